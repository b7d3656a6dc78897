/* pclean_detmath.h — the numeric contract of the PClean-MI355X boundary.
 *
 * The reference (probcomp/PClean, Julia) computes every score in Float64 and
 * draws with `rand(Categorical(exp.(lw .- logsumexp(lw))))`
 * (src/inference/proposal_compiler.jl:115-127,233-245, row_inference.jl:159-165,
 * src/utils.jl:16-26).  Julia's libm and its global RNG cannot be reproduced on
 * a GPU, so this header pins down ONE deterministic restatement that both the
 * HIP kernels (pclean_amd/csrc) and the CPU oracle (oracle/) compile:
 *
 *   pclean_exp / pclean_log : fdlibm-style double exp/log written with plain
 *       IEEE +,-,*,/ only (compile with -ffp-contract=off on BOTH sides so no
 *       fma is formed); < 1 ulp from libm, bit-identical host/device.
 *   PCLEAN_FIX_BITS fixed-point weights : a log-sum-exp or a categorical draw
 *       over scores s_k first takes m = max s_k (order independent), then
 *       u_k = floor(exp(s_k - m) * 2^40) as uint64.  Sums of u_k are integer,
 *       hence independent of reduction order / wavefront shape / GPU count.
 *         lse  = m + log((double)U * 2^-40),  U = sum u_k
 *         draw = min{k : u_0+..+u_k > r},     r = mulhi64(philox64, U)
 *       Quantisation error of the lse is <= n * 2^-40 relative (n candidates).
 *
 * Nothing here is PClean source; exp/log follow the published fdlibm
 * (e_exp.c / e_log.c, Sun Microsystems 1993, freely redistributable) method.
 */
#ifndef PCLEAN_DETMATH_H
#define PCLEAN_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PCLEAN_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define PCLEAN_HD static inline
#endif

#define PCLEAN_FIX_BITS 40
#define PCLEAN_FIX_ONE (1ull << PCLEAN_FIX_BITS)
#define PCLEAN_NEG_INF (-__builtin_inf())

PCLEAN_HD double pclean_bits2d(uint64_t b) {
  double d;
  __builtin_memcpy(&d, &b, 8);
  return d;
}
PCLEAN_HD uint64_t pclean_d2bits(double d) {
  uint64_t b;
  __builtin_memcpy(&b, &d, 8);
  return b;
}

/* exp(x), any finite x, -inf -> 0, +inf -> inf, NaN -> NaN. */
PCLEAN_HD double pclean_exp(double x) {
  const double ln2hi = 6.93147180369123816490e-01; /* 0x3fe62e42fee00000 */
  const double ln2lo = 1.90821492927058770002e-10; /* 0x3dea39ef35793c76 */
  const double invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.782712893383973096) return __builtin_inf();
  if (x < -745.13321910194110842) return 0.0;
  double hi, lo, r;
  int k;
  double ax = x < 0 ? -x : x;
  if (ax > 0.34657359027997264) { /* |x| > ln2/2 */
    k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
    hi = x - (double)k * ln2hi;
    lo = (double)k * ln2lo;
    r = hi - lo;
  } else if (ax < 3.7252902984619141e-09) { /* |x| < 2^-28 */
    return 1.0 + x;
  } else {
    k = 0;
    hi = x;
    lo = 0.0;
    r = x;
  }
  double t = r * r;
  double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  double y;
  if (k == 0) return 1.0 - ((r * c) / (c - 2.0) - r);
  y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  /* scale by 2^k without libm */
  if (k >= -1021) {
    uint64_t b = pclean_d2bits(y);
    b += (uint64_t)((int64_t)k << 52);
    return pclean_bits2d(b);
  } else {
    uint64_t b = pclean_d2bits(y);
    b += (uint64_t)((int64_t)(k + 1000) << 52);
    return pclean_bits2d(b) * 9.33263618503218878990e-302; /* 2^-1000 */
  }
}

/* log(x) for x > 0 (normal or subnormal); x == 0 -> -inf; x < 0 / NaN -> NaN;
 * +inf -> +inf. */
PCLEAN_HD double pclean_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  if (x != x || x < 0) return __builtin_nan("");
  if (x == 0) return PCLEAN_NEG_INF;
  if (x == __builtin_inf()) return x;
  int k = 0;
  uint64_t b = pclean_d2bits(x);
  if ((b >> 52) == 0) { /* subnormal: scale up by 2^54 */
    x *= 18014398509481984.0;
    k -= 54;
    b = pclean_d2bits(x);
  }
  int e = (int)(b >> 52) - 1023;
  uint64_t mant = b & 0x000fffffffffffffull;
  /* normalise to [sqrt(2)/2, sqrt(2)) */
  if (mant >= 0x6a09e667f3bcdull) { /* mantissa of sqrt(2) */
    e += 1;
    b = mant | ((uint64_t)1022 << 52);
  } else {
    b = mant | ((uint64_t)1023 << 52);
  }
  k += e;
  double f = pclean_bits2d(b) - 1.0;
  double dk = (double)k;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* Fixed-point weight of a score relative to the maximum: floor(exp(d)*2^40),
 * d = s - m <= 0.  -inf (or NaN) -> 0. */
PCLEAN_HD uint64_t pclean_fixw(double d) {
  if (!(d >= -28.5)) return 0; /* exp(-28.5)*2^40 < 1; also catches NaN/-inf */
  if (d >= 0) return PCLEAN_FIX_ONE;
  return (uint64_t)(pclean_exp(d) * 1099511627776.0);
}

/* lse from (m, U): m + log(U * 2^-40).  U == 0 only when every score is -inf. */
PCLEAN_HD double pclean_lse_from_fix(double m, uint64_t U) {
  if (U == 0) return PCLEAN_NEG_INF;
  return m + pclean_log((double)U * 9.094947017729282379150390625e-13);
}

/* floor(R * U / 2^64): a uniform integer in [0, U) from 64 random bits. */
PCLEAN_HD uint64_t pclean_mulhi64(uint64_t a, uint64_t b) {
  const uint64_t a0 = a & 0xffffffffull, a1 = a >> 32, b0 = b & 0xffffffffull, b1 = b >> 32;
  const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  const uint64_t mid = (p00 >> 32) + (p01 & 0xffffffffull) + (p10 & 0xffffffffull);
  return p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}

/* 53-bit uniform double in [0,1) from 64 random bits. */
PCLEAN_HD double pclean_u01(uint64_t r) { return (double)(r >> 11) * 1.1102230246251565404e-16; }

#endif /* PCLEAN_DETMATH_H */
