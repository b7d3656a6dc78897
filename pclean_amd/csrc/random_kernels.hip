// random(dist, args...) of the noise models (src/distributions/*.jl `random` methods), batched:
// one lane per element, counter-based Philox draws, so the output depends only on
// (seed, stream, element index) — never on the launch geometry or the GPU count.
// The draw order is fixed by include/pclean_hip.h ("random(dist, args...)" section).
#include "ctx.h"
#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"

namespace {

struct Draws {  // the element's private draw stream
  uint64_t seed;
  uint32_t elem, site, stream, t;
  __device__ uint64_t next() { return pclean_rand64(seed, elem, site, t++, stream); }
  __device__ uint32_t below(uint32_t n) { return (uint32_t)pclean_mulhi64(next(), (uint64_t)n); }
};

// add_typos.jl:9-45
__global__ void random_add_typos_kernel(int n, const uint32_t* __restrict__ cp, const int64_t* __restrict__ off,
                                        int max_typos, uint64_t seed, uint32_t stream, int stride,
                                        uint32_t* __restrict__ out_cp, int32_t* __restrict__ out_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draws d{seed, (uint32_t)i, PCLEAN_SITE_RANDOM(PCLEAN_RANDOM_ADD_TYPOS), stream, 0};
  uint32_t* w = out_cp + (size_t)i * stride;
  int len = (int)(off[i + 1] - off[i]);
  if (len > stride) len = stride;
  for (int k = 0; k < len; ++k) w[k] = cp[off[i] + k];
  // NegativeBinomial(ceil(len/5), 0.9): failures before the r-th success (add_typos.jl:37)
  const int r = (len + 4) / 5;
  int typos = 0;
  for (int succ = 0; succ < r;) {
    if (d.next() < PCLEAN_P10_U64)
      ++typos;
    else
      ++succ;
  }
  if (max_typos >= 0 && typos > max_typos) typos = max_typos;  // add_typos.jl:38
  for (int k = 0; k < typos; ++k) {
    const uint32_t kind = d.below(4);  // [:insert, :delete, :transpose, :substitute], add_typos.jl:41
    if (kind == 0) {
      if (len >= stride) continue;
      const int pos = (int)d.below((uint32_t)len + 1);
      const uint32_t letter = 'a' + d.below(26);
      for (int q = len; q > pos; --q) w[q] = w[q - 1];
      w[pos] = letter;
      ++len;
    } else if (kind == 1) {
      if (len < 1) continue;
      const int pos = (int)d.below((uint32_t)len);
      for (int q = pos; q + 1 < len; ++q) w[q] = w[q + 1];
      --len;
    } else if (kind == 2) {
      if (len < 2) continue;
      const int pos = (int)d.below((uint32_t)len - 1);
      const uint32_t tmp = w[pos];
      w[pos] = w[pos + 1];
      w[pos + 1] = tmp;
    } else {
      if (len < 1) continue;
      const int pos = (int)d.below((uint32_t)len);
      w[pos] = 'a' + d.below(26);
    }
  }
  out_len[i] = len;
}

// fixed-point inverse CDF over 28 probabilities
__device__ int draw28(const double* p, uint64_t r64) {
  uint64_t w[28];
  uint64_t total = 0;
  for (int j = 0; j < 28; ++j) {
    w[j] = (uint64_t)(p[j] * 1099511627776.0);  // floor(p * 2^40), p >= 0
    total += w[j];
  }
  const uint64_t r = pclean_mulhi64(r64, total);
  uint64_t acc = 0;
  for (int j = 0; j < 28; ++j) {
    acc += w[j];
    if (acc > r) return j;
  }
  return 27;
}

// string_prior.jl:28-39
__global__ void random_string_prior_kernel(int n, int min_len, int max_len, const double* __restrict__ init_p,
                                           const double* __restrict__ trans_p, uint64_t seed, uint32_t stream,
                                           int stride, uint8_t* __restrict__ out, int32_t* __restrict__ out_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draws d{seed, (uint32_t)i, PCLEAN_SITE_RANDOM(PCLEAN_RANDOM_STRING_PRIOR), stream, 0};
  const int len = min_len + (int)d.below((uint32_t)(max_len - min_len + 1));
  int prev = 0;
  for (int k = 0; k < len; ++k) {
    prev = draw28(k == 0 ? init_p : trans_p + (size_t)prev * 28, d.next());
    out[(size_t)i * stride + k] = (uint8_t)prev;
  }
  out_len[i] = len;
}

// the same with a private (key, counter row) per element (values of chosen ProposalDummyValues, pclean_dummy_seed)
__global__ void random_string_prior_at_kernel(int n, const uint64_t* __restrict__ seeds, const uint32_t* __restrict__ elems,
                                              int min_len, int max_len, const double* __restrict__ init_p,
                                              const double* __restrict__ trans_p, uint32_t stream, int stride,
                                              uint8_t* __restrict__ out, int32_t* __restrict__ out_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draws d{seeds[i], elems[i], PCLEAN_SITE_RANDOM(PCLEAN_RANDOM_STRING_PRIOR), stream, 0};
  const int len = min_len + (int)d.below((uint32_t)(max_len - min_len + 1));
  int prev = 0;
  for (int k = 0; k < len; ++k) {
    prev = draw28(k == 0 ? init_p : trans_p + (size_t)prev * 28, d.next());
    out[(size_t)i * stride + k] = (uint8_t)prev;
  }
  out_len[i] = len;
}

// choose_proportionally.jl:3-5 / choose_uniformly.jl:3-5
__global__ void random_categorical_kernel(int n, int n_options, const double* __restrict__ logp, uint64_t seed,
                                          uint32_t stream, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draws d{seed, (uint32_t)i, PCLEAN_SITE_RANDOM(PCLEAN_RANDOM_CATEGORICAL), stream, 0};
  double m = -INFINITY;
  for (int k = 0; k < n_options; ++k) m = logp[k] > m ? logp[k] : m;
  uint64_t total = 0;
  for (int k = 0; k < n_options; ++k) total += pclean_fixw(logp[k] - m);
  const uint64_t r = pclean_mulhi64(d.next(), total);
  uint64_t acc = 0;
  int pick = n_options - 1;
  for (int k = 0; k < n_options; ++k) {
    acc += pclean_fixw(logp[k] - m);
    if (acc > r) {
      pick = k;
      break;
    }
  }
  out[i] = pick;
}

__device__ double unit_pm1(uint64_t r) { return 2.0 * (((double)(r >> 11) + 0.5) * 0x1.0p-53) - 1.0; }

// add_noise.jl:5, transformed_gaussian.jl:13
__global__ void random_normal_kernel(int n, const double* __restrict__ mean, double std, double fwd_scale,
                                     uint64_t seed, uint32_t stream, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draws d{seed, (uint32_t)i, PCLEAN_SITE_RANDOM(PCLEAN_RANDOM_NORMAL), stream, 0};
  double z = 0.0;
  for (;;) {
    const double u = unit_pm1(d.next());
    const double v = unit_pm1(d.next());
    const double s = u * u + v * v;
    if (s > 0.0 && s < 1.0) {
      z = u * sqrt(-2.0 * pclean_log(s) / s);
      break;
    }
  }
  out[i] = fwd_scale * (mean[i] + std * z);
}

// maybe_swap.jl:5-11
__global__ void random_maybe_swap_kernel(int n, const double* __restrict__ prob, const int32_t* __restrict__ n_options,
                                         uint64_t seed, uint32_t stream, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draws d{seed, (uint32_t)i, PCLEAN_SITE_RANDOM(PCLEAN_RANDOM_MAYBE_SWAP), stream, 0};
  const double p = prob[i];
  const uint64_t r0 = d.next();
  const uint64_t r1 = d.next();
  const bool swap = p >= 1.0 || (p > 0.0 && r0 < (uint64_t)(p * 18446744073709551616.0));
  out[i] = swap ? (int32_t)pclean_mulhi64(r1, (uint64_t)n_options[i]) : -1;
}

// time_prior.jl:21-23
__global__ void random_time_prior_kernel(int n, uint64_t seed, uint32_t stream, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draws d{seed, (uint32_t)i, PCLEAN_SITE_RANDOM(PCLEAN_RANDOM_TIME_PRIOR), stream, 0};
  out[3 * i] = 1 + (int32_t)d.below(12);
  out[3 * i + 1] = 1 + (int32_t)d.below(60);
  out[3 * i + 2] = (int32_t)(d.next() >> 63);
}

// Host staging: copy inputs up, run, copy outputs down.  These samplers are not on the sweep's
// critical path (they run when a proposal picks a dummy value or a node is unobserved), so they
// use plain synchronous transfers.
template <typename T>
int up(pclean_ctx* ctx, DevBuf<T>& b, const T* src, size_t n) {
  if (b.alloc(n ? n : 1)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (n) HIPCHK(ctx, hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice));
  return PCLEAN_OK;
}
template <typename T>
int down(pclean_ctx* ctx, T* dst, const DevBuf<T>& b, size_t n) {
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipGetLastError());
  if (n) HIPCHK(ctx, hipMemcpy(dst, b.p, n * sizeof(T), hipMemcpyDeviceToHost));
  return PCLEAN_OK;
}
inline dim3 grid_for(int n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

#define TRY(x)           \
  do {                   \
    int rc_ = (x);       \
    if (rc_) return rc_; \
  } while (0)

extern "C" int pclean_random_add_typos(pclean_ctx* ctx, int32_t n, const uint32_t* cp, const int64_t* off,
                                       int32_t max_typos, uint64_t seed, uint32_t stream, int32_t out_stride,
                                       uint32_t* out_cp, int32_t* out_len) {
  if (!ctx || n < 0 || !off || out_stride <= 0 || (n > 0 && (!out_cp || !out_len)))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_random_add_typos: bad arguments");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<uint32_t> d_cp, d_out;
  DevBuf<int64_t> d_off;
  DevBuf<int32_t> d_len;
  TRY(up(ctx, d_cp, cp, (size_t)off[n]));
  TRY(up(ctx, d_off, off, (size_t)n + 1));
  if (d_out.alloc((size_t)n * out_stride) || d_len.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemsetAsync(d_out.p, 0, (size_t)n * out_stride * 4, ctx->stream));
  hipLaunchKernelGGL(random_add_typos_kernel, grid_for(n), dim3(256), 0, ctx->stream, n, d_cp.p, d_off.p, max_typos, seed,
                     stream, out_stride, d_out.p, d_len.p);
  TRY(down(ctx, out_cp, d_out, (size_t)n * out_stride));
  TRY(down(ctx, out_len, d_len, (size_t)n));
  d_cp.release(); d_out.release(); d_off.release(); d_len.release();
  return PCLEAN_OK;
}

extern "C" int pclean_random_string_prior(pclean_ctx* ctx, int32_t n, int32_t min_len, int32_t max_len,
                                          const double* init_p, const double* trans_p, uint64_t seed, uint32_t stream,
                                          int32_t out_stride, uint8_t* out_letters, int32_t* out_len) {
  if (!ctx || n < 0 || min_len < 0 || max_len < min_len || out_stride < max_len || out_stride <= 0 || !init_p || !trans_p ||
      (n > 0 && (!out_letters || !out_len)))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_random_string_prior: bad arguments");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<double> d_init, d_trans;
  DevBuf<uint8_t> d_out;
  DevBuf<int32_t> d_len;
  TRY(up(ctx, d_init, init_p, 28));
  TRY(up(ctx, d_trans, trans_p, 28 * 28));
  if (d_out.alloc((size_t)n * out_stride) || d_len.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemsetAsync(d_out.p, 0, (size_t)n * out_stride, ctx->stream));
  hipLaunchKernelGGL(random_string_prior_kernel, grid_for(n), dim3(256), 0, ctx->stream, n, min_len, max_len, d_init.p,
                     d_trans.p, seed, stream, out_stride, d_out.p, d_len.p);
  TRY(down(ctx, out_letters, d_out, (size_t)n * out_stride));
  TRY(down(ctx, out_len, d_len, (size_t)n));
  d_init.release(); d_trans.release(); d_out.release(); d_len.release();
  return PCLEAN_OK;
}

extern "C" int pclean_random_string_prior_at(pclean_ctx* ctx, int32_t n, const uint64_t* seeds, const uint32_t* elems,
                                             int32_t min_len, int32_t max_len, const double* init_p, const double* trans_p,
                                             uint32_t stream, int32_t out_stride, uint8_t* out_letters, int32_t* out_len) {
  if (!ctx || n < 0 || min_len < 0 || max_len < min_len || out_stride < max_len || out_stride <= 0 || !init_p || !trans_p ||
      (n > 0 && (!out_letters || !out_len || !seeds || !elems)))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_random_string_prior_at: bad arguments");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<double> d_init, d_trans;
  DevBuf<uint64_t> d_seeds;
  DevBuf<uint32_t> d_elems;
  DevBuf<uint8_t> d_out;
  DevBuf<int32_t> d_len;
  TRY(up(ctx, d_init, init_p, 28));
  TRY(up(ctx, d_trans, trans_p, 28 * 28));
  TRY(up(ctx, d_seeds, seeds, (size_t)n));
  TRY(up(ctx, d_elems, elems, (size_t)n));
  if (d_out.alloc((size_t)n * out_stride) || d_len.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemsetAsync(d_out.p, 0, (size_t)n * out_stride, ctx->stream));
  hipLaunchKernelGGL(random_string_prior_at_kernel, grid_for(n), dim3(256), 0, ctx->stream, n, d_seeds.p, d_elems.p, min_len,
                     max_len, d_init.p, d_trans.p, stream, out_stride, d_out.p, d_len.p);
  TRY(down(ctx, out_letters, d_out, (size_t)n * out_stride));
  TRY(down(ctx, out_len, d_len, (size_t)n));
  d_init.release(); d_trans.release(); d_seeds.release(); d_elems.release(); d_out.release(); d_len.release();
  return PCLEAN_OK;
}

extern "C" int pclean_random_categorical(pclean_ctx* ctx, int32_t n, int32_t n_options, const double* logp,
                                         uint64_t seed, uint32_t stream, int32_t* out) {
  if (!ctx || n < 0 || n_options <= 0 || !logp || (n > 0 && !out))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_random_categorical: bad arguments");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<double> d_logp;
  DevBuf<int32_t> d_out;
  TRY(up(ctx, d_logp, logp, (size_t)n_options));
  if (d_out.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  hipLaunchKernelGGL(random_categorical_kernel, grid_for(n), dim3(256), 0, ctx->stream, n, n_options, d_logp.p, seed,
                     stream, d_out.p);
  TRY(down(ctx, out, d_out, (size_t)n));
  d_logp.release(); d_out.release();
  return PCLEAN_OK;
}

extern "C" int pclean_random_normal(pclean_ctx* ctx, int32_t n, const double* mean, double std, double fwd_scale,
                                    uint64_t seed, uint32_t stream, double* out) {
  if (!ctx || n < 0 || !(std >= 0.0) || (n > 0 && (!mean || !out)))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_random_normal: bad arguments");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<double> d_mean, d_out;
  TRY(up(ctx, d_mean, mean, (size_t)n));
  if (d_out.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  hipLaunchKernelGGL(random_normal_kernel, grid_for(n), dim3(256), 0, ctx->stream, n, d_mean.p, std, fwd_scale, seed, stream,
                     d_out.p);
  TRY(down(ctx, out, d_out, (size_t)n));
  d_mean.release(); d_out.release();
  return PCLEAN_OK;
}

extern "C" int pclean_random_maybe_swap(pclean_ctx* ctx, int32_t n, const double* prob, const int32_t* n_options,
                                        uint64_t seed, uint32_t stream, int32_t* out) {
  if (!ctx || n < 0 || (n > 0 && (!prob || !n_options || !out)))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_random_maybe_swap: bad arguments");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<double> d_prob;
  DevBuf<int32_t> d_n, d_out;
  TRY(up(ctx, d_prob, prob, (size_t)n));
  TRY(up(ctx, d_n, n_options, (size_t)n));
  if (d_out.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  hipLaunchKernelGGL(random_maybe_swap_kernel, grid_for(n), dim3(256), 0, ctx->stream, n, d_prob.p, d_n.p, seed, stream,
                     d_out.p);
  TRY(down(ctx, out, d_out, (size_t)n));
  d_prob.release(); d_n.release(); d_out.release();
  return PCLEAN_OK;
}

extern "C" int pclean_random_time_prior(pclean_ctx* ctx, int32_t n, uint64_t seed, uint32_t stream, int32_t* out) {
  if (!ctx || n < 0 || (n > 0 && !out)) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_random_time_prior: bad arguments");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<int32_t> d_out;
  if (d_out.alloc((size_t)n * 3)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  hipLaunchKernelGGL(random_time_prior_kernel, grid_for(n), dim3(256), 0, ctx->stream, n, seed, stream, d_out.p);
  TRY(down(ctx, out, d_out, (size_t)n * 3));
  d_out.release();
  return PCLEAN_OK;
}
