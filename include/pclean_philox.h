/* pclean_philox.h — counter-based RNG shared by the HIP kernels and the oracle.
 *
 * The reference draws from Julia's global, unseeded RNG
 * (src/inference/row_inference.jl:99,162,164; proposal_compiler.jl:119,237;
 * block_proposal.jl:47,80) so its stream cannot be reproduced.  The build
 * replaces it by Philox4x32-10 (Salmon et al., SC'11 — published algorithm,
 * restated here) keyed by the user seed; the 128-bit counter names the draw
 * site, so any draw can be regenerated in any order on any number of GPUs:
 *
 *     counter = (row, site, particle, sweep)     key = (seed_lo, seed_hi)
 *
 * `site` values are listed in pclean_hip.h (PCLEAN_SITE_*).
 */
#ifndef PCLEAN_PHILOX_H
#define PCLEAN_PHILOX_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PCLEAN_RNG_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define PCLEAN_RNG_HD static inline
#endif

typedef struct {
  uint32_t v[4];
} pclean_u32x4;

PCLEAN_RNG_HD void pclean_philox_round(uint32_t* c, uint32_t k0, uint32_t k1) {
  const uint64_t M0 = 0xD2511F53ull, M1 = 0xCD9E8D57ull;
  uint64_t p0 = M0 * (uint64_t)c[0];
  uint64_t p1 = M1 * (uint64_t)c[2];
  uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  uint32_t n0 = hi1 ^ c[1] ^ k0;
  uint32_t n1 = lo1;
  uint32_t n2 = hi0 ^ c[3] ^ k1;
  uint32_t n3 = lo0;
  c[0] = n0;
  c[1] = n1;
  c[2] = n2;
  c[3] = n3;
}

PCLEAN_RNG_HD pclean_u32x4 pclean_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
  uint32_t c[4] = {c0, c1, c2, c3};
  for (int i = 0; i < 10; ++i) {
    pclean_philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  pclean_u32x4 o;
  o.v[0] = c[0];
  o.v[1] = c[1];
  o.v[2] = c[2];
  o.v[3] = c[3];
  return o;
}

/* 64 random bits for draw site (row, site, particle, sweep) under `seed`. */
PCLEAN_RNG_HD uint64_t pclean_rand64(uint64_t seed, uint32_t row, uint32_t site, uint32_t particle,
                                     uint32_t sweep) {
  pclean_u32x4 o =
      pclean_philox4x32_10(row, site, particle, sweep, (uint32_t)seed, (uint32_t)(seed >> 32));
  return ((uint64_t)o.v[1] << 32) | (uint64_t)o.v[0];
}

/* Key of the private draw stream of a value sampled for a chosen ProposalDummyValue (block_proposal.jl:58-60:
 * `random(node.dist, args...)`): one stream per (draw site of the enumerated node, particle, sweep); the row is the
 * counter's first word.  The sweep draws the string with this key when it corrects the particle's weight, the host
 * draws it again with the same key when the particle is chosen and its new row is committed. */
PCLEAN_RNG_HD uint64_t pclean_dummy_seed(uint64_t seed, uint32_t site, uint32_t particle, uint32_t sweep) {
  uint64_t x = seed ^ (((uint64_t)site << 32) | (uint64_t)sweep);
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x ^= (uint64_t)(particle + 1u) * 0x94d049bb133111ebull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

#endif /* PCLEAN_PHILOX_H */
