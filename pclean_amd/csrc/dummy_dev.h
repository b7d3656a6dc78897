// Value sampled for a chosen ProposalDummyValue, scored inside the sweep (block_proposal.jl:58-60):
// random(StringPrior) draw (string_prior.jl:28-39, the draw order of random_kernels.hip) as pool symbols, and its
// Damerau-Levenshtein distance to an observed pool string in either flavour (add_typos.jl:56) — single-thread
// device code for a rare path: a particle draws the dummy with an observation below the node only when no
// proposal atom explains that observation.
#pragma once
#include "ctx.h"
#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"

#define DUMMY_MAX_LEN 255  // StringPrior max_len and observed string length both fit a byte-wide pair table

// fixed-point inverse CDF over 28 probabilities (random_kernels.hip: draw28)
__device__ inline int dummy_draw28(const double* p, uint64_t r64) {
  uint64_t w[28];
  uint64_t total = 0;
  for (int j = 0; j < 28; ++j) {
    w[j] = (uint64_t)(p[j] * 1099511627776.0);
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

// the string of (key, row): symbols into out[0 .. len), returns len (random_string_prior_at_kernel's draws, stream 0)
__device__ inline int dummy_draw_string(uint64_t key, uint32_t row, int min_len, int max_len, const double* init_p,
                                        const double* trans_p, const uint16_t* letter_sym, uint16_t* out) {
  uint32_t t = 0;
  const uint32_t site = PCLEAN_SITE_RANDOM(PCLEAN_RANDOM_STRING_PRIOR);
  const int len = min_len + (int)pclean_mulhi64(pclean_rand64(key, row, site, t++, 0u), (uint64_t)(max_len - min_len + 1));
  int prev = 0;
  for (int k = 0; k < len; ++k) {
    prev = dummy_draw28(k == 0 ? init_p : trans_p + (size_t)prev * 28, pclean_rand64(key, row, site, t++, 0u));
    out[k] = letter_sym[prev];
  }
  return len;
}

// distance between a[0..la) and b[0..lb) on the full matrix H [(la + 2)][(lb + 2)] (int16, caller's scratch):
// restricted (optimal string alignment) or unrestricted (Lowrance-Wagner; the "last row where this symbol occurred"
// table is replaced by a backward search, the strings are short)
__device__ inline int dummy_distance(int dist_mode, const uint16_t* a, int la, const uint16_t* b, int lb, int16_t* H) {
  const int W = lb + 2;
  const int maxdist = la + lb;
  H[0] = (int16_t)maxdist;
  for (int i = 0; i <= la; ++i) {
    H[(i + 1) * W + 0] = (int16_t)maxdist;
    H[(i + 1) * W + 1] = (int16_t)i;
  }
  for (int j = 0; j <= lb; ++j) {
    H[0 * W + (j + 1)] = (int16_t)maxdist;
    H[1 * W + (j + 1)] = (int16_t)j;
  }
  for (int i = 1; i <= la; ++i) {
    int db = 0;
    for (int j = 1; j <= lb; ++j) {
      const int cost = a[i - 1] == b[j - 1] ? 0 : 1;
      int v = H[i * W + j] + cost;
      v = min(v, H[(i + 1) * W + j] + 1);
      v = min(v, H[i * W + (j + 1)] + 1);
      if (dist_mode == PCLEAN_DIST_OSA) {
        if (i > 1 && j > 1 && a[i - 1] == b[j - 2] && a[i - 2] == b[j - 1]) v = min(v, H[(i - 1) * W + (j - 1)] + 1);
      } else {
        int k = 0;  // last row < i whose symbol equals b[j-1]
        for (int q = i - 1; q >= 1; --q)
          if (a[q - 1] == b[j - 1]) {
            k = q;
            break;
          }
        const int l = db;
        v = min(v, H[k * W + l] + (i - k - 1) + 1 + (j - l - 1));
        if (cost == 0) db = j;
      }
      H[(i + 1) * W + (j + 1)] = (int16_t)v;
    }
  }
  return H[(la + 1) * W + (lb + 1)];
}
