// fk_root_fast_kernel — the dominant sweep kernel: candidate referents of a reference slot with many
// candidates (hospital Record block 1: K ~ 1e4 latent hospitals x 11 AddTypos terms; the nested Place /
// County slots of its new-row branch; the Measure root of block 2 including its ctx term).
//
// Same contract and bit-identical results as enum_node_kernel (enum_kernels.hip), restructured for
// MI355X around three facts:
//   * candidate-compact byte tables comp_f[o][k] (built by compact_pair_kernel when the latent table's
//     columns change) turn the pair-table gather into contiguous byte streams: every lane reads one
//     dword = 4 candidates, a wave reads 256 consecutive bytes -> fully coalesced, L2 resident for the
//     rows of one referent (groups are launched sorted by referent);
//   * a candidate whose score is more than 28.5 nats below the maximum has fixed-point weight
//     floor(exp(s-m) 2^40) == 0 exactly (pclean_fixw), so it can influence neither the log-sum-exp nor
//     a draw.  An INTEGER PRE-FILTER proves that for almost every candidate: score <= prior_max -
//     c_min * (summed byte distances of the three most discriminating terms), compared with a lower
//     bound of the maximum (the exact score of the rows' current referent, or of the candidate with the
//     smallest summed distance).  Only the survivors (a handful per row) are scored in fp64, in plan
//     order, with the density read from the global nb / logl tables;
//   * rows with identical (observed tuple, ctx, referent) share the score vector: ONE WORKGROUP PER
//     GROUP of such rows (ItemsDev::grp_off / members); every (member row, particle) pair then draws
//     with its own Philox counter by binary search over the survivors' prefix.
// Survivors are kept in a 4-entry register window per lane, filtered with the block maximum into a
// small LDS list (SURV2_CAP), rank-sorted by candidate index (the natural order of the inverse CDF)
// and prefix-summed.  Groups whose windows or list overflow (flat posteriors, < 1 % of the rows) are
// flagged and re-run by the host with the LDS-resident generic kernel — results are identical either way.
#include <algorithm>
#include <cstdlib>

#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"
#include "enum.h"

#define HALF_LOG26 1.629048269010741
#define ADD_TYPOS_IMPOSSIBLE (-1e5)
#define FAST_MAX_WAVES 16
#define SURV2_CAP 256      // after filtering with the true maximum
#define FIX_CUTOFF 28.5    // pclean_fixw(d) == 0 for d < -28.5
#define FAST_WINDOW 8      // per-lane register window of candidates within FIX_CUTOFF of the running maximum

__global__ void compact_pair_kernel(const uint8_t* __restrict__ pair, int n_obs, int n_lat,
                                    const int32_t* __restrict__ cand_col, int n_cand, int kpad,
                                    uint8_t* __restrict__ comp) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (k >= kpad) return;
  uint8_t v = 0;
  if (k < n_cand) v = pair[(size_t)o * n_lat + cand_col[k]];
  comp[(size_t)o * kpad + k] = v;
}
__global__ void compact_len_kernel(const uint16_t* __restrict__ lat_len, const int32_t* __restrict__ cand_col,
                                   int n_cand, int kpad, uint8_t* __restrict__ clen) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= kpad) return;
  clen[k] = k < n_cand ? (uint8_t)lat_len[cand_col[k]] : (uint8_t)0;
}
__global__ void priors_kernel(const int64_t* __restrict__ counts, const double* __restrict__ logc_full, int n_cand,
                              int kpad, double logden_e, double logden_n, double* __restrict__ prior_e,
                              double* __restrict__ prior_n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= kpad) return;
  const bool live = k < n_cand && counts[k] != 0;
  prior_e[k] = live ? logc_full[k] - logden_e : -__builtin_inf();
  prior_n[k] = live ? logc_full[k] - logden_n : -__builtin_inf();
}

int pclean_build_compact(pclean_ctx* ctx, const uint8_t* pair, int n_obs, int n_lat, const int32_t* cand_col,
                         const uint16_t* lat_len, int n_cand, int kpad, uint8_t* comp, uint8_t* clen) {
  if (n_obs > 0)
    hipLaunchKernelGGL(compact_pair_kernel, dim3((kpad + 255) / 256, n_obs), dim3(256), 0, ctx->stream, pair, n_obs,
                       n_lat, cand_col, n_cand, kpad, comp);
  hipLaunchKernelGGL(compact_len_kernel, dim3((kpad + 255) / 256), dim3(256), 0, ctx->stream, lat_len, cand_col, n_cand,
                     kpad, clen);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
int pclean_build_priors(pclean_ctx* ctx, const int64_t* counts, const double* logc_full, int n_cand, int kpad,
                        double logden_e, double logden_n, double* prior_e, double* prior_n) {
  hipLaunchKernelGGL(priors_kernel, dim3((kpad + 255) / 256), dim3(256), 0, ctx->stream, counts, logc_full, n_cand,
                     kpad, logden_e, logden_n, prior_e, prior_n);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

__device__ __forceinline__ double wave_max64(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ double add_typos_dens(const DensDev& dn, int L, int d) {
  // the fp64 operation order of term_density() (enum_kernels.hip), add_typos.jl:61-63
  const int r = (L + 4) / 5;
  double l = dn.nb[(size_t)r * dn.nb_stride + d];
  l -= dn.logl[L] * (double)d;
  l -= HALF_LOG26 * (double)d;
  return l;
}

template <int NT>
__global__ __launch_bounds__(1024) void fk_root_fast_kernel(const FastRootDev fr, const DensDev dn, const ItemsDev it,
                                                            const ChildrenDev ch, uint64_t seed, uint32_t sweep,
                                                            uint32_t site, int n_draws, int item_base,
                                                            double* __restrict__ lse_out,
                                                            int32_t* __restrict__ draws_out,
                                                            int32_t* __restrict__ overflow_flag,
                                                            unsigned int* __restrict__ overflow_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T = blockDim.x, nw = T >> 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x + item_base;
  const int m_lo = it.grp_off ? it.grp_off[g] : g, m_hi = it.grp_off ? it.grp_off[g + 1] : g + 1;
  const int t = it.grp_off ? it.members[m_lo] : g;  // the item whose scores stand for the whole group
  const int n = fr.n_cand;
  // LDS carve (all offsets multiples of 16 bytes)
  double* red = (double*)smem;                                  // [16] per-wave maxima
  uint64_t* redk = (uint64_t*)(red + FAST_MAX_WAVES);           // [16] per-wave (distance, candidate) minima
  uint64_t* u2 = redk + FAST_MAX_WAVES;                         // [SURV2_CAP] sorted weights, then inclusive prefix
  double* sc2 = (double*)(u2 + SURV2_CAP);                      // [SURV2_CAP] second-stage scores
  int32_t* k2 = (int32_t*)(sc2 + SURV2_CAP);                    // [SURV2_CAP] survivor ids (unsorted)
  int32_t* ks = k2 + SURV2_CAP;                                 // [SURV2_CAP] sorted ids
  unsigned int* cnt = (unsigned int*)(ks + SURV2_CAP);          // [4] counters
  double* bnd = (double*)(cnt + 4);                             // [1] lower bound of the maximum

  if (tid < 4) cnt[tid] = 0;
  const int row = it.row ? it.row[t] : t;
  const int excl = it.excl ? it.excl[t] : -1;
  const bool excluded = excl >= 0;
  const bool deleted = excluded && fr.counts[excl] <= 1;
  const double logden = excluded ? fr.scal[1] : fr.scal[0];
  const double* prior = excluded ? fr.prior_e : fr.prior_n;
  const double pmax = excluded ? fr.prior_max_e : fr.prior_max_n;
  // per-term row pointers (wave-uniform -> scalar registers)
  const uint32_t* crow[NT];
  const uint32_t* lrow[NT];
  const uint8_t* grow[NT];   // ctx terms: the observed value's row of the full pair table
  const int32_t* gfn[NT];    //            fn[ctx value][.]
  int mt[NT];
#pragma unroll
  for (int f = 0; f < NT; ++f) {
    crow[f] = nullptr;
    lrow[f] = nullptr;
    grow[f] = nullptr;
    gfn[f] = nullptr;
    mt[f] = -1;
    if (f < fr.n_terms) {
      const int o = fr.terms[f].obs_col[row];
      mt[f] = fr.terms[f].max_typos;
      if (fr.terms[f].ctx_slot >= 0) {
        if (o >= 0) {
          grow[f] = fr.terms[f].pair + (size_t)o * fr.terms[f].n_lat;
          gfn[f] = fr.terms[f].fn + (size_t)it.ctx[(size_t)t * PCLEAN_MAX_CTX + fr.terms[f].ctx_slot] * fr.terms[f].fn_nb;
        }
      } else {
        if (o >= 0) crow[f] = reinterpret_cast<const uint32_t*>(fr.terms[f].comp + (size_t)o * fr.kpad);
        lrow[f] = reinterpret_cast<const uint32_t*>(fr.terms[f].clen);
      }
    }
  }
  // density of ctx term f for candidate k (term_density() of enum_kernels.hip on the mapped value)
  auto ctx_term = [&](int f, int k) {
    const int val = gfn[f][fr.terms[f].cand_col[k]];
    const int d = grow[f][val], L = fr.terms[f].lat_len[val];
    return (mt[f] >= 0 && d > mt[f]) ? ADD_TYPOS_IMPOSSIBLE : add_typos_dens(dn, L, d);
  };
  // the (up to 3) most discriminating terms: byte rows summed by the integer pre-filter
  const uint32_t* prow[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    prow[p] = nullptr;
    if (p < fr.n_pre) {
      const int f = fr.pre[p];
      const int o = fr.terms[f].obs_col[row];
      if (o >= 0) prow[p] = reinterpret_cast<const uint32_t*>(fr.terms[f].comp + (size_t)o * fr.kpad);
    }
  }
  // exact score of one candidate, in the operation order of the main loop
  auto exact_score = [&](int k, double pr) {
    double b = pr;
#pragma unroll
    for (int f = 0; f < NT; ++f) {
      if (crow[f]) {
        const int d = reinterpret_cast<const uint8_t*>(crow[f])[k], L = reinterpret_cast<const uint8_t*>(lrow[f])[k];
        b += (mt[f] >= 0 && d > mt[f]) ? ADD_TYPOS_IMPOSSIBLE : add_typos_dens(dn, L, d);
      } else if (grow[f]) {
        b += ctx_term(f, k);
      }
    }
    return b;
  };
  const int nslots = fr.kpad >> 2;

  // ---- phase 0: a lower bound of the maximum = the exact score of one good candidate.
  // With a retained referent (rejuvenation of an assigned row) that candidate is the referent;
  // otherwise (nested reference slots of a new row, initialisation) the candidate with the smallest
  // summed edit distance over the pre-filter terms.  Only ever used as a filter, never as a score.
  double bound = -__builtin_inf();
  if (excluded && !deleted) {
    bound = exact_score(excl, fr.logc_m1[excl] - logden) - 1.0;
  } else if (fr.n_pre > 0) {
    uint64_t best = ~0ull;
    for (int slot = tid; slot < nslots; slot += T) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (prow[p]) {
          const uint32_t c = prow[p][slot];
          lo += c & 0x00ff00ffu;
          hi += (c >> 8) & 0x00ff00ffu;
        }
      const uint32_t D[4] = {lo & 0xffffu, hi & 0xffffu, lo >> 16, hi >> 16};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = (slot << 2) + e;
        const uint64_t key = ((uint64_t)D[e] << 32) | (uint32_t)k;
        if (key < best && k < n && k != excl && prior[k] > -__builtin_inf()) best = key;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t other = __shfl_xor(best, o, 64);
      best = other < best ? other : best;
    }
    if (lane == 0) redk[wave] = best;
    __syncthreads();
    if (tid == 0) {
      uint64_t b = redk[0];
      for (int w = 1; w < nw; ++w) b = redk[w] < b ? redk[w] : b;
      double v = -__builtin_inf();
      if (b != ~0ull) {
        const int k = (int)(uint32_t)b;
        v = exact_score(k, prior[k]) - 1.0;
      }
      bnd[0] = v;
    }
    __syncthreads();
    bound = bnd[0];
  }
  // Pre-filter threshold: a candidate whose summed edit distance D over the pre-filter terms exceeds
  // dcut scores at most pmax - c_min * D < bound - FIX_CUTOFF <= max - FIX_CUTOFF, i.e. its fixed-point
  // weight is exactly 0 (c_min = smallest cost of one edit, fr.inv_c = 1 / c_min; terms not summed
  // and missing observations only lower the score further).
  uint32_t dcut = 0xffffu;
  if (fr.n_pre > 0 && bound > -__builtin_inf()) {
    const double x = (pmax - bound + FIX_CUTOFF) * fr.inv_c;
    if (x >= 0.0 && x < 60000.0) dcut = (uint32_t)x + 2u;
  }

  // ---- phase 1: scores, 4 consecutive candidates per lane per round.  Each lane keeps the
  // candidates within FIX_CUTOFF of its running maximum in a 4-entry register window.
  double tmax = bound;      // filter threshold base (lower bound of the maximum)
  double wS[FAST_WINDOW];
  int wK[FAST_WINDOW];
#pragma unroll
  for (int i = 0; i < FAST_WINDOW; ++i) {
    wS[i] = -__builtin_inf();
    wK[i] = -1;
  }
  bool lost = false;        // a window candidate had to be dropped
  double lost_max = -__builtin_inf();
  // A lane reads 16 consecutive candidates per round (one 16-byte load per pre-filter row = 4 slots of
  // 4 candidates).  Rounds are processed in chunks of at most 16 (fr.chunk_rounds; 4 survivor bits per
  // round in one 64-bit word): 1a scans a chunk, 1b scores its survivors.
  const int nquads = fr.kpad >> 4;
  const int rounds_total = (nquads + T - 1) / T;
  for (int rb = 0; rb < rounds_total; rb += fr.chunk_rounds) {
  // 1a: branch-free integer scan (loads of several rounds in flight): bit 4r+w of `alive` = slot w of the
  //     lane's (rb + r)-th quad holds a candidate that may carry weight
  uint64_t alive = 0;
  {
    const int r_end = min(fr.chunk_rounds, rounds_total - rb);
#pragma unroll 2
    for (int r = 0; r < r_end; ++r) {
      const int quad = tid + (rb + r) * T;
      if (quad >= nquads) break;
      uint32_t lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (prow[p]) {
          const uint4 c = reinterpret_cast<const uint4*>(prow[p])[quad];
          const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            lo[w] += cw[w] & 0x00ff00ffu;
            hi[w] += (cw[w] >> 8) & 0x00ff00ffu;
          }
        }
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const bool pass = (lo[w] & 0xffffu) <= dcut || (hi[w] & 0xffffu) <= dcut || (lo[w] >> 16) <= dcut ||
                          (hi[w] >> 16) <= dcut;
        alive |= (uint64_t)(pass ? 1u : 0u) << (4 * r + w);
      }
    }
  }
  // 1b: exact fp64 scores of the surviving slots
  while (alive) {
    const int bit = __builtin_ctzll(alive);
    alive &= alive - 1;
    const int slot = ((tid + (rb + (bit >> 2)) * T) << 2) + (bit & 3);
    const int k0 = slot << 2;
    double acc[4];
    const double2 p01 = *reinterpret_cast<const double2*>(prior + k0);
    const double2 p23 = *reinterpret_cast<const double2*>(prior + k0 + 2);
    acc[0] = p01.x;
    acc[1] = p01.y;
    acc[2] = p23.x;
    acc[3] = p23.y;
    if (excluded && (unsigned)(excl - k0) < 4u) {
      const double pe = deleted ? -__builtin_inf() : fr.logc_m1[excl] - logden;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e == excl) acc[e] = pe;
    }
#pragma unroll
    for (int f = 0; f < NT; ++f) {
      if (crow[f]) {  // an explicitly missing observation contributes nothing (add_typos.jl:51-53)
        const uint32_t c4 = crow[f][slot], l4 = lrow[f][slot];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int d = (c4 >> (8 * e)) & 255, L = (l4 >> (8 * e)) & 255;
          double dens = add_typos_dens(dn, L, d);
          if (mt[f] >= 0 && d > mt[f]) dens = ADD_TYPOS_IMPOSSIBLE;
          acc[e] += dens;
        }
      } else if (grow[f]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k0 + e < n) acc[e] += ctx_term(f, k0 + e);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (k0 + e < n) tmax = fmax(tmax, acc[e]);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (k0 + e < n && acc[e] >= tmax - FIX_CUTOFF && acc[e] > -__builtin_inf()) {
        // insert into the window: reuse a slot that is empty or has fallen out of the window
        bool placed = false;
#pragma unroll
        for (int i = 0; i < FAST_WINDOW; ++i)
          if (!placed && (wK[i] < 0 || wS[i] < tmax - FIX_CUTOFF)) {
            wS[i] = acc[e];
            wK[i] = k0 + e;
            placed = true;
          }
        if (!placed) {
          lost = true;
          lost_max = fmax(lost_max, acc[e]);
        }
      }
  }
  }  // chunk of rounds
  double sn = -__builtin_inf();
  if (tid == 0) {  // the "new row" candidate (index n, last in natural order)
    double snew = 0.0;
    for (int c = 0; c < ch.n; ++c) {
      size_t idx = (size_t)t;
      if (ch.obs_col[c]) {
        const int o = ch.obs_col[c][row];
        idx = o < 0 ? (size_t)ch.n_obs[c] : (size_t)o;
      }
      snew += ch.arr[c][idx];
    }
    sn = ((deleted ? fr.scal[3] : fr.scal[2]) - logden) + snew;
    tmax = fmax(tmax, sn);
  }
  // ---- phase 2: block max -----------------------------------------------------------
  {
    const double wm = wave_max64(tmax);
    if (lane == 0) red[wave] = wm;
  }
  __syncthreads();
  double m = red[0];
  for (int w = 1; w < nw; ++w) m = fmax(m, red[w]);
  // ---- phase 3: candidates with non-zero fixed-point weight -> LDS list ----------------------
  if (lost && lost_max - m >= -FIX_CUTOFF) atomicAdd(&cnt[2], 1u);  // a dropped candidate mattered
#pragma unroll
  for (int i = 0; i < FAST_WINDOW; ++i)
    if (wK[i] >= 0 && wS[i] - m >= -FIX_CUTOFF) {
      const unsigned int pos = atomicAdd(&cnt[1], 1u);
      if (pos < SURV2_CAP) {
        sc2[pos] = wS[i];
        k2[pos] = wK[i];
      }
    }
  if (tid == 0 && sn - m >= -FIX_CUTOFF) {
    const unsigned int pos = atomicAdd(&cnt[1], 1u);
    if (pos < SURV2_CAP) {
      sc2[pos] = sn;
      k2[pos] = n;
    }
  }
  __syncthreads();
  const unsigned int n2 = cnt[1];
  if (n2 > SURV2_CAP || cnt[2] != 0) {  // flat posterior: the host re-runs these items with the generic kernel
    for (int mi = m_lo + tid; mi < m_hi; mi += T)
      overflow_flag[it.grp_off ? it.members[mi] : t] = PCLEAN_CHOICE_NEW;  // marker understood by compact_new_kernel
    if (tid == 0) atomicAdd(overflow_count, (unsigned int)(m_hi - m_lo));
    return;
  }
  for (int mi = m_lo + tid; mi < m_hi; mi += T) overflow_flag[it.grp_off ? it.members[mi] : t] = 0;
  // ---- phase 4: rank sort by candidate index (natural order), weights ------------------------
  for (unsigned int i = tid; i < n2; i += T) {
    const int ki = k2[i];
    unsigned int rank = 0;
    for (unsigned int j = 0; j < n2; ++j) rank += k2[j] < ki ? 1u : 0u;
    ks[rank] = ki;
    u2[rank] = pclean_fixw(sc2[i] - m);
  }
  __syncthreads();
  // inclusive prefix (n2 is small: one wave, chunked)
  if (wave == 0) {
    const unsigned int chunk = (n2 + 63) / 64;
    const unsigned int lo = min(lane * chunk, n2), hi = min(lo + chunk, n2);
    unsigned long long part = 0;
    for (unsigned int i = lo; i < hi; ++i) part += u2[i];
    unsigned long long incl = part;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long x = __shfl_up(incl, o, 64);
      if (lane >= o) incl += x;
    }
    unsigned long long run = incl - part;
    for (unsigned int i = lo; i < hi; ++i) {
      run += u2[i];
      u2[i] = run;
    }
  }
  __syncthreads();
  const uint64_t U = n2 ? u2[n2 - 1] : 0ull;
  // ---- phase 5: lse + draws ---------------------------------------------------------------
  const double lse = pclean_lse_from_fix(m, U);
  const int n_out = (m_hi - m_lo) * (n_draws > 0 ? n_draws : 1);
  for (int q = tid; q < n_out; q += T) {  // (member item, draw) pairs of the group
    const int mi = m_lo + (n_draws > 0 ? q / n_draws : q), j = n_draws > 0 ? q % n_draws : 0;
    const int tm = it.grp_off ? it.members[mi] : t;
    if (j == 0 && lse_out) lse_out[tm] = lse;
    if (n_draws > 0) {
      const int row_m = it.row ? it.row[tm] : tm;
      const uint32_t rng_row = (uint32_t)((int64_t)row_m + it.row_offset);
      const uint32_t pid = it.particle ? (uint32_t)it.particle[tm] : (uint32_t)j;
      int32_t res = PCLEAN_CHOICE_NEW;
      if (U != 0) {
        const uint64_t x = pclean_mulhi64(pclean_rand64(seed, rng_row, site, pid, sweep), U);
        unsigned int lo = 0, hi = n2 - 1;  // smallest index with prefix > x
        while (lo < hi) {
          const unsigned int mid = (lo + hi) >> 1;
          if (u2[mid] > x)
            hi = mid;
          else
            lo = mid + 1;
        }
        const int k = ks[lo];
        res = k == n ? PCLEAN_CHOICE_NEW : k;
      }
      draws_out[(size_t)tm * n_draws + j] = res;
    }
  }
}

typedef void (*fast_kernel_t)(const FastRootDev, const DensDev, const ItemsDev, const ChildrenDev, uint64_t, uint32_t,
                              uint32_t, int, int, double*, int32_t*, int32_t*, unsigned int*);

static fast_kernel_t pick_kernel(int n_terms) {
  if (n_terms <= 2) return fk_root_fast_kernel<2>;
  if (n_terms <= 4) return fk_root_fast_kernel<4>;
  if (n_terms <= 8) return fk_root_fast_kernel<8>;
  if (n_terms <= 12) return fk_root_fast_kernel<12>;
  return fk_root_fast_kernel<16>;
}

size_t pclean_fast_lds_bytes(int lmax, int dstride) {
  (void)lmax;
  (void)dstride;  // the density LUT no longer lives in LDS
  return (size_t)FAST_MAX_WAVES * 8 * 2 + (size_t)SURV2_CAP * 8 * 2 + (size_t)SURV2_CAP * 4 * 2 + 16 + 16;
}

int pclean_launch_root_fast(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const ChildrenDev& ch,
                            uint64_t seed, uint32_t sweep, uint32_t site, int n_draws, double* lse_out,
                            int32_t* draws_out, int32_t* overflow_flag, unsigned int* overflow_count) {
  if (it.n <= 0) return PCLEAN_OK;
  const size_t lds = pclean_fast_lds_bytes(fr.lmax, fr.dstride);
  // lanes: small workgroups (more of them resident per CU hide the per-row latency chain), at most
  // 64 rounds of 4-candidate slots per lane (the survivor bitmask is one 64-bit word)
  int T = 64;
  if (const char* e = getenv("PCLEAN_FAST_T")) T = std::max(64, std::min(1024, atoi(e) / 64 * 64));
  while (T < 1024 && ((fr.kpad >> 4) + T - 1) / T > 64) T += 64;  // at most 4 chunks of 16 rounds per lane
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, nullptr, nullptr, nullptr};
  fast_kernel_t kern = pick_kernel(fr.n_terms);
  HIPCHK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int kMaxBlocks = 2 * 1024 * 1024;  // < 2^32 threads per launch
  for (int base = 0; base < it.n; base += kMaxBlocks)
    hipLaunchKernelGGL(kern, dim3(std::min(kMaxBlocks, it.n - base)), dim3(T), lds, ctx->stream, fr, dn, it, ch, seed,
                       sweep, site, n_draws, base, lse_out, draws_out, overflow_flag, overflow_count);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
