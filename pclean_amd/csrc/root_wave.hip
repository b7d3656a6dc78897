// fk_root_wave_kernel — the dominant sweep kernel: candidate referents of a reference slot with many
// candidates (hospital Record block 1: K ~ 1e4 latent hospitals x 11 AddTypos terms; the nested Place /
// County slots of its new-row branch; any root whose table outgrew the LDS-resident generic kernel).
//
// Same contract and bit-identical results as enum_node_kernel (enum_kernels.hip) — the code the reference
// JIT-generates for a ForeignKeyNode (src/inference/proposal_compiler.jl:131-252) plus the CRP prior
// (165-171) and the AddTypos densities (src/distributions/add_typos.jl:50-66) — restructured for MI355X
// around four facts:
//   * candidate-compact byte tables comp_f[o][k] (compact_pair_kernel, rebuilt when the latent table's
//     columns change) turn the pair-table gather into contiguous byte streams: a lane reads 16 consecutive
//     candidates with one 16-byte load, a wave 1 KB per row per round -> fully coalesced;
//   * a candidate whose score is more than 28.5 nats below the maximum has fixed-point weight
//     floor(exp(s-m) 2^40) == 0 exactly (pclean_fixw), so it can influence neither the log-sum-exp nor a
//     draw.  An INTEGER PRE-FILTER proves that for almost every candidate: score <= prior_max -
//     c_min * (summed byte distances of the three most discriminating terms), compared with a lower bound
//     of the maximum (the exact score of the rows' current referent, or of the candidate with the smallest
//     summed distance).  Only the survivors (a handful per group) are scored in fp64, in plan order;
//   * rows with identical (observed tuple, ctx, referent) share the score vector: ONE WAVEFRONT PER GROUP of
//     such rows; every (member row, particle) pair draws with its own Philox counter by binary search over
//     the survivors' fixed-point prefix;
//   * the per-group dependent chain (group -> member -> row -> observed ids -> byte rows -> bound) is cut
//     out of the scan: group_desc_kernel (one thread per group, fully parallel) writes a 112-byte descriptor
//     per group; the persistent scan kernel reads it with one coalesced load and prefetches the next
//     group's descriptor while it scans.  No workgroup barrier anywhere: a wave owns its group and its
//     slice of LDS.
// Survivors are written to the wave's LDS slice in ascending candidate order (ballot + lane prefix), so the
// inverse CDF needs no sort.  Groups with more than WAVE_SURV_CAP pre-filter survivors (flat posteriors)
// are flagged and re-run by the host with the LDS-resident generic kernel — results are identical either way.
#include <algorithm>
#include <cstdlib>

#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"
#include "enum.h"

#define HALF_LOG26 1.629048269010741
#define ADD_TYPOS_IMPOSSIBLE (-1e5)
#define FIX_CUTOFF 28.5        // pclean_fixw(d) == 0 for d < -28.5
#define WAVE_SURV_CAP 256      // pre-filter survivors a wave keeps (4 per lane)
#ifndef WAVE_RB
#define WAVE_RB 4              // rounds (16 candidates per lane each) whose loads are in flight together
#endif
#ifndef WAVE_TC
#define WAVE_TC 6              // terms whose loads are in flight together in the exact scoring
#endif
#define WAVE_DCUT_OK 24        // a cut-off below this many summed edits is considered selective
#define GD_STRIDE 28           // int32 words per group descriptor
// descriptor words: 0 m_lo, 1 m_hi, 2 representative item, 3 row, 4 excl, 5 ctx0, 6 ctx1, 7 flags (bit 0: the
// excluded referent is garbage-collected), 8-9 bound (double), 10..25 observed value index of term f,
// 26-27 score of the "new row" candidate (double)

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
  // FK table: live rows only; option table (counts == null): every option, prior = its log-probability
  const bool live = k < n_cand && (counts ? counts[k] != 0 : true);
  prior_n[k] = live ? logc_full[k] - logden_n : -__builtin_inf();
  if (prior_e) prior_e[k] = live ? logc_full[k] - logden_e : -__builtin_inf();
}
// bit e of alive[q] = candidate 16 q + e can carry weight (live row / option with a finite prior)
__global__ void alive_kernel(const double* __restrict__ prior_n, int kpad, uint16_t* __restrict__ alive) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (kpad >> 4)) return;
  uint32_t m = 0;
  for (int e = 0; e < 16; ++e) m |= (prior_n[(q << 4) + e] > -__builtin_inf() ? 1u : 0u) << e;
  alive[q] = (uint16_t)m;
}

int pclean_build_compact(pclean_ctx* ctx, const uint8_t* pair, int n_obs, int n_lat, const int32_t* cand_col,
                         const uint16_t* lat_len, int n_cand, int kpad, uint8_t* comp, uint8_t* clen) {
  for (int o0 = 0; o0 < n_obs; o0 += 65535) {  // gridDim.y limit
    const int no = std::min(65535, n_obs - o0);
    hipLaunchKernelGGL(compact_pair_kernel, dim3((kpad + 255) / 256, no), dim3(256), 0, ctx->stream,
                       pair + (size_t)o0 * n_lat, no, n_lat, cand_col, n_cand, kpad, comp + (size_t)o0 * kpad);
  }
  hipLaunchKernelGGL(compact_len_kernel, dim3((kpad + 255) / 256), dim3(256), 0, ctx->stream, lat_len, cand_col, n_cand,
                     kpad, clen);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
int pclean_build_priors(pclean_ctx* ctx, const int64_t* counts, const double* logc_full, int n_cand, int kpad,
                        double logden_e, double logden_n, double* prior_e, double* prior_n, uint16_t* alive) {
  hipLaunchKernelGGL(priors_kernel, dim3((kpad + 255) / 256), dim3(256), 0, ctx->stream, counts, logc_full, n_cand,
                     kpad, logden_e, logden_n, prior_e, prior_n);
  hipLaunchKernelGGL(alive_kernel, dim3(((kpad >> 4) + 255) / 256), dim3(256), 0, ctx->stream, prior_n, kpad, alive);
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

// exact score of candidate k for the item described by (o[], ctx): prior first, then the terms in plan
// order — the operation order of candidate_score() (enum_kernels.hip)
__device__ __forceinline__ double fast_exact_score(const FastRootDev& fr, const DensDev& dn, const int* o, int ctx0,
                                                   int ctx1, int k, double pr) {
  double b = pr;
  for (int f = 0; f < fr.n_terms; ++f) {
    const FastTermDev& tm = fr.terms[f];
    if (o[f] < 0) continue;  // an explicitly missing observation contributes nothing (add_typos.jl:51-53)
    int d, L;
    if (tm.ctx_slot < 0) {
      d = tm.comp[(size_t)o[f] * fr.kpad + k];
      L = tm.clen[k];
    } else {
      const int c = tm.ctx_slot == 0 ? ctx0 : ctx1;
      const int val = tm.fn[(size_t)c * tm.fn_nb + tm.cand_col[k]];
      d = tm.pair[(size_t)o[f] * tm.n_lat + val];
      L = tm.lat_len[val];
    }
    b += (tm.max_typos >= 0 && d > tm.max_typos) ? ADD_TYPOS_IMPOSSIBLE : add_typos_dens(dn, L, d);
  }
  return b;
}

// One thread per group: the descriptor the scan kernel consumes, including the lower bound of the maximum
// from the exact score of the rows' current referent (only ever used as a filter, never as a score).
__global__ void group_desc_kernel(const FastRootDev fr, const DensDev dn, const ItemsDev it, const ChildrenDev ch,
                                  int n_groups, int32_t* __restrict__ gd) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const int m_lo = it.grp_off ? it.grp_off[g] : g, m_hi = it.grp_off ? it.grp_off[g + 1] : g + 1;
  const int t = it.grp_off ? it.members[m_lo] : g;
  const int row = it.row ? it.row[t] : t;
  const int excl = it.excl ? it.excl[t] : -1;
  const int ctx0 = it.ctx ? it.ctx[(size_t)t * PCLEAN_MAX_CTX] : 0;
  const int ctx1 = it.ctx ? it.ctx[(size_t)t * PCLEAN_MAX_CTX + 1] : 0;
  int o[PCLEAN_MAX_TERMS];
  for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) o[f] = f < fr.n_terms ? fr.terms[f].obs_col[row] : -1;
  const bool deleted = excl >= 0 && fr.counts && fr.counts[excl] <= 1;
  double bound = -__builtin_inf();
  if (excl >= 0 && !deleted && fr.logc_m1)
    bound = fast_exact_score(fr, dn, o, ctx0, ctx1, excl, fr.logc_m1[excl] - fr.scal[1]) - 1.0;
  int32_t* d = gd + (size_t)g * GD_STRIDE;
  d[0] = m_lo;
  d[1] = m_hi;
  d[2] = t;
  d[3] = row;
  d[4] = excl;
  d[5] = ctx0;
  d[6] = ctx1;
  d[7] = deleted ? 1 : 0;
  d[8] = __double2loint(bound);
  d[9] = __double2hiint(bound);
  for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) d[10 + f] = o[f];
  // score of the "new row" candidate (proposal_compiler.jl:221-230): CRP new-table term + log-marginals of the
  // children in plan order — new_score() of enum_kernels.hip; an option list (LEAF node) has none
  double sn = -__builtin_inf();
  if (!fr.is_leaf) {
    const double logden = excl >= 0 ? fr.scal[1] : fr.scal[0];
    double snew = 0.0;
    for (int c = 0; c < ch.n; ++c) {
      size_t idx = (size_t)t;
      if (ch.obs_col[c]) {
        const int oc = ch.obs_col[c][row];
        idx = oc < 0 ? (size_t)ch.n_obs[c] : (size_t)oc;
      }
      snew += ch.arr[c][idx];
    }
    sn = ((deleted ? fr.scal[3] : fr.scal[2]) - logden) + snew;
  }
  d[26] = __double2loint(sn);
  d[27] = __double2hiint(sn);
}

template <int NT>
__global__ __launch_bounds__(256) void fk_root_wave_kernel(const FastRootDev fr, const DensDev dn, const ItemsDev it,
                                                           uint64_t seed, uint32_t sweep, uint32_t site, int n_draws,
                                                           int n_groups, const int32_t* __restrict__ gd,
                                                           double* __restrict__ lse_out,
                                                           int32_t* __restrict__ draws_out,
                                                           int32_t* __restrict__ overflow_flag,
                                                           unsigned int* __restrict__ overflow_count) {
  __shared__ uint64_t s_pref[4][WAVE_SURV_CAP + 8];
  __shared__ double s_sc[4][WAVE_SURV_CAP + 8];
  __shared__ int32_t s_k[4][WAVE_SURV_CAP + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t* pref = s_pref[wave];
  double* scv = s_sc[wave];
  int32_t* ksv = s_k[wave];
  const int n = fr.n_cand;
  const int nquads = fr.kpad >> 4;
  const int draw_is = it.draw_is ? it.draw_is : n_draws, draw_ds = it.draw_ds ? it.draw_ds : 1;
  // XCD-aware persistent mapping: workgroup b runs on XCD b % 8 (each XCD has its own L2).  The groups arrive
  // sorted by referent, so consecutive groups stream the same byte rows: XCD x takes the x-th contiguous eighth
  // of the groups and its workgroups walk it with a stride — the rows of one referent stay in ONE L2.
  const int xcd = blockIdx.x & 7, nbx = (gridDim.x + 7 - xcd) >> 3;
  const int per = (n_groups + 7) >> 3;
  const int g_hi = min((xcd + 1) * per, n_groups);
  const int wstride = nbx * 4;
  int g = xcd * per + (blockIdx.x >> 3) * 4 + wave;
  int dv = (g < g_hi && lane < GD_STRIDE) ? gd[(size_t)g * GD_STRIDE + lane] : 0;
  for (; g < g_hi; g += wstride) {
    // ---- descriptor -> wave-uniform registers; the next group's descriptor is requested right away -------
    const int m_lo = __builtin_amdgcn_readlane(dv, 0), m_hi = __builtin_amdgcn_readlane(dv, 1);
    const int t = __builtin_amdgcn_readlane(dv, 2);
    const int excl = __builtin_amdgcn_readlane(dv, 4);
    const int ctx0 = __builtin_amdgcn_readlane(dv, 5), ctx1 = __builtin_amdgcn_readlane(dv, 6);
    const bool deleted = (__builtin_amdgcn_readlane(dv, 7) & 1) != 0;
    double bound = __hiloint2double(__builtin_amdgcn_readlane(dv, 9), __builtin_amdgcn_readlane(dv, 8));
    // score of the "new row" candidate (index n, last in natural order; -inf for an option list), from the descriptor
    const double sn = __hiloint2double(__builtin_amdgcn_readlane(dv, 27), __builtin_amdgcn_readlane(dv, 26));
    int o[NT];
#pragma unroll
    for (int f = 0; f < NT; ++f) o[f] = __builtin_amdgcn_readlane(dv, 10 + f);
    {
      const int gn = g + wstride;
      dv = (gn < g_hi && lane < GD_STRIDE) ? gd[(size_t)gn * GD_STRIDE + lane] : 0;
    }
    const bool excluded = excl >= 0;
    const double logden = excluded ? fr.scal[1] : fr.scal[0];
    const double* __restrict__ prior = (excluded && fr.prior_e) ? fr.prior_e : fr.prior_n;
    const double pmax = excluded ? fr.prior_max_e : fr.prior_max_n;
    // byte rows of the (up to 3) most discriminating terms, summed by the integer pre-filter
    const uint4* prow[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      prow[p] = nullptr;
      if (p < fr.n_pre) {
        const int f = fr.pre[p];
        int of = -1;
#pragma unroll
        for (int q = 0; q < NT; ++q)
          if (q == f) of = o[q];
        if (of >= 0) prow[p] = reinterpret_cast<const uint4*>(fr.terms[f].comp + (size_t)of * fr.kpad);
      }
    }
    // summed byte distances of the 16 candidates of quad q: D[4w + e], e = byte e of dword w
    auto quad_sums = [&](int q, uint32_t* lo, uint32_t* hi) {
#pragma unroll
      for (int w = 0; w < 4; ++w) lo[w] = hi[w] = 0u;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (prow[p]) {
          const uint4 c = prow[p][q];
          const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            lo[w] += cw[w] & 0x00ff00ffu;
            hi[w] += (cw[w] >> 8) & 0x00ff00ffu;
          }
        }
    };

    // Stage 0 (groups without a retained referent: nested slots of a new row, option lists, initialisation):
    // the live candidate with the smallest summed distance; its exact score - 1 is the lower bound of the
    // maximum.  Stage 1: the pre-filter scan.  Both stages share ONE copy of the exact-scoring code below.
    // A retained referent that explains the row badly (wrong entity, many typos) gives a useless bound: when its
    // cut-off would let candidates more than WAVE_DCUT_OK edits away through, stage 0 runs as well and the better
    // of the two bounds is used.
    bound = fmax(bound, sn);  // the new-row candidate is a candidate too: its exact score bounds the maximum from below
    bool need_bound = !(bound > -__builtin_inf()) && fr.n_pre > 0;
    if (!need_bound && fr.n_pre > 0) {
      const double x = (pmax - bound + FIX_CUTOFF) * fr.inv_c;
      need_bound = !(x < (double)WAVE_DCUT_OK);
    }
    const double bound0 = bound;
    int ns = 0;
    bool over = false;
    for (int stage = need_bound ? 0 : 1; stage < 2; ++stage) {
      ns = 0;
      if (stage == 0) {
        uint64_t best = ~0ull;
        for (int q0 = 0; q0 < nquads; q0 += 64 * WAVE_RB) {
#pragma unroll
          for (int r = 0; r < WAVE_RB; ++r) {
            const int q = q0 + r * 64 + lane;
            if (q < nquads) {
              uint32_t lo[4], hi[4];
              quad_sums(q, lo, hi);
              const uint32_t al = fr.alive[q];
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const uint32_t D[4] = {lo[w] & 0xffffu, hi[w] & 0xffffu, lo[w] >> 16, hi[w] >> 16};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int k = (q << 4) + (w << 2) + e;
                  const uint64_t key = ((uint64_t)D[e] << 32) | (uint32_t)k;
                  if (key < best && ((al >> (4 * w + e)) & 1u) && k != excl) best = key;
                }
              }
            }
          }
        }
        for (int sh = 32; sh > 0; sh >>= 1) {
          const uint64_t other = __shfl_xor(best, sh, 64);
          best = other < best ? other : best;
        }
        if (best != ~0ull) {
          if (lane == 0) ksv[0] = (int)(uint32_t)best;
          ns = 1;
        }
      } else {
        // Pre-filter threshold: a candidate whose summed edit distance D over the pre-filter terms exceeds dcut
        // scores at most pmax - c_min * D < bound - FIX_CUTOFF <= max - FIX_CUTOFF, i.e. its fixed-point weight
        // is exactly 0 (c_min = smallest cost of one edit, fr.inv_c = 1 / c_min; terms not summed and missing
        // observations only lower the score further).
        uint32_t dcut = 0xffffu;
        if (fr.n_pre > 0 && bound > -__builtin_inf()) {
          const double x = (pmax - bound + FIX_CUTOFF) * fr.inv_c;
          if (x >= 0.0 && x < 60000.0) dcut = (uint32_t)x + 2u;
        }
        // branch-free integer scan, 16 candidates per lane per round, WAVE_RB rounds of loads in flight;
        // survivors (live candidates with D <= dcut) go to the wave's LDS list in ascending candidate order
        for (int q0 = 0; q0 < nquads; q0 += 64 * WAVE_RB) {
          uint32_t m16[WAVE_RB];
#pragma unroll
          for (int r = 0; r < WAVE_RB; ++r) {
            const int q = q0 + r * 64 + lane;
            m16[r] = 0;
            if (q < nquads) {
              uint32_t lo[4], hi[4];
              quad_sums(q, lo, hi);
              uint32_t mk = 0;
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                mk |= ((lo[w] & 0xffffu) <= dcut ? 1u : 0u) << (4 * w);
                mk |= ((hi[w] & 0xffffu) <= dcut ? 1u : 0u) << (4 * w + 1);
                mk |= ((lo[w] >> 16) <= dcut ? 1u : 0u) << (4 * w + 2);
                mk |= ((hi[w] >> 16) <= dcut ? 1u : 0u) << (4 * w + 3);
              }
              m16[r] = mk & (uint32_t)fr.alive[q];  // padding and free slots never survive
            }
          }
          uint32_t any = 0;
#pragma unroll
          for (int r = 0; r < WAVE_RB; ++r) any |= m16[r];
          if (__ballot(any != 0) == 0ull) continue;  // wave-uniform: most rounds hold no survivor at all
#pragma unroll
          for (int r = 0; r < WAVE_RB; ++r) {
            if (__ballot(m16[r] != 0) == 0ull) continue;
            const int q = q0 + r * 64 + lane;
            const int cnt = __builtin_popcount(m16[r]);
            int incl = cnt;
            for (int sh = 1; sh < 64; sh <<= 1) {
              const int x = __shfl_up(incl, sh, 64);
              if (lane >= sh) incl += x;
            }
            int pos = ns + incl - cnt;
            for (uint32_t mm = m16[r]; mm; mm &= mm - 1) {
              if (pos < WAVE_SURV_CAP) ksv[pos] = (q << 4) + __builtin_ctz(mm);
              ++pos;
            }
            ns += __shfl(incl, 63, 64);
          }
        }
        if (ns > WAVE_SURV_CAP) {
          over = true;
          break;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // ---- exact fp64 scores of ksv[0..ns), one candidate per lane per pass -> scv.  The loads of a chunk of
      // terms are in flight together (byte distance + length, then the two density pieces); the fp64 additions
      // follow plan order (the operation order of candidate_score(), enum_kernels.hip).
      for (int base = 0; base < ns; base += 64) {
        const int j = base + lane;
        if (j < ns) {
          const int k = ksv[j];
          double b = prior[k];
          if (k == excl) b = deleted ? -__builtin_inf() : fr.logc_m1[excl] - logden;
#pragma unroll
          for (int f0 = 0; f0 < NT; f0 += WAVE_TC) {
            int dd[WAVE_TC], LL[WAVE_TC];
#pragma unroll
            for (int u = 0; u < WAVE_TC; ++u) {
              const int f = f0 + u;
              dd[u] = 0;
              LL[u] = 0;
              if (f < NT && f < fr.n_terms && o[f < NT ? f : 0] >= 0) {
                const FastTermDev& tm = fr.terms[f];
                if (tm.ctx_slot < 0) {
                  dd[u] = tm.comp[(size_t)o[f < NT ? f : 0] * fr.kpad + k];
                  LL[u] = tm.clen[k];
                } else {
                  const int c = tm.ctx_slot == 0 ? ctx0 : ctx1;
                  const int val = tm.fn[(size_t)c * tm.fn_nb + tm.cand_col[k]];
                  dd[u] = tm.pair[(size_t)o[f < NT ? f : 0] * tm.n_lat + val];
                  LL[u] = tm.lat_len[val];
                }
              }
            }
            double nbv[WAVE_TC], lgv[WAVE_TC];
#pragma unroll
            for (int u = 0; u < WAVE_TC; ++u) {
              nbv[u] = dn.nb[(size_t)((LL[u] + 4) / 5) * dn.nb_stride + dd[u]];
              lgv[u] = dn.logl[LL[u]];
            }
#pragma unroll
            for (int u = 0; u < WAVE_TC; ++u) {
              const int f = f0 + u;
              if (f < NT && f < fr.n_terms && o[f < NT ? f : 0] >= 0) {  // a missing observation contributes nothing
                double l = nbv[u];                                      // operation order of add_typos_dens()
                l -= lgv[u] * (double)dd[u];
                l -= HALF_LOG26 * (double)dd[u];
                const int mt = fr.terms[f].max_typos;
                b += (mt >= 0 && dd[u] > mt) ? ADD_TYPOS_IMPOSSIBLE : l;
              }
            }
          }
          scv[j] = b;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (stage == 0) bound = fmax(bound0, ns ? scv[0] - 1.0 : -__builtin_inf());
    }
    if (over) {  // flat posterior: the host re-runs these items with the generic kernel (flags are pre-zeroed)
      if (m_hi - m_lo == 1) {
        if (lane == 0) overflow_flag[t] = PCLEAN_CHOICE_NEW;  // marker understood by compact_new_kernel
      } else {
        for (int mi = m_lo + lane; mi < m_hi; mi += 64) overflow_flag[it.members[mi]] = PCLEAN_CHOICE_NEW;
      }
      if (lane == 0) atomicAdd(overflow_count, (unsigned int)(m_hi - m_lo));
      continue;
    }
    // ---- maximum, fixed-point weights, inclusive prefix (survivors in ascending order, then the new row) ---
    double m = sn;
    for (int base = 0; base < ns; base += 64) {
      const int j = base + lane;
      if (j < ns) m = fmax(m, scv[j]);
    }
    m = wave_max64(m);
    uint64_t carry = 0;
    for (int base = 0; base < ns; base += 64) {
      const int j = base + lane;
      const uint64_t u = (j < ns && m != -__builtin_inf()) ? pclean_fixw(scv[j] - m) : 0ull;
      unsigned long long incl = u;
      for (int sh = 1; sh < 64; sh <<= 1) {
        const unsigned long long x = __shfl_up(incl, sh, 64);
        if (lane >= sh) incl += x;
      }
      if (j < ns) pref[j] = carry + incl;
      carry += __shfl(incl, 63, 64);
    }
    const uint64_t U = carry + ((m == -__builtin_inf()) ? 0ull : pclean_fixw(sn - m));
    if (lane == 0) pref[ns] = U;
    __builtin_amdgcn_wave_barrier();
    // ---- lse + draws of every (member item, draw) pair of the group ------------------------------------------
    const double lse = pclean_lse_from_fix(m, U);
    const int nd_eff = n_draws > 0 ? n_draws : 1;
    const int n_mem = m_hi - m_lo;
    const int n_out = n_mem * nd_eff;
    for (int q = lane; q < n_out; q += 64) {
      const int mi = m_lo + q / nd_eff, j = q % nd_eff;
      const int tm = n_mem == 1 ? t : it.members[mi];
      if (j == 0 && lse_out) lse_out[tm] = lse;
      if (n_draws > 0) {
        int32_t res = fr.is_leaf ? n - 1 : PCLEAN_CHOICE_NEW;
        if (U != 0) {
          const int row_m = it.row ? it.row[tm] : tm;
          const uint32_t rng_row = it.rng_row ? (uint32_t)it.rng_row[tm] : (uint32_t)((int64_t)row_m + it.row_offset);
          const uint32_t pid = it.particle ? (uint32_t)it.particle[tm] : (uint32_t)j;
          const uint64_t x = pclean_mulhi64(pclean_rand64(seed, rng_row, site, pid, sweep), U);
          int a = 0, b = ns;  // smallest index with prefix > x (index ns = the new row)
          while (a < b) {
            const int mid = (a + b) >> 1;
            if (pref[mid] > x)
              b = mid;
            else
              a = mid + 1;
          }
          res = a == ns ? (fr.is_leaf ? n - 1 : PCLEAN_CHOICE_NEW) : ksv[a];
        }
        draws_out[(size_t)tm * draw_is + (size_t)j * draw_ds] = res;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

typedef void (*wave_kernel_t)(const FastRootDev, const DensDev, const ItemsDev, uint64_t, uint32_t, uint32_t, int, int,
                              const int32_t*, double*, int32_t*, int32_t*, unsigned int*);

static wave_kernel_t pick_kernel(int n_terms) {
  if (n_terms <= 2) return fk_root_wave_kernel<2>;
  if (n_terms <= 4) return fk_root_wave_kernel<4>;
  if (n_terms <= 8) return fk_root_wave_kernel<8>;
  if (n_terms <= 12) return fk_root_wave_kernel<12>;
  return fk_root_wave_kernel<16>;
}

size_t pclean_fast_desc_words(int n_groups) { return (size_t)std::max(n_groups, 1) * GD_STRIDE; }

int pclean_launch_root_fast(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const ChildrenDev& ch,
                            uint64_t seed, uint32_t sweep, uint32_t site, int n_draws, double* lse_out,
                            int32_t* draws_out, int32_t* overflow_flag, unsigned int* overflow_count,
                            int32_t* desc_scratch) {
  if (it.n <= 0) return PCLEAN_OK;
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(group_desc_kernel, dim3((it.n + 255) / 256), dim3(256), 0, ctx->stream, fr, dn, it, ch, it.n,
                     desc_scratch);
  // persistent grid: 4 groups (waves) per workgroup, up to 8 workgroups per CU
  // persistent grid = what is resident at once (a workgroup that starts late would still own its full share)
  wave_kernel_t kern = pick_kernel(fr.n_terms);
  static int resident[17] = {0};  // per kernel variant (indexed by its term capacity), queried once
  const int variant = fr.n_terms <= 2 ? 2 : fr.n_terms <= 4 ? 4 : fr.n_terms <= 8 ? 8 : fr.n_terms <= 12 ? 12 : 16;
  if (!resident[variant]) {
    int per_cu = 0, n_cu = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    per_cu = std::min(per_cu, 6);  // 256-thread workgroups with ~100 SGPRs: the hardware admits 6 per CU (MI355X_MICROARCH.md)
    resident[variant] = n_cu * per_cu;
  }
  int wgs = resident[variant];
  if (const char* e = getenv("PCLEAN_WAVE_WGS")) wgs = std::max(1, atoi(e));
  wgs = std::min(wgs, (it.n + 3) / 4);
  wgs = (wgs + 7) & ~7;  // a multiple of the 8 XCDs (the kernel splits the groups into 8 contiguous ranges)
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, ctx->stream, fr, dn, it, seed, sweep, site,
                     n_draws, it.n, desc_scratch, lse_out, draws_out, overflow_flag, overflow_count);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
