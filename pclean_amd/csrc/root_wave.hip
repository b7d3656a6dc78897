// fk_root_wave_kernel — the dominant sweep kernel: candidate referents of a reference slot with many
// candidates (hospital Record block 1: K ~ 1e4 latent hospitals x 11 AddTypos terms; the nested Place /
// County slots of its new-row branch; any root whose table outgrew the LDS-resident generic kernel).
//
// Same contract and bit-identical results as enum_node_kernel (enum_kernels.hip) — the code the reference
// JIT-generates for a ForeignKeyNode (src/inference/proposal_compiler.jl:131-252) plus the CRP prior
// (165-171) and the AddTypos densities (src/distributions/add_typos.jl:50-66) — restructured for MI355X
// (DESIGN.md §2, §5).  Round-2 counters showed the first version of this kernel to be VALU-issue bound
// (2 800 vector instructions per group, a third of them SGPR-spill traffic), not memory bound; this version
// is built to issue as few vector instructions per group as possible:
//   * candidate-compact byte tables comp_f[o][k] = min(D_f[o][value of candidate k], 42) (compact_pair_kernel,
//     rebuilt when the latent table's columns change): a lane reads 16 consecutive candidates of a row with one
//     16-byte load, fully coalesced.  Bytes saturate at 42 so that the sum of three rows fits a byte: the
//     INTEGER PRE-FILTER is three packed 32-bit additions per four candidates and one add + and for the compare;
//   * a candidate whose score is more than 28.5 nats below the maximum has fixed-point weight
//     floor(exp(s-m) 2^40) == 0 exactly (pclean_fixw), so it can influence neither the log-sum-exp nor a
//     draw.  The pre-filter proves that for almost every candidate: score <= prior_max - c_min * (summed
//     saturated byte distances of the three most discriminating terms), compared with a lower bound of the
//     maximum (exact score of the rows' current referent / of the "new row" candidate, computed per group by
//     group_desc_kernel).  Only the survivors (a handful) are scored in fp64, in plan order (a saturated byte
//     sends that one term to the pair table for its true distance);
//   * rows with identical (observed tuple, ctx, referent) share the score vector: ONE WAVEFRONT PER GROUP of
//     such rows; every (member row, particle) pair draws with its own Philox counter by binary search over
//     the survivors' fixed-point prefix.  Groups are sorted by (referent, pre-filter observed values) and a
//     wave takes WAVE_CHUNK consecutive groups: a group whose pre-filter rows and cut-off are covered by the
//     previous scan reuses its survivor list (any superset of the required list gives identical results);
//   * group_desc_kernel (one thread per group) writes a 128-byte descriptor per group; the scan kernel reads it
//     with one coalesced load and prefetches the next group's.  Chunks are handed out by one atomic counter per
//     XCD (each XCD walks its own contiguous eighth of the groups: the byte rows of one referent stay in ONE
//     L2) and stolen from the other XCDs at the tail.  No workgroup barrier anywhere;
//   * the kernel's table descriptor (FastRootDev, 1.3 KB) is read through the kernarg segment with scalar
//     loads where it is needed instead of living in (spilled) SGPRs; the per-group log-sum-exp is computed by
//     group_lse_kernel, one thread per group, instead of redundantly by 64 lanes.
// Groups with more than WAVE_SURV_CAP pre-filter survivors (flat posteriors) are flagged and re-run by the
// host with the LDS-resident generic kernel — results are identical either way.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstdio>

#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"
#include "enum.h"

#define ADD_TYPOS_IMPOSSIBLE (-1e5)
#define FIX_CUTOFF 28.5        // pclean_fixw(d) == 0 for d < -28.5
#define PRE_CLAMP 42           // compact bytes saturate here: three of them sum to <= 126 < 128
#define CUT_ALL 126u           // cut-off that lets every live candidate through
#define WAVE_SURV_CAP 256      // pre-filter survivors a wave keeps
#ifndef WAVE_RB
#define WAVE_RB 1              // rounds (16 candidates per lane each) in flight together in the whole-row fallback of the scan
#endif
#ifndef WAVE_TC
#define WAVE_TC 6              // terms whose loads are in flight together in the exact scoring
#endif
#ifndef WAVE_CHUNK
#define WAVE_CHUNK 4           // consecutive groups a wave takes at a time (PCLEAN_WAVE_CHUNK; measured at the end of round 6 on the Measure slot's 97 k groups: 1 / 2 / 3 / 4 / 6 / 8 / 16 / 32 -> 0.467 / 0.441 / 0.435 / 0.437 / 0.455 / 0.479 / 0.566 / 0.891 ms for its launch group — the balance over the waves matters more than the few list reuses a longer run of neighbours adds)
#endif
#ifndef WAVE_MIN_WAVES
#define WAVE_MIN_WAVES 5       // resident workgroups per CU the register allocation aims at (measured: 5 beats 6, 7 and 8)
#endif
#ifndef WAVE_BLK_CAP
#define WAVE_BLK_CAP 128       // passing 64-candidate blocks a scan's fine level takes (more: the rows are streamed whole)
#endif
#ifndef WAVE_EXACT_BY_TERM
#define WAVE_EXACT_BY_TERM 2   // exact scoring: 0 one candidate per lane, 1 one (candidate, term) per lane, 2 the latter for short lists
#endif
#define WAVE_RES_UNRESOLVED (-2147483647 - 1)  // g_res of a group the scan kernel has to walk
#define WAVE_DCUT_OK 24u       // a cut-off below this many summed edits is considered selective
#define WAVE_GUESS 4u          // first cut-off tried when the descriptor's bound is useless (widened until something survives)
#define WAVE_SLACK 3u          // scanned cut-off = required + slack: makes the list reusable by the next groups
#define WAVE_CTR_STRIDE 1024    // uint32 words between the eight per-XCD chunk counters: 4 KB apart.  Device-scope atomics on one
                                // cache line are served one after the other at the memory side (~30 ns each): with the eight
                                // counters in ONE line the 40 000 chunk grabs of a 1M-row launch were a 1.2 ms floor of the kernel
#define WAVE_WORK_CTR 512       // word of the chunk-counter area that counts the work list's entries
#define WL_SEGS 64              // segments of the work list while group_settle_kernel appends to it ...
#define WL_SEG_CTR(s) (16 + 32 * (s))  // ... and the word of the chunk-counter area that counts segment s (128 bytes apart)
#define GD_STRIDE 32           // int32 words per group descriptor (one 128-byte line)
// descriptor words: 0 m_lo, 1 m_hi, 2 representative item, 3 row, 4 excl, 5 ctx0, 6 ctx1, 7 flags (bit 0: the
// excluded referent is garbage-collected, bit 1: the bound is useless -> guess and refine, bit 2: words 30-31 hold
// the exact score of the current referent, bit 3: resolved by group_desc_kernel, the scan kernel skips it), 8-9 bound (double;
// already includes the new-row score), 10..25 observed value index of term f, 26-27 score of the "new row"
// candidate (double), 28 first cut-off, 29 cut-off the bound would require (refine mode: upper limit), 30-31 exact
// score of the current referent (double)

__global__ void compact_pair_kernel(const uint8_t* __restrict__ pair, int n_obs, int n_lat,
                                    const int32_t* __restrict__ cand_col, int n_cand, int kpad,
                                    uint8_t* __restrict__ comp) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (k >= kpad) return;
  uint8_t v = 0;
  if (k < n_cand) v = pair[(size_t)o * n_lat + cand_col[k]];
  comp[(size_t)o * kpad + k] = v < PRE_CLAMP ? v : (uint8_t)PRE_CLAMP;
}
// cmin[o][kb] = smallest byte of comp[o][64 kb .. 64 kb + 63] (padding blocks: 255).  One thread per (row, block).
__global__ void compact_min_kernel(const uint8_t* __restrict__ comp, int n_obs, int kpad, int cstride,
                                   uint8_t* __restrict__ cmin) {
  const int kb = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (kb >= cstride) return;
  uint32_t m = 255u;
  const int q0 = kb * 4, nq = kpad >> 4;
  for (int q = q0; q < q0 + 4 && q < nq; ++q) {
    const uint4 c = reinterpret_cast<const uint4*>(comp + (size_t)o * kpad)[q];
    const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = min(m, (cw[w] >> (8 * e)) & 0xffu);
  }
  cmin[(size_t)o * cstride + kb] = (uint8_t)m;
}
__global__ void compact_len_kernel(const uint16_t* __restrict__ lat_len, const int32_t* __restrict__ cand_col,
                                   int n_cand, int kpad, uint8_t* __restrict__ clen) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= kpad) return;
  clen[k] = k < n_cand ? (uint8_t)lat_len[cand_col[k]] : (uint8_t)0;
}
__global__ void priors_kernel(const int64_t* __restrict__ counts, const double* __restrict__ logc_full, int n_cand,
                              int kpad, double logden_e, double logden_n, double* __restrict__ prior_e,
                              double* __restrict__ prior_n, uint16_t* __restrict__ alive) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  // FK table: live rows only; option table (counts == null): every option, prior = its log-probability
  const bool live = k < n_cand && (counts ? counts[k] != 0 : true);
  const double pn = live ? logc_full[k] - logden_n : -__builtin_inf();
  if (k < kpad) {
    prior_n[k] = pn;
    if (prior_e) prior_e[k] = live ? logc_full[k] - logden_e : -__builtin_inf();
  }
  // bit e of alive[q] = candidate 16 q + e can carry weight (live row / option with a finite prior): the wavefront's 64
  // candidates are four words (kpad is a multiple of 16, the block size a multiple of 64)
  if (alive) {
    const uint64_t m = __ballot(pn > -__builtin_inf());
    const int lane = threadIdx.x & 63;
    if ((lane & 15) == 0 && k < kpad) alive[k >> 4] = (uint16_t)(m >> lane);
  }
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
// the same for the candidates rows[0 .. n_rows) alone (a device commit wrote those rows of the table)
__global__ void compact_update_kernel(const uint8_t* __restrict__ pair, int n_lat, const int32_t* __restrict__ cand_col,
                                      const uint16_t* __restrict__ lat_len, const int32_t* __restrict__ rows, int n_rows, int kpad,
                                      uint8_t* __restrict__ comp, uint8_t* __restrict__ clen) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (j >= n_rows) return;
  const int k = rows[j];
  const int val = cand_col[k];
  const uint8_t v = pair[(size_t)o * n_lat + val];
  comp[(size_t)o * kpad + k] = v < PRE_CLAMP ? v : (uint8_t)PRE_CLAMP;
  if (o == 0) clen[k] = (uint8_t)lat_len[val];
}
int pclean_update_compact(pclean_ctx* ctx, const uint8_t* pair, int n_obs, int n_lat, const int32_t* cand_col,
                          const uint16_t* lat_len, const int32_t* rows, int n_rows, int kpad, uint8_t* comp, uint8_t* clen) {
  if (n_rows <= 0) return PCLEAN_OK;
  for (int o0 = 0; o0 < n_obs; o0 += 65535) {
    const int no = std::min(65535, n_obs - o0);
    hipLaunchKernelGGL(compact_update_kernel, dim3((n_rows + 63) / 64, no), dim3(64), 0, ctx->stream, pair + (size_t)o0 * n_lat,
                       n_lat, cand_col, lat_len, rows, n_rows, kpad, comp + (size_t)o0 * kpad, clen);
  }
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
int pclean_build_compact_min(pclean_ctx* ctx, const uint8_t* comp, int n_obs, int kpad, int cstride, uint8_t* cmin) {
  for (int o0 = 0; o0 < n_obs; o0 += 65535) {  // gridDim.y limit
    const int no = std::min(65535, n_obs - o0);
    hipLaunchKernelGGL(compact_min_kernel, dim3((cstride + 63) / 64, no), dim3(64), 0, ctx->stream,
                       comp + (size_t)o0 * kpad, no, kpad, cstride, cmin + (size_t)o0 * cstride);
  }
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
// ... of the blocks that hold the candidates rows[0 .. n_rows) alone (after compact_update_kernel wrote those candidates'
// bytes): thread (j, o) recomputes block rows[j] / 64 of row o — several rows of one block write the same value
__global__ void compact_min_rows_kernel(const uint8_t* __restrict__ comp, int kpad, int cstride, const int32_t* __restrict__ rows,
                                        int n_rows, uint8_t* __restrict__ cmin) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (j >= n_rows) return;
  const int kb = rows[j] >> 6;
  uint32_t m = 255u;
  const int q0 = kb * 4, nq = kpad >> 4;
  for (int q = q0; q < q0 + 4 && q < nq; ++q) {
    const uint4 c = reinterpret_cast<const uint4*>(comp + (size_t)o * kpad)[q];
    const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = min(m, (cw[w] >> (8 * e)) & 0xffu);
  }
  cmin[(size_t)o * cstride + kb] = (uint8_t)m;
}
int pclean_update_compact_min(pclean_ctx* ctx, const uint8_t* comp, int n_obs, int kpad, int cstride, const int32_t* rows,
                              int n_rows, uint8_t* cmin) {
  if (n_rows <= 0) return PCLEAN_OK;
  // a few rows of a table of thousands (a latent sub-batch's commit): their blocks; a tenth of the table and more (the Measure
  // rows an observed-class commit wrote): every block is touched anyway — the whole rows, coalesced
  static const bool no_rows = getenv("PCLEAN_NO_COMPACT_MIN_ROWS") != nullptr;
  if (no_rows || (size_t)n_rows * 64 * 2 > (size_t)kpad) return pclean_build_compact_min(ctx, comp, n_obs, kpad, cstride, cmin);
  for (int o0 = 0; o0 < n_obs; o0 += 65535) {
    const int no = std::min(65535, n_obs - o0);
    hipLaunchKernelGGL(compact_min_rows_kernel, dim3((n_rows + 63) / 64, no), dim3(64), 0, ctx->stream, comp + (size_t)o0 * kpad,
                       kpad, cstride, rows, n_rows, cmin + (size_t)o0 * cstride);
  }
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
int pclean_build_priors(pclean_ctx* ctx, const int64_t* counts, const double* logc_full, int n_cand, int kpad,
                        double logden_e, double logden_n, double* prior_e, double* prior_n, uint16_t* alive) {
  static const bool unfused = getenv("PCLEAN_NO_FUSED_PRIORS") != nullptr;
  hipLaunchKernelGGL(priors_kernel, dim3((kpad + 255) / 256), dim3(256), 0, ctx->stream, counts, logc_full, n_cand,
                     kpad, logden_e, logden_n, prior_e, prior_n, unfused ? (uint16_t*)nullptr : alive);
  if (unfused) hipLaunchKernelGGL(alive_kernel, dim3(((kpad >> 4) + 255) / 256), dim3(256), 0, ctx->stream, prior_n, kpad, alive);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

__device__ __forceinline__ double wave_max64(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// AddTypos density of term tm for candidate k against observed value of (>= 0): fr.atd[L][d] holds the three
// fp64 operations of term_density() (enum_kernels.hip; add_typos.jl:61-63) evaluated once on the host
__device__ __forceinline__ double wave_term_dens(const FastRootDev& fr, const FastTermDev& tm, int of, int ctx0, int ctx1,
                                                 int k) {
  int d, L;
  if (tm.ctx_slot < 0) {
    d = tm.comp[(size_t)of * fr.kpad + k];
    L = tm.clen[k];
    if (d == PRE_CLAMP) d = tm.pair[(size_t)of * tm.n_lat + tm.cand_col[k]];  // saturated byte: the true distance
  } else {
    const int c = tm.ctx_slot == 0 ? ctx0 : ctx1;
    const int val = tm.fn[(size_t)c * tm.fn_nb + tm.cand_col[k]];
    d = tm.pair[(size_t)of * tm.n_lat + val];
    L = tm.lat_len[val];
  }
  return (tm.max_typos >= 0 && d > tm.max_typos) ? ADD_TYPOS_IMPOSSIBLE : fr.atd[(size_t)L * fr.atd_stride + d];
}

// exact score of candidate k for the item described by (o[], ctx): prior first, then the terms in plan
// order — the operation order of candidate_score() (enum_kernels.hip)
__device__ __forceinline__ double fast_exact_score(const FastRootDev& fr, const int* o, int ctx0, int ctx1, int k,
                                                   double pr) {
  double b = pr;
  for (int f = 0; f < fr.n_terms; ++f) {
    if (o[f] < 0) continue;  // an explicitly missing observation contributes nothing (add_typos.jl:51-53)
    b += wave_term_dens(fr, fr.terms[f], o[f], ctx0, ctx1, k);
  }
  return b;
}

// fast_exact_score() with every term's loads in flight together (one candidate per thread: group_gate_kernel,
// group_desc_kernel): the byte distance and the length of ALL terms first, then all the density-table entries — two round
// trips instead of two per term.  The loads are unconditional (a term the node does not have, a context term and a missing
// observation read the zero row: a load under a condition becomes a branch with its own wait); what does not count is left
// out of the sum, whose additions keep plan order: the same value as fast_exact_score(), bit for bit.
__device__ __forceinline__ double fast_exact_score_mlp(const FastRootDev& fr, const int* o, int ctx0, int ctx1, int k, double pr) {
  int d[PCLEAN_MAX_TERMS], L[PCLEAN_MAX_TERMS];
  bool sat = false;
#pragma unroll
  for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) {
    const bool plain = f < fr.n_terms && fr.terms[f].ctx_slot < 0;  // (wave-uniform)
    const uint8_t* crow = (plain && o[f] >= 0) ? fr.terms[f].comp + (size_t)o[f] * fr.kpad : fr.zero_row;
    const uint8_t* cl = plain ? fr.terms[f].clen : fr.zero_row;
    d[f] = crow[k];
    L[f] = cl[k];
  }
#pragma unroll
  for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) sat |= d[f] == PRE_CLAMP;  // (the zero row never is)
  if (sat) {  // saturated bytes (rare): the true distances
#pragma unroll
    for (int f = 0; f < PCLEAN_MAX_TERMS; ++f)
      if (d[f] == PRE_CLAMP) d[f] = fr.terms[f].pair[(size_t)o[f] * fr.terms[f].n_lat + fr.terms[f].cand_col[k]];
  }
  double l[PCLEAN_MAX_TERMS];
#pragma unroll
  for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) {
    const int mt = f < fr.n_terms ? fr.terms[f].max_typos : -1;
    l[f] = fr.atd[(size_t)L[f] * fr.atd_stride + ((mt >= 0 && d[f] > mt) ? 0 : d[f])];  // (beyond the limit: not looked up)
  }
  double b = pr;
#pragma unroll
  for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) {
    if (f < fr.n_terms) {
      if (fr.terms[f].ctx_slot >= 0) {  // (wave-uniform: a JuliaNode term goes through fn[ctx][.])
        if (o[f] >= 0) b += wave_term_dens(fr, fr.terms[f], o[f], ctx0, ctx1, k);
      } else {
        const int mt = fr.terms[f].max_typos;
        const double lf = (mt >= 0 && d[f] > mt) ? ADD_TYPOS_IMPOSSIBLE : l[f];
        if (o[f] >= 0) b += lf;  // an explicitly missing observation contributes nothing (add_typos.jl:51-53)
      }
    }
  }
  return b;
}

// observed values of `row` for the node's terms (-1 beyond n_terms): from the row-major copy when there is one
__device__ __forceinline__ void load_row_obs(const FastRootDev& fr, int row, int* o) {
  static_assert(PCLEAN_MAX_TERMS == 16, "four 16-byte loads");
  if (fr.obs_rm) {
    const int4* po = reinterpret_cast<const int4*>(fr.obs_rm + (size_t)row * PCLEAN_MAX_TERMS);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 v = po[q];
      o[4 * q] = v.x;
      o[4 * q + 1] = v.y;
      o[4 * q + 2] = v.z;
      o[4 * q + 3] = v.w;
    }
  } else {
    for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) o[f] = f < fr.n_terms ? fr.terms[f].obs_col[row] : -1;
  }
}
struct ObsColsDev {
  const int32_t* col[PCLEAN_MAX_TERMS];
};
__global__ void obs_rowmajor_kernel(ObsColsDev oc, int n_terms, int n_rows, int32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (row, term) word per thread: coalesced stores
  if (i >= (size_t)n_rows * PCLEAN_MAX_TERMS) return;
  const int row = (int)(i / PCLEAN_MAX_TERMS), f = (int)(i % PCLEAN_MAX_TERMS);
  out[i] = f < n_terms ? oc.col[f][row] : -1;
}
int pclean_build_obs_rowmajor(pclean_ctx* ctx, const int32_t* const* obs_cols, int n_terms, int n_rows, int32_t* out) {
  if (n_rows <= 0) return PCLEAN_OK;
  ObsColsDev oc{};
  for (int f = 0; f < n_terms && f < PCLEAN_MAX_TERMS; ++f) oc.col[f] = obs_cols[f];
  const size_t n = (size_t)n_rows * PCLEAN_MAX_TERMS;
  hipLaunchKernelGGL(obs_rowmajor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, oc, n_terms, n_rows, out);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// One thread per group: the descriptor the scan kernel consumes, including the lower bound of the maximum
// from the exact score of the rows' current referent (only ever used as a filter, never as a score) and the
// pre-filter cut-off that follows from it.
__global__ void group_desc_kernel(const FastRootDev fr, const ItemsDev it, const ChildrenDev ch, int n_groups,
                                  int32_t* __restrict__ gd, unsigned int* __restrict__ chunk_ctr, double* __restrict__ g_m,
                                  uint64_t* __restrict__ g_U, int32_t* __restrict__ g_res,
                                  unsigned int* __restrict__ scan_stats, const double* __restrict__ pre_score,
                                  const int32_t* __restrict__ pre_obs) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < 8) chunk_ctr[g * WAVE_CTR_STRIDE] = 0u;
  if (g >= 8 && g < 8 + WL_SEGS) chunk_ctr[WL_SEG_CTR(g - 8)] = 0u;  // segment counters of the scan kernel's work list (group_settle_kernel)
  if (g == 8 + WL_SEGS) chunk_ctr[WAVE_WORK_CTR] = 0u;
  unsigned int st_blocks = 0, st_resolved = 0;  // (summed over the wavefront at the end: one atomic per wave)
  if (g < n_groups) {
  const int m_lo = it.grp_off ? it.grp_off[g] : g, m_hi = it.grp_off ? it.grp_off[g + 1] : g + 1;
  const int t = it.grp_off ? it.members[m_lo] : g;
  const int row = it.row ? it.row[t] : t;
  const int excl = it.excl ? it.excl[t] : -1;
  const int ctx0 = it.ctx ? it.ctx[(size_t)t * PCLEAN_MAX_CTX] : 0;
  const int ctx1 = it.ctx ? it.ctx[(size_t)t * PCLEAN_MAX_CTX + 1] : 0;
  int o[PCLEAN_MAX_TERMS];
  if (pre_obs) {  // group_gate_kernel gathered the observed values of this group's row already: 64 contiguous bytes
    static_assert(PCLEAN_MAX_TERMS == 16, "four 16-byte loads");
    const int4* po = reinterpret_cast<const int4*>(pre_obs + (size_t)g * PCLEAN_MAX_TERMS);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 v = po[q];
      o[4 * q] = v.x;
      o[4 * q + 1] = v.y;
      o[4 * q + 2] = v.z;
      o[4 * q + 3] = v.w;
    }
  } else {
    load_row_obs(fr, row, o);
  }
  const bool deleted = excl >= 0 && fr.counts && fr.counts[excl] <= 1;
  double bound = -__builtin_inf(), score_cur = 0.0;
  bool have_cur = false;
  if (excl >= 0 && !deleted && fr.logc_m1) {
    // the current referent's exact score: the same operations in the same order as the scan kernel's own scoring of
    // candidate excl, so the scan kernel takes it from the descriptor when that candidate is its only survivor
    // (group_gate_kernel computed it already when the gate of the new-row branch ran on the groups: same function, same bits)
    score_cur = pre_score ? pre_score[g] : fast_exact_score_mlp(fr, o, ctx0, ctx1, excl, fr.logc_m1[excl] - fr.scal[1]);
    have_cur = true;
    bound = score_cur - 1.0;
  }
  // score of the "new row" candidate (proposal_compiler.jl:221-230): CRP new-table term + log-marginals of the
  // children in plan order — new_score() of enum_kernels.hip; an option list (LEAF node) has none
  double sn = -__builtin_inf();
  if (!fr.is_leaf) {
    const double logden = excl >= 0 ? fr.scal[1] : fr.scal[0];
    double snew = 0.0;
    for (int c = 0; c < ch.n; ++c) {
      size_t idx = (size_t)(it.out_pos ? it.out_pos[t] : t);  // per-item marginals are indexed by the output slot
      if (ch.obs_col[c]) {
        const int oc = ch.obs_col[c][row];
        idx = oc < 0 ? (size_t)ch.n_obs[c] : (size_t)oc;
      }
      snew += ch.arr[c][idx];
    }
    sn = ((deleted ? fr.scal[3] : fr.scal[2]) - logden) + snew;
  }
  bound = fmax(bound, sn);  // the new-row candidate is a candidate too: its exact score bounds the maximum from below
  // Pre-filter cut-off: a candidate whose summed (saturated) edit distance D over the pre-filter terms exceeds the
  // cut scores at most pmax - c_min * D < bound - FIX_CUTOFF <= max - FIX_CUTOFF, i.e. its fixed-point weight is
  // exactly 0 (c_min = smallest cost of one edit, fr.inv_c = 1 / c_min; terms not summed and missing observations
  // only lower the score further).  A bound that would let candidates WAVE_DCUT_OK edits away through (a referent
  // that explains the row badly, no referent at all) is replaced by guess-and-refine in the scan kernel.
  uint32_t cut = CUT_ALL, cut_max = CUT_ALL, refine = 0;
  if (fr.n_pre > 0) {
    const double pmax = excl >= 0 ? fr.prior_max_e : fr.prior_max_n;
    if (bound > -__builtin_inf()) {
      const double x = (pmax - bound + FIX_CUTOFF) * fr.inv_c;
      if (x >= 0.0 && x < (double)(CUT_ALL - 2u)) cut_max = (uint32_t)x + 2u;
    }
    if (cut_max < WAVE_DCUT_OK) {
      cut = cut_max;
    } else {
      cut = WAVE_GUESS;
      refine = 1;
    }
  }
  // ---- RESOLVED groups.  Nine groups in ten end the same way: the rows' current referent is the only candidate within
  // the cut-off, and it (or the new row) outweighs the other by more than 28.5 nats, i.e. the fixed-point total is
  // exactly one unit and every draw returns the same entry.  One THREAD settles such a group here — the two-level
  // pre-filter scan of its three byte rows with the required cut-off, looking for any candidate other than the
  // referent — instead of one WAVEFRONT walking it through the scan kernel's serial phases (~20 us per group per
  // wave whatever the phases contain).  The scan kernel skips the groups flagged here; their rows'
  // draws are written by resolved_draws_kernel, their log-marginals by group_lse_kernel as for every group.
  bool resolved = false;
  int res_val = 0;
  double res_m = 0.0;
  if (g_res && have_cur && !refine && fr.n_pre == 3 && fr.cstride > 0 && !fr.is_leaf) {
    const double hi = fmax(score_cur, sn), lo = fmin(score_cur, sn);
    if (hi > -__builtin_inf() && score_cur != sn && !(lo - hi >= -28.5)) {  // pclean_fixw: the smaller one weighs 0
      const uint8_t *r[3], *mn[3];
      for (int p = 0; p < 3; ++p) {
        const int op = o[fr.pre[p]];
        r[p] = op >= 0 ? fr.terms[fr.pre[p]].comp + (size_t)op * fr.kpad : fr.zero_row;
        mn[p] = op >= 0 ? fr.terms[fr.pre[p]].cmin + (size_t)op * fr.cstride : fr.zero_row;
      }
      const int nquads = fr.kscan >> 4, kblk = (fr.kscan + 63) >> 6;
      bool has_excl = false, other = false;
      for (int kb0 = 0; kb0 < kblk && !other; kb0 += 16) {
        const uint4 a = *reinterpret_cast<const uint4*>(mn[0] + kb0), b = *reinterpret_cast<const uint4*>(mn[1] + kb0),
                    c = *reinterpret_cast<const uint4*>(mn[2] + kb0);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, cw[4] = {c.x, c.y, c.z, c.w};
        for (int e = 0; e < 16 && !other; ++e) {
          const int kb = kb0 + e;
          if (kb >= kblk) break;
          const uint32_t sh = 8u * (e & 3);
          if (((aw[e >> 2] >> sh) & 0xffu) + ((bw[e >> 2] >> sh) & 0xffu) + ((cw[e >> 2] >> sh) & 0xffu) > cut) continue;
          ++st_blocks;
          for (int q = kb * 4; q < kb * 4 + 4 && q < nquads && !other; ++q) {
            const uint4 fa = reinterpret_cast<const uint4*>(r[0])[q], fb = reinterpret_cast<const uint4*>(r[1])[q],
                        fc = reinterpret_cast<const uint4*>(r[2])[q];
            const uint32_t al = fr.alive[q];
            const uint32_t xa[4] = {fa.x, fa.y, fa.z, fa.w}, xb[4] = {fb.x, fb.y, fb.z, fb.w}, xc[4] = {fc.x, fc.y, fc.z, fc.w};
            for (int i = 0; i < 16; ++i) {
              const uint32_t s2 = 8u * (i & 3);
              const uint32_t dsum = ((xa[i >> 2] >> s2) & 0xffu) + ((xb[i >> 2] >> s2) & 0xffu) + ((xc[i >> 2] >> s2) & 0xffu);
              if (dsum <= cut && ((al >> i) & 1u)) {
                if (q * 16 + i == excl)
                  has_excl = true;
                else
                  other = true;
              }
            }
          }
        }
      }
      if (has_excl && !other) {
        resolved = true;
        res_m = hi;
        res_val = sn > score_cur ? PCLEAN_CHOICE_NEW : excl;
      }
    }
  }
  int32_t* d = gd + (size_t)g * GD_STRIDE;
  d[0] = m_lo;
  d[1] = m_hi;
  d[2] = t;
  d[3] = row;
  d[4] = excl;
  d[5] = ctx0;
  d[6] = ctx1;
  d[7] = (deleted ? 1 : 0) | (refine ? 2 : 0) | (have_cur ? 4 : 0) | (resolved ? 8 : 0);
  d[8] = __double2loint(bound);
  d[9] = __double2hiint(bound);
  for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) d[10 + f] = o[f];
  d[26] = __double2loint(sn);
  d[27] = __double2hiint(sn);
  d[28] = (int32_t)cut;
  d[29] = (int32_t)cut_max;
  d[30] = __double2loint(score_cur);
  d[31] = __double2hiint(score_cur);
  if (g_res) g_res[g] = resolved ? res_val : WAVE_RES_UNRESOLVED;
  if (resolved) {
    g_m[g] = res_m;
    g_U[g] = PCLEAN_FIX_ONE;
    st_resolved = 1u;
  }
  }
  if (scan_stats) {  // the byte model's counters: the blocks these scans read, the groups settled here
    for (int sh = 32; sh > 0; sh >>= 1) {
      st_blocks += __shfl_xor(st_blocks, sh, 64);
      st_resolved += __shfl_xor(st_resolved, sh, 64);
    }
    if ((threadIdx.x & 63) == 0 && (st_blocks | st_resolved)) {
      unsigned int* sl = scan_stats + (size_t)(((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & 63) * 32;
      atomicAdd(&sl[1], st_blocks);
      atomicAdd(&sl[3], st_resolved);
    }
  }
}

// Gate of the new-row branch on GROUPS (enum_kernels.hip: gate_new_kernel does it per item through the pair tables): one
// thread per group scores the rows' current referent exactly through the compact byte rows — the value group_desc_kernel
// needs anyway (score_out) — and compares it with the upper bound of the new-row candidate: where that candidate's
// fixed-point weight is exactly 0 the children of the branch are never evaluated.  flag[g] = PCLEAN_CHOICE_NEW: needed.
__global__ void group_gate_kernel(const FastRootDev fr, const ItemsDev it, const GateDev gt, int n_groups,
                                  int32_t* __restrict__ flag, double* __restrict__ score_out,
                                  int32_t* __restrict__ obs_out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const int t = it.grp_off ? it.members[it.grp_off[g]] : g;
  const int row = it.row ? it.row[t] : t;
  const int excl = it.excl ? it.excl[t] : -1;
  const int ctx0 = it.ctx ? it.ctx[(size_t)t * PCLEAN_MAX_CTX] : 0;
  const int ctx1 = it.ctx ? it.ctx[(size_t)t * PCLEAN_MAX_CTX + 1] : 0;
  const bool deleted = excl >= 0 && fr.counts && fr.counts[excl] <= 1;
  bool need = true;
  double sc = 0.0;
  // the observed values of the group's row: gathered once, here, and handed on to group_desc_kernel (obs_out)
  int o[PCLEAN_MAX_TERMS];
  load_row_obs(fr, row, o);
  {
    int4* po = reinterpret_cast<int4*>(obs_out + (size_t)g * PCLEAN_MAX_TERMS);
#pragma unroll
    for (int q = 0; q < 4; ++q) po[q] = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
  }
  if (excl >= 0 && !deleted && fr.logc_m1) {
    sc = fast_exact_score_mlp(fr, o, ctx0, ctx1, excl, fr.logc_m1[excl] - fr.scal[1]);
    double ub = fr.scal[2] - fr.scal[1];
    for (int c = 0; c < gt.n; ++c) {
      if (gt.cache[c]) {
        const int oc = gt.obs_col[c][row];
        ub += gt.cache[c][oc < 0 ? gt.n_obs[c] : oc];
      } else {
        ub += gt.ub[c];
      }
    }
    need = !(ub + 1e-6 < sc - FIX_CUTOFF);
  }
  flag[g] = need ? PCLEAN_CHOICE_NEW : 0;
  score_out[g] = sc;
}
int pclean_launch_group_gate(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const GateDev& gt, int32_t* flag,
                             double* score_out, int32_t* obs_out) {
  if (it.n <= 0) return PCLEAN_OK;
  hipLaunchKernelGGL(group_gate_kernel, dim3((it.n + 255) / 256), dim3(256), 0, ctx->stream, fr, it, gt, it.n, flag, score_out,
                     obs_out);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// draws of the RESOLVED groups (group_desc_kernel): one thread per member position; its group by bisection of grp_off
__global__ void resolved_draws_kernel(int n_items, int n_groups, const int32_t* __restrict__ grp_off,
                                      const int32_t* __restrict__ members, const int32_t* __restrict__ out_pos,
                                      const int32_t* __restrict__ g_res, int n_draws, int draw_is, int draw_ds,
                                      int32_t* __restrict__ draws_out) {
  const int mi = blockIdx.x * blockDim.x + threadIdx.x;
  if (mi >= n_items) return;
  int g = mi, tm = mi;
  if (grp_off) {
    int lo = 0, hi = n_groups - 1;  // largest g with grp_off[g] <= mi
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (grp_off[mid] <= mi)
        lo = mid;
      else
        hi = mid - 1;
    }
    g = lo;
    tm = members[mi];
  }
  const int r = g_res[g];
  if (r == WAVE_RES_UNRESOLVED) return;
  int32_t* dst = draws_out + (size_t)(out_pos ? out_pos[tm] : tm) * draw_is;
  for (int j = 0; j < n_draws; ++j) dst[(size_t)j * draw_ds] = r;
}

// what the scan kernel needs of an item list
struct WaveItems {
  const int32_t* members;   // null: group g = item g
  const int32_t* row;       // identity when null
  const int32_t* rng_row;   // RNG row of an item (default: row + row_offset)
  const int32_t* particle;  // RNG particle of an item (n_draws == 1), default: the draw index
  const int32_t* out_pos;   // where item t writes its outputs (identity when null)
  int64_t row_offset;
  int32_t draw_is, draw_ds;
};

// ---- SETTLED groups, a few lanes each ---------------------------------------------------------------------------------
// Nine groups in ten of a steady-state sweep end the same way: the rows' current referent is the only live candidate
// within the descriptor's cut-off and it (or the new row) outweighs the other by more than 28.5 nats — the fixed-point
// total is exactly one unit and every draw returns the same entry.  The scan kernel spends a wavefront's serial phases
// (~20 us per group per wave) on each of them; one THREAD per group (group_desc_kernel with g_res,
// PCLEAN_RESOLVE_GROUPS) reads its block minima uncoalesced and was measured slower than what it saves.  Here
// SETTLE_L lanes share a group: lane l reads 16 consecutive block minima of each pre-filter term (one coalesced 16-byte
// load per term: a 10^4-candidate table is covered by 11 lanes), walks the fine blocks that pass, and the lanes agree by
// ballot; a settled group is flagged in its descriptor (the scan kernel skips it), gets its maximum / total for
// group_lse_kernel and its members' draws written right here.  Same arithmetic as the scan kernel's filter (integer
// sums against the same cut-off), so the set of settled groups is exactly the set it would have found with one survivor.
#define SETTLE_L 16
#define SETTLE_MAX_BLOCKS 4  // a group with more fine blocks to look at is left to the scan kernel
struct SettleArgs {          // the pre-filter terms' tables resolved on the host: no indexing of FastRootDev::terms[] in the kernel
  const uint8_t* comp[3];
  const uint8_t* cmin[3];
  int32_t pre[3];            // descriptor word of the term's observed value: 10 + term index (-1: no such term)
  int32_t kpad, kscan, cstride;
  const uint16_t* alive;
  const uint8_t* zero_row;
};
__global__ __launch_bounds__(256) void group_settle_kernel(const SettleArgs sa, const WaveItems wi, int n_groups, int n_draws,
                                                           int32_t* __restrict__ gd, double* __restrict__ g_m,
                                                           uint64_t* __restrict__ g_U, int32_t* __restrict__ draws_out,
                                                           unsigned int* __restrict__ scan_stats,
                                                           int32_t* __restrict__ worklist, unsigned int* __restrict__ work_n,
                                                           int seg_cap, int32_t* __restrict__ lz_ns, int32_t* __restrict__ lz_k,
                                                           uint64_t* __restrict__ lz_p, const int32_t* __restrict__ eager_rows) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = tid / SETTLE_L, l = tid % SETTLE_L;
  const int lane = threadIdx.x & 63, base = lane & ~(SETTLE_L - 1);
  const unsigned long long gmask = ((1ull << SETTLE_L) - 1ull) << base;
  // the group's 128-byte descriptor: lane l holds words l and 16 + l (two coalesced loads), fields by shuffle
  int32_t* d = gd + (size_t)(g < n_groups ? g : 0) * GD_STRIDE;
  const int w0 = g < n_groups ? d[l] : 0, w1 = g < n_groups ? d[16 + l] : 0;
  auto word = [&](int w) { return __shfl(w < 16 ? w0 : w1, base + (w & 15), 64); };
  const int flags = word(7), excl = word(4);
  const double sn = __hiloint2double(word(27), word(26)), score_cur = __hiloint2double(word(31), word(30));
  const double hi = fmax(score_cur, sn), lo = fmin(score_cur, sn);
  const uint32_t cut = (uint32_t)word(28);
  int op[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) op[p] = sa.pre[p] >= 0 ? word(sa.pre[p] >= 0 ? sa.pre[p] : 0) : -1;
  // (have_cur, no guess-and-refine, and the lighter of the two explicit candidates weighs exactly 0: pclean_fixw)
  const bool eligible = g < n_groups && (flags & 4) && !(flags & 2) && !(flags & 8) && hi > -__builtin_inf() &&
                        score_cur != sn && !(lo - hi >= -28.5);
  const uint8_t *r[3], *mn[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    r[p] = op[p] >= 0 ? sa.comp[p] + (size_t)op[p] * sa.kpad : sa.zero_row;
    mn[p] = op[p] >= 0 ? sa.cmin[p] + (size_t)op[p] * sa.cstride : sa.zero_row;
  }
  const int kblk = (sa.kscan + 63) >> 6;
  unsigned int st_blocks = 0;
  bool has_excl = false, other = false;
  // ---- level 1: block minima, 16 per lane and round; passing blocks of this lane as a bit mask per round
  uint32_t pass = 0;   // passing blocks of the CURRENT round (bit e: block kb0 + e)
  int kb_base = 0;
  int n_seen = 0;      // fine blocks this group has looked at
  for (int kb_round = 0; kb_round < kblk; kb_round += 16 * SETTLE_L) {  // (wave-uniform: kblk is a launch constant)
    const int kb0 = kb_round + 16 * l;
    pass = 0;
    kb_base = kb0;
    if (eligible && kb0 < kblk) {
      const uint4 a = *reinterpret_cast<const uint4*>(mn[0] + kb0), b = *reinterpret_cast<const uint4*>(mn[1] + kb0),
                  c = *reinterpret_cast<const uint4*>(mn[2] + kb0);
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t sh = 8u * (e & 3);
        const uint32_t dsum = ((aw[e >> 2] >> sh) & 0xffu) + ((bw[e >> 2] >> sh) & 0xffu) + ((cw[e >> 2] >> sh) & 0xffu);
        if (kb0 + e < kblk && dsum <= cut) pass |= 1u << e;
      }
    }
    // ---- level 2: the group's lanes walk its passing blocks together, 4 candidates per lane
    for (int step = 0; step <= SETTLE_MAX_BLOCKS; ++step) {
      const unsigned long long have = __ballot(pass != 0u), oth = __ballot(other);
      if (have == 0ull) break;                      // (no group of this wavefront has a block left in this round)
      const uint32_t mine = (uint32_t)((have & gmask) >> base);
      if (mine == 0u) continue;                      // this group is done with the round
      if (n_seen >= SETTLE_MAX_BLOCKS || (oth & gmask) != 0ull) {  // (group-uniform) too many blocks / already decided: the
                                                                   // group is left to the scan kernel
        other = true;
        pass = 0;
        continue;
      }
      const int src = __builtin_ctz(mine);
      const int kb = __shfl(kb_base + (pass ? __builtin_ctz(pass) : 0), base + src, 64);
      if (l == src) pass &= pass - 1u;
      ++n_seen;
      if (l == 0) ++st_blocks;
      const int k = kb * 64 + 4 * l;
      if (k < sa.kscan) {
        const uint32_t xa = *reinterpret_cast<const uint32_t*>(r[0] + k), xb = *reinterpret_cast<const uint32_t*>(r[1] + k),
                       xc = *reinterpret_cast<const uint32_t*>(r[2] + k);
        const uint32_t al = (uint32_t)sa.alive[k >> 4] >> (k & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t dsum = ((xa >> (8 * i)) & 0xffu) + ((xb >> (8 * i)) & 0xffu) + ((xc >> (8 * i)) & 0xffu);
          if (dsum <= cut && ((al >> i) & 1u)) {
            if (k + i == excl)
              has_excl = true;
            else
              other = true;
          }
        }
      }
    }
    if (pass != 0u) other = true;  // (blocks left over after the walk)
  }
  // the group's lanes agree (every lane of the wavefront takes part in the ballots)
  const unsigned long long b_other = __ballot(other), b_excl = __ballot(has_excl);
  const bool settled = eligible && (b_excl & gmask) != 0ull && (b_other & gmask) == 0ull;
  if (settled) {
    const int res_val = sn > score_cur ? PCLEAN_CHOICE_NEW : excl;
    // lazy draws: a settled group's list is its one live entry with the whole mass (or no entry at all: the new row), and
    // lazy_draw_kernel's search returns it for every random number — no draw is written per member here (the groups of a
    // lazy launch are not cut into pieces); the rows of eager_rows get every draw now, as in the scan kernel
    const bool lazy = lz_ns && !(eager_rows && eager_rows[word(3)] != 0);
    if (l == 0) {
      d[7] = flags | 8;
      g_m[g] = hi;
      g_U[g] = PCLEAN_FIX_ONE;
      if (lz_ns) lz_ns[g] = !lazy ? -1 : (res_val == excl ? 1 : 0);
      if (lazy && res_val == excl) {
        lz_k[(size_t)g * ROOT_LZ_CAP] = excl;
        lz_p[(size_t)g * ROOT_LZ_CAP] = PCLEAN_FIX_ONE;
      }
    }
    const int m_lo = word(0), m_hi = word(1), t0 = word(2);
    const int n_out = lazy ? 0 : (m_hi - m_lo) * n_draws;
    for (int q = l; q < n_out; q += SETTLE_L) {
      const int mi = m_lo + q / n_draws, j = q % n_draws;
      const int tm = wi.members ? wi.members[mi] : t0;
      const size_t to = (size_t)(wi.out_pos ? wi.out_pos[tm] : tm);
      draws_out[to * wi.draw_is + (size_t)j * wi.draw_ds] = res_val;
    }
  }
  // the groups left for the scan kernel: its WORK LIST (the scan kernel used to walk every descriptor to skip the nine in
  // ten settled here).  One atomic per WORKGROUP on one of WL_SEGS counters, 128 bytes apart — a single counter serves its
  // atomics one after the other (~30 ns each: 42 000 of them were 0.48 ms when no group of a launch could be settled) —,
  // each with its own segment of the list; worklist_pack_kernel makes the list dense.
  if (worklist) {
    __shared__ unsigned int wl_cnt[4], wl_base;
    const int wv = threadIdx.x >> 6;
    const bool todo = g < n_groups && l == 0 && !settled;
    const unsigned long long tm = __ballot(todo);
    if (lane == 0) wl_cnt[wv] = (unsigned int)__popcll(tm);
    __syncthreads();
    const int seg = blockIdx.x & (WL_SEGS - 1);
    if (threadIdx.x == 0) {
      const unsigned int tot = wl_cnt[0] + wl_cnt[1] + wl_cnt[2] + wl_cnt[3];
      wl_base = tot ? atomicAdd(work_n + WL_SEG_CTR(seg), tot) : 0u;
    }
    __syncthreads();
    if (todo) {
      unsigned int at = wl_base + (unsigned int)__popcll(tm & ((1ull << lane) - 1ull));
      for (int w = 0; w < wv; ++w) at += wl_cnt[w];
      worklist[(size_t)seg * seg_cap + at] = g;
    }
  }
  if (scan_stats) {
    unsigned int st_settled = (settled && l == 0) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) {
      st_blocks += __shfl_xor(st_blocks, o, 64);
      st_settled += __shfl_xor(st_settled, o, 64);
    }
    if (lane == 0 && (st_blocks | st_settled)) {
      unsigned int* sl = scan_stats + (size_t)((tid >> 6) & 63) * 32;
      atomicAdd(&sl[1], st_blocks);
      atomicAdd(&sl[3], st_settled);
    }
  }
}

// the segments of the work list -> one dense list (segment order; inside a segment the order the workgroups arrived in) and
// its length in chunk_ctr[WAVE_WORK_CTR]; stat (optional): the length once more, for the host's statistics
__global__ __launch_bounds__(1024) void worklist_pack_kernel(const int32_t* __restrict__ seg_list, int seg_cap,
                                                             unsigned int* __restrict__ chunk_ctr, int32_t* __restrict__ dense,
                                                             unsigned int* __restrict__ stat) {
  static_assert(WL_SEGS == 64, "one wavefront reads the segment counters");
  __shared__ unsigned int off[WL_SEGS + 1];
  if (threadIdx.x < 64) {  // exclusive prefix of the 64 segment lengths: lane s holds segment s
    const unsigned int n = chunk_ctr[WL_SEG_CTR(threadIdx.x)];
    unsigned int incl = n;
    for (int sh = 1; sh < 64; sh <<= 1) {
      const unsigned int x = __shfl_up(incl, sh, 64);
      if ((int)threadIdx.x >= sh) incl += x;
    }
    off[threadIdx.x] = incl - n;
    if (threadIdx.x == 63) {
      off[WL_SEGS] = incl;
      chunk_ctr[WAVE_WORK_CTR] = incl;
      if (stat) *stat = incl;
    }
  }
  __syncthreads();
  // every entry of the dense list in one pass: its segment by bisection of the offsets
  const unsigned int total = off[WL_SEGS];
  for (unsigned int i = threadIdx.x; i < total; i += 1024) {
    int lo = 0, hi = WL_SEGS - 1;  // largest s with off[s] <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (off[mid] <= i)
        lo = mid;
      else
        hi = mid - 1;
    }
    dense[i] = seg_list[(size_t)lo * seg_cap + (i - off[lo])];
  }
}

// The per-term pointers of FastRootDev (16 terms x 9 fields) must not live in SGPRs: the compiler hoists them out of
// the group loop and spills them to VGPR lanes (a third of the first version's vector instructions were that spill
// traffic).  FastRootDev is the kernel's FIRST parameter; its terms[] are only read through the kernarg segment
// pointer: once at kernel start into lane-resident registers (lane f holds term f), and with scalar loads in the
// rare ctx-term path.
typedef const __attribute__((address_space(4))) FastRootDev* fr_karg_t;
__device__ __forceinline__ fr_karg_t fr_karg() {
  uint64_t a = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return (fr_karg_t)a;
}
// global-memory views of addresses assembled from lane registers (a plain cast would give flat loads)
typedef const __attribute__((address_space(1))) uint8_t* g_u8_t;
typedef const __attribute__((address_space(1))) uint16_t* g_u16_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4_t* g_u4_t;
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
// exclusive prefix over the wave of a small per-lane count (< 32) + the wave total, from five ballots
__device__ __forceinline__ int wave_excl_prefix5(int cnt, int& total) {
  int ex = 0;
  total = 0;
#pragma unroll
  for (int bit = 0; bit < 5; ++bit) {
    const uint64_t m = __ballot((cnt >> bit) & 1);
    ex += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << bit;
    total += __builtin_popcountll(m) << bit;
  }
  return ex;
}

// -DWAVE_PHASE_CLOCK: per-phase cycle totals over all waves (s_memtime), printed by pclean_launch_root_fast for
// launches of more than 100 000 groups — a measurement build, not the product build.
#ifdef WAVE_PHASE_CLOCK
__device__ unsigned long long g_wave_clk[16];
#define WCLK(i)                                   \
  {                                               \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    clk_acc[i] += now_ - clk_t;                   \
    clk_t = now_;                                 \
  }
#else
#define WCLK(i)
#endif

// CAP = survivors a wave keeps, WPG = waves per workgroup (production shape: <256, 4>; a one-wave 4096-survivor shape
// was tried as a second chance for overflowed groups and is no faster than the generic kernel on flat posteriors).
template <int NT, int CAP, int WPG>
__global__ __launch_bounds__(64 * WPG, WPG == 4 ? WAVE_MIN_WAVES : 1) void fk_root_wave_kernel(
    const FastRootDev fr, const WaveItems wi, uint64_t seed, uint32_t sweep, uint32_t site, int n_draws, int n_groups,
    int chunk, const int32_t* __restrict__ gd, unsigned int* __restrict__ chunk_ctr, double* __restrict__ g_m,
    uint64_t* __restrict__ g_U, int32_t* __restrict__ draws_out, int32_t* __restrict__ overflow_flag,
    unsigned int* __restrict__ overflow_count, int32_t* __restrict__ overflow_list,
    unsigned int* __restrict__ scan_stats, const int32_t* __restrict__ worklist, int32_t* __restrict__ lz_k,
    uint64_t* __restrict__ lz_p, int32_t* __restrict__ lz_ns, const int32_t* __restrict__ eager_rows, int dense_mode) {
  // exact scores and, later, the fixed-point prefix share one array: entry j is converted in place by lane j
  __shared__ uint64_t s_pref[WPG][CAP + 8];
  __shared__ int32_t s_k[WPG][CAP + 8];
  __shared__ int32_t s_blk[WPG][WAVE_BLK_CAP];
  static_assert(CAP <= ROOT_LZ_CAP, "a group's survivor list fits its slot of the lazy-draw lists");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t* pref = s_pref[wave];
  double* scv = reinterpret_cast<double*>(s_pref[wave]);
  int32_t* ksv = s_k[wave];
  int32_t* blk = s_blk[wave];
  const int kpad = fr.kpad, nquads = fr.kscan >> 4, n_terms = fr.n_terms;
  const int nd_eff = n_draws > 0 ? n_draws : 1;
  const int draw_is = wi.draw_is ? wi.draw_is : n_draws, draw_ds = wi.draw_ds ? wi.draw_ds : 1;
  // lane -> (member slot, draw) of the draw phase, fixed for the launch
  const int mem_per_pass = nd_eff <= 64 ? 64 / nd_eff : 0;
  const int slot_l = nd_eff <= 64 ? lane / nd_eff : 0, draw_l = nd_eff <= 64 ? lane - slot_l * nd_eff : 0;
  // lane f < n_terms: term f's compact rows, candidate lengths and typo limit (ctx terms: the zero row)
  uint64_t t_comp = (uint64_t)fr.zero_row, t_clen = (uint64_t)fr.zero_row;
  int t_mt = -1;
  bool t_ctx = false;
  if (lane < n_terms) {
    fr_karg_t F = fr_karg();
    t_ctx = F->terms[lane].ctx_slot >= 0;
    t_mt = F->terms[lane].max_typos;
    if (!t_ctx) {
      t_comp = (uint64_t)F->terms[lane].comp;
      t_clen = (uint64_t)F->terms[lane].clen;
    }
  }
  const uint32_t ctx_mask = (uint32_t)__ballot(t_ctx), mt_mask = (uint32_t)__ballot(lane < n_terms && t_mt >= 0);
  // exact scoring: lane -> (survivor slot j_l of a pass, term f_l); that term's candidate lengths and typo limit
  constexpr int NTP = NT <= 2 ? 2 : NT <= 4 ? 4 : NT <= 8 ? 8 : 16;
  constexpr int SPP = 64 / NTP;  // survivors per pass
  constexpr bool BY_TERM_ONLY = WAVE_EXACT_BY_TERM == 1;  // (measured: taking this path for long lists too is slower)
  const int f_l = lane & (NTP - 1), j_l = lane / NTP;
  const g_u8_t clen_f = (g_u8_t)__shfl((unsigned long long)t_clen, f_l, 64);
  const int mt_f = __shfl(t_mt, f_l, 64);  // -1: no limit (and lanes >= n_terms)
  // XCD-aware chunk hand-out: workgroup b runs on XCD b % 8 (each XCD has its own L2).  The groups arrive sorted by
  // referent, so consecutive groups stream the same byte rows: XCD x owns the x-th contiguous eighth of the groups.
  const int xcd = blockIdx.x & 7;
  // work list (group_settle_kernel): the hand-out below runs over its entries instead of over all groups
  if (worklist) {
    n_groups = (int)__hip_atomic_load(&chunk_ctr[WAVE_WORK_CTR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chunk = max(1, min(chunk, n_groups / max((int)(gridDim.x * WPG), 1)));
  }
  // (chunk = consecutive groups a wave takes at a time: WAVE_CHUNK for the large launches — a wave reuses its survivor list
  // across neighbours —, fewer for short lists: the nested slots of a new-row branch hold a few hundred groups, and eight
  // at a time left all but 32 waves of the chip idle behind a serial chain of ~15 us per group)
  const int per = (((n_groups + 7) >> 3) + chunk - 1) / chunk * chunk;  // groups per XCD, whole chunks
  const int chunks_per = per / chunk;
  int steal = 0;  // chunks come from the counter of XCD (xcd + steal) & 7; 8 = everything is handed out
  auto grab = [&]() -> int {  // the returned value is valid in lane 0 and looked at by resolve() only
    int c = 0;
    if (lane == 0) c = (int)atomicAdd(&chunk_ctr[((xcd + steal) & 7) * WAVE_CTR_STRIDE], 1u);
    return c;
  };
  auto resolve = [&](int raw_lane0, int& g_lo, int& g_hi) {
    int c = __builtin_amdgcn_readfirstlane(raw_lane0);
    for (;;) {
      const int x = (xcd + steal) & 7;
      if (c < chunks_per) {
        g_lo = x * per + c * chunk;
        g_hi = min(g_lo + chunk, min((x + 1) * per, n_groups));
        if (g_lo < g_hi) return;
      }
      if (++steal >= 8) {
        g_lo = g_hi = 0;
        return;
      }
      int r = 0;
      if (lane == 0) r = (int)atomicAdd(&chunk_ctr[((xcd + steal) & 7) * WAVE_CTR_STRIDE], 1u);
      c = __builtin_amdgcn_readfirstlane(r);
    }
  };
  int dense_skip = 0;  // scans left that go straight to the streaming filter (see the scan)
  const bool dense_on = dense_mode != 0;
  // cache of the last scan (per wave): pre-filter observed values, scanned cut-off, survivors in ksv
  int c_o0 = -2, c_o1 = -2, c_o2 = -2, c_ns = 0;
  uint32_t c_cut = 0;
  bool c_valid = false;
  // the previous group of this wave: a group with the same (referent, ctx, observed values, bounds) — the pieces a
  // large group is split into (make_item_groups) — reuses its survivors, prefix and totals outright
  int dvp = 0, p_ns = 0;
  bool p_valid = false, p_over = false;
  double p_m = 0.0;
  uint64_t p_U = 0;

#ifdef WAVE_PHASE_CLOCK
  unsigned long long clk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long clk_t = __builtin_readcyclecounter();
  unsigned long long cnt_acc[6] = {0, 0, 0, 0, 0, 0};  // full scans, cached scans, survivors, -, -, -
#endif
  unsigned int st_scans = 0, st_blocks = 0, st_terms = 0, st_lazy = 0;  // bench.py's byte model: what the launch really read / wrote
  int g = 0, g_end = 0;
  resolve(grab(), g, g_end);
  int raw_next = steal < 8 ? grab() : 0;
  // entries [wl_lo, wl_lo + chunk) of the work list, one per lane (chunk <= WAVE_CHUNK <= 64); gid(x) = group of entry x
  int wl_lo = g, wl_v = 0, wl_lo_n = 0, wl_v_n = 0;
  if (worklist && g < g_end) wl_v = g + lane < g_end ? worklist[g + lane] : 0;
  auto gid = [&](int x) -> int { return worklist ? __builtin_amdgcn_readlane(wl_v, x - wl_lo) : x; };
  // descriptor of group x, lane i = its i-th word: scalar base + lane offset (a per-lane 64-bit base would be hoisted out
  // of the group loop and spilled)
  auto desc_word = [&](int x) -> int {
    uint64_t base = (uint64_t)gd + (uint64_t)(uint32_t)x * (uint64_t)(GD_STRIDE * 4);
    uint32_t off;  // byte offset of this lane's word, formed here (held across the loop it is spilled to scratch)
    asm volatile(
        "v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_and_b32 %0, %2, %0\n\tv_lshlrev_b32 %0, 2, %0"
        : "=v"(off), "+s"(base)
        : "n"(GD_STRIDE - 1));
    return *(const __attribute__((address_space(1))) int32_t*)(base + off);
  };
  static_assert((GD_STRIDE & (GD_STRIDE - 1)) == 0, "descriptor stride: a power of two");
  int dv = g < g_end ? desc_word(gid(g)) : 0;
  while (g < g_end) {
    const int g_id = gid(g);  // the group itself (g is its place in the hand-out: an entry of the work list, or the group)
    // ---- next group (possibly the first of the next chunk): its descriptor is requested right away ---------------
    int gn = g + 1, gn_end = g_end;
    bool new_chunk = false;
    if (gn >= g_end) {
      if (steal < 8) {
        resolve(raw_next, gn, gn_end);
        raw_next = steal < 8 ? grab() : 0;
        new_chunk = true;
        if (worklist && gn < gn_end) {
          wl_lo_n = gn;
          wl_v_n = gn + lane < gn_end ? worklist[gn + lane] : 0;
        }
      } else {
        gn = gn_end = 0;
      }
    }
    const int gn_id = !worklist ? gn : (new_chunk ? __builtin_amdgcn_readlane(wl_v_n, 0) : (gn < gn_end ? gid(gn) : 0));
    const int dvn = gn < gn_end ? desc_word(gn_id) : 0;
    WCLK(0)  // chunk hand-out + descriptor request
    // ---- descriptor -> wave-uniform registers ----------------------------------------------------------------------
    const int m_lo = __builtin_amdgcn_readlane(dv, 0), m_hi = __builtin_amdgcn_readlane(dv, 1);
    const int t = __builtin_amdgcn_readlane(dv, 2);
    const int excl = __builtin_amdgcn_readlane(dv, 4);
    const int flags = __builtin_amdgcn_readlane(dv, 7);
    if (flags & 8) {  // settled by group_desc_kernel (its rows' draws: resolved_draws_kernel): nothing to do here
      g = gn;
      g_end = gn_end;
      dv = dvn;
      if (new_chunk) {
        wl_lo = wl_lo_n;
        wl_v = wl_v_n;
      }
      continue;
    }
    const bool deleted = (flags & 1) != 0;
    bool refine = (flags & 2) != 0;
    const double bound = __hiloint2double(__builtin_amdgcn_readlane(dv, 9), __builtin_amdgcn_readlane(dv, 8));
    // score of the "new row" candidate (index n, last in natural order; -inf for an option list)
    const double sn = __hiloint2double(__builtin_amdgcn_readlane(dv, 27), __builtin_amdgcn_readlane(dv, 26));
    uint32_t cut = (uint32_t)__builtin_amdgcn_readlane(dv, 28);
    const uint32_t cut_max = (uint32_t)__builtin_amdgcn_readlane(dv, 29);
    const bool excluded = excl >= 0;
    // lane f: observed value of term f and the address of its byte row; on_mask: terms with an observation
    const int of_l = __shfl(dv, (lane & 15) + 10, 64);
    const uint32_t on_mask = (uint32_t)__ballot(lane < n_terms && of_l >= 0);
    const uint64_t t_row = t_comp + (uint64_t)(uint32_t)(of_l < 0 ? 0 : of_l) * (uint64_t)(uint32_t)kpad;
    // observed values of the pre-filter terms (-1: missing / fewer than three terms)
    const int po0 = fr.n_pre > 0 ? __builtin_amdgcn_readlane(dv, 10 + fr.pre[0]) : -1;
    const int po1 = fr.n_pre > 1 ? __builtin_amdgcn_readlane(dv, 10 + fr.pre[1]) : -1;
    const int po2 = fr.n_pre > 2 ? __builtin_amdgcn_readlane(dv, 10 + fr.pre[2]) : -1;

    const int n_mem = m_hi - m_lo;
    // raw attribute words of a member item (row, RNG row, particle, output position): requested together, combined
    // only where they are used (requesting them ahead of the draw phase was measured: no gain)
    auto item_attrs = [&](int tm, int& w0, int& w1, int& w2, int& w3) {
      w0 = w1 = w2 = w3 = tm;
      if (wi.row) w0 = wi.row[tm];
      if (wi.rng_row) w1 = wi.rng_row[tm];
      if (wi.particle) w2 = wi.particle[tm];
      if (wi.out_pos) w3 = wi.out_pos[tm];
    };

    // ---- pre-filter scan at cut-off `want` (or the cached list when it covers it): survivors -> ksv, ascending ----
    int ns = 0;
    auto scan = [&](uint32_t want) {
      if (c_valid && po0 == c_o0 && po1 == c_o1 && po2 == c_o2 && want <= c_cut && c_ns <= CAP) {
        ns = c_ns;
        cut = c_cut;
#ifdef WAVE_PHASE_CLOCK
        cnt_acc[1] += 1;
#endif
        return;
      }
      const uint64_t zr = (uint64_t)fr.zero_row;
      const g_u8_t r0 = (g_u8_t)(po0 >= 0 ? readlane64(t_row, fr.pre[0]) : zr);
      const g_u8_t r1 = (g_u8_t)(po1 >= 0 ? readlane64(t_row, fr.pre[1]) : zr);
      const g_u8_t r2 = (g_u8_t)(po2 >= 0 ? readlane64(t_row, fr.pre[2]) : zr);
#ifdef WAVE_PHASE_CLOCK
      cnt_acc[0] += 1;
#endif
      const g_u8_t alive = (g_u8_t)(uint64_t)fr.alive;
      st_scans += 1u;
      uint32_t cs = min(want + WAVE_SLACK, CUT_ALL);
      // coarse level: the block-minimum rows (one byte per 64 candidates).  sum of the three minima > cut-off: no
      // candidate of the block can pass (the minima bound every candidate's bytes from below) — the block's 3 x 64
      // bytes are never read.  A scan reads ~3 x kpad / 64 bytes + the few blocks that hold a near candidate.
      const uint32_t cst = (uint32_t)fr.cstride;
      fr_karg_t FK = fr_karg();  // (the block-minimum bases: scalar loads from the kernarg segment, once per scan)
      const g_u8_t m0 = (g_u8_t)(po0 >= 0 ? (uint64_t)FK->terms[fr.pre[0]].cmin + (uint64_t)((uint32_t)po0 * cst) : zr);
      const g_u8_t m1 = (g_u8_t)(po1 >= 0 ? (uint64_t)FK->terms[fr.pre[1]].cmin + (uint64_t)((uint32_t)po1 * cst) : zr);
      const g_u8_t m2 = (g_u8_t)(po2 >= 0 ? (uint64_t)FK->terms[fr.pre[2]].cmin + (uint64_t)((uint32_t)po2 * cst) : zr);
      const int kblk = (fr.kscan + 63) >> 6;
      for (;;) {
        // a byte of (c0 + c1 + c2 + addc) has bit 7 set iff its summed distance exceeds cs (sums <= 126: no carries)
        const uint32_t addc = 0x01010101u * (127u - cs);
        ns = 0;
        bool fine_done = false;
        // (dense_skip: the wave's last two-level scan found three blocks in four passing — a table of near-identical rows, the
        // Measure slot — so the coarse level is a round trip for nothing: the next scans stream the rows whole, then it is tried again)
        if (cst != 0u && dense_skip > 0) --dense_skip;
        else if (cst != 0u) {
          int nb = 0;  // passing blocks -> blk[], ascending
          for (int kb0 = 0; kb0 < kblk; kb0 += 64) {
            const int kb = kb0 + lane;
            const bool pass = kb < kblk && (uint32_t)m0[kb] + (uint32_t)m1[kb] + (uint32_t)m2[kb] <= cs;
            const uint64_t mk = __ballot(pass);
            if (pass) {
              const int pos = nb + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
              if (pos < WAVE_BLK_CAP) blk[pos] = kb;
            }
            nb += __builtin_popcountll(mk);
          }
          __builtin_amdgcn_wave_barrier();
#ifdef WAVE_PHASE_CLOCK
          cnt_acc[4] += nb <= WAVE_BLK_CAP ? 1 : 0;
          cnt_acc[5] += (unsigned long long)min(nb, 100000);
#endif
          if (dense_on && nb * 4 >= kblk * 3 && kblk >= 8) {
            dense_skip = 64;
          } else if (nb <= WAVE_BLK_CAP) {
            fine_done = true;
            st_blocks += (unsigned int)nb;
            for (int i0 = 0; i0 < nb; i0 += 16) {  // 16 blocks x 4 quads per pass, in ascending candidate order
              const int bi = i0 + (lane >> 2);
              const int q = bi < nb ? blk[bi] * 4 + (lane & 3) : nquads;
              uint32_t m16 = 0;
              if (q < nquads) {
                const u32x4_t ca = *(g_u4_t)(r0 + ((uint32_t)q << 4)), cb = *(g_u4_t)(r1 + ((uint32_t)q << 4)),
                              cc = *(g_u4_t)(r2 + ((uint32_t)q << 4));
                const uint32_t al = *(g_u16_t)(alive + ((uint32_t)q << 1));
                const uint32_t tw[4] = {ca.x + cb.x + cc.x + addc, ca.y + cb.y + cc.y + addc, ca.z + cb.z + cc.z + addc,
                                        ca.w + cb.w + cc.w + addc};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const uint32_t zz = ~tw[i] & 0x80808080u;
                  m16 |= (((zz >> 7) | (zz >> 14) | (zz >> 21) | (zz >> 28)) & 0xfu) << (4 * i);
                }
                m16 &= al;
              }
              if (__ballot(m16 != 0) == 0ull) continue;
              int total;
              int pos = ns + wave_excl_prefix5(__builtin_popcount(m16), total);
              for (uint32_t mm = m16; mm; mm &= mm - 1) {
                if (pos < CAP) ksv[pos] = (q << 4) + __builtin_ctz(mm);
                ++pos;
              }
              ns += total;
            }
          }
        }
        if (!fine_done) st_blocks += (unsigned int)kblk;
        if (!fine_done)
        for (int q0 = 0; q0 < nquads; q0 += 64 * WAVE_RB) {
          u32x4_t ca[WAVE_RB], cb[WAVE_RB], cc[WAVE_RB];
          uint32_t al[WAVE_RB];
#pragma unroll
          for (int r = 0; r < WAVE_RB; ++r) {  // every load of the batch first
            const uint32_t qq = (uint32_t)min(q0 + r * 64 + lane, nquads - 1);
            ca[r] = *(g_u4_t)(r0 + (qq << 4));
            cb[r] = *(g_u4_t)(r1 + (qq << 4));
            cc[r] = *(g_u4_t)(r2 + (qq << 4));
            al[r] = *(g_u16_t)(alive + (qq << 1));
          }
          // does the batch hold a survivor at all?  Wave-uniform, and mostly not.
          bool any = false;
#pragma unroll
          for (int r = 0; r < WAVE_RB; ++r) {
            if (q0 + r * 64 + lane >= nquads) al[r] = 0u;
            uint32_t tt = (ca[r].x + cb[r].x + cc[r].x + addc) & (ca[r].y + cb[r].y + cc[r].y + addc) &
                          (ca[r].z + cb[r].z + cc[r].z + addc) & (ca[r].w + cb[r].w + cc[r].w + addc);
            any |= ((tt & 0x80808080u) != 0x80808080u) && al[r] != 0u;
          }
          if (__ballot(any) == 0ull) continue;
#pragma unroll
          for (int r = 0; r < WAVE_RB; ++r) {
            const int q = q0 + r * 64 + lane;
            // 16-bit mask of the bytes of (x, y, z, w) whose bit 7 is clear: candidate 4 w + e passes
            auto pass16 = [&](uint32_t x, uint32_t y, uint32_t z, uint32_t w) -> uint32_t {
              const uint32_t tw[4] = {x, y, z, w};
              uint32_t m16 = 0;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const uint32_t zz = ~tw[i] & 0x80808080u;
                m16 |= (((zz >> 7) | (zz >> 14) | (zz >> 21) | (zz >> 28)) & 0xfu) << (4 * i);
              }
              return m16 & al[r];  // padding and free slots never survive
            };
            auto append = [&](uint32_t m16, int32_t* list, int cap, int& n) {
              if (__ballot(m16 != 0) == 0ull) return;
              int total;
              int pos = n + wave_excl_prefix5(__builtin_popcount(m16), total);
              for (uint32_t mm = m16; mm; mm &= mm - 1) {
                if (pos < cap) list[pos] = (q << 4) + __builtin_ctz(mm);
                ++pos;
              }
              n += total;
            };
            append(pass16(ca[r].x + cb[r].x + cc[r].x + addc, ca[r].y + cb[r].y + cc[r].y + addc,
                          ca[r].z + cb[r].z + cc[r].z + addc, ca[r].w + cb[r].w + cc[r].w + addc),
                   ksv, CAP, ns);
          }
        }
        if (ns > CAP && cs > want) {  // the slack alone overflowed the list: once more without it
          cs = want;
          continue;
        }
        break;
      }
      __builtin_amdgcn_wave_barrier();
      c_o0 = po0;
      c_o1 = po1;
      c_o2 = po2;
      c_cut = cs;
      c_ns = ns;
      c_valid = true;
      cut = cs;
    };

    bool over = false;
    double m = -__builtin_inf();
    uint64_t U = 0;
    const bool same = p_valid && __ballot(lane >= 4 && lane < 30 && dv != dvp) == 0ull;
    if (same) {
      ns = p_ns;
      over = p_over;
      m = p_m;
      U = p_U;
    } else {
    WCLK(1)  // descriptor decode
    for (;;) {
      scan(cut);
      WCLK(2)  // scan
#ifdef WAVE_PHASE_CLOCK
      cnt_acc[2] += (unsigned long long)min(ns, CAP);
#endif
      if (ns > CAP) {
        over = true;
        break;
      }
      if (refine && ns == 0) {  // nothing within the guessed cut-off: widen it
        if (cut >= cut_max) break;
        cut = min(2u * cut + 2u, cut_max);
        continue;
      }
      // ---- the only survivor is the rows' current referent (nine groups in ten): its exact score is in the descriptor
      // (group_desc_kernel computed it for the bound, bit for bit what the code below would compute) — no gather at all
      bool scored = false;
      if (ns == 1 && (flags & 4) != 0) {
        const int k0 = __builtin_amdgcn_readfirstlane(ksv[0]);
        if (k0 == excl) {
          if (lane == 0) scv[0] = __hiloint2double(__builtin_amdgcn_readlane(dv, 31), __builtin_amdgcn_readlane(dv, 30));
          scored = true;
        }
      }
      if (!scored) st_terms += (unsigned int)(ns * n_terms);
#if WAVE_EXACT_BY_TERM != 0
      // ---- exact fp64 scores of ksv[0..ns) -> scv: lane (j, f) = (survivor j of the pass, term f).  Every lane
      // issues ITS byte-distance and length loads at once and then its density load — two memory round trips per
      // pass whatever the number of terms (a per-candidate loop over the terms serialises them: the kernel is bound
      // by dependent round trips, not by bytes).  The fp64 additions then follow plan order through lane shuffles
      // (the operation order of candidate_score(), enum_kernels.hip).
      if (!scored && (BY_TERM_ONLY || ns <= SPP)) {
        const double* prior = (excluded && fr.prior_e) ? fr.prior_e : fr.prior_n;
        const bool f_on = ((on_mask >> f_l) & 1u) != 0u, f_ctx = ((ctx_mask >> f_l) & 1u) != 0u;
        const g_u8_t row_f = (g_u8_t)__shfl((unsigned long long)t_row, f_l, 64);
        const int of_f = __shfl(dv, 10 + f_l, 64);
        for (int base = 0; base < ns; base += SPP) {
          const int j = base + j_l;
          const uint32_t k = (uint32_t)ksv[j < ns ? j : base];
          double b = 0.0;
          if (f_l == 0) b = prior[k];
          int dd = 0, LL = 0;
          if (!f_ctx) {  // lanes >= n_terms and missing observations hold the zero row
            dd = row_f[k];
            LL = clen_f[k];
          } else if (f_on) {  // latent value through fn[ctx][.] (a JuliaNode of an earlier block's choice)
            fr_karg_t F = fr_karg();
            const int c = F->terms[f_l].ctx_slot == 0 ? __builtin_amdgcn_readlane(dv, 5) : __builtin_amdgcn_readlane(dv, 6);
            const int val = F->terms[f_l].fn[(size_t)c * F->terms[f_l].fn_nb + F->terms[f_l].cand_col[k]];
            dd = F->terms[f_l].pair[(size_t)of_f * F->terms[f_l].n_lat + val];
            LL = F->terms[f_l].lat_len[val];
          }
          if (dd == PRE_CLAMP && f_on && !f_ctx) {  // saturated: the true distance
            fr_karg_t F = fr_karg();
            dd = F->terms[f_l].pair[(size_t)of_f * F->terms[f_l].n_lat + F->terms[f_l].cand_col[k]];
          }
          double l = fr.atd[(uint32_t)(LL * fr.atd_stride + dd)];
          if (mt_f >= 0 && dd > mt_f) l = ADD_TYPOS_IMPOSSIBLE;
          if ((int)k == excl) b = deleted ? -__builtin_inf() : fr.logc_m1[excl] - fr.scal[1];
          const int l0 = lane & ~(NTP - 1);
#pragma unroll
          for (int f = 0; f < NT; ++f) {
            const double lf = __shfl(l, l0 + f, 64);
            if ((on_mask >> f) & 1u) b += lf;  // a missing observation contributes nothing (add_typos.jl:51-53)
          }
          if (f_l == 0 && j < ns) scv[j] = b;
        }
      }
#endif
#if WAVE_EXACT_BY_TERM != 1
      // ---- exact fp64 scores of ksv[0..ns), one candidate per lane per pass -> scv.  The loads of a chunk of
      // terms are in flight together (byte distance + length, then the density table); the fp64 additions follow
      // plan order (the operation order of candidate_score(), enum_kernels.hip).
      if (!scored && !BY_TERM_ONLY && (WAVE_EXACT_BY_TERM == 0 || ns > SPP)) {
        const double* prior = (excluded && fr.prior_e) ? fr.prior_e : fr.prior_n;
        for (int base = 0; base < ns; base += 64) {
          const int j = base + lane;
          const uint32_t k = (uint32_t)(j < ns ? ksv[j] : ksv[base]);
          double b = prior[k];
          if ((int)k == excl) b = deleted ? -__builtin_inf() : fr.logc_m1[excl] - fr.scal[1];
#pragma unroll
          for (int f0 = 0; f0 < NT; f0 += WAVE_TC) {
            int dd[WAVE_TC], LL[WAVE_TC];
#pragma unroll
            for (int u = 0; u < WAVE_TC; ++u) {
              const int f = f0 + u;
              dd[u] = 0;
              LL[u] = 0;
              if (f < NT) {
                if ((ctx_mask >> f) & 1u) {  // latent value through fn[ctx][.] (a JuliaNode of an earlier block's choice)
                  if ((on_mask >> f) & 1u) {
                    fr_karg_t F = fr_karg();
                    const int of = __builtin_amdgcn_readlane(dv, 10 + f);
                    const int c = F->terms[f].ctx_slot == 0 ? __builtin_amdgcn_readlane(dv, 5) : __builtin_amdgcn_readlane(dv, 6);
                    const int val = F->terms[f].fn[(size_t)c * F->terms[f].fn_nb + F->terms[f].cand_col[k]];
                    dd[u] = F->terms[f].pair[(size_t)of * F->terms[f].n_lat + val];
                    LL[u] = F->terms[f].lat_len[val];
                  }
                } else {  // lanes >= n_terms and missing observations hold the zero row
                  dd[u] = ((g_u8_t)readlane64(t_row, f))[k];
                  LL[u] = ((g_u8_t)readlane64(t_clen, f))[k];
                }
              }
            }
#pragma unroll
            for (int u = 0; u < WAVE_TC; ++u) {
              const int f = f0 + u;
              if (f < NT && dd[u] == PRE_CLAMP && ((on_mask & ~ctx_mask) >> f) & 1u) {  // saturated: the true distance
                fr_karg_t F = fr_karg();
                const int of = __builtin_amdgcn_readlane(dv, 10 + f);
                dd[u] = F->terms[f].pair[(size_t)of * F->terms[f].n_lat + F->terms[f].cand_col[k]];
              }
            }
            double av[WAVE_TC];
#pragma unroll
            for (int u = 0; u < WAVE_TC; ++u) av[u] = fr.atd[(uint32_t)(LL[u] * fr.atd_stride + dd[u])];
#pragma unroll
            for (int u = 0; u < WAVE_TC; ++u) {
              const int f = f0 + u;
              if (f < NT && ((on_mask >> f) & 1u)) {  // a missing observation contributes nothing (add_typos.jl:51-53)
                double l = av[u];
                if ((mt_mask >> f) & 1u) l = dd[u] > __builtin_amdgcn_readlane(t_mt, f) ? ADD_TYPOS_IMPOSSIBLE : l;
                b += l;
              }
            }
          }
          if (j < ns) scv[j] = b;
        }
      }
#endif
      __builtin_amdgcn_wave_barrier();
      WCLK(3)  // exact scores
      if (!refine) break;
      // refine: the best survivor's exact score (and the descriptor's bound) give the cut-off actually required
      double best = bound;
      for (int base = 0; base < ns; base += 64) {
        const int j = base + lane;
        if (j < ns) best = fmax(best, scv[j]);
      }
      best = wave_max64(best);
      refine = false;
      uint32_t need = CUT_ALL;
      {
        const double x = ((excluded ? fr.prior_max_e : fr.prior_max_n) - best + FIX_CUTOFF) * fr.inv_c;
        if (x >= 0.0 && x < (double)(CUT_ALL - 2u)) need = (uint32_t)x + 2u;
      }
      need = min(need, cut_max);
      if (need <= cut) break;  // the scanned list is a superset of the required one
      cut = need;
    }
    if (!over) {
      // ---- maximum, fixed-point weights, inclusive prefix: survivors in ascending order, then the new row (entry ns)
      if (lane == 0) scv[ns] = sn;
      __builtin_amdgcn_wave_barrier();
      const int n_e = ns + 1;
      for (int base = 0; base < n_e; base += 64) {
        const int j = base + lane;
        if (j < n_e) m = fmax(m, scv[j]);
      }
      m = wave_max64(m);
      uint64_t carry = 0;
      for (int base = 0; base < n_e; base += 64) {
        const int j = base + lane;
        const uint64_t u = (j < n_e && m != -__builtin_inf()) ? pclean_fixw(scv[j] - m) : 0ull;
        unsigned long long incl = u;
        const int n_here = min(64, n_e - base);  // entries of this pass: the usual list is two or three long
        for (int sh = 1; sh < n_here; sh <<= 1) {
          const unsigned long long x = __shfl_up(incl, sh, 64);
          if (lane >= sh) incl += x;
        }
        if (j < n_e) pref[j] = carry + incl;
        carry += __shfl(incl, n_here - 1, 64);
      }
      U = carry;
      __builtin_amdgcn_wave_barrier();
    }
    WCLK(4)  // weights
    p_ns = ns;
    p_over = over;
    p_m = m;
    p_U = U;
    }
    dvp = dv;
    p_valid = true;
    if (over) {  // flat posterior: these items are re-run over all candidates (flags are pre-zeroed); with an
      // overflow list the re-run kernel consumes it straight from the device (no count read-back)
      const int n_mem_o = m_hi - m_lo;
      int base_o = 0;
      if (lane == 0) {
        base_o = (int)atomicAdd(overflow_count, (unsigned int)n_mem_o);
        g_m[g_id] = __builtin_nan("");
        if (lz_ns) lz_ns[g_id] = -1;  // (lazy draws: the re-run writes every draw of these items)
      }
      base_o = __builtin_amdgcn_readfirstlane(base_o);
      if (n_mem_o == 1) {
        if (lane == 0) {
          overflow_flag[wi.out_pos ? wi.out_pos[t] : t] = PCLEAN_CHOICE_NEW;  // marker understood by compact_new_kernel
          if (overflow_list) overflow_list[base_o] = t;
        }
      } else {
        for (int mi = m_lo + lane; mi < m_hi; mi += 64) {
          const int tm = wi.members[mi];
          overflow_flag[wi.out_pos ? wi.out_pos[tm] : tm] = PCLEAN_CHOICE_NEW;
          if (overflow_list) overflow_list[base_o + (mi - m_lo)] = tm;
        }
      }
    } else {
      if (lane == 0) {
        g_m[g_id] = m;
        g_U[g_id] = U;
      }
      // ---- LAZY draws (RootExtra): the list and its prefix are kept, one draw per row follows the final choice ---------
      bool lazy = false;
      if (lz_ns) {
        lazy = n_draws > 0 && !(eager_rows && eager_rows[__builtin_amdgcn_readlane(dv, 3)] != 0);
        if (lazy) {  // (scalar bases + lane offsets: per-lane 64-bit bases would be held across the group loop and spilled)
          uint64_t bk = (uint64_t)lz_k + (uint64_t)(uint32_t)g_id * (uint64_t)(ROOT_LZ_CAP * 4);
          uint64_t bp = (uint64_t)lz_p + (uint64_t)(uint32_t)g_id * (uint64_t)(ROOT_LZ_CAP * 8);
          asm volatile("" : "+s"(bk), "+s"(bp));
          for (int j = lane; j < ns; j += 64) {
            ((__attribute__((address_space(1))) int32_t*)bk)[j] = ksv[j];
            ((__attribute__((address_space(1))) uint64_t*)bp)[j] = pref[j];
          }
        }
        if (lane == 0) lz_ns[g_id] = lazy ? ns : -1;
        if (lazy) st_lazy += (unsigned int)ns;
      }
      // ---- draws of every (member item, draw) pair of the group ------------------------------------------------------
      if (n_draws > 0 && !lazy) {
        const int res_new = fr.is_leaf ? fr.n_cand - 1 : PCLEAN_CHOICE_NEW;
        // A DECIDED group: the total is exactly one unit (PCLEAN_FIX_ONE) — the maximum's own weight — so every other
        // entry has fixed-point weight 0 and every draw, whatever its random number (x < U), returns the first entry
        // with a non-zero prefix.  No Philox, no multiply-high, no per-lane search, no RNG attributes: most groups
        // (the rows whose referent explains them 28.5 nats better than anything else) are of this kind.
        const bool decided = U == PCLEAN_FIX_ONE;
#ifdef WAVE_PHASE_CLOCK
        cnt_acc[3] += decided ? 1 : 0;
#endif
        int dec_res = res_new;
        if (decided) {
          int a = 0, b = ns;  // wave-uniform: smallest index with a non-zero prefix
          while (a < b) {
            const int mid = (a + b) >> 1;
            if (pref[mid] > 0ull)
              b = mid;
            else
              a = mid + 1;
          }
          if (a < ns) dec_res = ksv[a];
        }
        auto finish_draw = [&](int w0, int w1, int w2, int w3, int j) {
          int32_t res = res_new;
          if (U != 0) {
            const uint32_t rng_row = wi.rng_row ? (uint32_t)w1 : (uint32_t)((int64_t)w0 + wi.row_offset);
            const uint32_t pid = wi.particle ? (uint32_t)w2 : (uint32_t)j;
            const uint64_t x = pclean_mulhi64(pclean_rand64(seed, rng_row, site, pid, sweep), U);
            int a = 0, b = ns;  // smallest index with prefix > x (index ns = the new row)
            while (a < b) {
              const int mid = (a + b) >> 1;
              if (pref[mid] > x)
                b = mid;
              else
                a = mid + 1;
            }
            if (a < ns) res = ksv[a];
          }
          draws_out[(size_t)w3 * draw_is + (size_t)j * draw_ds] = res;
        };
        auto one_output = [&](int tm, int j) {
          if (decided) {
            const int op = wi.out_pos ? wi.out_pos[tm] : tm;
            draws_out[(size_t)op * draw_is + (size_t)j * draw_ds] = dec_res;
          } else {
            int w0, w1, w2, w3;
            item_attrs(tm, w0, w1, w2, w3);
            finish_draw(w0, w1, w2, w3, j);
          }
        };
        if (mem_per_pass > 0) {
          for (int m0 = 0; m0 < n_mem; m0 += mem_per_pass) {
            const int ms = m0 + slot_l;
            if (slot_l < mem_per_pass && ms < n_mem) one_output(n_mem == 1 ? t : wi.members[m_lo + ms], draw_l);
          }
        } else {
          const int n_out = n_mem * nd_eff;
          for (int q = lane; q < n_out; q += 64) one_output(n_mem == 1 ? t : wi.members[m_lo + q / nd_eff], q % nd_eff);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    WCLK(5)  // outputs + draws
    g = gn;
    g_end = gn_end;
    dv = dvn;
    if (new_chunk) {
      wl_lo = wl_lo_n;
      wl_v = wl_v_n;
    }
#ifdef WAVE_PHASE_CLOCK
    clk_acc[7] += 1;
#endif
  }
  if (scan_stats && lane == 0) {  // 64 slots a cache line apart: same-line atomics are served one after the other
    unsigned int* sl = scan_stats + (size_t)((blockIdx.x * WPG + wave) & 63) * 32;
    atomicAdd(&sl[0], st_scans);
    atomicAdd(&sl[1], st_blocks);
    atomicAdd(&sl[2], st_terms);
    if (st_lazy) atomicAdd(&sl[4], st_lazy);
  }
#ifdef WAVE_PHASE_CLOCK
  clk_acc[6] = __builtin_readcyclecounter() - clk_t;
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_wave_clk[i], clk_acc[i]);
  if (lane == 0) atomicAdd(&g_wave_clk[8], 1ull);
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_wave_clk[9 + i], cnt_acc[i]);
#endif
}

// log-sum-exp of every group from the (maximum, fixed-point total) the scan kernel left, scattered to the member items
__global__ void group_lse_kernel(int n_groups, const int32_t* __restrict__ grp_off, const int32_t* __restrict__ members,
                                 const int32_t* __restrict__ out_pos, const double* __restrict__ g_m,
                                 const uint64_t* __restrict__ g_U, double* __restrict__ lse_out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const double m = g_m[g];
  if (m != m) return;  // overflowed group: the re-run writes its items
  const double lse = pclean_lse_from_fix(m, g_U[g]);
  if (!grp_off) {
    lse_out[out_pos ? out_pos[g] : g] = lse;
    return;
  }
  const int hi = grp_off[g + 1];
  for (int mi = grp_off[g]; mi < hi; ++mi) lse_out[out_pos ? out_pos[members[mi]] : members[mi]] = lse;
}

// the same by MEMBER POSITION (uid[mi] - 1 = the group of position mi): a launch whose groups are not cut into pieces may hold
// groups of thousands of rows, and a thread walking one of them alone was the launch group's tail
__global__ void member_lse_kernel(int n_pos, const int32_t* __restrict__ uid, const int32_t* __restrict__ members,
                                  const int32_t* __restrict__ out_pos, const double* __restrict__ g_m,
                                  const uint64_t* __restrict__ g_U, double* __restrict__ lse_out) {
  const int mi = blockIdx.x * blockDim.x + threadIdx.x;
  if (mi >= n_pos) return;
  const int g = uid[mi] - 1;
  const double m = g_m[g];
  if (m != m) return;  // overflowed group: the re-run writes its items
  const int tm = members[mi];
  lse_out[out_pos ? out_pos[tm] : tm] = pclean_lse_from_fix(m, g_U[g]);
}

// ---- overflow_lds_kernel: the items whose pre-filter survivor list overflowed (flat posteriors), re-run over ALL
// candidates.  Same contract and results as enum_node_kernel (enum_kernels.hip) — scores in LDS, fixed-point
// weights in place, chunk sums + block scan, binary-search draws — but phase 1 scores through the candidate-compact
// byte rows (fast_exact_score: coalesced byte loads, one density-table lookup per term) instead of the generic
// kernel's dependent gather chains (candidate column -> pair byte -> length -> density pieces), which made 27
// overflowed rows cost 0.56 ms of every 1M-row sweep.  One workgroup per item; items are not grouped.
#define OVF_CPT 6   // candidates a thread of overflow_lds_kernel scores at a time (their loads in flight together)
#define OVF_T 1024  // threads per workgroup of overflow_lds_kernel (one workgroup per CU: the scores fill its LDS)
__global__ __launch_bounds__(OVF_T) void overflow_lds_kernel(const FastRootDev fr, const ItemsDev it, const ChildrenDev ch,
                                                           uint64_t seed, uint32_t sweep, uint32_t site, int n_draws,
                                                           double* __restrict__ lse_out,
                                                           int32_t* __restrict__ draws_out,
                                                           const int32_t* __restrict__ over_list,
                                                           const unsigned int* __restrict__ over_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_o[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // list mode: the items the scan kernel appended to over_list (count on the device), fixed grid; else item = workgroup
  const int n_work = over_list ? (int)*over_count : it.n;
  for (int wi_ = blockIdx.x; wi_ < n_work; wi_ += gridDim.x) {
  __syncthreads();  // the previous item's LDS is no longer read
  const int t = over_list ? over_list[wi_] : wi_;
  const int to = it.out_pos ? it.out_pos[t] : t;
  const int n = fr.n_cand;
  const bool fk = !fr.is_leaf;
  const int nc = n + (fk ? 1 : 0);
  double* s = (double*)smem_o;
  uint64_t* u = (uint64_t*)smem_o;
  double* red = (double*)(smem_o + (size_t)((nc + 1) & ~1) * 8);
  uint64_t* wsum = (uint64_t*)(red + 16);
  const int row = it.row ? it.row[t] : t;
  const int excl = it.excl ? it.excl[t] : -1;
  const int ctx0 = it.ctx ? it.ctx[(size_t)t * PCLEAN_MAX_CTX] : 0;
  const int ctx1 = it.ctx ? it.ctx[(size_t)t * PCLEAN_MAX_CTX + 1] : 0;
  int o[PCLEAN_MAX_TERMS];
  load_row_obs(fr, row, o);
  const bool deleted = excl >= 0 && fr.counts && fr.counts[excl] <= 1;
  const double* prior = (excl >= 0 && fr.prior_e) ? fr.prior_e : fr.prior_n;
  // ---- phase 1: exact scores, prior first, terms in plan order (candidate_score's operation order)
  // OVF_CPT candidates per thread at a time, term by term: the byte / length loads of the whole batch are in flight together,
  // then its density loads — one candidate at a time was a chain of ~3 dependent round trips per term and candidate (20
  // overflowed rows of block 0: 0.10 ms of every 1M-row sweep on 20 CUs).  Each candidate's additions keep plan order.
  double lmax = -__builtin_inf();
  const double pr_excl = excl >= 0 ? (deleted ? -__builtin_inf() : fr.logc_m1[excl] - fr.scal[1]) : 0.0;
  for (int k0 = tid; k0 < n; k0 += OVF_T * OVF_CPT) {
    // (every load below is unconditional — a candidate beyond n re-reads k0, a free slot's bytes are defined — and what must
    // not count is selected away afterwards: a load under a condition becomes a branch with its own wait, one round trip each)
    int kk[OVF_CPT];
    double b[OVF_CPT];
    bool in[OVF_CPT], on[OVF_CPT];
#pragma unroll
    for (int c = 0; c < OVF_CPT; ++c) {
      const int k = k0 + c * OVF_T;
      in[c] = k < n;
      kk[c] = in[c] ? k : k0;
    }
#pragma unroll
    for (int c = 0; c < OVF_CPT; ++c) b[c] = prior[kk[c]];
#pragma unroll
    for (int c = 0; c < OVF_CPT; ++c) {
      b[c] = kk[c] == excl ? pr_excl : b[c];
      on[c] = in[c] && b[c] != -__builtin_inf();  // (free slots: no defined values to look up)
    }
    for (int f = 0; f < fr.n_terms; ++f) {
      const int of = o[f];
      if (of < 0) continue;  // an explicitly missing observation contributes nothing (add_typos.jl:51-53)
      const FastTermDev& tm = fr.terms[f];
      if (tm.ctx_slot >= 0) {
#pragma unroll
        for (int c = 0; c < OVF_CPT; ++c)
          if (on[c]) b[c] += wave_term_dens(fr, tm, of, ctx0, ctx1, kk[c]);
        continue;
      }
      const uint8_t* crow = tm.comp + (size_t)of * fr.kpad;
      const uint8_t* clen = tm.clen;
      const int mt = tm.max_typos;
      int d[OVF_CPT], L[OVF_CPT];
#pragma unroll
      for (int c = 0; c < OVF_CPT; ++c) {
        d[c] = (int)crow[kk[c]];
        L[c] = (int)clen[kk[c]];
      }
      bool sat = false;
#pragma unroll
      for (int c = 0; c < OVF_CPT; ++c) sat |= on[c] && d[c] == PRE_CLAMP;
      if (sat) {  // saturated bytes (rare): the true distances
#pragma unroll
        for (int c = 0; c < OVF_CPT; ++c)
          if (on[c] && d[c] == PRE_CLAMP) d[c] = tm.pair[(size_t)of * tm.n_lat + tm.cand_col[kk[c]]];
      }
      double l[OVF_CPT];
#pragma unroll
      for (int c = 0; c < OVF_CPT; ++c) l[c] = fr.atd[(size_t)L[c] * fr.atd_stride + ((on[c] && !(mt >= 0 && d[c] > mt)) ? d[c] : 0)];
#pragma unroll
      for (int c = 0; c < OVF_CPT; ++c) {
        const double lc = (mt >= 0 && d[c] > mt) ? ADD_TYPOS_IMPOSSIBLE : l[c];
        b[c] += on[c] ? lc : 0.0;  // (b is -inf or never stored where on is false: adding 0.0 changes nothing)
      }
    }
#pragma unroll
    for (int c = 0; c < OVF_CPT; ++c)
      if (in[c]) {
        s[kk[c]] = b[c];
        lmax = fmax(lmax, b[c]);
      }
  }
  if (fk && tid == 0) {  // new_score() of enum_kernels.hip
    const double logden = excl >= 0 ? fr.scal[1] : fr.scal[0];
    double snew = 0.0;
    for (int c = 0; c < ch.n; ++c) {
      size_t idx = (size_t)to;
      if (ch.obs_col[c]) {
        const int oc = ch.obs_col[c][row];
        idx = oc < 0 ? (size_t)ch.n_obs[c] : (size_t)oc;
      }
      snew += ch.arr[c][idx];
    }
    const double sn = ((deleted ? fr.scal[3] : fr.scal[2]) - logden) + snew;
    s[n] = sn;
    lmax = fmax(lmax, sn);
  }
  lmax = wave_max64(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  double m = red[0];
  for (int w = 1; w < OVF_T / 64; ++w) m = fmax(m, red[w]);
  for (int k = tid; k < nc; k += OVF_T) {
    const double sk = s[k];
    u[k] = (m == -__builtin_inf()) ? 0ull : pclean_fixw(sk - m);
  }
  __syncthreads();
  const int chunk = (nc + OVF_T - 1) / OVF_T;
  const int lo = min(tid * chunk, nc), hi = min(lo + chunk, nc);
  uint64_t part = 0;
  for (int k = lo; k < hi; ++k) part += u[k];
  unsigned long long incl = part;
  for (int sh = 1; sh < 64; sh <<= 1) {
    const unsigned long long x = __shfl_up(incl, sh, 64);
    if (lane >= sh) incl += x;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint64_t base = 0, U = 0;
  for (int w = 0; w < OVF_T / 64; ++w) {
    if (w < wave) base += wsum[w];
    U += wsum[w];
  }
  {
    uint64_t run = base + incl - part;
    for (int k = lo; k < hi; ++k) {
      run += u[k];
      u[k] = run;
    }
  }
  __syncthreads();
  if (tid == 0 && lse_out) lse_out[to] = pclean_lse_from_fix(m, U);
  const int draw_is = it.draw_is ? it.draw_is : n_draws, draw_ds = it.draw_ds ? it.draw_ds : 1;
  for (int j = tid; j < n_draws; j += OVF_T) {
    int32_t res = fk ? PCLEAN_CHOICE_NEW : n - 1;
    if (U != 0) {
      const uint32_t rng_row = it.rng_row ? (uint32_t)it.rng_row[t] : (uint32_t)((int64_t)row + it.row_offset);
      const uint32_t pid = it.particle ? (uint32_t)it.particle[t] : (uint32_t)j;
      const uint64_t x = pclean_mulhi64(pclean_rand64(seed, rng_row, site, pid, sweep), U);
      int a = 0, b = nc - 1;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (u[mid] > x)
          b = mid;
        else
          a = mid + 1;
      }
      res = (fk && a == n) ? PCLEAN_CHOICE_NEW : a;
    }
    draws_out[(size_t)to * draw_is + (size_t)j * draw_ds] = res;
  }
  }
}

// returns 1 when the launch was made, 0 when the candidates do not fit the LDS of one workgroup (caller: generic kernel)
static size_t overflow_lds_bytes(const FastRootDev& fr) {
  const int nc = fr.n_cand + (fr.is_leaf ? 0 : 1);
  return (size_t)((nc + 1) & ~1) * 8 + (32 + 64) * 8;
}
// can the overflowed items of this launch be re-run by overflow_lds_kernel (candidates fit one workgroup's LDS)?
int pclean_overflow_fast_ok(const FastRootDev& fr, const ItemsDev& it) {
  return overflow_lds_bytes(fr) <= 160 * 1024 && !it.ev_lo;
}
// over_list == null: one workgroup per item of `it` (ungrouped).  Otherwise the items over_list[0 .. *over_count) of
// `it` (ungrouped view of the scan's items), fixed grid, nothing read back.
int pclean_launch_overflow_fast(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const ChildrenDev& ch,
                                uint64_t seed, uint32_t sweep, uint32_t site, int n_draws, double* lse_out,
                                int32_t* draws_out, const int32_t* over_list, const unsigned int* over_count) {
  if (it.n <= 0) return 1;
  const size_t lds = overflow_lds_bytes(fr);
  if (lds > 160 * 1024 || it.grp_off || it.ev_lo) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(ctx, hipFuncSetAttribute((const void*)overflow_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int grid = over_list ? std::min(it.n, 256) : it.n;
  hipLaunchKernelGGL(overflow_lds_kernel, dim3(grid), dim3(OVF_T), lds, ctx->stream, fr, it, ch, seed, sweep, site, n_draws,
                     lse_out, draws_out, over_list, over_count);
  HIPCHK(ctx, hipGetLastError());
  return 1;
}

// debug (pclean_debug_root_flags): per item of the last launch, bit 0 = re-run by the generic kernel (survivor list
// overflowed), bit 1 = its group was scanned in guess-and-refine mode
__global__ void root_flags_kernel(int n_groups, const int32_t* __restrict__ gd, const int32_t* __restrict__ grp_off,
                                  const int32_t* __restrict__ members, const int32_t* __restrict__ oflag,
                                  int32_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const int refine = (gd[(size_t)g * GD_STRIDE + 7] & 2) ? 2 : 0;
  const int lo = grp_off ? grp_off[g] : g, hi = grp_off ? grp_off[g + 1] : g + 1;
  for (int mi = lo; mi < hi; ++mi) {
    const int t = grp_off ? members[mi] : g;
    out[t] = refine | (oflag[t] != 0 ? 1 : 0);
  }
}
int pclean_launch_root_flags(pclean_ctx* ctx, int n_groups, const int32_t* gd, const int32_t* grp_off,
                             const int32_t* members, const int32_t* oflag, int32_t* out) {
  hipLaunchKernelGGL(root_flags_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, ctx->stream, n_groups, gd, grp_off,
                     members, oflag, out);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

typedef void (*wave_kernel_t)(const FastRootDev, const WaveItems, uint64_t, uint32_t, uint32_t, int, int, int, const int32_t*,
                              unsigned int*, double*, uint64_t*, int32_t*, int32_t*, unsigned int*, int32_t*, unsigned int*,
                              const int32_t*, int32_t*, uint64_t*, int32_t*, const int32_t*, int);

static wave_kernel_t pick_kernel(int n_terms) {
  if (n_terms <= 2) return fk_root_wave_kernel<2, WAVE_SURV_CAP, 4>;
  if (n_terms <= 4) return fk_root_wave_kernel<4, WAVE_SURV_CAP, 4>;
  if (n_terms <= 8) return fk_root_wave_kernel<8, WAVE_SURV_CAP, 4>;
  if (n_terms <= 12) return fk_root_wave_kernel<12, WAVE_SURV_CAP, 4>;
  return fk_root_wave_kernel<16, WAVE_SURV_CAP, 4>;
}

// int32 words of desc_scratch for n_groups groups: descriptors, 8 chunk counters, per-group (maximum, total)
size_t pclean_fast_desc_words(int n_groups) {
  const size_t ng = (size_t)std::max(n_groups, 1);
  // (+ the work list of the groups the settle kernel leaves: dense, and in WL_SEGS segments while it is being written)
  return ng * GD_STRIDE + 8 * WAVE_CTR_STRIDE + ng * 4 + ng + ng + (ng + 16 * (WL_SEGS + 1) + 64);
}

int pclean_launch_root_fast(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const ChildrenDev& ch,
                            uint64_t seed, uint32_t sweep, uint32_t site, int n_draws, double* lse_out,
                            int32_t* draws_out, int32_t* overflow_flag, unsigned int* overflow_count,
                            int32_t* desc_scratch, int32_t* overflow_list, unsigned int* scan_stats, int n_items,
                            const double* pre_score, bool want_worklist, unsigned int* wl_stat, const int32_t* pre_obs,
                            RootExtra* extra) {
  if (it.n <= 0) return PCLEAN_OK;
  const size_t ng = (size_t)it.n;
  unsigned int* chunk_ctr = reinterpret_cast<unsigned int*>(desc_scratch + ng * GD_STRIDE);
  double* g_m = reinterpret_cast<double*>(desc_scratch + ng * GD_STRIDE + 8 * WAVE_CTR_STRIDE);
  uint64_t* g_U = reinterpret_cast<uint64_t*>(g_m + ng);
  int32_t* g_res = reinterpret_cast<int32_t*>(g_U + ng);
  int32_t* worklist = g_res + ng;
  // groups are settled by group_desc_kernel only when their rows' draws can be written afterwards (the member count is
  // known) and the launch draws at all
  // OFF by default (PCLEAN_RESOLVE_GROUPS=1 turns it on): bit-identical on every parity test, and it takes the scan
  // kernel's full scans from 138k to 5.6k on the 1M-row table — but as written (one thread walking its group's block
  // minima with uncoalesced 16-byte loads and early exits) group_desc_kernel becomes the slower part: 1.25 -> 1.87 ms
  // for the launch pair.  Kept for the next round: the settling belongs in a kernel with a few lanes per group.
  static const bool want_resolve = getenv("PCLEAN_RESOLVE_GROUPS") != nullptr;
  const bool resolve = n_draws > 0 && want_resolve && (!it.grp_off || n_items > 0);
  const bool lazy = extra && extra->lz_ns && n_draws > 0 && !resolve;
  hipLaunchKernelGGL(group_desc_kernel, dim3((it.n + 255) / 256), dim3(256), 0, ctx->stream, fr, it, ch, it.n,
                     desc_scratch, chunk_ctr, g_m, g_U, resolve ? g_res : nullptr, scan_stats, pre_score, pre_obs);
  if (extra) extra->g_U = g_U;
  // the easy groups settled by a few lanes each (group_settle_kernel): launches that draw, large enough to matter
  static const bool no_settle = getenv("PCLEAN_NO_SETTLE") != nullptr;
  const bool settle = !no_settle && !resolve && n_draws > 0 && !fr.is_leaf && fr.n_pre >= 1 && fr.n_pre <= 3 && fr.cstride > 0 &&
                      it.excl != nullptr && it.n >= 4096;
  // persistent grid = what is resident at once (a workgroup that starts late would find the counters drained anyway)
  const int wpg = 4;
  wave_kernel_t kern = pick_kernel(fr.n_terms);
  static const bool no_dense = getenv("PCLEAN_NO_DENSE_SCAN") != nullptr;  // (A/B: the survivor sets are the same either way)
  static int resident[17] = {0};  // per kernel variant (indexed by its term capacity), queried once
  const int variant = fr.n_terms <= 2 ? 2 : fr.n_terms <= 4 ? 4 : fr.n_terms <= 8 ? 8 : fr.n_terms <= 12 ? 12 : 16;
  int& res = resident[variant];
  if (!res) {
    int per_cu = 0, n_cu = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 64 * wpg, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    per_cu = std::min(per_cu, 8);
    res = n_cu * per_cu;
  }
  int wgs = res;
  if (const char* e = getenv("PCLEAN_WAVE_WGS")) wgs = std::max(1, atoi(e));
  // groups per hand-out: every resident wave gets work before any wave takes several groups at a time
  int chunk = std::max(1, std::min(WAVE_CHUNK, it.n / std::max(res * wpg, 1)));
  if (const char* e = getenv("PCLEAN_WAVE_CHUNK")) chunk = std::max(1, atoi(e));
  wgs = std::min(wgs, (it.n + wpg * chunk - 1) / (wpg * chunk));
  wgs = (wgs + 7) & ~7;  // a multiple of the 8 XCDs
  WaveItems wi{it.grp_off ? it.members : nullptr, it.row, it.rng_row, it.particle, it.out_pos, it.row_offset, it.draw_is,
               it.draw_ds};
  bool use_worklist = false;
  if (settle) {
    WaveItems ws = wi;
    ws.draw_is = it.draw_is ? it.draw_is : n_draws;
    ws.draw_ds = it.draw_ds ? it.draw_ds : 1;
    SettleArgs sa{};
    for (int p = 0; p < 3; ++p) {
      const bool on = p < fr.n_pre && fr.terms[fr.pre[p]].comp && fr.terms[fr.pre[p]].cmin;
      sa.comp[p] = on ? fr.terms[fr.pre[p]].comp : fr.zero_row;
      sa.cmin[p] = on ? fr.terms[fr.pre[p]].cmin : fr.zero_row;
      sa.pre[p] = on ? 10 + fr.pre[p] : -1;
    }
    sa.kpad = fr.kpad;
    sa.kscan = fr.kscan;
    sa.cstride = fr.cstride;
    sa.alive = fr.alive;
    sa.zero_row = fr.zero_row;
    const size_t n_thr = (size_t)it.n * SETTLE_L;
    static const bool no_worklist = getenv("PCLEAN_NO_WORKLIST") != nullptr;
    use_worklist = !no_worklist && want_worklist;
    const unsigned int n_wg = (unsigned int)((n_thr + 255) / 256);
    const int seg_cap = (int)((n_wg / WL_SEGS + 1) * 16);  // 16 groups per workgroup, workgroups dealt round-robin
    int32_t* seg_list = worklist + ng;
    hipLaunchKernelGGL(group_settle_kernel, dim3(n_wg), dim3(256), 0, ctx->stream, sa, ws, it.n, n_draws, desc_scratch, g_m, g_U,
                       draws_out, scan_stats, use_worklist ? seg_list : nullptr, chunk_ctr, seg_cap, lazy ? extra->lz_ns : nullptr,
                       lazy ? extra->lz_k : nullptr, lazy ? extra->lz_p : nullptr, lazy ? extra->eager_rows : nullptr);
    if (use_worklist)
      hipLaunchKernelGGL(worklist_pack_kernel, dim3(1), dim3(1024), 0, ctx->stream, seg_list, seg_cap, chunk_ctr, worklist, wl_stat);
  }
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * wpg), 0, ctx->stream, fr, wi, seed, sweep, site, n_draws, it.n, chunk, desc_scratch,
                     chunk_ctr, g_m, g_U, draws_out, overflow_flag, overflow_count, overflow_list, scan_stats,
                     use_worklist ? worklist : nullptr, lazy ? extra->lz_k : nullptr, lazy ? extra->lz_p : nullptr,
                     lazy ? extra->lz_ns : nullptr, lazy ? extra->eager_rows : nullptr, no_dense ? 0 : 1);
#ifdef WAVE_PHASE_CLOCK
  if (it.n > 100000) {
    unsigned long long h[16];
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wave_clk), sizeof h);
    const char* nm[7] = {"handout+desc", "decode", "scan", "exact", "weights", "draws+out", "tail"};
    double tot = 0;
    for (int i = 0; i < 6; ++i) tot += (double)h[i];
    fprintf(stderr, "[wave clk] groups %d terms %d waves %llu (grid %d WGs): ", it.n, fr.n_terms, h[8], wgs);
    for (int i = 0; i < 6; ++i) fprintf(stderr, "%s %.1f%% ", nm[i], 100.0 * (double)h[i] / tot);
    fprintf(stderr, "| cycles/group %.0f, groups seen %llu; full scans %llu, cached scans %llu, decided groups %llu, survivors scored %llu; two-level scans %llu, passing blocks %llu; chunk %d, lazy %d\n",
            tot / (double)h[7], h[7], h[9], h[10], h[12], h[11], h[13], h[14], chunk, lazy ? 1 : 0);
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_clk), h, sizeof h);
  }
#endif
  if (resolve) {
    const int n_mem_all = it.grp_off ? n_items : it.n;
    hipLaunchKernelGGL(resolved_draws_kernel, dim3((n_mem_all + 255) / 256), dim3(256), 0, ctx->stream, n_mem_all, it.n,
                       it.grp_off, it.members, it.out_pos, g_res, n_draws, it.draw_is ? it.draw_is : n_draws,
                       it.draw_ds ? it.draw_ds : 1, draws_out);
  }
  if (lse_out && it.grp_off && it.grp_uid && n_items > 0)
    hipLaunchKernelGGL(member_lse_kernel, dim3((n_items + 255) / 256), dim3(256), 0, ctx->stream, n_items, it.grp_uid, it.members,
                       it.out_pos, g_m, g_U, lse_out);
  else if (lse_out)
    hipLaunchKernelGGL(group_lse_kernel, dim3((it.n + 255) / 256), dim3(256), 0, ctx->stream, it.n, it.grp_off, it.members,
                       it.out_pos, g_m, g_U, lse_out);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// ---- one draw per row after the final choice (RootExtra: lazy draws) --------------------------------------------------------
// Thread per member position of the root launch's grouping (the positions of a group are adjacent: its threads probe the same
// few cache lines of the group's prefix).  The draw is the one the scan kernel would have made for (item, chosen particle):
// same Philox counter (row, site, particle, sweep), same multiply-high against the group's total, same search.
__global__ __launch_bounds__(256) void lazy_draw_kernel(const LazyDrawArgs a, uint64_t seed, uint32_t sweep, uint32_t site) {
  const int mi = blockIdx.x * blockDim.x + threadIdx.x;
  if (mi >= a.n_pos) return;
  const int tm = a.members ? a.members[mi] : mi;
  const int row = a.item_row ? a.item_row[tm] : tm;
  if (a.eager_rows && a.eager_rows[row] != 0) return;  // (those rows' particles were updated one by one, with every draw)
  const int c = a.chosen[row];
  if (a.slot_item && a.slot_item[(size_t)c * a.n_rows + row] != tm) return;  // another item carries the chosen particle's context
  if (c == 0 && a.cur_b && a.cur_b[row] >= 0) return;  // the retained particle keeps its referent (row_inference.jl:143-145)
  const int g = a.uid ? a.uid[mi] - 1 : mi;
  const int ns = a.lz_ns[g];
  int32_t res;
  if (ns < 0) {
    res = a.draws_item[(size_t)tm * a.n_particles + c];
  } else {
    res = a.res_new;
    const uint64_t U = a.g_U[g];
    if (U != 0) {
      const uint64_t x = pclean_mulhi64(pclean_rand64(seed, (uint32_t)((int64_t)row + a.row_offset), site, (uint32_t)c, sweep), U);
      const uint64_t* pref = a.lz_p + (size_t)g * ROOT_LZ_CAP;
      int lo = 0, hi = ns;  // smallest index with prefix > x (index ns = the new row)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pref[mid] > x)
          hi = mid;
        else
          lo = mid + 1;
      }
      if (lo < ns) res = a.lz_k[(size_t)g * ROOT_LZ_CAP + lo];
    }
  }
  a.pchoice[(size_t)c * a.n_rows + row] = res;
}
int pclean_launch_lazy_draws(pclean_ctx* ctx, const LazyDrawArgs& a, uint64_t seed, uint32_t sweep, uint32_t site) {
  if (a.n_pos <= 0) return PCLEAN_OK;
  hipLaunchKernelGGL(lazy_draw_kernel, dim3((a.n_pos + 255) / 256), dim3(256), 0, ctx->stream, a, seed, sweep, site);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
