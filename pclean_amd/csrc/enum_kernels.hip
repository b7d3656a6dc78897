// Enumeration kernels: rows x candidate-referents proposal scoring, per-item
// fixed-point log-sum-exp and inverse-CDF categorical draws.
//
// Take over the enumeration loops the reference JIT-generates per block
// (src/inference/proposal_compiler.jl:131-252 ForeignKeyNode, 55-129 discrete
// RandomChoiceNode) plus the CRP prior (165-171) and the AddTypos densities
// they call (src/distributions/add_typos.jl:50-66, via the pair tables).
//
// enum_node_kernel — one workgroup (256 lanes = 4 wavefronts) per work item, or per GROUP of items
// with identical score vectors (ItemsDev::grp_off / members: scores once, draws per member item),
// candidate scores resident in LDS (<= ~20k candidates):
//   phase 1  lane-strided over candidates: coalesced loads of the flattened
//            latent table columns, gather of the pair-table byte, fp64 density
//            epilogue, score -> LDS
//   phase 2  wave-shuffle + LDS max reduction
//   phase 3  u_k = floor(exp(s_k - m) 2^40) in place (uint64; integer sums are
//            order independent => bit-identical to the sequential oracle)
//   phase 4  per-lane contiguous chunk sums, shuffle scan across the block
//   phase 5  lse = m + log(U 2^-40); the chunk sums become an inclusive prefix in place and every
//            (member item, draw) pair locates its Philox threshold by binary search, one pair per lane
// enum_node_big_kernel — same contract for candidate sets that do not fit in
// LDS (large option lists of StringPrior / ChooseUniformly leaves): scores are
// recomputed per pass instead of stored (max pass, weight pass, locate pass).
#include <vector>
#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"
#include <algorithm>

#include "enum.h"
#include "gauss_dev.h"

#define HALF_LOG26 1.629048269010741
#define ADD_TYPOS_IMPOSSIBLE (-1e5)
#define ENUM_CPT 5  // candidates a thread of enum_node_kernel scores at a time (candidate_score_batch)
#define CS_TC 4  // evidence entries of a term whose gathers are in flight together (candidate_score_ev)

__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ double term_density(const TermDev& tm, const DensDev& dn, int d, int val) {
  if (tm.dens_kind == PCLEAN_DENS_EQUAL) return d == 0 ? 0.0 : -__builtin_inf();
  if (tm.max_typos >= 0 && d > tm.max_typos) return ADD_TYPOS_IMPOSSIBLE;
  const int L = tm.lat_len[val];
  const int r = (L + 4) / 5;
  double l = dn.nb[(size_t)r * dn.nb_stride + d];
  l -= dn.logl[L] * (double)d;
  l -= HALF_LOG26 * (double)d;
  return l;
}

// MaybeSwap (maybe_swap.jl:13-28): o = observed value index (-1 missing), same = strings equal
__device__ __forceinline__ double maybe_swap_density(const TermDev& tm, const DensDev& dn, int o, int d, int val, int k,
                                                     int pidx) {
  if (o < 0) return val >= tm.other_val ? -1000.0 : 0.0;  // dummy and sampled strings (ids after it): not an option
  if (d == 0) return dn.prob_same[pidx];
  return dn.prob_diff[pidx] - dn.logn[tm.aux_col[k]];
}

struct ItemView {
  int row, excl, item;
  const int32_t* ctxv;
  bool deleted;
  double logden;
  int ev_lo, ev_hi;  // evidence range (ev_lo < 0: the single row `row`)
};

__device__ __forceinline__ ItemView item_view(const NodeDev& nd, const ItemsDev& it, int t) {
  ItemView v;
  v.item = t;
  v.row = it.row ? it.row[t] : t;
  v.excl = it.excl ? it.excl[t] : -1;
  v.ctxv = it.ctx ? it.ctx + (size_t)t * PCLEAN_MAX_CTX : nullptr;
  v.deleted = false;
  v.logden = 0.0;
  v.ev_lo = -1;
  v.ev_hi = -1;
  if (it.ev_lo) {
    v.ev_lo = it.ev_lo[t];
    v.ev_hi = it.ev_hi[t];
  }
  if (nd.kind == PCLEAN_NODE_FK) {
    const bool excluded = v.excl >= 0;
    v.deleted = excluded && nd.counts[v.excl] <= 1;
    v.logden = excluded ? nd.scal[1] : nd.scal[0];
  }
  return v;
}

// score of existing candidate k (the single definition both kernels use, so the
// fp64 operation order — prior, then terms in plan order — is identical)
// Gaussian term of candidate k for observed row `row` (evctx = ctx of the evidence row, may be null)
__device__ __forceinline__ double gauss_term(const NodeDev& nd, const ItemView& v, int k, int row, const int32_t* evctx) {
  const GaussDev& g = nd.g;
  const double xv = g.x[row];
  if (xv != xv) return 0.0;  // missing numeric observation
  double sc[16];
  int codes[16];
  const int n = gauss_combo_scores(
      g, row, evctx,
      [&](int d) -> int {
        switch (g.src_kind[d]) {
          case PCLEAN_GSRC_CAND: return g.src_ptr[d][k];
          case PCLEAN_GSRC_OBS: return g.src_ptr[d][row];
          case PCLEAN_GSRC_ITEMCTX: return v.ctxv[g.src_slot[d]];
          default: return evctx[g.src_slot[d]];
        }
      },
      sc, codes);
  return gauss_lse(sc, n);
}

// Evidence sets (latent-class rows scored against every observed row referring to them — the
// ExternalLikelihoodNodes of proposal_compiler.jl:306-350 / block_proposal.jl:119-155).  Order of the fp64
// additions (the oracle restates it, oracle/enumerate.h): terms in plan order; per term the DISTINCT (ctx value,
// observed value) pairs of the evidence rows in ascending order, each adding multiplicity x density — the ~100
// referring rows of a hospital hold a handful of distinct values per column; then the Gaussian terms row by row.
__device__ __forceinline__ double candidate_score_ev(const NodeDev& nd, const DensDev& dn, const ItemsDev& it,
                                                     const ItemView& v, int k, double sk) {
  const int oi = it.ev_item ? it.ev_item[v.item] : v.item;
  for (int ti = 0; ti < nd.n_terms; ++ti) {
    const TermDev& tm = nd.terms[ti];
    const AggDev ag = it.agg[ti];
    const int r1 = ag.end ? ag.end[oi] : ag.off[oi + 1];
    if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP) {
      for (int r = ag.off[oi]; r < r1; ++r) {
        const uint64_t key = ag.key[r];
        const int o = (int)(key & 0xffffffull) - 1;
        const int ec = (int)((key >> 24) & 0xffffull);
        const double mult = (double)ag.cnt[r];
        const int val = tm.cand_col[k];
        const int c = tm.ctx_mode == 0 ? v.ctxv[tm.ctx_slot] : ec;
        const int d = o < 0 ? 1 : (int)tm.pair[(size_t)o * tm.n_lat + val];
        sk += mult * maybe_swap_density(tm, dn, o, d, val, k, c);
      }
      continue;
    }
    // the entries CS_TC at a time: an entry is a chain of dependent gathers (key -> pair byte and word length -> density
    // pieces), and walked one after the other the chains added up — these kernels wait for memory, nothing else.  Keys and
    // multiplicities, then the pair bytes and word lengths, then the density pieces: each step's loads are issued for the
    // whole chunk before any is used; the additions keep the entry order (same operations on the same values: same bits)
    const int val0 = tm.cand_col[k];
    const int cv = (tm.ctx_slot >= 0 && tm.ctx_mode == 0) ? v.ctxv[tm.ctx_slot] : 0;
    const bool typos = tm.dens_kind != PCLEAN_DENS_EQUAL;
    for (int rb = ag.off[oi]; rb < r1; rb += CS_TC) {
      // (every load of a step unconditional, on a safe index — an entry past the item's last repeats the last one, a missing
      // observation reads row 0 — and what must not count selected away: a load under a per-entry condition is a branch with
      // its own wait, which put the chunk's chains one behind the other again)
      int o[CS_TC], val[CS_TC], d[CS_TC], L[CS_TC];
      double mult[CS_TC], a[CS_TC], b[CS_TC];
      int ec[CS_TC], cn[CS_TC];
#pragma unroll
      for (int u = 0; u < CS_TC; ++u) {
        const int r = min(rb + u, r1 - 1);
        const uint64_t key = ag.key[r];
        cn[u] = ag.cnt[r];
        o[u] = (int)(key & 0xffffffull) - 1;
        ec[u] = (int)((key >> 24) & 0xffffull);
      }
#pragma unroll
      for (int u = 0; u < CS_TC; ++u) {
        const bool in = rb + u < r1;
        o[u] = in ? o[u] : -1;
        mult[u] = in ? (double)cn[u] : 0.0;
        val[u] = val0;
      }
      if (tm.ctx_slot >= 0) {  // (wave-uniform)
#pragma unroll
        for (int u = 0; u < CS_TC; ++u) {
          const int c = o[u] < 0 ? 0 : (tm.ctx_mode == 0 ? cv : ec[u]);
          const int fv = tm.ctx_mode == 2 ? tm.fn[(size_t)val0 * tm.fn_nb + c] : tm.fn[(size_t)c * tm.fn_nb + val0];
          val[u] = o[u] >= 0 ? fv : val0;
        }
      }
      if (typos) {
#pragma unroll
        for (int u = 0; u < CS_TC; ++u) L[u] = tm.lat_len[val[u]];
      } else {
#pragma unroll
        for (int u = 0; u < CS_TC; ++u) L[u] = 0;
      }
      if (tm.elem_bytes == 1) {
#pragma unroll
        for (int u = 0; u < CS_TC; ++u) d[u] = (int)tm.pair[o[u] >= 0 ? (size_t)o[u] * tm.n_lat + val[u] : (size_t)0];  // (entry 0: there even when the column holds no observation at all)
      } else {
#pragma unroll
        for (int u = 0; u < CS_TC; ++u) d[u] = (int)((const uint16_t*)tm.pair)[o[u] >= 0 ? (size_t)o[u] * tm.n_lat + val[u] : (size_t)0];
      }
      if (typos) {
#pragma unroll
        for (int u = 0; u < CS_TC; ++u) {
          const bool look = o[u] >= 0 && !(tm.max_typos >= 0 && d[u] > tm.max_typos);
          a[u] = dn.nb[(size_t)((L[u] + 4) / 5) * dn.nb_stride + (look ? d[u] : 0)];
          b[u] = dn.logl[L[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < CS_TC; ++u) {
        double l;  // (term_density's operations, on the values loaded above)
        if (!typos) {
          l = d[u] == 0 ? 0.0 : -__builtin_inf();
        } else if (tm.max_typos >= 0 && d[u] > tm.max_typos) {
          l = ADD_TYPOS_IMPOSSIBLE;
        } else {
          l = a[u];
          l -= b[u] * (double)d[u];
          l -= HALF_LOG26 * (double)d[u];
        }
        if (o[u] >= 0) sk += mult[u] * l;  // (not: a missing observation, or past the item's last entry)
      }
    }
  }
  if (nd.g.on)
    for (int e = v.ev_lo; e < v.ev_hi; ++e)
      if (sk > -__builtin_inf())
        sk += gauss_term(nd, v, k, it.ev_rows[e], it.ev_ctx ? it.ev_ctx + (size_t)e * PCLEAN_MAX_CTX : nullptr);
  return sk;
}

// (EV = false: a launch whose items are single rows — the evidence path is compiled out of the kernel, which keeps the
// registers of enum_node_kernel's single-row instances where they were before candidate_score_ev held a chunk's loads in flight)
template <bool EV = true>
__device__ __forceinline__ double candidate_score(const NodeDev& nd, const DensDev& dn, const ItemsDev& it,
                                                  const ItemView& v, int k) {
  double sk;
  if (nd.kind == PCLEAN_NODE_FK) {
    if (nd.counts[k] == 0) return -__builtin_inf();  // free slot
    if (k == v.excl)
      sk = v.deleted ? -__builtin_inf() : nd.logc_m1[k] - v.logden;
    else
      sk = nd.logc_full[k] - v.logden;
  } else {
    sk = nd.logc_full[k];
  }
  if (EV && v.ev_lo >= 0) return candidate_score_ev(nd, dn, it, v, k, sk);
  for (int ti = 0; ti < nd.n_terms; ++ti) {
    const TermDev& tm = nd.terms[ti];
    const int o = tm.obs_col[v.row];
    if (o < 0) continue;  // explicitly missing observation (add_typos.jl:51-53)
    int val = tm.cand_col[k];
    if (tm.ctx_slot >= 0) val = tm.fn[(size_t)v.ctxv[tm.ctx_slot] * tm.fn_nb + val];
    const size_t idx = (size_t)o * tm.n_lat + val;
    const int d = tm.elem_bytes == 1 ? (int)tm.pair[idx] : (int)((const uint16_t*)tm.pair)[idx];
    sk += term_density(tm, dn, d, val);
  }
  if (nd.g.on && sk > -__builtin_inf()) sk += gauss_term(nd, v, k, v.row, nullptr);
  return sk;
}

// candidate_score() of CPT candidates of one item at once (single-row items without a Gaussian term): the same operations per
// candidate in the same order, but every level of the gather chain (candidate column -> ctx function -> pair entry + length ->
// density pieces) is loaded for the whole batch before the next level is touched.  A thread that scores its candidates one
// after the other waits for ~4 dependent round trips per term and candidate; a load under a per-candidate condition becomes
// a branch with its own wait, so every load here is unconditional on a safe index and what must not count is selected away.
// kk[c] < n_cand for every c (the caller repeats a valid candidate for the slots beyond the list).
template <int CPT>
__device__ __forceinline__ void candidate_score_batch(const NodeDev& nd, const DensDev& dn, const ItemView& v, const int* kk,
                                                      double* sk) {
  bool on[CPT];
  if (nd.kind == PCLEAN_NODE_FK) {
    long long cnt[CPT];
    double lf[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      cnt[c] = (long long)nd.counts[kk[c]];
      lf[c] = nd.logc_full[kk[c]];
    }
    const double pr_excl = (v.excl >= 0 && !v.deleted) ? nd.logc_m1[v.excl] - v.logden : -__builtin_inf();
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      on[c] = cnt[c] != 0;  // (a free slot: -inf, and no defined values to look up)
      sk[c] = !on[c] ? -__builtin_inf() : (kk[c] == v.excl ? pr_excl : lf[c] - v.logden);
    }
  } else {
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      sk[c] = nd.logc_full[kk[c]];
      on[c] = true;
    }
  }
  for (int ti = 0; ti < nd.n_terms; ++ti) {
    const TermDev& tm = nd.terms[ti];
    const int o = tm.obs_col[v.row];
    if (o < 0) continue;  // explicitly missing observation (add_typos.jl:51-53)
    int val[CPT], d[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) val[c] = tm.cand_col[kk[c]];
#pragma unroll
    for (int c = 0; c < CPT; ++c) val[c] = on[c] ? val[c] : 0;
    if (tm.ctx_slot >= 0) {
      const int32_t* fr = tm.fn + (size_t)v.ctxv[tm.ctx_slot] * tm.fn_nb;
#pragma unroll
      for (int c = 0; c < CPT; ++c) val[c] = fr[val[c]];
    }
    const size_t base = (size_t)o * tm.n_lat;
    const bool typos = tm.dens_kind != PCLEAN_DENS_EQUAL;
    int L[CPT];
    if (typos) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) L[c] = tm.lat_len[val[c]];
    }
    if (tm.elem_bytes == 1) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) d[c] = (int)tm.pair[base + val[c]];
    } else {
#pragma unroll
      for (int c = 0; c < CPT; ++c) d[c] = (int)((const uint16_t*)tm.pair)[base + val[c]];
    }
    if (!typos) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) sk[c] += !on[c] ? 0.0 : (d[c] == 0 ? 0.0 : -__builtin_inf());
      continue;
    }
    const int mt = tm.max_typos;
    double nbv[CPT], ll[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {  // term_density(): the same three operations on the same values
      const bool far = mt >= 0 && d[c] > mt;
      const int r = (L[c] + 4) / 5;
      nbv[c] = dn.nb[(size_t)r * dn.nb_stride + (far ? 0 : d[c])];
      ll[c] = dn.logl[L[c]];
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      double l = nbv[c];
      l -= ll[c] * (double)d[c];
      l -= HALF_LOG26 * (double)d[c];
      if (mt >= 0 && d[c] > mt) l = ADD_TYPOS_IMPOSSIBLE;
      sk[c] += on[c] ? l : 0.0;  // (sk is -inf where on is false: adding 0.0 changes nothing)
    }
  }
}

// likelihood terms of candidate k alone, added in candidate_score()'s order onto 0.0 — what p accumulates for a value
// that was sampled from its prior instead of being enumerated (use_dd_proposals = false, block_proposal.jl:62-64)
__device__ __forceinline__ double candidate_terms(const NodeDev& nd, const DensDev& dn, const ItemsDev& it,
                                                  const ItemView& v, int k) {
  if (v.ev_lo >= 0) return candidate_score_ev(nd, dn, it, v, k, 0.0);
  double sk = 0.0;
  for (int ti = 0; ti < nd.n_terms; ++ti) {
    const TermDev& tm = nd.terms[ti];
    const int o = tm.obs_col[v.row];
    if (o < 0) continue;
    int val = tm.cand_col[k];
    if (tm.ctx_slot >= 0) val = tm.fn[(size_t)v.ctxv[tm.ctx_slot] * tm.fn_nb + val];
    const size_t idx = (size_t)o * tm.n_lat + val;
    const int d = tm.elem_bytes == 1 ? (int)tm.pair[idx] : (int)((const uint16_t*)tm.pair)[idx];
    sk += term_density(tm, dn, d, val);
  }
  return sk;
}

// score of the "new row" candidate (proposal_compiler.jl:221-230): CRP new-table
// term + log-marginals of the children, added in plan order
__device__ __forceinline__ double new_score(const NodeDev& nd, const ChildrenDev& ch, const ItemView& v, int t) {
  double snew = 0.0;
  for (int c = 0; c < ch.n; ++c) {
    size_t idx = (size_t)t;
    if (ch.obs_col[c]) {
      const int o = ch.obs_col[c][v.row];
      idx = o < 0 ? (size_t)ch.n_obs[c] : (size_t)o;
    }
    snew += ch.arr[c][idx];
  }
  return ((v.deleted ? nd.scal[3] : nd.scal[2]) - v.logden) + snew;
}

// ---- gate of the "new row" branch ------------------------------------------------------------------
// The new-row candidate of an FK node needs the log-marginals of the node's sub-plans (its children).  For
// an item whose current referent e scores well they cannot matter: with
//     ub = CRP new-table term + sum over children of an upper bound of the child's log-marginal
// (exact cached value for cacheable leaves; log-sum of the option prior for other leaves; max(0, sum) for
// nested reference slots — every term density of the sub-tree is a probability mass <= 1), the new row's
// score is <= ub, and ub < score(e) - 28.5 <= max - 28.5 makes its fixed-point weight exactly 0
// (pclean_fixw).  Such items skip the evaluation of the children: flag 0.  flag = PCLEAN_CHOICE_NEW (the
// compaction marker) for items that need them.
template <bool EV>
__global__ void gate_new_kernel(const NodeDev nd, const DensDev dn, const ItemsDev it, const GateDev gt,
                                int32_t* __restrict__ flag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= it.n) return;
  const ItemView v = item_view(nd, it, t);
  bool need = true;
  if (v.excl >= 0 && !v.deleted) {
    const double bound = candidate_score<EV>(nd, dn, it, v, v.excl);
    double ub = nd.scal[2] - v.logden;
    for (int c = 0; c < gt.n; ++c) {
      if (gt.cache[c] && v.ev_lo < 0) {
        const int o = gt.obs_col[c][v.row];
        ub += gt.cache[c][o < 0 ? gt.n_obs[c] : o];
      } else {
        ub += gt.ub[c];
      }
    }
    need = !(ub + 1e-6 < bound - 28.5);
  }
  flag[t] = need ? PCLEAN_CHOICE_NEW : 0;
}

int pclean_launch_gate(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const GateDev& gt, int32_t* flag) {
  if (it.n <= 0) return PCLEAN_OK;
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, ctx->prob_same.p, ctx->prob_diff.p, ctx->logn.p};
  if (it.ev_lo)
    hipLaunchKernelGGL((gate_new_kernel<true>), dim3((it.n + 255) / 256), dim3(256), 0, ctx->stream, nd, dn, it, gt, flag);
  else
    hipLaunchKernelGGL((gate_new_kernel<false>), dim3((it.n + 255) / 256), dim3(256), 0, ctx->stream, nd, dn, it, gt, flag);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// ---- ev_leaf_block_kernel: a big option list (LEAF node) scored against an EVIDENCE SET -----------------------
// Latent-class sweeps re-propose every attribute of a latent row given all observed rows that refer to it
// (proposal_compiler.jl:306-350).  For an option list of 40k strings that is 40k x (evidence) densities per row
// in the generic kernels.  Same idea as root_wave.hip, with the aggregated evidence as the "rows":
//     score(k) <= prior_max - c_min * D(k),   D(k) = sum over entries (o, mult) of the node's plain AddTypos terms
//                                                     of mult * comp[o][k]      (32-bit integer arithmetic)
// pass A finds the live option with the smallest D (its exact score - 1 is a lower bound of the maximum), pass B
// keeps the options with D <= dcut; only those are scored exactly — through candidate_score(), the very
// function the generic kernels use, so the result is bit-identical.  ONE WORKGROUP of EV_T threads per latent row:
// a sub-batch of a latent sweep holds a few hundred rows (a handful for the small classes, each with the evidence
// of 10^5 observed rows), so one wavefront per row (the first version) left most of the chip idle behind a few
// stragglers.  The integer sums, the maximum and the fixed-point prefix do not depend on how the options are split
// over threads.  Rows with more than EV_SURV_CAP survivors are flagged for the generic kernel.
#define EV_T 512
#define EV_W (EV_T / 64)
#define EV_SURV_CAP 2048
#define EV_FIX_CUTOFF 28.5
#define EV_RU 2            // evidence entries whose byte-row loads are in flight together (wsum_batch)
#define EV_ENT_CAP 512     // evidence entries of an item kept in LDS (more: read through the aggregated arrays)

// Round 6: REFERENCE SLOTS too (nd.kind == PCLEAN_NODE_FK: a latent Place re-choosing its County against the ~260 observed
// rows below it): the candidates' priors are the CRP terms (candidate_score handles the excluded referent), the "new row"
// candidate (new_score: the children's marginals, evaluated before) is one more entry behind the survivors and one more
// lower bound of the maximum, and the largest prior of the inequality is the slot's (with / without an excluded reference).
// EV_QB = quads of options per thread whose byte-row loads are in flight together, MINW = waves per SIMD the register
// allocation aims at: <4, 2> one workgroup per CU with 256 registers, <2, 4> two with 128 (launches of more rows than CUs)
template <int EV_QB, int MINW>
__global__ __launch_bounds__(EV_T, MINW) void ev_leaf_block_kernel(const NodeDev nd, const DensDev dn, const ItemsDev it, const ChildrenDev ch,
                                                             const FastRootDev fr, uint64_t seed, uint32_t sweep,
                                                             uint32_t site, int n_draws, double* __restrict__ lse_out,
                                                             int32_t* __restrict__ draws_out,
                                                             int32_t* __restrict__ overflow_flag,
                                                             unsigned int* __restrict__ overflow_count,
                                                             int32_t* __restrict__ overflow_list, int prior_cut,
                                                             const uint32_t* __restrict__ dsum) {
  __shared__ uint64_t s_pref[EV_SURV_CAP + 1];  // exact scores (as doubles), then the fixed-point inclusive prefix (+ the new row)
  __shared__ int32_t s_k[EV_SURV_CAP];
  // the item's evidence entries of the plain terms, once per item: byte row of the entry's observed value and its multiplicity
  // (the passes below walk them for every quad of options: read through the aggregated arrays they were a chain of two
  // dependent loads per (quad, entry), one quad at a time)
  __shared__ uint64_t s_erow[EV_ENT_CAP];
  __shared__ uint32_t s_emul[EV_ENT_CAP];
  __shared__ uint64_t s_w64[EV_W];
  __shared__ double s_wd[EV_W];
  __shared__ int s_wi[EV_W];
  __shared__ double s_bound, s_new;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* scv = reinterpret_cast<double*>(s_pref);
  const int n = nd.n_cand;
  const bool fk = nd.kind == PCLEAN_NODE_FK;
  const int nquads = fr.kpad >> 4;
  const int draw_is = it.draw_is ? it.draw_is : n_draws;
  for (int t = blockIdx.x; t < it.n; t += gridDim.x) {
    const ItemView v = item_view(nd, it, t);
    const int oi = it.ev_item ? it.ev_item[t] : t;
    // ---- the entries -> LDS (uniform counts: every thread reads the same offsets)
    int n_ent = 0;
    if (!dsum) {
      for (int f = 0; f < fr.n_terms; ++f) {
        if (!fr.terms[f].comp) continue;
        const AggDev ag = it.agg[f];
        const int r0 = ag.off[oi], r1 = ag.end ? ag.end[oi] : ag.off[oi + 1];
        for (int r = r0 + tid; r < r1; r += EV_T) {
          const int idx = n_ent + (r - r0);
          if (idx < EV_ENT_CAP) {
            const int o = (int)(ag.key[r] & 0xffffffull) - 1;
            s_erow[idx] = o < 0 ? (uint64_t)fr.zero_row : (uint64_t)(fr.terms[f].comp + (size_t)o * fr.kpad);
            s_emul[idx] = o < 0 ? 0u : (uint32_t)ag.cnt[r];
          }
        }
        n_ent += r1 - r0;
      }
    }
    const bool ent_lds = !dsum && n_ent <= EV_ENT_CAP;
    __syncthreads();
    // weighted distance sums of the 16 options of quad q over the entries of the plain (compact-table) terms
    auto wsum = [&](int q, uint32_t* acc) {
      if (dsum) {  // summed beforehand by ev_wsum_kernel (a few latent rows with evidence sets of 10^5 rows and more)
        const uint4* ds = reinterpret_cast<const uint4*>(dsum + (size_t)t * fr.kpad) + ((size_t)q << 2);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint4 c = ds[w];
          acc[4 * w] = c.x;
          acc[4 * w + 1] = c.y;
          acc[4 * w + 2] = c.z;
          acc[4 * w + 3] = c.w;
        }
        return;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0u;
      for (int f = 0; f < fr.n_terms; ++f) {
        if (!fr.terms[f].comp) continue;
        const AggDev ag = it.agg[f];
        const int r1 = ag.end ? ag.end[oi] : ag.off[oi + 1];
        for (int r = ag.off[oi]; r < r1; ++r) {
          const uint64_t key = ag.key[r];
          const int o = (int)(key & 0xffffffull) - 1;
          if (o < 0) continue;
          const uint32_t mult = (uint32_t)ag.cnt[r];
          const uint4 c = reinterpret_cast<const uint4*>(fr.terms[f].comp + (size_t)o * fr.kpad)[q];
          const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
          for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * w + e] += mult * ((cw[w] >> (8 * e)) & 0xffu);
        }
      }
    };
    // ... of the EV_QB quads qb, qb + EV_T, ... at once: per entry the EV_QB loads are in flight together (integer sums: the
    // same whatever the order)
    auto wsum_batch = [&](int qb, uint32_t (*acc)[16]) {
      if (!ent_lds) {
#pragma unroll
        for (int u = 0; u < EV_QB; ++u)
          if (qb + u * EV_T < nquads) wsum(qb + u * EV_T, acc[u]);
        return;
      }
#pragma unroll
      for (int u = 0; u < EV_QB; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0u;
      // EV_RU entries at a time, every load unconditional (a quad beyond the list re-reads the last one — its sums are never
      // looked at —, an entry beyond the set repeats the last with multiplicity 0): one entry after the other, each load
      // behind its own condition, was a serial chain of n_ent round trips per pass — 260 of them for a Hospital's evidence
      int qs[EV_QB];
#pragma unroll
      for (int u = 0; u < EV_QB; ++u) qs[u] = min(qb + u * EV_T, nquads - 1);
      for (int r = 0; r < n_ent; r += EV_RU) {
        uint4 c[EV_RU][EV_QB];
        uint32_t mult[EV_RU];
#pragma unroll
        for (int x = 0; x < EV_RU; ++x) {
          const int rr = min(r + x, n_ent - 1);
          const uint4* row = reinterpret_cast<const uint4*>(s_erow[rr]);
          mult[x] = r + x < n_ent ? s_emul[rr] : 0u;
#pragma unroll
          for (int u = 0; u < EV_QB; ++u) c[x][u] = row[qs[u]];
        }
#pragma unroll
        for (int x = 0; x < EV_RU; ++x)
#pragma unroll
          for (int u = 0; u < EV_QB; ++u) {
            const uint32_t cw[4] = {c[x][u].x, c[x][u].y, c[x][u].z, c[x][u].w};
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[u][4 * w + e] += mult[x] * ((cw[w] >> (8 * e)) & 0xffu);
          }
      }
    };
    // ---- pass A: the live option with the smallest weighted distance -> lower bound of the maximum
    // (one_round: the whole list is one batch per thread — pass B looks at the very sums pass A made, which are kept)
    const bool one_round = nquads <= EV_QB * EV_T;
    uint32_t acc[EV_QB][16];
    uint64_t best = ~0ull;
    for (int qb = tid; qb < nquads; qb += EV_QB * EV_T) {
      wsum_batch(qb, acc);
#pragma unroll
      for (int u = 0; u < EV_QB; ++u) {
        const int q = qb + u * EV_T;
        if (q >= nquads) break;
        const uint32_t al = fr.alive[q];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const uint64_t key = ((uint64_t)acc[u][e] << 32) | (uint32_t)((q << 4) + e);
          if (((al >> e) & 1u) && key < best) best = key;
        }
      }
    }
    for (int sh = 32; sh > 0; sh >>= 1) {
      const uint64_t other = __shfl_xor(best, sh, 64);
      best = other < best ? other : best;
    }
    if (lane == 0) s_w64[wave] = best;
    __syncthreads();
    const int to = it.out_pos ? it.out_pos[t] : t;
    if (tid == 0) {
      for (int w = 1; w < EV_W; ++w) best = s_w64[w] < best ? s_w64[w] : best;
      double bd = best != ~0ull ? candidate_score(nd, dn, it, v, (int)(uint32_t)best) - 1.0 : -__builtin_inf();
      double sn = -__builtin_inf();
      if (fk) {  // the new-row candidate is a candidate too: its exact score bounds the maximum from below
        sn = new_score(nd, ch, v, to);
        bd = fmax(bd, sn);
      }
      s_new = sn;
      s_bound = bd;
    }
    __syncthreads();
    const double bound = s_bound;
    uint32_t dcut = 0xffffffffu;
    const double* pri = nullptr;  // per-option priors of the refined cut (null: no usable bound)
    if (fr.inv_c > 0.0 && bound > -__builtin_inf()) {
      const double x = ((fk && v.excl >= 0 ? fr.prior_max_e : fr.prior_max_n) - bound + EV_FIX_CUTOFF) * fr.inv_c;
      if (x >= 0.0 && x < 4.0e9) dcut = (uint32_t)x + 2u;
      if (prior_cut) pri = (fk && v.excl >= 0) ? fr.prior_e : fr.prior_n;
    }
    // ---- pass B: survivors in ascending option order (rounds of EV_T quads, EV_QB rounds summed at a time)
    int ns = 0;
    for (int q0 = 0; q0 < nquads; q0 += EV_QB * EV_T) {
      uint32_t masks[EV_QB];
      {
        if (!one_round) wsum_batch(q0 + tid, acc);  // (one_round: pass A's sums of the same quads; a thread without a quad masks nothing)
#pragma unroll
        for (int u = 0; u < EV_QB; ++u) {
          const int q = q0 + u * EV_T + tid;
          uint32_t mask16 = 0;
          if (q < nquads) {
#pragma unroll
            for (int e = 0; e < 16; ++e) mask16 |= (acc[u][e] <= dcut ? 1u : 0u) << e;
            mask16 &= (uint32_t)fr.alive[q];
            // the same inequality with the option's OWN prior in place of the largest one: score(k) <= prior(k) - c_min D(k).
            // A latent row with one or two referring rows has a weak bound (the best option's letter-model prior is tens of
            // nats below the shortest string's): under the common cut every option within ~20 edits survived and the row went
            // to the generic kernel (a fifth of a Hospital sub-batch's time for 7 % of its rows)
            if (pri && mask16) {
#pragma unroll
              for (int e = 0; e < 16; ++e)
                if ((mask16 >> e) & 1u) {
                  const double x = (pri[(q << 4) + e] - bound + EV_FIX_CUTOFF) * fr.inv_c + 2.0;
                  if (!((double)acc[u][e] <= x)) mask16 &= ~(1u << e);
                }
            }
          }
          masks[u] = mask16;
        }
      }
#pragma unroll
      for (int u = 0; u < EV_QB; ++u) {
        if (q0 + u * EV_T >= nquads) break;  // (uniform)
        const int q = q0 + u * EV_T + tid;
        const uint32_t mask16 = masks[u];
        const int cnt = __builtin_popcount(mask16);
        int incl = cnt;
        for (int sh = 1; sh < 64; sh <<= 1) {
          const int x = __shfl_up(incl, sh, 64);
          if (lane >= sh) incl += x;
        }
        if (lane == 63) s_wi[wave] = incl;
        __syncthreads();
        int pos = ns + incl - cnt, round_total = 0;
        for (int w = 0; w < EV_W; ++w) {
          if (w < wave) pos += s_wi[w];
          round_total += s_wi[w];
        }
        for (uint32_t mm = mask16; mm; mm &= mm - 1) {
          if (pos < EV_SURV_CAP) s_k[pos] = (q << 4) + __builtin_ctz(mm);
          ++pos;
        }
        ns += round_total;
        __syncthreads();  // s_wi is rewritten by the next round
      }
    }
    if (ns > EV_SURV_CAP) {
      if (tid == 0) {
        overflow_flag[to] = PCLEAN_CHOICE_NEW;
        const unsigned int at = atomicAdd(overflow_count, 1u);
        if (overflow_list) overflow_list[at] = t;  // (item index: the re-run reads the same item arrays)
      }
      continue;  // (every thread: ns is uniform; the loop's first barrier follows writes to other arrays only)
    }
    // ---- exact scores (the generic kernels' own function), maximum, fixed-point prefix (flags are pre-zeroed)
    double m = -__builtin_inf();
    for (int j = tid; j < ns; j += EV_T) {
      const double sc = candidate_score(nd, dn, it, v, s_k[j]);
      scv[j] = sc;
      m = fmax(m, sc);
    }
    const int n_e = ns + (fk ? 1 : 0);  // entries of the prefix: the survivors in ascending order, then the new row
    if (fk && tid == 0) {
      scv[ns] = s_new;
      m = fmax(m, s_new);
    }
    __syncthreads();  // (scv[ns] is read by the thread that owns entry ns below)
    m = wave_max(m);
    if (lane == 0) s_wd[wave] = m;
    __syncthreads();
    m = s_wd[0];
    for (int w = 1; w < EV_W; ++w) m = fmax(m, s_wd[w]);
    uint64_t carry = 0;
    for (int j0 = 0; j0 < n_e; j0 += EV_T) {
      const int j = j0 + tid;
      const uint64_t u = (j < n_e && m != -__builtin_inf()) ? pclean_fixw(scv[j] - m) : 0ull;
      unsigned long long incl = u;
      for (int sh = 1; sh < 64; sh <<= 1) {
        const unsigned long long x = __shfl_up(incl, sh, 64);
        if (lane >= sh) incl += x;
      }
      __syncthreads();  // the previous chunk's readers of s_w64 are done
      if (lane == 63) s_w64[wave] = incl;
      __syncthreads();
      uint64_t base = carry, chunk_total = 0;
      for (int w = 0; w < EV_W; ++w) {
        if (w < wave) base += s_w64[w];
        chunk_total += s_w64[w];
      }
      if (j < n_e) s_pref[j] = base + incl;  // entry j: read as a score by this thread only, above
      carry += chunk_total;
    }
    const uint64_t U = carry;
    __syncthreads();
    if (tid == 0) {
      if (lse_out) lse_out[to] = pclean_lse_from_fix(m, U);
      if (n_draws > 0) {
        int32_t res = fk ? PCLEAN_CHOICE_NEW : n - 1;
        if (U != 0) {
          const uint32_t rng_row = it.rng_row ? (uint32_t)it.rng_row[t] : (uint32_t)((int64_t)v.row + it.row_offset);
          const uint32_t pid = it.particle ? (uint32_t)it.particle[t] : 0u;
          const uint64_t x = pclean_mulhi64(pclean_rand64(seed, rng_row, site, pid, sweep), U);
          int a = 0, b = n_e - 1;  // smallest index with prefix > x
          while (a < b) {
            const int mid = (a + b) >> 1;
            if (s_pref[mid] > x)
              b = mid;
            else
              a = mid + 1;
          }
          res = (fk && a == ns) ? PCLEAN_CHOICE_NEW : s_k[a];
        }
        draws_out[(size_t)to * draw_is] = res;
      }
    }
    __syncthreads();  // the next row overwrites s_k / s_pref
  }
}

// The weighted distance sums D(k) of ev_leaf_block_kernel for launches of a FEW latent rows with huge evidence sets (the one
// HospitalType row that 10^6 observed rows refer to: options x distinct evidence entries = 4 x 10^8 byte products in ONE
// workgroup, 12 ms): a grid over (quads of options, item, slices of the evidence entries), every thread adds its slice's
// share of 16 options with integer atomics — the sums are the same whatever the split.
__global__ __launch_bounds__(256) void ev_wsum_kernel(const FastRootDev fr, const ItemsDev it, uint32_t* __restrict__ dsum) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
  const int nquads = fr.kpad >> 4;
  if (q >= nquads) return;
  const int oi = it.ev_item ? it.ev_item[t] : t;
  uint32_t acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0u;
  bool any = false;
  for (int f = 0; f < fr.n_terms; ++f) {
    if (!fr.terms[f].comp) continue;
    const AggDev ag = it.agg[f];
    const int r1 = ag.end ? ag.end[oi] : ag.off[oi + 1];
    for (int r = ag.off[oi] + (int)blockIdx.z; r < r1; r += (int)gridDim.z) {
      const uint64_t key = ag.key[r];
      const int o = (int)(key & 0xffffffull) - 1;
      if (o < 0) continue;
      const uint32_t mult = (uint32_t)ag.cnt[r];
      const uint4 c = reinterpret_cast<const uint4*>(fr.terms[f].comp + (size_t)o * fr.kpad)[q];
      const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * w + e] += mult * ((cw[w] >> (8 * e)) & 0xffu);
      any = true;
    }
  }
  if (!any) return;
  uint32_t* d = dsum + (size_t)t * fr.kpad + ((size_t)q << 4);
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if (acc[e]) atomicAdd(&d[e], acc[e]);
}

int pclean_launch_ev_leaf(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const FastRootDev& fr, uint64_t seed,
                          uint32_t sweep, uint32_t site, int n_draws, double* lse_out, int32_t* draws_out,
                          int32_t* overflow_flag, unsigned int* overflow_count, int32_t* overflow_list, const ChildrenDev* ch,
                          uint32_t* dsum) {
  if (it.n <= 0) return PCLEAN_OK;
  if (n_draws > 1) return pclean_fail(ctx, PCLEAN_ERR_ARG, "evidence-set option lists draw at most once per item");
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, ctx->prob_same.p, ctx->prob_diff.p, ctx->logn.p};
  const int wgs = std::min(256 * 8, it.n);
  static const bool no_prior_cut = getenv("PCLEAN_NO_EV_PRIOR_CUT") != nullptr;  // (A/B switch: results are bit-identical)
  if ((nd.kind == PCLEAN_NODE_FK) != (ch != nullptr))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "evidence-set scan: a reference slot comes with its children's marginals, an option list without");
  const ChildrenDev none{};
  if (dsum) {  // (zeroed by the caller) the sums first, over the whole chip
    const int nquads = fr.kpad >> 4;
    const int slices = std::max(1, std::min(64, 2048 / std::max(it.n * ((nquads + 255) / 256), 1)));
    hipLaunchKernelGGL(ev_wsum_kernel, dim3((nquads + 255) / 256, it.n, slices), dim3(256), 0, ctx->stream, fr, it, dsum);
  }
  static const int force_qb = getenv("PCLEAN_EV_QB") ? atoi(getenv("PCLEAN_EV_QB")) : 0;  // (A/B: 1 = the one-quad walk, 2, 4)
  const int qb = force_qb ? force_qb : (it.n > 256 ? 2 : 4);
  auto kern = qb == 1 ? ev_leaf_block_kernel<1, 3> : qb == 2 ? ev_leaf_block_kernel<2, 4> : ev_leaf_block_kernel<4, 2>;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(EV_T), 0, ctx->stream, nd, dn, it, ch ? *ch : none, fr, seed, sweep, site, n_draws,
                     lse_out, draws_out, overflow_flag, overflow_count, overflow_list, no_prior_cut ? 0 : 1, (const uint32_t*)dsum);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// exclusive block scan of one uint64 per lane (NW waves); returns lane prefix, sets total
template <int NW = 4>
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t part, uint64_t* wsum, uint64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = part;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long x = __shfl_up(incl, o, 64);
    if (lane >= o) incl += x;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint64_t base = 0;
  uint64_t tot = 0;
  for (int w = 0; w < NW; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  *total = tot;
  return base + incl - part;
}

template <int BT, bool EV>
__global__ __launch_bounds__(BT) void enum_node_kernel(const NodeDev nd, const DensDev dn, const ItemsDev it,
                                                        const ChildrenDev ch, uint64_t seed, uint32_t sweep,
                                                        uint32_t site, int n_draws, int item_base,
                                                        double* __restrict__ lse_out,
                                                        double* __restrict__ scores_out,
                                                        int32_t* __restrict__ draws_out,
                                                        const double* __restrict__ scores_in, int slot_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  int g = blockIdx.x + item_base;
  const int slot_in = g;  // (scores_in: row of the scores enum_scores_kernel left for this workgroup)
  if (it.sel) {  // (never together with groups)
    if ((unsigned int)g >= *it.sel_n) return;
    g = it.sel[g];
  }
  const int m_lo = it.grp_off ? it.grp_off[g] : g, m_hi = it.grp_off ? it.grp_off[g + 1] : g + 1;
  const int t = it.grp_off ? it.members[m_lo] : g;  // the item whose scores stand for the whole group
  const int to = it.out_pos ? it.out_pos[t] : t;    // output slot of this item
  const int n = nd.n_cand;
  const bool fk = nd.kind == PCLEAN_NODE_FK;
  const int nc = n + (fk ? 1 : 0);
  double* s = (double*)smem;                                    // [nc]
  uint64_t* u = (uint64_t*)smem;                                // same storage, after phase 3
  double* red = (double*)(smem + (size_t)((nc + 1) & ~1) * 8);  // [16]
  uint64_t* wsum = (uint64_t*)(red + 16);                       // [16]
  const ItemView v = item_view(nd, it, t);

  // ---- phase 1: scores ----------------------------------------------------
  double lmax = -__builtin_inf();
  if (scores_in && slot_in < slot_cap) {  // computed beforehand by enum_scores_kernel, one candidate per thread over the whole chip
    const double* si = scores_in + (size_t)slot_in * nc;
    for (int k = tid; k < nc; k += BT) {
      const double sk = si[k];
      s[k] = sk;
      lmax = fmax(lmax, sk);
    }
  } else {
    if ((!EV || v.ev_lo < 0) && !nd.g.on && n > BT) {  // several candidates per thread: their gather chains level by level
      for (int k0 = tid; k0 < n; k0 += BT * ENUM_CPT) {
        int kk[ENUM_CPT];
        double sc[ENUM_CPT];
#pragma unroll
        for (int c = 0; c < ENUM_CPT; ++c) kk[c] = k0 + c * BT < n ? k0 + c * BT : k0;
        candidate_score_batch<ENUM_CPT>(nd, dn, v, kk, sc);
#pragma unroll
        for (int c = 0; c < ENUM_CPT; ++c)
          if (k0 + c * BT < n) {
            s[kk[c]] = sc[c];
            if (scores_out) scores_out[(size_t)to * nc + kk[c]] = sc[c];
            lmax = fmax(lmax, sc[c]);
          }
      }
    } else {
      for (int k = tid; k < n; k += BT) {
        const double sk = candidate_score<EV>(nd, dn, it, v, k);
        s[k] = sk;
        if (scores_out) scores_out[(size_t)to * nc + k] = sk;
        lmax = fmax(lmax, sk);
      }
    }
    if (fk && tid == 0) {
      const double sn = new_score(nd, ch, v, to);
      s[n] = sn;
      if (scores_out) scores_out[(size_t)to * nc + n] = sn;
      lmax = fmax(lmax, sn);
    }
  }
  // ---- phase 2: max ----------------------------------------------------------
  lmax = wave_max(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  double m = red[0];
  for (int w = 1; w < BT / 64; ++w) m = fmax(m, red[w]);
  // ---- phase 3: fixed-point weights in place ----------------------------------
  for (int k = tid; k < nc; k += BT) {
    const double sk = s[k];
    u[k] = (m == -__builtin_inf()) ? 0ull : pclean_fixw(sk - m);
  }
  __syncthreads();
  // ---- phase 4: chunk sums + block scan ---------------------------------------
  const int chunk = (nc + BT - 1) / BT;
  const int lo = min(tid * chunk, nc), hi = min(lo + chunk, nc);
  uint64_t part = 0;
  for (int k = lo; k < hi; ++k) part += u[k];
  uint64_t U;
  const uint64_t pre = block_excl_scan<BT / 64>(part, wsum, &U);
  // ---- phase 5: lse + draws of every member item of the group -------------------------------------
  // The lane's chunk becomes an inclusive prefix in place; every (member item, draw) pair then locates
  // its threshold by binary search (smallest k with prefix[k] > x — a zero-weight candidate is never hit,
  // exactly as in the sequential scan of the oracle), one pair per lane, no further barriers.
  {
    uint64_t run = pre;
    for (int k = lo; k < hi; ++k) {
      run += u[k];
      u[k] = run;
    }
  }
  __syncthreads();
  const double lse = pclean_lse_from_fix(m, U);
  const int nd_eff = n_draws > 0 ? n_draws : 1;
  const int n_out = (m_hi - m_lo) * nd_eff;
  const int draw_is = it.draw_is ? it.draw_is : n_draws, draw_ds = it.draw_ds ? it.draw_ds : 1;
  for (int q = tid; q < n_out; q += BT) {
    const int mi = m_lo + q / nd_eff, j = q % nd_eff;
    const int tm = it.grp_off ? it.members[mi] : t;
    const int tom = it.out_pos ? it.out_pos[tm] : tm;
    if (j == 0 && lse_out) lse_out[tom] = lse;
    if (n_draws <= 0) continue;
    int32_t res = fk ? PCLEAN_CHOICE_NEW : n - 1;
    if (U != 0) {
      const int row_m = it.row ? it.row[tm] : tm;
      const uint32_t rng_row = it.rng_row ? (uint32_t)it.rng_row[tm] : (uint32_t)((int64_t)row_m + it.row_offset);
      const uint32_t pid = it.particle ? (uint32_t)it.particle[tm] : (uint32_t)j;
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
    draws_out[(size_t)tom * draw_is + (size_t)j * draw_ds] = res;
  }
}

// Few items with long candidate lists and evidence sets (a latent sub-batch's rows the evidence scan could not settle, the
// generic launches of its other nodes): one workgroup per item walks ~10 candidates per thread, each a chain of dependent
// gathers through the aggregated evidence and the pair tables — 60-100 us of latency with most of the chip idle.  Here the
// exact scores come from ONE CANDIDATE PER THREAD over (item, candidate) — the same candidate_score() / new_score(), so the
// same bits — and enum_node_kernel's workgroup per item starts from them (scores_in).  Workgroup (x, y): candidates
// 256 x .. 256 x + 255 of the slots y, y + gridDim.y, ...; an indirect launch stops at the list's length.
template <bool EV>
__global__ __launch_bounds__(256) void enum_scores_kernel(const NodeDev nd, const DensDev dn, const ItemsDev it, const ChildrenDev ch,
                                                          double* __restrict__ scores, int slot_cap) {
  const int n = nd.n_cand;
  const bool fk = nd.kind == PCLEAN_NODE_FK;
  const int nc = n + (fk ? 1 : 0);
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nc) return;
  // (slot_cap: rows of `scores`; the workgroups of later slots compute their scores themselves)
  const int n_slots = min(it.sel ? (int)min(*it.sel_n, (unsigned int)it.n) : it.n, slot_cap);
  for (int slot = blockIdx.y; slot < n_slots; slot += gridDim.y) {
    const int t = it.sel ? it.sel[slot] : slot;
    const ItemView v = item_view(nd, it, t);
    double sk;
    if (k < n) {
      sk = candidate_score<EV>(nd, dn, it, v, k);
    } else {
      const int to = it.out_pos ? it.out_pos[t] : t;
      sk = new_score(nd, ch, v, to);
    }
    scores[(size_t)slot * nc + k] = sk;
  }
}

// BT threads per item: 256 for launches that fill the chip, 1024 for the few-item launches of the latent sweeps (the
// re-run of a sub-batch's overflowed rows: a dozen workgroups, each walking the whole option list three times)
template <int BT, bool EV>
__global__ __launch_bounds__(BT) void enum_node_big_kernel(const NodeDev nd, const DensDev dn, const ItemsDev it,
                                                            const ChildrenDev ch, uint64_t seed, uint32_t sweep,
                                                            uint32_t site, int n_draws, int item_base,
                                                            double* __restrict__ lse_out,
                                                            double* __restrict__ scores_out,
                                                            int32_t* __restrict__ draws_out,
                                                            const double* __restrict__ scores_in, int slot_cap) {
  __shared__ double red[BT / 64];
  __shared__ uint64_t wsum[BT / 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x + item_base;
  const int slot_in = t;  // (scores_in: row of the scores enum_scores_kernel left for this workgroup)
  if (it.sel) {
    if ((unsigned int)t >= *it.sel_n) return;
    t = it.sel[t];
  }
  const int to = it.out_pos ? it.out_pos[t] : t;  // output slot of this item
  const int n = nd.n_cand;
  const bool fk = nd.kind == PCLEAN_NODE_FK;
  const int nc = n + (fk ? 1 : 0);
  const ItemView v = item_view(nd, it, t);
  // the three passes below re-compute a score wherever they need it — or read it from the row enum_scores_kernel left (a list
  // beyond LDS of a few-item launch: 40 candidates per thread and pass, each a chain of gathers, became three streamed reads)
  const double* si = (scores_in && slot_in < slot_cap) ? scores_in + (size_t)slot_in * nc : nullptr;
  const double sn = !fk ? -__builtin_inf() : (si ? si[n] : new_score(nd, ch, v, to));

  // pass A: max (lane-strided, coalesced)
  double lmax = -__builtin_inf();
  for (int k = tid; k < n; k += BT) {
    const double sk = si ? si[k] : candidate_score<EV>(nd, dn, it, v, k);
    if (scores_out) scores_out[(size_t)to * nc + k] = sk;
    lmax = fmax(lmax, sk);
  }
  if (fk) {
    lmax = fmax(lmax, sn);
    if (scores_out && tid == 0) scores_out[(size_t)to * nc + n] = sn;
  }
  lmax = wave_max(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  double m = red[0];
  for (int w = 1; w < BT / 64; ++w) m = fmax(m, red[w]);

  // pass B: fixed-point weights over contiguous per-lane chunks (natural order prefix)
  const int chunk = (nc + BT - 1) / BT;
  const int lo = min(tid * chunk, nc), hi = min(lo + chunk, nc);
  uint64_t part = 0;
  if (m != -__builtin_inf())
    for (int k = lo; k < hi; ++k) {
      const double sk = (k == n) ? sn : (si ? si[k] : candidate_score<EV>(nd, dn, it, v, k));
      part += pclean_fixw(sk - m);
    }
  uint64_t U;
  const uint64_t pre = block_excl_scan<BT / 64>(part, wsum, &U);
  if (tid == 0 && lse_out) lse_out[to] = pclean_lse_from_fix(m, U);

  // pass C: draws, located by recomputing the owning lane's chunk
  if (n_draws > 0) {
    const uint32_t rng_row = it.rng_row ? (uint32_t)it.rng_row[t] : (uint32_t)((int64_t)v.row + it.row_offset);
    for (int j = 0; j < n_draws; ++j) {
      const uint32_t pid = it.particle ? (uint32_t)it.particle[t] : (uint32_t)j;
      int32_t* dst = draws_out + (size_t)to * (it.draw_is ? it.draw_is : n_draws) + (size_t)j * (it.draw_ds ? it.draw_ds : 1);
      if (U == 0) {
        if (tid == 0) *dst = fk ? PCLEAN_CHOICE_NEW : n - 1;
        continue;
      }
      const uint64_t R = pclean_rand64(seed, rng_row, site, pid, sweep);
      const uint64_t x = pclean_mulhi64(R, U);
      if (x >= pre && x < pre + part) {
        uint64_t acc = pre;
        int k = lo;
        for (; k < hi; ++k) {
          const double sk = (k == n) ? sn : (si ? si[k] : candidate_score<EV>(nd, dn, it, v, k));
          acc += pclean_fixw(sk - m);
          if (acc > x) break;
        }
        *dst = (fk && k == n) ? PCLEAN_CHOICE_NEW : k;
      }
    }
  }
}

// ---- cacheable option lists: per-observed-value (maximum, fixed-point total, coarse prefix) -----------------------
// A cacheable LEAF (one term on one observed column, no ctx) has a score vector that depends on the evidence row only
// through that row's observed value o.  leaf_coarse_build_kernel enumerates every o once (item t "observes" value t,
// the last item a missing value): maximum m[o], total U[o] and the inclusive fixed-point prefix at the end of every
// block of LEAF_CB consecutive options.  A draw then needs one binary search over the coarse prefix and the exact
// weights of ONE block of options (leaf_coarse_draw_kernel) instead of three passes over the whole list — the same
// index "min{k : prefix_k > x}" because integer sums do not depend on how they are bracketed.  The log-marginal
// m + log(U 2^-40) is what the generic kernels return, bit for bit.
#define LEAF_CB 256

template <bool EV>
__global__ __launch_bounds__(256) void leaf_coarse_build_kernel(const NodeDev nd, const DensDev dn, const ItemsDev it,
                                                                int n_blocks, double* __restrict__ lse_out,
                                                                double* __restrict__ m_out, uint64_t* __restrict__ U_out,
                                                                uint64_t* __restrict__ coarse, int dummy_k,
                                                                uint64_t* __restrict__ udummy_out) {
  __shared__ double red[4];
  __shared__ uint64_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.x;
  const int n = nd.n_cand;
  const ItemView v = item_view(nd, it, t);
  double lmax = -__builtin_inf();
  for (int k = tid; k < n; k += 256) lmax = fmax(lmax, candidate_score<EV>(nd, dn, it, v, k));
  lmax = wave_max(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  uint64_t run = 0;
  for (int b = 0; b < n_blocks; ++b) {
    const int k = b * LEAF_CB + tid;
    uint64_t u = 0;
    if (k < n && m != -__builtin_inf()) u = pclean_fixw(candidate_score<EV>(nd, dn, it, v, k) - m);
    if (k == dummy_k && udummy_out) udummy_out[t] = u;  // weight of the ProposalDummyValue option (0: it cannot be drawn)
    for (int o = 32; o > 0; o >>= 1) u += __shfl_xor((unsigned long long)u, o, 64);
    __syncthreads();  // wsum of the previous block has been read
    if (lane == 0) wsum[wave] = u;
    __syncthreads();
    run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (tid == 0) coarse[(size_t)t * n_blocks + b] = run;
  }
  if (tid == 0) {
    m_out[t] = m;
    U_out[t] = run;
    lse_out[t] = pclean_lse_from_fix(m, run);
  }
}

// one wavefront per (item, draw): coarse search, then the exact weights of the LEAF_CB options of the block found
template <bool EV>
__global__ __launch_bounds__(256) void leaf_coarse_draw_kernel(const NodeDev nd, const DensDev dn, const ItemsDev it,
                                                               const int32_t* __restrict__ obs_col, int n_obs,
                                                               int n_blocks, const double* __restrict__ lse_c,
                                                               const double* __restrict__ m_c,
                                                               const uint64_t* __restrict__ U_c,
                                                               const uint64_t* __restrict__ coarse, uint64_t seed,
                                                               uint32_t sweep, uint32_t site, int n_draws,
                                                               double* __restrict__ lse_out,
                                                               int32_t* __restrict__ draws_out) {
  const int lane = threadIdx.x & 63;
  const int nd_eff = n_draws > 0 ? n_draws : 1;  // n_draws == 0: log-marginals only
  const long long n_pairs = (long long)it.n * nd_eff;
  const int n = nd.n_cand;
  const int draw_is = it.draw_is ? it.draw_is : n_draws, draw_ds = it.draw_ds ? it.draw_ds : 1;
  for (long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_pairs; q += (long long)gridDim.x * 4) {
    const int t = (int)(q / nd_eff), j = (int)(q - (long long)t * nd_eff);
    const ItemView v = item_view(nd, it, t);
    const int to = it.out_pos ? it.out_pos[t] : t;
    const int ov = obs_col[v.row];
    const int o = ov < 0 ? n_obs : ov;
    if (j == 0 && lse_out && lane == 0) lse_out[to] = lse_c[o];
    if (n_draws <= 0) continue;
    const uint64_t U = U_c[o];
    int32_t res = n - 1;
    if (U != 0) {
      const double m = m_c[o];
      const uint32_t rng_row = it.rng_row ? (uint32_t)it.rng_row[t] : (uint32_t)((int64_t)v.row + it.row_offset);
      const uint32_t pid = it.particle ? (uint32_t)it.particle[t] : (uint32_t)j;
      const uint64_t x = pclean_mulhi64(pclean_rand64(seed, rng_row, site, pid, sweep), U);
      const uint64_t* cp = coarse + (size_t)o * n_blocks;
      int a = 0, b = n_blocks - 1;  // smallest block whose inclusive prefix exceeds x
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (cp[mid] > x)
          b = mid;
        else
          a = mid + 1;
      }
      const uint64_t base = a > 0 ? cp[a - 1] : 0ull;
      // lane l: options 4 l .. 4 l + 3 of the block, natural order
      uint64_t u4[4];
      uint64_t mine = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = a * LEAF_CB + lane * 4 + e;
        u4[e] = k < n ? pclean_fixw(candidate_score<EV>(nd, dn, it, v, k) - m) : 0ull;
        mine += u4[e];
      }
      unsigned long long incl = mine;
      for (int sh = 1; sh < 64; sh <<= 1) {
        const unsigned long long y = __shfl_up(incl, sh, 64);
        if (lane >= sh) incl += y;
      }
      uint64_t acc = base + incl - mine;
      int found = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += u4[e];
        if (found == 0x7fffffff && acc > x) found = a * LEAF_CB + lane * 4 + e;
      }
      for (int sh = 32; sh > 0; sh >>= 1) found = min(found, __shfl_xor(found, sh, 64));
      if (found != 0x7fffffff) res = found;
    }
    if (lane == 0) draws_out[(size_t)to * draw_is + (size_t)j * draw_ds] = res;
  }
}

int pclean_launch_leaf_coarse_build(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, int n_blocks, double* lse_out,
                                    double* m_out, uint64_t* U_out, uint64_t* coarse, int dummy_k, uint64_t* udummy_out) {
  if (it.n <= 0) return PCLEAN_OK;
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, ctx->prob_same.p, ctx->prob_diff.p, ctx->logn.p};
  if (it.ev_lo)
    hipLaunchKernelGGL((leaf_coarse_build_kernel<true>), dim3(it.n), dim3(256), 0, ctx->stream, nd, dn, it, n_blocks, lse_out, m_out,
                     U_out, coarse, dummy_k, udummy_out);
  else
    hipLaunchKernelGGL((leaf_coarse_build_kernel<false>), dim3(it.n), dim3(256), 0, ctx->stream, nd, dn, it, n_blocks, lse_out, m_out,
                     U_out, coarse, dummy_k, udummy_out);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
int pclean_leaf_coarse_blocks(int n_options) { return (n_options + LEAF_CB - 1) / LEAF_CB; }

int pclean_launch_leaf_coarse_draw(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const int32_t* obs_col, int n_obs,
                                   int n_blocks, const double* lse_c, const double* m_c, const uint64_t* U_c,
                                   const uint64_t* coarse, uint64_t seed, uint32_t sweep, uint32_t site, int n_draws,
                                   double* lse_out, int32_t* draws_out) {
  if (it.n <= 0) return PCLEAN_OK;
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, ctx->prob_same.p, ctx->prob_diff.p, ctx->logn.p};
  const long long pairs = (long long)it.n * std::max(n_draws, 1);
  const int wgs = (int)std::min<long long>((pairs + 3) / 4, 256 * 8);
  if (it.ev_lo)
    hipLaunchKernelGGL((leaf_coarse_draw_kernel<true>), dim3(wgs), dim3(256), 0, ctx->stream, nd, dn, it, obs_col, n_obs, n_blocks,
                     lse_c, m_c, U_c, coarse, seed, sweep, site, n_draws, lse_out, draws_out);
  else
    hipLaunchKernelGGL((leaf_coarse_draw_kernel<false>), dim3(wgs), dim3(256), 0, ctx->stream, nd, dn, it, obs_col, n_obs, n_blocks,
                     lse_c, m_c, U_c, coarse, seed, sweep, site, n_draws, lse_out, draws_out);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// ---- prior proposals (use_dd_proposals = false): weight of a particle = likelihood of its sampled sub-tree -------
// One thread per particle slot (slot = particle * N + row).  choice[slot] >= 0: the terms of the block's root at that
// referent; NEW: the nodes of the sampled new row, children summed in plan order inside their parent (the oracle's
// recursion, sweep.h: subtree_terms; node ids are in pre-order, so a reverse pass sees the children first).
#define PRIOR_MAX_NODES 64
__global__ void prior_terms_kernel(size_t n_slots, int N, int n_nodes, const NodeDev* __restrict__ nds,
                                   const int32_t* __restrict__ n_children, const int32_t* __restrict__ child_begin,
                                   const int32_t* __restrict__ children, DensDev dn, const int32_t* __restrict__ pchoice,
                                   const int32_t* __restrict__ pnewpos, const int32_t* __restrict__ vals,
                                   const int32_t* __restrict__ it_ctx, double* __restrict__ w) {
  const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_slots) return;
  int32_t cv[PCLEAN_MAX_CTX];
  for (int c = 0; c < PCLEAN_MAX_CTX; ++c) cv[c] = it_ctx ? it_ctx[(size_t)c * n_slots + slot] : 0;
  ItemsDev it{};
  ItemView v{};
  v.row = (int)(slot % (size_t)N);
  v.item = v.row;
  v.excl = -1;
  v.ctxv = cv;
  v.ev_lo = v.ev_hi = -1;
  const int c0 = pchoice[slot];
  double L;
  if (c0 >= 0) {
    L = candidate_terms(nds[0], dn, it, v, c0);
  } else {
    const int32_t* vv = vals + (size_t)pnewpos[slot] * n_nodes;
    double acc[PRIOR_MAX_NODES];
    for (int node = n_nodes - 1; node >= 0; --node) {
      const int k = node == 0 ? PCLEAN_CHOICE_NEW : vv[node];
      double a = 0.0;
      if (k >= 0) {
        a = candidate_terms(nds[node], dn, it, v, k);
      } else if (k == PCLEAN_CHOICE_NEW) {
        for (int c = 0; c < n_children[node]; ++c) a += acc[children[child_begin[node] + c]];
      }
      acc[node] = a;
    }
    L = acc[0];
  }
  w[slot] += L;
}

int pclean_launch_prior_terms(pclean_ctx* ctx, size_t n_slots, int N, int n_nodes, const NodeDev* nds,
                              const int32_t* n_children, const int32_t* child_begin, const int32_t* children,
                              const int32_t* pchoice, const int32_t* pnewpos, const int32_t* vals, const int32_t* it_ctx,
                              double* w) {
  if (n_slots == 0) return PCLEAN_OK;
  if (n_nodes > PRIOR_MAX_NODES) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "prior proposals: more than %d plan nodes", PRIOR_MAX_NODES);
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, ctx->prob_same.p, ctx->prob_diff.p, ctx->logn.p};
  hipLaunchKernelGGL(prior_terms_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, ctx->stream, n_slots, N,
                     n_nodes, nds, n_children, child_begin, children, dn, pchoice, pnewpos, vals, it_ctx, w);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// ---- prior proposals for the rows of a LATENT class: likelihood of the referring rows given particle p's values ---
// slot = item * P + particle; vals[slot][node] = the particle's choices (option / referent / NEW / -2 unused).
// roots in order; each root's sub-tree as in prior_terms_kernel, every term summed over the item's evidence set
// (candidate_score_ev onto 0.0).  aggs[node] = aggregated evidence of the node's terms (latent.hip: ensure_agg).
__global__ void prior_terms_ev_kernel(int n_items, int P, int n_nodes, const NodeDev* __restrict__ nds,
                                      const AggDev* const* __restrict__ aggs, const int32_t* __restrict__ n_children,
                                      const int32_t* __restrict__ child_begin, const int32_t* __restrict__ children,
                                      int n_roots, const int32_t* __restrict__ roots, DensDev dn, ItemsDev it,
                                      const int32_t* __restrict__ vals, double* __restrict__ w) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_items * P) return;
  const int t = slot / P;
  ItemView v{};
  v.item = t;
  v.row = 0;
  v.excl = -1;
  // (as item_view() does it: a RUN-TIME null.  With a literal nullptr here the Gaussian evidence term — whose value lambda has
  // a `v.ctxv[...]` case for sources this plan does not have — faulted on address 0 although that case is never taken)
  v.ctxv = it.ctx ? it.ctx + (size_t)t * PCLEAN_MAX_CTX : nullptr;
  v.ev_lo = it.ev_lo[t];
  v.ev_hi = it.ev_hi[t];
  const int32_t* vv = vals + (size_t)slot * n_nodes;
  double acc[PRIOR_MAX_NODES];
  for (int node = n_nodes - 1; node >= 0; --node) {
    const int k = vv[node];
    double a = 0.0;
    if (k >= 0) {
      ItemsDev itn = it;
      itn.agg = aggs[node];
      a = candidate_score_ev(nds[node], dn, itn, v, k, 0.0);
    } else if (k == PCLEAN_CHOICE_NEW) {
      for (int c = 0; c < n_children[node]; ++c) a += acc[children[child_begin[node] + c]];
    }
    acc[node] = a;
  }
  double L = 0.0;
  for (int r = 0; r < n_roots; ++r) L += acc[roots[r]];
  w[slot] = L;
}

// fault hunting (PCLEAN_DEBUG_LATENT): the indices the Gaussian evidence term of `node` would dereference, range-checked
// instead of dereferenced.  out[slot] = {k, n_cand, ev_lo, ev_hi, rows out of range, smallest / largest mean index, flags}
__global__ void gauss_ev_probe_kernel(int n_slots, int P, int n_nodes, const NodeDev* __restrict__ nds, int node, ItemsDev it,
                                      const int32_t* __restrict__ vals, int n_rows, int n_mean, int n_ev, long long* __restrict__ out, int mode) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_slots) return;
  const NodeDev& nd = nds[node];
  const GaussDev& g = nd.g;
  const int t = slot / P;
  const int k = vals[(size_t)slot * n_nodes + node];
  const int lo = it.ev_lo[t], hi = it.ev_hi[t];
  long long bad_rows = 0, min_idx = 1ll << 40, max_idx = -(1ll << 40), flags = 0;
  if (k < 0 || k >= nd.n_cand) flags |= 1;
  if (lo < 0 || hi > n_ev || lo > hi) flags |= 2;
  if (!(flags & 3))
    for (int e = lo; e < hi; ++e) {
      const int row = it.ev_rows[e];
      if (row < 0 || row >= n_rows) {
        ++bad_rows;
        continue;
      }
      const int32_t* evctx = it.ev_ctx ? it.ev_ctx + (size_t)e * PCLEAN_MAX_CTX : nullptr;
      long long idx = 0;
      for (int d = 0; d < g.n_dims; ++d) {
        long long v = 0;
        if (g.src_kind[d] == PCLEAN_GSRC_CAND) {
          if (!g.src_ptr[d]) flags |= 4; else v = g.src_ptr[d][k];
        } else if (g.src_kind[d] == PCLEAN_GSRC_OBS) {
          if (!g.src_ptr[d]) flags |= 8; else v = g.src_ptr[d][row];
        } else if (g.src_kind[d] == PCLEAN_GSRC_EVCTX) {
          if (!evctx) flags |= 16; else v = evctx[g.src_slot[d]];
        } else {
          flags |= 32;
        }
        idx += (long long)g.stride[d] * v;
      }
      if (g.t_kind == PCLEAN_GSRC_EVCTX && evctx) {
        const int u = evctx[g.t_src];
        if (u < 0 || u > 3) flags |= 64;
      }
      min_idx = idx < min_idx ? idx : min_idx;
      max_idx = idx > max_idx ? idx : max_idx;
      if (idx < 0 || idx >= n_mean) flags |= 128;
      else {
        const double xv = g.x[row], mv = g.mu[idx];
        if (xv == xv && !(mv == mv)) flags |= 256;
        if (mode >= 1) {  // the real thing
          ItemView v{};
          v.item = t;
          v.ctxv = (mode == 1 || !it.ctx) ? nullptr : it.ctx;  // (mode 1: a compile-time null, mode 2: a run-time one)
          if (mode == 2) v.ctxv = it.ctx ? it.ctx + (size_t)t * PCLEAN_MAX_CTX : nullptr;
          v.excl = -1;
          const double gt = gauss_term(nd, v, k, row, evctx);
          if (!(gt == gt)) flags |= 512;
        }
      }
    }
  long long* o = out + (size_t)slot * 8;
  o[0] = k; o[1] = nd.n_cand; o[2] = lo; o[3] = hi; o[4] = bad_rows; o[5] = min_idx; o[6] = max_idx; o[7] = flags;
}
int pclean_debug_gauss_ev_probe(pclean_ctx* ctx, int n_items, int P, int n_nodes, const NodeDev* nds, int node, const ItemsDev& it,
                                const int32_t* vals, int n_mean, int n_ev) {
  const int n_slots = n_items * P;
  long long* d = nullptr;
  if (hipMalloc((void**)&d, (size_t)n_slots * 64) != hipSuccess) return PCLEAN_OK;
  std::vector<long long> h((size_t)n_slots * 8);
  for (int mode = 0; mode < 3; mode += 2) {
    hipLaunchKernelGGL(gauss_ev_probe_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, ctx->stream, n_slots, P, n_nodes, nds, node, it,
                       vals, ctx->n_rows, n_mean, n_ev, d, mode);
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    fprintf(stderr, "[gauss probe] node %d mode %d (0: range checks, 2: gauss_term itself): %s\n", node, mode, hipGetErrorString(e));
    fflush(stderr);
  }
  (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  int shown = 0;
  long long any = 0;
  for (int sl = 0; sl < n_slots; ++sl) {
    const long long* o = &h[(size_t)sl * 8];
    any |= o[7];
    if ((o[7] || o[4]) && shown < 12) {
      fprintf(stderr, "[gauss probe] node %d slot %d: k %lld of %lld, evidence [%lld, %lld), bad rows %lld, mean index %lld..%lld of %d, flags %llx\n",
              node, sl, o[0], o[1], o[2], o[3], o[4], o[5], o[6], n_mean, (unsigned long long)o[7]);
      ++shown;
    }
  }
  fprintf(stderr, "[gauss probe] node %d: %d slots, flags seen %llx (1 k, 2 evidence range, 4/8 null source, 16 null evidence ctx, 32 kind, 64 unit, 128 mean index)\n",
          node, n_slots, (unsigned long long)any);
  return PCLEAN_OK;
}

int pclean_launch_prior_terms_ev(pclean_ctx* ctx, int n_items, int P, int n_nodes, const NodeDev* nds, const AggDev* const* aggs,
                                 const int32_t* n_children, const int32_t* child_begin, const int32_t* children, int n_roots,
                                 const int32_t* roots, const ItemsDev& it, const int32_t* vals, double* w) {
  if (n_items <= 0) return PCLEAN_OK;
  if (n_nodes > PRIOR_MAX_NODES) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "prior proposals: more than %d plan nodes", PRIOR_MAX_NODES);
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, ctx->prob_same.p, ctx->prob_diff.p, ctx->logn.p};
  hipLaunchKernelGGL(prior_terms_ev_kernel, dim3((n_items * P + 255) / 256), dim3(256), 0, ctx->stream, n_items, P, n_nodes,
                     nds, aggs, n_children, child_begin, children, n_roots, roots, dn, it, vals, w);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// rows of the split launch's score scratch: every item of a direct launch within 256 MB, and of an indirect one (its
// device-side list is usually a few per cent of the items) at most 64 — later workgroups compute their scores themselves
static int pclean_enum_split_slots(const ItemsDev& it, int nc) {
  const size_t fit = std::max<size_t>(((size_t)32 << 20) / (size_t)std::max(nc, 1), 1);
  return (int)std::min<size_t>(std::min<size_t>((size_t)it.n, fit), it.sel ? 64 : (size_t)it.n);
}
// doubles of scratch with which pclean_launch_enum computes the scores of this launch one candidate per thread first
// (enum_scores_kernel); 0: the launch would not use them (many items, short lists, no evidence sets, scores beyond LDS)
size_t pclean_enum_split_scores(const NodeDev& nd, const ItemsDev& it) {
  static const bool off = getenv("PCLEAN_NO_ENUM_SPLIT") != nullptr;
  const int nc = nd.n_cand + (nd.kind == PCLEAN_NODE_FK ? 1 : 0);
  const size_t lds = (size_t)((nc + 1) & ~1) * 8 + (16 + 64) * 8;
  const bool few = it.n <= 1024 || (it.sel && it.n <= 16384);
  (void)lds;
  // (without evidence sets a candidate is one short chain of gathers: worth the extra launch for a handful of long lists only —
  // the nested slots of a new-row branch, seven groups x 14 000 candidates per 1M-row sweep)
  if (off || !few || it.grp_off || nc < 2048) return 0;
  if (!it.ev_lo && (it.n > 64 || nc < 4096)) return 0;
  return (size_t)pclean_enum_split_slots(it, nc) * nc;
}

int pclean_launch_enum(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const ChildrenDev& ch, uint64_t seed,
                       uint32_t sweep, uint32_t site, int n_draws, double* lse_out, double* scores_out,
                       int32_t* draws_out, double* scores_tmp) {
  if (it.n <= 0) return PCLEAN_OK;
  const int nc = nd.n_cand + (nd.kind == PCLEAN_NODE_FK ? 1 : 0);
  const size_t lds = (size_t)((nc + 1) & ~1) * 8 + (16 + 64) * 8;
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0, ctx->prob_same.p, ctx->prob_diff.p, ctx->logn.p};
  // a launch may not exceed 2^32 threads: chunk the items (grid = chunk, 256 lanes each)
  const int kMaxBlocks = 4 * 1024 * 1024;
  // The LDS-resident kernel keeps every score in LDS: beyond 80 KB only one workgroup fits on a CU and the
  // recompute-per-pass kernel (no LDS, 8 workgroups per CU) wins — unless the launch is grouped (scores shared by
  // the member items), which only the LDS kernel supports, or draws many times per item (one more pass per draw).
  static const size_t big_from = getenv("PCLEAN_BIG_FROM") ? (size_t)atol(getenv("PCLEAN_BIG_FROM")) : (size_t)80 * 1024;
  // (a launch of at most 1024 items — the sub-batches of the latent sweeps — cannot fill the chip with workgroups whatever
  // their LDS footprint: the LDS-resident kernel's single pass beats three recomputing passes there, measured 0.22 vs
  // 0.40 ms per launch on the Hospital sub-batches)
  static const bool big_few = getenv("PCLEAN_BIG_FEW_ITEMS") != nullptr;
  // (an indirect launch runs as many workgroups as its device-side list holds — the items a scan could not settle, few)
  // (... and a launch whose score arrays leave room for fewer than eight workgroups of 256 threads on a CU — lists from ~2 500
  // candidates on: the kernels wait for gathers, and 1024 threads per item keep twice to four times the waves in flight)
  static const bool no_wide = getenv("PCLEAN_NO_WIDE_ENUM") != nullptr;
  const bool few = it.n <= 1024 || (it.sel && it.n <= 16384) || (!no_wide && lds > 20 * 1024 && it.n <= 65536);
  if (it.sel && it.grp_off) return pclean_fail(ctx, PCLEAN_ERR_ARG, "indirect launches are not grouped");
  if (lds > 160 * 1024 || (lds > big_from && !it.grp_off && !scores_out && n_draws <= 1 && (!few || big_few))) {
    if (few) {  // too few workgroups to fill the chip: more threads per item
      const double* scores_in = nullptr;
      int slot_cap = 0;
      if (scores_tmp && !it.grp_off && !scores_out) {  // ... and the scores first, one candidate per thread (enum_scores_kernel)
        slot_cap = pclean_enum_split_slots(it, nc);
        if (it.ev_lo)
          hipLaunchKernelGGL((enum_scores_kernel<true>), dim3((nc + 255) / 256, std::min(slot_cap, 1024)), dim3(256), 0, ctx->stream, nd, dn, it,
                           ch, scores_tmp, slot_cap);
        else
          hipLaunchKernelGGL((enum_scores_kernel<false>), dim3((nc + 255) / 256, std::min(slot_cap, 1024)), dim3(256), 0, ctx->stream, nd, dn, it,
                           ch, scores_tmp, slot_cap);
        scores_in = scores_tmp;
      }
      if (it.ev_lo)
        hipLaunchKernelGGL((enum_node_big_kernel<1024, true>), dim3(it.n), dim3(1024), 0, ctx->stream, nd, dn, it, ch, seed, sweep, site,
                         n_draws, 0, lse_out, scores_out, draws_out, scores_in, slot_cap);
      else
        hipLaunchKernelGGL((enum_node_big_kernel<1024, false>), dim3(it.n), dim3(1024), 0, ctx->stream, nd, dn, it, ch, seed, sweep, site,
                         n_draws, 0, lse_out, scores_out, draws_out, scores_in, slot_cap);
    } else {
      for (int base = 0; base < it.n; base += kMaxBlocks)
        if (it.ev_lo)
          hipLaunchKernelGGL((enum_node_big_kernel<256, true>), dim3(std::min(kMaxBlocks, it.n - base)), dim3(256), 0, ctx->stream, nd,
                           dn, it, ch, seed, sweep, site, n_draws, base, lse_out, scores_out, draws_out, (const double*)nullptr, 0);
        else
          hipLaunchKernelGGL((enum_node_big_kernel<256, false>), dim3(std::min(kMaxBlocks, it.n - base)), dim3(256), 0, ctx->stream, nd,
                           dn, it, ch, seed, sweep, site, n_draws, base, lse_out, scores_out, draws_out, (const double*)nullptr, 0);
    }
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      HIPCHK(ctx, hipFuncSetAttribute((const void*)enum_node_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(ctx, hipFuncSetAttribute((const void*)enum_node_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(ctx, hipFuncSetAttribute((const void*)enum_node_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(ctx, hipFuncSetAttribute((const void*)enum_node_kernel<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    if (few) {  // (groups or items) too few workgroups to fill the chip: more threads per item
      // ... and, with scratch for them (pclean_enum_split_scores), the scores first, one candidate per thread
      const double* scores_in = nullptr;
      int slot_cap = 0;
      if (scores_tmp && !it.grp_off && !scores_out) {
        slot_cap = pclean_enum_split_slots(it, nc);
        if (it.ev_lo)
          hipLaunchKernelGGL((enum_scores_kernel<true>), dim3((nc + 255) / 256, std::min(slot_cap, 1024)), dim3(256), 0, ctx->stream, nd, dn, it,
                           ch, scores_tmp, slot_cap);
        else
          hipLaunchKernelGGL((enum_scores_kernel<false>), dim3((nc + 255) / 256, std::min(slot_cap, 1024)), dim3(256), 0, ctx->stream, nd, dn, it,
                           ch, scores_tmp, slot_cap);
        scores_in = scores_tmp;
      }
      if (it.ev_lo)
        hipLaunchKernelGGL((enum_node_kernel<1024, true>), dim3(it.n), dim3(1024), lds, ctx->stream, nd, dn, it, ch, seed, sweep, site,
                           n_draws, 0, lse_out, scores_out, draws_out, scores_in, slot_cap);
      else
        hipLaunchKernelGGL((enum_node_kernel<1024, false>), dim3(it.n), dim3(1024), lds, ctx->stream, nd, dn, it, ch, seed, sweep, site,
                           n_draws, 0, lse_out, scores_out, draws_out, scores_in, slot_cap);
    } else {
      for (int base = 0; base < it.n; base += kMaxBlocks)
        if (it.ev_lo)
          hipLaunchKernelGGL((enum_node_kernel<256, true>), dim3(std::min(kMaxBlocks, it.n - base)), dim3(256), lds, ctx->stream, nd, dn,
                             it, ch, seed, sweep, site, n_draws, base, lse_out, scores_out, draws_out, (const double*)nullptr, 0);
        else
          hipLaunchKernelGGL((enum_node_kernel<256, false>), dim3(std::min(kMaxBlocks, it.n - base)), dim3(256), lds, ctx->stream, nd, dn,
                             it, ch, seed, sweep, site, n_draws, base, lse_out, scores_out, draws_out, (const double*)nullptr, 0);
    }
  }
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
