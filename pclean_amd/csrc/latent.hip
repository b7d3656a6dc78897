// Latent-class sweeps: a latent row re-proposes its attributes given every observed row that refers to it
// (ExternalLikelihoodNodes, proposal_compiler.jl:306-350): aggregated evidence, pclean_sweep_latent (option lists on
// side streams, reference slots, prior proposals), pclean_score_node_ev (one node against evidence sets, for parity tests).
#include "sweep_internal.h"

// ---- aggregated evidence of latent-class sweeps -------------------------------------------------------------
// (contract in enum_kernels.hip: candidate_score_ev)
__global__ void item_of_pos_kernel(int n_ev, int n_items, const int32_t* __restrict__ off, int32_t* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ev) return;
  int lo = 0, hi = n_items - 1;  // largest t with off[t] <= e
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= e)
      lo = mid;
    else
      hi = mid - 1;
  }
  out[e] = lo;
}
__global__ void agg_key_kernel(int n_ev, const int32_t* __restrict__ item_of_pos, const int32_t* __restrict__ ev_rows,
                               const int32_t* __restrict__ ev_ctx, int ctx_slot, const int32_t* __restrict__ obs_col,
                               uint64_t* __restrict__ key) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ev) return;
  const uint64_t o1 = (uint64_t)(uint32_t)(obs_col[ev_rows[e]] + 1) & 0xffffffull;
  const uint64_t c = ctx_slot >= 0 ? ((uint64_t)(uint32_t)ev_ctx[(size_t)e * PCLEAN_MAX_CTX + ctx_slot] & 0xffffull) : 0ull;
  key[e] = ((uint64_t)(uint32_t)item_of_pos[e] << 40) | (c << 24) | o1;
}
__global__ void agg_off_kernel(int n_items, const uint64_t* __restrict__ uniq, const int32_t* __restrict__ n_runs,
                               int32_t* __restrict__ off) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_items) return;
  const uint64_t want = (uint64_t)(uint32_t)t << 40;
  int lo = 0, hi = *n_runs;  // first run with key >= want
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (uniq[mid] < want)
      lo = mid + 1;
    else
      hi = mid;
  }
  off[t] = lo;
}
struct AggPack {
  AggDev a[PCLEAN_MAX_TERMS];
};
// The same aggregation with one workgroup per (original item, term) when no item has more than AGG_LDS_CAP evidence
// rows — the sub-batches of a large latent class (a few hundred rows with ~100 referring rows each): the keys of the
// item's rows are sorted in LDS (bitonic), run-length encoded and written at the item's own offset of the evidence
// list.  Same runs in the same order as the global sort + run-length encoding below (which costs ~18 launches per
// term); the scores that walk them are unchanged.
#define AGG_LDS_CAP 2048
struct AggTermArgs {
  const int32_t* obs_col[PCLEAN_MAX_TERMS];
  int32_t ctx_slot[PCLEAN_MAX_TERMS];
  uint64_t* uniq[PCLEAN_MAX_TERMS];
  int32_t* cnt[PCLEAN_MAX_TERMS];
  int32_t* end[PCLEAN_MAX_TERMS];
};
// every term of every node of a latent plan at once (ensure_agg_all)
#define AGG_ALL_MAX 48
struct AggAllArgs {
  const int32_t* obs_col[AGG_ALL_MAX];
  int32_t ctx_slot[AGG_ALL_MAX];
  uint64_t* uniq[AGG_ALL_MAX];
  int32_t* cnt[AGG_ALL_MAX];
  int32_t* end[AGG_ALL_MAX];
};
struct AggAllPack {
  AggDev a[AGG_ALL_MAX];
  int32_t dst[AGG_ALL_MAX];  // where entry i goes in the per-node arrays: node * PCLEAN_MAX_TERMS + term
};
template <typename Args>
__global__ __launch_bounds__(256) void agg_item_kernel(int n_items, const int32_t* __restrict__ ev_off,
                                                       const int32_t* __restrict__ ev_rows, const int32_t* __restrict__ ev_ctx,
                                                       Args a) {
  __shared__ uint64_t s_key[AGG_LDS_CAP];
  __shared__ int32_t s_run[AGG_LDS_CAP];  // run id of sorted position i, then the run lengths
  __shared__ int s_w[4];
  const int t = blockIdx.x, ti = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = ev_off[t], L = ev_off[t + 1] - lo;
  if (L <= 0) {
    if (tid == 0) a.end[ti][t] = lo;
    return;
  }
  int np2 = 1;
  while (np2 < L) np2 <<= 1;
  const int32_t* oc = a.obs_col[ti];
  const int cs = a.ctx_slot[ti];
  for (int i = tid; i < np2; i += 256) {
    uint64_t key = ~0ull;  // padding sorts last
    if (i < L) {
      const int e = lo + i;
      const uint64_t o1 = (uint64_t)(uint32_t)(oc[ev_rows[e]] + 1) & 0xffffffull;
      const uint64_t c = cs >= 0 ? ((uint64_t)(uint32_t)ev_ctx[(size_t)e * PCLEAN_MAX_CTX + cs] & 0xffffull) : 0ull;
      key = ((uint64_t)(uint32_t)t << 40) | (c << 24) | o1;
    }
    s_key[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = s_key[i], y = s_key[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            s_key[i] = y;
            s_key[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  // run ids: inclusive count of heads over the sorted keys, in chunks of 256 positions
  int base = 0;
  for (int i0 = 0; i0 < L; i0 += 256) {
    const int i = i0 + tid;
    const int head = (i < L && (i == 0 || s_key[i] != s_key[i - 1])) ? 1 : 0;
    int incl = head;
    for (int sh = 1; sh < 64; sh <<= 1) {
      const int x = __shfl_up(incl, sh, 64);
      if (lane >= sh) incl += x;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int before = base, total = 0;
    for (int w = 0; w < 4; ++w) {
      if (w < wave) before += s_w[w];
      total += s_w[w];
    }
    if (i < L) s_run[i] = before + incl - 1;
    base += total;
    __syncthreads();
  }
  const int n_runs = base;
  // heads write their key; lengths = distance to the next head
  for (int i = tid; i < L; i += 256) {
    if (i == 0 || s_key[i] != s_key[i - 1]) {
      const int r = s_run[i];
      int j = i + 1;
      while (j < L && s_key[j] == s_key[i]) ++j;
      a.uniq[ti][lo + r] = s_key[i];
      a.cnt[ti][lo + r] = j - i;
    }
  }
  if (tid == 0) a.end[ti][t] = lo + n_runs;
}
__global__ void write_agg_kernel(AggPack p, int n, AggDev* dst) {
  const int i = threadIdx.x;
  if (i < n) dst[i] = p.a[i];
}
__global__ void write_agg_all_kernel(AggAllPack p, int n, AggDev* dst) {
  const int i = threadIdx.x;
  if (i < n) dst[p.dst[i]] = p.a[i];
}

// A per-evidence-row ctx value occupies 16 bits of an aggregation key (agg_key_kernel / agg_item_kernel): its domain
// (the ctx side of the term's fn table, or the error-probability table of a MaybeSwap term) must stay below 2^16 or runs
// of different values would alias.
static int agg_ctx_fits(pclean_ctx* ctx, const pclean_term& tm, int ctx_slot) {
  if (ctx_slot < 0) return PCLEAN_OK;
  int64_t dom = 0;
  if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP)
    dom = ctx->n_prob;
  else if (tm.fn_table >= 0 && tm.fn_table < PCLEAN_MAX_TABLES && ctx->fn[tm.fn_table].valid)
    dom = tm.ctx_mode == 2 ? ctx->fn[tm.fn_table].n_b : ctx->fn[tm.fn_table].n_a;
  if (dom >= (1 << 16))
    return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "evidence aggregation: a per-evidence-row ctx domain of %lld values does not "
                                                 "fit the 16 key bits", (long long)dom);
  return PCLEAN_OK;
}

// The aggregated evidence of EVERY node of the plan in one launch, at the start of a pclean_sweep_latent call whose items fit
// the LDS aggregation (the sub-batches of a large latent class: a dozen nodes each, one ~25 us launch + one pointer-table
// write per node was 12 ms of a Hospital class sweep).  Whatever this leaves out (a plan with more terms than AGG_ALL_MAX, a
// term without its per-evidence-row ctx, ...) is built, or refused, by ensure_agg when a node asks for it.
static int ensure_agg_all(pclean_ctx* ctx, int block_id, const int32_t* ev_rows, const int32_t* ev_ctx) {
  static const bool off = getenv("PCLEAN_NO_AGG_ALL") != nullptr;
  SweepState* s = st(ctx);
  const Block& b = ctx->block[block_id];
  const int n_ev = s->lat_ev, n_items = s->lat_items, n_nodes = (int)b.nodes.size();
  if (off || n_ev <= 0 || n_items <= 0 || n_items >= (1 << 24) || s->lat_max_ev > AGG_LDS_CAP || ctx->no_item_agg) return PCLEAN_OK;
  int total = 0;
  for (int node = 0; node < n_nodes; ++node) {
    const pclean_node& n = b.nodes[node];
    if (n.n_terms > PCLEAN_MAX_TERMS) return PCLEAN_OK;
    for (int ti = 0; ti < n.n_terms; ++ti) {
      const pclean_term& tm = b.terms[n.term_begin + ti];
      const PairTable& pt = ctx->pair[tm.pair_table];
      const int ctx_slot = (tm.ctx_slot >= 0 && tm.ctx_mode != 0) ? tm.ctx_slot : -1;
      if (tm.obs_col < 0 || tm.obs_col >= ctx->n_cols || (pt.valid && pt.n_obs + 1 >= (1 << 24)) || (ctx_slot >= 0 && !ev_ctx)) return PCLEAN_OK;
      if (ctx_slot >= 0) {  // (16 key bits per ctx value: agg_ctx_fits, which fails the call — left to ensure_agg)
        int64_t dom = 0;
        if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP)
          dom = ctx->n_prob;
        else if (tm.fn_table >= 0 && tm.fn_table < PCLEAN_MAX_TABLES && ctx->fn[tm.fn_table].valid)
          dom = tm.ctx_mode == 2 ? ctx->fn[tm.fn_table].n_b : ctx->fn[tm.fn_table].n_a;
        if (dom >= (1 << 16)) return PCLEAN_OK;
      }
    }
    total += n.n_terms;
  }
  if (total <= 0 || total > AGG_ALL_MAX) return PCLEAN_OK;
  ProfScope ps(ctx, "evidence_aggregation");
  AggDev* dst = (AggDev*)scratch<unsigned char>(ctx, sizeof(AggDev) * PCLEAN_MAX_TERMS * (size_t)n_nodes);
  uint64_t* uniq = scratch<uint64_t>(ctx, (size_t)n_ev * total);
  int32_t* cnt = scratch<int32_t>(ctx, (size_t)n_ev * total);
  int32_t* end = scratch<int32_t>(ctx, (size_t)n_items * total);
  if (!dst || !uniq || !cnt || !end) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  AggAllArgs at{};
  AggAllPack pack{};
  int k = 0;
  for (int node = 0; node < n_nodes; ++node) {
    const pclean_node& n = b.nodes[node];
    for (int ti = 0; ti < n.n_terms; ++ti, ++k) {
      const pclean_term& tm = b.terms[n.term_begin + ti];
      at.obs_col[k] = ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows;
      at.ctx_slot[k] = (tm.ctx_slot >= 0 && tm.ctx_mode != 0) ? tm.ctx_slot : -1;
      at.uniq[k] = uniq + (size_t)n_ev * k;
      at.cnt[k] = cnt + (size_t)n_ev * k;
      at.end[k] = end + (size_t)n_items * k;
      pack.a[k] = AggDev{at.uniq[k], at.cnt[k], s->lat_off, at.end[k]};
      pack.dst[k] = node * PCLEAN_MAX_TERMS + ti;
    }
  }
  hipLaunchKernelGGL(agg_item_kernel<AggAllArgs>, dim3(n_items, total), dim3(256), 0, ctx->stream, n_items, s->lat_off, ev_rows,
                     ev_ctx, at);
  hipLaunchKernelGGL(write_agg_all_kernel, dim3(1), dim3(64), 0, ctx->stream, pack, total, dst);
  HIPCHK(ctx, hipGetLastError());
  for (int node = 0; node < n_nodes; ++node)
    if (b.nodes[node].n_terms > 0) s->lat_agg[node] = dst + (size_t)node * PCLEAN_MAX_TERMS;
  return PCLEAN_OK;
}

// Aggregated evidence of every term of node `node_id` over the original items of the running
// pclean_sweep_latent call; built once per (call, node).
int ensure_agg(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const AggDev** out) {
  SweepState* s = st(ctx);
  auto itc = s->lat_agg.find(node_id);
  if (itc != s->lat_agg.end()) {
    *out = itc->second;
    return PCLEAN_OK;
  }
  ProfScope ps(ctx, "evidence_aggregation");
  const Block& b = ctx->block[block_id];
  const pclean_node& n = b.nodes[node_id];
  const int n_ev = s->lat_ev, n_items = s->lat_items;
  if (n_items >= (1 << 24)) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "too many latent rows in one latent sweep");
  AggPack pack{};
  AggDev* dst = (AggDev*)scratch<unsigned char>(ctx, sizeof(AggDev) * PCLEAN_MAX_TERMS);
  if (!dst) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  if (n_ev > 0 && s->lat_max_ev <= AGG_LDS_CAP && n.n_terms > 0 && n.n_terms <= PCLEAN_MAX_TERMS && !ctx->no_item_agg) {
    AggTermArgs at{};
    for (int ti = 0; ti < n.n_terms; ++ti) {
      const pclean_term& tm = b.terms[n.term_begin + ti];
      const PairTable& pt = ctx->pair[tm.pair_table];
      if (tm.obs_col < 0 || tm.obs_col >= ctx->n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term column out of range");
      if (pt.valid && pt.n_obs + 1 >= (1 << 24)) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "observed domain too large for the evidence keys");
      const int ctx_slot = (tm.ctx_slot >= 0 && tm.ctx_mode != 0) ? tm.ctx_slot : -1;
      if (ctx_slot >= 0 && !il.ev_ctx) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term %d needs per-evidence-row ctx", n.term_begin + ti);
      { const int rck = agg_ctx_fits(ctx, tm, ctx_slot); if (rck) return rck; }  // (16 key bits per ctx value)
      at.obs_col[ti] = ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows;
      at.ctx_slot[ti] = ctx_slot;
      at.uniq[ti] = scratch<uint64_t>(ctx, (size_t)n_ev);
      at.cnt[ti] = scratch<int32_t>(ctx, (size_t)n_ev);
      at.end[ti] = scratch<int32_t>(ctx, (size_t)n_items);
      if (!at.uniq[ti] || !at.cnt[ti] || !at.end[ti]) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      pack.a[ti] = AggDev{at.uniq[ti], at.cnt[ti], s->lat_off, at.end[ti]};
    }
    hipLaunchKernelGGL(agg_item_kernel<AggTermArgs>, dim3(n_items, n.n_terms), dim3(256), 0, ctx->stream, n_items, s->lat_off,
                       il.ev_rows, il.ev_ctx, at);
    hipLaunchKernelGGL(write_agg_kernel, dim3(1), dim3(64), 0, ctx->stream, pack, n.n_terms, dst);
    HIPCHK(ctx, hipGetLastError());
    s->lat_agg[node_id] = dst;
    *out = dst;
    return PCLEAN_OK;
  }
  for (int ti = 0; ti < n.n_terms; ++ti) {
    const pclean_term& tm = b.terms[n.term_begin + ti];
    const PairTable& pt = ctx->pair[tm.pair_table];
    if (tm.obs_col < 0 || tm.obs_col >= ctx->n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term column out of range");
    if (pt.valid && pt.n_obs + 1 >= (1 << 24)) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "observed domain too large for the evidence keys");
    const int ctx_slot = (tm.ctx_slot >= 0 && tm.ctx_mode != 0) ? tm.ctx_slot : -1;
    if (ctx_slot >= 0 && !il.ev_ctx) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term %d needs per-evidence-row ctx", n.term_begin + ti);
      { const int rck = agg_ctx_fits(ctx, tm, ctx_slot); if (rck) return rck; }  // (16 key bits per ctx value)
    const size_t ne = (size_t)std::max(n_ev, 1);
    uint64_t* key = scratch<uint64_t>(ctx, ne);
    uint64_t* key_s = scratch<uint64_t>(ctx, ne);
    uint64_t* uniq = scratch<uint64_t>(ctx, ne);
    int32_t* cnt = scratch<int32_t>(ctx, ne);
    int32_t* n_runs = scratch<int32_t>(ctx, 4);
    int32_t* off = scratch<int32_t>(ctx, (size_t)n_items + 2);
    if (!key || !key_s || !uniq || !cnt || !n_runs || !off) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    HIPCHK(ctx, hipMemsetAsync(n_runs, 0, sizeof(int32_t), ctx->stream));
    if (n_ev > 0) {
      hipLaunchKernelGGL(agg_key_kernel, grid1(n_ev), dim3(256), 0, ctx->stream, n_ev, s->lat_item_of_pos, il.ev_rows,
                         il.ev_ctx, ctx_slot, ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows, key);
      size_t tmp_sort = 0, tmp_rle = 0;
      HIPCHK(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_sort, key, key_s, n_ev, 0, 64, ctx->stream));
      HIPCHK(ctx, hipcub::DeviceRunLengthEncode::Encode(nullptr, tmp_rle, key_s, uniq, cnt, n_runs, n_ev, ctx->stream));
      unsigned char* tmp = scratch<unsigned char>(ctx, std::max(tmp_sort, tmp_rle));
      if (!tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      HIPCHK(ctx, hipcub::DeviceRadixSort::SortKeys(tmp, tmp_sort, key, key_s, n_ev, 0, 64, ctx->stream));
      HIPCHK(ctx, hipcub::DeviceRunLengthEncode::Encode(tmp, tmp_rle, key_s, uniq, cnt, n_runs, n_ev, ctx->stream));
    }
    hipLaunchKernelGGL(agg_off_kernel, grid1((size_t)n_items + 1), dim3(256), 0, ctx->stream, n_items, uniq, n_runs, off);
    pack.a[ti] = AggDev{uniq, cnt, off, nullptr};
  }
  hipLaunchKernelGGL(write_agg_kernel, dim3(1), dim3(64), 0, ctx->stream, pack, n.n_terms, dst);
  HIPCHK(ctx, hipGetLastError());
  s->lat_agg[node_id] = dst;
  *out = dst;
  return PCLEAN_OK;
}

// particle choice of a latent row: every particle has the same weight (all sub-plans enumerated)
__global__ void latent_choice_kernel(int n, int P, int use_mh, const int32_t* keys, uint64_t seed, uint32_t sweep,
                                     uint32_t block_id, int32_t* chosen) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t rr = (uint32_t)keys[i];
  const uint32_t pid = 0x1000u + block_id;
  int c;
  if (use_mh && P >= 2) {  // row_inference.jl:161-162 with w1 == w0
    const double ratio = 0.5 / (1e-10 + 0.5);
    c = pclean_u01(pclean_rand64(seed, rr, PCLEAN_SITE_MH, pid, sweep)) < ratio ? 1 : 0;
  } else {
    const uint64_t U = (uint64_t)P << PCLEAN_FIX_BITS;
    c = (int)(pclean_mulhi64(pclean_rand64(seed, rr, PCLEAN_SITE_FINAL, pid, sweep), U) >> PCLEAN_FIX_BITS);
  }
  chosen[i] = c;
}
__global__ void mark_positive_kernel(int n, const int32_t* v, int32_t* flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = v[i] > 0 ? PCLEAN_CHOICE_NEW : 0;
}
__global__ void latent_items_kernel(int n, const int32_t* list, const int32_t* keys, const int32_t* ev_off,
                                    const int32_t* chosen, int32_t* rng, int32_t* ev_lo, int32_t* ev_hi,
                                    int32_t* particle, int32_t* origin) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int t = list[j];
  rng[j] = keys[t];
  ev_lo[j] = ev_off[t];
  ev_hi[j] = ev_off[t + 1];
  particle[j] = chosen[t];
  origin[j] = t;
}

// ---- prior proposals for a latent class (use_dd_proposals = false) -----------------------------------------------
__global__ void retain_first_kernel(int n_items, int P, const int32_t* __restrict__ cur, int32_t* __restrict__ draws) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_items) draws[(size_t)t * P] = cur[t];  // particle 0 keeps the row's current value
}
__global__ void set_node_col_kernel(int n, const int32_t* __restrict__ src, int n_nodes, int node, int32_t* __restrict__ vals) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) vals[(size_t)j * n_nodes + node] = src[j];
}
__global__ void latent_prior_items_kernel(int n, int P, int n_items, const int32_t* __restrict__ list,
                                          const int32_t* __restrict__ keys, const int32_t* __restrict__ cur,
                                          int32_t* __restrict__ rng, int32_t* __restrict__ particle,
                                          int32_t* __restrict__ origin, int32_t* __restrict__ excl,
                                          int32_t* __restrict__ ev_lo, int32_t* __restrict__ ev_hi,
                                          const int32_t* __restrict__ off) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int idx = list[j], t = idx / P;
  rng[j] = keys[t];
  particle[j] = idx - t * P;
  origin[j] = idx;
  excl[j] = cur[t];
  ev_lo[j] = off[t];
  ev_hi[j] = off[t + 1];
}
// final choice among the P particles of every latent row (row_inference.jl:158-165), weights row-major [n_items][P]
template <int PMAX>
__global__ void latent_prior_choice_kernel(int n_items, int P, int use_mh, const double* __restrict__ w,
                                           const int32_t* __restrict__ keys, uint64_t seed, uint32_t sweep, uint32_t block_id,
                                           int32_t* __restrict__ chosen) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_items) return;
  FixW<PMAX> f;
  fix_weights<PMAX>(w + (size_t)t * P, (size_t)1, P, f);
  const uint32_t rr = (uint32_t)keys[t], pid = 0x1000u + block_id;
  int c;
  if (use_mh && P >= 2) {
    const double Ud = (double)f.U;
    const double w0 = (double)f.u[0] / Ud, w1 = (double)f.u[PMAX > 1 ? 1 : 0] / Ud;
    double ratio = w1 / (1e-10 + w0);
    if (ratio > 1.0) ratio = 1.0;
    c = (f.U != 0 && pclean_u01(pclean_rand64(seed, rr, PCLEAN_SITE_MH, pid, sweep)) < ratio) ? 1 : 0;
  } else {
    c = fix_pick<PMAX>(f, P, pclean_rand64(seed, rr, PCLEAN_SITE_FINAL, pid, sweep));
  }
  chosen[t] = c;
}
__global__ void gather_chosen_vals_kernel(int n_items, int P, int n_nodes, const int32_t* __restrict__ chosen,
                                          const int32_t* __restrict__ pv, int32_t* __restrict__ vals) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (size_t)n_items * n_nodes) return;
  const int t = (int)(q / n_nodes), k = (int)(q - (size_t)t * n_nodes);
  const int c = chosen[t];
  vals[q] = c > 0 ? pv[((size_t)t * P + c) * n_nodes + k] : -2;
}

// Side streams of pclean_sweep_latent (sweep_state.h); PCLEAN_LATENT_STREAMS=0 keeps everything on the library's stream.
static int side_streams(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  if (s->n_side >= 0) return s->n_side;
  const char* e = getenv("PCLEAN_LATENT_STREAMS");
  int want = e ? atoi(e) : 6;
  want = std::max(0, std::min(want, (int)SweepState::MAX_SIDE));
  s->n_side = 0;
  if (want > 0 && hipEventCreateWithFlags(&s->side_fork, hipEventDisableTiming) != hipSuccess) return 0;
  for (int k = 0; k < want; ++k) {
    if (hipStreamCreateWithFlags(&s->side[k], hipStreamNonBlocking) != hipSuccess) break;
    if (hipEventCreateWithFlags(&s->side_join[k], hipEventDisableTiming) != hipSuccess) break;
    if (hipEventCreateWithFlags(&s->side_mid[k], hipEventDisableTiming) != hipSuccess) break;
    s->n_side = k + 1;
  }
  return s->n_side;
}
// Work of one call spread over the side streams: fork() after the inputs are queued on the library's stream, use(i) to
// issue the i-th independent piece, join() before anything on the library's stream reads the results.  An early
// return (error) waits for the side streams on the host: the scratch pool they use is rewound by the next call.
struct SideFork {
  pclean_ctx* ctx;
  SweepState* s;
  hipStream_t main;
  bool forked = false, used[SweepState::MAX_SIDE] = {};
  explicit SideFork(pclean_ctx* c) : ctx(c), s(st(c)), main(c->stream) {}
  int fork() {  // what is queued on the library's stream so far is what the side streams wait for
    if (side_streams(ctx) <= 0 || forked) return PCLEAN_OK;
    HIPCHK(ctx, hipEventRecord(s->side_fork, main));
    forked = true;
    return PCLEAN_OK;
  }
  int use(int i) {
    const int K = side_streams(ctx);
    if (K <= 0) return PCLEAN_OK;
    if (!forked) {
      const int rc = fork();
      if (rc) return rc;
    }
    const int k = i % K;
    if (!used[k]) {
      HIPCHK(ctx, hipStreamWaitEvent(s->side[k], s->side_fork, 0));
      used[k] = true;
    }
    ctx->stream = s->side[k];
    return PCLEAN_OK;
  }
  void back() { ctx->stream = main; }
  int mark() {  // the library's stream waits for what the current side stream holds so far (not for what follows on it)
    if (ctx->stream == main) return PCLEAN_OK;
    for (int k = 0; k < SweepState::MAX_SIDE; ++k)
      if (ctx->stream == s->side[k]) {
        HIPCHK(ctx, hipEventRecord(s->side_mid[k], s->side[k]));
        HIPCHK(ctx, hipStreamWaitEvent(main, s->side_mid[k], 0));
      }
    return PCLEAN_OK;
  }
  int join() {
    back();
    for (int k = 0; k < SweepState::MAX_SIDE; ++k)
      if (used[k]) {
        used[k] = false;
        HIPCHK(ctx, hipEventRecord(s->side_join[k], s->side[k]));
        HIPCHK(ctx, hipStreamWaitEvent(main, s->side_join[k], 0));
      }
    return PCLEAN_OK;
  }
  ~SideFork() {
    back();
    for (int k = 0; k < SweepState::MAX_SIDE; ++k)
      if (used[k]) (void)hipStreamSynchronize(s->side[k]);
  }
};

// ---- stable argsort of small integer ids (the evidence CSR of a latent class: observed rows ordered by the latent row they
// refer to, inference.jl:60-81's "every row of the class with everything that refers to it") -------------------------------
__global__ void ids_to_keys_kernel(int n, const int32_t* __restrict__ ids, uint32_t* __restrict__ key, int32_t* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  key[i] = (uint32_t)(ids[i] + 1);  // (-1 = no referent sorts first)
  idx[i] = i;
}
extern "C" int pclean_argsort_ids(pclean_ctx* ctx, int32_t n, const int32_t* ids, int32_t id_max, int32_t* order_out) {
  if (!ctx || n < 0 || (n > 0 && (!ids || !order_out)) || id_max < -1)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_argsort_ids: bad arguments");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  SweepState* s = st(ctx);
  int rc = begin_call(ctx);
  if (rc) return rc;
  int32_t* d_ids = scratch<int32_t>(ctx, n);
  uint32_t* key = scratch<uint32_t>(ctx, n);
  uint32_t* key_s = scratch<uint32_t>(ctx, n);
  int32_t* idx = scratch<int32_t>(ctx, n);
  int32_t* idx_s = scratch<int32_t>(ctx, n);
  if (!d_ids || !key || !key_s || !idx || !idx_s) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  int bits = 1;
  while (bits < 32 && ((uint64_t)1 << bits) <= (uint64_t)id_max + 1ull) ++bits;
  size_t tmp_bytes = 0;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key, key_s, idx, idx_s, n, 0, bits, ctx->stream));
  unsigned char* tmp = scratch<unsigned char>(ctx, std::max<size_t>(tmp_bytes, 16));
  if (!tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  const size_t bytes = (size_t)n * sizeof(int32_t);
  if (ctx->stage.grow(2 * bytes + 1024)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
  ctx->stage.rewind();
  void* h_in = ctx->stage.take(bytes);
  void* h_out = ctx->stage.take(bytes);
  if (!h_in || !h_out) return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
  memcpy(h_in, ids, bytes);
  HIPCHK(ctx, hipMemcpyAsync(d_ids, h_in, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(ids_to_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_ids, key, idx);
  // (LSD radix sort: stable — rows of one latent row stay in ascending order, as np.argsort(kind="stable") leaves them)
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key, key_s, idx, idx_s, n, 0, bits, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(h_out, idx_s, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PCLEAN_SYNC(ctx);
  memcpy(order_out, h_out, bytes);
  (void)s;
  return PCLEAN_OK;
}

// res_rows / res_ctx: the items' evidence rows (and per-row ctx values) already on the device (pclean_sweep_latent_resident);
// the host arrays ev_rows / ev_ctx are not looked at then (has_ctx says whether the plan's terms read per-row ctx values)
static int sweep_latent_impl(pclean_ctx* ctx, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep_idx,
                             int32_t block_id, int32_t n_roots, const int32_t* roots, int32_t n_items,
                             const int32_t* keys, const int32_t* ev_off, const int32_t* ev_rows,
                             const int32_t* ev_ctx, const int32_t* excl, int32_t* chosen, int32_t* vals,
                             const int32_t* res_rows, const int32_t* res_ctx) {
  if (!ctx || !cfg || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || n_roots <= 0 ||
      !roots || n_items < 0 || !keys || !ev_off || !excl || !chosen || !vals)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent: bad arguments");
  if (!cfg->use_dd_proposals) {
    const int rcp = prior_mode_supported(ctx, ctx->block[block_id], "pclean_sweep_latent");
    if (rcp) return rcp;
  }
  if (n_items == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Block& b = ctx->block[block_id];
  const int nn = (int)b.nodes.size();
  int P = cfg->num_particles;
  const int use_mh = cfg->use_mh_instead_of_pg != 0;
  if (use_mh) P = 2;
  if (P < 1 || P > MAXP) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent: bad particle count");
  for (int r = 0; r < n_roots; ++r)
    if (roots[r] < 0 || roots[r] >= nn) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent: bad root");
  SweepState* s = st(ctx);
  {
    const int rcb = begin_call(ctx);
    if (rcb) return rcb;
  }
  if (s->counter.alloc(4)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  const int n_ev = ev_off[n_items];
  if (n_ev > 0 && !ev_rows && !res_rows) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent: evidence rows missing");
  if (res_rows) {  // (resident evidence: nothing of it is staged)
    ev_rows = nullptr;
    ev_ctx = nullptr;
  }
  int32_t* d_keys = scratch<int32_t>(ctx, n_items);
  int32_t* d_off = scratch<int32_t>(ctx, (size_t)n_items + 1);
  int32_t* d_evr = res_rows ? const_cast<int32_t*>(res_rows) : scratch<int32_t>(ctx, std::max(n_ev, 1));
  int32_t* d_evc = res_rows ? const_cast<int32_t*>(res_ctx)
                            : (ev_ctx ? scratch<int32_t>(ctx, (size_t)std::max(n_ev, 1) * PCLEAN_MAX_CTX) : nullptr);
  int32_t* d_excl = scratch<int32_t>(ctx, (size_t)n_roots * n_items);
  int32_t* d_chosen = scratch<int32_t>(ctx, n_items);
  int32_t* d_vals = scratch<int32_t>(ctx, (size_t)n_items * nn);
  int32_t* d_flag = scratch<int32_t>(ctx, n_items);
  if (!d_keys || !d_off || !d_evr || (ev_ctx && !d_evc) || !d_excl || !d_chosen || !d_vals || !d_flag)
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  // inputs and outputs travel through the library's page-locked staging area (ctx.h: HostStage — never the caller's pages)
  const size_t b_keys = (size_t)n_items * 4, b_off = ((size_t)n_items + 1) * 4, b_evr = res_rows ? 0 : (size_t)n_ev * 4,
               b_evc = (ev_ctx && n_ev) ? (size_t)n_ev * PCLEAN_MAX_CTX * 4 : 0, b_excl = (size_t)n_roots * n_items * 4,
               b_vals = (size_t)n_items * nn * 4;
  if (ctx->stage.grow(2 * b_keys + b_off + b_evr + b_evc + b_excl + b_vals + 8 * 256))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
  ctx->stage.rewind();
  auto stage_up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    void* h = ctx->stage.take(bytes);
    memcpy(h, src, bytes);
    return hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, ctx->stream);
  };
  HIPCHK(ctx, stage_up(d_keys, keys, b_keys));
  HIPCHK(ctx, stage_up(d_off, ev_off, b_off));
  HIPCHK(ctx, stage_up(d_evr, ev_rows, b_evr));
  HIPCHK(ctx, stage_up(d_evc, ev_ctx, b_evc));
  HIPCHK(ctx, stage_up(d_excl, excl, b_excl));
  int32_t* h_chosen = (int32_t*)ctx->stage.take(b_keys);
  int32_t* h_vals = (int32_t*)ctx->stage.take(b_vals);
  int32_t* d_iop = scratch<int32_t>(ctx, std::max(n_ev, 1));
  if (!d_iop) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  if (n_ev) hipLaunchKernelGGL(item_of_pos_kernel, grid1(n_ev), dim3(256), 0, ctx->stream, n_ev, n_items, d_off, d_iop);
  s->lat_off = d_off;
  s->lat_item_of_pos = d_iop;
  s->lat_items = n_items;
  s->lat_ev = n_ev;
  s->lat_max_ev = 0;
  for (int t = 0; t < n_items; ++t) s->lat_max_ev = std::max(s->lat_max_ev, ev_off[t + 1] - ev_off[t]);
  s->lat_agg.clear();
  { const int rca = ensure_agg_all(ctx, block_id, d_evr, d_evc); if (rca) return rca; }
  if (!cfg->use_dd_proposals) {
    // Prior proposals (block_proposal.jl:168): particle 0 keeps the row's current values (excl[r][t]: current referent
    // of a reference slot, current OPTION of a choice), every other particle draws each attribute from its prior;
    // weight = likelihood of the referring rows given the particle's values; final choice among the particles.
    const size_t NPi = (size_t)n_items * P;
    static const bool lat_dbg = getenv("PCLEAN_DEBUG_LATENT") != nullptr;  // synchronise and say where we are (fault hunting)
    auto LATDBG = [&](const char* tag, int k) {
      if (!lat_dbg) return;
      const hipError_t e = hipStreamSynchronize(ctx->stream);
      fprintf(stderr, "[latent prior] block %d %s %d: %s\n", block_id, tag, k, hipGetErrorString(e));
      fflush(stderr);
    };
    int32_t* pv = scratch<int32_t>(ctx, NPi * nn);
    int32_t* draws = scratch<int32_t>(ctx, NPi);
    double* wl = scratch<double>(ctx, NPi);
    if (!pv || !draws || !wl) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    hipLaunchKernelGGL(fill_i32_kernel, grid1(NPi * nn), dim3(256), 0, ctx->stream, pv, NPi * nn, -2);
    for (int r = 0; r < n_roots; ++r) {
      const int root = roots[r];
      const pclean_node& rn = b.nodes[root];
      const int32_t* cur_r = d_excl + (size_t)r * n_items;
      ItemList ilp{n_items, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_keys};
      ctx->prior_mode = true;
      int rc = eval_node(ctx, block_id, root, ilp, rn.kind == PCLEAN_NODE_FK ? cur_r : nullptr, seed, sweep_idx, P, nullptr,
                         draws, nullptr, nullptr, false);
      if (rc) {
        ctx->prior_mode = false;
        return rc;
      }
      LATDBG("prior draws of root", root);
      hipLaunchKernelGGL(retain_first_kernel, grid1(n_items), dim3(256), 0, ctx->stream, n_items, P, cur_r, draws);
      hipLaunchKernelGGL(set_node_col_kernel, grid1(NPi), dim3(256), 0, ctx->stream, (int)NPi, draws, nn, root, pv);
      if (rn.kind == PCLEAN_NODE_FK && rn.n_children > 0) {
        HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
        int32_t* l2 = scratch<int32_t>(ctx, NPi);
        if (!l2) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        hipLaunchKernelGGL(compact_new_kernel, grid1(NPi), dim3(256), 0, ctx->stream, NPi, draws, 1, s->counter.p, l2, nullptr);
        unsigned int c2 = 0;
        PCLEAN_READ_COUNT(ctx, s->counter.p, &c2);
        if (c2) {
          int32_t* rng2 = scratch<int32_t>(ctx, c2);
          int32_t* part2 = scratch<int32_t>(ctx, c2);
          int32_t* org2 = scratch<int32_t>(ctx, c2);
          int32_t* ex2 = scratch<int32_t>(ctx, c2);
          int32_t* evl2 = scratch<int32_t>(ctx, c2);
          int32_t* evh2 = scratch<int32_t>(ctx, c2);
          if (!rng2 || !part2 || !org2 || !ex2 || !evl2 || !evh2) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          hipLaunchKernelGGL(latent_prior_items_kernel, grid1(c2), dim3(256), 0, ctx->stream, (int)c2, P, n_items, l2, d_keys,
                             cur_r, rng2, part2, org2, ex2, evl2, evh2, d_off);
          ItemList sub{(int)c2, nullptr, nullptr, part2, org2, nullptr, nullptr, nullptr, nullptr, rng2};
          rc = sample_children(ctx, block_id, root, sub, ex2, seed, sweep_idx, pv, nn);
          if (rc) {
            ctx->prior_mode = false;
            return rc;
          }
        }
      }
      ctx->prior_mode = false;
    }
    // likelihood of every (row, particle)
    const NodeDev* nds;
    const int32_t *dnc, *dcb, *dch;
    int rc = upload_plan_nodes(ctx, block_id, &nds, &dnc, &dcb, &dch);
    if (rc) return rc;
    int32_t* d_roots = scratch<int32_t>(ctx, n_roots);
    const AggDev** d_aggs = (const AggDev**)scratch<unsigned char>(ctx, sizeof(void*) * nn);
    if (!d_roots || !d_aggs) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    ItemList ilev{n_items, nullptr, nullptr, nullptr, nullptr, d_off, d_off + 1, d_evr, d_evc, d_keys};
    std::vector<const AggDev*> h_aggs(nn, nullptr);
    for (int node = 0; node < nn; ++node)
      if (b.nodes[node].n_terms > 0) {
        rc = ensure_agg(ctx, block_id, node, ilev, &h_aggs[node]);
        if (rc) return rc;
      }
    HIPCHK(ctx, hipMemcpyAsync(d_aggs, h_aggs.data(), sizeof(void*) * nn, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_roots, roots, (size_t)n_roots * 4, hipMemcpyHostToDevice, ctx->stream));
    PCLEAN_SYNC(ctx);  // h_aggs goes out of scope
    ItemsDev itd{n_items, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, d_off, d_off + 1, d_evr, d_evc, d_keys, nullptr,
                 nullptr, 0, 0, nullptr, nullptr};
    LATDBG("aggregated evidence", nn);
    if (lat_dbg)
      for (int node = 0; node < nn; ++node)
        if (node < (int)b.node_gauss.size() && b.node_gauss[node] >= 0) {
          (void)pclean_debug_gauss_ev_probe(ctx, n_items, P, nn, nds, node, itd, pv, (int)ctx->mean[b.gauss[b.node_gauss[node]].mean_table].v.n, n_ev);
        }
    rc = pclean_launch_prior_terms_ev(ctx, n_items, P, nn, nds, d_aggs, dnc, dcb, dch, n_roots, d_roots, itd, pv, wl);
    if (rc) return rc;
    LATDBG("prior_terms_ev_kernel", n_items);
    DISPATCH_PMAX(P, hipLaunchKernelGGL(latent_prior_choice_kernel<PMAX>, grid1(n_items), dim3(256), 0, ctx->stream, n_items, P,
                                        use_mh, wl, d_keys, seed, sweep_idx, (uint32_t)block_id, d_chosen));
    hipLaunchKernelGGL(gather_chosen_vals_kernel, grid1((size_t)n_items * nn), dim3(256), 0, ctx->stream, n_items, P, nn,
                       d_chosen, pv, d_vals);
    HIPCHK(ctx, hipMemcpyAsync(h_chosen, d_chosen, b_keys, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_vals, d_vals, b_vals, hipMemcpyDeviceToHost, ctx->stream));
    PCLEAN_SYNC(ctx);
    memcpy(chosen, h_chosen, b_keys);
    memcpy(vals, h_vals, b_vals);
    s->lat_agg.clear();
    if (s->prof_on) prof_collect(ctx);
    return finish_call(ctx);
  }
  hipLaunchKernelGGL(latent_choice_kernel, grid1(n_items), dim3(256), 0, ctx->stream, n_items, P, use_mh, d_keys, seed,
                     sweep_idx, (uint32_t)block_id, d_chosen);
  hipLaunchKernelGGL(fill_i32_kernel, grid1((size_t)n_items * nn), dim3(256), 0, ctx->stream, d_vals,
                     (size_t)n_items * nn, -2);
  // rows that take a fresh particle
  hipLaunchKernelGGL(mark_positive_kernel, grid1(n_items), dim3(256), 0, ctx->stream, n_items, d_chosen, d_flag);
  HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
  hipLaunchKernelGGL(compact_new_kernel, grid1(n_items), dim3(256), 0, ctx->stream, (size_t)n_items, d_flag, 0,
                     s->counter.p, nullptr, nullptr);
  unsigned int cnt = 0;
  PCLEAN_READ_COUNT(ctx, s->counter.p, &cnt);
  if (cnt) {
    int32_t* list = scratch<int32_t>(ctx, cnt);
    int32_t* rng = scratch<int32_t>(ctx, cnt);
    int32_t* evl = scratch<int32_t>(ctx, cnt);
    int32_t* evh = scratch<int32_t>(ctx, cnt);
    int32_t* part = scratch<int32_t>(ctx, cnt);
    int32_t* org = scratch<int32_t>(ctx, cnt);
    if (!list || !rng || !evl || !evh || !part || !org) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(compact_new_kernel, grid1(n_items), dim3(256), 0, ctx->stream, (size_t)n_items, d_flag, 1,
                       s->counter.p, list, nullptr);
    hipLaunchKernelGGL(latent_items_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, list, d_keys, d_off,
                       d_chosen, rng, evl, evh, part, org);
    ItemList il{(int)cnt, nullptr, nullptr, part, org, evl, evh, d_evr, d_evc, rng};
    // The roots of a latent row's plan are independent given its evidence (each writes its own column of d_vals, draws
    // at its own RNG site).  Three passes: (A) the reference slots' enumerations on the library's stream — queued first,
    // nothing in a sub-batch's pass waits for the host; (B) the option lists, each on a side stream, overlapping (A) and
    // each other; (C) back on the library's stream, per reference slot: how many rows proposed a NEW referent (the one
    // count the host needs) and the sampling of those referents' contents.  A batch large enough for the gate of the
    // new-row branch (a count read-back inside eval_node) queues its option lists before (A) instead.
    SideFork sf(ctx);
    std::vector<int> fk_roots, leaf_roots;
    for (int r = 0; r < n_roots; ++r) (b.nodes[roots[r]].kind == PCLEAN_NODE_LEAF ? leaf_roots : fk_roots).push_back(r);
    unsigned int* c2ctr = scratch<unsigned int>(ctx, std::max(n_roots, 1));
    if (!c2ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    HIPCHK(ctx, hipMemsetAsync(c2ctr, 0, (size_t)std::max(n_roots, 1) * sizeof(unsigned int), ctx->stream));
    std::vector<int32_t*> fk_ex(n_roots, nullptr), fk_l2(n_roots, nullptr);
    {
      const int rcf = sf.fork();  // (before pass A is queued: the side streams wait for the inputs, not for pass A)
      if (rcf) return rcf;
    }
    const char* gm = getenv("PCLEAN_GATE_MIN");
    const bool fk_first = (int)cnt < (gm ? atoi(gm) : 2048);
    int n_side_used = 0;
    auto pass_fk = [&]() -> int {
      for (int r : fk_roots) {
        const int root = roots[r];
        const pclean_node& rn = b.nodes[root];
        if (fk_first) {  // (no count read-back, no shared counter on this path below the gate's size: a stream of its own)
          const int rcs = sf.use(n_side_used++);
          if (rcs) return rcs;
        }
        int32_t* ex = scratch<int32_t>(ctx, cnt);
        int32_t* draws = scratch<int32_t>(ctx, cnt);
        if (!ex || !draws) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        hipLaunchKernelGGL(gather_i32_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, list,
                           d_excl + (size_t)r * n_items, ex);
        fk_ex[r] = ex;
        int rc = eval_node(ctx, block_id, root, il, ex, seed, sweep_idx, 1, nullptr, draws, nullptr, nullptr, false);
        if (rc) return rc;
        hipLaunchKernelGGL(scatter_vals_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, org, draws, nn, root,
                           d_vals);
        if (rn.n_children > 0) {  // rows that proposed a NEW referent: listed now, counted by the host in pass C
          fk_l2[r] = scratch<int32_t>(ctx, cnt);
          if (!fk_l2[r]) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          hipLaunchKernelGGL(compact_new_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (size_t)cnt, draws, 1, c2ctr + r,
                             fk_l2[r], nullptr);
        }
        const int rcm = sf.mark();  // pass C (library's stream) follows this root's pass A, not the option lists
        if (rcm) return rcm;
      }
      sf.back();
      return PCLEAN_OK;
    };
    auto pass_leaf = [&]() -> int {
      for (size_t oi = 0; oi < leaf_roots.size(); ++oi) {
        const int r = leaf_roots[oi];
        const int root = roots[r];
        // (the reference slots keep their streams to themselves: theirs are the longest chains of a sub-batch)
        const int K = side_streams(ctx), n_fk_side = fk_first ? std::min((int)fk_roots.size(), std::max(K - 1, 0)) : 0;
        const int rcs = sf.use(K > n_fk_side ? n_fk_side + (int)oi % (K - n_fk_side) : (int)oi);
        if (rcs) return rcs;
        int32_t* draws = scratch<int32_t>(ctx, cnt);
        if (!draws) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        int rc = eval_node(ctx, block_id, root, il, nullptr, seed, sweep_idx, 1, nullptr, draws, nullptr, nullptr, false);
        if (rc) return rc;
        hipLaunchKernelGGL(scatter_vals_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, org, draws, nn, root,
                           d_vals);
      }
      sf.back();
      return PCLEAN_OK;
    };
    int rc = fk_first ? pass_fk() : pass_leaf();
    if (!rc) rc = fk_first ? pass_leaf() : pass_fk();
    if (rc) return rc;
    for (int r : fk_roots) {
      const int root = roots[r];
      if (!fk_l2[r]) continue;
      // referents proposed as NEW: sample their contents with the same evidence
      unsigned int c2 = 0;
      PCLEAN_READ_COUNT(ctx, c2ctr + r, &c2);
      if (c2) {
        int32_t* l2 = fk_l2[r];
        int32_t* row2 = scratch<int32_t>(ctx, c2);
        int32_t* cx2 = scratch<int32_t>(ctx, (size_t)c2 * PCLEAN_MAX_CTX);
        int32_t* part2 = scratch<int32_t>(ctx, c2);
        int32_t* org2 = scratch<int32_t>(ctx, c2);
        int32_t* evl2 = scratch<int32_t>(ctx, c2);
        int32_t* evh2 = scratch<int32_t>(ctx, c2);
        int32_t* rng2 = scratch<int32_t>(ctx, c2);
        int32_t* ex2 = scratch<int32_t>(ctx, c2);
        if (!row2 || !cx2 || !part2 || !org2 || !evl2 || !evh2 || !rng2 || !ex2)
          return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        hipLaunchKernelGGL(sublist_items_kernel, grid1(c2), dim3(256), 0, ctx->stream, (int)c2, l2, il.row, il.ctx,
                           il.particle, il.origin, row2, cx2, part2, org2, il.ev_lo, il.ev_hi, il.rng_row, evl2, evh2,
                           rng2);
        hipLaunchKernelGGL(gather_i32_kernel, grid1(c2), dim3(256), 0, ctx->stream, (int)c2, l2, fk_ex[r], ex2);
        ItemList sub{(int)c2, nullptr, nullptr, part2, org2, evl2, evh2, d_evr, d_evc, rng2};
        rc = sample_children(ctx, block_id, root, sub, ex2, seed, sweep_idx, d_vals, nn);
        if (rc) return rc;
      }
    }
    const int rcj = sf.join();
    if (rcj) return rcj;
  }
  HIPCHK(ctx, hipMemcpyAsync(h_chosen, d_chosen, b_keys, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(h_vals, d_vals, b_vals, hipMemcpyDeviceToHost, ctx->stream));
  {
    int rcq = queue_over_copy(ctx);  // the sync-free re-runs' counts ride on the call's one synchronisation
    if (!rcq) rcq = d2h_flush(ctx);
    if (rcq) return rcq;
  }
  PCLEAN_SYNC(ctx);
  memcpy(chosen, h_chosen, b_keys);
  memcpy(vals, h_vals, b_vals);
  apply_over_stats(ctx);
  s->lat_agg.clear();
  if (s->prof_on) prof_collect(ctx);
  return finish_call(ctx);
}

extern "C" int pclean_sweep_latent(pclean_ctx* ctx, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep_idx,
                                   int32_t block_id, int32_t n_roots, const int32_t* roots, int32_t n_items,
                                   const int32_t* keys, const int32_t* ev_off, const int32_t* ev_rows,
                                   const int32_t* ev_ctx, const int32_t* excl, int32_t* chosen, int32_t* vals) {
  return sweep_latent_impl(ctx, cfg, seed, sweep_idx, block_id, n_roots, roots, n_items, keys, ev_off, ev_rows, ev_ctx, excl,
                           chosen, vals, nullptr, nullptr);
}

// ---- the evidence CSR of a latent class on the device ----------------------------------------------------------------
// What the host's build_evidence (inference.py) does with NumPy on 10^6 rows per class and iteration — follow the reference
// slots from every observed row to the class's row, order the observed rows by that row (stable), gather the per-row context
// values — with the referents and the tables where they already are (pclean_set_cur / the device-resident commit,
// pclean_set_table).  The ordered rows and their ctx values stay on the device for the sub-batches of the class sweep.
struct EvFollowDev {
  int32_t n_steps;
  const int32_t* col[PCLEAN_EV_MAX_STEPS];  // step s: the reference-slot column of the table the walk stands in
  int32_t n_rows[PCLEAN_EV_MAX_STEPS];
};
struct EvSrcDev {
  int32_t n;
  const int32_t* cur[PCLEAN_MAX_CTX];  // the observed rows' referents in the source's block
  const int32_t* col[PCLEAN_MAX_CTX];  // the value column of that block's root table
  int32_t n_rows[PCLEAN_MAX_CTX];
};
__global__ void ev_follow_kernel(int n, const int32_t* __restrict__ cur, EvFollowDev f, uint32_t* __restrict__ key,
                                 int32_t* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int k = cur[i];
  for (int s = 0; s < f.n_steps; ++s) k = (k >= 0 && k < f.n_rows[s]) ? f.col[s][k] : -1;
  key[i] = (uint32_t)(k + 1);  // (-1 = no referent sorts first)
  idx[i] = i;
}
// off[k] = number of sorted keys below k + 1 = first position of latent row k's evidence rows, k = 0 .. n_target
__global__ void ev_offsets_kernel(int n, const uint32_t* __restrict__ key_s, int n_target, int32_t* __restrict__ off) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_target) return;
  const uint32_t want = (uint32_t)k + 1u;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (key_s[mid] < want) lo = mid + 1; else hi = mid;
  }
  off[k] = lo;
}
__global__ void ev_ctx_kernel(int n, const int32_t* __restrict__ order, EvSrcDev sv, int32_t* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int r = order[e];
  int32_t v[PCLEAN_MAX_CTX];
#pragma unroll
  for (int q = 0; q < PCLEAN_MAX_CTX; ++q) {
    v[q] = 0;
    if (q < sv.n) {
      const int k = sv.cur[q][r];
      v[q] = (k >= 0 && k < sv.n_rows[q]) ? sv.col[q][k] : 0;
    }
  }
#pragma unroll
  for (int q = 0; q < PCLEAN_MAX_CTX; ++q) out[(size_t)e * PCLEAN_MAX_CTX + q] = v[q];
}

extern "C" int pclean_build_evidence(pclean_ctx* ctx, int32_t cur_block, int32_t n_steps, const int32_t* step_table,
                                     const int32_t* step_col, int32_t n_target_rows, int32_t n_src, const int32_t* src_block,
                                     const int32_t* src_table, const int32_t* src_col, int32_t* off_out) {
  if (!ctx || n_steps < 0 || n_steps > PCLEAN_EV_MAX_STEPS || (n_steps > 0 && (!step_table || !step_col)) || n_target_rows < 0 ||
      n_src < 0 || n_src > PCLEAN_MAX_CTX || (n_src > 0 && (!src_block || !src_table || !src_col)) || !off_out)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_build_evidence: bad arguments");
  if (!ctx->dev_cur_valid || cur_block < 0 || cur_block >= ctx->dev_cur_blocks || ctx->n_rows <= 0)
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_build_evidence: no device-resident referents (pclean_set_cur)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->ev_res_n = 0;
  const int n = ctx->n_rows;
  EvFollowDev f{};
  f.n_steps = n_steps;
  for (int s = 0; s < n_steps; ++s) {
    if (step_table[s] < 0 || step_table[s] >= PCLEAN_MAX_TABLES || !ctx->cand[step_table[s]].valid ||
        ctx->cand[step_table[s]].is_options || step_col[s] < 0 || step_col[s] >= ctx->cand[step_table[s]].n_cols)
      return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_build_evidence: bad step %d", s);
    const CandTable& t = ctx->cand[step_table[s]];
    f.col[s] = t.cols.p + (size_t)step_col[s] * t.n_rows;
    f.n_rows[s] = t.n_rows;
  }
  EvSrcDev sv{};
  sv.n = n_src;
  for (int q = 0; q < n_src; ++q) {
    if (src_block[q] < 0 || src_block[q] >= ctx->dev_cur_blocks || src_table[q] < 0 || src_table[q] >= PCLEAN_MAX_TABLES ||
        !ctx->cand[src_table[q]].valid || ctx->cand[src_table[q]].is_options || src_col[q] < 0 ||
        src_col[q] >= ctx->cand[src_table[q]].n_cols)
      return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_build_evidence: bad ctx source %d", q);
    const CandTable& t = ctx->cand[src_table[q]];
    sv.cur[q] = ctx->dev_cur.p + (size_t)src_block[q] * ctx->n_rows;
    sv.col[q] = t.cols.p + (size_t)src_col[q] * t.n_rows;
    sv.n_rows[q] = t.n_rows;
  }
  int rc = begin_call(ctx);
  if (rc) return rc;
  if (ctx->ev_res_rows.n < (size_t)n && ctx->ev_res_rows.alloc((size_t)n + (size_t)n / 16))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (n_src > 0 && ctx->ev_res_ctx.n < (size_t)n * PCLEAN_MAX_CTX && ctx->ev_res_ctx.alloc(((size_t)n + (size_t)n / 16) * PCLEAN_MAX_CTX))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  uint32_t* key = scratch<uint32_t>(ctx, n);
  uint32_t* key_s = scratch<uint32_t>(ctx, n);
  int32_t* idx = scratch<int32_t>(ctx, n);
  int32_t* d_off = scratch<int32_t>(ctx, (size_t)n_target_rows + 1);
  if (!key || !key_s || !idx || !d_off) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  int bits = 1;
  while (bits < 32 && ((uint64_t)1 << bits) <= (uint64_t)n_target_rows + 1ull) ++bits;
  size_t tmp_bytes = 0;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key, key_s, idx, ctx->ev_res_rows.p, n, 0, bits, ctx->stream));
  unsigned char* tmp = scratch<unsigned char>(ctx, std::max<size_t>(tmp_bytes, 16));
  if (!tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  const size_t b_off = ((size_t)n_target_rows + 1) * sizeof(int32_t);
  if (ctx->stage.grow(b_off + 1024)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
  ctx->stage.rewind();
  void* h_off = ctx->stage.take(b_off);
  if (!h_off) return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
  hipLaunchKernelGGL(ev_follow_kernel, grid1(n), dim3(256), 0, ctx->stream, n, ctx->dev_cur.p + (size_t)cur_block * ctx->n_rows, f,
                     key, idx);
  // (LSD radix sort: stable — the rows of one latent row stay in ascending order, as np.argsort(kind="stable") leaves them)
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key, key_s, idx, ctx->ev_res_rows.p, n, 0, bits, ctx->stream));
  hipLaunchKernelGGL(ev_offsets_kernel, grid1((size_t)n_target_rows + 1), dim3(256), 0, ctx->stream, n, key_s, n_target_rows, d_off);
  if (n_src > 0)
    hipLaunchKernelGGL(ev_ctx_kernel, grid1(n), dim3(256), 0, ctx->stream, n, ctx->ev_res_rows.p, sv, ctx->ev_res_ctx.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(h_off, d_off, b_off, hipMemcpyDeviceToHost, ctx->stream));
  PCLEAN_SYNC(ctx);
  memcpy(off_out, h_off, b_off);
  ctx->ev_res_n = n;
  ctx->ev_res_has_ctx = n_src > 0;
  return PCLEAN_OK;
}

extern "C" int pclean_get_evidence(pclean_ctx* ctx, int32_t begin, int32_t n, int32_t* rows_out, int32_t* ctx_out) {
  if (!ctx || begin < 0 || n < 0 || (n > 0 && !rows_out)) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_get_evidence: bad arguments");
  if ((int64_t)begin + n > ctx->ev_res_n) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_get_evidence: range outside the resident evidence");
  if (n == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(rows_out, ctx->ev_res_rows.p + begin, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (ctx_out) {
    if (ctx->ev_res_has_ctx)
      HIPCHK(ctx, hipMemcpyAsync(ctx_out, ctx->ev_res_ctx.p + (size_t)begin * PCLEAN_MAX_CTX, (size_t)n * PCLEAN_MAX_CTX * 4,
                                 hipMemcpyDeviceToHost, ctx->stream));
    else
      memset(ctx_out, 0, (size_t)n * PCLEAN_MAX_CTX * 4);
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PCLEAN_OK;
}

extern "C" int pclean_sweep_latent_resident(pclean_ctx* ctx, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep_idx,
                                            int32_t block_id, int32_t n_roots, const int32_t* roots, int32_t n_items,
                                            const int32_t* keys, const int32_t* ev_off, int32_t ev_begin,
                                            const int32_t* excl, int32_t* chosen, int32_t* vals) {
  if (!ctx || !ev_off || n_items < 0) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent_resident: bad arguments");
  if (ctx->ev_res_n <= 0) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_sweep_latent_resident: no resident evidence (pclean_build_evidence)");
  if (ev_begin < 0 || ev_off[0] != 0 || ev_off[n_items] < 0 || (int64_t)ev_begin + ev_off[n_items] > ctx->ev_res_n)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent_resident: evidence range outside the resident rows");
  return sweep_latent_impl(ctx, cfg, seed, sweep_idx, block_id, n_roots, roots, n_items, keys, ev_off, nullptr, nullptr, excl,
                           chosen, vals, ctx->ev_res_rows.p + ev_begin,
                           ctx->ev_res_has_ctx ? ctx->ev_res_ctx.p + (size_t)ev_begin * PCLEAN_MAX_CTX : nullptr);
}

// pclean_score_node for EVIDENCE SETS: item t is a latent row scored against the observed rows
// ev_rows[ev_off[t] .. ev_off[t + 1]) (with their per-row ctx) — one plan node of a latent class's plan, as
// pclean_sweep_latent evaluates it (same aggregation, same kernels), with the per-candidate scores returned.
extern "C" int pclean_score_node_ev(pclean_ctx* ctx, int32_t block_id, int32_t node_id, int32_t n_items,
                                    const int32_t* keys, const int32_t* ev_off, const int32_t* ev_rows,
                                    const int32_t* ev_ctx, const int32_t* excl, uint64_t seed, uint32_t sweep,
                                    int32_t n_draws, double* lse, double* scores, int32_t* draws) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || n_items <= 0 || !keys ||
      !ev_off || n_draws < 0 || n_draws > 1 || (n_draws > 0 && !draws))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_score_node_ev: bad arguments");
  Block& b = ctx->block[block_id];
  if (node_id < 0 || node_id >= (int)b.nodes.size()) return pclean_fail(ctx, PCLEAN_ERR_ARG, "bad node id");
  const int n_ev = ev_off[n_items];
  if (n_ev < 0 || (n_ev > 0 && !ev_rows)) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_score_node_ev: evidence rows missing");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  {
    const int rcb = begin_call(ctx);
    if (rcb) return rcb;
  }
  SweepState* s = st(ctx);
  const pclean_node& n = b.nodes[node_id];
  const CandTable& t = ctx->cand[n.table];
  const int nc = t.n_rows + (n.kind == PCLEAN_NODE_FK ? 1 : 0);
  int32_t* d_keys = scratch<int32_t>(ctx, n_items);
  int32_t* d_off = scratch<int32_t>(ctx, (size_t)n_items + 1);
  int32_t* d_evr = scratch<int32_t>(ctx, std::max(n_ev, 1));
  int32_t* d_evc = ev_ctx ? scratch<int32_t>(ctx, (size_t)std::max(n_ev, 1) * PCLEAN_MAX_CTX) : nullptr;
  int32_t* d_iop = scratch<int32_t>(ctx, std::max(n_ev, 1));
  int32_t* d_excl = excl ? scratch<int32_t>(ctx, n_items) : nullptr;
  double* d_lse = scratch<double>(ctx, n_items);
  double* d_scores = scores ? scratch<double>(ctx, (size_t)n_items * nc) : nullptr;
  int32_t* d_draws = n_draws ? scratch<int32_t>(ctx, n_items) : nullptr;
  if (!d_keys || !d_off || !d_evr || (ev_ctx && !d_evc) || !d_iop || (excl && !d_excl) || !d_lse || (scores && !d_scores) ||
      (n_draws && !d_draws))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  HIPCHK(ctx, hipMemcpyAsync(d_keys, keys, (size_t)n_items * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_off, ev_off, ((size_t)n_items + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_ev) HIPCHK(ctx, hipMemcpyAsync(d_evr, ev_rows, (size_t)n_ev * 4, hipMemcpyHostToDevice, ctx->stream));
  if (ev_ctx && n_ev)
    HIPCHK(ctx, hipMemcpyAsync(d_evc, ev_ctx, (size_t)n_ev * PCLEAN_MAX_CTX * 4, hipMemcpyHostToDevice, ctx->stream));
  if (excl) HIPCHK(ctx, hipMemcpyAsync(d_excl, excl, (size_t)n_items * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_ev) hipLaunchKernelGGL(item_of_pos_kernel, grid1(n_ev), dim3(256), 0, ctx->stream, n_ev, n_items, d_off, d_iop);
  s->lat_off = d_off;
  s->lat_item_of_pos = d_iop;
  s->lat_items = n_items;
  s->lat_ev = n_ev;
  s->lat_max_ev = 0;
  for (int i = 0; i < n_items; ++i) s->lat_max_ev = std::max(s->lat_max_ev, ev_off[i + 1] - ev_off[i]);
  s->lat_agg.clear();
  ItemList il{n_items, nullptr, nullptr, nullptr, nullptr, d_off, d_off + 1, d_evr, d_evc, d_keys};
  int rc = eval_node(ctx, block_id, node_id, il, d_excl, seed, sweep, n_draws, d_lse, d_draws, d_scores, nullptr, false);
  s->lat_agg.clear();
  if (rc) return rc;
  if (lse) HIPCHK(ctx, hipMemcpyAsync(lse, d_lse, (size_t)n_items * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (scores) HIPCHK(ctx, hipMemcpyAsync(scores, d_scores, (size_t)n_items * nc * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (n_draws) HIPCHK(ctx, hipMemcpyAsync(draws, d_draws, (size_t)n_items * 4, hipMemcpyDeviceToHost, ctx->stream));
  PCLEAN_SYNC(ctx);
  return finish_call(ctx);
}
