// Evaluation of one plan node for a list of work items (process_plan!, proposal_compiler.jl:363-388): option lists
// (LEAF nodes) and reference slots (FK nodes) with the log-marginals of a new row's children; the compact-table fast
// path (root_wave.hip) with its caches, item de-duplication (sorted groups of identical score vectors), the memo of
// option-list marginals, sampling of the contents of rows proposed as NEW.  Called by the observed-class sweep
// (sweep.hip) and the latent-class sweep (latent.hip).
#include "sweep_internal.h"

int build_gauss_dev(pclean_ctx* ctx, const pclean_gauss& g, const CandTable* t, GaussDev& d) {
  memset(&d, 0, sizeof d);
  if (g.x_col < 0 || g.x_col >= ctx->n_xcols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "gauss: numeric column out of range");
  const MeanTable& m = ctx->mean[g.mean_table];
  if (!m.valid) return pclean_fail(ctx, PCLEAN_ERR_STATE, "gauss: mean table %d not set", g.mean_table);
  d.on = 1;
  d.n_dims = g.n_dims;
  d.n_locals = g.n_locals;
  d.x = ctx->xnum.p + (size_t)g.x_col * ctx->n_rows + ctx->active_begin;
  d.mu = m.v.p;
  for (int i = 0; i < g.n_dims; ++i) {
    d.src_kind[i] = g.src_kind[i];
    d.src_slot[i] = g.src[i];
    d.stride[i] = g.stride[i];
    d.src_ptr[i] = nullptr;
    if (g.src_kind[i] == PCLEAN_GSRC_CAND) {
      if (!t || g.src[i] < 0 || g.src[i] >= t->n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "gauss: candidate column out of range");
      d.src_ptr[i] = t->cols.p + (size_t)g.src[i] * t->n_rows;
    } else if (g.src_kind[i] == PCLEAN_GSRC_OBS) {
      if (g.src[i] < 0 || g.src[i] >= ctx->n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "gauss: observed column out of range");
      d.src_ptr[i] = ctx->obs.p + (size_t)g.src[i] * ctx->n_rows + ctx->active_begin;
    }
  }
  for (int l = 0; l < 2; ++l) {
    d.local_n[l] = l < g.n_locals ? g.local_n[l] : 1;
    d.local_logp[l] = l < g.n_locals ? -std::log((double)g.local_n[l]) : 0.0;  // choose_uniformly.jl:7-10
    d.local_obs[l] = nullptr;
    if (l < g.n_locals && g.local_obs_col[l] >= 0) {
      if (g.local_obs_col[l] >= ctx->n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "gauss: local observed column out of range");
      d.local_obs[l] = ctx->obs.p + (size_t)g.local_obs_col[l] * ctx->n_rows + ctx->active_begin;
    }
  }
  d.t_kind = g.transform_src_kind;
  d.t_src = g.transform_src;
  for (int u = 0; u < 4; ++u) {
    d.t_scale[u] = g.t_scale[u];
    d.t_lad[u] = g.t_logabsderiv[u];
    d.tx[u] = d.tl[u] = nullptr;
    if (g.t_x_col[u] >= 0 || g.t_lad_col[u] >= 0) {  // a non-linear Transformation: both per-row columns
      if (g.t_x_col[u] < 0 || g.t_x_col[u] >= ctx->n_xcols || g.t_lad_col[u] < 0 || g.t_lad_col[u] >= ctx->n_xcols)
        return pclean_fail(ctx, PCLEAN_ERR_ARG, "gauss: transformation column out of range");
      d.tx[u] = ctx->xnum.p + (size_t)g.t_x_col[u] * ctx->n_rows + ctx->active_begin;
      d.tl[u] = ctx->xnum.p + (size_t)g.t_lad_col[u] * ctx->n_rows + ctx->active_begin;
    }
  }
  d.sigma = g.sigma;
  d.log_sigma = std::log(g.sigma);
  return PCLEAN_OK;
}

int build_node_dev(pclean_ctx* ctx, const Block& b, int node_id, NodeDev& nd) {
  const pclean_node& n = b.nodes[node_id];
  const CandTable& t = ctx->cand[n.table];
  if (!t.valid) return pclean_fail(ctx, PCLEAN_ERR_STATE, "node %d: candidate table %d not set", node_id, n.table);
  if (n.n_terms > PCLEAN_MAX_TERMS) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "too many terms on one node");
  if ((n.kind == PCLEAN_NODE_FK) == t.is_options)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "node %d: kind does not match table %d", node_id, n.table);
  nd.kind = n.kind;
  nd.n_cand = t.n_rows;
  nd.n_terms = n.n_terms;
  nd.counts = t.counts.p;
  nd.logc_full = t.logc_full.p;
  nd.logc_m1 = t.logc_m1.p;
  memcpy(nd.scal, t.scal, sizeof nd.scal);
  memset(&nd.g, 0, sizeof nd.g);
  if (node_id < (int)b.node_gauss.size() && b.node_gauss[node_id] >= 0) {
    int rc = build_gauss_dev(ctx, b.gauss[b.node_gauss[node_id]], &t, nd.g);
    if (rc) return rc;
  }
  if (ctx->prior_mode) {  // prior proposals: candidates are drawn from the prior alone (block_proposal.jl:42-56, 68-84)
    nd.n_terms = 0;
    memset(&nd.g, 0, sizeof nd.g);
    return PCLEAN_OK;
  }
  for (int i = 0; i < n.n_terms; ++i) {
    const pclean_term& tm = b.terms[n.term_begin + i];
    const PairTable& pt = ctx->pair[tm.pair_table];
    if (!pt.valid) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pair table %d not built", tm.pair_table);
    if (tm.obs_col < 0 || tm.obs_col >= ctx->n_cols || tm.cand_col < 0 || tm.cand_col >= t.n_cols)
      return pclean_fail(ctx, PCLEAN_ERR_ARG, "term %d: column out of range", n.term_begin + i);
    TermDev& td = nd.terms[i];
    td.obs_col = ctx->obs_override ? ctx->obs_override : ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows + ctx->active_begin;
    td.ctx_mode = tm.ctx_mode;
    td.pad = 0;
    td.cand_col = t.cols.p + (size_t)tm.cand_col * t.n_rows;
    td.pair = pt.d.p;
    td.lat_len = pt.lat_len.p;
    td.n_lat = pt.n_lat;
    td.elem_bytes = pt.elem_bytes;
    td.dens_kind = tm.dens_kind;
    td.max_typos = tm.max_typos;
    td.ctx_slot = tm.ctx_slot;
    td.fn = nullptr;
    td.fn_nb = 0;
    td.aux_col = nullptr;
    td.other_val = -1;
    td.pad2 = 0;
    if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP) {
      if (tm.max_typos < 0 || tm.max_typos >= t.n_cols || tm.ctx_slot < 0 || ctx->n_prob == 0)
        return pclean_fail(ctx, PCLEAN_ERR_ARG, "MaybeSwap term %d: needs an option-count column, a ctx slot and a prob table",
                           n.term_begin + i);
      td.aux_col = t.cols.p + (size_t)tm.max_typos * t.n_rows;
      td.other_val = tm.fn_table;
      continue;
    }
    if (tm.ctx_slot >= 0) {
      const FnTable& f = ctx->fn[tm.fn_table];
      if (!f.valid) return pclean_fail(ctx, PCLEAN_ERR_STATE, "fn table %d not set", tm.fn_table);
      td.fn = f.fn.p;
      td.fn_nb = f.n_b;
    }
  }
  return PCLEAN_OK;
}

struct ItemList;
// Per-unique-observed-value marginal of a cacheable leaf (one term, no ctx):
// cache[u] for u < n_obs, cache[n_obs] for a missing observation.
int ensure_leaf_cache(pclean_ctx* ctx, int block_id, int node_id, const double** out, const int32_t** obs_col,
                             int* n_obs) {
  Block& b = ctx->block[block_id];
  const pclean_node& n = b.nodes[node_id];
  if (n.n_terms != 1) return pclean_fail(ctx, PCLEAN_ERR_ARG, "cacheable leaf %d must have exactly one term", node_id);
  const pclean_term& tm = b.terms[n.term_begin];
  if (tm.ctx_slot >= 0) return pclean_fail(ctx, PCLEAN_ERR_ARG, "cacheable leaf %d must not use ctx", node_id);
  const PairTable& pt = ctx->pair[tm.pair_table];
  SweepState* s = st(ctx);
  const int key = block_id * 256 + node_id;
  const int U = pt.n_obs;
  DevBuf<int32_t>& io = s->leaf_iota[key];
  if (io.n < (size_t)U + 1) {
    if (io.alloc(U + 1)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "alloc");
    hipLaunchKernelGGL(iota_missing_kernel, grid1(U + 1), dim3(256), 0, ctx->stream, io.p, U);
  }
  DevBuf<double>& cache = b.leaf_cache[node_id];
  // the marginal only depends on the option table and the pair table: recompute when either was re-uploaded
  const uint64_t ver = ctx->cand[n.table].version * 1000003ull + pt.version;
  auto itv = s->leaf_version.find(key);
  if (itv == s->leaf_version.end() || itv->second != ver || cache.n < (size_t)U + 1) {
    // item t observes value t (or a missing value for t == U): every option of every value once, leaving the
    // log-marginal, the maximum, the fixed-point total and the coarse prefix (enum_kernels.hip: leaf_coarse_build_kernel)
    ProfScope ps(ctx, "leaf_cache_rebuild");
    const int nblk = pclean_leaf_coarse_blocks(ctx->cand[n.table].n_rows);
    if (cache.alloc(U + 1) || b.leaf_m[node_id].alloc(U + 1) || b.leaf_U[node_id].alloc(U + 1) ||
        b.leaf_coarse[node_id].alloc((size_t)(U + 1) * nblk))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "alloc");
    NodeDev nd;
    ctx->obs_override = io.p;
    // the cache holds the DATA-DRIVEN marginal whatever the running sweep proposes from: build_node_dev drops a node's
    // terms in prior mode (use_dd_proposals = false), which must never reach a cache keyed by table versions alone
    const bool prior_saved = ctx->prior_mode;
    ctx->prior_mode = false;
    int rc = build_node_dev(ctx, b, node_id, nd);
    ctx->prior_mode = prior_saved;
    ctx->obs_override = nullptr;
    if (rc) return rc;
    ItemsDev it{U + 1, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                nullptr, nullptr, 0, 0, nullptr, nullptr};
    // the ProposalDummyValue option, if the list has one: its fixed-point weight per observed value tells whether a
    // particle can draw it at all (block_dummy_drawable)
    int dummy_k = -1;
    if (n.dummy_value != 0) {
      const std::vector<int32_t>& hv = ctx->cand[n.table].h_vals;
      for (size_t k = 0; k < hv.size(); ++k)
        if (hv[k] == n.dummy_value - 1) dummy_k = (int)k;
      if (dummy_k >= 0) {
        if (b.leaf_udummy[node_id].alloc(U + 1)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "alloc");
        HIPCHK(ctx, hipMemsetAsync(b.leaf_udummy[node_id].p, 0, (size_t)(U + 1) * sizeof(uint64_t), ctx->stream));
      }
    }
    rc = pclean_launch_leaf_coarse_build(ctx, nd, it, nblk, cache.p, b.leaf_m[node_id].p, b.leaf_U[node_id].p,
                                         b.leaf_coarse[node_id].p, dummy_k, dummy_k >= 0 ? b.leaf_udummy[node_id].p : nullptr);
    if (rc) return rc;
    b.leaf_drawable[node_id] = 0;
    if (dummy_k >= 0) {  // (once per rebuild of the cache: a read-back is affordable)
      std::vector<uint64_t> hu((size_t)U + 1);
      HIPCHK(ctx, hipMemcpyAsync(hu.data(), b.leaf_udummy[node_id].p, hu.size() * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                 ctx->stream));
      PCLEAN_SYNC(ctx);
      bool any = tm.obs_col >= 0 && tm.obs_col < (int)ctx->col_has_missing.size() && ctx->col_has_missing[tm.obs_col] &&
                 hu[U] != 0;
      int first_o = -1;
      for (int o = 0; o < U && !any; ++o)
        if (hu[o] != 0) {
          any = true;
          first_o = o;
        }
      b.leaf_drawable[node_id] = any ? 1 : 0;
      if (getenv("PCLEAN_DEBUG_DUMMY"))
        fprintf(stderr, "[pclean] block %d node %d: dummy option %d, drawable %d (first observed value %d, weight %llu; missing-value weight %llu)\n",
                block_id, node_id, dummy_k, any ? 1 : 0, first_o, first_o >= 0 ? (unsigned long long)hu[first_o] : 0ull,
                (unsigned long long)hu[U]);
    }
    s->leaf_version[key] = ver;
  }
  *out = cache.p;
  *obs_col = ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows + ctx->active_begin;
  *n_obs = U;
  return PCLEAN_OK;
}

// The (up to three) terms of node n whose byte rows the integer pre-filter of root_wave.hip sums: plain
// (compact-table) terms, longest latent strings first.  Returns their number; pre[p] = index within the node.
static int prefilter_terms(pclean_ctx* ctx, const Block& b, const pclean_node& n, int32_t pre[3]) {
  int order[PCLEAN_MAX_TERMS];
  const int nt = std::min(n.n_terms, PCLEAN_MAX_TERMS);
  for (int i = 0; i < nt; ++i) order[i] = i;
  auto plain = [&](const pclean_term& tm) {
    return tm.ctx_slot < 0 && tm.dens_kind == PCLEAN_DENS_ADD_TYPOS && tm.pair_table >= 0 && tm.pair_table < PCLEAN_MAX_TABLES &&
           ctx->pair[tm.pair_table].valid;
  };
  std::stable_sort(order, order + nt, [&](int a, int c) {
    const pclean_term& ta = b.terms[n.term_begin + a];
    const pclean_term& tc = b.terms[n.term_begin + c];
    if (plain(ta) != plain(tc)) return plain(ta);  // compact-table terms first
    if (!plain(ta)) return false;
    return ctx->pair[ta.pair_table].max_lat_len > ctx->pair[tc.pair_table].max_lat_len;
  });
  int n_compact = 0;
  for (int i = 0; i < nt; ++i) n_compact += plain(b.terms[n.term_begin + i]) ? 1 : 0;
  const int n_pre = std::min(3, n_compact);
  for (int p = 0; p < 3; ++p) pre[p] = p < n_pre ? order[p] : 0;
  return n_pre;
}

// Fast path of a reference slot (root_wave.hip): returns 1 and fills `fr` when the node is an FK
// with many candidates whose terms are all plain AddTypos lookups in byte tables; 0 otherwise.
// The rows the host's re-uploads of table t rewrote between cols_version `base` and the current one (CandTable::delta_log),
// on the device (t.union_rows, memoised per base).  Returns 1 with *n_out rows, 0 when no chain of known deltas leads from
// `base` to the current version (or the union is not worth it), < 0 on errors.
static int cols_union(pclean_ctx* ctx, CandTable& t, uint64_t base, uint64_t col_mask, int* n_out) {
  *n_out = 0;
  if (t.union_n >= 0 && t.union_base == base && t.union_head == t.cols_version && t.union_mask == col_mask) {
    *n_out = t.union_n;
    return 1;
  }
  size_t first = t.delta_log.size();
  for (size_t i = 0; i < t.delta_log.size(); ++i)
    if (t.delta_log[i].base == base) {
      first = i;
      break;
    }
  if (first == t.delta_log.size() || t.delta_log.back().next != t.cols_version) return 0;
  std::vector<uint8_t> mark((size_t)std::max(t.n_rows, 1), 0);
  std::vector<int32_t> rows;
  for (size_t i = first; i < t.delta_log.size(); ++i) {
    if (i > first && t.delta_log[i].base != t.delta_log[i - 1].next) return 0;  // (a gap: never by construction)
    const CandTable::DeltaEntry& e = t.delta_log[i];
    for (size_t q = 0; q < e.rows.size(); ++q) {
      const int32_t r = e.rows[q];
      if ((e.masks[q] & col_mask) && r >= 0 && r < t.n_rows && !mark[r]) {  // (rows changed in columns nobody here reads: skipped)
        mark[r] = 1;
        rows.push_back(r);
      }
    }
    if (rows.size() * 8 > (size_t)t.n_rows) return 0;
  }
  std::sort(rows.begin(), rows.end());
  if (!rows.empty()) {
    const size_t bytes = rows.size() * sizeof(int32_t);
    if (t.union_rows.alloc(rows.size() + rows.size() / 4 + 64)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    if (ctx->ustage.grow(bytes + 256)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
    ctx->ustage.rewind();
    void* h = ctx->ustage.take(bytes);
    if (!h) return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
    memcpy(h, rows.data(), bytes);
    HIPCHK(ctx, hipMemcpyAsync(t.union_rows.p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (the staging area is reused by the next union)
  }
  t.union_base = base;
  t.union_mask = col_mask;
  t.union_head = t.cols_version;
  t.union_n = (int32_t)rows.size();
  *n_out = t.union_n;
  return 1;
}

static int try_fast_root(pclean_ctx* ctx, int block_id, int node_id, FastRootDev& fr, bool ev_mode = false) {
  Block& b = ctx->block[block_id];
  if (node_id >= 64) return 0;
  const pclean_node& n = b.nodes[node_id];
  const CandTable& t = ctx->cand[n.table];
  const bool leaf = n.kind == PCLEAN_NODE_LEAF;
  static const bool no_leaf = getenv("PCLEAN_NO_FAST_LEAF") != nullptr;
  if (!t.valid || t.n_rows < 1024 || n.n_terms < 1 || n.n_terms > PCLEAN_MAX_TERMS || (leaf && no_leaf)) return 0;
  if (leaf != t.is_options) return 0;
  int lmax = 0, dmax = 0;
  for (int i = 0; i < n.n_terms; ++i) {
    const pclean_term& tm = b.terms[n.term_begin + i];
    const PairTable& pt = ctx->pair[tm.pair_table];
    if (!pt.valid || tm.dens_kind != PCLEAN_DENS_ADD_TYPOS || pt.elem_bytes != 1) return 0;
    // evidence sets (ev_leaf_block_kernel): ctx terms are only ever scored exactly (by candidate_score), any mode goes
    if (tm.ctx_slot >= 0 && ((!ev_mode && tm.ctx_mode != 0) || !ctx->fn[tm.fn_table].valid)) return 0;
    if (tm.ctx_slot >= 2) return 0;  // the wave kernel's group descriptor carries two context values
    lmax = std::max(lmax, pt.max_lat_len);
    dmax = std::max(dmax, std::max(pt.max_lat_len, pt.max_obs_len));
  }
  if (lmax > 255 || dmax > 255) return 0;
  if (leaf && !ev_mode) {
    // An option list scored against ONE observed string: the integer pre-filter keeps every option within
    // ~10 edits of it (28.5 nats / cost of an edit), i.e. everything when the strings are short (codes, zip
    // codes, phone numbers) — only long strings (names, addresses) are worth the compact tables.
    double best = 0.0;
    for (int i = 0; i < n.n_terms; ++i) {
      const pclean_term& tm = b.terms[n.term_begin + i];
      if (tm.ctx_slot < 0) best = std::max(best, ctx->pair[tm.pair_table].mean_lat_len);
    }
    if (best < 16.0) return 0;
  }
  const int kpad = (t.n_rows + 15) & ~15;
  FastRoot& f = st(ctx)->fast[block_id * 64 + node_id];
  if (f.disabled > 0) {
    --f.disabled;
    return 0;
  }
  if ((int)f.comp.size() != n.n_terms) {
    for (auto& c : f.comp) c.release();
    for (auto& c : f.clen) c.release();
    for (auto& c : f.cblk) c.release();
    f.cblk.assign(n.n_terms, DevBuf<uint8_t>());
    f.comp.assign(n.n_terms, DevBuf<uint8_t>());
    f.clen.assign(n.n_terms, DevBuf<uint8_t>());
    f.ver.assign(n.n_terms, 0);
    f.kpad = kpad;
    f.prior_ver = 0;
  } else if (f.kpad != kpad) {
    // the table grew (or shrank) by a few rows: the byte tables are rebuilt with the new stride INTO the buffers they
    // have (allocated with headroom below) — freeing and allocating a dozen buffers of up to a GB each costs tens of
    // milliseconds on some boxes (the first full iteration was measured at 0.4 s or 0.8 s depending on it)
    f.ver.assign(n.n_terms, 0);
    f.kpad = kpad;
    f.prior_ver = 0;
  }
  auto grow = [](DevBuf<uint8_t>& b, size_t need) -> int {  // 1/8 of headroom whenever it has to grow
    need = std::max<size_t>(need, 16);
    return need <= b.n && b.p ? 0 : b.alloc(need + need / 8);
  };
  // block minima of the compact rows (one byte per 64 candidates): the coarse level of the pre-filter scan
  const int cstride = ((((kpad + 63) >> 6) + 15) & ~15);
  for (int i = 0; i < n.n_terms; ++i) {
    const pclean_term& tm = b.terms[n.term_begin + i];
    const PairTable& pt = ctx->pair[tm.pair_table];
    fr.terms[i] = FastTermDev{};
    fr.terms[i].obs_col = ctx->obs_override ? ctx->obs_override : ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows + ctx->active_begin;
    fr.terms[i].max_typos = tm.max_typos;
    fr.terms[i].ctx_slot = tm.ctx_slot;
    fr.terms[i].pair = (const uint8_t*)pt.d.p;  // ctx terms gather from it; plain terms look up the true distance
    fr.terms[i].lat_len = pt.lat_len.p;         // behind a saturated compact byte
    fr.terms[i].cand_col = t.cols.p + (size_t)tm.cand_col * t.n_rows;
    fr.terms[i].n_lat = pt.n_lat;
    if (tm.ctx_slot >= 0) {  // scored by gathering (few survivors reach it)
      const FnTable& fnt = ctx->fn[tm.fn_table];
      fr.terms[i].fn = fnt.fn.p;
      fr.terms[i].fn_nb = fnt.n_b;
      continue;
    }
    const uint64_t ver = t.cols_version * 1000003ull + pt.version;
    static const bool no_delta = getenv("PCLEAN_NO_COMPACT_DELTA") != nullptr;
    if (f.ver[i] != ver && f.comp[i].p && f.cblk[i].p && !no_delta && t.cols_delta_n >= 0 && t.cols_delta_n * 8 <= t.n_rows &&
        f.ver[i] == t.cols_delta_base * 1000003ull + pt.version) {
      // built from the columns as they were before the last device commit, which wrote a few rows: refresh those rows
      // (and the block minima), not the whole table
      ProfScope psd(ctx, "compact_table_update");
      int rc = PCLEAN_OK;
      if (t.cols_delta_n > 0)
        rc = pclean_update_compact(ctx, pt.d.p, pt.n_obs, pt.n_lat, t.cols.p + (size_t)tm.cand_col * t.n_rows, pt.lat_len.p,
                                     t.cols_delta_rows, t.cols_delta_n, kpad, f.comp[i].p, f.clen[i].p);
      if (rc) return rc;
      if (t.cols_delta_n > 0)
        rc = pclean_update_compact_min(ctx, f.comp[i].p, pt.n_obs, kpad, cstride, t.cols_delta_rows, t.cols_delta_n, f.cblk[i].p);
      if (rc) return rc;
      f.ver[i] = ver;
    }
    // ... or from an older version that a chain of known deltas connects with the current one: the device commit's rows
    // (still on the device), then the union of the rows the host's re-uploads rewrote since (CandTable::delta_log) — the
    // observed class's tables after the sub-batches of the latent classes' sweeps (14 rebuilds = 9 ms per iteration at 1M rows)
    static const bool no_chain = getenv("PCLEAN_NO_DELTA_CHAIN") != nullptr;
    if (f.ver[i] != ver && f.comp[i].p && f.cblk[i].p && !no_delta && !no_chain) {
      CandTable& tw = ctx->cand[n.table];
      uint64_t at = 0;
      bool have = false, via_commit = false;
      if (tw.commit_delta_n >= 0 && f.ver[i] == tw.commit_delta_base * 1000003ull + pt.version) {
        at = tw.commit_delta_next;
        have = via_commit = true;
      } else {
        for (const auto& e : tw.delta_log)
          if (f.ver[i] == e.base * 1000003ull + pt.version) {
            at = e.base;
            have = true;
            break;
          }
      }
      int un = 0;
      bool ok = have;
      if (ok && at != tw.cols_version) {
        uint64_t node_mask = 0;  // the value columns the node's byte tables are built from
        for (int q = 0; q < n.n_terms; ++q)
          if (b.terms[n.term_begin + q].ctx_slot < 0) node_mask |= 1ull << std::min(b.terms[n.term_begin + q].cand_col, 63);
        const int rcu = cols_union(ctx, tw, at, node_mask, &un);
        if (rcu < 0) return rcu;
        ok = rcu == 1;
      }
      const int64_t total = (int64_t)(via_commit ? tw.commit_delta_n : 0) + un;
      static const bool dbg_chain = getenv("PCLEAN_DEBUG_CHAIN") != nullptr;
      if (dbg_chain && i == 0)
        fprintf(stderr, "[chain] block %d node %d table %d (%d rows): have %d via_commit %d (commit rows %d) log %zu entries, union %s %d rows\n",
                block_id, node_id, (int)n.table, tw.n_rows, (int)have, (int)via_commit, tw.commit_delta_n, tw.delta_log.size(),
                ok ? "ok" : "none", un);
      if (ok && total * 8 <= tw.n_rows) {
        ProfScope psd(ctx, "compact_table_update");
        int rc = PCLEAN_OK;
        const int32_t* colp = tw.cols.p + (size_t)tm.cand_col * tw.n_rows;
        if (via_commit && tw.commit_delta_n > 0) {
          rc = pclean_update_compact(ctx, pt.d.p, pt.n_obs, pt.n_lat, colp, pt.lat_len.p, tw.commit_delta_rows, tw.commit_delta_n, kpad,
                                     f.comp[i].p, f.clen[i].p);
          if (!rc) rc = pclean_update_compact_min(ctx, f.comp[i].p, pt.n_obs, kpad, cstride, tw.commit_delta_rows, tw.commit_delta_n, f.cblk[i].p);
          if (rc) return rc;
        }
        if (un > 0) {
          rc = pclean_update_compact(ctx, pt.d.p, pt.n_obs, pt.n_lat, colp, pt.lat_len.p, tw.union_rows.p, un, kpad, f.comp[i].p, f.clen[i].p);
          if (!rc) rc = pclean_update_compact_min(ctx, f.comp[i].p, pt.n_obs, kpad, cstride, tw.union_rows.p, un, f.cblk[i].p);
          if (rc) return rc;
        }
        f.ver[i] = ver;
      }
    }
    if (f.ver[i] != ver || !f.comp[i].p) {
      ProfScope psd(ctx, "compact_table_rebuild");
      if (grow(f.comp[i], (size_t)pt.n_obs * kpad) || grow(f.clen[i], (size_t)kpad))
        return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed (compact tables)");
      int rc = pclean_build_compact(ctx, pt.d.p, pt.n_obs, pt.n_lat, t.cols.p + (size_t)tm.cand_col * t.n_rows,
                                    pt.lat_len.p, t.n_rows, kpad, f.comp[i].p, f.clen[i].p);
      if (rc) return rc;
      if (grow(f.cblk[i], (size_t)pt.n_obs * cstride))
        return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed (compact tables)");
      rc = pclean_build_compact_min(ctx, f.comp[i].p, pt.n_obs, kpad, cstride, f.cblk[i].p);
      if (rc) return rc;
      f.ver[i] = ver;
    }
    fr.terms[i].comp = f.comp[i].p;
    fr.terms[i].clen = f.clen[i].p;
    fr.terms[i].cmin = f.cblk[i].p;
  }
  if ((int)f.zero_row.n < kpad || !f.zero_row.p) {
    if (f.zero_row.alloc((size_t)kpad + 4096)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    HIPCHK(ctx, hipMemsetAsync(f.zero_row.p, 0, f.zero_row.n, ctx->stream));
  }
  if (f.prior_ver != t.version || !f.prior_n.p) {
    if ((!leaf && f.prior_e.alloc(kpad)) || f.prior_n.alloc(kpad) || f.alive.alloc(std::max(kpad >> 4, 1)))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    int rc = pclean_build_priors(ctx, leaf ? nullptr : t.counts.p, t.logc_full.p, t.n_rows, kpad, t.scal[1], t.scal[0],
                                 leaf ? nullptr : f.prior_e.p, f.prior_n.p, f.alive.p);
    if (rc) return rc;
    f.prior_ver = t.version;
    f.logc_max = t.logc_max;  // (maintained with the table: pclean_set_table / pclean_set_options / pclean_commit_device)
  }
  // pre-filter: the three terms with the longest latent strings discriminate best; c_min = the
  // smallest density cost of one edit over every (length, distance) the tables hold
  {
    fr.n_pre = prefilter_terms(ctx, b, n, fr.pre);
    const int stride = ctx->max_d + 1;
    const uint64_t ckey = ((uint64_t)lmax << 40) | ((uint64_t)dmax << 20) | (uint64_t)stride;
    if (f.cmin_key != ckey) {  // ~lmax x dmax host iterations: once per (table shape), not per launch
      double cm = INFINITY;
      for (int L = 1; L <= lmax; ++L)
        for (int d = 1; d <= dmax; ++d) {
          const int r = (L + 4) / 5;
          double l = ctx->h_nb[(size_t)r * stride + d];
          l -= ctx->h_logl[L] * (double)d;
          l -= 1.629048269010741 * (double)d;
          if (l == l) cm = std::min(cm, -l / (double)d);
        }
      f.cmin = cm;
      f.cmin_key = ckey;
    }
    const double cmin = f.cmin;
    if (!(cmin > 1e-6) || !std::isfinite(cmin)) {
      fr.n_pre = 0;  // no usable bound: evaluate every candidate exactly
      fr.inv_c = 0.0;
    } else {
      fr.inv_c = 1.0 / (cmin * (1.0 - 1e-9));
    }
    fr.cstride = cstride;
    fr.prior_max_e = f.logc_max - t.scal[1];
    fr.prior_max_n = f.logc_max - t.scal[0];
  }
  // the observed values row-major, for the launches that look at one row per group (group_gate_kernel, group_desc_kernel);
  // large tables only; not for evidence sets and not under an observation override (leaf caches)
  fr.obs_rm = nullptr;
  static const bool no_obs_rm = getenv("PCLEAN_NO_OBS_ROWMAJOR") != nullptr;
  if (!leaf && !ev_mode && !ctx->obs_override && !no_obs_rm && ctx->n_rows >= 4096) {
    const uint64_t key = ctx->obs_version * 1000003ull + (uint64_t)ctx->n_rows + 1ull;
    if (f.obs_rm_key != key || !f.obs_rm.p) {
      if (f.obs_rm.alloc((size_t)ctx->n_rows * PCLEAN_MAX_TERMS)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
      const int32_t* cols[PCLEAN_MAX_TERMS] = {};
      for (int i = 0; i < n.n_terms; ++i) cols[i] = ctx->obs.p + (size_t)b.terms[n.term_begin + i].obs_col * ctx->n_rows;
      int rc = pclean_build_obs_rowmajor(ctx, cols, n.n_terms, ctx->n_rows, f.obs_rm.p);
      if (rc) return rc;
      f.obs_rm_key = key;
    }
    fr.obs_rm = f.obs_rm.p + (size_t)ctx->active_begin * PCLEAN_MAX_TERMS;
  }
  fr.n_cand = t.n_rows;
  fr.kpad = kpad;
  {
    static const bool no_kscan = getenv("PCLEAN_NO_KSCAN") != nullptr;
    const int n_used = (t.n_used > 0 && t.n_used <= t.n_rows && !no_kscan) ? t.n_used : t.n_rows;
    fr.kscan = std::min(kpad, (n_used + 63) & ~63);
  }
  fr.n_terms = n.n_terms;
  fr.lmax = lmax;
  fr.dstride = dmax + 1;
  fr.is_leaf = leaf ? 1 : 0;
  fr.atd = ctx->atd.p;
  fr.atd_stride = ctx->max_d + 1;
  fr.zero_row = f.zero_row.p;
  fr.alive = f.alive.p;
  fr.prior_e = leaf ? nullptr : f.prior_e.p;
  fr.prior_n = f.prior_n.p;
  fr.logc_m1 = leaf ? nullptr : t.logc_m1.p;
  fr.counts = leaf ? nullptr : t.counts.p;
  memcpy(fr.scal, t.scal, sizeof fr.scal);
  return 1;
}

// Bottom-up evaluation of one plan sub-tree for a list of items
// (process_plan!, proposal_compiler.jl:363-388).  excl = per-item excluded row of
// THIS node's table (device, may be null).  When n_draws > 0 the node also draws.
static int eval_node_lse(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                         uint64_t seed, uint32_t sweep, double* lse_out);
static int make_item_groups(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                            ItemGroups& g, int split_m = 0, bool want_members = true);

// Upper bound of the log-marginal of plan sub-tree `node_id` (gate_new_kernel, enum_kernels.hip): every term
// density of the sub-tree must be a probability mass (<= 1); +inf when it is not (Gaussian terms).
static double subtree_ub(pclean_ctx* ctx, const Block& b, int node_id) {
  const pclean_node& n = b.nodes[node_id];
  if (node_id < (int)b.node_gauss.size() && b.node_gauss[node_id] >= 0) return INFINITY;
  CandTable& t = ctx->cand[n.table];
  if (n.kind == PCLEAN_NODE_LEAF) {
    if (t.h_lse_ver != t.version) {  // log-sum of the option prior, once per upload
      double m = -INFINITY, acc = 0.0;
      for (double v : t.h_logc_full) m = std::max(m, v);
      if (m > -INFINITY)
        for (double v : t.h_logc_full) acc += std::exp(v - m);
      t.h_lse = m > -INFINITY ? m + std::log(acc) + 1e-9 : -INFINITY;
      t.h_lse_ver = t.version;
    }
    return t.h_lse;
  }
  double sum = 0.0;
  for (int c = 0; c < n.n_children; ++c) sum += subtree_ub(ctx, b, b.children[n.child_begin + c]);
  return std::max(0.0, sum);  // log(a + b e^X) <= max(0, X) for a + b <= 1 (CRP prior over rows + new)
}

__global__ void scatter_f64_kernel(int n, const int32_t* list, const double* src, double* dst) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[list[j]] = src[j];
}
// attributes of the items list[j] of a parent list
__global__ void sub_items_kernel(int n, const int32_t* list, const int32_t* p_row, const int32_t* p_ctx,
                                 const int32_t* p_excl, const int32_t* p_ev_lo, const int32_t* p_ev_hi,
                                 const int32_t* p_rng, const int32_t* p_origin, int32_t* row, int32_t* ctxv,
                                 int32_t* excl, int32_t* ev_lo, int32_t* ev_hi, int32_t* rng, int32_t* origin) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int s = list[j];
  row[j] = p_row ? p_row[s] : s;
  excl[j] = p_excl ? p_excl[s] : -1;
  for (int c = 0; c < PCLEAN_MAX_CTX; ++c) ctxv[j * PCLEAN_MAX_CTX + c] = p_ctx ? p_ctx[(size_t)s * PCLEAN_MAX_CTX + c] : 0;
  if (p_ev_lo) {
    ev_lo[j] = p_ev_lo[s];
    ev_hi[j] = p_ev_hi[s];
  }
  if (p_rng) rng[j] = p_rng[s];
  if (origin) origin[j] = p_origin ? p_origin[s] : s;
}

// attributes of the representatives of groups list[j] (all groups when list is null)
__global__ void group_rep_items_kernel(int n, const int32_t* list, const int32_t* grp_off, const int32_t* members,
                                       const int32_t* p_row, const int32_t* p_ctx, const int32_t* p_excl, int32_t* row,
                                       int32_t* ctxv, int32_t* excl) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int g = list ? list[j] : j;
  const int s = members[grp_off[g]];
  row[j] = p_row ? p_row[s] : s;
  excl[j] = p_excl ? p_excl[s] : -1;
  for (int c = 0; c < PCLEAN_MAX_CTX; ++c) ctxv[j * PCLEAN_MAX_CTX + c] = p_ctx ? p_ctx[(size_t)s * PCLEAN_MAX_CTX + c] : 0;
}
// value src[j] of group list[j] -> every member item of the group
__global__ void group_scatter_f64_kernel(int n, const int32_t* list, const int32_t* grp_off, const int32_t* members,
                                         const double* src, double* dst) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int g = list ? list[j] : j;
  const double v = src[j];
  const int hi = grp_off[g + 1];
  for (int mi = grp_off[g]; mi < hi; ++mi) dst[members[mi]] = v;
}

// ... of EVERY group, by member position (uid[mi] - 1 = the group of position mi): no thread walks a large group alone
__global__ void member_scatter_f64_kernel(int n_pos, const int32_t* uid, const int32_t* members, const double* src, double* dst) {
  const int mi = blockIdx.x * blockDim.x + threadIdx.x;
  if (mi < n_pos) dst[members[mi]] = src[uid[mi] - 1];
}

int eval_node(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                     uint64_t seed, uint32_t sweep, int n_draws, double* lse_out, int32_t* draws_out,
                     double* scores_out, const double* snew_override, bool time_it) {
  Block& b = ctx->block[block_id];
  const pclean_node& n = b.nodes[node_id];
  SweepState* s = st(ctx);
  const SweepState::LazyReq lazy_req = s->lazy_req;  // (this evaluation's; the children's evaluations below must not see it)
  s->lazy_req = SweepState::LazyReq();
  s->lazy_out.valid = false;
  // lazy draws (the sweep's last block, see RootExtra): asked for and possible for this list of items
  static const bool no_lazy = getenv("PCLEAN_NO_LAZY_DRAWS") != nullptr;
  const bool lazy_ok = lazy_req.on && !no_lazy && n_draws > 1 && !il.rng_row && !il.ev_lo;
  // groups of a compact-table launch: a large group is cut into pieces of at most ~2 x 256 / n_draws member items (a wave
  // writes n_draws draws per member, see item_head_kernel) — unless the launch leaves lists instead of draws: nothing is
  // written per member then, and the pieces (a third of the Measure slot's groups) were descriptors, hand-outs and list
  // copies for nothing (PCLEAN_LAZY_SPLIT=1: cut them all the same)
  static const bool lazy_split = getenv("PCLEAN_LAZY_SPLIT") != nullptr;
  // (il.n bounds the number of groups: the lists of the launch are sure to fit, see the launch below)
  const bool lazy_sure = lazy_ok && (size_t)il.n * ROOT_LZ_CAP * 12 <= ((size_t)4 << 30);
  const int fast_split_m = (lazy_sure && !lazy_split) ? 0 : std::max(4, 256 / std::max(n_draws, 1));
  NodeDev nd;
  int rc = build_node_dev(ctx, b, node_id, nd);
  if (rc) return rc;
  ChildrenDev ch{};
  ItemsDev it{il.n, 0, il.row, il.ctx, excl, n_draws == 1 ? il.particle : nullptr, s->row_offset + ctx->active_begin,
              nullptr, il.ev_lo, il.ev_hi, il.ev_rows, il.ev_ctx, il.rng_row, nullptr, nullptr, il.draw_is, il.draw_ds,
              nullptr, nullptr};
  if (il.ev_lo) {  // evidence sets: aggregated per original latent item (il.origin)
    rc = ensure_agg(ctx, block_id, node_id, il, &it.agg);
    if (rc) return rc;
    it.ev_item = il.origin;
  }
  FastRootDev fr;
  int fast = 0, fast_ev = 0;
  // a reference slot whose groups are made BEFORE its new-row branch is looked at (below): the gate and the children of
  // the branch then run once per group of identical rows instead of once per row
  ItemGroups g_pre;
  bool fast_tried = false, groups_tried = false;
  const double* pre_score = nullptr;  // exact score of every group's current referent (group_gate_kernel)
  const int32_t* pre_obs = nullptr;   // observed values of every group's row (group_gate_kernel)
  if (n.kind == PCLEAN_NODE_FK && ctx->prior_mode) {
    ch.n = 0;  // the new row's choices are sampled from their priors: the branch carries its CRP term alone
  } else if (n.kind == PCLEAN_NODE_FK) {
    if (snew_override) {
      ch.n = 1;
      ch.arr[0] = snew_override;
      ch.obs_col[0] = nullptr;
    } else {
      if (n.n_children > PCLEAN_MAX_CHILDREN) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "too many children");
      ch.n = n.n_children;
      const CandTable& t = ctx->cand[n.table];
      // cacheable leaves first: exact marginal per unique observed value (versioned cache)
      GateDev gt{};
      gt.n = n.n_children;
      int n_open = 0;
      // (short lists — new-row sampling, tests — are not worth the extra launches and the count read-back)
      const char* gm = getenv("PCLEAN_GATE_MIN");
      const int gate_min = gm ? atoi(gm) : 2048;
      bool gate = excl && !scores_out && il.n >= gate_min && !getenv("PCLEAN_NO_GATE");
      // (rejuvenation sweeps only: the initialisation's batches have no current referent to compare with)
      SweepState::GateStat* gstat = (gate && node_id < 64 && sweep < 0x7ffffff0u && !il.ev_lo && !getenv("PCLEAN_GATE_ALWAYS"))
                                        ? &s->gate_stat[block_id * 64 + node_id] : nullptr;
      if (gstat && gstat->skip > 0) {
        --gstat->skip;
        gate = false;
        gstat = nullptr;
      }
      for (int c = 0; c < n.n_children; ++c) {
        const int cid = b.children[n.child_begin + c];
        const pclean_node& cn = b.nodes[cid];
        if (cn.kind == PCLEAN_NODE_LEAF && cn.cacheable) {
          rc = ensure_leaf_cache(ctx, block_id, cid, &ch.arr[c], &ch.obs_col[c], &ch.n_obs[c]);
          if (rc) return rc;
          gt.cache[c] = ch.arr[c];
          gt.obs_col[c] = ch.obs_col[c];
          gt.n_obs[c] = ch.n_obs[c];
          gt.ub[c] = 0.0;
          if (il.ev_lo) gt.ub[c] = subtree_ub(ctx, b, cid);  // evidence sets: no single observed row to look up
        } else {
          ++n_open;
          gt.cache[c] = nullptr;
          gt.ub[c] = subtree_ub(ctx, b, cid);
          if (!(gt.ub[c] < INFINITY)) gate = false;
        }
        if (il.ev_lo && !(gt.ub[c] < INFINITY)) gate = false;
      }
      // Gate of the new-row branch (gate_new_kernel): items whose current referent scores so well that
      // the new row's fixed-point weight is exactly 0 skip the evaluation of the open children.
      static const bool no_group_gate = getenv("PCLEAN_NO_GROUP_GATE") != nullptr;
      bool pre_grouped = false;
      if (n_open > 0 && n_draws > 0 && excl && !scores_out && !ctx->force_generic && !nd.g.on && !il.rng_row && !il.ev_lo &&
          !no_group_gate) {
        fast = try_fast_root(ctx, block_id, node_id, fr);
        if (fast < 0) return fast;
        fast_tried = true;
        if (fast) {
          rc = make_item_groups(ctx, block_id, node_id, il, excl, g_pre, fast_split_m);
          if (rc) return rc;
          groups_tried = true;
          pre_grouped = g_pre.n_groups > 0;
        }
      }
      if (pre_grouped) {
        // ---- the new-row branch per GROUP: rows with the same (observed tuple, ctx, referent) share the gate's verdict
        // and the children's marginals; the gate scores the current referent through the compact byte rows
        // (group_gate_kernel) and hands the score on to the group descriptors
        const int ng = g_pre.n_groups;
        ItemsDev itg = it;
        itg.n = ng;
        itg.grp_off = g_pre.grp_off;
        itg.members = g_pre.members;
        int32_t* list_g = nullptr;
        unsigned int n_need = (unsigned int)ng;
        if (gate) {
          ProfScope ps(ctx, "gate_new_branch");
          int32_t* flag = scratch<int32_t>(ctx, ng);
          list_g = scratch<int32_t>(ctx, ng);
          double* sc = scratch<double>(ctx, ng);
          int32_t* ow = scratch<int32_t>(ctx, (size_t)ng * PCLEAN_MAX_TERMS);
          if (!flag || !list_g || !sc || !ow) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          if (time_it && s->evg0) (void)hipEventRecord(s->evg0, ctx->stream);
          rc = pclean_launch_group_gate(ctx, fr, itg, gt, flag, sc, ow);
          if (rc) return rc;
          if (time_it && s->evg0) {
            (void)hipEventRecord(s->evg1, ctx->stream);
            s->gate_timed = true;
          }
          unsigned int* need_ctr = fresh_counter(ctx);
          if (!need_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "counter bank: device alloc failed");
          hipLaunchKernelGGL(compact_new_kernel, grid1(ng), dim3(256), 0, ctx->stream, (size_t)ng, flag, 1, need_ctr, list_g,
                             nullptr);
          PCLEAN_READ_COUNT(ctx, need_ctr, &n_need);
          pre_score = sc;
          pre_obs = ow;
          if (gstat) {
            gstat->all_need_run = n_need == (unsigned int)ng ? gstat->all_need_run + 1 : 0;
            if (gstat->all_need_run >= 3) {
              gstat->skip = 16;
              gstat->all_need_run = 2;
            }
          }
          if (n_need == (unsigned int)ng) list_g = nullptr;  // (every group: the identity)
        }
        ItemList sil;
        const int32_t* sexcl = nullptr;
        if (n_need > 0) {
          int32_t* row2 = scratch<int32_t>(ctx, n_need);
          int32_t* ctx2 = scratch<int32_t>(ctx, (size_t)n_need * PCLEAN_MAX_CTX);
          int32_t* excl2 = scratch<int32_t>(ctx, n_need);
          if (!row2 || !ctx2 || !excl2) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          hipLaunchKernelGGL(group_rep_items_kernel, grid1(n_need), dim3(256), 0, ctx->stream, (int)n_need, list_g, g_pre.grp_off,
                             g_pre.members, il.row, il.ctx, excl, row2, ctx2, excl2);
          sil = ItemList{(int)n_need, row2, il.ctx ? ctx2 : nullptr, nullptr, nullptr};
          sexcl = excl2;
        }
        for (int c = 0; c < n.n_children; ++c) {
          const int cid = b.children[n.child_begin + c];
          const pclean_node& cn = b.nodes[cid];
          if (cn.kind == PCLEAN_NODE_LEAF && cn.cacheable) continue;
          double* child_lse = scratch<double>(ctx, il.n);
          if (!child_lse) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          ch.arr[c] = child_lse;
          ch.obs_col[c] = nullptr;
          if (n_need < (unsigned int)ng)  // gated groups: the child's marginal is never looked at with a non-zero weight
            hipLaunchKernelGGL(fill_f64_kernel, grid1(il.n), dim3(256), 0, ctx->stream, child_lse, (size_t)il.n, -__builtin_inf());
          if (n_need == 0) continue;
          const int32_t* child_excl = nullptr;
          if (cn.kind == PCLEAN_NODE_FK) {
            if (cn.parent_fk_col < 0 || cn.parent_fk_col >= t.n_cols)
              return pclean_fail(ctx, PCLEAN_ERR_ARG, "node %d: parent_fk_col out of range", cid);
            int32_t* ce = scratch<int32_t>(ctx, sil.n);
            if (!ce) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
            hipLaunchKernelGGL(derive_excl_kernel, grid1(sil.n), dim3(256), 0, ctx->stream, sil.n, sexcl, t.counts.p,
                               t.cols.p + (size_t)cn.parent_fk_col * t.n_rows, ce);
            child_excl = ce;
          }
          double* dst = scratch<double>(ctx, sil.n);
          if (!dst) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          rc = eval_node_lse(ctx, block_id, cid, sil, child_excl, seed, sweep, dst);
          if (rc) return rc;
          if (!list_g && g_pre.uid && fast_split_m == 0)  // (unsplit groups may be large)
            hipLaunchKernelGGL(member_scatter_f64_kernel, grid1(il.n), dim3(256), 0, ctx->stream, il.n, g_pre.uid, g_pre.members, dst,
                               child_lse);
          else
            hipLaunchKernelGGL(group_scatter_f64_kernel, grid1(sil.n), dim3(256), 0, ctx->stream, sil.n, list_g, g_pre.grp_off,
                               g_pre.members, dst, child_lse);
        }
      } else {
      int32_t* list = nullptr;
      unsigned int n_need = (unsigned int)il.n;
      if (gate && n_open > 0) {
        ProfScope ps(ctx, "gate_new_branch");
        int32_t* flag = scratch<int32_t>(ctx, il.n);
        list = scratch<int32_t>(ctx, il.n);
        if (!flag || !list) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        if (s->counter.alloc(4)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
        rc = pclean_launch_gate(ctx, nd, it, gt, flag);
        if (rc) return rc;
        unsigned int* need_ctr = fresh_counter(ctx);
        if (!need_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "counter bank: device alloc failed");
        hipLaunchKernelGGL(compact_new_kernel, grid1(il.n), dim3(256), 0, ctx->stream, (size_t)il.n, flag, 1, need_ctr, list,
                           nullptr);
        PCLEAN_READ_COUNT(ctx, need_ctr, &n_need);
        if (gstat) {
          gstat->all_need_run = n_need == (unsigned int)il.n ? gstat->all_need_run + 1 : 0;
          if (gstat->all_need_run >= 3) {
            gstat->skip = 16;
            gstat->all_need_run = 2;  // one more useless evaluation after the pause starts the next one
          }
        }
      } else {
        gate = false;
      }
      const bool sub = gate && n_need < (unsigned int)il.n;
      ItemList sil = il;
      const int32_t* sexcl = excl;
      if (sub && n_need > 0) {
        int32_t* row2 = scratch<int32_t>(ctx, n_need);
        int32_t* ctx2 = scratch<int32_t>(ctx, (size_t)n_need * PCLEAN_MAX_CTX);
        int32_t* excl2 = scratch<int32_t>(ctx, n_need);
        int32_t* evl2 = il.ev_lo ? scratch<int32_t>(ctx, n_need) : nullptr;
        int32_t* evh2 = il.ev_lo ? scratch<int32_t>(ctx, n_need) : nullptr;
        int32_t* rng2 = il.rng_row ? scratch<int32_t>(ctx, n_need) : nullptr;
        int32_t* org2 = il.ev_lo ? scratch<int32_t>(ctx, n_need) : nullptr;
        if (!row2 || !ctx2 || !excl2 || (il.ev_lo && (!evl2 || !evh2 || !org2)) || (il.rng_row && !rng2))
          return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        hipLaunchKernelGGL(sub_items_kernel, grid1(n_need), dim3(256), 0, ctx->stream, (int)n_need, list, il.row, il.ctx,
                           excl, il.ev_lo, il.ev_hi, il.rng_row, il.origin, row2, ctx2, excl2, evl2, evh2, rng2, org2);
        sil = ItemList{(int)n_need, row2, il.ctx ? ctx2 : nullptr, nullptr, org2, evl2, evh2, il.ev_rows, il.ev_ctx, rng2};
        sexcl = excl2;
      }
      for (int c = 0; c < n.n_children; ++c) {
        const int cid = b.children[n.child_begin + c];
        const pclean_node& cn = b.nodes[cid];
        if (cn.kind == PCLEAN_NODE_LEAF && cn.cacheable) continue;
        double* child_lse = scratch<double>(ctx, il.n);
        if (!child_lse) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        ch.arr[c] = child_lse;
        ch.obs_col[c] = nullptr;
        if (sub) {  // gated items: the child's marginal is never looked at with a non-zero weight
          hipLaunchKernelGGL(fill_f64_kernel, grid1(il.n), dim3(256), 0, ctx->stream, child_lse, (size_t)il.n,
                             -__builtin_inf());
          if (n_need == 0) continue;
        }
        const int32_t* child_excl = nullptr;
        if (cn.kind == PCLEAN_NODE_FK && sexcl) {
          if (cn.parent_fk_col < 0 || cn.parent_fk_col >= t.n_cols)
            return pclean_fail(ctx, PCLEAN_ERR_ARG, "node %d: parent_fk_col out of range", cid);
          int32_t* ce = scratch<int32_t>(ctx, sil.n);
          if (!ce) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          hipLaunchKernelGGL(derive_excl_kernel, grid1(sil.n), dim3(256), 0, ctx->stream, sil.n, sexcl, t.counts.p,
                             t.cols.p + (size_t)cn.parent_fk_col * t.n_rows, ce);
          child_excl = ce;
        }
        double* dst = child_lse;
        if (sub) {
          dst = scratch<double>(ctx, sil.n);
          if (!dst) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        }
        rc = eval_node_lse(ctx, block_id, cid, sil, child_excl, seed, sweep, dst);
        if (rc) return rc;
        if (sub)
          hipLaunchKernelGGL(scatter_f64_kernel, grid1(sil.n), dim3(256), 0, ctx->stream, sil.n, list, dst, child_lse);
      }
      }  // (item-level gate)
    }
  }
  // cacheable option list: log-marginal and draws from the per-observed-value coarse prefix (leaf_coarse_draw_kernel)
  static const bool no_coarse = getenv("PCLEAN_NO_COARSE_LEAF") != nullptr;
  if (n.kind == PCLEAN_NODE_LEAF && n.cacheable && !il.ev_lo && !scores_out && !ctx->force_generic && !ctx->obs_override &&
      !no_coarse && !nd.g.on && !ctx->prior_mode) {
    const double* cache = nullptr;
    const int32_t* ocol = nullptr;
    int n_obs = 0;
    rc = ensure_leaf_cache(ctx, block_id, node_id, &cache, &ocol, &n_obs);
    if (rc) return rc;
    ProfScope ps(ctx, "option_list_coarse_draw");
    return pclean_launch_leaf_coarse_draw(ctx, nd, it, ocol, n_obs, pclean_leaf_coarse_blocks(nd.n_cand), cache,
                                          b.leaf_m[node_id].p, b.leaf_U[node_id].p, b.leaf_coarse[node_id].p, seed, sweep,
                                          PCLEAN_SITE_NODE(block_id, node_id), n_draws, lse_out, draws_out);
  }
  // A marginal-only evaluation of a few hundred items (the nested slots of a new-row branch that passed the gate: ~250
  // groups per 1M-row sweep) is one launch of the LDS-resident generic kernel with 1024 threads per item; the compact-table
  // path would spend ~10 launches on it (prior rows, alive bits, descriptors, settle, scan, log-sum-exp, overflow re-run) and
  // refresh the node's prior rows after every commit for nothing.  Same results either way (wave == generic is tested).
  static const bool no_small_generic = getenv("PCLEAN_NO_SMALL_GENERIC") != nullptr;
  const bool small_lse = !no_small_generic && n_draws == 0 && il.n <= 1024 && !il.ev_lo &&
                         (size_t)(nd.n_cand + 2) * 8 + (16 + 64) * 8 <= (size_t)160 * 1024;
  if (!fast_tried && !scores_out && !snew_override && !ctx->force_generic && !nd.g.on && !ctx->prior_mode && !small_lse) {
    static const int ev_slot_min_items = getenv("PCLEAN_EV_SLOT_MIN_ITEMS") ? atoi(getenv("PCLEAN_EV_SLOT_MIN_ITEMS")) : 64;
    if (!il.ev_lo)
      fast = try_fast_root(ctx, block_id, node_id, fr);
    // option lists, and since round 6 reference slots (the latent Places re-choosing their County in one batch, 20 ms ->
    // 1.6 ms).  The slot's candidate-compact tables follow the referred table's columns: a sub-batch's host commit re-uploads
    // them with a handful of rows created / collected, and pclean_set_table hands the changed rows on (cols_delta_*) so that
    // the tables are refreshed for those rows alone — rebuilt whole they cost 0.32 ms per call against the 0.29 ms of generic
    // enumeration they replace (Hospital sub-batches of 333 rows), which is what PCLEAN_EV_SLOT_MIN_ITEMS=1024 goes back to
    else if ((n.kind == PCLEAN_NODE_LEAF || (il.n >= ev_slot_min_items && !getenv("PCLEAN_NO_FAST_EV_SLOTS"))) && n_draws <= 1 &&
             !getenv("PCLEAN_NO_FAST_EV"))
      fast_ev = try_fast_root(ctx, block_id, node_id, fr, true);
    if (fast < 0) return fast;
    if (fast_ev < 0) return fast_ev;
    if (fast_ev) {  // needs at least one plain (compact-table) term to filter on
      bool any = false;
      for (int i = 0; i < fr.n_terms; ++i) any |= fr.terms[i].comp != nullptr;
      if (!any) fast_ev = 0;
    }
  }
  // Items with identical score vectors (same observed tuple, ctx and excluded row) share one
  // wavefront / workgroup: scores once, draws per member item.
  ItemGroups ggrp;  // the grouping of the launch below (n_groups == 0: none)
  {
    const int nc = nd.n_cand + (n.kind == PCLEAN_NODE_FK ? 1 : 0);
    const bool lds_kernel = (size_t)((nc + 1) & ~1) * 8 + (16 + 64) * 8 <= 160 * 1024;
    if (n_draws > 0 && !scores_out && !snew_override && !ctx->force_generic && !il.rng_row && !il.ev_lo &&
        (fast || lds_kernel)) {
      ItemGroups g = g_pre;
      // wave kernel: at most ~2 x 256 draws per group (see item_head_kernel)
      if (!groups_tried) {
        rc = make_item_groups(ctx, block_id, node_id, il, excl, g, fast ? fast_split_m : 0);
        if (rc) return rc;
      }
      if (g.n_groups > 0) {
        it.n = g.n_groups;
        it.grp_off = g.grp_off;
        it.members = g.members;
        if (fast && fast_split_m == 0) it.grp_uid = g.uid;  // (unsplit groups: per-member outputs by position)
        ggrp = g;
      }
    }
  }
  const uint32_t site = PCLEAN_SITE_NODE(block_id, node_id);
  if (time_it) {
    pclean_root_stats& rs = ctx->root_stats;
    rs = pclean_root_stats{};
    rs.fast = fast;
    rs.n_items = il.n;
    rs.n_groups = it.n;
    rs.n_cand = nd.n_cand;
    rs.n_terms = n.n_terms;
    rs.n_draws = n_draws;
    rs.pre_scored = pre_score ? 1 : 0;
    if (fast) {
      rs.kpad = fr.kscan;  // (what the scans walk: the candidates below the table's high-water mark, FastRootDev::kscan)
      rs.cstride = std::min(fr.cstride, (((fr.kscan + 63) >> 6) + 15) & ~15);
      rs.n_pre = fr.n_pre;
      for (int p = 0; p < 3; ++p) rs.pre_obs_col[p] = p < fr.n_pre ? b.terms[n.term_begin + fr.pre[p]].obs_col : -1;
    }
  }
  if (!fast && !fast_ev) {
    ProfScope ps(ctx, n.kind == PCLEAN_NODE_FK ? "enum_fk_generic" : "enum_leaf_generic");
    if (time_it) (void)hipEventRecord(s->ev0, ctx->stream);
    double* sc_tmp = nullptr;  // (few items with evidence sets and long lists: the scores one candidate per thread first)
    if (!scores_out) {
      const size_t sc_n = pclean_enum_split_scores(nd, it);
      if (sc_n && !(sc_tmp = scratch<double>(ctx, sc_n))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    }
    rc = pclean_launch_enum(ctx, nd, it, ch, seed, sweep, site, n_draws, lse_out, scores_out, draws_out, sc_tmp);
    if (time_it) (void)hipEventRecord(s->ev1, ctx->stream);
    return rc;
  }
  // compact-table kernels; items whose survivor list overflows are re-run over all candidates
  int32_t* oflag = scratch<int32_t>(ctx, il.n);
  if (!oflag || s->counter.alloc(4)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  // Sync-free re-run: the scan kernel appends the overflowed items to a device list that overflow_lds_kernel
  // (root_wave.hip) consumes with a fixed grid; the count is only read at the end of the call, for the statistics.
  static const bool no_fast_over = getenv("PCLEAN_NO_FAST_OVERFLOW") != nullptr;
  const bool list_mode = fast && !no_fast_over && pclean_overflow_fast_ok(fr, it) && s->over_rec.size() < OVER_SLOTS &&
                         s->over_ctr.p != nullptr;
  // Evidence sets: the scan appends the items it could not settle to a device list as well, and the generic kernel
  // re-runs them as an indirect launch (ItemsDev::sel) of il.n workgroups that retire beyond the list's length — a
  // latent sub-batch evaluates a dozen option lists, each of which used to wait for its count here.
  static const bool no_ev_list = getenv("PCLEAN_NO_EV_LIST") != nullptr;
  const bool ev_list_mode = fast_ev && !no_ev_list && s->over_rec.size() < OVER_SLOTS && s->over_ctr.p != nullptr;
  unsigned int* over_count = (list_mode || ev_list_mode) ? s->over_ctr.p + s->over_rec.size() : s->counter.p + 1;
  int32_t* over_list = nullptr;
  if (list_mode || ev_list_mode) {
    over_list = scratch<int32_t>(ctx, il.n);
    if (!over_list) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    s->over_rec.push_back(SweepState::OverRec{block_id, node_id, il.n, time_it, n.kind == PCLEAN_NODE_LEAF, fast_ev ? 64 : 1024});
  } else {
    { const int rcz = dev_zero(ctx, s->counter.p + 1, sizeof(unsigned int)); if (rcz) return rcz; }
  }
  if (!ev_list_mode)  // (the list stands for the markers there)
    { const int rcz = dev_zero(ctx, oflag, (size_t)il.n * sizeof(int32_t)); if (rcz) return rcz; }  // kernels only set overflow markers
  if (fast) {
    int32_t* desc = scratch<int32_t>(ctx, pclean_fast_desc_words(it.n));
    if (!desc) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    ProfScope ps(ctx, (time_it && block_id == 0) ? "root_scan_block0" : (n.kind == PCLEAN_NODE_FK ? "slot_scan" : "option_scan"));
    if (time_it) (void)hipEventRecord(s->ev0, ctx->stream);
    unsigned int* scan_stats = nullptr;
    if (time_it && s->over_ctr.p) {  // the timed launch (block 0's root): what it read, for bench.py's byte model
      scan_stats = s->over_ctr.p + OVER_SLOTS;
      s->scan_stats_used = true;
    }
    // the work list of the groups the settle kernel leaves pays when it settles most of them; how many it left last time
    // comes back with the call's statistics (apply_over_stats)
    FastRoot& fwl = s->fast[block_id * 64 + node_id];
    unsigned int* wl_stat = nullptr;
    if (!fwl.wl_off && node_id < 64 && s->over_rec.size() < OVER_SLOTS && s->over_ctr.p) {
      wl_stat = s->over_ctr.p + s->over_rec.size();
      s->over_rec.push_back(SweepState::OverRec{block_id, node_id, it.n, false, false, -1});  // min_items -1: a work-list record
    }
    if (fwl.wl_off > 0) --fwl.wl_off;
    RootExtra ex{};
    bool use_ex = false;
    // lazy draws (the sweep's last block): lists instead of n_draws draws per member item
    if (lazy_ok && !it.particle && !it.out_pos &&
        (size_t)it.n * ROOT_LZ_CAP * 12 <= ((size_t)4 << 30)) {
      ex.lz_k = scratch<int32_t>(ctx, (size_t)it.n * ROOT_LZ_CAP);
      ex.lz_p = scratch<uint64_t>(ctx, (size_t)it.n * ROOT_LZ_CAP);
      ex.lz_ns = scratch<int32_t>(ctx, (size_t)it.n);
      if (!ex.lz_k || !ex.lz_p || !ex.lz_ns) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      ex.eager_rows = lazy_req.eager_rows;
      use_ex = true;
    }
    rc = pclean_launch_root_fast(ctx, fr, it, ch, seed, sweep, site, n_draws, lse_out, draws_out, oflag, over_count, desc,
                                 over_list, scan_stats, il.n, pre_score, wl_stat != nullptr, wl_stat, pre_obs, use_ex ? &ex : nullptr);
    if (!rc && ex.lz_ns) {
      SweepState::LazyOut& lo = s->lazy_out;
      lo.valid = true;
      lo.site = site;
      lo.args = LazyDrawArgs{};
      lo.args.n_pos = il.n;
      lo.args.members = it.grp_off ? it.members : nullptr;
      lo.args.uid = it.grp_off ? ggrp.uid : nullptr;
      lo.args.item_row = il.row;
      lo.args.eager_rows = lazy_req.eager_rows;
      lo.args.draws_item = draws_out;
      lo.args.lz_k = ex.lz_k;
      lo.args.lz_p = ex.lz_p;
      lo.args.lz_ns = ex.lz_ns;
      lo.args.g_U = ex.g_U;
      lo.args.row_offset = it.row_offset;
      lo.args.res_new = fr.is_leaf ? fr.n_cand - 1 : PCLEAN_CHOICE_NEW;
    }
    if (time_it) {
      (void)hipEventRecord(s->ev1, ctx->stream);
      s->dbg_desc = desc;
      s->dbg_grp_off = it.grp_off;
      s->dbg_members = it.members;
      s->dbg_oflag = oflag;
      s->dbg_groups = it.n;
      s->dbg_items = il.n;
    }
    if (!rc && list_mode) {
      ProfScope ps2(ctx, "overflow_rerun");
      ItemsDev itf = it;  // the scan's items, ungrouped: list entries index them
      itf.n = il.n;
      itf.grp_off = nullptr;
      itf.members = nullptr;
      const int done = pclean_launch_overflow_fast(ctx, fr, itf, ch, seed, sweep, site, n_draws, lse_out, draws_out, over_list,
                                                   over_count);
      return done < 0 ? done : PCLEAN_OK;
    }
  } else {
    {
      ProfScope ps(ctx, "evidence_option_scan");
      // a handful of latent rows with evidence sets of 10^4 rows and more each: one workgroup per row would walk options x
      // evidence entries alone (12 ms for the one HospitalType row of the 1M-row table) — the sums come from a chip-wide kernel
      uint32_t* dsum = nullptr;
      static const bool no_split = getenv("PCLEAN_NO_EV_SPLIT") != nullptr;
      if (!no_split && il.n <= 64 && s->lat_max_ev >= 16384 && fr.kpad >= 256 && (size_t)il.n * fr.kpad * 4 <= ((size_t)1 << 28)) {
        dsum = scratch<uint32_t>(ctx, (size_t)il.n * fr.kpad);
        if (!dsum) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        const int rcz = dev_zero(ctx, dsum, (size_t)il.n * fr.kpad * sizeof(uint32_t));
        if (rcz) return rcz;
      }
      rc = pclean_launch_ev_leaf(ctx, nd, it, fr, seed, sweep, site, n_draws, lse_out, draws_out, oflag, over_count, over_list,
                                 n.kind == PCLEAN_NODE_FK ? &ch : nullptr, dsum);
    }
    if (!rc && ev_list_mode) {
      ProfScope ps2(ctx, "overflow_rerun");
      ItemsDev itr = it;
      itr.sel = over_list;
      itr.sel_n = over_count;
      double* sc_tmp = nullptr;
      const size_t sc_n = pclean_enum_split_scores(nd, itr);
      if (sc_n && !(sc_tmp = scratch<double>(ctx, sc_n))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      return pclean_launch_enum(ctx, nd, itr, ch, seed, sweep, site, n_draws, lse_out, nullptr, draws_out, sc_tmp);
    }
  }
  if (rc) return rc;
  unsigned int n_over = 0;
  PCLEAN_READ_COUNT(ctx, s->counter.p + 1, &n_over);
  ctx->timing.reserved += (int32_t)n_over;  // items that fell back to the generic kernel
  if (time_it) ctx->root_stats.overflow_items = (int32_t)n_over;
  // short strings / flat posteriors: when a quarter of the items overflow the survivor list the integer pre-filter
  // does not pay for this option list -> its next evaluations go straight to the generic kernel (64, then 128, 256, ...
  // between retries)
  // (latent sub-batches hold a few hundred rows: the same rule from 64 items on — an option list of short strings, where
  // the pre-filter keeps everything, otherwise pays a scan AND a full re-run in every sub-batch)
  if (n.kind == PCLEAN_NODE_LEAF && il.n >= (il.ev_lo ? 64 : 1024) &&
      (il.ev_lo ? n_over >= (unsigned int)il.n : (size_t)n_over * 4 > (size_t)il.n)) {  // (see apply_over_stats)
    FastRoot& f = s->fast[block_id * 64 + node_id];
    f.disabled = f.backoff;
    f.backoff = std::min(f.backoff * 2, 1 << 20);
  }
  if (n_over && getenv("PCLEAN_DEBUG_OVERFLOW"))
    fprintf(stderr, "[pclean] block %d node %d: %u of %d items re-run by the generic kernel\n", block_id, node_id, n_over,
            il.n);
  if (n_over) {
    ProfScope ps(ctx, "overflow_rerun");
    int32_t* list = scratch<int32_t>(ctx, n_over);
    int32_t* row2 = scratch<int32_t>(ctx, n_over);
    int32_t* excl2 = scratch<int32_t>(ctx, n_over);
    int32_t* ctx2 = scratch<int32_t>(ctx, (size_t)n_over * PCLEAN_MAX_CTX);
    int32_t* part2 = scratch<int32_t>(ctx, n_over);
    int32_t* evl2 = il.ev_lo ? scratch<int32_t>(ctx, n_over) : nullptr;
    int32_t* evh2 = il.ev_lo ? scratch<int32_t>(ctx, n_over) : nullptr;
    int32_t* org2 = il.ev_lo ? scratch<int32_t>(ctx, n_over) : nullptr;
    int32_t* rng2 = il.rng_row ? scratch<int32_t>(ctx, n_over) : nullptr;
    if (!list || !row2 || !excl2 || !ctx2 || !part2 || (il.ev_lo && (!evl2 || !evh2 || !org2)) || (il.rng_row && !rng2))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    { const int rcz = dev_zero(ctx, s->counter.p + 1, sizeof(unsigned int)); if (rcz) return rcz; }
    hipLaunchKernelGGL(compact_new_kernel, grid1(il.n), dim3(256), 0, ctx->stream, (size_t)il.n, oflag, 1,
                       s->counter.p + 1, list, nullptr);
    hipLaunchKernelGGL(sub_items_kernel, grid1(n_over), dim3(256), 0, ctx->stream, (int)n_over, list, il.row, il.ctx, excl,
                       il.ev_lo, il.ev_hi, il.rng_row, il.origin, row2, ctx2, excl2, evl2, evh2, rng2, org2);
    if (it.particle)
      hipLaunchKernelGGL(gather_i32_kernel, grid1(n_over), dim3(256), 0, ctx->stream, (int)n_over, list, it.particle, part2);
    ItemsDev it2{(int)n_over, 0, row2, il.ctx ? ctx2 : nullptr, excl ? excl2 : nullptr, it.particle ? part2 : nullptr,
                 s->row_offset + ctx->active_begin, list, evl2, evh2, il.ev_rows, il.ev_ctx, rng2, nullptr, nullptr,
                 il.draw_is, il.draw_ds, it.agg, org2};
    // compact-row exact scoring of every candidate (root_wave.hip: overflow_lds_kernel); evidence sets, groups and
    // tables beyond one workgroup's LDS go through the generic kernel
    int done = 0;
    static const bool no_fast_over = getenv("PCLEAN_NO_FAST_OVERFLOW") != nullptr;
    if (fast && !no_fast_over) {
      done = pclean_launch_overflow_fast(ctx, fr, it2, ch, seed, sweep, site, n_draws, lse_out, draws_out, nullptr, nullptr);
      if (done < 0) return done;
    }
    if (!done) {
      double* sc_tmp = nullptr;
      const size_t sc_n = pclean_enum_split_scores(nd, it2);
      if (sc_n && !(sc_tmp = scratch<double>(ctx, sc_n))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      rc = pclean_launch_enum(ctx, nd, it2, ch, seed, sweep, site, n_draws, lse_out, nullptr, draws_out, sc_tmp);
    }
  }
  return rc;
}

// rocPRIM's radix sort switches to a merge sort for inputs of up to 2^20 keys (radix_sort_config's MergeSortLimit);
// for (32-bit key, 32-bit value) pairs of a 1M-row sweep its Onesweep path is ~3x faster (measured: 165 -> ~55 us).
using pclean_sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
template <typename KeyT>
static hipError_t pclean_sort_pairs(void* tmp, size_t& tmp_bytes, KeyT* key, KeyT* key_s, int32_t* val, int32_t* val_s, int n,
                                    int key_bits, hipStream_t stream) {
  static const bool merge = getenv("PCLEAN_SORT_MERGE") != nullptr;
  if (merge || n < 100000)
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key, key_s, val, val_s, n, 0, key_bits, stream);
  return rocprim::radix_sort_pairs<pclean_sort_config>(tmp, tmp_bytes, key, key_s, val, val_s, (size_t)n, 0u, (unsigned)key_bits,
                                                       stream);
}

// ---- item de-duplication ------------------------------------------------------------------------
// The log marginal of a plan sub-tree is a pure function of (observed values of the sub-tree's
// terms, ctx values, excluded row).  On a 1M-row table most rows share that tuple with other rows
// (same hospital, same dirty cells), so the sub-tree is evaluated once per distinct tuple and the
// result scattered back.  Distinct tuples are found by sorting a 64-bit hash and comparing adjacent
// tuples exactly (a hash collision can only split a group, never merge two).
struct KeyColsDev {
  int32_t n_cols, use_ctx;
  const int32_t* col[32];
  int32_t n_pre, pad;         // observed columns of the scan kernel's pre-filter terms (prefilter_terms): groups that
  const int32_t* pre_col[3];  // share them are made adjacent so that a wave can reuse its survivor list
  // static per-row ids (ensure_tuple_ids): dense id of the row's tuple of key columns (two rows hold the same observed
  // tuple iff their ids are equal) and a hash of its pre-filter values — the data never changes, so the exact
  // comparison of the columns is paid once, not in every sweep.  Null: hash / compare the columns themselves.
  const int32_t* tuple_id;
  const uint32_t* pre_hash;
};

__device__ __forceinline__ uint64_t mix64(uint64_t h, uint32_t v) {
  h ^= (uint64_t)v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  return h ^ (h >> 32);
}
template <typename KeyT>
__global__ void item_key_kernel(int n, KeyColsDev kc, const int32_t* row, const int32_t* ctxv, const int32_t* excl,
                                int low_bits, KeyT* key, int32_t* idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = row ? row[i] : i;
  uint64_t h = 0x2545f4914f6cdd1dull;
  if (kc.tuple_id)
    h = mix64(h, (uint32_t)kc.tuple_id[r]);
  else
    for (int c = 0; c < kc.n_cols; ++c) h = mix64(h, (uint32_t)kc.col[c][r]);
  if (kc.use_ctx && ctxv)
    for (int s = 0; s < PCLEAN_MAX_CTX; ++s) h = mix64(h, (uint32_t)ctxv[(size_t)i * PCLEAN_MAX_CTX + s]);
  // Sort order = (referent, hash of the pre-filter observed values, hash of the whole tuple): groups of one
  // referent end up adjacent (their waves run back to back and re-read the same byte rows from L2), and within
  // a referent the groups that share the pre-filter rows are adjacent too (root_wave.hip reuses the scan).
  // Short keys = few radix passes: low_bits hash bits below the referent id (half of them from the pre-filter
  // values), 32 hash bits without a referent; a collision of two different tuples can only split a group
  // (item_head_kernel compares exactly).  With a referent the whole key fits 32 bits whenever the table has fewer
  // than 2^(32 - 16) rows (make_item_groups picks KeyT): half the sort's memory traffic.
  uint64_t hp = 0x9e3779b97f4a7c15ull;
  if (kc.pre_hash)
    hp = (uint64_t)kc.pre_hash[r] << 32;
  else
    for (int c = 0; c < kc.n_pre; ++c) hp = mix64(hp, (uint32_t)kc.pre_col[c][r]);
  if (excl) {
    h = mix64(h, (uint32_t)excl[i]);
    const int hb = low_bits >> 1, lb = low_bits - hb;  // pre-filter hash bits, tuple hash bits
    const uint64_t low = kc.n_pre > 0 ? (((hp >> (64 - hb)) << lb) | (h >> (64 - lb))) : (h >> (64 - low_bits));
    h = ((uint64_t)(uint32_t)(excl[i] + 1) << low_bits) | low;
  } else {
    h = kc.n_pre > 0 ? (((hp >> 48) << 16) | (h >> 48)) : (h >> 32);
  }
  key[i] = (KeyT)h;
  idx[i] = i;
}
// split_m > 0: a run of more than split_m items with one key is cut at every multiple of split_m (pieces of
// split_m .. 2 split_m - 1 items): the scan kernel serialises the draws of a group in ONE wave, and its hand-out
// of work balances at group granularity (the pieces are adjacent: the wave reuses the previous piece's scores).
template <typename KeyT>
__global__ void item_head_kernel(int n, KeyColsDev kc, const int32_t* row, const int32_t* ctxv, const int32_t* excl,
                                 const KeyT* key, const int32_t* idx, int32_t* head, int split_m) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int h = 1;
  if (split_m > 0 && j >= split_m && (j % split_m) == 0 && key[j] == key[j - split_m]) {
    head[j] = 1;
    return;
  }
  if (j > 0 && key[j] == key[j - 1]) {
    const int a = idx[j], b = idx[j - 1];
    const int ra = row ? row[a] : a, rb = row ? row[b] : b;
    bool same = true;
    if (kc.tuple_id)
      same = kc.tuple_id[ra] == kc.tuple_id[rb];
    else
      for (int c = 0; c < kc.n_cols && same; ++c) same = kc.col[c][ra] == kc.col[c][rb];
    if (same && kc.use_ctx && ctxv)
      for (int s = 0; s < PCLEAN_MAX_CTX && same; ++s)
        same = ctxv[(size_t)a * PCLEAN_MAX_CTX + s] == ctxv[(size_t)b * PCLEAN_MAX_CTX + s];
    if (same && excl) same = excl[a] == excl[b];
    h = same ? 0 : 1;
  }
  head[j] = h;
}
__global__ void item_unique_kernel(int n, const int32_t* idx, const int32_t* head, const int32_t* uid_incl,
                                   const int32_t* row, const int32_t* ctxv, const int32_t* excl, int32_t* uid_of_item,
                                   int32_t* row2, int32_t* ctx2, int32_t* excl2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int i = idx[j], u = uid_incl[j] - 1;
  uid_of_item[i] = u;
  if (head[j]) {
    row2[u] = row ? row[i] : i;
    if (ctxv)
      for (int s = 0; s < PCLEAN_MAX_CTX; ++s) ctx2[(size_t)u * PCLEAN_MAX_CTX + s] = ctxv[(size_t)i * PCLEAN_MAX_CTX + s];
    if (excl) excl2[u] = excl[i];
  }
}
__global__ void gather_f64_kernel(int n, const int32_t* src_of, const double* src, double* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[src_of[i]];
}

// observed columns / ctx use of a plan sub-tree; false when the sub-tree cannot be keyed (numeric terms)
static bool subtree_key(pclean_ctx* ctx, const Block& b, int node_id, std::set<int>& cols, bool& use_ctx) {
  const pclean_node& n = b.nodes[node_id];
  if (node_id < (int)b.node_gauss.size() && b.node_gauss[node_id] >= 0) return false;
  for (int i = 0; i < n.n_terms; ++i) {
    const pclean_term& tm = b.terms[n.term_begin + i];
    if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP) return false;
    cols.insert(tm.obs_col);
    if (tm.ctx_slot >= 0) use_ctx = true;
  }
  for (int c = 0; c < n.n_children; ++c)
    if (!subtree_key(ctx, b, b.children[n.child_begin + c], cols, use_ctx)) return false;
  return true;
}

__global__ void group_offsets_kernel(int n, const int32_t* head, const int32_t* uid, int32_t* grp_off) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n) return;
  if (j == n)
    grp_off[uid[n - 1]] = n;
  else if (head[j])
    grp_off[uid[j] - 1] = j;
}

// ---- static per-row tuple ids --------------------------------------------------------------------------------------
__global__ void tuple_hash_kernel(int n, KeyColsDev kc, uint64_t* key, int32_t* idx, uint32_t* pre_hash) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t h = 0x2545f4914f6cdd1dull;
  for (int c = 0; c < kc.n_cols; ++c) h = mix64(h, (uint32_t)kc.col[c][i]);
  key[i] = h;
  idx[i] = i;
  uint64_t hp = 0x9e3779b97f4a7c15ull;
  for (int c = 0; c < kc.n_pre; ++c) hp = mix64(hp, (uint32_t)kc.pre_col[c][i]);
  pre_hash[i] = (uint32_t)(hp >> 32);
}
__global__ void tuple_id_scatter_kernel(int n, const int32_t* idx, const int32_t* uid_incl, int32_t* tuple_id) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) tuple_id[idx[j]] = uid_incl[j] - 1;
}
// tuple ids / pre-filter hashes of ALL loaded rows for the key columns of (block, node); built once per loaded table
static int ensure_tuple_ids(pclean_ctx* ctx, int block_id, int node_id, const std::set<int>& cols, const int32_t pre_cols[3],
                            int n_pre, const int32_t** tuple_id, const uint32_t** pre_hash) {
  SweepState* s = st(ctx);
  SweepState::TupleIds& t = s->tuple_ids[block_id * 64 + node_id];
  uint64_t sig = (uint64_t)ctx->n_rows * 0x9e3779b97f4a7c15ull + (uint64_t)(uintptr_t)ctx->obs.p + ctx->obs_version * 0xd6e8feb86659fd93ull;
  for (int c : cols) sig = sig * 1000003ull + (uint64_t)(c + 1);
  for (int q = 0; q < n_pre; ++q) sig = sig * 1000003ull + (uint64_t)(pre_cols[q] + 7);
  if (t.sig != sig || !t.id.p) {
    const int n = ctx->n_rows;
    if (t.id.alloc(std::max(n, 1)) || t.pre.alloc(std::max(n, 1))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    KeyColsDev kc{};
    for (int c : cols) kc.col[kc.n_cols++] = ctx->obs.p + (size_t)c * ctx->n_rows;
    kc.n_pre = n_pre;
    for (int q = 0; q < n_pre; ++q) kc.pre_col[q] = ctx->obs.p + (size_t)pre_cols[q] * ctx->n_rows;
    DevBuf<uint64_t> key, key_s;
    DevBuf<int32_t> idx, idx_s, head, uid;
    DevBuf<unsigned char> tmp;
    if (key.alloc(n) || key_s.alloc(n) || idx.alloc(n) || idx_s.alloc(n) || head.alloc(n) || uid.alloc(n))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    hipLaunchKernelGGL(tuple_hash_kernel, grid1(n), dim3(256), 0, ctx->stream, n, kc, key.p, idx.p, t.pre.p);
    size_t tmp_sort = 0, tmp_scan = 0;
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, key.p, key_s.p, idx.p, idx_s.p, n, 0, 64, ctx->stream));
    HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan, head.p, uid.p, n, ctx->stream));
    if (tmp.alloc(std::max(tmp_sort, tmp_scan))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_sort, key.p, key_s.p, idx.p, idx_s.p, n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(item_head_kernel<uint64_t>, grid1(n), dim3(256), 0, ctx->stream, n, kc, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, key_s.p, idx_s.p, head.p, 0);
    HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(tmp.p, tmp_scan, head.p, uid.p, n, ctx->stream));
    hipLaunchKernelGGL(tuple_id_scatter_kernel, grid1(n), dim3(256), 0, ctx->stream, n, idx_s.p, uid.p, t.id.p);
    PCLEAN_SYNC(ctx);
    key.release(); key_s.release(); idx.release(); idx_s.release(); head.release(); uid.release(); tmp.release();
    t.sig = sig;
  }
  *tuple_id = t.id.p + ctx->active_begin;
  *pre_hash = t.pre.p + ctx->active_begin;
  return PCLEAN_OK;
}

// ---- grouping through a hash table ----------------------------------------------------------------------------------
// The groups only have to be RIGHT (members of a group share the whole key: a group may be split, never merged), not sorted:
// every result is a function of the item alone.  A radix sort of 10^6 (key, item) pairs is ~16 dispatches and ~0.2 ms, and
// a sweep made three of them; here every item claims or joins a slot of an open-addressing table (the slot holds a
// representative ITEM, the key comparison reads that item's attributes: no half-written keys) and one scan over the slots
// numbers the groups.  Callers that only need "which group is item i in" (eval_node_lse_core) stop there; the root scans
// also need the members of every group side by side: an item takes its position in the group from the slot's counter
// (the lanes of a wavefront that landed in the same slot share ONE atomic) and the same scan turns (pieces, members) per
// slot into offsets.  The slot order keeps what the scan kernels like about the sorted order where it is cheap to keep:
// the table index is hash(referent, pre-filter values) * HG_BUCKET + (hash(tuple, ctx) mod HG_BUCKET), so groups that
// share the referent and the pre-filter rows sit next to each other (root_wave.hip reuses the previous group's survivor
// list).  PCLEAN_SORT_GROUPS=1: the radix-sort path below.
#define HG_WBITS 12   // slots per window: 4096
#define HG_INWIN 32   // probes inside the window before a key goes looking elsewhere
struct HashGroupDev {
  int32_t* rep;        // [cap] 0: empty, else representative item + 1
  unsigned int* cnt;   // [cap] members so far (null: group ids only)
  uint32_t mask;       // cap - 1
  int32_t split_m;
  uint32_t wbits;      // log2 of the window size (at most the table)
};
__device__ __forceinline__ bool hg_same_key(const KeyColsDev& kc, const int32_t* row, const int32_t* ctxv, const int32_t* excl,
                                            int a, int b) {
  const int ra = row ? row[a] : a, rb = row ? row[b] : b;
  bool same = true;
  if (kc.tuple_id)
    same = kc.tuple_id[ra] == kc.tuple_id[rb];
  else
    for (int c = 0; c < kc.n_cols && same; ++c) same = kc.col[c][ra] == kc.col[c][rb];
  if (same && kc.use_ctx && ctxv)
    for (int s = 0; s < PCLEAN_MAX_CTX && same; ++s)
      same = ctxv[(size_t)a * PCLEAN_MAX_CTX + s] == ctxv[(size_t)b * PCLEAN_MAX_CTX + s];
  if (same && excl) same = excl[a] == excl[b];
  return same;
}
// probe sequence of a key: HG_INWIN slots of its family's window (the window is chosen by hash(referent, pre-filter values),
// the start inside it by hash(tuple, ctx): a family's groups sit in one window, spread evenly — a family of thousands of
// groups neither piles up behind one bucket nor loses its neighbourhood), then linear probing from a uniform position
__device__ __forceinline__ uint32_t hg_slot(uint32_t base, uint32_t pos0, uint32_t alt, uint32_t wmask, uint32_t mask, int t) {
  return t < HG_INWIN ? base + ((pos0 + (uint32_t)t) & wmask) : (alt + (uint32_t)(t - HG_INWIN)) & mask;
}
// classes of the lanes with in == true by the value of `key`: leader = first lane of this lane's class, rank = lanes of the
// class before this one, size = lanes in the class (wave-uniform loop: one pass per distinct key, ballots only)
__device__ __forceinline__ void hg_classes(bool in, uint32_t key, int lane, int& leader, int& rank, int& size) {
  leader = lane;
  rank = 0;
  size = 1;
  for (unsigned long long pend = __ballot(in); pend;) {
    const int ld = __builtin_ctzll(pend);
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, ld);
    const unsigned long long cls = __ballot(in && key == k) & pend;
    if ((cls >> lane) & 1ull) {
      leader = ld;
      rank = __popcll(cls & ((1ull << lane) - 1ull));
      size = __popcll(cls);
    }
    pend &= ~cls;
  }
}
__global__ __launch_bounds__(256) void hg_insert_kernel(int n, KeyColsDev kc, const int32_t* __restrict__ row,
                                                        const int32_t* __restrict__ ctxv, const int32_t* __restrict__ excl,
                                                        HashGroupDev hg, int32_t* __restrict__ slot_of,
                                                        int32_t* __restrict__ pos_of) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool on = i < n;
  uint32_t base = 0, pos0 = 0, alt = 0;
  if (on) {
    const int r = row ? row[i] : i;
    uint64_t h = 0x2545f4914f6cdd1dull;
    if (kc.tuple_id)
      h = mix64(h, (uint32_t)kc.tuple_id[r]);
    else
      for (int c = 0; c < kc.n_cols; ++c) h = mix64(h, (uint32_t)kc.col[c][r]);
    if (kc.use_ctx && ctxv)
      for (int s = 0; s < PCLEAN_MAX_CTX; ++s) h = mix64(h, (uint32_t)ctxv[(size_t)i * PCLEAN_MAX_CTX + s]);
    if (excl) h = mix64(h, (uint32_t)excl[i]);
    uint64_t hp = 0x9e3779b97f4a7c15ull;
    if (kc.pre_hash)
      hp = mix64(hp, kc.pre_hash[r]);
    else
      for (int c = 0; c < kc.n_pre; ++c) hp = mix64(hp, (uint32_t)kc.pre_col[c][r]);
    if (excl) hp = mix64(hp, (uint32_t)excl[i]);
    if (kc.n_pre == 0 && !excl) hp = h;  // nothing to keep adjacent: plain hashing
    base = ((uint32_t)(hp >> 24) & (hg.mask >> hg.wbits)) << hg.wbits;
    pos0 = (uint32_t)(h >> 40);
    alt = (uint32_t)(h >> 8) & hg.mask;
  }
  const uint32_t wmask = (1u << hg.wbits) - 1u;
  // Claim or join a slot, in wave-uniform rounds.  The lanes of the wavefront that look at the SAME slot in a round form a
  // class (hg_classes); its first lane reads the slot, tries the compare-and-swap when it is empty, and hands what the slot
  // holds to the others by shuffle: 10^5 rows of one popular key cost one load (and at most one atomic) per wavefront and
  // round instead of 10^5 requests queueing on one address.
  bool active = on;
  int t = 0;
  uint32_t slot = hg_slot(base, pos0, alt, wmask, hg.mask, 0);
  int leader = lane, rank = 0, size = 1;
  while (__ballot(active)) {
    hg_classes(active, slot, lane, leader, rank, size);
    int val = 0;
    if (active && leader == lane) {
      val = __hip_atomic_load(&hg.rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (val == 0) {
        val = atomicCAS(&hg.rep[slot], 0, i + 1);
        if (val == 0) val = i + 1;  // claimed: this item represents the group
      }
    }
    val = __shfl(val, leader, 64);
    if (active) {
      if (val - 1 == i || hg_same_key(kc, row, ctxv, excl, i, val - 1)) {
        active = false;
      } else {
        ++t;
        slot = hg_slot(base, pos0, alt, wmask, hg.mask, t);
      }
    }
  }
  if (on) slot_of[i] = (int32_t)slot;
  if (!hg.cnt) return;
  // position within the group: the lanes that ended in the same slot take consecutive positions from ONE atomic of their
  // first lane — all classes' atomics in flight together
  hg_classes(on, slot, lane, leader, rank, size);
  unsigned int bse = 0;
  if (on && leader == lane) bse = atomicAdd(&hg.cnt[slot], (unsigned int)size);
  bse = (unsigned int)__shfl((int)bse, leader, 64);
  if (on) pos_of[i] = (int)bse + rank;
}
// per slot: groups only: 1 per claimed slot; with members: (groups the slot contributes) << 32 | members — a group of more
// than 2 split_m - 1 members is cut into pieces of split_m (the last piece takes the remainder: split_m .. 2 split_m - 1
// members), see item_head_kernel
struct HgClaimed {
  __host__ __device__ uint64_t operator()(const int32_t& rep) const { return rep != 0 ? 1ull : 0ull; }
};
struct HgPacked {
  int32_t split_m;
  __host__ __device__ uint64_t operator()(const unsigned int& c) const {
    const unsigned int pieces = c == 0u ? 0u : (split_m > 0 ? (c / (unsigned int)split_m > 1u ? c / (unsigned int)split_m : 1u) : 1u);
    return ((uint64_t)pieces << 32) | c;
  }
};
__global__ void hg_fill_kernel(int n, uint32_t cap, const uint64_t* __restrict__ incl, const unsigned int* __restrict__ cnt,
                               const int32_t* __restrict__ slot_of, const int32_t* __restrict__ pos_of, int split_m,
                               int32_t* __restrict__ members, int32_t* __restrict__ head, int32_t* __restrict__ uid,
                               int32_t* __restrict__ grp_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) grp_off[(uint32_t)(incl[cap - 1] >> 32)] = n;
  if (i >= n) return;
  const int sl = slot_of[i], pos = pos_of[i];
  const uint64_t pk = HgPacked{split_m}(cnt[sl]), ex = incl[sl] - pk;
  const int pieces = (int)(pk >> 32), g0 = (int)(ex >> 32), off0 = (int)(uint32_t)ex;
  const int piece = split_m > 0 ? min(pos / split_m, pieces - 1) : 0;
  const bool first = split_m > 0 ? pos == piece * split_m : pos == 0;
  const int at = off0 + pos;
  members[at] = i;
  uid[at] = g0 + piece + 1;
  if (head) head[at] = first ? 1 : 0;
  if (first) grp_off[g0 + piece] = at;
}
// group ids only: uid_of_item[i], and the attributes of every group's representative
__global__ void hg_unique_kernel(int n, const uint64_t* __restrict__ incl, const int32_t* __restrict__ rep,
                                 const int32_t* __restrict__ slot_of, const int32_t* __restrict__ row,
                                 const int32_t* __restrict__ ctxv, const int32_t* __restrict__ excl,
                                 int32_t* __restrict__ uid_of_item, int32_t* __restrict__ row2, int32_t* __restrict__ ctx2,
                                 int32_t* __restrict__ excl2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int sl = slot_of[i];
  const int u = (int)incl[sl] - 1;
  uid_of_item[i] = u;
  if (rep[sl] - 1 == i) {
    row2[u] = row ? row[i] : i;
    if (ctxv)
      for (int s = 0; s < PCLEAN_MAX_CTX; ++s) ctx2[(size_t)u * PCLEAN_MAX_CTX + s] = ctxv[(size_t)i * PCLEAN_MAX_CTX + s];
    if (excl) excl2[u] = excl[i];
  }
}
// members == false: g.n_groups and the state hg_unique_kernel needs (g.hg_incl, g.hg_rep, g.hg_slot_of) only
static int make_item_groups_hash(pclean_ctx* ctx, const ItemList& il, const int32_t* excl, const KeyColsDev& kc, int split_m,
                                 bool want_members, ItemGroups& g) {
  const int n = il.n;
  uint32_t cap = 1024;
  while (cap < 2u * (uint32_t)n) cap <<= 1;
  int32_t* rep = scratch<int32_t>(ctx, (size_t)cap * 2);  // rep[cap], then cnt[cap]: one memset
  uint64_t* incl = scratch<uint64_t>(ctx, cap);
  int32_t* slot_of = scratch<int32_t>(ctx, n);
  int32_t* pos_of = want_members ? scratch<int32_t>(ctx, n) : nullptr;
  size_t tmp_scan = 0, tmp_scan2 = 0;
  hipcub::TransformInputIterator<uint64_t, HgClaimed, const int32_t*> in_claimed(rep, HgClaimed());
  hipcub::TransformInputIterator<uint64_t, HgPacked, const unsigned int*> in_packed((const unsigned int*)(rep + cap), HgPacked{split_m});
  HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan, in_claimed, incl, (int)cap, ctx->stream));
  HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan2, in_packed, incl, (int)cap, ctx->stream));
  unsigned char* tmp = scratch<unsigned char>(ctx, std::max<size_t>(std::max(tmp_scan, tmp_scan2), 16));
  if (!rep || !incl || !slot_of || (want_members && !pos_of) || !tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  uint32_t wbits = HG_WBITS;
  while ((1u << wbits) > cap) --wbits;
  HashGroupDev hg{rep, want_members ? (unsigned int*)(rep + cap) : nullptr, cap - 1, split_m, wbits};
  { const int rcz = dev_zero(ctx, rep, (size_t)cap * (want_members ? 2 : 1) * sizeof(int32_t)); if (rcz) return rcz; }
  hipLaunchKernelGGL(hg_insert_kernel, grid1(n), dim3(256), 0, ctx->stream, n, kc, il.row, il.ctx, excl, hg, slot_of, pos_of);
  int32_t n_unique = 0;
  if (want_members) {
    HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(tmp, tmp_scan2, in_packed, incl, (int)cap, ctx->stream));
    // (little-endian: the high word of the last inclusive sum = number of groups)
    PCLEAN_READ_COUNT(ctx, reinterpret_cast<const uint32_t*>(incl + (cap - 1)) + 1, &n_unique);
  } else {
    HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(tmp, tmp_scan, in_claimed, incl, (int)cap, ctx->stream));
    PCLEAN_READ_COUNT(ctx, reinterpret_cast<const uint32_t*>(incl + (cap - 1)), &n_unique);
  }
  if (n_unique <= 0 || (double)n_unique > 0.75 * n) return PCLEAN_OK;  // not worth the indirection
  g.n_groups = n_unique;
  g.hg_incl = incl;
  g.hg_rep = rep;
  g.hg_slot_of = slot_of;
  if (!want_members) return PCLEAN_OK;
  int32_t* members = scratch<int32_t>(ctx, n);
  int32_t* uid = scratch<int32_t>(ctx, n);
  int32_t* grp_off = scratch<int32_t>(ctx, (size_t)n_unique + 1);
  if (!members || !uid || !grp_off) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  // (no `head` flags: their only reader is item_unique_kernel, which the hash path never reaches — hg_unique_kernel takes the
  // representatives from the table itself; a third scattered 4-byte write per item for nothing)
  hipLaunchKernelGGL(hg_fill_kernel, grid1(n), dim3(256), 0, ctx->stream, n, cap, incl, hg.cnt, slot_of, pos_of, split_m, members,
                     (int32_t*)nullptr, uid, grp_off);
  g.grp_off = grp_off;
  g.members = members;
  g.head = nullptr;
  g.uid = uid;
  return PCLEAN_OK;
}

// Groups the items of `il` by (observed values of the sub-tree of node_id, ctx, excl).  g.n_groups == 0
// when the sub-tree cannot be keyed, the list is small, or fewer than a quarter of the items are duplicates.
static int make_item_groups(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                            ItemGroups& g, int split_m, bool want_members) {
  Block& b = ctx->block[block_id];
  std::set<int> cols;
  bool use_ctx = false;
  static const bool disabled = getenv("PCLEAN_NO_DEDUP") != nullptr;
  g = ItemGroups();
  if (ctx->prior_mode) {  // the prior vector of a slot depends on the excluded row alone
    if (disabled || il.n < 4096 || il.ev_lo) return PCLEAN_OK;
  } else if (disabled || il.n < 4096 || il.ev_lo || !subtree_key(ctx, b, node_id, cols, use_ctx) || cols.size() > 32) {
    return PCLEAN_OK;
  }
  const int n = il.n;
  KeyColsDev kc{};
  kc.use_ctx = use_ctx ? 1 : 0;
  for (int c : cols) {
    if (c < 0 || c >= ctx->n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term column out of range");
    kc.col[kc.n_cols++] = ctx->obs.p + (size_t)c * ctx->n_rows + ctx->active_begin;
  }
  {
    const pclean_node& nn = b.nodes[node_id];
    int32_t pre[3];
    kc.n_pre = (nn.n_terms <= PCLEAN_MAX_TERMS && !ctx->prior_mode) ? prefilter_terms(ctx, b, nn, pre) : 0;
    for (int q = 0; q < kc.n_pre; ++q) {
      const int c = b.terms[nn.term_begin + pre[q]].obs_col;
      kc.pre_col[q] = ctx->obs.p + (size_t)c * ctx->n_rows + ctx->active_begin;
    }
  }
  // static tuple ids of the loaded rows replace the per-sweep column hashing / compares
  static const bool no_tuple_ids = getenv("PCLEAN_NO_TUPLE_IDS") != nullptr;
  if (!ctx->prior_mode && !ctx->obs_override && !cols.empty() && node_id < 64 && !no_tuple_ids) {
    int32_t pre_cols[3] = {-1, -1, -1};
    const pclean_node& nn2 = b.nodes[node_id];
    int32_t pre2[3];
    const int np2 = kc.n_pre > 0 ? prefilter_terms(ctx, b, nn2, pre2) : 0;
    for (int q = 0; q < np2; ++q) pre_cols[q] = b.terms[nn2.term_begin + pre2[q]].obs_col;
    int rc = ensure_tuple_ids(ctx, block_id, node_id, cols, pre_cols, np2, &kc.tuple_id, &kc.pre_hash);
    if (rc) return rc;
    if (np2 == 0) kc.pre_hash = nullptr;
  }
  static const bool sort_groups = getenv("PCLEAN_SORT_GROUPS") != nullptr;
  if (!sort_groups) return make_item_groups_hash(ctx, il, excl, kc, split_m, want_members, g);
  uint64_t* key = scratch<uint64_t>(ctx, n);
  uint64_t* key_s = scratch<uint64_t>(ctx, n);
  int32_t* idx = scratch<int32_t>(ctx, n);
  int32_t* idx_s = scratch<int32_t>(ctx, n);
  int32_t* head = scratch<int32_t>(ctx, n);
  int32_t* uid = scratch<int32_t>(ctx, n);
  if (!key || !key_s || !idx || !idx_s || !head || !uid) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  size_t tmp_sort = 0, tmp_scan = 0;
  int key_bits = 32, low_bits = 24;
  bool k32 = true;  // without a referent the key is 32 hash bits
  if (excl) {  // referent ids are < rows of this node's table (+1 for "none")
    const int kmax = ctx->cand[b.nodes[node_id].table].n_rows + 2;
    int rb = 1;
    while ((1ll << rb) < kmax) ++rb;
    static const bool force64 = getenv("PCLEAN_SORT_KEY64") != nullptr;
    k32 = rb <= 16 && !force64;
    low_bits = k32 ? 32 - rb : 24;
    // (fewer hash bits would save a radix pass, but two tuples of one referent that collide are interleaved by the
    // stable sort and fall apart into one group per item: measured, 10 bits cost more in the scan than the pass saves)
    key_bits = low_bits + rb;
  }
  uint32_t* key32 = (uint32_t*)key;
  uint32_t* key32_s = (uint32_t*)key_s;
  if (k32) {
    hipLaunchKernelGGL(item_key_kernel<uint32_t>, grid1(n), dim3(256), 0, ctx->stream, n, kc, il.row, il.ctx, excl, low_bits,
                       key32, idx);
    HIPCHK(ctx, pclean_sort_pairs<uint32_t>(nullptr, tmp_sort, key32, key32_s, idx, idx_s, n, key_bits, ctx->stream));
  } else {
    hipLaunchKernelGGL(item_key_kernel<uint64_t>, grid1(n), dim3(256), 0, ctx->stream, n, kc, il.row, il.ctx, excl, low_bits,
                       key, idx);
    HIPCHK(ctx, pclean_sort_pairs<uint64_t>(nullptr, tmp_sort, key, key_s, idx, idx_s, n, key_bits, ctx->stream));
  }
  HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan, head, uid, n, ctx->stream));
  unsigned char* tmp = scratch<unsigned char>(ctx, std::max(tmp_sort, tmp_scan));
  if (!tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  if (k32) {
    HIPCHK(ctx, pclean_sort_pairs<uint32_t>(tmp, tmp_sort, key32, key32_s, idx, idx_s, n, key_bits, ctx->stream));
    hipLaunchKernelGGL(item_head_kernel<uint32_t>, grid1(n), dim3(256), 0, ctx->stream, n, kc, il.row, il.ctx, excl, key32_s,
                       idx_s, head, split_m);
  } else {
    HIPCHK(ctx, pclean_sort_pairs<uint64_t>(tmp, tmp_sort, key, key_s, idx, idx_s, n, key_bits, ctx->stream));
    hipLaunchKernelGGL(item_head_kernel<uint64_t>, grid1(n), dim3(256), 0, ctx->stream, n, kc, il.row, il.ctx, excl, key_s,
                       idx_s, head, split_m);
  }
  HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(tmp, tmp_scan, head, uid, n, ctx->stream));
  int32_t n_unique = 0;
  PCLEAN_READ_COUNT(ctx, uid + (n - 1), &n_unique);
  if (n_unique <= 0 || (double)n_unique > 0.75 * n) return PCLEAN_OK;  // not worth the indirection
  int32_t* grp_off = scratch<int32_t>(ctx, (size_t)n_unique + 1);
  if (!grp_off) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  hipLaunchKernelGGL(group_offsets_kernel, grid1((size_t)n + 1), dim3(256), 0, ctx->stream, n, head, uid, grp_off);
  g.n_groups = n_unique;
  g.grp_off = grp_off;
  g.members = idx_s;
  g.head = head;
  g.uid = uid;
  return PCLEAN_OK;
}

// log marginal of sub-tree `node_id` for every item (no draws), evaluated once per distinct item tuple
// ---- memo of option-list marginals ---------------------------------------------------------------------------
// The log-marginal of an option list (LEAF node) is a pure function of (observed values of its terms, ctx) as long
// as its option table, pair tables and fn tables stay what they are — the data never changes, so the same tuples
// come back sweep after sweep (the reference memoises its AddTypos densities the same way, add_typos.jl:47,55).
// Open-addressing table in HBM: 3 x uint64 key (up to 6 values, each stored +1) + the fp64 marginal.  Lookups
// and inserts run in different kernels, so a reader never meets a half-written entry; two inserts of one key may
// land in two slots (harmless: equal values).
#define MEMO_PROBES 32
struct MemoDev {
  uint64_t* keys;
  double* vals;
  unsigned int* count;
  unsigned int cap_mask, max_fill;
};
__device__ __forceinline__ void memo_key(const KeyColsDev& kc, int r, const int32_t* ctxv, size_t i, uint64_t* k) {
  uint32_t v[6] = {0u, 0u, 0u, 0u, 0u, 0u};
  int nv = 0;
  for (int c = 0; c < kc.n_cols; ++c) v[nv++] = (uint32_t)(kc.col[c][r] + 1);
  if (kc.use_ctx && ctxv)
    for (int q = 0; q < PCLEAN_MAX_CTX; ++q) v[nv++] = (uint32_t)(ctxv[i * PCLEAN_MAX_CTX + q] + 1);
  k[0] = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
  k[1] = (uint64_t)v[2] | ((uint64_t)v[3] << 32);
  k[2] = (uint64_t)v[4] | ((uint64_t)v[5] << 32);
}
__device__ __forceinline__ uint32_t memo_hash(const uint64_t* k) {
  uint64_t h = k[0] * 0x9e3779b97f4a7c15ull;
  h ^= (h >> 29) + k[1] * 0xbf58476d1ce4e5b9ull;
  h ^= (h >> 31) + k[2] * 0x94d049bb133111ebull;
  h *= 0xff51afd7ed558ccdull;
  return (uint32_t)(h >> 32);
}
__global__ void memo_lookup_kernel(int n, KeyColsDev kc, const int32_t* row, const int32_t* ctxv, MemoDev m,
                                   double* __restrict__ lse_out, int32_t* __restrict__ miss_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t k[3];
  memo_key(kc, row ? row[i] : i, ctxv, (size_t)i, k);
  uint32_t slot = memo_hash(k) & m.cap_mask;
  int32_t miss = PCLEAN_CHOICE_NEW;
  for (int p = 0; p < MEMO_PROBES; ++p) {
    const uint64_t* e = m.keys + (size_t)slot * 3;
    const uint64_t k0 = e[0];
    if (k0 == ~0ull) break;
    if (k0 == k[0] && e[1] == k[1] && e[2] == k[2]) {
      lse_out[i] = m.vals[slot];
      miss = 0;
      break;
    }
    slot = (slot + 1) & m.cap_mask;
  }
  miss_flag[i] = miss;
}
// items list[j] (or all items when list is null) with freshly computed marginals src[j] -> table
__global__ void memo_insert_kernel(int n, const int32_t* list, KeyColsDev kc, const int32_t* row, const int32_t* ctxv,
                                   MemoDev m, const double* __restrict__ src) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (*m.count >= m.max_fill) return;
  const int i = list ? list[j] : j;
  uint64_t k[3];
  memo_key(kc, row ? row[i] : i, ctxv, (size_t)i, k);
  uint32_t slot = memo_hash(k) & m.cap_mask;
  for (int p = 0; p < MEMO_PROBES; ++p) {
    unsigned long long* e = (unsigned long long*)(m.keys + (size_t)slot * 3);
    const unsigned long long old = atomicCAS(e, ~0ull, (unsigned long long)k[0]);
    if (old == ~0ull) {
      e[1] = k[1];
      e[2] = k[2];
      m.vals[slot] = src[j];
      atomicAdd(m.count, 1u);
      return;
    }
    slot = (slot + 1) & m.cap_mask;
  }
}

static int eval_node_lse_core(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                              uint64_t seed, uint32_t sweep, double* lse_out);

// log marginal of sub-tree `node_id` for every item (no draws)
static int eval_node_lse(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                         uint64_t seed, uint32_t sweep, double* lse_out) {
  Block& b = ctx->block[block_id];
  const pclean_node& n = b.nodes[node_id];
  SweepState* s = st(ctx);
  std::set<int> cols;
  bool use_ctx = false;
  static const bool no_memo = getenv("PCLEAN_NO_MEMO") != nullptr;
  const bool memo_ok = !no_memo && n.kind == PCLEAN_NODE_LEAF && !il.ev_lo && il.n >= 4096 && node_id < 64 &&
                       subtree_key(ctx, b, node_id, cols, use_ctx) &&
                       (int)cols.size() + (use_ctx ? PCLEAN_MAX_CTX : 0) <= 6 && (!use_ctx || il.ctx);
  if (!memo_ok) return eval_node_lse_core(ctx, block_id, node_id, il, excl, seed, sweep, lse_out);
  ProfScope ps(ctx, "option_marginal_memo");
  // version of everything the marginal depends on
  uint64_t ver = ctx->cand[n.table].version;
  for (int i = 0; i < n.n_terms; ++i) {
    const pclean_term& tm = b.terms[n.term_begin + i];
    ver = ver * 1000003ull + ctx->pair[tm.pair_table].version;
    if (tm.ctx_slot >= 0) ver = ver * 1000003ull + (uint64_t)(tm.fn_table + 1);
  }
  SweepState::LeafMemo& mm = s->memo[block_id * 64 + node_id];
  const int cap = 1 << 21;
  if (mm.cap != cap) {
    if (mm.keys.alloc((size_t)cap * 3) || mm.vals.alloc(cap) || mm.count.alloc(4))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed (memo)");
    mm.cap = cap;
    mm.ver = 0;
  }
  if (mm.ver != ver) {
    HIPCHK(ctx, hipMemsetAsync(mm.keys.p, 0xff, (size_t)cap * 3 * sizeof(uint64_t), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(mm.count.p, 0, sizeof(unsigned int), ctx->stream));
    mm.ver = ver;
  }
  MemoDev md{mm.keys.p, mm.vals.p, mm.count.p, (unsigned int)(cap - 1), (unsigned int)(cap / 2)};
  KeyColsDev kc{};
  kc.use_ctx = use_ctx ? 1 : 0;
  for (int c : cols) kc.col[kc.n_cols++] = ctx->obs.p + (size_t)c * ctx->n_rows + ctx->active_begin;
  const int N = il.n;
  int32_t* flag = scratch<int32_t>(ctx, N);
  int32_t* list = scratch<int32_t>(ctx, N);
  if (!flag || !list || s->counter.alloc(4)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  hipLaunchKernelGGL(memo_lookup_kernel, grid1(N), dim3(256), 0, ctx->stream, N, kc, il.row, il.ctx, md, lse_out, flag);
  unsigned int* miss_ctr = fresh_counter(ctx);
  if (!miss_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "counter bank: device alloc failed");
  hipLaunchKernelGGL(compact_new_kernel, grid1(N), dim3(256), 0, ctx->stream, (size_t)N, flag, 1, miss_ctr, list, nullptr);
  unsigned int n_miss = 0;
  PCLEAN_READ_COUNT(ctx, miss_ctr, &n_miss);
  if (n_miss == 0) return PCLEAN_OK;
  if (n_miss == (unsigned int)N) {
    int rc = eval_node_lse_core(ctx, block_id, node_id, il, excl, seed, sweep, lse_out);
    if (rc) return rc;
    hipLaunchKernelGGL(memo_insert_kernel, grid1(N), dim3(256), 0, ctx->stream, N, (const int32_t*)nullptr, kc, il.row,
                       il.ctx, md, lse_out);
    return PCLEAN_OK;
  }
  int32_t* row2 = scratch<int32_t>(ctx, n_miss);
  int32_t* ctx2 = scratch<int32_t>(ctx, (size_t)n_miss * PCLEAN_MAX_CTX);
  int32_t* excl2 = scratch<int32_t>(ctx, n_miss);
  double* dst = scratch<double>(ctx, n_miss);
  if (!row2 || !ctx2 || !excl2 || !dst) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  hipLaunchKernelGGL(sub_items_kernel, grid1(n_miss), dim3(256), 0, ctx->stream, (int)n_miss, list, il.row, il.ctx, excl,
                     (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, row2,
                     ctx2, excl2, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
  ItemList sil{(int)n_miss, row2, il.ctx ? ctx2 : nullptr, nullptr, nullptr};
  int rc = eval_node_lse_core(ctx, block_id, node_id, sil, excl ? excl2 : nullptr, seed, sweep, dst);
  if (rc) return rc;
  hipLaunchKernelGGL(scatter_f64_kernel, grid1(n_miss), dim3(256), 0, ctx->stream, (int)n_miss, list, dst, lse_out);
  hipLaunchKernelGGL(memo_insert_kernel, grid1(n_miss), dim3(256), 0, ctx->stream, (int)n_miss, list, kc, il.row, il.ctx, md,
                     dst);
  return PCLEAN_OK;
}

// evaluated once per distinct item tuple
static int eval_node_lse_core(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                              uint64_t seed, uint32_t sweep, double* lse_out) {
  ItemGroups g;
  int rc0 = make_item_groups(ctx, block_id, node_id, il, excl, g, 0, false);  // (which group is item i in: no member lists)
  if (rc0) return rc0;
  if (g.n_groups == 0)
    return eval_node(ctx, block_id, node_id, il, excl, seed, sweep, 0, lse_out, nullptr, nullptr, nullptr, false);
  const int n = il.n;
  const int32_t n_unique = g.n_groups;
  int32_t* uid_of_item = scratch<int32_t>(ctx, n);
  if (!uid_of_item) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  int32_t* row2 = scratch<int32_t>(ctx, n_unique);
  int32_t* ctx2 = scratch<int32_t>(ctx, (size_t)n_unique * PCLEAN_MAX_CTX);
  int32_t* excl2 = scratch<int32_t>(ctx, n_unique);
  double* lse_u = scratch<double>(ctx, n_unique);
  if (!row2 || !ctx2 || !excl2 || !lse_u) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  if (g.hg_incl)
    hipLaunchKernelGGL(hg_unique_kernel, grid1(n), dim3(256), 0, ctx->stream, n, g.hg_incl, g.hg_rep, g.hg_slot_of, il.row, il.ctx,
                       excl, uid_of_item, row2, ctx2, excl2);
  else
    hipLaunchKernelGGL(item_unique_kernel, grid1(n), dim3(256), 0, ctx->stream, n, g.members, g.head, g.uid, il.row, il.ctx, excl,
                       uid_of_item, row2, ctx2, excl2);
  ItemList il2;
  il2.n = n_unique;
  il2.row = row2;
  il2.ctx = il.ctx ? ctx2 : nullptr;
  int rc = eval_node(ctx, block_id, node_id, il2, excl ? excl2 : nullptr, seed, sweep, 0, lse_u, nullptr, nullptr, nullptr,
                     false);
  if (rc) return rc;
  hipLaunchKernelGGL(gather_f64_kernel, grid1(n), dim3(256), 0, ctx->stream, n, uid_of_item, lse_u, lse_out);
  return PCLEAN_OK;
}

// Top-down sampling of the children of freshly proposed rows
// (the per-branch draws of proposal_compiler.jl:115-127 / 233-245 for the blind
// new-row branch, done lazily only for (row, particle) pairs that picked it).
int sample_children(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                           uint64_t seed, uint32_t sweep, int32_t* vals, int n_nodes) {
  Block& b = ctx->block[block_id];
  const pclean_node& n = b.nodes[node_id];
  const CandTable& t = ctx->cand[n.table];
  for (int c = 0; c < n.n_children; ++c) {
    const int cid = b.children[n.child_begin + c];
    const pclean_node& cn = b.nodes[cid];
    int32_t* draws = scratch<int32_t>(ctx, il.n);
    if (!draws) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    const int32_t* child_excl = nullptr;
    if (cn.kind == PCLEAN_NODE_FK && excl) {
      int32_t* ce = scratch<int32_t>(ctx, il.n);
      if (!ce) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      hipLaunchKernelGGL(derive_excl_kernel, grid1(il.n), dim3(256), 0, ctx->stream, il.n, excl, t.counts.p,
                         t.cols.p + (size_t)cn.parent_fk_col * t.n_rows, ce);
      child_excl = ce;
    }
    int rc = eval_node(ctx, block_id, cid, il, child_excl, seed, sweep, 1, nullptr, draws, nullptr, nullptr, false);
    if (rc) return rc;
    hipLaunchKernelGGL(scatter_vals_kernel, grid1(il.n), dim3(256), 0, ctx->stream, il.n, il.origin, draws, n_nodes,
                       cid, vals);
    if (cn.kind == PCLEAN_NODE_FK && cn.n_children > 0) {
      // rows of this child that were themselves proposed as NEW
      unsigned int* new_ctr = fresh_counter(ctx);
      if (!new_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "counter bank: device alloc failed");
      hipLaunchKernelGGL(compact_new_kernel, grid1(il.n), dim3(256), 0, ctx->stream, (size_t)il.n, draws, 0, new_ctr, nullptr,
                         nullptr);
      unsigned int cnt = 0;
      PCLEAN_READ_COUNT(ctx, new_ctr, &cnt);
      if (cnt) {
        int32_t* list = scratch<int32_t>(ctx, cnt);
        int32_t* row = scratch<int32_t>(ctx, cnt);
        int32_t* cx = scratch<int32_t>(ctx, (size_t)cnt * PCLEAN_MAX_CTX);
        int32_t* part = scratch<int32_t>(ctx, cnt);
        int32_t* org = scratch<int32_t>(ctx, cnt);
        int32_t* sub_excl = scratch<int32_t>(ctx, cnt);
        if (!list || !row || !cx || !part || !org || !sub_excl) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        unsigned int* list_ctr = fresh_counter(ctx);
        if (!list_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "counter bank: device alloc failed");
        hipLaunchKernelGGL(compact_new_kernel, grid1(il.n), dim3(256), 0, ctx->stream, (size_t)il.n, draws, 1, list_ctr, list, nullptr);
        int32_t* evl = il.ev_lo ? scratch<int32_t>(ctx, cnt) : nullptr;
        int32_t* evh = il.ev_lo ? scratch<int32_t>(ctx, cnt) : nullptr;
        int32_t* rng = il.rng_row ? scratch<int32_t>(ctx, cnt) : nullptr;
        if ((il.ev_lo && (!evl || !evh)) || (il.rng_row && !rng)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        hipLaunchKernelGGL(sublist_items_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, list, il.row, il.ctx,
                           il.particle, il.origin, row, cx, part, org, il.ev_lo, il.ev_hi, il.rng_row, evl, evh, rng);
        // exclusion of the child's table for the sub-list = gather of child_excl
        if (child_excl)
          hipLaunchKernelGGL(gather_i32_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, list, child_excl,
                             sub_excl);
        ItemList sub{(int)cnt, row, cx, part, org, evl, evh, il.ev_rows, il.ev_ctx, rng};
        rc = sample_children(ctx, block_id, cid, sub, child_excl ? sub_excl : nullptr, seed, sweep, vals, n_nodes);
        if (rc) return rc;
      }
    }
  }
  return PCLEAN_OK;
}

// The candidate-compact tables (and prior rows) of a reference slot brought up to date on ANOTHER stream: the refresh after a
// commit wrote a few hundred rows of the table (compact_update / compact_min, ~0.18 ms for the Measure slot at 1M rows) does
// not depend on the blocks before it, so pclean_sweep lets it run beside block 0 (sweep.hip).  The launches go to `side`;
// the host-side versions move at once, so the later call on the library's stream finds nothing left to do.
int prefetch_fast_root(pclean_ctx* ctx, int block_id, int node_id, hipStream_t side) {
  hipStream_t saved = ctx->stream;
  ctx->stream = side;
  FastRootDev fr;
  const int rc = try_fast_root(ctx, block_id, node_id, fr);
  ctx->stream = saved;
  return rc < 0 ? rc : PCLEAN_OK;
}

// One-time set-up of everything the first sweeps would otherwise build on their way: the candidate-compact byte tables,
// block minima and prior rows of every reference slot / long-string option list of every loaded block (observed-class
// blocks and latent-class plans alike: the same cache entries whichever kind of sweep asks first), the per-value caches of
// cacheable option lists.  The buffers are a few GB on the 1M-row workload and allocating them is what made the FIRST
// run_inference iteration 0.4-0.8 s where the later ones take 0.3 s (and twice as long again on some boxes).  Call it once
// the latent tables are uploaded (with the capacity the device-resident commit wants, when that is on).
extern "C" int pclean_prepare(pclean_ctx* ctx, uint32_t ev_blocks) {  // bit bi: block bi is a latent-class plan (evidence sets)
  if (!ctx) return PCLEAN_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  {
    const int rcb = begin_call(ctx);
    if (rcb) return rcb;
  }
  for (int bi = 0; bi < PCLEAN_MAX_BLOCKS; ++bi) {
    Block& b = ctx->block[bi];
    if (!b.valid || b.is_score) continue;
    for (int node = 0; node < (int)b.nodes.size() && node < 64; ++node) {
      const pclean_node& n = b.nodes[node];
      if (n.table < 0 || n.table >= PCLEAN_MAX_TABLES || !ctx->cand[n.table].valid) continue;
      if (node < (int)b.node_gauss.size() && b.node_gauss[node] >= 0) continue;  // (never takes the compact-table kernels)
      FastRootDev fr;
      const int rc = try_fast_root(ctx, bi, node, fr, n.kind == PCLEAN_NODE_LEAF && ((ev_blocks >> bi) & 1u));
      if (rc < 0) return rc;
      if (n.kind == PCLEAN_NODE_LEAF && n.cacheable && n.n_terms == 1 && b.terms[n.term_begin].ctx_slot < 0) {
        const double* cache;
        const int32_t* ocol;
        int n_obs;
        const int rcl = ensure_leaf_cache(ctx, bi, node, &cache, &ocol, &n_obs);
        if (rcl) return rcl;
      }
    }
  }
  PCLEAN_SYNC(ctx);
  return PCLEAN_OK;
}

int ensure_plan_dev(pclean_ctx* ctx, int block_id) {
  SweepState* s = st(ctx);
  BlockRun& r = s->run[block_id];
  const Block& b = ctx->block[block_id];
  const int nn = (int)b.nodes.size();
  std::vector<int32_t> kind(nn), nrows(nn), cmb(nn);
  std::vector<const int32_t*> cols(nn);
  for (int i = 0; i < nn; ++i) {
    const CandTable& t = ctx->cand[b.nodes[i].table];
    kind[i] = b.nodes[i].kind;
    nrows[i] = t.n_rows;
    cmb[i] = b.nodes[i].colmap_begin;
    cols[i] = t.cols.p;
  }
  // unchanged since the last upload (same tables at the same addresses with the same shapes): nothing to do — the
  // arrays are tiny, but five copies and a synchronisation per block and sweep are not
  if (r.plan_ready && r.plan_sig_block == b.version && r.plan_sig_nrows == nrows && r.plan_sig_cols == cols && r.plan_sig_colmap == b.colmap.size() &&
      r.plan_sig_kind == kind && r.plan_sig_cmb == cmb)
    return PCLEAN_OK;
  if (r.plan_kind.alloc(nn) || r.plan_nrows.alloc(nn) || r.plan_cmb.alloc(nn) || r.plan_cols.alloc(nn) ||
      r.plan_colmap.alloc(std::max<size_t>(b.colmap.size(), 2)))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "alloc");
  HIPCHK(ctx, hipMemcpyAsync(r.plan_kind.p, kind.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(r.plan_nrows.p, nrows.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(r.plan_cmb.p, cmb.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(r.plan_cols.p, cols.data(), nn * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  if (!b.colmap.empty())
    HIPCHK(ctx, hipMemcpyAsync(r.plan_colmap.p, b.colmap.data(), b.colmap.size() * 4, hipMemcpyHostToDevice,
                               ctx->stream));
  PCLEAN_SYNC(ctx);  // host vectors go out of scope
  r.plan = PlanDev{nn, r.plan_kind.p, r.plan_cols.p, r.plan_nrows.p, r.plan_cmb.p, r.plan_colmap.p};
  r.plan_sig_nrows = nrows;
  r.plan_sig_cols = cols;
  r.plan_sig_kind = kind;
  r.plan_sig_cmb = cmb;
  r.plan_sig_colmap = b.colmap.size();
  r.plan_sig_block = b.version;
  r.plan_ready = true;
  return PCLEAN_OK;
}
