// C ABI entry points: lifecycle, string pool, observed columns, pair tables,
// density tables, StringPrior scores, candidate tables, plan upload.
// The enumeration / sweep entry points live in sweep.hip (observed class), latent.hip (latent classes) and eval.hip (plan nodes).
#include <chrono>
#include <cmath>
#include <limits>

#include <map>
#include <mutex>

#include "ctx.h"

static const double kNegInf = -std::numeric_limits<double>::infinity();
uint64_t g_pclean_version = 0;

extern "C" const char* pclean_version(void) { return "pclean-hip 0.1 (gfx950)"; }

extern "C" int pclean_ctx_create(int device_id, pclean_ctx** out) {
  if (!out) return PCLEAN_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PCLEAN_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= n) return PCLEAN_ERR_ARG;
  if (hipSetDevice(device_id) != hipSuccess) return PCLEAN_ERR_HIP;
  pclean_ctx* ctx = new pclean_ctx();
  ctx->device = device_id;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return PCLEAN_ERR_HIP;
  }
  *out = ctx;
  return PCLEAN_OK;
}

extern "C" int pclean_ctx_destroy(pclean_ctx* ctx) {
  if (!ctx) return PCLEAN_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)pclean_comm_destroy(ctx);
  pclean_commit_state_free(ctx);
  pclean_sweep_state_free(ctx);
  ctx->stage.release();
  ctx->ustage.release();
  ctx->stats_pack.release();
  ctx->sym.release();
  ctx->off.release();
  ctx->obs.release();
  ctx->iota.release();
  ctx->nb.release();
  ctx->logl.release();
  ctx->atd.release();
  for (auto& p : ctx->pair) {
    p.d.release();
    p.lat_len.release();
    p.obs_ids.release();
  }
  ctx->lm_init.release();
  ctx->lm_trans.release();
  ctx->letter_sym.release();
  for (auto& c : ctx->cand) {
    c.cols.release();
    c.counts.release();
    c.logc_full.release();
    c.logc_m1.release();
    c.stats.release();
  }
  for (auto& f : ctx->fn) f.fn.release();
  for (auto& b : ctx->block) {
    b.d_terms.release();
    for (auto& l : b.leaf_cache) l.release();
    for (auto& l : b.leaf_m) l.release();
    for (auto& l : b.leaf_U) l.release();
    for (auto& l : b.leaf_coarse) l.release();
    for (auto& l : b.leaf_udummy) l.release();
  }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return PCLEAN_OK;
}

extern "C" const char* pclean_last_error(const pclean_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// ---------------------------------------------------------------------------
extern "C" int pclean_load_strings(pclean_ctx* ctx, int32_t n_strings, const uint16_t* sym, const int64_t* off) {
  if (!ctx || n_strings < 0 || !off || (n_strings > 0 && !sym && off[n_strings] > 0))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_strings: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int64_t total = off[n_strings];
  int nsym = 0;
  for (int64_t i = 0; i < total; ++i) nsym = std::max(nsym, (int)sym[i] + 1);
  if (nsym >= 0xfffe) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "too many distinct symbols");
  if (ctx->sym.alloc((size_t)std::max<int64_t>(total, 1)) || ctx->off.alloc((size_t)n_strings + 1))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (total) HIPCHK(ctx, hipMemcpy(ctx->sym.p, sym, total * sizeof(uint16_t), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->off.p, off, ((size_t)n_strings + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  ctx->h_off.assign(off, off + n_strings + 1);
  ctx->n_strings = n_strings;
  ctx->n_symbols = nsym;
  return PCLEAN_OK;
}

__global__ void iota_kernel(int32_t* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

extern "C" int pclean_load_columns(pclean_ctx* ctx, int32_t n_rows, int32_t n_cols, const int32_t* obs) {
  if (!ctx || n_rows < 0 || n_cols < 0 || (!obs && (int64_t)n_rows * n_cols > 0))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_columns: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)n_rows * n_cols;
  if (ctx->obs.alloc(std::max<size_t>(n, 1))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (n) HIPCHK(ctx, hipMemcpy(ctx->obs.p, obs, n * sizeof(int32_t), hipMemcpyHostToDevice));
  ctx->n_rows = n_rows;
  ctx->n_cols = n_cols;
  ++ctx->obs_version;
  ctx->col_has_missing.assign(n_cols, 0);
  for (int c = 0; c < n_cols; ++c)
    for (int i = 0; i < n_rows; ++i)
      if (obs[(size_t)c * n_rows + i] < 0) {
        ctx->col_has_missing[c] = 1;
        break;
      }
  return PCLEAN_OK;
}

// ---------------------------------------------------------------------------
// AddTypos density pieces (add_typos.jl:61-63), computed once with host libm
// and shared bit-for-bit by the kernels and (via pclean_get_density_tables) by
// the parity oracle.
int pclean_ensure_density(pclean_ctx* ctx, int max_len) {
  if (max_len <= ctx->max_len) return PCLEAN_OK;
  const int ml = std::max(max_len, 64);
  const int mr = (ml + 4) / 5, md = ml;
  std::vector<double> nb((size_t)(mr + 1) * (md + 1)), logl(ml + 1);
  for (int r = 0; r <= mr; ++r)
    for (int d = 0; d <= md; ++d) {
      double v;
      if (r == 0)
        v = d == 0 ? 0.0 : kNegInf;
      else
        v = std::lgamma((double)d + r) - std::lgamma(d + 1.0) - std::lgamma((double)r) + r * std::log(0.9) +
            d * std::log1p(-0.9);
      nb[(size_t)r * (md + 1) + d] = v;
    }
  logl[0] = 0.0;
  for (int L = 1; L <= ml; ++L) logl[L] = std::log((double)L);
  // the full density of (latent length L, distance d): the three fp64 operations of term_density()
  // (enum_kernels.hip) evaluated here once — same IEEE operations, same order, same bits
  std::vector<double> atd((size_t)(ml + 1) * (md + 1));
  for (int L = 0; L <= ml; ++L)
    for (int d = 0; d <= md; ++d) {
      double l = nb[(size_t)((L + 4) / 5) * (md + 1) + d];
      l -= logl[L] * (double)d;
      l -= 1.629048269010741 * (double)d;
      atd[(size_t)L * (md + 1) + d] = l;
    }
  if (ctx->nb.alloc(nb.size()) || ctx->logl.alloc(logl.size()) || ctx->atd.alloc(atd.size()))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpy(ctx->atd.p, atd.data(), atd.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->nb.p, nb.data(), nb.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->logl.p, logl.data(), logl.size() * sizeof(double), hipMemcpyHostToDevice));
  ctx->h_nb.swap(nb);
  ctx->h_logl.swap(logl);
  ctx->max_r = mr;
  ctx->max_d = md;
  ctx->max_len = ml;
  return PCLEAN_OK;
}

extern "C" int pclean_get_density_tables(pclean_ctx* ctx, int32_t* max_r, int32_t* max_d, int32_t* max_len,
                                         double* nb, double* logl) {
  if (!ctx) return PCLEAN_ERR_ARG;
  if (ctx->max_len < 0) {
    int rc = pclean_ensure_density(ctx, 64);
    if (rc) return rc;
  }
  if (max_r) *max_r = ctx->max_r;
  if (max_d) *max_d = ctx->max_d;
  if (max_len) *max_len = ctx->max_len;
  if (nb) memcpy(nb, ctx->h_nb.data(), ctx->h_nb.size() * sizeof(double));
  if (logl) memcpy(logl, ctx->h_logl.data(), ctx->h_logl.size() * sizeof(double));
  return PCLEAN_OK;
}

extern "C" int pclean_build_pair_table(pclean_ctx* ctx, int32_t table_id, int32_t n_obs, const int32_t* obs_ids,
                                       int32_t n_lat, const int32_t* lat_ids, int32_t dist_mode) {
  // n_obs == 0 is legal: a column whose every cell is missing has an empty observed domain
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || n_obs < 0 || n_lat <= 0 || (n_obs > 0 && !obs_ids) ||
      !lat_ids || (dist_mode != PCLEAN_DIST_OSA && dist_mode != PCLEAN_DIST_DL))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_build_pair_table: bad arguments");
  if (ctx->n_strings == 0) return pclean_fail(ctx, PCLEAN_ERR_STATE, "load strings first");
  if (n_obs > 65535) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "n_obs > 65535 not supported yet");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairTable& pt = ctx->pair[table_id];
  int max_la = 0, max_lb = 0;
  for (int i = 0; i < n_obs; ++i) {
    if (obs_ids[i] < 0 || obs_ids[i] >= ctx->n_strings) return pclean_fail(ctx, PCLEAN_ERR_ARG, "obs id out of range");
    max_la = std::max(max_la, (int)(ctx->h_off[obs_ids[i] + 1] - ctx->h_off[obs_ids[i]]));
  }
  int64_t sum_lb = 0;
  for (int i = 0; i < n_lat; ++i) {
    if (lat_ids[i] < 0 || lat_ids[i] >= ctx->n_strings) return pclean_fail(ctx, PCLEAN_ERR_ARG, "lat id out of range");
    const int len = (int)(ctx->h_off[lat_ids[i] + 1] - ctx->h_off[lat_ids[i]]);
    max_lb = std::max(max_lb, len);
    sum_lb += len;
  }
  pt.mean_lat_len = (double)sum_lb / (double)n_lat;
  pt.n_obs = n_obs;
  pt.n_lat = n_lat;
  pt.max_obs_len = max_la;
  pt.max_lat_len = max_lb;
  pt.elem_bytes = std::max(max_la, max_lb) <= 255 ? 1 : 2;
  int rc = pclean_ensure_density(ctx, std::max(max_la, max_lb));
  if (rc) return rc;
  if (pt.d.alloc(std::max<size_t>((size_t)n_obs * n_lat * pt.elem_bytes, 16)) || pt.lat_len.alloc(n_lat))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  DevBuf<int32_t> d_lat;
  DevBuf<int32_t>& d_obs = pt.obs_ids;  // kept: the weight of a chosen dummy value scores drawn strings against them
  pt.dist_mode = dist_mode;
  if (d_obs.alloc(std::max(n_obs, 1)) || d_lat.alloc(n_lat)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (n_obs) HIPCHK(ctx, hipMemcpy(d_obs.p, obs_ids, n_obs * sizeof(int32_t), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(d_lat.p, lat_ids, n_lat * sizeof(int32_t), hipMemcpyHostToDevice));
  rc = pclean_launch_dist(ctx, pt, d_obs.p, d_lat.p, dist_mode, lat_ids);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  d_lat.release();
  if (rc) return rc;
  if (e != hipSuccess) return pclean_fail(ctx, PCLEAN_ERR_HIP, "distance kernel failed: %s", hipGetErrorString(e));
  pt.valid = true;
  pt.version = ++g_pclean_version;
  return PCLEAN_OK;
}

extern "C" int pclean_set_lm_tables(pclean_ctx* ctx, const double* init_p, const double* trans_p, const uint16_t* letter_sym) {
  if (!ctx || !init_p || !trans_p || !letter_sym) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_lm_tables: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->lm_init.alloc(28) || ctx->lm_trans.alloc(28 * 28) || ctx->letter_sym.alloc(28))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpy(ctx->lm_init.p, init_p, 28 * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->lm_trans.p, trans_p, 28 * 28 * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->letter_sym.p, letter_sym, 28 * sizeof(uint16_t), hipMemcpyHostToDevice));
  ctx->lm_valid = true;
  return PCLEAN_OK;
}

extern "C" int pclean_set_pair_table(pclean_ctx* ctx, int32_t table_id, int32_t n_obs, int32_t n_lat,
                                     const uint8_t* table) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || n_obs <= 0 || n_lat <= 0 || !table)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_pair_table: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairTable& pt = ctx->pair[table_id];
  pt.n_obs = n_obs;
  pt.n_lat = n_lat;
  pt.elem_bytes = 1;
  pt.max_lat_len = pt.max_obs_len = 0;
  int rc = pclean_ensure_density(ctx, 64);
  if (rc) return rc;
  if (pt.d.alloc((size_t)n_obs * n_lat) || pt.lat_len.alloc(n_lat))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpy(pt.d.p, table, (size_t)n_obs * n_lat, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemset(pt.lat_len.p, 0, n_lat * sizeof(uint16_t)));
  pt.valid = true;
  pt.version = ++g_pclean_version;
  return PCLEAN_OK;
}

extern "C" int pclean_get_pair_table(pclean_ctx* ctx, int32_t table_id, uint16_t* out) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || !out || !ctx->pair[table_id].valid)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_get_pair_table: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairTable& pt = ctx->pair[table_id];
  const size_t n = (size_t)pt.n_obs * pt.n_lat;
  if (pt.elem_bytes == 2) {
    HIPCHK(ctx, hipMemcpy(out, pt.d.p, n * 2, hipMemcpyDeviceToHost));
  } else {
    std::vector<uint8_t> tmp(n);
    HIPCHK(ctx, hipMemcpy(tmp.data(), pt.d.p, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) out[i] = tmp[i];
  }
  return PCLEAN_OK;
}

extern "C" int pclean_get_pair_rows(pclean_ctx* ctx, int32_t table_id, int32_t n, const int32_t* obs_rows,
                                    uint16_t* out) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || !out || n <= 0 || !obs_rows ||
      !ctx->pair[table_id].valid)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_get_pair_rows: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairTable& pt = ctx->pair[table_id];
  std::vector<uint8_t> tmp((size_t)pt.n_lat * pt.elem_bytes);
  for (int j = 0; j < n; ++j) {
    if (obs_rows[j] < 0 || obs_rows[j] >= pt.n_obs) return pclean_fail(ctx, PCLEAN_ERR_ARG, "obs row out of range");
    HIPCHK(ctx, hipMemcpy(tmp.data(), pt.d.p + (size_t)obs_rows[j] * pt.n_lat * pt.elem_bytes, tmp.size(),
                          hipMemcpyDeviceToHost));
    uint16_t* o = out + (size_t)j * pt.n_lat;
    if (pt.elem_bytes == 2)
      memcpy(o, tmp.data(), tmp.size());
    else
      for (int v = 0; v < pt.n_lat; ++v) o[v] = tmp[v];
  }
  return PCLEAN_OK;
}

// ---------------------------------------------------------------------------
// StringPrior (string_prior.jl:43-61): one lane per string, sequential bigram
// chain (the sum order is part of the parity contract).
__global__ void string_prior_kernel(const uint8_t* __restrict__ lm, const int64_t* __restrict__ off, int n,
                                    int min_len, int max_len, double base, double log28,
                                    const double* __restrict__ init_logp, const double* __restrict__ trans_logp,
                                    double* __restrict__ out) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int64_t o = off[s];
  const int len = (int)(off[s + 1] - o);
  if (len < min_len || len > max_len) {
    out[s] = -__builtin_inf();
    return;
  }
  double score = base;
  int prev = -1;
  for (int i = 0; i < len; ++i) {
    const double* dist = prev < 0 ? init_logp : trans_logp + prev * 28;
    const uint8_t c = lm[o + i];
    prev = c == 255 ? -1 : (int)c;
    score += prev < 0 ? -log28 : dist[prev];
  }
  out[s] = score;
}

extern "C" int pclean_string_prior_scores(pclean_ctx* ctx, int32_t n_strings, const uint8_t* lm, const int64_t* off,
                                          int32_t min_len, int32_t max_len, const double* init_logp,
                                          const double* trans_logp, double* out) {
  if (!ctx || n_strings < 0 || !off || !init_logp || !trans_logp || !out || max_len < min_len)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_string_prior_scores: bad arguments");
  if (n_strings == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<uint8_t> d_lm;
  DevBuf<int64_t> d_off;
  DevBuf<double> d_init, d_trans, d_out;
  const int64_t total = off[n_strings];
  int rc = PCLEAN_OK;
  if (d_lm.alloc(std::max<int64_t>(total, 1)) || d_off.alloc(n_strings + 1) || d_init.alloc(28) ||
      d_trans.alloc(28 * 28) || d_out.alloc(n_strings))
    rc = pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  hipError_t e = hipSuccess;
  if (!rc) {
    if (total) e = hipMemcpy(d_lm.p, lm, total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off.p, off, (n_strings + 1) * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_init.p, init_logp, 28 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_trans.p, trans_logp, 28 * 28 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      const double base = -std::log((double)(max_len - min_len + 1));
      hipLaunchKernelGGL(string_prior_kernel, dim3((n_strings + 255) / 256), dim3(256), 0, ctx->stream, d_lm.p,
                         d_off.p, n_strings, min_len, max_len, base, std::log(28.0), d_init.p, d_trans.p, d_out.p);
      e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out.p, n_strings * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = pclean_fail(ctx, PCLEAN_ERR_HIP, "string prior kernel failed: %s", hipGetErrorString(e));
  }
  d_lm.release();
  d_off.release();
  d_init.release();
  d_trans.release();
  d_out.release();
  return rc;
}

// ---------------------------------------------------------------------------
// Candidate tables.  CRP prior pieces follow proposal_compiler.jl:165-171:
//   existing k: log(count_k - discount) - log(total + strength)
//   new       : log(strength + discount * n_rows) - log(total + strength)
// with the evidence row's own reference removed first (row_inference.jl:115-126),
// hence the "_m1" variants.
extern "C" int pclean_set_table(pclean_ctx* ctx, int32_t table_id, int32_t n_rows, int32_t n_cols,
                                const int32_t* cols, const int64_t* counts, double strength, double discount) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || n_rows < 0 || n_cols < 0 || !counts)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_table: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  static const bool dbg_upload = getenv("PCLEAN_DEBUG_UPLOAD") != nullptr;
  const auto dbg_t0 = std::chrono::steady_clock::now();
  auto dbg_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0).count(); };
  double dbg_a = 0, dbg_b = 0, dbg_c = 0;
  CandTable& t = ctx->cand[table_id];
  // cols == NULL: keep the columns uploaded before (same shape), refresh counts / CRP pieces only
  const bool keep_cols = !cols && (int64_t)n_rows * n_cols > 0;
  if (keep_cols && (!t.valid || t.is_options || t.n_rows != n_rows || t.n_cols != n_cols))
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_set_table: cols == NULL needs a previous upload of the same shape");
  const bool same_shape = t.valid && !t.is_options && t.n_rows == n_rows && t.n_cols == n_cols;
  const bool mirror_was_stale = t.h_cols_stale;  // (a device commit wrote value columns: h_cols is not what the device holds)
  const uint64_t prev_cols_version = t.cols_version;
  int32_t upload_delta_n = -1;
  t.is_options = false;
  t.is_options_1col = false;
  t.n_used = 0;  // (pclean_commit_set_table_state tells)
  pclean_commit_table_reuploaded(ctx, table_id);  // the device commit's live flags / free stack belong to the previous upload
  t.n_rows = n_rows;
  t.n_cols = n_cols;
  const size_t n = (size_t)n_rows * n_cols;
  const size_t nr = std::max<size_t>(n_rows, 1);
  if (t.cols.alloc(std::max<size_t>(n, 1)) || t.counts.alloc(nr) || t.logc_full.alloc(nr) || t.logc_m1.alloc(nr) ||
      t.stats.alloc(nr))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  dbg_a = dbg_ms();
  int64_t total = 0, live = 0;
  // (a latent class is re-uploaded after every sub-batch of its sweep with a handful of counts moved: a row whose count and
  // discount are what the previous upload held keeps its two logarithms — the same function of the same arguments)
  const bool keep_logs = same_shape && t.discount == discount && t.h_counts.size() == (size_t)n_rows &&
                         t.h_logc_full.size() == (size_t)n_rows && t.h_logc_m1.size() == (size_t)n_rows;
  t.h_logc_full.resize(n_rows);
  t.h_logc_m1.resize(n_rows);
  for (int k = 0; k < n_rows; ++k) {
    const int64_t c = counts[k];
    if (c < 0) return pclean_fail(ctx, PCLEAN_ERR_ARG, "negative reference count");
    total += c;
    live += c > 0;
    if (keep_logs && t.h_counts[k] == c) continue;
    t.h_logc_full[k] = c > 0 ? std::log((double)c - discount) : kNegInf;
    t.h_logc_m1[k] = c > 1 ? std::log((double)(c - 1) - discount) : kNegInf;
  }
  t.h_counts.assign(counts, counts + n_rows);
  t.scal[0] = std::log((double)total + strength);
  t.scal[1] = std::log((double)(total - 1) + strength);
  t.scal[2] = std::log(strength + discount * (double)live);
  t.scal[3] = std::log(strength + discount * (double)(live - 1));
  dbg_b = dbg_ms();
  // (copies on the library's own stream, one wait at the end: a blocking hipMemcpy goes through the null stream, whose
  // first operation after the library's stream has been busy was measured at ~21 ms for 48 bytes)
  // through the library's page-locked staging area (ctx.h: HostStage), on the library's stream, one wait at the end
  {
    const size_t b_cols = (n && !keep_cols) ? n * sizeof(int32_t) : 0, b_r = (size_t)n_rows * 8;
    if (ctx->stage.grow(b_cols + 3 * b_r + (size_t)n_rows / 2 + 8 * 256)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
    ctx->stage.rewind();
    if (b_cols) {
      void* h = ctx->stage.take(b_cols);
      memcpy(h, cols, b_cols);
      HIPCHK(ctx, hipMemcpyAsync(t.cols.p, h, b_cols, hipMemcpyHostToDevice, ctx->stream));
      // which rows differ from the previous upload of the same shape, and in which columns (the compact byte tables built
      // from it are then refreshed for those rows alone: eval.hip, try_fast_root)
      static const bool no_upload_delta = getenv("PCLEAN_NO_UPLOAD_DELTA") != nullptr;
      std::vector<int32_t> rows;
      std::vector<uint64_t> masks;  // bit min(c, 63) of masks[q]: column c of rows[q] changed
      bool diffed = false;
      if (same_shape && !mirror_was_stale && !no_upload_delta && t.h_cols.size() == n) {
        diffed = true;
        std::vector<uint64_t> m((size_t)n_rows, 0);
        for (int c = 0; c < n_cols; ++c) {  // (column-major arrays: one sequential pass per column)
          const int32_t* a = cols + (size_t)c * n_rows;
          const int32_t* b0 = t.h_cols.data() + (size_t)c * n_rows;
          const uint64_t bit = 1ull << std::min(c, 63);
          for (int k = 0; k < n_rows; ++k)
            if (a[k] != b0[k]) m[k] |= bit;
        }
        for (int k = 0; k < n_rows; ++k)
          if (m[k]) {
            rows.push_back(k);
            masks.push_back(m[k]);
          }
        if (rows.size() <= (size_t)n_rows / 8) {  // the delta of this upload alone, whatever the column (t.cols_delta_*)
          upload_delta_n = (int32_t)rows.size();
          if (upload_delta_n > 0) {
            if (t.upload_delta_rows.alloc(rows.size())) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
            void* hr = ctx->stage.take(rows.size() * sizeof(int32_t));
            memcpy(hr, rows.data(), rows.size() * sizeof(int32_t));
            HIPCHK(ctx, hipMemcpyAsync(t.upload_delta_rows.p, hr, rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
          }
        }
      }
      t.h_cols.assign(cols, cols + n);
      t.h_cols_stale = false;
      // the chain of deltas (ctx.h): this upload's joins it, or — rows unknown — ends it
      if (!diffed || (!t.delta_log.empty() && t.delta_log.back().next != prev_cols_version)) t.delta_log.clear();
      if (diffed) {
        if (t.delta_log.size() >= 256) t.delta_log.erase(t.delta_log.begin(), t.delta_log.begin() + 128);
        t.delta_log.push_back(CandTable::DeltaEntry{prev_cols_version, 0, std::move(rows), std::move(masks)});  // (next: below)
      }
      t.union_n = -1;
      static const bool dbg_chain = getenv("PCLEAN_DEBUG_CHAIN") != nullptr;
      if (dbg_chain)
        fprintf(stderr, "[chain] set_table %d (%d rows): same_shape %d mirror_stale %d delta %d -> log %zu (%zu rows in the last entry)\n",
                table_id, n_rows, (int)same_shape, (int)mirror_was_stale, upload_delta_n, t.delta_log.size(),
                t.delta_log.empty() ? (size_t)0 : t.delta_log.back().rows.size());
    }
    dbg_c = dbg_ms();
    if (n_rows) {
      void* h0 = ctx->stage.take(b_r);
      void* h1 = ctx->stage.take(b_r);
      void* h2 = ctx->stage.take(b_r);
      memcpy(h0, counts, b_r);
      memcpy(h1, t.h_logc_full.data(), b_r);
      memcpy(h2, t.h_logc_m1.data(), b_r);
      HIPCHK(ctx, hipMemcpyAsync(t.counts.p, h0, b_r, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(t.logc_full.p, h1, b_r, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(t.logc_m1.p, h2, b_r, hipMemcpyHostToDevice, ctx->stream));
    }
    if (b_cols || n_rows) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  t.strength = strength;
  t.discount = discount;
  t.logc_max = kNegInf;
  for (double v : t.h_logc_full) t.logc_max = std::max(t.logc_max, v);
  t.h_mirror_stale = false;
  t.valid = true;
  t.version = ++g_pclean_version;
  if (!keep_cols) {
    t.cols_version = t.version;
    if (!t.delta_log.empty()) t.delta_log.back().next = t.cols_version;
    t.cols_delta_n = upload_delta_n;
    if (upload_delta_n >= 0) {
      t.cols_delta_base = prev_cols_version;
      t.cols_delta_rows = t.upload_delta_rows.p;
    }
  }
  if (dbg_upload && dbg_ms() > 1.0)
    fprintf(stderr, "[pclean_set_table] table %d (%d x %d, cols %s): alloc %.2f, host logs %.2f, cols copy %.2f, rest %.2f ms\n", table_id,
            n_rows, n_cols, keep_cols ? "kept" : "sent", dbg_a, dbg_b - dbg_a, dbg_c - dbg_b, dbg_ms() - dbg_c);
  return PCLEAN_OK;
}

extern "C" int pclean_set_options(pclean_ctx* ctx, int32_t table_id, int32_t n_options, const int32_t* values,
                                  const double* logp) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || n_options <= 0 || !values || !logp)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_options: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  CandTable& t = ctx->cand[table_id];
  t.is_options = true;
  t.n_rows = n_options;
  t.n_cols = 1;
  if (t.cols.alloc(n_options) || t.logc_full.alloc(n_options))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  // the candidate-compact byte tables of the wave kernel depend on the option VALUES only: keep their version when
  // just the log-probabilities changed (ChooseProportionally options are re-uploaded with every parameter move)
  const bool same_vals = t.valid && t.is_options_1col && (int)t.h_vals.size() == n_options &&
                         memcmp(t.h_vals.data(), values, (size_t)n_options * sizeof(int32_t)) == 0;
  {
    const size_t bv = (size_t)n_options * sizeof(int32_t), bl = (size_t)n_options * sizeof(double);
    if (ctx->stage.grow(bv + bl + 2 * 256)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
    ctx->stage.rewind();
    void* hv = ctx->stage.take(bv);
    void* hl = ctx->stage.take(bl);
    memcpy(hv, values, bv);
    memcpy(hl, logp, bl);
    HIPCHK(ctx, hipMemcpyAsync(t.cols.p, hv, bv, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(t.logc_full.p, hl, bl, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  t.h_vals.assign(values, values + n_options);
  t.is_options_1col = true;
  t.h_logc_full.assign(logp, logp + n_options);
  t.logc_max = kNegInf;
  for (int k = 0; k < n_options; ++k) t.logc_max = std::max(t.logc_max, logp[k]);
  t.h_mirror_stale = false;
  t.h_logc_m1.clear();
  t.h_counts.clear();
  t.scal[0] = t.scal[1] = 0.0;
  t.scal[2] = t.scal[3] = kNegInf;
  t.valid = true;
  t.version = ++g_pclean_version;
  if (!same_vals) t.cols_version = t.version;
  return PCLEAN_OK;
}

extern "C" int pclean_set_options_cols(pclean_ctx* ctx, int32_t table_id, int32_t n_options, int32_t n_cols,
                                       const int32_t* cols, const double* logp) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || n_options <= 0 || n_cols <= 0 || !cols || !logp)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_options_cols: bad arguments");
  int rc = pclean_set_options(ctx, table_id, n_options, cols, logp);
  if (rc) return rc;
  CandTable& t = ctx->cand[table_id];
  t.n_cols = n_cols;
  t.is_options_1col = false;  // value columns beyond the first: always treated as changed
  t.cols_version = t.version;
  if (t.cols.alloc((size_t)n_options * n_cols)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpy(t.cols.p, cols, (size_t)n_options * n_cols * sizeof(int32_t), hipMemcpyHostToDevice));
  return PCLEAN_OK;
}

extern "C" int pclean_load_numeric_columns(pclean_ctx* ctx, int32_t n_rows, int32_t n_cols, const double* x) {
  if (!ctx || n_rows < 0 || n_cols < 0 || (!x && (int64_t)n_rows * n_cols > 0))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_numeric_columns: bad arguments");
  if (ctx->n_rows != n_rows) return pclean_fail(ctx, PCLEAN_ERR_STATE, "load the dictionary-encoded columns first (same n_rows)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)n_rows * n_cols;
  if (ctx->xnum.alloc(std::max<size_t>(n, 1))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (n) HIPCHK(ctx, hipMemcpy(ctx->xnum.p, x, n * sizeof(double), hipMemcpyHostToDevice));
  ctx->n_xcols = n_cols;
  return PCLEAN_OK;
}

// Page-locked registrations are per process, not per context: two contexts sweeping the same trace (fast vs generic
// comparisons, two distance flavours) pin the same `cur` array.  Registrations are reference-counted by address.
static std::mutex g_pin_mu;
static std::map<void*, std::pair<size_t, int>> g_pins;  // ptr -> (bytes, references)
extern "C" int pclean_pin_host(pclean_ctx* ctx, void* ptr, size_t bytes) {
  if (!ctx || !ptr || bytes == 0) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_pin_host: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pins.find(ptr);
  if (it != g_pins.end() && it->second.first >= bytes) {
    ++it->second.second;
    return PCLEAN_OK;
  }
  if (it != g_pins.end())
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_pin_host: %p is already page-locked with a smaller size", ptr);
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  if (e == hipErrorHostMemoryAlreadyRegistered) {  // registered by someone else (torch, the caller): theirs to release
    (void)hipGetLastError();
    return PCLEAN_OK;
  }
  if (e != hipSuccess) return pclean_fail(ctx, PCLEAN_ERR_HIP, "hipHostRegister failed: %s", hipGetErrorString(e));
  g_pins[ptr] = std::make_pair(bytes, 1);
  return PCLEAN_OK;
}
extern "C" int pclean_unpin_host(pclean_ctx* ctx, void* ptr) {
  if (!ctx || !ptr) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_unpin_host: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pins.find(ptr);
  if (it == g_pins.end()) return PCLEAN_OK;  // not ours (see pclean_pin_host)
  if (--it->second.second > 0) return PCLEAN_OK;
  g_pins.erase(it);
  HIPCHK(ctx, hipHostUnregister(ptr));
  return PCLEAN_OK;
}

extern "C" int pclean_set_mean_table(pclean_ctx* ctx, int32_t table_id, int32_t n, const double* mean) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || n <= 0 || !mean)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_mean_table: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  MeanTable& m = ctx->mean[table_id];
  if (m.v.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpy(m.v.p, mean, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  m.n = n;
  m.valid = true;
  return PCLEAN_OK;
}

extern "C" int pclean_set_node_gauss(pclean_ctx* ctx, int32_t block_id, int32_t node_id, const pclean_gauss* g) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || !g)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_node_gauss: bad arguments");
  Block& b = ctx->block[block_id];
  if (node_id < 0 || node_id >= (int)b.nodes.size() || g->n_dims < 0 || g->n_dims > 4 || g->n_locals < 0 ||
      g->n_locals > 2 || g->mean_table < 0 || g->mean_table >= PCLEAN_MAX_TABLES)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_node_gauss: malformed spec");
  int combos = 1;
  for (int l = 0; l < g->n_locals; ++l) combos *= g->local_n[l];
  if (combos > 16 || (g->n_locals == 2 && g->local_n[1] > 16))
    return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "pclean_set_node_gauss: more than 16 local combinations");
  if (b.node_gauss.size() != b.nodes.size()) b.node_gauss.assign(b.nodes.size(), -1);
  b.node_gauss[node_id] = (int32_t)b.gauss.size();
  b.gauss.push_back(*g);
  return PCLEAN_OK;
}

extern "C" int pclean_set_prob_table(pclean_ctx* ctx, int32_t n, const double* p) {
  if (!ctx || n <= 0 || !p) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_prob_table: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->h_prob_same.resize(n);
  ctx->h_prob_diff.resize(n);
  for (int i = 0; i < n; ++i) {
    if (!(p[i] > 0.0 && p[i] < 1.0)) return pclean_fail(ctx, PCLEAN_ERR_ARG, "probability %d outside (0,1)", i);
    ctx->h_prob_same[i] = std::log1p(-p[i]);  // maybe_swap.jl:25
    ctx->h_prob_diff[i] = std::log(p[i]);     // maybe_swap.jl:27
  }
  if (ctx->h_logn.empty()) {
    ctx->h_logn.resize(4096);
    ctx->h_logn[0] = 0.0;
    for (int k = 1; k < 4096; ++k) ctx->h_logn[k] = std::log((double)k);
    if (ctx->logn.alloc(4096)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    HIPCHK(ctx, hipMemcpy(ctx->logn.p, ctx->h_logn.data(), 4096 * sizeof(double), hipMemcpyHostToDevice));
  }
  if (ctx->prob_same.alloc(n) || ctx->prob_diff.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpy(ctx->prob_same.p, ctx->h_prob_same.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->prob_diff.p, ctx->h_prob_diff.data(), n * sizeof(double), hipMemcpyHostToDevice));
  ctx->n_prob = n;
  return PCLEAN_OK;
}

extern "C" int pclean_load_score_block(pclean_ctx* ctx, int32_t block_id, int32_t n_terms, const int32_t* obs_col,
                                       const int32_t* pair_table, const int32_t* val_src, const int32_t* key_src,
                                       const int32_t* nopt_fn, const int32_t* other_val, int32_t prob_fn,
                                       const int32_t* prob_a_src, const int32_t* prob_b_src) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || n_terms <= 0 || n_terms > 8 || !obs_col || !pair_table ||
      !val_src || !key_src || !nopt_fn || !other_val || !prob_a_src || !prob_b_src || prob_fn < 0 ||
      prob_fn >= PCLEAN_MAX_TABLES)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_score_block: bad arguments");
  Block& b = ctx->block[block_id];
  b = Block();
  b.is_score = true;
  for (int t = 0; t < n_terms; ++t)
    b.score_terms.push_back(ScoreTerm{obs_col[t], pair_table[t], val_src[2 * t], val_src[2 * t + 1], key_src[2 * t],
                                      key_src[2 * t + 1], nopt_fn[t], other_val[t]});
  b.prob_fn = prob_fn;
  b.prob_a_block = prob_a_src[0];
  b.prob_a_col = prob_a_src[1];
  b.prob_b_block = prob_b_src[0];
  b.prob_b_col = prob_b_src[1];
  b.valid = true;
  return PCLEAN_OK;
}

extern "C" int pclean_set_block_group(pclean_ctx* ctx, int32_t block_id, int32_t group) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_block_group: bad arguments");
  ctx->block[block_id].group = group;
  return PCLEAN_OK;
}

extern "C" int pclean_set_fn_table(pclean_ctx* ctx, int32_t fn_id, int32_t n_a, int32_t n_b, const int32_t* fn) {
  if (!ctx || fn_id < 0 || fn_id >= PCLEAN_MAX_TABLES || n_a <= 0 || n_b <= 0 || !fn)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_fn_table: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  FnTable& f = ctx->fn[fn_id];
  if (f.fn.alloc((size_t)n_a * n_b)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpy(f.fn.p, fn, (size_t)n_a * n_b * sizeof(int32_t), hipMemcpyHostToDevice));
  f.n_a = n_a;
  f.n_b = n_b;
  f.valid = true;
  return PCLEAN_OK;
}

extern "C" int pclean_get_table_priors(pclean_ctx* ctx, int32_t table_id, double* logc_full, double* logc_m1,
                                       double* scal4) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || !ctx->cand[table_id].valid)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_get_table_priors: bad arguments");
  CandTable& t = ctx->cand[table_id];
  if (t.h_mirror_stale && !t.is_options) {  // a device-resident commit moved the counts: read the device arrays
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (logc_full && t.n_rows) HIPCHK(ctx, hipMemcpy(logc_full, t.logc_full.p, (size_t)t.n_rows * sizeof(double), hipMemcpyDeviceToHost));
    if (logc_m1 && t.n_rows) HIPCHK(ctx, hipMemcpy(logc_m1, t.logc_m1.p, (size_t)t.n_rows * sizeof(double), hipMemcpyDeviceToHost));
  } else {
    if (logc_full) memcpy(logc_full, t.h_logc_full.data(), t.h_logc_full.size() * sizeof(double));
    if (logc_m1 && !t.is_options) memcpy(logc_m1, t.h_logc_m1.data(), t.h_logc_m1.size() * sizeof(double));
  }
  if (scal4) memcpy(scal4, t.scal, 4 * sizeof(double));
  return PCLEAN_OK;
}

/* rows / columns of candidate table `table_id` as the library holds it (a table uploaded with spare capacity for the
 * device-resident commit has more rows than the trace's table) */
extern "C" int pclean_table_shape(pclean_ctx* ctx, int32_t table_id, int32_t* n_rows, int32_t* n_cols) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || !ctx->cand[table_id].valid)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_table_shape: bad arguments");
  if (n_rows) *n_rows = ctx->cand[table_id].n_rows;
  if (n_cols) *n_cols = ctx->cand[table_id].n_cols;
  return PCLEAN_OK;
}

extern "C" int pclean_load_block(pclean_ctx* ctx, int32_t block_id, int32_t n_nodes, const pclean_node* nodes,
                                 int32_t n_terms, const pclean_term* terms, int32_t n_children,
                                 const int32_t* children, int32_t n_colmap, const int32_t* colmap, int32_t n_ctx,
                                 const int32_t* ctx_src_block, const int32_t* ctx_src_col) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || n_nodes <= 0 || !nodes || n_terms < 0 ||
      (n_terms > 0 && !terms) || n_children < 0 || (n_children > 0 && !children) || n_colmap < 0 ||
      (n_colmap > 0 && !colmap) || (n_colmap & 1) || n_ctx < 0 || n_ctx > PCLEAN_MAX_CTX)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_block: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Block& b = ctx->block[block_id];
  for (int i = 0; i < n_nodes; ++i) {
    const pclean_node& nd = nodes[i];
    if (nd.table < 0 || nd.table >= PCLEAN_MAX_TABLES || nd.term_begin < 0 || nd.term_begin + nd.n_terms > n_terms ||
        nd.child_begin < 0 || nd.child_begin + nd.n_children > n_children || nd.parent >= n_nodes)
      return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_block: node %d malformed", i);
  }
  for (int i = 0; i < n_children; ++i)
    if (children[i] <= 0 || children[i] >= n_nodes)
      return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_block: child id out of range");
  for (int i = 0; i < n_terms; ++i) {
    const pclean_term& tm = terms[i];
    if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP) {  // ctx = evidence-row prob index, fn_table = "other" value
      if (tm.pair_table < 0 || tm.pair_table >= PCLEAN_MAX_TABLES || tm.ctx_slot < 0 || tm.ctx_slot >= PCLEAN_MAX_CTX ||
          tm.ctx_mode != 1 || tm.max_typos < 0)
        return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_block: MaybeSwap term %d malformed", i);
      continue;
    }
    if (tm.pair_table < 0 || tm.pair_table >= PCLEAN_MAX_TABLES || tm.ctx_slot >= n_ctx ||
        (tm.ctx_slot >= 0 && (tm.fn_table < 0 || tm.fn_table >= PCLEAN_MAX_TABLES)))
      return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_load_block: term %d malformed", i);
  }
  b.nodes.assign(nodes, nodes + n_nodes);
  b.terms.assign(terms, terms + n_terms);
  b.children.assign(children, children + n_children);
  b.colmap.assign(colmap, colmap + n_colmap);
  b.n_ctx = n_ctx;
  for (int s = 0; s < n_ctx; ++s) {
    b.ctx_src_block[s] = ctx_src_block[s];
    b.ctx_src_col[s] = ctx_src_col[s];
  }
  if (b.d_terms.alloc(std::max(n_terms, 1))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (n_terms) HIPCHK(ctx, hipMemcpy(b.d_terms.p, terms, n_terms * sizeof(pclean_term), hipMemcpyHostToDevice));
  for (auto& l : b.leaf_cache) l.release();
  for (auto& l : b.leaf_m) l.release();
  for (auto& l : b.leaf_U) l.release();
  for (auto& l : b.leaf_coarse) l.release();
  b.leaf_cache.clear();
  b.leaf_cache.resize(n_nodes);
  b.leaf_m.clear();
  b.leaf_m.resize(n_nodes);
  b.leaf_U.clear();
  b.leaf_U.resize(n_nodes);
  b.leaf_coarse.clear();
  b.leaf_coarse.resize(n_nodes);
  for (auto& l : b.leaf_udummy) l.release();
  b.leaf_udummy.clear();
  b.leaf_udummy.resize(n_nodes);
  b.leaf_drawable.assign(n_nodes, -1);
  for (int i = 0; i < n_nodes; ++i)  // the density tables must cover the strings random(StringPrior) can return
    if (nodes[i].kind == PCLEAN_NODE_LEAF && nodes[i].dummy_value != 0 &&
        (nodes[i].dummy_spec & 0xff) == PCLEAN_DUMMY_STRING_PRIOR) {
      const int rcd = pclean_ensure_density(ctx, (nodes[i].dummy_spec >> 16) & 0xff);
      if (rcd) return rcd;
    }
  b.gauss.clear();
  b.node_gauss.assign(n_nodes, -1);
  b.group = -1;
  b.version = ++g_pclean_version;
  b.valid = true;
  return PCLEAN_OK;
}
