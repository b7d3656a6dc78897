/* oracle/pclean_oracle.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * C entry points (ctypes-loaded by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg ONLY) over the single-thread CPU restatement of
 * PClean's hot path.  The product (pclean_amd/) never links or loads this.
 * PARITY UNPINNED — see oracle/densities.h header and DESIGN.md §Oracle.
 */
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/pclean_detmath.h"
#include "../include/pclean_philox.h"
#include "densities.h"
#include "enumerate.h"
#include "sweep.h"
#include "pruned.h"

extern "C" {

/* ---- string distances -------------------------------------------------- */
int pco_osa(const uint16_t* a, int la, const uint16_t* b, int lb) { return pco::osa_distance(a, la, b, lb); }
int pco_dl(const uint16_t* a, int la, const uint16_t* b, int lb) { return pco::dl_distance(a, la, b, lb); }

/* D[u][v] for a whole pair table (what the memo dict add_typos.jl:47 ends up holding). */
void pco_pair_table(const uint16_t* sym, const int64_t* off, int n_obs, const int32_t* obs_ids, int n_lat,
                    const int32_t* lat_ids, int mode, uint16_t* out) {
  for (int u = 0; u < n_obs; ++u)
    for (int v = 0; v < n_lat; ++v) {
      const uint16_t* a = sym + off[obs_ids[u]];
      const uint16_t* b = sym + off[lat_ids[v]];
      int la = (int)(off[obs_ids[u] + 1] - off[obs_ids[u]]), lb = (int)(off[lat_ids[v] + 1] - off[lat_ids[v]]);
      out[(size_t)u * n_lat + v] =
          (uint16_t)(mode == 0 ? pco::osa_distance(a, la, b, lb) : pco::dl_distance(a, la, b, lb));
    }
}

/* ---- densities --------------------------------------------------------- */
double pco_negbin_logpdf(double r, double p, int k) { return pco::negbin_logpdf(r, p, k); }
double pco_normal_logpdf(double x, double mu, double sigma) { return pco::normal_logpdf(x, mu, sigma); }
double pco_add_typos(int num_typos, int word_len, int max_typos) {
  return pco::add_typos_from_distance(num_typos, word_len, max_typos);
}
double pco_string_prior(const uint8_t* lm, int len, int min_len, int max_len, const double* init_p,
                        const double* trans_p) {
  return pco::string_prior_logdensity(lm, len, min_len, max_len, init_p, trans_p);
}
double pco_dummy_logmass(const double* atom_logps, int n) { return pco::dummy_logmass(atom_logps, (size_t)n); }
double pco_choose_uniformly(int n) { return pco::choose_uniformly_logdensity(n); }
double pco_choose_proportionally(int observed, const int32_t* options, const double* probs, int n) {
  return pco::choose_proportionally_logdensity(observed, options, probs, n);
}
double pco_transformed_gaussian(double backward_obs, double abs_deriv, double mean, double std) {
  return pco::transformed_gaussian_logdensity(backward_obs, abs_deriv, mean, std);
}
double pco_maybe_swap(int obs_missing, int val_in_options, int same, int n_options, double prob) {
  return pco::maybe_swap_logdensity(obs_missing != 0, val_in_options != 0, same != 0, n_options, prob);
}
int pco_time_regex(const uint32_t* s, int n) { return pco::time_regex_match(s, n) ? 1 : 0; }
double pco_time_prior(void) { return pco::time_prior_logdensity(); }
double pco_logsumexp(const double* x, int n) { return pco::logsumexp(x, (size_t)n); }
double pco_py_existing(int64_t count, int64_t total, double strength, double discount) {
  return pco::py_existing_logprob(count, total, strength, discount);
}
double pco_py_new(int64_t n_rows, int64_t total, double strength, double discount) {
  return pco::py_new_logprob(n_rows, total, strength, discount);
}
double pco_pitman_yor_score(double strength, double discount, const int64_t* counts, int n) {
  return pco::pitman_yor_score(strength, discount, counts, (size_t)n);
}
double pco_ess(const double* logw, int n) { return pco::effective_sample_size(logw, (size_t)n); }

/* ---- deterministic math / rng contract (include/) ----------------------- */
double pco_det_exp(double x) { return pclean_exp(x); }
double pco_det_log(double x) { return pclean_log(x); }
uint64_t pco_fixw(double d) { return pclean_fixw(d); }
void pco_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4) {
  pclean_u32x4 o = pclean_philox4x32_10(c0, c1, c2, c3, k0, k1);
  memcpy(out4, o.v, 16);
}
uint64_t pco_rand64(uint64_t seed, uint32_t row, uint32_t site, uint32_t particle, uint32_t sweep) {
  return pclean_rand64(seed, row, site, particle, sweep);
}

/* ---- world construction (host copies of what the product uploads) ------- */
pco::World* pco_world_create(void) { return new pco::World(); }
void pco_world_destroy(pco::World* w) { delete w; }
void pco_world_set_obs(pco::World* w, int n_rows, int n_cols, const int32_t* obs) {
  w->n_rows = n_rows;
  w->n_cols = n_cols;
  w->obs.assign(obs, obs + (size_t)n_rows * n_cols);
}
void pco_world_set_density(pco::World* w, int max_r, int max_d, int max_len, const double* nb, const double* logl) {
  w->max_r = max_r;
  w->max_d = max_d;
  w->max_len = max_len;
  w->nb.assign(nb, nb + (size_t)(max_r + 1) * (max_d + 1));
  w->logl.assign(logl, logl + max_len + 1);
}
void pco_world_set_pair(pco::World* w, int id, int n_obs, int n_lat, const uint16_t* d, const uint16_t* lat_len) {
  pco::OPair& p = w->pair[id];
  p.n_obs = n_obs;
  p.n_lat = n_lat;
  p.d.assign(d, d + (size_t)n_obs * n_lat);
  p.lat_len.assign(lat_len, lat_len + n_lat);
}
/* what the weight of a chosen ProposalDummyValue needs (sweep.h: dummy_correction) */
void pco_world_set_strings(pco::World* w, int n_strings, const uint16_t* sym, const int64_t* off) {
  w->off.assign(off, off + n_strings + 1);
  w->sym.assign(sym, sym + off[n_strings]);
}
void pco_world_set_pair_strings(pco::World* w, int id, int n_obs, const int32_t* obs_ids, int dist_mode) {
  w->pair[id].obs_ids.assign(obs_ids, obs_ids + n_obs);
  w->pair[id].dist_mode = dist_mode;
}
void pco_world_set_lm(pco::World* w, const double* init_p, const double* trans_p, const uint16_t* letter_sym) {
  w->lm_init.assign(init_p, init_p + 28);
  w->lm_trans.assign(trans_p, trans_p + 28 * 28);
  w->letter_sym.assign(letter_sym, letter_sym + 28);
}
void pco_world_set_table(pco::World* w, int id, int n_rows, int n_cols, const int32_t* cols, const int64_t* counts,
                         const double* logc_full, const double* logc_m1, const double* scal4) {
  pco::OTable& t = w->table[id];
  t.is_options = false;
  t.n_rows = n_rows;
  t.n_cols = n_cols;
  t.cols.assign(cols, cols + (size_t)n_rows * n_cols);
  t.counts.assign(counts, counts + n_rows);
  t.logc_full.assign(logc_full, logc_full + n_rows);
  t.logc_m1.assign(logc_m1, logc_m1 + n_rows);
  memcpy(t.scal, scal4, sizeof t.scal);
}
void pco_world_set_options(pco::World* w, int id, int n, const int32_t* values, const double* logp) {
  pco::OTable& t = w->table[id];
  t.is_options = true;
  t.n_rows = n;
  t.n_cols = 1;
  t.cols.assign(values, values + n);
  t.counts.assign(n, 1);
  t.logc_full.assign(logp, logp + n);
  t.logc_m1.clear();
}
void pco_world_set_block_group(pco::World* w, int id, int group) { w->block[id].group = group; }
void pco_world_set_ev_row_by_row(pco::World* w, int on) { w->ev_row_by_row = on != 0; }
void pco_world_set_fn(pco::World* w, int id, int n_a, int n_b, const int32_t* fn) {
  pco::OFn& f = w->fn[id];
  f.n_a = n_a;
  f.n_b = n_b;
  f.fn.assign(fn, fn + (size_t)n_a * n_b);
}
void pco_world_load_block(pco::World* w, int id, int n_nodes, const pclean_node* nodes, int n_terms,
                          const pclean_term* terms, int n_children, const int32_t* children, int n_colmap,
                          const int32_t* colmap, int n_ctx, const int32_t* ctx_src_block,
                          const int32_t* ctx_src_col) {
  pco::OBlock& b = w->block[id];
  b.nodes.assign(nodes, nodes + n_nodes);
  b.terms.assign(terms, terms + n_terms);
  b.children.assign(children, children + n_children);
  b.colmap.assign(colmap, colmap + n_colmap);
  b.gauss.clear();
  b.node_gauss.assign(n_nodes, -1);
  b.n_ctx = n_ctx;
  for (int s = 0; s < n_ctx; ++s) {
    b.ctx_src_block[s] = ctx_src_block[s];
    b.ctx_src_col[s] = ctx_src_col[s];
  }
}
/* CRP prior pieces computed the oracle's own way (densities.h), for checking
 * what the product uploads (pclean_get_table_priors). */
void pco_table_priors(int n_rows, const int64_t* counts, double strength, double discount, double* logc_full,
                      double* logc_m1, double* scal4) {
  int64_t total = 0, live = 0;
  for (int k = 0; k < n_rows; ++k) {
    total += counts[k];
    live += counts[k] > 0;
  }
  for (int k = 0; k < n_rows; ++k) {
    logc_full[k] = counts[k] > 0 ? std::log((double)counts[k] - discount) : pco::NEG_INF;
    logc_m1[k] = counts[k] > 1 ? std::log((double)(counts[k] - 1) - discount) : pco::NEG_INF;
  }
  scal4[0] = std::log((double)total + strength);
  scal4[1] = std::log((double)(total - 1) + strength);
  scal4[2] = std::log(strength + discount * (double)live);
  scal4[3] = std::log(strength + discount * (double)(live - 1));
}

/* ---- batched enumeration spec (enumerate.h) ------------------------------ */
int pco_score_node(const pco::World* w, int block_id, int node_id, int n_items, const int32_t* rows,
                   const int32_t* ctxv, const int32_t* excl, const double* snew, uint64_t seed, uint32_t sweep,
                   int n_draws, double* lse, double* scores, int32_t* draws) {
  return pco::score_node(*w, block_id, node_id, n_items, rows, ctxv, excl, snew, seed, sweep, n_draws, lse, scores,
                         draws);
}

/* Scores of every candidate of `node_id` (+ the new-row candidate, last, for reference slots) for one row, the
 * new-row branch evaluated recursively (process_plan!, proposal_compiler.jl:363-388).  Returns the log-marginal. */
double pco_eval_tree(const pco::World* w, int block_id, int node_id, int row, const int32_t* ctxv, int excl,
                     double* scores, int n_scores) {
  pco::RowCtx rc{w, block_id, row, ctxv, 0, 0, 0};
  std::vector<double> s;
  const double lse = pco::eval_tree(rc, node_id, excl, &s);
  for (int k = 0; k < n_scores && k < (int)s.size(); ++k) scores[k] = s[k];
  return lse;
}

/* The same for a latent-class work item: `node_id` of the latent plan `block_id` scored against the EVIDENCE SET
 * (observed rows ev_rows[0..n_ev), their per-row ctx values ev_ctx[n_ev][PCLEAN_MAX_CTX] or null) — what
 * sweep_latent() draws from (proposal_compiler.jl:306-350). */
double pco_eval_tree_ev(const pco::World* w, int block_id, int node_id, int n_ev, const int32_t* ev_rows,
                        const int32_t* ev_ctx, int excl, double* scores, int n_scores) {
  pco::Evidence ev;
  ev.rows = ev_rows;
  ev.ctx = ev_ctx;
  ev.n = n_ev;
  pco::RowCtx rc{w, block_id, 0, nullptr, 0, 0, 0, &ev, 0};
  std::vector<double> s;
  const double lse = pco::eval_tree(rc, node_id, excl, &s);
  for (int k = 0; k < n_scores && k < (int)s.size(); ++k) scores[k] = s[k];
  return lse;
}

/* ---- particle primitives -------------------------------------------------- */
void pco_maybe_resample(int n_rows, int P, const double* logw, int retain_first, uint64_t seed, uint32_t sweep,
                        uint32_t block, int64_t row_offset, int32_t* ancestors, double* logml_inc, double* ess) {
  for (int i = 0; i < n_rows; ++i) {
    std::vector<double> w(logw + (size_t)i * P, logw + (size_t)(i + 1) * P);
    std::vector<int> anc;
    double inc, e;
    pco::maybe_resample(w, retain_first != 0, seed, (uint32_t)(i + row_offset), sweep, block, anc, inc, &e);
    for (int p = 0; p < P; ++p) ancestors[(size_t)i * P + p] = anc[p];
    logml_inc[i] = inc;
    if (ess) ess[i] = e;
  }
}
void pco_final_choice(int n_rows, int P, const double* logw, int use_mh, int is_csmc, uint64_t seed, uint32_t sweep,
                      int64_t row_offset, int32_t* chosen, double* log_total) {
  for (int i = 0; i < n_rows; ++i) {
    std::vector<double> w(logw + (size_t)i * P, logw + (size_t)(i + 1) * P);
    double lt;
    chosen[i] = pco::final_choice(w, use_mh != 0, is_csmc != 0, seed, (uint32_t)(i + row_offset), sweep, &lt);
    if (log_total) log_total[i] = lt;
  }
}

/* ---- whole sweep, batched schedule (the GPU's parity target) --------------- */
static std::vector<pco::NewRow> g_new_rows;
static std::vector<int32_t> g_locals; /* [n_rows][n_blocks][2] of the last batched sweep */
static int g_locals_blocks = 0;
int pco_sweep_batched(const pco::World* w, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep, int n_blocks,
                      int64_t row_offset, const int32_t* cur, int32_t* choice, int32_t* chosen, double* logml) {
  const int N = w->n_rows;
  if (!cfg->use_dd_proposals)
    for (int b = 0; b < n_blocks; ++b)
      if (!pco::prior_mode_supported(w->block[b])) return -1; /* as the product: PCLEAN_ERR_ARG */
  g_new_rows.clear();
  g_locals.assign((size_t)N * n_blocks * 2, -1);
  g_locals_blocks = n_blocks;
  std::vector<int32_t> c(n_blocks), ch(n_blocks);
  for (int i = 0; i < N; ++i) {
    for (int b = 0; b < n_blocks; ++b) c[b] = cur[(size_t)b * N + i];
    int cp;
    double ml;
    pco::run_smc_row(*w, *cfg, seed, sweep, n_blocks, i, row_offset, c.data(), ch.data(), &cp, &ml, g_new_rows,
                     &g_locals[(size_t)i * n_blocks * 2]);
    for (int b = 0; b < n_blocks; ++b) choice[(size_t)b * N + i] = ch[b];
    if (chosen) chosen[i] = cp;
    if (logml) logml[i] = ml;
  }
  return 0;
}
/* the batched sweep with grouping + exact pruning on one thread (oracle/pruned.h): the same outputs as
 * pco_sweep_batched, bit for bit (tests/test_oracle_pruned.py); stats[9] = root evaluations asked, served by the memo,
 * candidates scored exactly, candidates pruned, new-row branches skipped, evaluated, child memo hits, misses, full fallbacks */
int pco_sweep_batched_pruned(const pco::World* w, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep, int n_blocks,
                             int64_t row_offset, int n_rows, const int32_t* cur, int32_t* choice, int32_t* chosen, double* logml,
                             uint64_t* stats) {
  const int N = w->n_rows;
  if (!cfg->use_dd_proposals || n_rows > N) return -1;
  g_new_rows.clear();
  g_locals.assign((size_t)N * n_blocks * 2, -1);
  g_locals_blocks = n_blocks;
  pco::Pruner pr(*w);
  std::vector<int32_t> c(n_blocks), ch(n_blocks);
  for (int i = 0; i < n_rows; ++i) {
    for (int b = 0; b < n_blocks; ++b) c[b] = cur[(size_t)b * N + i];
    int cp;
    double ml;
    pco::run_smc_row(*w, *cfg, seed, sweep, n_blocks, i, row_offset, c.data(), ch.data(), &cp, &ml, g_new_rows,
                     &g_locals[(size_t)i * n_blocks * 2], &pr);
    for (int b = 0; b < n_blocks; ++b) choice[(size_t)b * N + i] = ch[b];
    if (chosen) chosen[i] = cp;
    if (logml) logml[i] = ml;
  }
  if (stats) {
    const pco::PrunedStats& st = pr.st;
    const uint64_t v[9] = {st.roots, st.root_hits, st.cand_exact, st.cand_pruned, st.new_skipped, st.new_evaluated, st.child_hits,
                           st.child_miss, st.full_fallback};
    for (int i = 0; i < 9; ++i) stats[i] = v[i];
  }
  return 0;
}
/* sequential schedule = CPU baseline (bench.py cpu_baseline leg) */
int pco_sweep_sequential(pco::World* w, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep, int n_blocks,
                         int64_t row_offset, int32_t* cur, const double* py, int64_t* n_moved, int64_t* n_new) {
  pco::sweep_sequential(*w, *cfg, seed, sweep, n_blocks, row_offset, cur, py, n_moved, n_new);
  return 0;
}
int pco_sweep_latent(const pco::World* w, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep, int block_id,
                     int n_roots, const int32_t* roots, int n_items, const int32_t* keys, const int32_t* ev_off,
                     const int32_t* ev_rows, const int32_t* ev_ctx, const int32_t* excl, int32_t* chosen,
                     int32_t* vals) {
  pco::sweep_latent(*w, *cfg, seed, sweep, block_id, n_roots, roots, n_items, keys, ev_off, ev_rows, ev_ctx, excl,
                    chosen, vals);
  return 0;
}
/* current own choices of the rows of `block` ([n_rows][2], the sweep window's row index): the retained particle of a
 * prior-proposal sweep keeps them; n_rows = 0 clears */
void pco_world_set_cur_locals(pco::World* w, int block, int n_rows, const int32_t* locals) {
  w->cur_locals[block].assign(locals, locals + (size_t)n_rows * 2);
}
void pco_get_locals(int block, int n_rows, int32_t* out) {
  for (int i = 0; i < n_rows; ++i) {
    out[2 * i] = g_locals[((size_t)i * g_locals_blocks + block) * 2];
    out[2 * i + 1] = g_locals[((size_t)i * g_locals_blocks + block) * 2 + 1];
  }
}
void pco_world_set_numeric(pco::World* w, int n_rows, int n_cols, const double* x) {
  w->n_xcols = n_cols;
  w->xnum.assign(x, x + (size_t)n_rows * n_cols);
}
void pco_world_set_mean(pco::World* w, int id, int n, const double* mean) { w->mean[id].assign(mean, mean + n); }
void pco_world_set_gauss(pco::World* w, int block, int node, const pclean_gauss* g) {
  pco::OBlock& b = w->block[block];
  if (b.node_gauss.size() != b.nodes.size()) b.node_gauss.assign(b.nodes.size(), -1);
  b.node_gauss[node] = (int)b.gauss.size();
  b.gauss.push_back(*g);
}
void pco_world_set_options_cols(pco::World* w, int id, int n, int n_cols, const int32_t* cols, const double* logp) {
  pco::OTable& t = w->table[id];
  t.is_options = true;
  t.n_rows = n;
  t.n_cols = n_cols;
  t.cols.assign(cols, cols + (size_t)n * n_cols);
  t.counts.assign(n, 1);
  t.logc_full.assign(logp, logp + n);
  t.logc_m1.clear();
}
void pco_world_set_prob(pco::World* w, int n, const double* p) { w->prob.assign(p, p + n); }
void pco_world_load_score_block(pco::World* w, int id, int n_terms, const int32_t* obs_col, const int32_t* pair_table,
                                const int32_t* val_src, const int32_t* key_src, const int32_t* nopt_fn,
                                const int32_t* other_val, int prob_fn, const int32_t* pa, const int32_t* pb) {
  pco::OBlock& b = w->block[id];
  b = pco::OBlock();
  b.is_score = true;
  for (int t = 0; t < n_terms; ++t)
    b.score_terms.push_back({obs_col[t], pair_table[t], val_src[2 * t], val_src[2 * t + 1], key_src[2 * t],
                             key_src[2 * t + 1], nopt_fn[t], other_val[t]});
  b.prob_fn = prob_fn;
  b.prob_a_block = pa[0];
  b.prob_a_col = pa[1];
  b.prob_b_block = pb[0];
  b.prob_b_col = pb[1];
}
int pco_new_rows_count(int block) {
  int n = 0;
  for (auto& r : g_new_rows) n += r.block == block;
  return n;
}
void pco_new_rows_get(int block, int n_nodes, int32_t* rows, int32_t* vals) {
  int j = 0;
  for (auto& r : g_new_rows)
    if (r.block == block) {
      rows[j] = r.row;
      for (int k = 0; k < n_nodes; ++k) vals[(size_t)j * n_nodes + k] = r.vals[k];
      ++j;
    }
}

} /* extern "C" */

/* ---- random(dist, args...) (random.h) ------------------------------------ */
#include "random.h"
extern "C" {
void pco_random_add_typos(int n, const uint32_t* cp, const int64_t* off, int max_typos, uint64_t seed, uint32_t stream,
                          int stride, uint32_t* out_cp, int32_t* out_len) {
  for (int i = 0; i < n; ++i) {
    std::vector<uint32_t> w = pco::random_add_typos(cp + off[i], (int)(off[i + 1] - off[i]), max_typos, seed,
                                                    (uint32_t)i, stream, (size_t)stride);
    for (size_t k = 0; k < (size_t)stride; ++k) out_cp[(size_t)i * stride + k] = k < w.size() ? w[k] : 0u;
    out_len[i] = (int32_t)w.size();
  }
}
void pco_random_string_prior(int n, int min_len, int max_len, const double* init, const double* trans, uint64_t seed,
                             uint32_t stream, int stride, uint8_t* out, int32_t* out_len) {
  for (int i = 0; i < n; ++i) {
    std::vector<uint8_t> w = pco::random_string_prior(min_len, max_len, init, trans, seed, (uint32_t)i, stream);
    for (size_t k = 0; k < (size_t)stride; ++k) out[(size_t)i * stride + k] = k < w.size() ? w[k] : 0;
    out_len[i] = (int32_t)w.size();
  }
}
void pco_random_string_prior_at(int n, const uint64_t* seeds, const uint32_t* elems, int min_len, int max_len,
                                const double* init, const double* trans, uint32_t stream, int stride, uint8_t* out,
                                int32_t* out_len) {
  for (int i = 0; i < n; ++i) {
    std::vector<uint8_t> l = pco::random_string_prior(min_len, max_len, init, trans, seeds[i], elems[i], stream);
    out_len[i] = (int32_t)l.size();
    for (size_t k = 0; k < l.size() && (int)k < stride; ++k) out[(size_t)i * stride + k] = l[k];
  }
}
uint64_t pco_dummy_seed(uint64_t seed, uint32_t site, uint32_t particle, uint32_t sweep) {
  return pclean_dummy_seed(seed, site, particle, sweep);
}
void pco_random_categorical(int n, int n_options, const double* logp, uint64_t seed, uint32_t stream, int32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = pco::random_categorical(logp, n_options, seed, (uint32_t)i, stream);
}
void pco_random_normal(int n, const double* mean, double std, double fwd_scale, uint64_t seed, uint32_t stream,
                       double* out) {
  for (int i = 0; i < n; ++i) out[i] = pco::random_normal(mean[i], std, fwd_scale, seed, (uint32_t)i, stream);
}
void pco_random_maybe_swap(int n, const double* prob, const int32_t* n_options, uint64_t seed, uint32_t stream,
                           int32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = pco::random_maybe_swap(prob[i], n_options[i], seed, (uint32_t)i, stream);
}
void pco_random_time_prior(int n, uint64_t seed, uint32_t stream, int32_t* out) {
  for (int i = 0; i < n; ++i) pco::random_time_prior(seed, (uint32_t)i, stream, out + 3 * i);
}
}
