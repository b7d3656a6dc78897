/* oracle/enumerate.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the enumeration proposal that the reference generates per
 * (class, block, observed-key-set) in src/inference/proposal_compiler.jl:5-418
 * and re-scores in src/inference/block_proposal.jl:24-157, specialised to the
 * static plan IR of include/pclean_hip.h and to the fixed-point draw rule of
 * include/pclean_detmath.h ("batched-schedule" semantics: latent tables frozen,
 * the evidence row's own reference removed).
 *
 * score_node() = one ForeignKeyNode / RandomChoiceNode enumeration:
 *   proposal_compiler.jl:131-252 (FK: candidate loop 198-219, blind new-row
 *   branch 221-230, logsumexp + categorical 233-245) and 55-129 (discrete
 *   RandomChoiceNode: option loop 96-113, choice 115-127).
 * eval_tree() = process_plan! recursion (363-388): children of a "new row"
 *   are enumerated independently and their log-marginals added.
 */
#ifndef PCLEAN_ORACLE_ENUMERATE_H
#define PCLEAN_ORACLE_ENUMERATE_H

#include <cstdint>
#include <map>
#include <vector>

#include "../include/pclean_detmath.h"
#include "../include/pclean_hip.h"
#include "../include/pclean_philox.h"
#include "densities.h"

namespace pco {

struct OPair {
  int n_obs = 0, n_lat = 0;
  std::vector<uint16_t> d;        /* [n_obs][n_lat] */
  std::vector<uint16_t> lat_len;  /* [n_lat] */
  std::vector<int32_t> obs_ids;   /* pool string of every observed value (AddTypos tables; dummy-value weights) */
  int dist_mode = PCLEAN_DIST_DL; /* flavour of DamerauLevenshtein() the table was built with */
};
struct OTable {
  bool is_options = false;
  int n_rows = 0, n_cols = 0;
  std::vector<int32_t> cols;  /* [n_cols][n_rows] */
  std::vector<int64_t> counts;
  std::vector<double> logc_full, logc_m1;
  double scal[4] = {0, 0, 0, 0};
};
struct OFn {
  int n_a = 0, n_b = 0;
  std::vector<int32_t> fn;
};
struct OBlock {
  std::vector<pclean_node> nodes;
  std::vector<pclean_term> terms;
  std::vector<int32_t> children;
  std::vector<int32_t> colmap; /* pairs (child node, child col) per table column of each FK node */
  int n_ctx = 0;
  int ctx_src_block[PCLEAN_MAX_CTX] = {-1, -1};
  int ctx_src_col[PCLEAN_MAX_CTX] = {-1, -1};
  int group = -1;                  /* blocks of one model block share a group: no resampling between them */
  bool is_score = false;           /* block without a reference slot (flights Obs block 3) */
  struct ScoreTerm {
    int obs_col, pair_table, val_block, val_col, key_block, key_col, nopt_fn, other_val;
  };
  std::vector<ScoreTerm> score_terms;
  int prob_fn = -1, prob_a_block = -1, prob_a_col = -1, prob_b_block = -1, prob_b_col = -1;
  std::vector<pclean_gauss> gauss; /* Gaussian terms (experiments/rents/run.jl:19-25) */
  std::vector<int> node_gauss;     /* per node: index into gauss or -1 */
};

struct World {
  std::vector<double> prob; /* MaybeSwap error probabilities (ProbParameter values / constants) */
  int n_xcols = 0;
  std::vector<double> xnum;              /* numeric observed columns [n_xcols][n_rows] */
  std::vector<std::vector<double>> mean; /* MeanParameter value tables */
  int n_rows = 0, n_cols = 0;
  std::vector<int32_t> obs; /* [n_cols][n_rows] */
  /* AddTypos density pieces: handed over from the product (pclean_get_density_tables)
   * so both sides add bit-identical doubles; checked against densities.h in tests. */
  int max_r = 0, max_d = 0, max_len = 0;
  std::vector<double> nb, logl;
  std::vector<OPair> pair;
  std::vector<OTable> table;
  std::vector<OFn> fn;
  std::vector<OBlock> block;
  /* string pool + letter model of random(StringPrior): what the weight of a particle that chose a
   * ProposalDummyValue needs (block_proposal.jl:58-60; sweep.h: dummy_correction) */
  std::vector<uint16_t> sym;
  std::vector<int64_t> off;
  /* current own choices of the observed rows per block ([n_rows][2], local row index; empty: none): what the retained
   * particle of a prior-proposal sweep keeps (sweep.h: gauss_prior_term) */
  std::vector<std::vector<int32_t>> cur_locals;
  std::vector<double> lm_init, lm_trans; /* [28], [28*28] ([prev][next]) */
  std::vector<uint16_t> letter_sym;      /* [28] pool symbol of every alphabet letter, 0xFFFF = absent */
  /* evidence sets added ROW BY ROW (every term of an evidence row, then the next row): the reference's own order of
   * operations (one ExternalLikelihoodNode per referring row, proposal_compiler.jl:306-350).  Off: the aggregated
   * order the HIP path uses (per term, distinct (ctx, observed value) pairs x multiplicity).  A CPU test holds the
   * two within 1e-9 of each other (tests/test_literal_fixtures.py). */
  bool ev_row_by_row = false;
  World() : mean(64), pair(64), table(64), fn(64), block(16), cur_locals(16) {}
};

static const double HALF_LOG26 = 1.629048269010741; /* log(26)/2, add_typos.jl:63 */

inline double term_density(const World& w, const pclean_term& tm, const OPair& pt, int d, int val) {
  if (tm.dens_kind == PCLEAN_DENS_EQUAL) return d == 0 ? 0.0 : NEG_INF;
  if (tm.max_typos >= 0 && d > tm.max_typos) return IMPOSSIBLE; /* add_typos.jl:57-59 */
  const int L = pt.lat_len[val];
  const int r = (L + 4) / 5; /* ceil(length(word)/5), add_typos.jl:61 */
  double l = w.nb[(size_t)r * (w.max_d + 1) + d];
  l -= w.logl[L] * (double)d;  /* add_typos.jl:62 */
  l -= HALF_LOG26 * (double)d; /* add_typos.jl:63 */
  return l;
}

/* ---- MaybeSwap: maybe_swap.jl:13-28 (o < 0: explicitly missing observation) ------------------ */
inline double maybe_swap_term(const World& w, bool missing, bool same, bool val_in_options, int n_options, int pidx) {
  return maybe_swap_logdensity(missing, val_in_options, same, n_options, w.prob[pidx]);
}

/* ---- Gaussian term with enumerated locals (pclean_gauss): transformed_gaussian.jl:15-16,
 * add_noise.jl:7, choose_uniformly.jl:7-10; the locals are the inner enumeration loops of
 * proposal_compiler.jl:96-113 nested inside a candidate branch. ------------------------------ */
struct GaussCombos {
  int n = 0;
  double sc[16];
  int codes[16];
};
template <typename ValFn>
inline GaussCombos gauss_combo_scores(const World& w, const pclean_gauss& g, int row, const int32_t* evctx, ValFn val) {
  GaussCombos out;
  const double xv = w.xnum[(size_t)g.x_col * w.n_rows + row];
  const std::vector<double>& mu = w.mean[g.mean_table];
  const double log_sigma = std::log(g.sigma);
  int base = 0, lstride[2] = {0, 0};
  for (int d = 0; d < g.n_dims; ++d) {
    if (g.src_kind[d] == PCLEAN_GSRC_LOCAL)
      lstride[g.src[d]] = g.stride[d];
    else
      base += g.stride[d] * val(d);
  }
  int lo[2] = {0, 0}, hi[2] = {1, 1};
  double lp[2] = {0.0, 0.0};
  for (int l = 0; l < g.n_locals; ++l) {
    hi[l] = g.local_n[l];
    lp[l] = -std::log((double)g.local_n[l]);
    if (g.local_obs_col[l] >= 0) {
      const int v = w.obs[(size_t)g.local_obs_col[l] * w.n_rows + row];
      if (v >= 0) {
        lo[l] = v;
        hi[l] = v + 1;
      }
    }
  }
  for (int l0 = lo[0]; l0 < hi[0]; ++l0)
    for (int l1 = lo[1]; l1 < hi[1]; ++l1) {
      const int idx = base + lstride[0] * l0 + lstride[1] * l1;
      int u = 0;
      if (g.transform_src_kind == PCLEAN_GSRC_LOCAL)
        u = g.transform_src == 0 ? l0 : l1;
      else if (g.transform_src_kind == PCLEAN_GSRC_EVCTX)
        u = evctx[g.transform_src];
      double s = lp[0] + lp[1];
      /* (a non-linear Transformation: backward(x) and log|deriv(backward(x))| per row, pclean_gauss::t_x_col / t_lad_col) */
      const double bx = g.t_x_col[u] >= 0 ? w.xnum[(size_t)g.t_x_col[u] * w.n_rows + row] : xv * g.t_scale[u];
      const double z = (bx - mu[idx]) / g.sigma;
      s += -0.5 * z * z - log_sigma - 0.91893853320467274178;
      s -= g.t_lad_col[u] >= 0 ? w.xnum[(size_t)g.t_lad_col[u] * w.n_rows + row] : g.t_logabsderiv[u];
      out.sc[out.n] = s;
      out.codes[out.n] = l0 * 16 + l1;
      ++out.n;
    }
  return out;
}
inline double gauss_lse(const GaussCombos& c) {
  if (c.n == 1) return c.sc[0];
  double m = NEG_INF;
  for (int i = 0; i < c.n; ++i) m = c.sc[i] > m ? c.sc[i] : m;
  if (m == NEG_INF) return m;
  uint64_t U = 0;
  for (int i = 0; i < c.n; ++i) U += pclean_fixw(c.sc[i] - m);
  return pclean_lse_from_fix(m, U);
}

/* Evidence set of a latent-class work item: the observed rows that refer to the latent row
 * (ExternalLikelihoodNodes, proposal_compiler.jl:306-350; block_proposal.jl:119-155). */
struct Evidence {
  const int32_t* rows = nullptr; /* observed rows */
  const int32_t* ctx = nullptr;  /* [n][PCLEAN_MAX_CTX] per-evidence-row ctx, may be null */
  int n = 0;
};

/* Scores of all candidates (+ new-row candidate for FK nodes, last) of one
 * node for one work item.  out.size() == n_rows + (FK ? 1 : 0). */
enum ScoreMode { SCORE_FULL = 0, SCORE_PRIOR = 1, SCORE_TERMS = 2 };
/* SCORE_PRIOR: the prior part alone (CRP / option prior: what a prior proposal draws from, block_proposal.jl:42-56,
 * 68-84); SCORE_TERMS: the likelihood terms alone, added in the same order onto 0.0 (what p accumulates for a value
 * that was NOT enumerated: use_dd_proposals = false). */
inline void node_scores(const World& w, int block_id, int node_id, int row, const int32_t* ctxv, int excl,
                        double snew_in, std::vector<double>& out, const Evidence* ev = nullptr, int mode = SCORE_FULL) {
  const OBlock& b = w.block[block_id];
  const pclean_node& nd = b.nodes[node_id];
  const OTable& t = w.table[nd.table];
  const int n = t.n_rows;
  const bool fk = nd.kind == PCLEAN_NODE_FK;
  out.assign(n + (fk ? 1 : 0), NEG_INF);
  if (fk) {
    const bool excluded = excl >= 0;
    const bool deleted = excluded && t.counts[excl] <= 1; /* dependency_tracking.jl:189-201 */
    const double logden = excluded ? t.scal[1] : t.scal[0];
    for (int k = 0; k < n; ++k) {
      if (t.counts[k] == 0) continue;
      if (mode == SCORE_TERMS)
        out[k] = 0.0;
      else if (k == excl)
        out[k] = deleted ? NEG_INF : t.logc_m1[k] - logden;
      else
        out[k] = t.logc_full[k] - logden;
    }
    out[n] = mode == SCORE_TERMS ? 0.0 : ((deleted ? t.scal[3] : t.scal[2]) - logden) + snew_in;
  } else {
    for (int k = 0; k < n; ++k) out[k] = mode == SCORE_TERMS ? 0.0 : t.logc_full[k];
  }
  if (mode == SCORE_PRIOR) return;
  const pclean_gauss* gs = (node_id < (int)b.node_gauss.size() && b.node_gauss[node_id] >= 0)
                               ? &b.gauss[b.node_gauss[node_id]] : nullptr;
  if (ev && w.ev_row_by_row) {
    for (int e = 0; e < ev->n; ++e) {
      const int er = ev->rows[e];
      const int32_t* ecx = ev->ctx ? ev->ctx + (size_t)e * PCLEAN_MAX_CTX : nullptr;
      for (int ti = 0; ti < nd.n_terms; ++ti) {
        const pclean_term& tm = b.terms[nd.term_begin + ti];
        const int o = w.obs[(size_t)tm.obs_col * w.n_rows + er];
        const bool ev_ctx_term = tm.ctx_slot >= 0 && tm.ctx_mode != 0;
        const int ec = ev_ctx_term ? ecx[tm.ctx_slot] : 0;
        for (int k = 0; k < n; ++k) {
          if (fk && t.counts[k] == 0) continue;
          if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP) {
            const OPair& ptm = w.pair[tm.pair_table];
            const int v2 = t.cols[(size_t)tm.cand_col * n + k];
            const int c = tm.ctx_mode == 0 ? ctxv[tm.ctx_slot] : ec;
            const bool same = o >= 0 && ptm.d[(size_t)o * ptm.n_lat + v2] == 0;
            out[k] += maybe_swap_term(w, o < 0, same, v2 < tm.fn_table, t.cols[(size_t)tm.max_typos * n + k], c);
            continue;
          }
          if (o < 0) continue;
          const OPair& pt = w.pair[tm.pair_table];
          int val = t.cols[(size_t)tm.cand_col * n + k];
          if (tm.ctx_slot >= 0) {
            const OFn& f = w.fn[tm.fn_table];
            const int c = tm.ctx_mode == 0 ? ctxv[tm.ctx_slot] : ec;
            val = tm.ctx_mode == 2 ? f.fn[(size_t)val * f.n_b + c] : f.fn[(size_t)c * f.n_b + val];
          }
          out[k] += term_density(w, tm, pt, pt.d[(size_t)o * pt.n_lat + val], val);
        }
      }
      if (gs && w.xnum[(size_t)gs->x_col * w.n_rows + er] == w.xnum[(size_t)gs->x_col * w.n_rows + er])
        for (int k = 0; k < n; ++k) {
          if ((fk && t.counts[k] == 0) || !(out[k] > NEG_INF)) continue;
          out[k] += gauss_lse(gauss_combo_scores(w, *gs, er, ecx, [&](int d) -> int {
            switch (gs->src_kind[d]) {
              case PCLEAN_GSRC_CAND: return t.cols[(size_t)gs->src[d] * n + k];
              case PCLEAN_GSRC_OBS: return w.obs[(size_t)gs->src[d] * w.n_rows + er];
              case PCLEAN_GSRC_ITEMCTX: return ctxv[gs->src[d]];
              default: return ecx[gs->src[d]];
            }
          }));
        }
    }
    return;
  }
  if (ev) {
    /* Evidence sets: terms in plan order; per term the distinct (ctx value, observed value) pairs of the
     * evidence rows in ascending order (missing observation = -1 first), each adding multiplicity x density;
     * then the Gaussian terms row by row in list order.  (The reference adds row by row in Dict order,
     * proposal_compiler.jl:306-350 — the sum is the same up to fp64 rounding.) */
    for (int ti = 0; ti < nd.n_terms; ++ti) {
      const pclean_term& tm = b.terms[nd.term_begin + ti];
      std::map<std::pair<int, int>, int64_t> agg;
      const bool ev_ctx_term = tm.ctx_slot >= 0 && tm.ctx_mode != 0;
      for (int e = 0; e < ev->n; ++e) {
        const int o = w.obs[(size_t)tm.obs_col * w.n_rows + ev->rows[e]];
        const int c = ev_ctx_term ? ev->ctx[(size_t)e * PCLEAN_MAX_CTX + tm.ctx_slot] : 0;
        agg[std::make_pair(c, o)] += 1;
      }
      for (const auto& kv : agg) {
        const int ec = kv.first.first, o = kv.first.second;
        const double mult = (double)kv.second;
        for (int k = 0; k < n; ++k) {
          if (fk && t.counts[k] == 0) continue;
          if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP) {
            const OPair& ptm = w.pair[tm.pair_table];
            const int v2 = t.cols[(size_t)tm.cand_col * n + k];
            const int c = tm.ctx_mode == 0 ? ctxv[tm.ctx_slot] : ec;
            const bool same = o >= 0 && ptm.d[(size_t)o * ptm.n_lat + v2] == 0;
            out[k] += mult * maybe_swap_term(w, o < 0, same, v2 < tm.fn_table, t.cols[(size_t)tm.max_typos * n + k], c);
            continue;
          }
          if (o < 0) continue;
          const OPair& pt = w.pair[tm.pair_table];
          int val = t.cols[(size_t)tm.cand_col * n + k];
          if (tm.ctx_slot >= 0) {
            const OFn& f = w.fn[tm.fn_table];
            const int c = tm.ctx_mode == 0 ? ctxv[tm.ctx_slot] : ec;
            val = tm.ctx_mode == 2 ? f.fn[(size_t)val * f.n_b + c] : f.fn[(size_t)c * f.n_b + val];
          }
          out[k] += mult * term_density(w, tm, pt, pt.d[(size_t)o * pt.n_lat + val], val);
        }
      }
    }
    if (gs)
      for (int k = 0; k < n; ++k) {
        if (fk && t.counts[k] == 0) continue;
        for (int e = 0; e < ev->n; ++e) {
          const int er = ev->rows[e];
          if (!(out[k] > NEG_INF) || w.xnum[(size_t)gs->x_col * w.n_rows + er] != w.xnum[(size_t)gs->x_col * w.n_rows + er])
            continue;
          const int32_t* ec = ev->ctx ? ev->ctx + (size_t)e * PCLEAN_MAX_CTX : nullptr;
          out[k] += gauss_lse(gauss_combo_scores(w, *gs, er, ec, [&](int d) -> int {
            switch (gs->src_kind[d]) {
              case PCLEAN_GSRC_CAND: return t.cols[(size_t)gs->src[d] * n + k];
              case PCLEAN_GSRC_OBS: return w.obs[(size_t)gs->src[d] * w.n_rows + er];
              case PCLEAN_GSRC_ITEMCTX: return ctxv[gs->src[d]];
              default: return ec[gs->src[d]];
            }
          }));
        }
      }
    return;
  }
  for (int ti = 0; ti < nd.n_terms; ++ti) {
    const pclean_term& tm = b.terms[nd.term_begin + ti];
    const int o = w.obs[(size_t)tm.obs_col * w.n_rows + row];
    if (o < 0) continue; /* explicitly missing observation: add_typos.jl:51-53 */
    const OPair& pt = w.pair[tm.pair_table];
    for (int k = 0; k < n; ++k) {
      if (fk && t.counts[k] == 0) continue; /* free slot of the latent table */
      int val = t.cols[(size_t)tm.cand_col * n + k];
      if (tm.ctx_slot >= 0) {
        const OFn& f = w.fn[tm.fn_table];
        val = f.fn[(size_t)ctxv[tm.ctx_slot] * f.n_b + val];
      }
      const int d = pt.d[(size_t)o * pt.n_lat + val];
      out[k] += term_density(w, tm, pt, d, val);
    }
  }
  /* (SCORE_TERMS on a single row = a prior-proposal sweep of the observed class: the Gaussian term is scored by
   * gauss_prior_term at the sampled own choices, sweep.h — not marginalised here) */
  if (gs && mode != SCORE_TERMS && w.xnum[(size_t)gs->x_col * w.n_rows + row] == w.xnum[(size_t)gs->x_col * w.n_rows + row])
    for (int k = 0; k < n; ++k) {
      if (!(out[k] > NEG_INF)) continue;
      out[k] += gauss_lse(gauss_combo_scores(w, *gs, row, nullptr, [&](int d) -> int {
        switch (gs->src_kind[d]) {
          case PCLEAN_GSRC_CAND: return t.cols[(size_t)gs->src[d] * n + k];
          case PCLEAN_GSRC_OBS: return w.obs[(size_t)gs->src[d] * w.n_rows + row];
          default: return ctxv[gs->src[d]];
        }
      }));
    }
}

struct FixSum {
  double m;
  uint64_t U;
};
inline FixSum fix_sum(const std::vector<double>& s) {
  FixSum f{NEG_INF, 0};
  for (double v : s)
    if (v > f.m) f.m = v;
  if (f.m == NEG_INF) return f;
  for (double v : s) f.U += pclean_fixw(v - f.m);
  return f;
}
/* min{k : u_0+..+u_k > x}; falls back to the last candidate when U == 0. */
inline int fix_draw(const std::vector<double>& s, const FixSum& f, uint64_t R) {
  const int n = (int)s.size();
  if (f.U == 0) return n - 1;
  const uint64_t x = pclean_mulhi64(R, f.U);
  uint64_t acc = 0;
  for (int k = 0; k < n; ++k) {
    acc += pclean_fixw(s[k] - f.m);
    if (acc > x) return k;
  }
  return n - 1;
}

inline int score_node(const World& w, int block_id, int node_id, int n_items, const int32_t* rows,
                      const int32_t* ctxv, const int32_t* excl, const double* snew, uint64_t seed, uint32_t sweep,
                      int n_draws, double* lse, double* scores, int32_t* draws) {
  const OBlock& b = w.block[block_id];
  const pclean_node& nd = b.nodes[node_id];
  const OTable& t = w.table[nd.table];
  const bool fk = nd.kind == PCLEAN_NODE_FK;
  const int nc = t.n_rows + (fk ? 1 : 0);
  std::vector<double> s;
  for (int it = 0; it < n_items; ++it) {
    node_scores(w, block_id, node_id, rows[it], ctxv ? ctxv + (size_t)it * PCLEAN_MAX_CTX : nullptr,
                excl ? excl[it] : -1, snew ? snew[it] : NEG_INF, s);
    FixSum f = fix_sum(s);
    if (lse) lse[it] = pclean_lse_from_fix(f.m, f.U);
    if (scores)
      for (int k = 0; k < nc; ++k) scores[(size_t)it * nc + k] = s[k];
    for (int j = 0; j < n_draws; ++j) {
      uint64_t R = pclean_rand64(seed, (uint32_t)rows[it], PCLEAN_SITE_NODE(block_id, node_id), (uint32_t)j, sweep);
      int k = fix_draw(s, f, R);
      draws[(size_t)it * n_draws + j] = (fk && k == t.n_rows) ? PCLEAN_CHOICE_NEW : k;
    }
  }
  return 0;
}

} /* namespace pco */
#endif
