/* oracle/sweep.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of one rejuvenation sweep over the rows of the observed
 * class: pgibbs_sweep!'s inner loop (src/inference/inference.jl:66-79) calling
 * run_smc! (src/inference/row_inference.jl:108-187) with
 *   - block proposals by enumeration (proposal_compiler.jl, restated in enumerate.h),
 *   - the incremental weight p - q of a fully enumerated block = its log marginal
 *     (block_proposal.jl:160-190),
 *   - maybe_resample between blocks (row_inference.jl:87-105,152-155),
 *   - final Categorical / MH choice (row_inference.jl:158-165),
 *   - return value log_ml + log_total - log P (186).
 *
 * Two schedules:
 *   batched     — every row is updated against the latent tables as given (what
 *                 the GPU computes; bit-exact parity target);
 *   sequential  — row i sees the reference counts left by rows < i, as the
 *                 reference does (row_inference.jl:169-185); used as the
 *                 single-thread CPU baseline.  New-row creation / deletion inside
 *                 the sequential sweep is applied to counts only when the
 *                 chosen referent exists.
 */
#ifndef PCLEAN_ORACLE_SWEEP_H
#define PCLEAN_ORACLE_SWEEP_H

#include <cmath>
#include <type_traits>
#include <vector>

#include "enumerate.h"
#include "random.h"

namespace pco {

struct RowCtx {
  const World* w;
  int block;
  int row;
  const int32_t* ctxv;
  uint64_t seed;
  uint32_t sweep;
  int64_t row_offset;
  const Evidence* ev = nullptr; /* latent-class items */
  int64_t rng_row = -1;         /* RNG row id override (latent row key) */
};

inline int child_excl_of(const World& w, const OBlock& b, int parent_node, int child_node, int parent_excl) {
  const pclean_node& pn = b.nodes[parent_node];
  const pclean_node& cn = b.nodes[child_node];
  if (cn.kind != PCLEAN_NODE_FK || parent_excl < 0) return -1;
  const OTable& t = w.table[pn.table];
  if (t.counts[parent_excl] > 1) return -1;
  return t.cols[(size_t)cn.parent_fk_col * t.n_rows + parent_excl];
}

/* scores of a node with its new-row branch evaluated recursively (process_plan!) */
inline double eval_tree(const RowCtx& rc, int node, int excl, std::vector<double>* scores_out);

inline double snew_of(const RowCtx& rc, int node, int excl) {
  const OBlock& b = rc.w->block[rc.block];
  const pclean_node& nd = b.nodes[node];
  double snew = 0.0;
  for (int c = 0; c < nd.n_children; ++c) {
    const int cid = b.children[nd.child_begin + c];
    snew += eval_tree(rc, cid, child_excl_of(*rc.w, b, node, cid, excl), nullptr);
  }
  return snew;
}

inline double eval_tree(const RowCtx& rc, int node, int excl, std::vector<double>* scores_out) {
  const OBlock& b = rc.w->block[rc.block];
  const pclean_node& nd = b.nodes[node];
  double snew = NEG_INF;
  if (nd.kind == PCLEAN_NODE_FK) snew = snew_of(rc, node, excl);
  std::vector<double> s;
  node_scores(*rc.w, rc.block, node, rc.row, rc.ctxv, excl, snew, s, rc.ev);
  FixSum f = fix_sum(s);
  if (scores_out) scores_out->swap(s);
  return pclean_lse_from_fix(f.m, f.U);
}

/* draw sub-choices of a freshly proposed row of `node`'s table */
inline void sample_new(const RowCtx& rc, int node, int excl, uint32_t particle, int32_t* vals) {
  const OBlock& b = rc.w->block[rc.block];
  const pclean_node& nd = b.nodes[node];
  const uint32_t rr = rc.rng_row >= 0 ? (uint32_t)rc.rng_row : (uint32_t)((int64_t)rc.row + rc.row_offset);
  for (int c = 0; c < nd.n_children; ++c) {
    const int cid = b.children[nd.child_begin + c];
    const pclean_node& cn = b.nodes[cid];
    const int cex = child_excl_of(*rc.w, b, node, cid, excl);
    std::vector<double> s;
    eval_tree(rc, cid, cex, &s);
    FixSum f = fix_sum(s);
    const int k = fix_draw(s, f, pclean_rand64(rc.seed, rr, PCLEAN_SITE_NODE(rc.block, cid), particle, rc.sweep));
    const int n = rc.w->table[cn.table].n_rows;
    if (cn.kind == PCLEAN_NODE_FK && k == n) {
      vals[cid] = PCLEAN_CHOICE_NEW;
      sample_new(rc, cid, cex, particle, vals);
    } else {
      vals[cid] = k;
    }
  }
}

/* ---- use_dd_proposals = false (block_proposal.jl:168): nothing is enumerated against the observations.
 * propose_non_enumerable! (24-157) samples every reference slot from its CRP prior (68-84) and every unobserved
 * discrete choice of a new row from its prior proposal (42-56: q_cont += lprobs[chosen], p += logdensity of the same
 * value — they cancel; a chosen ProposalDummyValue leaves -log(dummy mass) and gets random(dist), 58-60), then
 * p accumulates the log-density of every observed choice given the sampled values (62-64).  The particle's weight
 * increment is therefore the likelihood of its sampled sub-tree: AddTypos observations (plain or through a JuliaNode),
 * noise-free observations (equality: 0 or -inf), MaybeSwap observations of the referring rows (latent classes); a
 * scoring block has nothing to propose and scores as always (62-64).  A block with a Gaussian term on its slot
 * (experiments/rents/run.jl:19-25): the own choices the data-driven proposal enumerates inside the candidate branch are
 * sampled from their priors as well (gauss_prior_term); the Gaussian evidence of a latent class, whose rows' own choices are
 * given, is a likelihood term like any other.  Refused: Gaussian terms on other nodes or with context sources. ---- */
inline bool prior_mode_supported(const OBlock& b) {
  for (const pclean_term& tm : b.terms)
    if (tm.dens_kind != PCLEAN_DENS_ADD_TYPOS && tm.dens_kind != PCLEAN_DENS_EQUAL && tm.dens_kind != PCLEAN_DENS_MAYBE_SWAP)
      return false;
  /* (observed class) the slot's Gaussian term is scored once per particle by gauss_prior_term, whatever was sampled below
   * the slot: the copies of the term on the nodes of a new row's choices — there for the enumeration — are not looked at */
  const bool root_g = !b.node_gauss.empty() && b.node_gauss[0] >= 0;
  for (size_t i = 0; i < b.node_gauss.size(); ++i) {
    if (b.node_gauss[i] < 0) continue;
    const pclean_gauss& g = b.gauss[b.node_gauss[i]];
    if (i != 0) {
      if (g.n_locals > 0 && !root_g) return false;
      continue; /* n_locals == 0: the evidence of a latent class, a likelihood term like any other */
    }
    for (int d = 0; d < g.n_dims; ++d)
      if (g.src_kind[d] != PCLEAN_GSRC_CAND && g.src_kind[d] != PCLEAN_GSRC_OBS && g.src_kind[d] != PCLEAN_GSRC_LOCAL) return g.n_locals == 0;
    if (g.transform_src_kind == PCLEAN_GSRC_EVCTX) return g.n_locals == 0;
  }
  return true;
}

/* Prior proposal of the own choices of a block with a Gaussian term (block_proposal.jl:42-56,62-64): an OBSERVED own
 * choice is scored (its ChooseUniformly density enters p), an unobserved one is sampled from its prior — uniform, the
 * proposal's and the model's densities cancel — by particle `particle` (the retained particle keeps the row's current
 * ones, `keep`), and the observed number is scored given the sampled values.  Returns what p accumulates; loc[2] = the
 * particle's own choices.  Uniform draw of choice l: floor(R * n_l / 2^64), R from the stream of
 * (site PCLEAN_SITE_LOCALS(block), particle | (l + 1) << 16).  Additions in this order onto 0.0: observed choices'
 * densities (choice 0, then 1), the Normal log-density, minus the transformation's log |derivative|. */
template <typename ValFn>
inline double gauss_prior_term(const World& w, const pclean_gauss& g, int block, int row, uint32_t rr, uint32_t particle,
                               uint64_t seed, uint32_t sweep, const int32_t* keep, ValFn val, int32_t* loc) {
  int l[2] = {0, 0};
  double s = 0.0;
  loc[0] = loc[1] = -1;
  for (int i = 0; i < g.n_locals; ++i) {
    const int o = g.local_obs_col[i] >= 0 ? w.obs[(size_t)g.local_obs_col[i] * w.n_rows + row] : -1;
    if (o >= 0) {
      l[i] = o;
      s += -std::log((double)g.local_n[i]);
    } else if (keep && keep[i] >= 0) {
      l[i] = keep[i];
    } else {
      l[i] = (int)pclean_mulhi64(pclean_rand64(seed, rr, PCLEAN_SITE_LOCALS(block), particle | ((uint32_t)(i + 1) << 16), sweep),
                                 (uint64_t)g.local_n[i]);
    }
    loc[i] = l[i];
  }
  const double xv = w.xnum[(size_t)g.x_col * w.n_rows + row];
  if (xv != xv) return s; /* missing number: nothing to score */
  const std::vector<double>& mu = w.mean[g.mean_table];
  int idx = 0;
  for (int d = 0; d < g.n_dims; ++d) idx += g.stride[d] * (g.src_kind[d] == PCLEAN_GSRC_LOCAL ? l[g.src[d]] : val(d));
  const int u = g.transform_src_kind == PCLEAN_GSRC_LOCAL ? l[g.transform_src] : 0;
  const double bx = g.t_x_col[u] >= 0 ? w.xnum[(size_t)g.t_x_col[u] * w.n_rows + row] : xv * g.t_scale[u];
  const double z = (bx - mu[idx]) / g.sigma;
  s += -0.5 * z * z - std::log(g.sigma) - 0.91893853320467274178;
  s -= g.t_lad_col[u] >= 0 ? w.xnum[(size_t)g.t_lad_col[u] * w.n_rows + row] : g.t_logabsderiv[u];
  return s;
}

/* prior draws of the sub-choices of a freshly proposed row of `node`'s table */
inline void sample_new_prior(const RowCtx& rc, int node, int excl, uint32_t particle, int32_t* vals) {
  const OBlock& b = rc.w->block[rc.block];
  const pclean_node& nd = b.nodes[node];
  const uint32_t rr = rc.rng_row >= 0 ? (uint32_t)rc.rng_row : (uint32_t)((int64_t)rc.row + rc.row_offset);
  for (int c = 0; c < nd.n_children; ++c) {
    const int cid = b.children[nd.child_begin + c];
    const pclean_node& cn = b.nodes[cid];
    const int cex = child_excl_of(*rc.w, b, node, cid, excl);
    std::vector<double> s;
    node_scores(*rc.w, rc.block, cid, rc.row, rc.ctxv, cex, 0.0, s, rc.ev, SCORE_PRIOR);
    FixSum f = fix_sum(s);
    const int k = fix_draw(s, f, pclean_rand64(rc.seed, rr, PCLEAN_SITE_NODE(rc.block, cid), particle, rc.sweep));
    const int n = rc.w->table[cn.table].n_rows;
    if (cn.kind == PCLEAN_NODE_FK && k == n) {
      vals[cid] = PCLEAN_CHOICE_NEW;
      sample_new_prior(rc, cid, cex, particle, vals);
    } else {
      vals[cid] = k;
    }
  }
}

/* likelihood of the observations below `node` given its sampled choice (k, or NEW with the sub-choices in vals):
 * terms of the chosen candidate, or — new row — of its children, added in plan order onto 0.0 */
inline double subtree_terms(const RowCtx& rc, int node, int choice, int excl, const int32_t* vals) {
  const OBlock& b = rc.w->block[rc.block];
  const pclean_node& nd = b.nodes[node];
  if (choice >= 0) {
    std::vector<double> s;
    node_scores(*rc.w, rc.block, node, rc.row, rc.ctxv, excl, 0.0, s, rc.ev, SCORE_TERMS);
    return s[choice];
  }
  double acc = 0.0;
  for (int c = 0; c < nd.n_children; ++c) {
    const int cid = b.children[nd.child_begin + c];
    acc += subtree_terms(rc, cid, vals[cid], child_excl_of(*rc.w, b, node, cid, excl), vals);
  }
  return acc;
}

inline int32_t resolve_new_value(const World& w, const OBlock& b, int node, int col, const int32_t* vals) {
  for (int depth = 0; depth < 16; ++depth) {
    const int cn = b.colmap[2 * (b.nodes[node].colmap_begin + col)];
    const int cc = b.colmap[2 * (b.nodes[node].colmap_begin + col) + 1];
    if (cn < 0) return -1;
    const int choice = vals[cn];
    const OTable& t = w.table[b.nodes[cn].table];
    if (b.nodes[cn].kind == PCLEAN_NODE_LEAF) return t.cols[choice];
    if (choice >= 0) return t.cols[(size_t)cc * t.n_rows + choice];
    node = cn;
    col = cc;
  }
  return -1;
}

/* Weight correction of a fresh particle whose enumerated proposal chose a ProposalDummyValue somewhere in its new
 * row (block_proposal.jl:58-60 after proposal_compiler.jl:96-127).  The enumeration scored the dummy option with
 * its prior mass log(dm) and the PLACEHOLDER string's likelihood, and q_disc holds exactly that; propose_non_enumerable!
 * then replaces the placeholder by random(node.dist, ...) — no density of its own enters p — and the observations
 * below the node are scored on the DRAWN string.  So  p - q_disc = block marginal + sum over chosen dummies of
 *     - log(dm) + sum over the node's AddTypos observations [ logdensity(obs | drawn) - logdensity(obs | placeholder) ].
 * The string is drawn with the private stream pclean_dummy_seed(seed, site of the node, particle, sweep) at the row's
 * global id; the host draws it again with the same key when the particle is chosen (inference.resample_dummies).
 * Supported: StringPrior placeholders under plain AddTypos observations; TimePrior placeholders have no such
 * observation in their own block, their correction is the -log(dm) term.  Terms through a JuliaNode (ctx) keep the
 * placeholder's likelihood. */
inline double dummy_correction(const RowCtx& rc, const int32_t* vals, uint32_t particle) {
  const World& w = *rc.w;
  const OBlock& b = w.block[rc.block];
  const uint32_t rr = rc.rng_row >= 0 ? (uint32_t)rc.rng_row : (uint32_t)((int64_t)rc.row + rc.row_offset);
  double corr = 0.0;
  for (size_t node = 0; node < b.nodes.size(); ++node) {
    const pclean_node& nd = b.nodes[node];
    if (nd.kind != PCLEAN_NODE_LEAF || nd.dummy_value == 0) continue;
    const int k = vals[node];
    if (k < 0) continue; /* -2: not sampled (its slot joined an existing row) */
    const OTable& t = w.table[nd.table];
    if (t.cols[k] != nd.dummy_value - 1) continue; /* an atom was chosen */
    double c = -t.logc_full[k];
    if ((nd.dummy_spec & 0xff) == PCLEAN_DUMMY_STRING_PRIOR) {
      std::vector<uint16_t> drawn;
      bool have = false;
      for (int ti = 0; ti < nd.n_terms; ++ti) {
        const pclean_term& tm = b.terms[nd.term_begin + ti];
        if (tm.dens_kind != PCLEAN_DENS_ADD_TYPOS || tm.ctx_slot >= 0) continue;
        const int o = w.obs[(size_t)tm.obs_col * w.n_rows + rc.row];
        if (o < 0) continue;
        const OPair& pt = w.pair[tm.pair_table];
        if (!have) {
          const int mn = (nd.dummy_spec >> 8) & 0xff, mx = (nd.dummy_spec >> 16) & 0xff;
          const uint64_t key = pclean_dummy_seed(rc.seed, PCLEAN_SITE_NODE(rc.block, (int)node), particle, rc.sweep);
          std::vector<uint8_t> letters = random_string_prior(mn, mx, w.lm_init.data(), w.lm_trans.data(), key, rr, 0u);
          for (uint8_t l : letters) drawn.push_back(w.letter_sym[l]);
          have = true;
        }
        const int sid = pt.obs_ids[o];
        const uint16_t* os = w.sym.data() + w.off[sid];
        const int ol = (int)(w.off[sid + 1] - w.off[sid]);
        const int d = pt.dist_mode == PCLEAN_DIST_OSA ? osa_distance(os, ol, drawn.data(), (int)drawn.size())
                                                      : dl_distance(os, ol, drawn.data(), (int)drawn.size());
        /* term_density() on (d, word length of the drawn string): add_typos.jl:57-63 */
        double l;
        const int L = (int)drawn.size();
        if (tm.max_typos >= 0 && d > tm.max_typos) {
          l = IMPOSSIBLE;
        } else {
          l = w.nb[(size_t)((L + 4) / 5) * (w.max_d + 1) + d];
          l -= w.logl[L] * (double)d;
          l -= HALF_LOG26 * (double)d;
        }
        const int ph = nd.dummy_value - 1;
        c += l - term_density(w, tm, pt, pt.d[(size_t)o * pt.n_lat + ph], ph);
      }
    }
    corr += c;
  }
  return corr;
}

struct NewRow {
  int block, row;
  std::vector<int32_t> vals;
};

struct PartW {
  double m;
  uint64_t U;
  std::vector<uint64_t> u;
};
inline PartW part_weights(const std::vector<double>& w) {
  PartW p{NEG_INF, 0, std::vector<uint64_t>(w.size())};
  for (double v : w) p.m = v > p.m ? v : p.m;
  for (size_t i = 0; i < w.size(); ++i) {
    p.u[i] = p.m == NEG_INF ? 0 : pclean_fixw(w[i] - p.m);
    p.U += p.u[i];
  }
  return p;
}
inline int part_pick(const PartW& p, uint64_t R) {
  const int P = (int)p.u.size();
  if (p.U == 0) return P - 1;
  const uint64_t x = pclean_mulhi64(R, p.U);
  uint64_t acc = 0;
  for (int i = 0; i < P; ++i) {
    acc += p.u[i];
    if (acc > x) return i;
  }
  return P - 1;
}

/* maybe_resample (row_inference.jl:87-105) in the fixed-point contract */
inline bool maybe_resample(const std::vector<double>& w, bool retain_first, uint64_t seed, uint32_t rr, uint32_t sweep,
                           uint32_t block, std::vector<int>& anc, double& inc, double* ess_out) {
  const int P = (int)w.size();
  PartW pw = part_weights(w);
  const double Ud = (double)pw.U;
  double s2 = 0.0;
  for (int p = 0; p < P; ++p) s2 += (double)pw.u[p] * (double)pw.u[p];
  const double ess = pw.U ? (Ud * Ud) / s2 : 0.0;
  if (ess_out) *ess_out = ess;
  anc.resize(P);
  if (ess < (double)P / 2.0) {
    for (int p = 0; p < P; ++p)
      anc[p] = (p == 0 && retain_first) ? 0
                                        : part_pick(pw, pclean_rand64(seed, rr, PCLEAN_SITE_RESAMPLE(block), (uint32_t)p,
                                                                      sweep));
    inc = pclean_lse_from_fix(pw.m, pw.U) - pclean_log((double)P);
    return true;
  }
  for (int p = 0; p < P; ++p) anc[p] = p;
  inc = 0.0;
  return false;
}

inline int final_choice(const std::vector<double>& w, bool use_mh, bool csmc, uint64_t seed, uint32_t rr, uint32_t sweep,
                        double* log_total) {
  const int P = (int)w.size();
  PartW pw = part_weights(w);
  if (log_total) *log_total = pclean_lse_from_fix(pw.m, pw.U);
  if (use_mh && csmc && P >= 2) { /* row_inference.jl:161-162 */
    const double Ud = (double)pw.U;
    const double w0 = (double)pw.u[0] / Ud, w1 = (double)pw.u[1] / Ud;
    double ratio = w1 / (1e-10 + w0);
    if (ratio > 1.0) ratio = 1.0;
    const double x = pclean_u01(pclean_rand64(seed, rr, PCLEAN_SITE_MH, 0u, sweep));
    return (pw.U != 0 && x < ratio) ? 1 : 0;
  }
  return part_pick(pw, pclean_rand64(seed, rr, PCLEAN_SITE_FINAL, 0u, sweep));
}

/* One row of run_smc! (CSMC when cur >= 0).  Outputs the chosen referent per block and
 * the sampled contents of new rows.  pr (oracle/pruned.h: Pruner): the data-driven root enumerations go through its
 * grouped + pruned evaluation — same results, less work (the CPU baseline's second leg). */
struct NoPruner {};
template <typename PR = NoPruner>
inline void run_smc_row(const World& w, const pclean_infer_config& cfg, uint64_t seed, uint32_t sweep, int n_blocks,
                        int row, int64_t row_offset, const int32_t* cur /*[n_blocks]*/, int32_t* choice /*[n_blocks]*/,
                        int32_t* chosen_particle, double* logml, std::vector<NewRow>& new_rows,
                        int32_t* locals_out = nullptr /*[n_blocks][2]*/, PR* pr = nullptr) {
  std::vector<std::vector<int32_t>> plocals(n_blocks); /* prior proposals: every particle's own choices [P][2] */
  const bool use_mh = cfg.use_mh_instead_of_pg != 0;
  const int P = use_mh ? 2 : cfg.num_particles;
  const uint32_t rr = (uint32_t)((int64_t)row + row_offset);
  std::vector<double> wts(P, 0.0);
  std::vector<std::vector<int32_t>> pch(n_blocks, std::vector<int32_t>(P));
  std::vector<std::vector<std::vector<int32_t>>> pvals(n_blocks, std::vector<std::vector<int32_t>>(P));
  double acc = 0.0;
  std::vector<char> has_dummy(n_blocks, 0);
  for (int bi = 0; bi < n_blocks; ++bi)
    for (const pclean_node& nd : w.block[bi].nodes) has_dummy[bi] |= nd.kind == PCLEAN_NODE_LEAF && nd.dummy_value != 0;
  for (int bi = 0; bi < n_blocks; ++bi) {
    const OBlock& b = w.block[bi];
    if (b.is_score) { /* p += logdensity(...) of the block's observed choices (block_proposal.jl:62-64) */
      auto src = [&](int blk, int col, int p) {
        const OBlock& sbk = w.block[blk];
        const OTable& rt = w.table[sbk.nodes[0].table];
        const int ch = pch[blk][p];
        return ch >= 0 ? rt.cols[(size_t)col * rt.n_rows + ch] : resolve_new_value(w, sbk, 0, col, pvals[blk][p].data());
      };
      const OFn& pf = w.fn[b.prob_fn];
      for (int p = 0; p < P; ++p) {
        const int pidx = pf.fn[(size_t)src(b.prob_a_block, b.prob_a_col, p) * pf.n_b + src(b.prob_b_block, b.prob_b_col, p)];
        double acc2 = 0.0;
        for (const auto& stt : b.score_terms) {
          const int val = src(stt.val_block, stt.val_col, p);
          const int o = w.obs[(size_t)stt.obs_col * w.n_rows + row];
          const OPair& pt = w.pair[stt.pair_table];
          const bool same = o >= 0 && pt.d[(size_t)o * pt.n_lat + val] == 0;
          const int nopt = w.fn[stt.nopt_fn].fn[src(stt.key_block, stt.key_col, p)];
          acc2 += maybe_swap_term(w, o < 0, same, val < stt.other_val, nopt, pidx);  /* ids from the dummy's on: not an option */
        }
        wts[p] += acc2;
      }
      continue;
    }
    const int n_root = w.table[b.nodes[0].table].n_rows;
    const int excl = cur[bi];
    auto ctx_of = [&](int p, int32_t* out) {
      for (int s = 0; s < PCLEAN_MAX_CTX; ++s) {
        out[s] = 0;
        if (s >= b.n_ctx) continue;
        const int sb = b.ctx_src_block[s], sc = b.ctx_src_col[s];
        const OBlock& src = w.block[sb];
        const OTable& rt = w.table[src.nodes[0].table];
        const int ch = pch[sb][p];
        out[s] = ch >= 0 ? rt.cols[(size_t)sc * rt.n_rows + ch] : resolve_new_value(w, src, 0, sc, pvals[sb][p].data());
      }
    };
    /* draws of particle p; returns the weight correction of the ProposalDummyValues its new row chose (0 otherwise) —
     * added AFTER the block's log marginal (the order of the two fp64 additions is part of the contract) */
    auto finish_draw = [&](int p, const RowCtx& rc, int k) -> double {
      int c = k == n_root ? PCLEAN_CHOICE_NEW : k;
      if (p == 0 && cur[bi] >= 0) c = cur[bi]; /* retained particle, row_inference.jl:143-145 */
      pch[bi][p] = c;
      if (c == PCLEAN_CHOICE_NEW) {
        pvals[bi][p].assign(b.nodes.size(), -2);
        pvals[bi][p][0] = PCLEAN_CHOICE_NEW;
        sample_new(rc, 0, excl, (uint32_t)p, pvals[bi][p].data());
        if (has_dummy[bi]) return dummy_correction(rc, pvals[bi][p].data(), (uint32_t)p);
      }
      return 0.0;
    };
    auto finish_particle = [&](int p, const RowCtx& rc, const std::vector<double>& s, const FixSum& f) -> double {
      return finish_draw(p, rc, fix_draw(s, f, pclean_rand64(seed, rr, PCLEAN_SITE_NODE(bi, 0), (uint32_t)p, sweep)));
    };
    if (!cfg.use_dd_proposals) { /* prior proposals: every particle draws on its own, weight = likelihood of the draw */
      for (int p = 0; p < P; ++p) {
        int32_t cv[PCLEAN_MAX_CTX];
        ctx_of(p, cv);
        RowCtx rc{&w, bi, row, b.n_ctx ? cv : nullptr, seed, sweep, row_offset};
        std::vector<double> s;
        node_scores(w, bi, 0, row, rc.ctxv, excl, 0.0, s, nullptr, SCORE_PRIOR);
        FixSum f = fix_sum(s);
        const int k = fix_draw(s, f, pclean_rand64(seed, rr, PCLEAN_SITE_NODE(bi, 0), (uint32_t)p, sweep));
        int c = k == n_root ? PCLEAN_CHOICE_NEW : k;
        if (p == 0 && cur[bi] >= 0) c = cur[bi];
        pch[bi][p] = c;
        double corr = 0.0;
        if (c == PCLEAN_CHOICE_NEW) {
          pvals[bi][p].assign(b.nodes.size(), -2);
          pvals[bi][p][0] = PCLEAN_CHOICE_NEW;
          sample_new_prior(rc, 0, excl, (uint32_t)p, pvals[bi][p].data());
          if (has_dummy[bi]) corr = dummy_correction(rc, pvals[bi][p].data(), (uint32_t)p);
        }
        wts[p] += subtree_terms(rc, 0, c, excl, pvals[bi][p].data());
        if (!b.node_gauss.empty() && b.node_gauss[0] >= 0) {
          const pclean_gauss& g = b.gauss[b.node_gauss[0]];
          const OTable& rt = w.table[b.nodes[0].table];
          if (plocals[bi].empty()) plocals[bi].assign((size_t)P * 2, -1);
          const std::vector<int32_t>& cl = w.cur_locals[bi];
          const int32_t* keep = (p == 0 && cur[bi] >= 0 && cl.size() >= (size_t)(row + 1) * 2) ? &cl[(size_t)row * 2] : nullptr;
          wts[p] += gauss_prior_term(w, g, bi, row, rr, (uint32_t)p, seed, sweep, keep, [&](int d) -> int {
            if (g.src_kind[d] == PCLEAN_GSRC_CAND)
              return c >= 0 ? rt.cols[(size_t)g.src[d] * rt.n_rows + c] : resolve_new_value(w, b, 0, g.src[d], pvals[bi][p].data());
            return w.obs[(size_t)g.src[d] * w.n_rows + row];
          }, &plocals[bi][(size_t)p * 2]);
        }
        wts[p] += corr;
      }
    } else if (b.n_ctx == 0) {
      RowCtx rc{&w, bi, row, nullptr, seed, sweep, row_offset};
      bool done = false;
      if constexpr (!std::is_same<PR, NoPruner>::value) {
        if (pr) {
          const auto& R = pr->root(rc, excl);
          for (int p = 0; p < P; ++p) {
            const double corr = finish_draw(p, rc, R.draw(pclean_rand64(seed, rr, PCLEAN_SITE_NODE(bi, 0), (uint32_t)p, sweep)));
            wts[p] += R.lse;
            wts[p] += corr;
          }
          done = true;
        }
      }
      if (!done) {
        std::vector<double> s;
        const double lse = eval_tree(rc, 0, excl, &s);
        FixSum f = fix_sum(s);
        for (int p = 0; p < P; ++p) {
          const double corr = finish_particle(p, rc, s, f);
          wts[p] += lse;
          wts[p] += corr;
        }
      }
    } else {
      for (int p = 0; p < P; ++p) {
        int32_t cv[PCLEAN_MAX_CTX];
        ctx_of(p, cv);
        RowCtx rc{&w, bi, row, cv, seed, sweep, row_offset};
        bool done = false;
        if constexpr (!std::is_same<PR, NoPruner>::value) {
          if (pr) {
            const auto& R = pr->root(rc, excl);
            const double corr = finish_draw(p, rc, R.draw(pclean_rand64(seed, rr, PCLEAN_SITE_NODE(bi, 0), (uint32_t)p, sweep)));
            wts[p] += R.lse;
            wts[p] += corr;
            done = true;
          }
        }
        if (done) continue;
        std::vector<double> s;
        const double lse = eval_tree(rc, 0, excl, &s);
        FixSum f = fix_sum(s);
        const double corr = finish_particle(p, rc, s, f);
        wts[p] += lse;
        wts[p] += corr;
      }
    }
    const int grp_here = b.group >= 0 ? b.group : bi;
    const int grp_next = bi + 1 < n_blocks ? (w.block[bi + 1].group >= 0 ? w.block[bi + 1].group : bi + 1) : -2;
    if (!use_mh && bi < n_blocks - 1 && grp_here != grp_next) { /* row_inference.jl:152-155: between the MODEL's blocks */
      std::vector<int> anc;
      double inc;
      if (maybe_resample(wts, cur[bi] >= 0, seed, rr, sweep, (uint32_t)bi, anc, inc, nullptr)) {
        for (int k = 0; k <= bi; ++k) {
          if (w.block[k].is_score) continue;
          std::vector<int32_t> nc(P);
          std::vector<std::vector<int32_t>> nv(P);
          for (int p = 0; p < P; ++p) {
            nc[p] = pch[k][anc[p]];
            nv[p] = pvals[k][anc[p]];
          }
          pch[k].swap(nc);
          pvals[k].swap(nv);
          if (!plocals[k].empty()) {
            std::vector<int32_t> nl((size_t)P * 2);
            for (int p = 0; p < P; ++p) {
              nl[(size_t)p * 2] = plocals[k][(size_t)anc[p] * 2];
              nl[(size_t)p * 2 + 1] = plocals[k][(size_t)anc[p] * 2 + 1];
            }
            plocals[k].swap(nl);
          }
        }
        for (int p = 0; p < P; ++p) wts[p] = 0.0;
      }
      acc += inc;
    }
  }
  double log_total;
  const int c = final_choice(wts, use_mh, cur[0] >= 0, seed, rr, sweep, &log_total);
  if (chosen_particle) *chosen_particle = c;
  if (logml) *logml = acc + log_total - pclean_log((double)P);
  for (int bi = 0; bi < n_blocks; ++bi) {
    if (w.block[bi].is_score) {
      choice[bi] = 0;
      continue;
    }
    choice[bi] = pch[bi][c];
    if (pch[bi][c] == PCLEAN_CHOICE_NEW) {
      new_rows.push_back(NewRow{bi, row, pvals[bi][c]});
      new_rows.back().vals[0] = -1 - c; /* the chosen particle names the draw stream of its dummy values */
    }
    /* own enumerated choices (locals) of the chosen particle given its referent */
    const OBlock& b = w.block[bi];
    if (locals_out && !cfg.use_dd_proposals && !plocals[bi].empty()) { /* prior proposals: what the chosen particle sampled */
      locals_out[2 * bi] = plocals[bi][(size_t)c * 2];
      locals_out[2 * bi + 1] = plocals[bi][(size_t)c * 2 + 1];
    } else if (locals_out && !b.node_gauss.empty() && b.node_gauss[0] >= 0 && b.gauss[b.node_gauss[0]].n_locals > 0) {
      const pclean_gauss& g = b.gauss[b.node_gauss[0]];
      const OTable& rt = w.table[b.nodes[0].table];
      const int ch = pch[bi][c];
      GaussCombos gc;
      const double xv = w.xnum[(size_t)g.x_col * w.n_rows + row];
      if (xv == xv) {
        gc = gauss_combo_scores(w, g, row, nullptr, [&](int d) -> int {
          if (g.src_kind[d] == PCLEAN_GSRC_CAND)
            return ch >= 0 ? rt.cols[(size_t)g.src[d] * rt.n_rows + ch]
                           : resolve_new_value(w, b, 0, g.src[d], pvals[bi][c].data());
          return w.obs[(size_t)g.src[d] * w.n_rows + row];
        });
      } else {
        const int n0 = g.local_n[0], n1 = g.n_locals > 1 ? g.local_n[1] : 1;
        for (int l0 = 0; l0 < n0; ++l0)
          for (int l1 = 0; l1 < n1; ++l1) {
            auto ok = [&](int l, int v) {
              if (l >= g.n_locals || g.local_obs_col[l] < 0) return true;
              const int o = w.obs[(size_t)g.local_obs_col[l] * w.n_rows + row];
              return o < 0 || o == v;
            };
            if (ok(0, l0) && ok(1, l1)) {
              gc.sc[gc.n] = 0.0;
              gc.codes[gc.n] = l0 * 16 + l1;
              ++gc.n;
            }
          }
      }
      std::vector<double> sv(gc.sc, gc.sc + gc.n);
      FixSum f = fix_sum(sv);
      int pick = gc.n - 1;
      if (f.U) pick = fix_draw(sv, f, pclean_rand64(seed, rr, PCLEAN_SITE_LOCALS(bi), (uint32_t)c, sweep));
      locals_out[2 * bi] = gc.n ? gc.codes[pick] >> 4 : -1;
      locals_out[2 * bi + 1] = (gc.n && g.n_locals > 1) ? (gc.codes[pick] & 15) : -1;
    }
  }
}

/* Rejuvenation of the rows of a latent class (pgibbs_sweep! over a latent class): every
 * independent sub-plan of the class's attributes is enumerated against the referring rows; all
 * particles get the same weight, so the chosen particle is uniform (PG) / accepted with
 * min(1, w1/(1e-10+w0)), w0 == w1 (MH).  Only the chosen particle's draws are materialised. */
inline void sweep_latent(const World& w, const pclean_infer_config& cfg, uint64_t seed, uint32_t sweep, int block_id,
                         int n_roots, const int32_t* roots, int n_items, const int32_t* keys, const int32_t* ev_off,
                         const int32_t* ev_rows, const int32_t* ev_ctx, const int32_t* excl, int32_t* chosen,
                         int32_t* vals) {
  const OBlock& b = w.block[block_id];
  const int nn = (int)b.nodes.size();
  const bool use_mh = cfg.use_mh_instead_of_pg != 0;
  const int P = use_mh ? 2 : cfg.num_particles;
  if (!cfg.use_dd_proposals) {
    /* prior proposals (block_proposal.jl:168): particle 0 keeps the row's current values (excl[r][t]: current referent
     * of a reference slot, current OPTION of a choice), every other particle draws each attribute from its prior;
     * weight = likelihood of the referring rows given the particle's values; then the usual final choice. */
    for (int t = 0; t < n_items; ++t) {
      const uint32_t rr = (uint32_t)keys[t];
      const uint32_t pid = 0x1000u + (uint32_t)block_id;
      Evidence ev;
      ev.rows = ev_rows + ev_off[t];
      ev.ctx = ev_ctx ? ev_ctx + (size_t)ev_off[t] * PCLEAN_MAX_CTX : nullptr;
      ev.n = ev_off[t + 1] - ev_off[t];
      RowCtx rc{&w, block_id, 0, nullptr, seed, sweep, 0, &ev, (int64_t)keys[t]};
      std::vector<double> wts(P, 0.0);
      std::vector<std::vector<int32_t>> pv(P, std::vector<int32_t>(nn, -2));
      for (int p = 0; p < P; ++p)
        for (int r = 0; r < n_roots; ++r) {
          const int root = roots[r];
          const pclean_node& rn = b.nodes[root];
          const int curv = excl[(size_t)r * n_items + t];
          const int rex = rn.kind == PCLEAN_NODE_FK ? curv : -1;
          int c = curv;
          if (p > 0) {
            std::vector<double> s;
            node_scores(w, block_id, root, 0, nullptr, rex, 0.0, s, &ev, SCORE_PRIOR);
            FixSum f = fix_sum(s);
            const int k = fix_draw(s, f, pclean_rand64(seed, rr, PCLEAN_SITE_NODE(block_id, root), (uint32_t)p, sweep));
            c = (rn.kind == PCLEAN_NODE_FK && k == w.table[rn.table].n_rows) ? PCLEAN_CHOICE_NEW : k;
            if (c == PCLEAN_CHOICE_NEW) sample_new_prior(rc, root, rex, (uint32_t)p, pv[p].data());
          }
          pv[p][root] = c;
          wts[p] += subtree_terms(rc, root, c, rex, pv[p].data());
        }
      PartW pw = part_weights(wts);
      int c;
      if (use_mh && P >= 2) {
        const double Ud = (double)pw.U;
        const double w0 = (double)pw.u[0] / Ud, w1 = (double)pw.u[1] / Ud;
        double ratio = w1 / (1e-10 + w0);
        if (ratio > 1.0) ratio = 1.0;
        c = (pw.U != 0 && pclean_u01(pclean_rand64(seed, rr, PCLEAN_SITE_MH, pid, sweep)) < ratio) ? 1 : 0;
      } else {
        c = part_pick(pw, pclean_rand64(seed, rr, PCLEAN_SITE_FINAL, pid, sweep));
      }
      chosen[t] = c;
      for (int k = 0; k < nn; ++k) vals[(size_t)t * nn + k] = c > 0 ? pv[c][k] : -2;
    }
    return;
  }
  for (int t = 0; t < n_items; ++t) {
    const uint32_t rr = (uint32_t)keys[t];
    const uint32_t pid = 0x1000u + (uint32_t)block_id;
    int c;
    if (use_mh && P >= 2) {
      const double ratio = 0.5 / (1e-10 + 0.5);
      c = pclean_u01(pclean_rand64(seed, rr, PCLEAN_SITE_MH, pid, sweep)) < ratio ? 1 : 0;
    } else {
      const uint64_t U = (uint64_t)P << PCLEAN_FIX_BITS;
      c = (int)(pclean_mulhi64(pclean_rand64(seed, rr, PCLEAN_SITE_FINAL, pid, sweep), U) >> PCLEAN_FIX_BITS);
    }
    chosen[t] = c;
    for (int k = 0; k < nn; ++k) vals[(size_t)t * nn + k] = -2;
    if (c == 0) continue;
    Evidence ev;
    ev.rows = ev_rows + ev_off[t];
    ev.ctx = ev_ctx ? ev_ctx + (size_t)ev_off[t] * PCLEAN_MAX_CTX : nullptr;
    ev.n = ev_off[t + 1] - ev_off[t];
    RowCtx rc{&w, block_id, 0, nullptr, seed, sweep, 0, &ev, (int64_t)keys[t]};
    for (int r = 0; r < n_roots; ++r) {
      const int root = roots[r];
      const pclean_node& rn = b.nodes[root];
      const int rex = rn.kind == PCLEAN_NODE_FK ? excl[(size_t)r * n_items + t] : -1;
      std::vector<double> s;
      eval_tree(rc, root, rex, &s);
      FixSum f = fix_sum(s);
      const int k = fix_draw(s, f, pclean_rand64(seed, rr, PCLEAN_SITE_NODE(block_id, root), (uint32_t)c, sweep));
      const int n = w.table[rn.table].n_rows;
      if (rn.kind == PCLEAN_NODE_FK && k == n) {
        vals[(size_t)t * nn + root] = PCLEAN_CHOICE_NEW;
        sample_new(rc, root, rex, (uint32_t)c, vals + (size_t)t * nn);
      } else {
        vals[(size_t)t * nn + root] = k;
      }
    }
  }
}

/* Sequential-schedule sweep (the reference's schedule, row_inference.jl:169-185):
 * after each row the reference counts and CRP prior pieces of the referents it
 * left / joined are updated, so row i+1 sees them.  Rows that choose a NEW referent
 * keep their old referent here (creating latent rows is bookkeeping outside the
 * timed scoring work; counted in n_new).  Used as the single-thread CPU baseline. */
inline void sweep_sequential(World& w, const pclean_infer_config& cfg, uint64_t seed, uint32_t sweep, int n_blocks,
                             int64_t row_offset, int32_t* cur /*[n_blocks][N] in/out*/, const double* py /*[64][2]*/,
                             int64_t* n_moved, int64_t* n_new) {
  const int N = w.n_rows;
  std::vector<int32_t> c(n_blocks), ch(n_blocks);
  std::vector<NewRow> news;
  *n_moved = 0;
  *n_new = 0;
  for (int i = 0; i < N; ++i) {
    for (int b = 0; b < n_blocks; ++b) c[b] = cur[(size_t)b * N + i];
    news.clear();
    run_smc_row(w, cfg, seed, sweep, n_blocks, i, row_offset, c.data(), ch.data(), nullptr, nullptr, news);
    for (int b = 0; b < n_blocks; ++b) {
      if (ch[b] == PCLEAN_CHOICE_NEW) {
        ++*n_new;
        continue;
      }
      if (ch[b] == c[b]) continue;
      ++*n_moved;
      OTable& t = w.table[w.block[b].nodes[0].table];
      const double discount = py[2 * w.block[b].nodes[0].table + 1];
      const int upd[2] = {c[b], ch[b]};
      t.counts[c[b]] -= 1;
      t.counts[ch[b]] += 1;
      for (int k : upd) {
        const int64_t cnt = t.counts[k];
        t.logc_full[k] = cnt > 0 ? std::log((double)cnt - discount) : NEG_INF;
        t.logc_m1[k] = cnt > 1 ? std::log((double)(cnt - 1) - discount) : NEG_INF;
      }
      cur[(size_t)b * N + i] = ch[b];
    }
  }
}

} /* namespace pco */
#endif
