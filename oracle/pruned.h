/* oracle/pruned.h — TEST INFRASTRUCTURE ONLY (the CPU baseline's second leg; never loaded by the product).
 *
 * The batched sweep of oracle/sweep.h with the two EXACT work savers of the HIP path restated for one CPU thread, so
 * that bench.py can say what grouping + pruning buy without a GPU ("a one-thread CPU run of the same
 * pruned algorithm"):
 *
 *   grouping   the enumeration of a block's root for a row depends on the row only through the observed values of the
 *              block's terms, the contexts and the current referent: rows (and particles) that agree on those share
 *              ONE evaluation (the HIP path: make_item_groups / leaf caches / memo tables).  The batched schedule
 *              freezes the latent tables for the sweep, so the memo is exact.  Leaf children of a new-row branch are
 *              memoised the same way per (node, observed values).
 *   pruning    a candidate k of a reference slot scores at most  prior_k - c_min * D_k  (D_k = summed edit distance of
 *              its plain AddTypos terms, c_min = smallest density cost of one edit: enum_kernels.hip / eval.hip); when
 *              that is more than 28.5 nats below the exact score of the row's CURRENT referent its fixed-point weight is
 *              exactly 0 (pclean_fixw) and its exact score is never computed.  Likewise the new-row branch: when its
 *              upper bound (CRP term + exact marginals of its memoised leaf children + 0 for the others — a log
 *              marginal of discrete observations is <= 0) is that far below, its children are never enumerated.
 *
 * Results are the ones of run_smc_row bit for bit (tests/test_oracle_pruned.py): the maximum is attained by a candidate
 * that is kept, dropped candidates weigh exactly 0, and a draw walks the kept candidates in the same (ascending) order.
 * Follows: proposal_compiler.jl:131-252 (the enumeration), block_proposal.jl:160-190 (weights), row_inference.jl:108-187.
 */
#ifndef PCLEAN_ORACLE_PRUNED_H
#define PCLEAN_ORACLE_PRUNED_H

#include <cstring>
#include <unordered_map>
#include <vector>

#include "sweep.h"

namespace pco {

struct VecHash {
  size_t operator()(const std::vector<int32_t>& v) const {
    uint64_t h = 1469598103934665603ull;
    for (int32_t x : v) {
      h ^= (uint32_t)x;
      h *= 1099511628211ull;
    }
    return (size_t)h;
  }
};

struct PrunedRoot { /* the candidates of a root enumeration that weigh anything, ascending (new row = n_root, last) */
  double m = NEG_INF, lse = NEG_INF;
  uint64_t U = 0;
  int n_all = 0; /* number of candidates incl. the new row: what a draw falls back to when U == 0 */
  std::vector<int32_t> k;
  std::vector<uint64_t> pref;
  int draw(uint64_t R) const { /* fix_draw() on the full score vector */
    if (U == 0) return n_all - 1;
    const uint64_t x = pclean_mulhi64(R, U);
    size_t a = 0, b = pref.size();
    while (a < b) {
      const size_t mid = (a + b) >> 1;
      if (pref[mid] > x)
        b = mid;
      else
        a = mid + 1;
    }
    return a < k.size() ? k[a] : n_all - 1;
  }
};

struct PrunedStats {
  uint64_t roots = 0, root_hits = 0, cand_exact = 0, cand_pruned = 0, new_skipped = 0, new_evaluated = 0, child_hits = 0,
           child_miss = 0, full_fallback = 0;
};

struct Pruner {
  const World& w;
  std::unordered_map<std::vector<int32_t>, PrunedRoot, VecHash> roots;
  std::unordered_map<std::vector<int32_t>, double, VecHash> child_lse; /* eval_tree(child) per (block, node, excl, ctx, observed values below) */
  std::vector<std::vector<std::vector<int32_t>>> sub_cols;           /* [block][node]: observed columns of the terms in the sub-tree */
  std::vector<std::vector<char>> sub_ctx;                            /* [block][node]: a term of the sub-tree reads a context value */
  double cmin = 0.0;
  PrunedStats st;
  PrunedRoot scratch; /* the result of an evaluation that is not memoised */

  explicit Pruner(const World& world) : w(world) {
    /* smallest density cost of one edit over every (length, distance) the density tables hold (eval.hip: try_fast_root) */
    double cm = INFINITY;
    for (int L = 1; L <= w.max_len; ++L)
      for (int d = 1; d <= w.max_d; ++d) {
        const int r = (L + 4) / 5;
        if (r > w.max_r) continue;
        double l = w.nb[(size_t)r * (w.max_d + 1) + d];
        l -= w.logl[L] * (double)d;
        l -= HALF_LOG26 * (double)d;
        if (l == l) cm = std::min(cm, -l / (double)d);
      }
    cmin = (cm > 1e-6 && std::isfinite(cm)) ? cm * (1.0 - 1e-9) : 0.0;
    sub_cols.resize(w.block.size());
    sub_ctx.resize(w.block.size());
    for (size_t bi = 0; bi < w.block.size(); ++bi) {
      const OBlock& b = w.block[bi];
      sub_cols[bi].resize(b.nodes.size());
      sub_ctx[bi].assign(b.nodes.size(), 0);
      for (int node = (int)b.nodes.size() - 1; node >= 0; --node) { /* children have larger ids than their parents */
        std::vector<int32_t>& c = sub_cols[bi][node];
        const pclean_node& nd = b.nodes[node];
        for (int ti = 0; ti < nd.n_terms; ++ti) {
          c.push_back(b.terms[nd.term_begin + ti].obs_col);
          if (b.terms[nd.term_begin + ti].ctx_slot >= 0) sub_ctx[bi][node] = 1;
        }
        for (int ci = 0; ci < nd.n_children; ++ci) {
          const int cid = b.children[nd.child_begin + ci];
          const std::vector<int32_t>& cc = sub_cols[bi][cid];
          c.insert(c.end(), cc.begin(), cc.end());
          if (sub_ctx[bi][cid]) sub_ctx[bi][node] = 1;
        }
      }
    }
  }

  std::vector<int32_t> key_of(const RowCtx& rc, int node, int excl) const {
    std::vector<int32_t> key;
    const std::vector<int32_t>& cols = sub_cols[rc.block][node];
    key.reserve(cols.size() + 4 + PCLEAN_MAX_CTX);
    key.push_back(rc.block);
    key.push_back(node);
    key.push_back(excl);
    if (sub_ctx[rc.block][node])
      for (int s = 0; s < PCLEAN_MAX_CTX; ++s) key.push_back(rc.ctxv ? rc.ctxv[s] : 0);
    for (int32_t c : cols) key.push_back(w.obs[(size_t)c * w.n_rows + rc.row]);
    return key;
  }

  /* eval_tree(rc, child, excl) through the memo (a Gaussian term reads the row's number: not memoised) */
  double child(const RowCtx& rc, int node, int excl) {
    const OBlock& b = w.block[rc.block];
    bool gauss = false;
    for (size_t i = 0; i < b.node_gauss.size(); ++i) gauss = gauss || b.node_gauss[i] >= 0;
    if (gauss) return eval_tree(rc, node, excl, nullptr);
    std::vector<int32_t> key = key_of(rc, node, excl);
    auto it = child_lse.find(key);
    if (it != child_lse.end()) {
      ++st.child_hits;
      return it->second;
    }
    ++st.child_miss;
    const double v = eval_tree(rc, node, excl, nullptr);
    child_lse.emplace(std::move(key), v);
    return v;
  }
  bool child_known(const RowCtx& rc, int node, int excl, double* v) const {
    auto it = child_lse.find(key_of(rc, node, excl));
    if (it == child_lse.end()) return false;
    *v = it->second;
    return true;
  }

  static void compact(const std::vector<double>& s, PrunedRoot& R) {
    const FixSum f = fix_sum(s);
    R.m = f.m;
    R.U = f.U;
    R.lse = pclean_lse_from_fix(f.m, f.U);
    R.n_all = (int)s.size();
    uint64_t acc = 0;
    if (f.m != NEG_INF)
      for (size_t k = 0; k < s.size(); ++k) {
        const uint64_t u = pclean_fixw(s[k] - f.m);
        if (!u) continue;
        acc += u;
        R.k.push_back((int32_t)k);
        R.pref.push_back(acc);
      }
  }

  /* the root enumeration of rc.block for rc.row (eval_tree(rc, 0, excl, &s) + fix_sum), grouped and pruned */
  const PrunedRoot& root(const RowCtx& rc, int excl) {
    ++st.roots;
    const OBlock& b = w.block[rc.block];
    bool gauss = false; /* a Gaussian term reads the row's NUMBER: rows that agree on the dictionary-encoded values do not share it */
    for (size_t i = 0; i < b.node_gauss.size(); ++i) gauss = gauss || b.node_gauss[i] >= 0;
    if (gauss) {
      ++st.full_fallback;
      std::vector<double> s;
      eval_tree(rc, 0, excl, &s);
      scratch = PrunedRoot();
      compact(s, scratch);
      return scratch;
    }
    std::vector<int32_t> key = key_of(rc, 0, excl);
    auto it = roots.find(key);
    if (it != roots.end()) {
      ++st.root_hits;
      return it->second;
    }
    PrunedRoot R;
    const pclean_node& nd = b.nodes[0];
    const OTable& t = w.table[nd.table];
    const int n = t.n_rows;
    bool plain = nd.kind == PCLEAN_NODE_FK && cmin > 0.0 && !rc.ev && excl >= 0 && t.counts[excl] > 1;
    for (int ti = 0; ti < nd.n_terms && plain; ++ti) plain = b.terms[nd.term_begin + ti].dens_kind == PCLEAN_DENS_ADD_TYPOS;
    for (size_t i = 0; i < b.node_gauss.size() && plain; ++i) plain = b.node_gauss[i] < 0;
    if (!plain) { /* no current referent to compare with / other term kinds: the full enumeration, grouped only */
      ++st.full_fallback;
      std::vector<double> s;
      double snew = NEG_INF;
      if (nd.kind == PCLEAN_NODE_FK) {
        snew = 0.0;
        for (int c = 0; c < nd.n_children; ++c) {
          const int cid = b.children[nd.child_begin + c];
          snew += child(rc, cid, child_excl_of(w, b, 0, cid, excl));
        }
      }
      node_scores(w, rc.block, 0, rc.row, rc.ctxv, excl, snew, s, rc.ev);
      compact(s, R);
      return roots.emplace(std::move(key), std::move(R)).first->second;
    }
    const double logden = t.scal[1]; /* (excluded, not deleted: node_scores) */
    /* observed value, pair table and candidate column of every term, in plan order */
    struct T {
      const pclean_term* tm;
      const OPair* pt;
      int o;
      const int32_t* col;
    };
    std::vector<T> terms;
    for (int ti = 0; ti < nd.n_terms; ++ti) {
      const pclean_term& tm = b.terms[nd.term_begin + ti];
      terms.push_back(T{&tm, &w.pair[tm.pair_table], w.obs[(size_t)tm.obs_col * w.n_rows + rc.row], &t.cols[(size_t)tm.cand_col * n]});
    }
    auto exact = [&](int k, double prior) { /* node_scores(): prior, then the terms in plan order */
      double sc = prior;
      for (const T& x : terms) {
        if (x.o < 0) continue;
        int val = x.col[k];
        if (x.tm->ctx_slot >= 0) {
          const OFn& f = w.fn[x.tm->fn_table];
          val = f.fn[(size_t)rc.ctxv[x.tm->ctx_slot] * f.n_b + val];
        }
        sc += term_density(w, *x.tm, *x.pt, x.pt->d[(size_t)x.o * x.pt->n_lat + val], val);
      }
      return sc;
    };
    const double score_cur = exact(excl, t.logc_m1[excl] - logden);
    /* the new-row branch: an upper bound first (memoised children exactly, the others at 0) */
    const double crp_new = t.scal[2] - logden;
    double ub = crp_new;
    for (int c = 0; c < nd.n_children; ++c) {
      const int cid = b.children[nd.child_begin + c];
      double v;
      if (b.nodes[cid].kind == PCLEAN_NODE_LEAF)
        v = child(rc, cid, -1); /* an option list: one enumeration per distinct observed value, shared by every row */
      else if (!child_known(rc, cid, child_excl_of(w, b, 0, cid, excl), &v))
        v = 0.0;
      ub += v;
    }
    double sn = NEG_INF;
    if (!(ub + 1e-6 < score_cur - 28.5)) {
      ++st.new_evaluated;
      double snew = 0.0;
      for (int c = 0; c < nd.n_children; ++c) {
        const int cid = b.children[nd.child_begin + c];
        snew += child(rc, cid, child_excl_of(w, b, 0, cid, excl));
      }
      sn = crp_new + snew; /* node_scores(): ((deleted ? scal[3] : scal[2]) - logden) + snew_in */
    } else {
      ++st.new_skipped;
    }
    const double bound = score_cur > sn ? score_cur : sn;
    const double thresh = bound - 28.5 - 1e-6;
    std::vector<int32_t> ks;
    std::vector<double> sc;
    for (int k = 0; k < n; ++k) {
      if (t.counts[k] == 0) continue;
      const double prior = (k == excl ? t.logc_m1[k] : t.logc_full[k]) - logden;
      double ubk = prior;
      bool out = ubk < thresh;
      for (size_t ti = 0; ti < terms.size() && !out; ++ti) {
        const T& x = terms[ti];
        if (x.o < 0 || x.tm->ctx_slot >= 0) continue;
        ubk -= cmin * (double)x.pt->d[(size_t)x.o * x.pt->n_lat + x.col[k]];
        out = ubk < thresh;
      }
      if (out) {
        ++st.cand_pruned;
        continue;
      }
      ++st.cand_exact;
      ks.push_back(k);
      sc.push_back(k == excl ? score_cur : exact(k, prior));
    }
    ks.push_back(n);
    sc.push_back(sn);
    double m = NEG_INF;
    for (double v : sc) m = v > m ? v : m;
    R.m = m;
    R.n_all = n + 1;
    uint64_t acc = 0;
    if (m != NEG_INF)
      for (size_t j = 0; j < sc.size(); ++j) {
        const uint64_t u = pclean_fixw(sc[j] - m);
        if (!u) continue;
        acc += u;
        R.k.push_back(ks[j]);
        R.pref.push_back(acc);
      }
    R.U = acc;
    R.lse = pclean_lse_from_fix(m, acc);
    return roots.emplace(std::move(key), std::move(R)).first->second;
  }
};

} /* namespace pco */
#endif
