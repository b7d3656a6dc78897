/* oracle/densities.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Single-thread CPU restatement of the log-densities, string distances and
 * small numeric helpers on PClean's inference hot path.  Every function cites
 * the reference file:line (relative to /root/reference) it follows.
 *
 * PARITY UNPINNED: the reference is Julia, cannot run in this image, ships no
 * tests / golden vectors / RNG seed.  These functions are
 * pinned instead by formula-derived known-answer values (tests/golden/kat.json,
 * generated with scipy by tests/golden/make_kat.py) and brute-force property
 * tests.  Third-party arithmetic restated here from published definitions:
 *   StringDistances.jl (unpinned): DamerauLevenshtein — both the restricted
 *     (optimal-string-alignment) form used before v0.11 and the unrestricted
 *     Lowrance–Wagner form used since;
 *   Distributions.jl (unpinned): NegativeBinomial / Normal log-pdf.
 */
#ifndef PCLEAN_ORACLE_DENSITIES_H
#define PCLEAN_ORACLE_DENSITIES_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <limits>

namespace pco {

static const double NEG_INF = -std::numeric_limits<double>::infinity();

/* ---- string distances (StringDistances.jl `evaluate(DamerauLevenshtein(), a, b)`,
 *      called at src/distributions/add_typos.jl:56) ------------------------- */

/* Restricted Damerau–Levenshtein = optimal string alignment (adjacent
 * transposition counted once, no substring edited twice). */
inline int osa_distance(const uint16_t* a, int la, const uint16_t* b, int lb) {
  std::vector<int> p2(lb + 1), p1(lb + 1), cur(lb + 1);
  for (int j = 0; j <= lb; ++j) p1[j] = j;
  for (int i = 1; i <= la; ++i) {
    cur[0] = i;
    for (int j = 1; j <= lb; ++j) {
      int cost = a[i - 1] == b[j - 1] ? 0 : 1;
      int v = std::min(std::min(p1[j] + 1, cur[j - 1] + 1), p1[j - 1] + cost);
      if (i > 1 && j > 1 && a[i - 1] == b[j - 2] && a[i - 2] == b[j - 1]) v = std::min(v, p2[j - 2] + 1);
      cur[j] = v;
    }
    std::swap(p2, p1);
    std::swap(p1, cur);
  }
  return p1[lb];
}

/* Unrestricted Damerau–Levenshtein (Lowrance & Wagner 1975). Symbols are
 * dense ids < 65536. */
inline int dl_distance(const uint16_t* a, int la, const uint16_t* b, int lb) {
  const int W = lb + 2;
  std::vector<int> H((size_t)(la + 2) * W);
  const int maxdist = la + lb;
  std::vector<int> da(65536, 0);
  H[0] = maxdist;
  for (int i = 0; i <= la; ++i) {
    H[(i + 1) * W + 0] = maxdist;
    H[(i + 1) * W + 1] = i;
  }
  for (int j = 0; j <= lb; ++j) {
    H[0 * W + (j + 1)] = maxdist;
    H[1 * W + (j + 1)] = j;
  }
  for (int i = 1; i <= la; ++i) {
    int db = 0;
    for (int j = 1; j <= lb; ++j) {
      int k = da[b[j - 1]];
      int l = db;
      int cost = 1;
      if (a[i - 1] == b[j - 1]) {
        cost = 0;
        db = j;
      }
      int v = H[i * W + j] + cost;                             /* substitute / match */
      v = std::min(v, H[(i + 1) * W + j] + 1);                 /* insert  */
      v = std::min(v, H[i * W + (j + 1)] + 1);                 /* delete  */
      v = std::min(v, H[k * W + l] + (i - k - 1) + 1 + (j - l - 1)); /* transpose */
      H[(i + 1) * W + (j + 1)] = v;
    }
    da[a[i - 1]] = i;
  }
  return H[(la + 1) * W + (lb + 1)];
}

/* ---- Distributions.jl log-pdfs ---------------------------------------- */

/* logpdf(NegativeBinomial(r, p), k): failures k before the r-th success. */
inline double negbin_logpdf(double r, double p, int k) {
  return std::lgamma(k + r) - std::lgamma(k + 1.0) - std::lgamma(r) + r * std::log(p) +
         k * std::log1p(-p);
}

inline double normal_logpdf(double x, double mu, double sigma) {
  double z = (x - mu) / sigma;
  return -0.5 * z * z - std::log(sigma) - 0.91893853320467274178; /* log(sqrt(2pi)) */
}

/* ---- src/utils.jl:16-26 ------------------------------------------------ */
inline double logsumexp(const double* x, size_t n) {
  if (n == 0) return NEG_INF;
  double m = NEG_INF;
  for (size_t i = 0; i < n; ++i) m = std::max(m, x[i]);
  if (m == NEG_INF) return NEG_INF;
  double s = 0;
  for (size_t i = 0; i < n; ++i) s += std::exp(x[i] - m);
  return m + std::log(s);
}

/* ---- AddTypos: src/distributions/add_typos.jl:50-66 -------------------- */
static const double IMPOSSIBLE = -1e5;        /* add_typos.jl:34 */
static const double LETTERS_PER_TYPO = 5.0;   /* add_typos.jl:48 */

/* density given the edit distance; `word_len` = length(word) in characters
 * (add_typos.jl:61-63). max_typos < 0 means `nothing`. */
inline double add_typos_from_distance(int num_typos, int word_len, int max_typos) {
  if (max_typos >= 0 && num_typos > max_typos) return IMPOSSIBLE;
  double l = negbin_logpdf(std::ceil(word_len / LETTERS_PER_TYPO), 0.9, num_typos);
  l -= std::log((double)word_len) * num_typos;
  l -= std::log(26.0) * num_typos / 2;
  return l;
}

/* ---- StringPrior: src/distributions/string_prior.jl:43-61 --------------
 * lm[i] in 0..27 = index in the alphabet a-z,' ','.' of lowercase(char), 255 =
 * not in alphabet.  init_p[28], trans_p[28*28] column-major: trans_p[prev*28+next]
 * is english_letter_transitions[next, prev] (string_prior.jl:32,55). */
static const double UNUSUAL_LETTER_PENALTY = 1000; /* string_prior.jl:41 */

inline double string_prior_logdensity(const uint8_t* lm, int len, int min_len, int max_len,
                                      const double* init_p, const double* trans_p) {
  if (len < min_len || len > max_len) return NEG_INF;
  double score = -std::log((double)(max_len - min_len + 1));
  if (len == 0) return score;
  int prev = -1;
  for (int i = 0; i < len; ++i) {
    const double* dist = prev < 0 ? init_p : trans_p + prev * 28;
    prev = lm[i] == 255 ? -1 : (int)lm[i];
    score += prev < 0 ? -std::log(28.0) : std::max(std::log(dist[prev]), -UNUSUAL_LETTER_PENALTY);
  }
  return score;
}

/* dummy-value mass of a discrete proposal (string_prior.jl:16-22,
 * time_prior.jl:8-14): log1p(-exp(logsumexp(atom_logps))). */
inline double dummy_logmass(const double* atom_logps, size_t n) {
  return std::log1p(-std::exp(logsumexp(atom_logps, n)));
}

/* ---- ChooseUniformly: choose_uniformly.jl:7-10 ------------------------- */
inline double choose_uniformly_logdensity(int n_options) { return -std::log((double)n_options); }

/* ---- ChooseProportionally: choose_proportionally.jl:7-11 (log-sum-exp over
 * duplicate options; logprobs is plain log, utils.jl:33-36) -------------- */
inline double choose_proportionally_logdensity(int observed, const int32_t* options, const double* probs,
                                               int n) {
  std::vector<double> rel;
  for (int i = 0; i < n; ++i)
    if (options[i] == observed) rel.push_back(std::log(probs[i]));
  if (rel.empty()) return NEG_INF;
  return logsumexp(rel.data(), rel.size());
}

/* ---- AddNoise / TransformedGaussian: add_noise.jl:7, transformed_gaussian.jl:15-16
 * Transformation given as (backward value, |g'(backward value)|). */
inline double transformed_gaussian_logdensity(double backward_obs, double abs_deriv, double mean,
                                              double std) {
  return normal_logpdf(backward_obs, mean, std) - std::log(std::fabs(abs_deriv));
}

/* ---- MaybeSwap: maybe_swap.jl:13-28 ------------------------------------ */
inline double maybe_swap_logdensity(bool obs_missing, bool val_in_options, bool same, int n_options,
                                    double prob) {
  if (obs_missing) return val_in_options ? 0.0 : -1000.0;
  if (same) return std::log1p(-prob);
  return std::log(prob) - std::log((double)n_options);
}

/* ---- TimePrior: time_prior.jl:8-14,25-27; regex ^\d?\d:\d\d [ap]\.m\.$ -- */
inline bool time_regex_match(const uint32_t* s, int n) {
  auto dig = [](uint32_t c) { return c >= '0' && c <= '9'; };
  int i = 0;
  if (n < 9 || n > 10) return false;
  if (n == 10) {
    if (!dig(s[i++])) return false;
  }
  if (!dig(s[i++])) return false;
  if (s[i++] != ':') return false;
  if (!dig(s[i++]) || !dig(s[i++])) return false;
  if (s[i++] != ' ') return false;
  if (s[i] != 'a' && s[i] != 'p') return false;
  ++i;
  if (s[i++] != '.') return false;
  if (s[i++] != 'm') return false;
  if (s[i++] != '.') return false;
  return i == n;
}
inline double time_prior_logdensity() { return -std::log(1440.0); }

/* ---- CRP / Pitman–Yor: proposal_compiler.jl:165-171, block_proposal.jl:85-96,
 *      src/model/trace.jl:53-61 ------------------------------------------ */
inline double py_existing_logprob(int64_t count, int64_t total, double strength, double discount) {
  return std::log((double)count - discount) - std::log((double)total + strength);
}
inline double py_new_logprob(int64_t n_rows, int64_t total, double strength, double discount) {
  return std::log(strength + discount * (double)n_rows) - std::log((double)total + strength);
}

/* trace.jl:65-78 */
inline double pitman_yor_score(double strength, double discount, const int64_t* counts, size_t n) {
  double logprob = 0.0;
  int64_t n_refs = 0;
  for (size_t idx = 0; idx < n; ++idx) {
    int64_t n_objects = (int64_t)idx + 1, size = counts[idx];
    logprob += std::log(n_objects * discount + strength) - std::log(n_refs + strength);
    for (int64_t i = 1; i <= size - 1; ++i)
      logprob += std::log(i - discount) - std::log(n_refs + i + strength);
    n_refs += size;
  }
  return logprob;
}

/* ---- particle helpers: row_inference.jl:76-85 --------------------------- */
inline double effective_sample_size(const double* logw, size_t n) {
  double tot = logsumexp(logw, n);
  std::vector<double> t(n);
  for (size_t i = 0; i < n; ++i) t[i] = 2.0 * (logw[i] - tot);
  return std::exp(-logsumexp(t.data(), n));
}

} /* namespace pco */
#endif
