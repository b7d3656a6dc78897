/* oracle/random.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of `random(dist, args...)` of the reference's noise models, each written
 * after its Julia method and consuming Philox draws in the order fixed by
 * include/pclean_hip.h (section "random(dist, args...)"):
 *   AddTypos        src/distributions/add_typos.jl:9-45   (perform_typo, NegativeBinomial count)
 *   StringPrior     src/distributions/string_prior.jl:28-39
 *   ChooseProportionally / ChooseUniformly   choose_proportionally.jl:3-5, choose_uniformly.jl:3-5
 *   AddNoise / TransformedGaussian           add_noise.jl:5, transformed_gaussian.jl:13
 *   MaybeSwap       maybe_swap.jl:5-11
 *   TimePrior       time_prior.jl:21-23
 * The reference draws from Julia's global RNG; bit-level parity with Julia is impossible
 * (parity unpinned, see README.md) — the tests pin the *distributions* (moments, frequencies,
 * edit-distance bounds) and the device against this restatement draw for draw.
 */
#ifndef PCLEAN_ORACLE_RANDOM_H
#define PCLEAN_ORACLE_RANDOM_H

#include <cmath>
#include <cstdint>
#include <vector>

#include "../include/pclean_detmath.h"
#include "../include/pclean_hip.h"
#include "../include/pclean_philox.h"
#include "enumerate.h"

namespace pco {

class DrawStream {
 public:
  DrawStream(uint64_t seed, uint32_t elem, uint32_t kind, uint32_t stream)
      : seed_(seed), elem_(elem), site_(PCLEAN_SITE_RANDOM(kind)), stream_(stream) {}
  uint64_t bits() { return pclean_rand64(seed_, elem_, site_, t_++, stream_); }
  /* rand(DiscreteUniform(lo, hi)) */
  int uniform_int(int lo, int hi) { return lo + (int)pclean_mulhi64(bits(), (uint64_t)(hi - lo + 1)); }

 private:
  uint64_t seed_;
  uint32_t elem_, site_, stream_, t_ = 0;
};

/* rand(NegativeBinomial(r, 0.9)): failures before the r-th success of Bernoulli(0.9) trials */
inline int negative_binomial_p90(DrawStream& d, int r) {
  int failures = 0, successes = 0;
  while (successes < r) {
    if (d.bits() < PCLEAN_P10_U64)
      ++failures;
    else
      ++successes;
  }
  return failures;
}

/* add_typos.jl:9-33 on a code-point vector; `cap` bounds the length (device buffer stride) */
enum Typo { INSERT = 1, DELETE = 2, TRANSPOSE = 3, SUBSTITUTE = 4 };
inline void perform_typo(DrawStream& d, Typo typo, std::vector<uint32_t>& word, size_t cap) {
  const int L = (int)word.size();
  switch (typo) {
    case INSERT: {
      if (word.size() >= cap) return;
      int index = d.uniform_int(0, L); /* letters before the inserted one */
      uint32_t letter = 'a' + (uint32_t)(d.uniform_int(1, 26) - 1);
      word.insert(word.begin() + index, letter);
      return;
    }
    case DELETE: {
      if (L == 0) return;
      int index = d.uniform_int(1, L);
      word.erase(word.begin() + (index - 1));
      return;
    }
    case SUBSTITUTE: {
      if (L == 0) return;
      int index = d.uniform_int(1, L);
      word[index - 1] = 'a' + (uint32_t)(d.uniform_int(1, 26) - 1);
      return;
    }
    case TRANSPOSE: {
      if (L < 2) return; /* the reference returns `nothing` for one-letter words (26-28) */
      int index = d.uniform_int(1, L - 1);
      std::swap(word[index - 1], word[index]);
      return;
    }
  }
}

/* add_typos.jl:36-45 */
inline std::vector<uint32_t> random_add_typos(const uint32_t* cp, int len, int max_typos, uint64_t seed, uint32_t elem,
                                              uint32_t stream, size_t cap) {
  DrawStream d(seed, elem, PCLEAN_RANDOM_ADD_TYPOS, stream);
  std::vector<uint32_t> word(cp, cp + std::min<size_t>((size_t)len, cap));
  int num_typos = negative_binomial_p90(d, (int)std::ceil((double)word.size() / 5.0));
  if (max_typos >= 0) num_typos = std::min(max_typos, num_typos);
  static const Typo kinds[4] = {INSERT, DELETE, TRANSPOSE, SUBSTITUTE};
  for (int i = 0; i < num_typos; ++i) {
    Typo typo = kinds[d.uniform_int(1, 4) - 1];
    perform_typo(d, typo, word, cap);
  }
  return word;
}

/* rand(Categorical(normalize(p))) over non-negative weights, fixed-point inverse CDF */
inline int categorical_fixed(const double* p, int n, uint64_t R) {
  std::vector<uint64_t> w(n);
  uint64_t total = 0;
  for (int j = 0; j < n; ++j) total += (w[j] = (uint64_t)std::floor(p[j] * 1099511627776.0));
  const uint64_t x = pclean_mulhi64(R, total);
  uint64_t acc = 0;
  for (int j = 0; j < n; ++j)
    if ((acc += w[j]) > x) return j;
  return n - 1;
}

/* string_prior.jl:28-39; trans[prev*28 + next] */
inline std::vector<uint8_t> random_string_prior(int min_len, int max_len, const double* init, const double* trans,
                                                uint64_t seed, uint32_t elem, uint32_t stream) {
  DrawStream d(seed, elem, PCLEAN_RANDOM_STRING_PRIOR, stream);
  const int len = d.uniform_int(min_len, max_len);
  std::vector<uint8_t> letters;
  for (int i = 1; i <= len; ++i) {
    const double* dist = (i == 1) ? init : trans + 28 * (size_t)letters.back();
    letters.push_back((uint8_t)categorical_fixed(dist, 28, d.bits()));
  }
  return letters;
}

/* choose_proportionally.jl:3-5 with log-weights (logprobs(), utils.jl:33-36) */
inline int random_categorical(const double* logp, int n, uint64_t seed, uint32_t elem, uint32_t stream) {
  DrawStream d(seed, elem, PCLEAN_RANDOM_CATEGORICAL, stream);
  std::vector<double> s(logp, logp + n);
  FixSum f = fix_sum(s);
  return fix_draw(s, f, d.bits());
}

/* rand(Normal(mean, std)) by the Marsaglia polar method, then t.forward = multiplication */
inline double random_normal(double mean, double std, double fwd_scale, uint64_t seed, uint32_t elem, uint32_t stream) {
  DrawStream d(seed, elem, PCLEAN_RANDOM_NORMAL, stream);
  auto pm1 = [&]() { return 2.0 * (((double)(d.bits() >> 11) + 0.5) * 0x1.0p-53) - 1.0; };
  for (;;) {
    const double u = pm1();
    const double v = pm1();
    const double s = u * u + v * v;
    if (s > 0.0 && s < 1.0) {
      const double z = u * std::sqrt(-2.0 * pclean_log(s) / s);
      return fwd_scale * (mean + std * z);
    }
  }
}

/* maybe_swap.jl:5-11: -1 = val, else 0-based index into options */
inline int random_maybe_swap(double prob, int n_options, uint64_t seed, uint32_t elem, uint32_t stream) {
  DrawStream d(seed, elem, PCLEAN_RANDOM_MAYBE_SWAP, stream);
  const uint64_t b = d.bits();
  const uint64_t o = d.bits();
  const bool swap = prob >= 1.0 || (prob > 0.0 && b < (uint64_t)(prob * 18446744073709551616.0));
  return swap ? (int)pclean_mulhi64(o, (uint64_t)n_options) : -1;
}

/* time_prior.jl:21-23 */
inline void random_time_prior(uint64_t seed, uint32_t elem, uint32_t stream, int32_t out[3]) {
  DrawStream d(seed, elem, PCLEAN_RANDOM_TIME_PRIOR, stream);
  out[0] = d.uniform_int(1, 12);
  out[1] = d.uniform_int(1, 60);
  out[2] = (int32_t)(d.bits() >> 63);
}

}  // namespace pco
#endif
