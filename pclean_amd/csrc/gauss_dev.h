// Device/host-shared evaluation of the Gaussian term (pclean_gauss): marginal over the
// enumerated locals of  logp(locals) + logpdf(Normal(mean, sigma), backward(x)) - log|g'|
// (src/distributions/transformed_gaussian.jl:15-16, add_noise.jl:7, choose_uniformly.jl:7-10).
// The oracle (oracle/enumerate.h) restates the same operation order.
#pragma once
#include "../../include/pclean_detmath.h"
#include "enum.h"

#define PCLEAN_LOG_SQRT_2PI 0.91893853320467274178

__device__ __forceinline__ double gauss_normal_logpdf(double x, double mu, double sigma, double log_sigma) {
  const double z = (x - mu) / sigma;
  return -0.5 * z * z - log_sigma - PCLEAN_LOG_SQRT_2PI;
}

// Scores of the (l0, l1) combinations for candidate value source `cand(d)`; returns the
// number of combinations written to sc[] / codes[] (code = l0 * 16 + l1).  base_fn(d) gives
// the value of index dimension d for the non-local kinds.
template <typename ValFn>
__device__ __forceinline__ int gauss_combo_scores(const GaussDev& g, int row, const int32_t* evctx, ValFn val, double* sc,
                                                  int* codes) {
  const double xv = g.x[row];
  int base = 0, lstride[2] = {0, 0};
  for (int d = 0; d < g.n_dims; ++d) {
    if (g.src_kind[d] == PCLEAN_GSRC_LOCAL)
      lstride[g.src_slot[d]] = g.stride[d];
    else
      base += g.stride[d] * val(d);
  }
  int lo[2] = {0, 0}, hi[2] = {1, 1};
  double lp[2] = {0.0, 0.0};
  for (int l = 0; l < g.n_locals; ++l) {
    lo[l] = 0;
    hi[l] = g.local_n[l];
    lp[l] = g.local_logp[l];
    if (g.local_obs[l]) {
      const int v = g.local_obs[l][row];
      if (v >= 0) {
        lo[l] = v;
        hi[l] = v + 1;
      }
    }
  }
  int n = 0;
  for (int l0 = lo[0]; l0 < hi[0]; ++l0)
    for (int l1 = lo[1]; l1 < hi[1]; ++l1) {
      const int idx = base + lstride[0] * l0 + lstride[1] * l1;
      int u = 0;
      if (g.t_kind == PCLEAN_GSRC_LOCAL)
        u = g.t_src == 0 ? l0 : l1;
      else if (g.t_kind == PCLEAN_GSRC_EVCTX)
        u = evctx[g.t_src];
      double s = lp[0] + lp[1];
      s += gauss_normal_logpdf(g.tx[u] ? g.tx[u][row] : xv * g.t_scale[u], g.mu[idx], g.sigma, g.log_sigma);
      s -= g.tl[u] ? g.tl[u][row] : g.t_lad[u];
      sc[n] = s;
      codes[n] = l0 * 16 + l1;
      ++n;
    }
  return n;
}

// marginal (fixed-point log-sum-exp) of the combination scores; a single combination is returned as is
__device__ __forceinline__ double gauss_lse(const double* sc, int n) {
  if (n == 1) return sc[0];
  double m = -__builtin_inf();
  for (int i = 0; i < n; ++i) m = fmax(m, sc[i]);
  if (m == -__builtin_inf()) return m;
  uint64_t U = 0;
  for (int i = 0; i < n; ++i) U += pclean_fixw(sc[i] - m);
  return pclean_lse_from_fix(m, U);
}
