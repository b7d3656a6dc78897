// fk_root_fast_kernel — the dominant kernel: rows x candidate referents of a block
// root (hospital Record block 1: K ~ 1e4 latent hospitals x 11 AddTypos terms).
//
// Same contract and bit-identical results as enum_node_kernel (enum_kernels.hip),
// restructured for the memory system of MI355X:
//   * candidate-compact byte tables comp_f[o][k] (built by compact_pair_kernel when
//     the latent table's columns change) turn the pair-table gather into F
//     contiguous byte streams per row: every lane reads one dword = 4 candidates,
//     a wave reads 256 consecutive bytes -> fully coalesced, HBM/L2 friendly
//     (consecutive rows of the same hospital re-read the same byte rows from L2);
//   * candidate word lengths clen_f[k] are streamed the same way (L2 resident);
//   * the AddTypos density (add_typos.jl:61-63) is a (length, distance) LUT built
//     once per workgroup in LDS with the very fp64 operation order of
//     term_density(), so per term the inner loop is: 2 dword loads, 4 byte
//     extracts, 4 ds_read_b64, 4 fp64 adds;
//   * CRP priors are precomputed per candidate (prior_e / prior_n);
//   * one workgroup of up to 1024 lanes (16 wavefronts) per row keeps the score
//     vector (8 B x K) in LDS at 1 workgroup/CU with enough waves to hide latency.
// Scores are stored to LDS and then go through exactly the phases 2-5 of the generic
// kernel (max, fixed-point weights, natural-order chunk scan, Philox draws).
#include <algorithm>

#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"
#include "enum.h"

#define HALF_LOG26 1.629048269010741
#define ADD_TYPOS_IMPOSSIBLE (-1e5)
#define FAST_MAX_WAVES 16

__global__ void compact_pair_kernel(const uint8_t* __restrict__ pair, int n_obs, int n_lat,
                                    const int32_t* __restrict__ cand_col, int n_cand, int kpad,
                                    uint8_t* __restrict__ comp) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (k >= kpad) return;
  uint8_t v = 0;
  if (k < n_cand) v = pair[(size_t)o * n_lat + cand_col[k]];
  comp[(size_t)o * kpad + k] = v;
}
__global__ void compact_len_kernel(const uint16_t* __restrict__ lat_len, const int32_t* __restrict__ cand_col,
                                   int n_cand, int kpad, uint8_t* __restrict__ clen) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= kpad) return;
  clen[k] = k < n_cand ? (uint8_t)lat_len[cand_col[k]] : (uint8_t)0;
}
__global__ void priors_kernel(const int64_t* __restrict__ counts, const double* __restrict__ logc_full, int n_cand,
                              int kpad, double logden_e, double logden_n, double* __restrict__ prior_e,
                              double* __restrict__ prior_n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= kpad) return;
  const bool live = k < n_cand && counts[k] != 0;
  prior_e[k] = live ? logc_full[k] - logden_e : -__builtin_inf();
  prior_n[k] = live ? logc_full[k] - logden_n : -__builtin_inf();
}

int pclean_build_compact(pclean_ctx* ctx, const uint8_t* pair, int n_obs, int n_lat, const int32_t* cand_col,
                         const uint16_t* lat_len, int n_cand, int kpad, uint8_t* comp, uint8_t* clen) {
  hipLaunchKernelGGL(compact_pair_kernel, dim3((kpad + 255) / 256, n_obs), dim3(256), 0, ctx->stream, pair, n_obs,
                     n_lat, cand_col, n_cand, kpad, comp);
  hipLaunchKernelGGL(compact_len_kernel, dim3((kpad + 255) / 256), dim3(256), 0, ctx->stream, lat_len, cand_col, n_cand,
                     kpad, clen);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
int pclean_build_priors(pclean_ctx* ctx, const int64_t* counts, const double* logc_full, int n_cand, int kpad,
                        double logden_e, double logden_n, double* prior_e, double* prior_n) {
  hipLaunchKernelGGL(priors_kernel, dim3((kpad + 255) / 256), dim3(256), 0, ctx->stream, counts, logc_full, n_cand,
                     kpad, logden_e, logden_n, prior_e, prior_n);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

__device__ __forceinline__ double wave_max64(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(1024) void fk_root_fast_kernel(const FastRootDev fr, const DensDev dn, const ItemsDev it,
                                                            const ChildrenDev ch, uint64_t seed, uint32_t sweep,
                                                            uint32_t site, int n_draws, int item_base,
                                                            double* __restrict__ lse_out,
                                                            int32_t* __restrict__ draws_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T = blockDim.x, nw = T >> 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.x + item_base;
  const int n = fr.n_cand, nc = n + 1;
  double* s = (double*)smem;                                   // [nc] scores, later uint64 weights
  uint64_t* u = (uint64_t*)smem;
  double* lut = (double*)(smem + (size_t)((nc + 1) & ~1) * 8);  // [(lmax+1) * dstride]
  const int lut_n = (fr.lmax + 1) * fr.dstride;
  double* red = lut + ((lut_n + 1) & ~1);                       // [16]
  uint64_t* wsum = (uint64_t*)(red + FAST_MAX_WAVES);           // [16]
  uint64_t* xs = wsum + FAST_MAX_WAVES;                         // [64] draw thresholds
  uint64_t* rowp = xs + 64;                                     // [16] comp row pointers (0 = missing obs)

  // density LUT: the fp64 operation order of term_density() (enum_kernels.hip)
  for (int i = tid; i < lut_n; i += T) {
    const int L = i / fr.dstride, d = i - L * fr.dstride;
    const int r = (L + 4) / 5;
    double l = dn.nb[(size_t)r * dn.nb_stride + d];
    l -= dn.logl[L] * (double)d;
    l -= HALF_LOG26 * (double)d;
    lut[i] = l;
  }
  const int row = it.row ? it.row[t] : t;
  const int excl = it.excl ? it.excl[t] : -1;
  const bool excluded = excl >= 0;
  const bool deleted = excluded && fr.counts[excl] <= 1;
  const double logden = excluded ? fr.scal[1] : fr.scal[0];
  const double* prior = excluded ? fr.prior_e : fr.prior_n;
  if (tid < PCLEAN_MAX_TERMS) {
    uint64_t p = 0;
    if (tid < fr.n_terms) {
      const int o = fr.terms[tid].obs_col[row];
      if (o >= 0) p = (uint64_t)(fr.terms[tid].comp + (size_t)o * fr.kpad);
    }
    rowp[tid] = p;
  }
  __syncthreads();

  // ---- phase 1: scores, 4 consecutive candidates per lane per round -----------
  double lmax = -__builtin_inf();
  const int nslots = fr.kpad >> 2;
  for (int slot = tid; slot < nslots; slot += T) {
    const int k0 = slot << 2;
    double acc[4];
    const double2 p01 = *reinterpret_cast<const double2*>(prior + k0);
    const double2 p23 = *reinterpret_cast<const double2*>(prior + k0 + 2);
    acc[0] = p01.x;
    acc[1] = p01.y;
    acc[2] = p23.x;
    acc[3] = p23.y;
    if (excluded && (unsigned)(excl - k0) < 4u) {
      const double pe = deleted ? -__builtin_inf() : fr.logc_m1[excl] - logden;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e == excl) acc[e] = pe;
    }
    uint32_t c4[PCLEAN_MAX_TERMS], l4[PCLEAN_MAX_TERMS];
#pragma unroll
    for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) {
      if (f < fr.n_terms) {
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(rowp[f]);
        c4[f] = rp ? rp[slot] : 0u;
        l4[f] = reinterpret_cast<const uint32_t*>(fr.terms[f].clen)[slot];
      }
    }
#pragma unroll
    for (int f = 0; f < PCLEAN_MAX_TERMS; ++f) {
      if (f < fr.n_terms) {
        if (rowp[f]) {  // explicitly missing observation contributes nothing (add_typos.jl:51-53)
          const int mt = fr.terms[f].max_typos;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int d = (c4[f] >> (8 * e)) & 255, L = (l4[f] >> (8 * e)) & 255;
            double dens = lut[L * fr.dstride + d];
            if (mt >= 0 && d > mt) dens = ADD_TYPOS_IMPOSSIBLE;
            acc[e] += dens;
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (k0 + e < n) {
        s[k0 + e] = acc[e];
        lmax = fmax(lmax, acc[e]);
      }
  }
  if (tid == 0) {
    double snew = 0.0;
    for (int c = 0; c < ch.n; ++c) {
      size_t idx = (size_t)t;
      if (ch.obs_col[c]) {
        const int o = ch.obs_col[c][row];
        idx = o < 0 ? (size_t)ch.n_obs[c] : (size_t)o;
      }
      snew += ch.arr[c][idx];
    }
    const double sn = ((deleted ? fr.scal[3] : fr.scal[2]) - logden) + snew;
    s[n] = sn;
    lmax = fmax(lmax, sn);
  }
  // ---- phase 2: max ------------------------------------------------------------
  lmax = wave_max64(lmax);
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  double m = red[0];
  for (int w = 1; w < nw; ++w) m = fmax(m, red[w]);
  // ---- phase 3: fixed-point weights in place --------------------------------------
  for (int k = tid; k < nc; k += T) {
    const double sk = s[k];
    u[k] = (m == -__builtin_inf()) ? 0ull : pclean_fixw(sk - m);
  }
  __syncthreads();
  // ---- phase 4: contiguous chunk sums + block scan -----------------------------------
  const int chunk = (nc + T - 1) / T;
  const int lo = min(tid * chunk, nc), hi = min(lo + chunk, nc);
  uint64_t part = 0;
  for (int k = lo; k < hi; ++k) part += u[k];
  unsigned long long incl = part;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long x = __shfl_up(incl, o, 64);
    if (lane >= o) incl += x;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint64_t base = 0, U = 0;
  for (int w = 0; w < nw; ++w) {
    if (w < wave) base += wsum[w];
    U += wsum[w];
  }
  const uint64_t pre = base + incl - part;
  // ---- phase 5: lse + draws -------------------------------------------------------------
  if (tid == 0 && lse_out) lse_out[t] = pclean_lse_from_fix(m, U);
  if (n_draws > 0) {
    // one Philox evaluation per draw for the whole workgroup (lane j of wave 0), broadcast through LDS
    const uint32_t rng_row = (uint32_t)((int64_t)row + it.row_offset);
    for (int j0 = 0; j0 < n_draws; j0 += 64) {
      __syncthreads();
      if (tid < 64 && j0 + tid < n_draws) {
        const uint32_t pid = it.particle ? (uint32_t)it.particle[t] : (uint32_t)(j0 + tid);
        xs[tid] = U ? pclean_mulhi64(pclean_rand64(seed, rng_row, site, pid, sweep), U) : 0ull;
      }
      __syncthreads();
      const int jn = min(64, n_draws - j0);
      for (int j = 0; j < jn; ++j) {
        int32_t* dst = draws_out + (size_t)t * n_draws + j0 + j;
        if (U == 0) {
          if (tid == 0) *dst = PCLEAN_CHOICE_NEW;
          continue;
        }
        const uint64_t x = xs[j];
        if (x >= pre && x < pre + part) {
          uint64_t acc = pre;
          int k = lo;
          for (; k < hi; ++k) {
            acc += u[k];
            if (acc > x) break;
          }
          *dst = k == n ? PCLEAN_CHOICE_NEW : k;
        }
      }
    }
  }
}

int pclean_launch_root_fast(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const ChildrenDev& ch,
                            uint64_t seed, uint32_t sweep, uint32_t site, int n_draws, double* lse_out,
                            int32_t* draws_out) {
  if (it.n <= 0) return PCLEAN_OK;
  const int nc = fr.n_cand + 1;
  const int lut_n = (fr.lmax + 1) * fr.dstride;
  const size_t lds = (size_t)((nc + 1) & ~1) * 8 + (size_t)((lut_n + 1) & ~1) * 8 + (2 * FAST_MAX_WAVES + 64 + 16) * 8;
  // lanes: as few rounds of 4-candidate slots as possible with little idle tail
  const int nslots = fr.kpad >> 2;
  const int rounds = (nslots + 1023) / 1024;
  int T = ((nslots + rounds - 1) / rounds + 63) / 64 * 64;
  T = std::max(256, std::min(1024, T));
  DensDev dn{ctx->nb.p, ctx->logl.p, ctx->max_d + 1, 0};
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(ctx, hipFuncSetAttribute((const void*)fk_root_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
    attr_set = true;
  }
  const int kMaxBlocks = 2 * 1024 * 1024;  // < 2^32 threads per launch
  for (int base = 0; base < it.n; base += kMaxBlocks)
    hipLaunchKernelGGL(fk_root_fast_kernel, dim3(std::min(kMaxBlocks, it.n - base)), dim3(T), lds, ctx->stream, fr, dn,
                       it, ch, seed, sweep, site, n_draws, base, lse_out, draws_out);
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
