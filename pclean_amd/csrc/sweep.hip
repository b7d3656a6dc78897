// Batched rejuvenation sweep of the observed class: host orchestration of the
// static enumeration plan + the small particle kernels.
//
// Reference semantics followed (files under /root/reference/src):
//   run_smc!           inference/row_inference.jl:108-187 (particles, blocks, final choice)
//   maybe_resample     row_inference.jl:87-105 (ESS < P/2, multinomial, retained first)
//   make_block_proposal! / propose_non_enumerable!  block_proposal.jl:24-191: when every
//       latent choice of a block is enumerated the incremental weight p - q equals the
//       block's log-marginal (SURVEY §3.3 "key structural fact"), so one enumeration per
//       (row, block, context) serves all particles and only the draws are per particle.
//   process_plan!      proposal_compiler.jl:363-388 (children of a new row enumerated
//       independently; log-marginals added)
#include <algorithm>
#include <cmath>
#include <map>
#include <set>

#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include "../../include/pclean_detmath.h"
#include "../../include/pclean_philox.h"
#include "dummy_dev.h"
#include "enum.h"
#include "gauss_dev.h"
#include "sweep_state.h"

// every blocking point of the orchestration goes through here: PCLEAN_TRACE_SYNC=1 lists them per call
static int g_sync_count = 0;

// One 32-bit count from the device to the host in the middle of a call (how many items need the next step): instead of a
// copy + stream synchronisation (~25 us until the host thread is woken) a one-thread kernel publishes the value in
// page-locked memory and the host spins on a sequence number — the remaining host round trips of a sweep cost a few
// microseconds each.  PCLEAN_NO_POLL=1: the plain copy + synchronisation.
__global__ void publish_count_kernel(const unsigned int* __restrict__ src, volatile unsigned int* __restrict__ dst, unsigned int seq) {
  dst[0] = *src;
  __threadfence_system();
  dst[1] = seq;
}
static int read_count(pclean_ctx* ctx, const void* dev, void* out, const char* func, int line) {
  SweepState* s = st(ctx);
  static const bool no_poll = getenv("PCLEAN_NO_POLL") != nullptr;
  static const bool trace = getenv("PCLEAN_TRACE_SYNC") != nullptr;
  if (trace) fprintf(stderr, "[pclean sync %d] %s:%d (count)\n", ++g_sync_count, func, line);
  if (!no_poll && !s->h_poll) {
    if (hipHostMalloc((void**)&s->h_poll, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) s->h_poll = nullptr;
    if (s->h_poll) {
      s->h_poll[0] = s->h_poll[1] = 0;
      if (hipHostGetDevicePointer((void**)&s->d_poll, (void*)s->h_poll, 0) != hipSuccess) {
        (void)hipHostFree((void*)s->h_poll);
        s->h_poll = nullptr;
      }
    }
  }
  if (no_poll || !s->h_poll) {
    HIPCHK(ctx, hipMemcpyAsync(out, dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PCLEAN_OK;
  }
  const unsigned int seq = ++s->poll_seq;
  hipLaunchKernelGGL(publish_count_kernel, dim3(1), dim3(1), 0, ctx->stream, (const unsigned int*)dev, s->d_poll, seq);
  volatile unsigned int* h = s->h_poll;
  for (long spins = 0; h[1] != seq; ++spins)
    if (spins > 2000000) {  // (a failed launch would spin for ever: let the runtime report it)
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (h[1] != seq) return pclean_fail(ctx, PCLEAN_ERR_HIP, "count read-back: the publishing kernel did not run");
    }
  *(unsigned int*)out = h[0];
  return PCLEAN_OK;
}
#define PCLEAN_READ_COUNT(ctx, dev, out)                                  \
  do {                                                                    \
    const int rc_ = read_count(ctx, (dev), (out), __func__, __LINE__);    \
    if (rc_) return rc_;                                                  \
  } while (0)

#define PCLEAN_SYNC(ctx)                                                                                   \
  do {                                                                                                     \
    static const bool trace_ = getenv("PCLEAN_TRACE_SYNC") != nullptr;                                     \
    if (trace_) fprintf(stderr, "[pclean sync %d] %s:%d\n", ++g_sync_count, __func__, __LINE__);          \
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                                                        \
  } while (0)

__device__ int32_t resolve_new_value(const PlanDev& pl, int node, int col, const int32_t* vals) {
  for (int depth = 0; depth < 16; ++depth) {
    const int cn = pl.colmap[2 * (pl.colmap_begin[node] + col)];
    const int cc = pl.colmap[2 * (pl.colmap_begin[node] + col) + 1];
    if (cn < 0) return -1;
    const int choice = vals[cn];
    if (pl.kind[cn] == PCLEAN_NODE_LEAF) return pl.cols[cn][choice];
    if (choice >= 0) return pl.cols[cn][(size_t)cc * pl.n_rows[cn] + choice];
    node = cn;
    col = cc;
  }
  return -1;
}

// ---------------------------------------------------------------------------
// small kernels
__global__ void iota_missing_kernel(int32_t* p, int n_obs) {  // [0..n_obs-1, -1]
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_obs) p[i] = i < n_obs ? i : -1;
}

__global__ void fill_i32_kernel(int32_t* p, size_t n, int32_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void fill_f64_kernel(double* p, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// excl_child[t] = row of the child's table that loses a reference because the
// parent's excluded row is garbage-collected (dependency_tracking.jl:189-201)
__global__ void derive_excl_kernel(int n, const int32_t* parent_excl, const int64_t* parent_counts,
                                   const int32_t* parent_fk_col, int32_t* out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int e = parent_excl ? parent_excl[t] : -1;
  out[t] = (e >= 0 && parent_counts[e] <= 1) ? parent_fk_col[e] : -1;
}

// Particle arrays of a sweep are PARTICLE-MAJOR: slot(p, i) = p * N + i (w, pchoice, pnewpos, draws, ctx), so
// that one-thread-per-row kernels read them coalesced and per-slot kernels stay coalesced as well.
struct CtxSrc {
  int32_t n_ctx;
  const int32_t* pchoice[PCLEAN_MAX_CTX];  // [P][N] of the source block
  const int32_t* pnewpos[PCLEAN_MAX_CTX];
  const int32_t* vals[PCLEAN_MAX_CTX];     // [n_new][n_nodes] of the source block
  int32_t n_nodes[PCLEAN_MAX_CTX];
  const int32_t* root_col[PCLEAN_MAX_CTX];  // column of the source block's root table
  int32_t col[PCLEAN_MAX_CTX];
  PlanDev plan[PCLEAN_MAX_CTX];
};

// ctx value c of slot t -> it_ctx[c * NP + t]
__global__ void gather_ctx_kernel(size_t NP, CtxSrc cs, int32_t* it_ctx) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NP) return;
  for (int s = 0; s < cs.n_ctx; ++s) {  // (slots >= n_ctx of the buffer were zeroed once, when it was sized)
    int32_t v;
    const int choice = cs.pchoice[s][t];
    if (choice >= 0)
      v = cs.root_col[s][choice];
    else
      v = resolve_new_value(cs.plan[s], 0, cs.col[s], cs.vals[s] + (size_t)cs.pnewpos[s][t] * cs.n_nodes[s]);
    it_ctx[(size_t)s * NP + t] = v;
  }
}

// distinct contexts among the particles of a row: rep[slot(p,i)] = first particle q <= p of the row with the
// same ctx tuple; n_distinct[i] = number of representatives.  Typical rows: every particle shares one context.
__global__ void ctx_count_kernel(int N, int P, int n_ctx, const int32_t* __restrict__ it_ctx, int32_t* __restrict__ rep,
                                 int32_t* __restrict__ n_distinct) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const size_t NP = (size_t)N * P;
  int nd = 0;
  for (int p = 0; p < P; ++p) {
    const size_t sp = (size_t)p * N + i;
    int c[PCLEAN_MAX_CTX];
    for (int k = 0; k < PCLEAN_MAX_CTX; ++k) c[k] = k < n_ctx ? it_ctx[(size_t)k * NP + sp] : 0;
    int q = 0;
    for (; q < p; ++q) {
      bool same = true;
      for (int k = 0; k < PCLEAN_MAX_CTX; ++k)
        if (k < n_ctx) same &= it_ctx[(size_t)k * NP + (size_t)q * N + i] == c[k];
      if (same) break;
    }
    rep[sp] = q;
    nd += q == p ? 1 : 0;
  }
  n_distinct[i] = nd;
}
// items off[i] .. off[i] + n_distinct[i]) of row i, in particle order of their representatives
__global__ void ctx_fill_kernel(int N, int P, int n_ctx, const int32_t* __restrict__ it_ctx, const int32_t* __restrict__ rep,
                                const int32_t* __restrict__ off, const int32_t* __restrict__ cur_b,
                                int32_t* __restrict__ slot_item, int32_t* __restrict__ row, int32_t* __restrict__ ctxv,
                                int32_t* __restrict__ excl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const size_t NP = (size_t)N * P;
  int j = off[i];
  for (int p = 0; p < P; ++p) {
    const size_t sp = (size_t)p * N + i;
    const int q = rep[sp];
    if (q == p) {
      row[j] = i;
      excl[j] = cur_b ? cur_b[i] : -1;
      for (int k = 0; k < PCLEAN_MAX_CTX; ++k) ctxv[(size_t)j * PCLEAN_MAX_CTX + k] = k < n_ctx ? it_ctx[(size_t)k * NP + sp] : 0;
      slot_item[sp] = j++;
    } else {
      slot_item[sp] = slot_item[(size_t)q * N + i];  // written above by this very thread
    }
  }
}
// (PU_T = 1024 threads per workgroup: the ONE returning atomic per workgroup on the same counter is served at ~30 ns
// apiece at the memory side — 3 900 workgroups of 256 were a 0.12 ms floor of a 0.17 ms kernel)
#define PU_T 1024
// One thread per row, after a block's root enumeration: particle p of row i takes its draw (row-major from the
// root kernels: draws_rm[i * P + p], or draw p of its context's item) and the block's log-marginal; particle 0
// keeps the retained referent under CSMC (row_inference.jl:143-145).  Writes the particle-major arrays
// coalesced and emits the list of the particle slots that proposed a NEW referent (new_list[pos] = slot,
// pnewpos[slot] = pos; positions reserved with one atomic per workgroup, order irrelevant: every use is keyed
// by (row, particle)).
__global__ __launch_bounds__(PU_T) void particle_update_kernel(int N, int P, const int32_t* __restrict__ draws_rm,
                                                              const double* __restrict__ lse,
                                                              const int32_t* __restrict__ slot_item,
                                                              const int32_t* __restrict__ draws_item,
                                                              const double* __restrict__ lse_item,
                                                              const int32_t* __restrict__ cur_b,
                                                              int32_t* __restrict__ pchoice, double* __restrict__ w,
                                                              unsigned int* __restrict__ n_new,
                                                              int32_t* __restrict__ new_list,
                                                              int32_t* __restrict__ pnewpos, int first) {
  __shared__ unsigned int wsum[PU_T / 64];
  __shared__ unsigned int bbase;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t newmask = 0;
  if (i < N) {
    const int keep = cur_b ? cur_b[i] : -1;
    for (int p = 0; p < P; ++p) {
      const size_t sp = (size_t)p * N + i;
      int d;
      double l;
      if (slot_item) {
        const int item = slot_item[sp];
        d = draws_item[(size_t)item * P + p];
        l = lse_item[item];
      } else {
        d = draws_rm[(size_t)i * P + p];
        l = lse[i];
      }
      const int c = (p == 0 && keep >= 0) ? keep : d;
      pchoice[sp] = c;
      w[sp] = first ? 0.0 + l : w[sp] + l;  // (first block of the sweep: the weights start at +0.0)
      if (c == PCLEAN_CHOICE_NEW) newmask |= 1ull << p;
    }
  }
  const unsigned int mine = (unsigned int)__popcll(newmask);
  unsigned int incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int x = __shfl_up(incl, o, 64);
    if (lane >= o) incl += x;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int total = 0;
    for (int k = 0; k < PU_T / 64; ++k) total += wsum[k];
    bbase = total ? atomicAdd(n_new, total) : 0u;
  }
  __syncthreads();
  if (!mine) return;
  unsigned int pos = bbase + incl - mine;
  for (int k = 0; k < wave; ++k) pos += wsum[k];
  for (uint64_t mm = newmask; mm; mm &= mm - 1) {
    const int p = __builtin_ctzll(mm);
    const size_t sp = (size_t)p * N + i;
    new_list[pos] = (int32_t)sp;
    pnewpos[sp] = (int32_t)pos;
    ++pos;
  }
}

// rows whose CHOSEN particle proposed a NEW referent: slot list + positions (deferred new-row sampling of the last block)
__global__ void chosen_new_kernel(int N, const int32_t* __restrict__ chosen, const int32_t* __restrict__ pchoice,
                                  unsigned int* __restrict__ counter, int32_t* __restrict__ list,
                                  int32_t* __restrict__ pnewpos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const size_t s = (size_t)chosen[i] * N + i;
  if (pchoice[s] != PCLEAN_CHOICE_NEW) return;
  const unsigned int pos = atomicAdd(counter, 1u);
  list[pos] = (int32_t)s;
  pnewpos[s] = (int32_t)pos;
}

// ---- pure scoring block (flights Obs block 3): p += logdensity(MaybeSwap, ...) ---------------------
struct SrcDev {
  const int32_t* pchoice;
  const int32_t* pnewpos;
  const int32_t* vals;
  const int32_t* root_col;
  int32_t n_nodes, col;
  PlanDev plan;
};
__device__ __forceinline__ int src_value(const SrcDev& s, size_t slot) {
  const int choice = s.pchoice[slot];
  if (choice >= 0) return s.root_col[choice];
  return resolve_new_value(s.plan, 0, s.col, s.vals + (size_t)s.pnewpos[slot] * s.n_nodes);
}
struct ScoreTermDev {
  const int32_t* obs_col;
  const uint8_t* pair;
  const int32_t* nopt_fn;
  int32_t n_lat, other_val;
  SrcDev val, key;
};
struct ScoreBlockDev {
  int32_t n_terms, prob_nb;
  const int32_t* prob_fn;
  const double* prob_same;
  const double* prob_diff;
  const double* logn;
  SrcDev pa, pb;
  ScoreTermDev t[8];
};
__global__ void score_block_kernel(int n_rows, int P, ScoreBlockDev sb, double* w) {
  size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= (size_t)n_rows * P) return;
  const int i = (int)(slot % n_rows);
  const int pidx = sb.prob_fn[(size_t)src_value(sb.pa, slot) * sb.prob_nb + src_value(sb.pb, slot)];
  double acc = 0.0;
  for (int k = 0; k < sb.n_terms; ++k) {
    const ScoreTermDev& t = sb.t[k];
    const int val = src_value(t.val, slot);
    const int o = t.obs_col[i];
    double dens;
    if (o < 0)
      dens = val >= t.other_val ? -1000.0 : 0.0;  // the dummy, or a string drawn for one (ids after the dummy's)
    else if (t.pair[(size_t)o * t.n_lat + val] == 0)
      dens = sb.prob_same[pidx];
    else
      dens = sb.prob_diff[pidx] - sb.logn[t.nopt_fn[src_value(t.key, slot)]];
    acc += dens;
  }
  w[slot] += acc;
}

__global__ void add_weight_kernel(size_t n, const double* lse, double* w) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) w[t] += lse[t];
}

// compaction of NEW choices: pass 0 counts, pass 1 fills
// (block-aggregated: one global atomic per 256 elements instead of one per hit)
__global__ __launch_bounds__(256) void compact_new_kernel(size_t n, const int32_t* choice, int fill,
                                                          unsigned int* counter, int32_t* list, int32_t* pos_out) {
  __shared__ unsigned int wcnt[4];
  __shared__ unsigned int bbase;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool hit = t < n && choice[t] == PCLEAN_CHOICE_NEW;
  const unsigned long long mask = __ballot(hit);
  if (lane == 0) wcnt[wave] = (unsigned int)__popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    bbase = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  if (t >= n || !fill) return;
  if (!hit) {
    if (pos_out) pos_out[t] = -1;
    return;
  }
  unsigned int pos = bbase + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += wcnt[w];
  list[pos] = (int32_t)t;
  if (pos_out) pos_out[t] = (int32_t)pos;
}

// sub-list items from a parent list: list[j] indexes the parent's items
__global__ void sublist_items_kernel(int n, const int32_t* list, const int32_t* p_row, const int32_t* p_ctx,
                                     const int32_t* p_particle, const int32_t* p_origin, int32_t* row, int32_t* ctxv,
                                     int32_t* particle, int32_t* origin, const int32_t* p_ev_lo,
                                     const int32_t* p_ev_hi, const int32_t* p_rng, int32_t* ev_lo, int32_t* ev_hi,
                                     int32_t* rng) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int s = list[j];
  if (p_ev_lo) {
    ev_lo[j] = p_ev_lo[s];
    ev_hi[j] = p_ev_hi[s];
  }
  if (p_rng) rng[j] = p_rng[s];
  row[j] = p_row ? p_row[s] : s;
  particle[j] = p_particle[s];
  origin[j] = p_origin ? p_origin[s] : j;
  for (int c = 0; c < PCLEAN_MAX_CTX; ++c) ctxv[j * PCLEAN_MAX_CTX + c] = p_ctx ? p_ctx[s * PCLEAN_MAX_CTX + c] : 0;
}

// first-level new list from (row, particle) slots of a block (slot = particle * N + row)
__global__ void rootlist_items_kernel(int n, int N, size_t NP, const int32_t* list, const int32_t* b_ctx,
                                      const int32_t* cur_b, int32_t* row, int32_t* ctxv, int32_t* particle,
                                      int32_t* origin, int32_t* excl) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int slot = list[j];
  const int i = slot % N;
  row[j] = i;
  particle[j] = slot / N;
  origin[j] = j;
  excl[j] = cur_b ? cur_b[i] : -1;
  for (int c = 0; c < PCLEAN_MAX_CTX; ++c) ctxv[j * PCLEAN_MAX_CTX + c] = b_ctx ? b_ctx[(size_t)c * NP + slot] : 0;
}

__global__ void scatter_vals_kernel(int n, const int32_t* origin, const int32_t* draws, int n_nodes, int node,
                                    int32_t* vals) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) vals[(size_t)origin[j] * n_nodes + node] = draws[j];
}
__global__ void set_col_kernel(int n, int n_nodes, int node, int32_t v, int32_t* vals) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) vals[(size_t)j * n_nodes + node] = v;
}
// item attributes of a sub-list (list[j] = index in the parent item list)
__global__ void gather_items_kernel(int n, const int32_t* list, const int32_t* row, const int32_t* ctxv,
                                    const int32_t* excl, const int32_t* particle, int32_t* row2, int32_t* ctx2,
                                    int32_t* excl2, int32_t* part2) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int s = list[j];
  row2[j] = row ? row[s] : s;
  excl2[j] = excl ? excl[s] : -1;
  part2[j] = particle ? particle[s] : 0;
  for (int c = 0; c < PCLEAN_MAX_CTX; ++c) ctx2[j * PCLEAN_MAX_CTX + c] = ctxv ? ctxv[(size_t)s * PCLEAN_MAX_CTX + c] : 0;
}
__global__ void gather_i32_kernel(int n, const int32_t* list, const int32_t* src, int32_t* dst) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = src[list[j]];
}

// ---- particle kernels ----------------------------------------------------------
// One thread per row; the P particle weights of row i live at logw[i * sr + p * sp] (sweep: particle-major,
// sr = 1, sp = N -> coalesced; parity entry points: row-major, sr = P, sp = 1).  PMAX (compile-time bound
// of P) keeps the fixed-point weights in registers.
#define MAXP 64

template <int PMAX>
struct FixW {
  double m;
  uint64_t U;
  uint64_t u[PMAX];
};
template <int PMAX>
__device__ __forceinline__ void fix_weights(const double* w, size_t sp, int P, FixW<PMAX>& f) {
  f.m = -__builtin_inf();
  f.U = 0;
#pragma unroll
  for (int p = 0; p < PMAX; ++p)
    if (p < P) f.m = fmax(f.m, w[(size_t)p * sp]);
#pragma unroll
  for (int p = 0; p < PMAX; ++p) {
    f.u[p] = (p < P && f.m != -__builtin_inf()) ? pclean_fixw(w[(size_t)p * sp] - f.m) : 0ull;
    f.U += f.u[p];
  }
}
template <int PMAX>
__device__ __forceinline__ int fix_pick(const FixW<PMAX>& f, int P, uint64_t R) {
  if (f.U == 0) return P - 1;
  const uint64_t x = pclean_mulhi64(R, f.U);
  uint64_t acc = 0;
  int res = P - 1;
  bool found = false;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) {
    acc += f.u[p];
    if (!found && p < P && acc > x) {
      res = p;
      found = true;
    }
  }
  return res;
}

// row_inference.jl:87-105
template <int PMAX>
__global__ void maybe_resample_kernel(int n_rows, int P, const double* logw, size_t sr, size_t sp, int retain_first,
                                      const int32_t* csmc_flag, uint64_t seed, uint32_t sweep, uint32_t block,
                                      int64_t row_offset, int32_t* ancestors, double* logml_inc, double* ess_out,
                                      int32_t* did) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  FixW<PMAX> f;
  fix_weights<PMAX>(logw + (size_t)i * sr, sp, P, f);
  const double Ud = (double)f.U;
  double s2 = 0.0;
#pragma unroll
  for (int p = 0; p < PMAX; ++p)
    if (p < P) s2 += (double)f.u[p] * (double)f.u[p];
  const double ess = f.U ? (Ud * Ud) / s2 : 0.0;
  if (ess_out) ess_out[i] = ess;
  const bool retain = csmc_flag ? (csmc_flag[i] >= 0) : (retain_first != 0);
  int32_t* anc = ancestors + (size_t)i * sr;
  if (ess < (double)P / 2.0) {
    const uint32_t rr = (uint32_t)((int64_t)i + row_offset);
    for (int p = 0; p < P; ++p) {
      if (p == 0 && retain)
        anc[0] = 0;
      else
        anc[(size_t)p * sp] = fix_pick<PMAX>(f, P, pclean_rand64(seed, rr, PCLEAN_SITE_RESAMPLE(block), (uint32_t)p, sweep));
    }
    logml_inc[i] = pclean_lse_from_fix(f.m, f.U) - pclean_log((double)P);
    if (did) did[i] = 1;
  } else {
    for (int p = 0; p < P; ++p) anc[(size_t)p * sp] = p;
    logml_inc[i] = 0.0;
    if (did) did[i] = 0;
  }
}

// apply ancestors: particle-major int32 arrays and weights (clone_with_zero_weight, 17-21); the log-ML
// increment of the resampling step is accumulated here as well
struct AncestorArrays {  // the particle-major arrays a resampling step permutes, by value (no upload, no synchronisation)
  int32_t* p[2 * PCLEAN_MAX_BLOCKS];
};
__global__ void apply_ancestors_kernel(int n_rows, int P, const int32_t* ancestors, int n_arrays, AncestorArrays arrays,
                                       double* w, const int32_t* did, const double* logml_inc, double* logml_acc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  logml_acc[i] += logml_inc[i];
  if (!did[i]) return;
  int32_t tmp[MAXP];
  for (int a = 0; a < n_arrays; ++a) {
    int32_t* arr = arrays.p[a] + i;
    for (int p = 0; p < P; ++p) tmp[p] = arr[(size_t)ancestors[(size_t)p * n_rows + i] * n_rows];
    for (int p = 0; p < P; ++p) arr[(size_t)p * n_rows] = tmp[p];
  }
  for (int p = 0; p < P; ++p) w[(size_t)p * n_rows + i] = 0.0;
}

// row_inference.jl:158-165 + return value 186 (logml = accumulated resampling increments + log mean weight)
template <int PMAX>
__global__ void final_choice_kernel(int n_rows, int P, const double* logw, size_t sr, size_t sp, int use_mh,
                                    int is_csmc, const int32_t* csmc_flag, uint64_t seed, uint32_t sweep,
                                    int64_t row_offset, int32_t* chosen, double* log_total, const double* logml_acc,
                                    double* logml) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  FixW<PMAX> f;
  fix_weights<PMAX>(logw + (size_t)i * sr, sp, P, f);
  const uint32_t rr = (uint32_t)((int64_t)i + row_offset);
  const bool csmc = csmc_flag ? (csmc_flag[i] >= 0) : (is_csmc != 0);
  int c;
  if (use_mh && csmc && P >= 2) {
    const double Ud = (double)f.U;
    const double w0 = (double)f.u[0] / Ud, w1 = (double)f.u[PMAX > 1 ? 1 : 0] / Ud;
    double ratio = w1 / (1e-10 + w0);
    if (ratio > 1.0) ratio = 1.0;
    const double x = pclean_u01(pclean_rand64(seed, rr, PCLEAN_SITE_MH, 0u, sweep));
    c = (f.U != 0 && x < ratio) ? 1 : 0;
  } else {
    c = fix_pick<PMAX>(f, P, pclean_rand64(seed, rr, PCLEAN_SITE_FINAL, 0u, sweep));
  }
  chosen[i] = c;
  const double lt = pclean_lse_from_fix(f.m, f.U);
  if (log_total) log_total[i] = lt;
  if (logml) logml[i] = (logml_acc ? logml_acc[i] : 0.0) + lt - pclean_log((double)P);
}

#define DISPATCH_PMAX(P, ...)   \
  do {                          \
    if ((P) <= 2) {             \
      constexpr int PMAX = 2;   \
      __VA_ARGS__;              \
    } else if ((P) <= 8) {      \
      constexpr int PMAX = 8;   \
      __VA_ARGS__;              \
    } else if ((P) <= 32) {     \
      constexpr int PMAX = 32;  \
      __VA_ARGS__;              \
    } else {                    \
      constexpr int PMAX = 64;  \
      __VA_ARGS__;              \
    }                           \
  } while (0)

// per block after the final choice: the chosen particle's referent, its new-row record, the delta reference
// counts (the all-reduce payload) and the flags of moved rows / rows with a new referent.  Tables with few rows
// (hist_rows > 0: a handful of very popular referents, e.g. 28 measures for 1M records) accumulate the deltas in an
// LDS histogram per workgroup first — thousands of global atomics on the same few addresses would serialise.
__global__ __launch_bounds__(256) void finalize_block_kernel(int n_rows, const int32_t* chosen, const int32_t* pchoice,
                                                             const int32_t* pnewpos, const int32_t* cur_b,
                                                             int32_t* choice, int32_t* chosen_newpos,
                                                             unsigned long long* stats, int hist_rows,
                                                             int32_t* moved_flag, int32_t* new_flag) {
  extern __shared__ int32_t hist[];
  for (int k = threadIdx.x; k < hist_rows; k += 256) hist[k] = 0;
  if (hist_rows) __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rows) {
    const size_t s = (size_t)chosen[i] * n_rows + i;
    const int c = pchoice[s];
    const int o = cur_b[i];
    choice[i] = c;
    const int np = c == PCLEAN_CHOICE_NEW ? pnewpos[s] : -1;
    chosen_newpos[i] = np;
    new_flag[i] = np >= 0 ? 1 : 0;
    moved_flag[i] = c != o ? 1 : 0;
    if (o != c) {
      if (hist_rows) {
        if (o >= 0) atomicAdd(&hist[o], -1);
        if (c >= 0) atomicAdd(&hist[c], 1);
      } else {
        if (o >= 0) atomicAdd(&stats[o], (unsigned long long)(-1ll));
        if (c >= 0) atomicAdd(&stats[c], 1ull);
      }
    }
  }
  if (hist_rows) {
    __syncthreads();
    for (int k = threadIdx.x; k < hist_rows; k += 256) {
      const int v = hist[k];
      if (v) atomicAdd(&stats[k], (unsigned long long)(long long)v);
    }
  }
}

// own enumerated choices (locals) of the chosen particle, drawn from their conditional given
// the chosen referent: the inner draws of the nested enumeration (proposal_compiler.jl:115-127)
__global__ void locals_tail_kernel(int n_rows, int P, GaussDev g, PlanDev plan, const int32_t* chosen,
                                   const int32_t* pchoice, const int32_t* pnewpos, const int32_t* vals, int n_nodes,
                                   uint64_t seed, uint32_t sweep, uint32_t block, int64_t row_offset, int32_t* locals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const int c = chosen[i];
  const size_t slot = (size_t)c * n_rows + i;
  const int choice = pchoice[slot];
  const int32_t* v = choice >= 0 ? nullptr : vals + (size_t)pnewpos[slot] * n_nodes;
  locals[2 * i] = locals[2 * i + 1] = -1;
  const double xv = g.x[i];
  double sc[16];
  int codes[16];
  int n = 0;
  if (xv == xv)
    n = gauss_combo_scores(
        g, i, nullptr,
        [&](int d) -> int {
          if (g.src_kind[d] == PCLEAN_GSRC_CAND)
            return choice >= 0 ? g.src_ptr[d][choice] : resolve_new_value(plan, 0, g.src_slot[d], v);
          return g.src_ptr[d][i];  // PCLEAN_GSRC_OBS
        },
        sc, codes);
  else {  // no numeric evidence: the locals follow their (uniform) priors, observed ones stay fixed
    for (int l0 = 0; l0 < g.local_n[0]; ++l0)
      for (int l1 = 0; l1 < g.local_n[1]; ++l1) {
        const bool ok0 = !g.local_obs[0] || g.local_obs[0][i] < 0 || g.local_obs[0][i] == l0;
        const bool ok1 = !g.local_obs[1] || g.local_obs[1][i] < 0 || g.local_obs[1][i] == l1;
        if (ok0 && ok1) {
          sc[n] = 0.0;
          codes[n] = l0 * 16 + l1;
          ++n;
        }
      }
  }
  double m = -__builtin_inf();
  for (int k = 0; k < n; ++k) m = fmax(m, sc[k]);
  uint64_t u[16], U = 0;
  for (int k = 0; k < n; ++k) {
    u[k] = m == -__builtin_inf() ? 0ull : pclean_fixw(sc[k] - m);
    U += u[k];
  }
  int pick = n - 1;
  if (U) {
    const uint64_t x = pclean_mulhi64(
        pclean_rand64(seed, (uint32_t)((int64_t)i + row_offset), PCLEAN_SITE_LOCALS(block), (uint32_t)c, sweep), U);
    uint64_t acc = 0;
    for (int k = 0; k < n; ++k) {
      acc += u[k];
      if (acc > x) {
        pick = k;
        break;
      }
    }
  }
  if (n > 0) {
    locals[2 * i] = codes[pick] >> 4;
    locals[2 * i + 1] = g.n_locals > 1 ? (codes[pick] & 15) : -1;
  }
}

__global__ void gather_new_rows_kernel(int n, const int32_t* list, const int32_t* chosen_newpos, const int32_t* vals,
                                       int n_nodes, const int32_t* chosen, int32_t* rows_out, int32_t* vals_out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int i = list[j];
  rows_out[j] = i;
  const int32_t* v = vals + (size_t)chosen_newpos[i] * n_nodes;
  for (int k = 0; k < n_nodes; ++k) vals_out[(size_t)j * n_nodes + k] = v[k];
  vals_out[(size_t)j * n_nodes] = -1 - chosen[i];  // the chosen particle names the draw stream of its dummy values
}
// ---- weight of a particle whose new row chose a ProposalDummyValue (block_proposal.jl:58-60) -------------------------
// The enumeration scored the dummy option with its prior mass and the PLACEHOLDER's likelihood (q_disc holds that);
// propose_non_enumerable! then replaces the placeholder by random(node.dist, ...) and scores the observations below the
// node on the drawn string, so  p - q_disc = block marginal + sum over chosen dummies of
//     - log(dummy mass) + sum over the node's plain AddTypos observations [ logdensity(obs | drawn) - logdensity(obs | placeholder) ]
// (oracle/sweep.h: dummy_correction restates it; fp64 operation order: per leaf c = -logp, c += (l_drawn - l_placeholder)
// per term, corr += c in node order; w += corr after the block's log marginal).  One thread per NEW slot.
#define DUMMY_MAX_LEAVES 8
#define DUMMY_MAX_TERMS 2
#define DUMMY_DP_ARENA ((size_t)128 << 20)  // int16 cells of distance matrices per launch (256 MB)
struct DummyTermDev {
  const int32_t* obs_col;
  const uint8_t* pair;
  const uint16_t* lat_len;
  const int32_t* obs_ids;
  int32_t n_lat, elem_bytes, max_typos, dist_mode;
  int64_t dp_off;  // this term's distance matrix inside a slot's region of the arena (int16 cells)
};
struct DummyLeafDev {
  int32_t node, dummy_val, kind, min_len, max_len, n_terms;
  const int32_t* opt_vals;
  const double* opt_logp;
  DummyTermDev t[DUMMY_MAX_TERMS];
};
struct DummyPackDev {
  int32_t n_leaves, site_block;
  DummyLeafDev leaf[DUMMY_MAX_LEAVES];
  const uint16_t* sym;
  const int64_t* off;
  const double* lm_init;
  const double* lm_trans;
  const uint16_t* letter_sym;
  const double* nb;
  const double* logl;
  int32_t nb_stride, pad;
  int16_t* dp;             // arena: one region of slot_cells cells per NEW slot of the launch
  int64_t slot_cells;
  unsigned int* dp_ctr;    // [1] set when an observed string is longer than DUMMY_MAX_LEN
};
__global__ void dummy_correction_kernel(int j0, int n_new, int N, const int32_t* __restrict__ new_slots,
                                        const int32_t* __restrict__ vals, int n_nodes, DummyPackDev dp, uint64_t seed,
                                        uint32_t sweep, int64_t row_offset, double* __restrict__ w) {
  const int jl = blockIdx.x * blockDim.x + threadIdx.x;  // slot of this launch's slice [j0, j0 + n_new)
  if (jl >= n_new) return;
  const int j = j0 + jl;
  const int slot = new_slots[j];
  const int row = slot % N, particle = slot / N;
  const int32_t* v = vals + (size_t)j * n_nodes;
  double corr = 0.0;
  bool any = false;
  for (int li = 0; li < dp.n_leaves; ++li) {
    const DummyLeafDev& lf = dp.leaf[li];
    const int k = v[lf.node];
    if (k < 0 || lf.opt_vals[k] != lf.dummy_val) continue;
    any = true;
    double c = -lf.opt_logp[k];
    if (lf.kind == PCLEAN_DUMMY_STRING_PRIOR) {
      uint16_t drawn[DUMMY_MAX_LEN + 1];
      int L = -1;
      for (int ti = 0; ti < lf.n_terms; ++ti) {
        const DummyTermDev& tm = lf.t[ti];
        const int o = tm.obs_col[row];
        if (o < 0) continue;
        if (L < 0) {
          const uint64_t key = pclean_dummy_seed(seed, PCLEAN_SITE_NODE(dp.site_block, lf.node), (uint32_t)particle, sweep);
          L = dummy_draw_string(key, (uint32_t)((int64_t)row + row_offset), lf.min_len, lf.max_len, dp.lm_init, dp.lm_trans,
                                dp.letter_sym, drawn);
        }
        const int sid = tm.obs_ids[o];
        const uint16_t* os = dp.sym + dp.off[sid];
        const int ol = (int)(dp.off[sid + 1] - dp.off[sid]);
        if (ol > DUMMY_MAX_LEN) {
          dp.dp_ctr[1] = 1u;
          continue;
        }
        int16_t* H = dp.dp + (size_t)jl * dp.slot_cells + tm.dp_off;
        const int d = dummy_distance(tm.dist_mode, os, ol, drawn, L, H);
        double l;
        if (tm.max_typos >= 0 && d > tm.max_typos) {
          l = -1e5;
        } else {
          l = dp.nb[(size_t)((L + 4) / 5) * dp.nb_stride + d];
          l -= dp.logl[L] * (double)d;
          l -= 1.629048269010741 * (double)d;
        }
        const size_t pi = (size_t)o * tm.n_lat + lf.dummy_val;
        const int dph = tm.elem_bytes == 1 ? (int)tm.pair[pi] : (int)((const uint16_t*)tm.pair)[pi];
        double lph;
        if (tm.max_typos >= 0 && dph > tm.max_typos) {
          lph = -1e5;
        } else {
          const int Lp = tm.lat_len[lf.dummy_val];
          lph = dp.nb[(size_t)((Lp + 4) / 5) * dp.nb_stride + dph];
          lph -= dp.logl[Lp] * (double)dph;
          lph -= 1.629048269010741 * (double)dph;
        }
        c += l - lph;
      }
    }
    corr += c;
  }
  if (any) w[slot] += corr;
}

// ---------------------------------------------------------------------------
// host side
static inline dim3 grid1(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

struct ItemList {  // device arrays describing enumeration work items
  int n = 0;
  const int32_t* row = nullptr;
  const int32_t* ctx = nullptr;
  const int32_t* particle = nullptr;
  const int32_t* origin = nullptr;
  // evidence sets (latent-class sweeps): per-item [ev_lo, ev_hi) into ev_rows / ev_ctx; RNG row ids
  const int32_t* ev_lo = nullptr;
  const int32_t* ev_hi = nullptr;
  const int32_t* ev_rows = nullptr;
  const int32_t* ev_ctx = nullptr;
  const int32_t* rng_row = nullptr;
  int draw_is = 0, draw_ds = 0;  // ItemsDev::draw_is / draw_ds of the draws this list produces
};

void pclean_sweep_state_free(pclean_ctx* ctx) {
  if (!ctx->sweep_state) return;
  SweepState* s = (SweepState*)ctx->sweep_state;
  for (auto& b : s->pool) b.release();
  for (auto& r : s->run) {
    r.pchoice.release(); r.pnewpos.release(); r.draws.release(); r.moved_flag.release(); r.new_flag.release();
    r.moved_list.release(); r.new_list.release(); r.new_slots.release();
    r.locals.release(); r.it_ctx.release(); r.choice.release(); r.chosen_newpos.release(); r.vals.release();
    r.lse.release(); r.plan_kind.release(); r.plan_nrows.release(); r.plan_cmb.release(); r.plan_colmap.release();
    r.plan_cols.release();
  }
  s->did.release(); s->cur.release(); s->chosen.release(); s->ancestors.release(); s->csmc_flag.release(); s->w.release();
  s->log_total.release(); s->logml_inc.release(); s->logml_acc.release(); s->logml.release(); s->counter.release();
  s->arr_ptrs.release();
  for (auto& kv : s->leaf_iota) kv.second.release();
  for (auto& f : s->fast) {
    for (auto& c : f.comp) c.release();
    for (auto& c : f.clen) c.release();
    for (auto& c : f.cblk) c.release();
    f.prior_e.release();
    f.prior_n.release();
    f.alive.release();
    f.zero_row.release();
  }
  s->tail_counts.release();
  for (auto& kv : s->tuple_ids) {
    kv.second.id.release();
    kv.second.pre.release();
  }
  for (auto& kv : s->memo) {
    kv.second.keys.release();
    kv.second.vals.release();
    kv.second.count.release();
  }
  if (s->h_counts) (void)hipHostFree(s->h_counts);
  if (s->h_over) (void)hipHostFree(s->h_over);
  for (int k = 0; k < SweepState::MAX_SIDE; ++k) {
    if (s->side[k]) (void)hipStreamDestroy(s->side[k]);
    if (s->side_join[k]) (void)hipEventDestroy(s->side_join[k]);
    if (s->side_mid[k]) (void)hipEventDestroy(s->side_mid[k]);
  }
  if (s->side_fork) (void)hipEventDestroy(s->side_fork);
  if (s->h_poll) (void)hipHostFree((void*)s->h_poll);
  s->dummy_dp.release();
  s->dummy_ctr.release();
  s->over_ctr.release();
  for (auto e : s->prof_ev) (void)hipEventDestroy(e);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->evs) (void)hipEventDestroy(s->evs);
  if (s->eve) (void)hipEventDestroy(s->eve);
  delete s;
  ctx->sweep_state = nullptr;
}

// Start of an entry point that evaluates plan nodes: scratch pool rewound, overflow counters cleared.
static int begin_call(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  s->pool_used = 0;
  s->dbg_desc = nullptr;
  s->dummy_used = false;
  ctx->prior_mode = false;
  s->over_rec.clear();
  if (s->over_ctr.alloc(OVER_SLOTS + STAT_WORDS)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemsetAsync(s->over_ctr.p, 0, (OVER_SLOTS + STAT_WORDS) * sizeof(unsigned int), ctx->stream));
  s->scan_stats_used = false;
  return PCLEAN_OK;
}
// End of such a call, after its last stream synchronisation has been queued: the overflow counts of the sync-free
// launches go into the statistics and the "does the pre-filter pay for this option list" heuristic.
static int queue_over_copy(pclean_ctx* ctx) {  // before a stream synchronisation of the caller
  SweepState* s = st(ctx);
  if (s->over_rec.empty() && !s->scan_stats_used) return PCLEAN_OK;
  if (!s->h_over) HIPCHK(ctx, hipHostMalloc((void**)&s->h_over, (OVER_SLOTS + STAT_WORDS) * sizeof(unsigned int), hipHostMallocDefault));
  if (!s->over_rec.empty())
    HIPCHK(ctx, hipMemcpyAsync(s->h_over, s->over_ctr.p, s->over_rec.size() * sizeof(unsigned int), hipMemcpyDeviceToHost,
                               ctx->stream));
  if (s->scan_stats_used)
    HIPCHK(ctx, hipMemcpyAsync(s->h_over + OVER_SLOTS, s->over_ctr.p + OVER_SLOTS, STAT_WORDS * sizeof(unsigned int),
                               hipMemcpyDeviceToHost, ctx->stream));
  return PCLEAN_OK;
}
static void apply_over_stats(pclean_ctx* ctx) {  // after that synchronisation
  SweepState* s = st(ctx);
  for (size_t i = 0; i < s->over_rec.size(); ++i) {
    const SweepState::OverRec& r = s->over_rec[i];
    const unsigned int h = s->h_over[i];
    ctx->timing.reserved += (int32_t)h;
    if (r.time_it) ctx->root_stats.overflow_items = (int32_t)h;
    // (evidence sets, min_items 64: the re-run is the generic kernel over every candidate and the scan is the cheap part
    // — measured at 1M rows, full iteration: 459 ms with the quarter rule, 412 ms at three quarters, 369 ms without — so
    // the pre-filter is only given up for a list when EVERY item comes back)
    const bool ev = r.min_items < 1024;
    if (r.leaf && r.n_items >= r.min_items && (ev ? h >= (unsigned int)r.n_items : (size_t)h * 4 > (size_t)r.n_items)) {
      FastRoot& f = s->fast[r.block * 64 + r.node];
      f.disabled = f.backoff;
      f.backoff = std::min(f.backoff * 2, 1 << 20);
    }
    if (h && getenv("PCLEAN_DEBUG_OVERFLOW"))
      fprintf(stderr, "[pclean] block %d node %d: %u of %d items re-run over all candidates\n", r.block, r.node, h, r.n_items);
  }
  s->over_rec.clear();
  if (s->scan_stats_used) {
    unsigned long long tot[4] = {0, 0, 0, 0};  // 64 slots a cache line apart (root_wave.hip: WAVE_STAT_SLOTS)
    for (int sl = 0; sl < 64; ++sl)
      for (int i = 0; i < 4; ++i) tot[i] += s->h_over[OVER_SLOTS + sl * 32 + i];
    ctx->root_stats.full_scans = (int32_t)tot[0];
    ctx->root_stats.fine_blocks = (int32_t)tot[1];
    ctx->root_stats.scored_terms = (int32_t)tot[2];
    ctx->root_stats.resolved_groups = (int32_t)tot[3];
    s->scan_stats_used = false;
  }
}
static int finish_call(pclean_ctx* ctx) {
  if (st(ctx)->over_rec.empty() && !st(ctx)->scan_stats_used) return PCLEAN_OK;
  int rc = queue_over_copy(ctx);
  if (rc) return rc;
  PCLEAN_SYNC(ctx);
  apply_over_stats(ctx);
  return PCLEAN_OK;
}

// bump-style scratch: buffers persist across sweeps, handed out in order
template <typename T>
static T* scratch(pclean_ctx* ctx, size_t count) {
  SweepState* s = st(ctx);
  if (s->pool_used == s->pool.size()) s->pool.emplace_back();
  DevBuf<unsigned char>& b = s->pool[s->pool_used++];
  // a slot that has to grow takes half as much again: the sizes of a sweep's lists (groups, items that need a step)
  // drift from sweep to sweep, and a hipFree + hipMalloc in the middle of a sweep stalls the stream for ~0.3 ms
  const size_t need = std::max<size_t>(count * sizeof(T), 16);
  if (need > b.n && b.alloc(std::max(need, b.n + b.n / 2))) {
    if (b.alloc(need)) return nullptr;
  }
  return (T*)b.p;
}

// ---- per-phase profile (pclean_set_profiling): HIP events on the library's stream around groups of launches
static int prof_phase_id(SweepState* s, const char* name) {
  for (size_t i = 0; i < s->prof_names.size(); ++i)
    if (s->prof_names[i] == name) return (int)i;
  s->prof_names.push_back(name);
  s->prof_ms.push_back(0.f);
  s->prof_launches.push_back(0);
  return (int)s->prof_names.size() - 1;
}
struct ProfScope {  // records start at construction, stop at destruction
  pclean_ctx* ctx;
  SweepState* s;
  size_t rec = (size_t)-1;
  ProfScope(pclean_ctx* c, const char* name) : ctx(c), s(st(c)) {
    if (!s->prof_on) return;
    rec = s->prof_used++;
    while (s->prof_ev.size() < 2 * (rec + 1)) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) {
        rec = (size_t)-1;
        --s->prof_used;
        return;
      }
      s->prof_ev.push_back(e);
    }
    if (s->prof_phase.size() <= rec) s->prof_phase.resize(rec + 1);
    s->prof_phase[rec] = prof_phase_id(s, name);
    (void)hipEventRecord(s->prof_ev[2 * rec], ctx->stream);
  }
  ~ProfScope() {
    if (rec != (size_t)-1) (void)hipEventRecord(s->prof_ev[2 * rec + 1], ctx->stream);
  }
};
static void prof_collect(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  (void)hipStreamSynchronize(ctx->stream);  // the last scope's stop event has only just been recorded
  for (size_t r = 0; r < s->prof_used; ++r) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->prof_ev[2 * r], s->prof_ev[2 * r + 1]) == hipSuccess) {
      s->prof_ms[s->prof_phase[r]] += ms;
      s->prof_launches[s->prof_phase[r]] += 1;
    }
  }
  s->prof_used = 0;
}

extern "C" int pclean_set_profiling(pclean_ctx* ctx, int32_t on) {
  if (!ctx) return PCLEAN_ERR_ARG;
  SweepState* s = st(ctx);
  s->prof_on = on != 0;
  s->prof_used = 0;
  std::fill(s->prof_ms.begin(), s->prof_ms.end(), 0.f);
  std::fill(s->prof_launches.begin(), s->prof_launches.end(), 0);
  return PCLEAN_OK;
}
extern "C" int pclean_get_profile(pclean_ctx* ctx, int32_t cap, char* names, float* ms, int32_t* launches,
                                  int32_t* n_out) {
  if (!ctx || !n_out || cap < 0) return PCLEAN_ERR_ARG;
  SweepState* s = st(ctx);
  *n_out = (int32_t)s->prof_names.size();
  for (int i = 0; i < cap && i < *n_out; ++i) {
    if (names) {
      strncpy(names + (size_t)i * 32, s->prof_names[i].c_str(), 31);
      names[(size_t)i * 32 + 31] = 0;
    }
    if (ms) ms[i] = s->prof_ms[i];
    if (launches) launches[i] = s->prof_launches[i];
  }
  return PCLEAN_OK;
}

static int build_gauss_dev(pclean_ctx* ctx, const pclean_gauss& g, const CandTable* t, GaussDev& d) {
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
  }
  d.sigma = g.sigma;
  d.log_sigma = std::log(g.sigma);
  return PCLEAN_OK;
}

static int build_node_dev(pclean_ctx* ctx, const Block& b, int node_id, NodeDev& nd) {
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
static int eval_node(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                     uint64_t seed, uint32_t sweep, int n_draws, double* lse_out, int32_t* draws_out,
                     double* scores_out, const double* snew_override, bool time_it);
// Per-unique-observed-value marginal of a cacheable leaf (one term, no ctx):
// cache[u] for u < n_obs, cache[n_obs] for a missing observation.
static int ensure_leaf_cache(pclean_ctx* ctx, int block_id, int node_id, const double** out, const int32_t** obs_col,
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
  if ((int)f.comp.size() != n.n_terms || f.kpad != kpad) {
    for (auto& c : f.comp) c.release();
    for (auto& c : f.clen) c.release();
    for (auto& c : f.cblk) c.release();
    f.cblk.assign(n.n_terms, DevBuf<uint8_t>());
    f.comp.assign(n.n_terms, DevBuf<uint8_t>());
    f.clen.assign(n.n_terms, DevBuf<uint8_t>());
    f.ver.assign(n.n_terms, 0);
    f.kpad = kpad;
    f.prior_ver = 0;
  }
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
      int rc = pclean_update_compact(ctx, pt.d.p, pt.n_obs, pt.n_lat, t.cols.p + (size_t)tm.cand_col * t.n_rows, pt.lat_len.p,
                                     t.cols_delta_rows, t.cols_delta_n, kpad, f.comp[i].p, f.clen[i].p);
      if (rc) return rc;
      rc = pclean_build_compact_min(ctx, f.comp[i].p, pt.n_obs, kpad, cstride, f.cblk[i].p);
      if (rc) return rc;
      f.ver[i] = ver;
    }
    if (f.ver[i] != ver || !f.comp[i].p) {
      ProfScope psd(ctx, "compact_table_rebuild");
      if (f.comp[i].alloc(std::max<size_t>((size_t)pt.n_obs * kpad, 16)) || f.clen[i].alloc(kpad))
        return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed (compact tables)");
      int rc = pclean_build_compact(ctx, pt.d.p, pt.n_obs, pt.n_lat, t.cols.p + (size_t)tm.cand_col * t.n_rows,
                                    pt.lat_len.p, t.n_rows, kpad, f.comp[i].p, f.clen[i].p);
      if (rc) return rc;
      if (f.cblk[i].alloc(std::max<size_t>((size_t)pt.n_obs * cstride, 16)))
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
  fr.n_cand = t.n_rows;
  fr.kpad = kpad;
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
struct ItemGroups {
  int n_groups = 0;              // 0: grouping not applicable / not worth it
  const int32_t* grp_off = nullptr;  // [n_groups + 1] into members
  const int32_t* members = nullptr;  // [n] item ids, groups contiguous
  const int32_t* head = nullptr;     // [n] 1 at the first member of each group (sorted order)
  const int32_t* uid = nullptr;      // [n] inclusive scan of head
};
static int make_item_groups(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                            ItemGroups& g, int split_m = 0);
// ---- aggregated evidence of latent-class sweeps -------------------------------------------------------------
// (contract in enum_kernels.hip: candidate_score_ev)
__global__ void item_of_pos_kernel(int n_ev, int n_items, const int32_t* __restrict__ off, int32_t* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ev) return;
  int lo = 0, hi = n_items - 1;  // largest t with off[t] <= e
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= e)
      lo = mid;
    else
      hi = mid - 1;
  }
  out[e] = lo;
}
__global__ void agg_key_kernel(int n_ev, const int32_t* __restrict__ item_of_pos, const int32_t* __restrict__ ev_rows,
                               const int32_t* __restrict__ ev_ctx, int ctx_slot, const int32_t* __restrict__ obs_col,
                               uint64_t* __restrict__ key) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ev) return;
  const uint64_t o1 = (uint64_t)(uint32_t)(obs_col[ev_rows[e]] + 1) & 0xffffffull;
  const uint64_t c = ctx_slot >= 0 ? ((uint64_t)(uint32_t)ev_ctx[(size_t)e * PCLEAN_MAX_CTX + ctx_slot] & 0xffffull) : 0ull;
  key[e] = ((uint64_t)(uint32_t)item_of_pos[e] << 40) | (c << 24) | o1;
}
__global__ void agg_off_kernel(int n_items, const uint64_t* __restrict__ uniq, const int32_t* __restrict__ n_runs,
                               int32_t* __restrict__ off) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_items) return;
  const uint64_t want = (uint64_t)(uint32_t)t << 40;
  int lo = 0, hi = *n_runs;  // first run with key >= want
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (uniq[mid] < want)
      lo = mid + 1;
    else
      hi = mid;
  }
  off[t] = lo;
}
struct AggPack {
  AggDev a[PCLEAN_MAX_TERMS];
};
// The same aggregation with one workgroup per (original item, term) when no item has more than AGG_LDS_CAP evidence
// rows — the sub-batches of a large latent class (a few hundred rows with ~100 referring rows each): the keys of the
// item's rows are sorted in LDS (bitonic), run-length encoded and written at the item's own offset of the evidence
// list.  Same runs in the same order as the global sort + run-length encoding below (which costs ~18 launches per
// term); the scores that walk them are unchanged.
#define AGG_LDS_CAP 2048
struct AggTermArgs {
  const int32_t* obs_col[PCLEAN_MAX_TERMS];
  int32_t ctx_slot[PCLEAN_MAX_TERMS];
  uint64_t* uniq[PCLEAN_MAX_TERMS];
  int32_t* cnt[PCLEAN_MAX_TERMS];
  int32_t* end[PCLEAN_MAX_TERMS];
};
__global__ __launch_bounds__(256) void agg_item_kernel(int n_items, const int32_t* __restrict__ ev_off,
                                                       const int32_t* __restrict__ ev_rows, const int32_t* __restrict__ ev_ctx,
                                                       AggTermArgs a) {
  __shared__ uint64_t s_key[AGG_LDS_CAP];
  __shared__ int32_t s_run[AGG_LDS_CAP];  // run id of sorted position i, then the run lengths
  __shared__ int s_w[4];
  const int t = blockIdx.x, ti = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = ev_off[t], L = ev_off[t + 1] - lo;
  if (L <= 0) {
    if (tid == 0) a.end[ti][t] = lo;
    return;
  }
  int np2 = 1;
  while (np2 < L) np2 <<= 1;
  const int32_t* oc = a.obs_col[ti];
  const int cs = a.ctx_slot[ti];
  for (int i = tid; i < np2; i += 256) {
    uint64_t key = ~0ull;  // padding sorts last
    if (i < L) {
      const int e = lo + i;
      const uint64_t o1 = (uint64_t)(uint32_t)(oc[ev_rows[e]] + 1) & 0xffffffull;
      const uint64_t c = cs >= 0 ? ((uint64_t)(uint32_t)ev_ctx[(size_t)e * PCLEAN_MAX_CTX + cs] & 0xffffull) : 0ull;
      key = ((uint64_t)(uint32_t)t << 40) | (c << 24) | o1;
    }
    s_key[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = s_key[i], y = s_key[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            s_key[i] = y;
            s_key[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  // run ids: inclusive count of heads over the sorted keys, in chunks of 256 positions
  int base = 0;
  for (int i0 = 0; i0 < L; i0 += 256) {
    const int i = i0 + tid;
    const int head = (i < L && (i == 0 || s_key[i] != s_key[i - 1])) ? 1 : 0;
    int incl = head;
    for (int sh = 1; sh < 64; sh <<= 1) {
      const int x = __shfl_up(incl, sh, 64);
      if (lane >= sh) incl += x;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int before = base, total = 0;
    for (int w = 0; w < 4; ++w) {
      if (w < wave) before += s_w[w];
      total += s_w[w];
    }
    if (i < L) s_run[i] = before + incl - 1;
    base += total;
    __syncthreads();
  }
  const int n_runs = base;
  // heads write their key; lengths = distance to the next head
  for (int i = tid; i < L; i += 256) {
    if (i == 0 || s_key[i] != s_key[i - 1]) {
      const int r = s_run[i];
      int j = i + 1;
      while (j < L && s_key[j] == s_key[i]) ++j;
      a.uniq[ti][lo + r] = s_key[i];
      a.cnt[ti][lo + r] = j - i;
    }
  }
  if (tid == 0) a.end[ti][t] = lo + n_runs;
}
__global__ void write_agg_kernel(AggPack p, int n, AggDev* dst) {
  const int i = threadIdx.x;
  if (i < n) dst[i] = p.a[i];
}

// A per-evidence-row ctx value occupies 16 bits of an aggregation key (agg_key_kernel / agg_item_kernel): its domain
// (the ctx side of the term's fn table, or the error-probability table of a MaybeSwap term) must stay below 2^16 or runs
// of different values would alias.
static int agg_ctx_fits(pclean_ctx* ctx, const pclean_term& tm, int ctx_slot) {
  if (ctx_slot < 0) return PCLEAN_OK;
  int64_t dom = 0;
  if (tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP)
    dom = ctx->n_prob;
  else if (tm.fn_table >= 0 && tm.fn_table < PCLEAN_MAX_TABLES && ctx->fn[tm.fn_table].valid)
    dom = tm.ctx_mode == 2 ? ctx->fn[tm.fn_table].n_b : ctx->fn[tm.fn_table].n_a;
  if (dom >= (1 << 16))
    return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "evidence aggregation: a per-evidence-row ctx domain of %lld values does not "
                                                 "fit the 16 key bits", (long long)dom);
  return PCLEAN_OK;
}

// Aggregated evidence of every term of node `node_id` over the original items of the running
// pclean_sweep_latent call; built once per (call, node).
static int ensure_agg(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const AggDev** out) {
  SweepState* s = st(ctx);
  auto itc = s->lat_agg.find(node_id);
  if (itc != s->lat_agg.end()) {
    *out = itc->second;
    return PCLEAN_OK;
  }
  ProfScope ps(ctx, "evidence_aggregation");
  const Block& b = ctx->block[block_id];
  const pclean_node& n = b.nodes[node_id];
  const int n_ev = s->lat_ev, n_items = s->lat_items;
  if (n_items >= (1 << 24)) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "too many latent rows in one latent sweep");
  AggPack pack{};
  AggDev* dst = (AggDev*)scratch<unsigned char>(ctx, sizeof(AggDev) * PCLEAN_MAX_TERMS);
  if (!dst) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  if (n_ev > 0 && s->lat_max_ev <= AGG_LDS_CAP && n.n_terms > 0 && n.n_terms <= PCLEAN_MAX_TERMS && !ctx->no_item_agg) {
    AggTermArgs at{};
    for (int ti = 0; ti < n.n_terms; ++ti) {
      const pclean_term& tm = b.terms[n.term_begin + ti];
      const PairTable& pt = ctx->pair[tm.pair_table];
      if (tm.obs_col < 0 || tm.obs_col >= ctx->n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term column out of range");
      if (pt.valid && pt.n_obs + 1 >= (1 << 24)) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "observed domain too large for the evidence keys");
      const int ctx_slot = (tm.ctx_slot >= 0 && tm.ctx_mode != 0) ? tm.ctx_slot : -1;
      if (ctx_slot >= 0 && !il.ev_ctx) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term %d needs per-evidence-row ctx", n.term_begin + ti);
      { const int rck = agg_ctx_fits(ctx, tm, ctx_slot); if (rck) return rck; }  // (16 key bits per ctx value)
      at.obs_col[ti] = ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows;
      at.ctx_slot[ti] = ctx_slot;
      at.uniq[ti] = scratch<uint64_t>(ctx, (size_t)n_ev);
      at.cnt[ti] = scratch<int32_t>(ctx, (size_t)n_ev);
      at.end[ti] = scratch<int32_t>(ctx, (size_t)n_items);
      if (!at.uniq[ti] || !at.cnt[ti] || !at.end[ti]) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      pack.a[ti] = AggDev{at.uniq[ti], at.cnt[ti], s->lat_off, at.end[ti]};
    }
    hipLaunchKernelGGL(agg_item_kernel, dim3(n_items, n.n_terms), dim3(256), 0, ctx->stream, n_items, s->lat_off, il.ev_rows,
                       il.ev_ctx, at);
    hipLaunchKernelGGL(write_agg_kernel, dim3(1), dim3(64), 0, ctx->stream, pack, n.n_terms, dst);
    HIPCHK(ctx, hipGetLastError());
    s->lat_agg[node_id] = dst;
    *out = dst;
    return PCLEAN_OK;
  }
  for (int ti = 0; ti < n.n_terms; ++ti) {
    const pclean_term& tm = b.terms[n.term_begin + ti];
    const PairTable& pt = ctx->pair[tm.pair_table];
    if (tm.obs_col < 0 || tm.obs_col >= ctx->n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term column out of range");
    if (pt.valid && pt.n_obs + 1 >= (1 << 24)) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "observed domain too large for the evidence keys");
    const int ctx_slot = (tm.ctx_slot >= 0 && tm.ctx_mode != 0) ? tm.ctx_slot : -1;
    if (ctx_slot >= 0 && !il.ev_ctx) return pclean_fail(ctx, PCLEAN_ERR_ARG, "term %d needs per-evidence-row ctx", n.term_begin + ti);
      { const int rck = agg_ctx_fits(ctx, tm, ctx_slot); if (rck) return rck; }  // (16 key bits per ctx value)
    const size_t ne = (size_t)std::max(n_ev, 1);
    uint64_t* key = scratch<uint64_t>(ctx, ne);
    uint64_t* key_s = scratch<uint64_t>(ctx, ne);
    uint64_t* uniq = scratch<uint64_t>(ctx, ne);
    int32_t* cnt = scratch<int32_t>(ctx, ne);
    int32_t* n_runs = scratch<int32_t>(ctx, 4);
    int32_t* off = scratch<int32_t>(ctx, (size_t)n_items + 2);
    if (!key || !key_s || !uniq || !cnt || !n_runs || !off) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    HIPCHK(ctx, hipMemsetAsync(n_runs, 0, sizeof(int32_t), ctx->stream));
    if (n_ev > 0) {
      hipLaunchKernelGGL(agg_key_kernel, grid1(n_ev), dim3(256), 0, ctx->stream, n_ev, s->lat_item_of_pos, il.ev_rows,
                         il.ev_ctx, ctx_slot, ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows, key);
      size_t tmp_sort = 0, tmp_rle = 0;
      HIPCHK(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_sort, key, key_s, n_ev, 0, 64, ctx->stream));
      HIPCHK(ctx, hipcub::DeviceRunLengthEncode::Encode(nullptr, tmp_rle, key_s, uniq, cnt, n_runs, n_ev, ctx->stream));
      unsigned char* tmp = scratch<unsigned char>(ctx, std::max(tmp_sort, tmp_rle));
      if (!tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      HIPCHK(ctx, hipcub::DeviceRadixSort::SortKeys(tmp, tmp_sort, key, key_s, n_ev, 0, 64, ctx->stream));
      HIPCHK(ctx, hipcub::DeviceRunLengthEncode::Encode(tmp, tmp_rle, key_s, uniq, cnt, n_runs, n_ev, ctx->stream));
    }
    hipLaunchKernelGGL(agg_off_kernel, grid1((size_t)n_items + 1), dim3(256), 0, ctx->stream, n_items, uniq, n_runs, off);
    pack.a[ti] = AggDev{uniq, cnt, off, nullptr};
  }
  hipLaunchKernelGGL(write_agg_kernel, dim3(1), dim3(64), 0, ctx->stream, pack, n.n_terms, dst);
  HIPCHK(ctx, hipGetLastError());
  s->lat_agg[node_id] = dst;
  *out = dst;
  return PCLEAN_OK;
}

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

static int eval_node(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                     uint64_t seed, uint32_t sweep, int n_draws, double* lse_out, int32_t* draws_out,
                     double* scores_out, const double* snew_override, bool time_it) {
  Block& b = ctx->block[block_id];
  const pclean_node& n = b.nodes[node_id];
  SweepState* s = st(ctx);
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
        HIPCHK(ctx, hipMemsetAsync(s->counter.p + 2, 0, sizeof(unsigned int), ctx->stream));
        hipLaunchKernelGGL(compact_new_kernel, grid1(il.n), dim3(256), 0, ctx->stream, (size_t)il.n, flag, 1,
                           s->counter.p + 2, list, nullptr);
        PCLEAN_READ_COUNT(ctx, s->counter.p + 2, &n_need);
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
  FastRootDev fr;
  int fast = 0, fast_ev = 0;
  if (!scores_out && !snew_override && !ctx->force_generic && !nd.g.on && !ctx->prior_mode) {
    if (!il.ev_lo)
      fast = try_fast_root(ctx, block_id, node_id, fr);
    else if (n.kind == PCLEAN_NODE_LEAF && n_draws <= 1 && !getenv("PCLEAN_NO_FAST_EV"))
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
  {
    const int nc = nd.n_cand + (n.kind == PCLEAN_NODE_FK ? 1 : 0);
    const bool lds_kernel = (size_t)((nc + 1) & ~1) * 8 + (16 + 64) * 8 <= 160 * 1024;
    if (n_draws > 0 && !scores_out && !snew_override && !ctx->force_generic && !il.rng_row && !il.ev_lo &&
        (fast || lds_kernel)) {
      ItemGroups g;
      // wave kernel: at most ~2 x 256 draws per group (see item_head_kernel)
      rc = make_item_groups(ctx, block_id, node_id, il, excl, g, fast ? std::max(4, 256 / std::max(n_draws, 1)) : 0);
      if (rc) return rc;
      if (g.n_groups > 0) {
        it.n = g.n_groups;
        it.grp_off = g.grp_off;
        it.members = g.members;
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
    if (fast) {
      rs.kpad = fr.kpad;
      rs.cstride = fr.cstride;
      rs.n_pre = fr.n_pre;
      for (int p = 0; p < 3; ++p) rs.pre_obs_col[p] = p < fr.n_pre ? b.terms[n.term_begin + fr.pre[p]].obs_col : -1;
    }
  }
  if (!fast && !fast_ev) {
    ProfScope ps(ctx, n.kind == PCLEAN_NODE_FK ? "enum_fk_generic" : "enum_leaf_generic");
    if (time_it) (void)hipEventRecord(s->ev0, ctx->stream);
    rc = pclean_launch_enum(ctx, nd, it, ch, seed, sweep, site, n_draws, lse_out, scores_out, draws_out);
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
    HIPCHK(ctx, hipMemsetAsync(s->counter.p + 1, 0, sizeof(unsigned int), ctx->stream));
  }
  if (!ev_list_mode)  // (the list stands for the markers there)
    HIPCHK(ctx, hipMemsetAsync(oflag, 0, (size_t)il.n * sizeof(int32_t), ctx->stream));  // kernels only set overflow markers
  if (fast) {
    int32_t* desc = scratch<int32_t>(ctx, pclean_fast_desc_words(it.n));
    if (!desc) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    ProfScope ps(ctx, time_it ? "root_scan_block0" : (n.kind == PCLEAN_NODE_FK ? "slot_scan" : "option_scan"));
    if (time_it) (void)hipEventRecord(s->ev0, ctx->stream);
    unsigned int* scan_stats = nullptr;
    if (time_it && s->over_ctr.p) {  // the timed launch (block 0's root): what it read, for bench.py's byte model
      scan_stats = s->over_ctr.p + OVER_SLOTS;
      s->scan_stats_used = true;
    }
    rc = pclean_launch_root_fast(ctx, fr, it, ch, seed, sweep, site, n_draws, lse_out, draws_out, oflag, over_count, desc,
                                 over_list, scan_stats, il.n);
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
      rc = pclean_launch_ev_leaf(ctx, nd, it, fr, seed, sweep, site, n_draws, lse_out, draws_out, oflag, over_count, over_list);
    }
    if (!rc && ev_list_mode) {
      ProfScope ps2(ctx, "overflow_rerun");
      ItemsDev itr = it;
      itr.sel = over_list;
      itr.sel_n = over_count;
      return pclean_launch_enum(ctx, nd, itr, ch, seed, sweep, site, n_draws, lse_out, nullptr, draws_out);
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
    HIPCHK(ctx, hipMemsetAsync(s->counter.p + 1, 0, sizeof(unsigned int), ctx->stream));
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
    if (!done) rc = pclean_launch_enum(ctx, nd, it2, ch, seed, sweep, site, n_draws, lse_out, nullptr, draws_out);
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

// Groups the items of `il` by (observed values of the sub-tree of node_id, ctx, excl).  g.n_groups == 0
// when the sub-tree cannot be keyed, the list is small, or fewer than a quarter of the items are duplicates.
static int make_item_groups(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                            ItemGroups& g, int split_m) {
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
  HIPCHK(ctx, hipMemsetAsync(s->counter.p + 3, 0, sizeof(unsigned int), ctx->stream));
  hipLaunchKernelGGL(compact_new_kernel, grid1(N), dim3(256), 0, ctx->stream, (size_t)N, flag, 1, s->counter.p + 3, list,
                     nullptr);
  unsigned int n_miss = 0;
  PCLEAN_READ_COUNT(ctx, s->counter.p + 3, &n_miss);
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
  int rc0 = make_item_groups(ctx, block_id, node_id, il, excl, g);
  if (rc0) return rc0;
  if (g.n_groups == 0)
    return eval_node(ctx, block_id, node_id, il, excl, seed, sweep, 0, lse_out, nullptr, nullptr, nullptr, false);
  const int n = il.n;
  const int32_t n_unique = g.n_groups;
  const int32_t* idx_s = g.members;
  const int32_t* head = g.head;
  const int32_t* uid = g.uid;
  int32_t* uid_of_item = scratch<int32_t>(ctx, n);
  if (!uid_of_item) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  int32_t* row2 = scratch<int32_t>(ctx, n_unique);
  int32_t* ctx2 = scratch<int32_t>(ctx, (size_t)n_unique * PCLEAN_MAX_CTX);
  int32_t* excl2 = scratch<int32_t>(ctx, n_unique);
  double* lse_u = scratch<double>(ctx, n_unique);
  if (!row2 || !ctx2 || !excl2 || !lse_u) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  hipLaunchKernelGGL(item_unique_kernel, grid1(n), dim3(256), 0, ctx->stream, n, idx_s, head, uid, il.row, il.ctx, excl,
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
static int sample_children(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                           uint64_t seed, uint32_t sweep, int32_t* vals, int n_nodes) {
  Block& b = ctx->block[block_id];
  const pclean_node& n = b.nodes[node_id];
  const CandTable& t = ctx->cand[n.table];
  SweepState* s = st(ctx);
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
      HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
      hipLaunchKernelGGL(compact_new_kernel, grid1(il.n), dim3(256), 0, ctx->stream, (size_t)il.n, draws, 0,
                         s->counter.p, nullptr, nullptr);
      unsigned int cnt = 0;
      PCLEAN_READ_COUNT(ctx, s->counter.p, &cnt);
      if (cnt) {
        int32_t* list = scratch<int32_t>(ctx, cnt);
        int32_t* row = scratch<int32_t>(ctx, cnt);
        int32_t* cx = scratch<int32_t>(ctx, (size_t)cnt * PCLEAN_MAX_CTX);
        int32_t* part = scratch<int32_t>(ctx, cnt);
        int32_t* org = scratch<int32_t>(ctx, cnt);
        int32_t* sub_excl = scratch<int32_t>(ctx, cnt);
        if (!list || !row || !cx || !part || !org || !sub_excl) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
        hipLaunchKernelGGL(compact_new_kernel, grid1(il.n), dim3(256), 0, ctx->stream, (size_t)il.n, draws, 1,
                           s->counter.p, list, nullptr);
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

static int ensure_plan_dev(pclean_ctx* ctx, int block_id) {
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

// ---------------------------------------------------------------------------
extern "C" int pclean_set_active_rows(pclean_ctx* ctx, int32_t begin, int32_t count) {
  if (!ctx || begin < 0 || count < -1 || (count >= 0 && (int64_t)begin + count > ctx->n_rows))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_active_rows: window outside the loaded rows");
  ctx->active_begin = count < 0 ? 0 : begin;
  ctx->active_count = count;
  return PCLEAN_OK;
}

// particle choice of a latent row: every particle has the same weight (all sub-plans enumerated)
__global__ void latent_choice_kernel(int n, int P, int use_mh, const int32_t* keys, uint64_t seed, uint32_t sweep,
                                     uint32_t block_id, int32_t* chosen) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t rr = (uint32_t)keys[i];
  const uint32_t pid = 0x1000u + block_id;
  int c;
  if (use_mh && P >= 2) {  // row_inference.jl:161-162 with w1 == w0
    const double ratio = 0.5 / (1e-10 + 0.5);
    c = pclean_u01(pclean_rand64(seed, rr, PCLEAN_SITE_MH, pid, sweep)) < ratio ? 1 : 0;
  } else {
    const uint64_t U = (uint64_t)P << PCLEAN_FIX_BITS;
    c = (int)(pclean_mulhi64(pclean_rand64(seed, rr, PCLEAN_SITE_FINAL, pid, sweep), U) >> PCLEAN_FIX_BITS);
  }
  chosen[i] = c;
}
__global__ void mark_positive_kernel(int n, const int32_t* v, int32_t* flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = v[i] > 0 ? PCLEAN_CHOICE_NEW : 0;
}
__global__ void latent_items_kernel(int n, const int32_t* list, const int32_t* keys, const int32_t* ev_off,
                                    const int32_t* chosen, int32_t* rng, int32_t* ev_lo, int32_t* ev_hi,
                                    int32_t* particle, int32_t* origin) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int t = list[j];
  rng[j] = keys[t];
  ev_lo[j] = ev_off[t];
  ev_hi[j] = ev_off[t + 1];
  particle[j] = chosen[t];
  origin[j] = t;
}

// ---- prior proposals for a latent class (use_dd_proposals = false) -----------------------------------------------
static int prior_mode_supported(pclean_ctx* ctx, const Block& b, const char* who);
static int upload_plan_nodes(pclean_ctx* ctx, int bi, const NodeDev** nds, const int32_t** n_children,
                             const int32_t** child_begin, const int32_t** children);
__global__ void retain_first_kernel(int n_items, int P, const int32_t* __restrict__ cur, int32_t* __restrict__ draws) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_items) draws[(size_t)t * P] = cur[t];  // particle 0 keeps the row's current value
}
__global__ void set_node_col_kernel(int n, const int32_t* __restrict__ src, int n_nodes, int node, int32_t* __restrict__ vals) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) vals[(size_t)j * n_nodes + node] = src[j];
}
__global__ void latent_prior_items_kernel(int n, int P, int n_items, const int32_t* __restrict__ list,
                                          const int32_t* __restrict__ keys, const int32_t* __restrict__ cur,
                                          int32_t* __restrict__ rng, int32_t* __restrict__ particle,
                                          int32_t* __restrict__ origin, int32_t* __restrict__ excl,
                                          int32_t* __restrict__ ev_lo, int32_t* __restrict__ ev_hi,
                                          const int32_t* __restrict__ off) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int idx = list[j], t = idx / P;
  rng[j] = keys[t];
  particle[j] = idx - t * P;
  origin[j] = idx;
  excl[j] = cur[t];
  ev_lo[j] = off[t];
  ev_hi[j] = off[t + 1];
}
// final choice among the P particles of every latent row (row_inference.jl:158-165), weights row-major [n_items][P]
template <int PMAX>
__global__ void latent_prior_choice_kernel(int n_items, int P, int use_mh, const double* __restrict__ w,
                                           const int32_t* __restrict__ keys, uint64_t seed, uint32_t sweep, uint32_t block_id,
                                           int32_t* __restrict__ chosen) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_items) return;
  FixW<PMAX> f;
  fix_weights<PMAX>(w + (size_t)t * P, (size_t)1, P, f);
  const uint32_t rr = (uint32_t)keys[t], pid = 0x1000u + block_id;
  int c;
  if (use_mh && P >= 2) {
    const double Ud = (double)f.U;
    const double w0 = (double)f.u[0] / Ud, w1 = (double)f.u[PMAX > 1 ? 1 : 0] / Ud;
    double ratio = w1 / (1e-10 + w0);
    if (ratio > 1.0) ratio = 1.0;
    c = (f.U != 0 && pclean_u01(pclean_rand64(seed, rr, PCLEAN_SITE_MH, pid, sweep)) < ratio) ? 1 : 0;
  } else {
    c = fix_pick<PMAX>(f, P, pclean_rand64(seed, rr, PCLEAN_SITE_FINAL, pid, sweep));
  }
  chosen[t] = c;
}
__global__ void gather_chosen_vals_kernel(int n_items, int P, int n_nodes, const int32_t* __restrict__ chosen,
                                          const int32_t* __restrict__ pv, int32_t* __restrict__ vals) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (size_t)n_items * n_nodes) return;
  const int t = (int)(q / n_nodes), k = (int)(q - (size_t)t * n_nodes);
  const int c = chosen[t];
  vals[q] = c > 0 ? pv[((size_t)t * P + c) * n_nodes + k] : -2;
}

// Side streams of pclean_sweep_latent (sweep_state.h); PCLEAN_LATENT_STREAMS=0 keeps everything on the library's stream.
static int side_streams(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  if (s->n_side >= 0) return s->n_side;
  const char* e = getenv("PCLEAN_LATENT_STREAMS");
  int want = e ? atoi(e) : 6;
  want = std::max(0, std::min(want, (int)SweepState::MAX_SIDE));
  s->n_side = 0;
  if (want > 0 && hipEventCreateWithFlags(&s->side_fork, hipEventDisableTiming) != hipSuccess) return 0;
  for (int k = 0; k < want; ++k) {
    if (hipStreamCreateWithFlags(&s->side[k], hipStreamNonBlocking) != hipSuccess) break;
    if (hipEventCreateWithFlags(&s->side_join[k], hipEventDisableTiming) != hipSuccess) break;
    if (hipEventCreateWithFlags(&s->side_mid[k], hipEventDisableTiming) != hipSuccess) break;
    s->n_side = k + 1;
  }
  return s->n_side;
}
// Work of one call spread over the side streams: fork() after the inputs are queued on the library's stream, use(i) to
// issue the i-th independent piece, join() before anything on the library's stream reads the results.  An early
// return (error) waits for the side streams on the host: the scratch pool they use is rewound by the next call.
struct SideFork {
  pclean_ctx* ctx;
  SweepState* s;
  hipStream_t main;
  bool forked = false, used[SweepState::MAX_SIDE] = {};
  explicit SideFork(pclean_ctx* c) : ctx(c), s(st(c)), main(c->stream) {}
  int fork() {  // what is queued on the library's stream so far is what the side streams wait for
    if (side_streams(ctx) <= 0 || forked) return PCLEAN_OK;
    HIPCHK(ctx, hipEventRecord(s->side_fork, main));
    forked = true;
    return PCLEAN_OK;
  }
  int use(int i) {
    const int K = side_streams(ctx);
    if (K <= 0) return PCLEAN_OK;
    if (!forked) {
      const int rc = fork();
      if (rc) return rc;
    }
    const int k = i % K;
    if (!used[k]) {
      HIPCHK(ctx, hipStreamWaitEvent(s->side[k], s->side_fork, 0));
      used[k] = true;
    }
    ctx->stream = s->side[k];
    return PCLEAN_OK;
  }
  void back() { ctx->stream = main; }
  int mark() {  // the library's stream waits for what the current side stream holds so far (not for what follows on it)
    if (ctx->stream == main) return PCLEAN_OK;
    for (int k = 0; k < SweepState::MAX_SIDE; ++k)
      if (ctx->stream == s->side[k]) {
        HIPCHK(ctx, hipEventRecord(s->side_mid[k], s->side[k]));
        HIPCHK(ctx, hipStreamWaitEvent(main, s->side_mid[k], 0));
      }
    return PCLEAN_OK;
  }
  int join() {
    back();
    for (int k = 0; k < SweepState::MAX_SIDE; ++k)
      if (used[k]) {
        used[k] = false;
        HIPCHK(ctx, hipEventRecord(s->side_join[k], s->side[k]));
        HIPCHK(ctx, hipStreamWaitEvent(main, s->side_join[k], 0));
      }
    return PCLEAN_OK;
  }
  ~SideFork() {
    back();
    for (int k = 0; k < SweepState::MAX_SIDE; ++k)
      if (used[k]) (void)hipStreamSynchronize(s->side[k]);
  }
};

extern "C" int pclean_sweep_latent(pclean_ctx* ctx, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep_idx,
                                   int32_t block_id, int32_t n_roots, const int32_t* roots, int32_t n_items,
                                   const int32_t* keys, const int32_t* ev_off, const int32_t* ev_rows,
                                   const int32_t* ev_ctx, const int32_t* excl, int32_t* chosen, int32_t* vals) {
  if (!ctx || !cfg || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || n_roots <= 0 ||
      !roots || n_items < 0 || !keys || !ev_off || !excl || !chosen || !vals)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent: bad arguments");
  if (!cfg->use_dd_proposals) {
    const int rcp = prior_mode_supported(ctx, ctx->block[block_id], "pclean_sweep_latent");
    if (rcp) return rcp;
  }
  if (n_items == 0) return PCLEAN_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Block& b = ctx->block[block_id];
  const int nn = (int)b.nodes.size();
  int P = cfg->num_particles;
  const int use_mh = cfg->use_mh_instead_of_pg != 0;
  if (use_mh) P = 2;
  if (P < 1 || P > MAXP) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent: bad particle count");
  for (int r = 0; r < n_roots; ++r)
    if (roots[r] < 0 || roots[r] >= nn) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent: bad root");
  SweepState* s = st(ctx);
  {
    const int rcb = begin_call(ctx);
    if (rcb) return rcb;
  }
  if (s->counter.alloc(4)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  const int n_ev = ev_off[n_items];
  if (n_ev > 0 && !ev_rows) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep_latent: evidence rows missing");
  int32_t* d_keys = scratch<int32_t>(ctx, n_items);
  int32_t* d_off = scratch<int32_t>(ctx, (size_t)n_items + 1);
  int32_t* d_evr = scratch<int32_t>(ctx, std::max(n_ev, 1));
  int32_t* d_evc = ev_ctx ? scratch<int32_t>(ctx, (size_t)std::max(n_ev, 1) * PCLEAN_MAX_CTX) : nullptr;
  int32_t* d_excl = scratch<int32_t>(ctx, (size_t)n_roots * n_items);
  int32_t* d_chosen = scratch<int32_t>(ctx, n_items);
  int32_t* d_vals = scratch<int32_t>(ctx, (size_t)n_items * nn);
  int32_t* d_flag = scratch<int32_t>(ctx, n_items);
  if (!d_keys || !d_off || !d_evr || (ev_ctx && !d_evc) || !d_excl || !d_chosen || !d_vals || !d_flag)
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  // inputs and outputs travel through the library's page-locked staging area (ctx.h: HostStage — never the caller's pages)
  const size_t b_keys = (size_t)n_items * 4, b_off = ((size_t)n_items + 1) * 4, b_evr = (size_t)n_ev * 4,
               b_evc = (ev_ctx && n_ev) ? (size_t)n_ev * PCLEAN_MAX_CTX * 4 : 0, b_excl = (size_t)n_roots * n_items * 4,
               b_vals = (size_t)n_items * nn * 4;
  if (ctx->stage.grow(2 * b_keys + b_off + b_evr + b_evc + b_excl + b_vals + 8 * 256))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "page-locked staging alloc failed");
  ctx->stage.rewind();
  auto stage_up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    void* h = ctx->stage.take(bytes);
    memcpy(h, src, bytes);
    return hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, ctx->stream);
  };
  HIPCHK(ctx, stage_up(d_keys, keys, b_keys));
  HIPCHK(ctx, stage_up(d_off, ev_off, b_off));
  HIPCHK(ctx, stage_up(d_evr, ev_rows, b_evr));
  HIPCHK(ctx, stage_up(d_evc, ev_ctx, b_evc));
  HIPCHK(ctx, stage_up(d_excl, excl, b_excl));
  int32_t* h_chosen = (int32_t*)ctx->stage.take(b_keys);
  int32_t* h_vals = (int32_t*)ctx->stage.take(b_vals);
  int32_t* d_iop = scratch<int32_t>(ctx, std::max(n_ev, 1));
  if (!d_iop) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  if (n_ev) hipLaunchKernelGGL(item_of_pos_kernel, grid1(n_ev), dim3(256), 0, ctx->stream, n_ev, n_items, d_off, d_iop);
  s->lat_off = d_off;
  s->lat_item_of_pos = d_iop;
  s->lat_items = n_items;
  s->lat_ev = n_ev;
  s->lat_max_ev = 0;
  for (int t = 0; t < n_items; ++t) s->lat_max_ev = std::max(s->lat_max_ev, ev_off[t + 1] - ev_off[t]);
  s->lat_agg.clear();
  if (!cfg->use_dd_proposals) {
    // Prior proposals (block_proposal.jl:168): particle 0 keeps the row's current values (excl[r][t]: current referent
    // of a reference slot, current OPTION of a choice), every other particle draws each attribute from its prior;
    // weight = likelihood of the referring rows given the particle's values; final choice among the particles.
    const size_t NPi = (size_t)n_items * P;
    int32_t* pv = scratch<int32_t>(ctx, NPi * nn);
    int32_t* draws = scratch<int32_t>(ctx, NPi);
    double* wl = scratch<double>(ctx, NPi);
    if (!pv || !draws || !wl) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    hipLaunchKernelGGL(fill_i32_kernel, grid1(NPi * nn), dim3(256), 0, ctx->stream, pv, NPi * nn, -2);
    for (int r = 0; r < n_roots; ++r) {
      const int root = roots[r];
      const pclean_node& rn = b.nodes[root];
      const int32_t* cur_r = d_excl + (size_t)r * n_items;
      ItemList ilp{n_items, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_keys};
      ctx->prior_mode = true;
      int rc = eval_node(ctx, block_id, root, ilp, rn.kind == PCLEAN_NODE_FK ? cur_r : nullptr, seed, sweep_idx, P, nullptr,
                         draws, nullptr, nullptr, false);
      if (rc) {
        ctx->prior_mode = false;
        return rc;
      }
      hipLaunchKernelGGL(retain_first_kernel, grid1(n_items), dim3(256), 0, ctx->stream, n_items, P, cur_r, draws);
      hipLaunchKernelGGL(set_node_col_kernel, grid1(NPi), dim3(256), 0, ctx->stream, (int)NPi, draws, nn, root, pv);
      if (rn.kind == PCLEAN_NODE_FK && rn.n_children > 0) {
        HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
        int32_t* l2 = scratch<int32_t>(ctx, NPi);
        if (!l2) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        hipLaunchKernelGGL(compact_new_kernel, grid1(NPi), dim3(256), 0, ctx->stream, NPi, draws, 1, s->counter.p, l2, nullptr);
        unsigned int c2 = 0;
        PCLEAN_READ_COUNT(ctx, s->counter.p, &c2);
        if (c2) {
          int32_t* rng2 = scratch<int32_t>(ctx, c2);
          int32_t* part2 = scratch<int32_t>(ctx, c2);
          int32_t* org2 = scratch<int32_t>(ctx, c2);
          int32_t* ex2 = scratch<int32_t>(ctx, c2);
          int32_t* evl2 = scratch<int32_t>(ctx, c2);
          int32_t* evh2 = scratch<int32_t>(ctx, c2);
          if (!rng2 || !part2 || !org2 || !ex2 || !evl2 || !evh2) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          hipLaunchKernelGGL(latent_prior_items_kernel, grid1(c2), dim3(256), 0, ctx->stream, (int)c2, P, n_items, l2, d_keys,
                             cur_r, rng2, part2, org2, ex2, evl2, evh2, d_off);
          ItemList sub{(int)c2, nullptr, nullptr, part2, org2, nullptr, nullptr, nullptr, nullptr, rng2};
          rc = sample_children(ctx, block_id, root, sub, ex2, seed, sweep_idx, pv, nn);
          if (rc) {
            ctx->prior_mode = false;
            return rc;
          }
        }
      }
      ctx->prior_mode = false;
    }
    // likelihood of every (row, particle)
    const NodeDev* nds;
    const int32_t *dnc, *dcb, *dch;
    int rc = upload_plan_nodes(ctx, block_id, &nds, &dnc, &dcb, &dch);
    if (rc) return rc;
    int32_t* d_roots = scratch<int32_t>(ctx, n_roots);
    const AggDev** d_aggs = (const AggDev**)scratch<unsigned char>(ctx, sizeof(void*) * nn);
    if (!d_roots || !d_aggs) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    ItemList ilev{n_items, nullptr, nullptr, nullptr, nullptr, d_off, d_off + 1, d_evr, d_evc, d_keys};
    std::vector<const AggDev*> h_aggs(nn, nullptr);
    for (int node = 0; node < nn; ++node)
      if (b.nodes[node].n_terms > 0) {
        rc = ensure_agg(ctx, block_id, node, ilev, &h_aggs[node]);
        if (rc) return rc;
      }
    HIPCHK(ctx, hipMemcpyAsync(d_aggs, h_aggs.data(), sizeof(void*) * nn, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_roots, roots, (size_t)n_roots * 4, hipMemcpyHostToDevice, ctx->stream));
    PCLEAN_SYNC(ctx);  // h_aggs goes out of scope
    ItemsDev itd{n_items, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, d_off, d_off + 1, d_evr, d_evc, d_keys, nullptr,
                 nullptr, 0, 0, nullptr, nullptr};
    rc = pclean_launch_prior_terms_ev(ctx, n_items, P, nn, nds, d_aggs, dnc, dcb, dch, n_roots, d_roots, itd, pv, wl);
    if (rc) return rc;
    DISPATCH_PMAX(P, hipLaunchKernelGGL(latent_prior_choice_kernel<PMAX>, grid1(n_items), dim3(256), 0, ctx->stream, n_items, P,
                                        use_mh, wl, d_keys, seed, sweep_idx, (uint32_t)block_id, d_chosen));
    hipLaunchKernelGGL(gather_chosen_vals_kernel, grid1((size_t)n_items * nn), dim3(256), 0, ctx->stream, n_items, P, nn,
                       d_chosen, pv, d_vals);
    HIPCHK(ctx, hipMemcpyAsync(h_chosen, d_chosen, b_keys, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_vals, d_vals, b_vals, hipMemcpyDeviceToHost, ctx->stream));
    PCLEAN_SYNC(ctx);
    memcpy(chosen, h_chosen, b_keys);
    memcpy(vals, h_vals, b_vals);
    s->lat_agg.clear();
    if (s->prof_on) prof_collect(ctx);
    return finish_call(ctx);
  }
  hipLaunchKernelGGL(latent_choice_kernel, grid1(n_items), dim3(256), 0, ctx->stream, n_items, P, use_mh, d_keys, seed,
                     sweep_idx, (uint32_t)block_id, d_chosen);
  hipLaunchKernelGGL(fill_i32_kernel, grid1((size_t)n_items * nn), dim3(256), 0, ctx->stream, d_vals,
                     (size_t)n_items * nn, -2);
  // rows that take a fresh particle
  hipLaunchKernelGGL(mark_positive_kernel, grid1(n_items), dim3(256), 0, ctx->stream, n_items, d_chosen, d_flag);
  HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
  hipLaunchKernelGGL(compact_new_kernel, grid1(n_items), dim3(256), 0, ctx->stream, (size_t)n_items, d_flag, 0,
                     s->counter.p, nullptr, nullptr);
  unsigned int cnt = 0;
  PCLEAN_READ_COUNT(ctx, s->counter.p, &cnt);
  if (cnt) {
    int32_t* list = scratch<int32_t>(ctx, cnt);
    int32_t* rng = scratch<int32_t>(ctx, cnt);
    int32_t* evl = scratch<int32_t>(ctx, cnt);
    int32_t* evh = scratch<int32_t>(ctx, cnt);
    int32_t* part = scratch<int32_t>(ctx, cnt);
    int32_t* org = scratch<int32_t>(ctx, cnt);
    if (!list || !rng || !evl || !evh || !part || !org) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(compact_new_kernel, grid1(n_items), dim3(256), 0, ctx->stream, (size_t)n_items, d_flag, 1,
                       s->counter.p, list, nullptr);
    hipLaunchKernelGGL(latent_items_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, list, d_keys, d_off,
                       d_chosen, rng, evl, evh, part, org);
    ItemList il{(int)cnt, nullptr, nullptr, part, org, evl, evh, d_evr, d_evc, rng};
    // The roots of a latent row's plan are independent given its evidence (each writes its own column of d_vals, draws
    // at its own RNG site).  Three passes: (A) the reference slots' enumerations on the library's stream — queued first,
    // nothing in a sub-batch's pass waits for the host; (B) the option lists, each on a side stream, overlapping (A) and
    // each other; (C) back on the library's stream, per reference slot: how many rows proposed a NEW referent (the one
    // count the host needs) and the sampling of those referents' contents.  A batch large enough for the gate of the
    // new-row branch (a count read-back inside eval_node) queues its option lists before (A) instead.
    SideFork sf(ctx);
    std::vector<int> fk_roots, leaf_roots;
    for (int r = 0; r < n_roots; ++r) (b.nodes[roots[r]].kind == PCLEAN_NODE_LEAF ? leaf_roots : fk_roots).push_back(r);
    unsigned int* c2ctr = scratch<unsigned int>(ctx, std::max(n_roots, 1));
    if (!c2ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    HIPCHK(ctx, hipMemsetAsync(c2ctr, 0, (size_t)std::max(n_roots, 1) * sizeof(unsigned int), ctx->stream));
    std::vector<int32_t*> fk_ex(n_roots, nullptr), fk_l2(n_roots, nullptr);
    {
      const int rcf = sf.fork();  // (before pass A is queued: the side streams wait for the inputs, not for pass A)
      if (rcf) return rcf;
    }
    const char* gm = getenv("PCLEAN_GATE_MIN");
    const bool fk_first = (int)cnt < (gm ? atoi(gm) : 2048);
    int n_side_used = 0;
    auto pass_fk = [&]() -> int {
      for (int r : fk_roots) {
        const int root = roots[r];
        const pclean_node& rn = b.nodes[root];
        if (fk_first) {  // (no count read-back, no shared counter on this path below the gate's size: a stream of its own)
          const int rcs = sf.use(n_side_used++);
          if (rcs) return rcs;
        }
        int32_t* ex = scratch<int32_t>(ctx, cnt);
        int32_t* draws = scratch<int32_t>(ctx, cnt);
        if (!ex || !draws) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        hipLaunchKernelGGL(gather_i32_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, list,
                           d_excl + (size_t)r * n_items, ex);
        fk_ex[r] = ex;
        int rc = eval_node(ctx, block_id, root, il, ex, seed, sweep_idx, 1, nullptr, draws, nullptr, nullptr, false);
        if (rc) return rc;
        hipLaunchKernelGGL(scatter_vals_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, org, draws, nn, root,
                           d_vals);
        if (rn.n_children > 0) {  // rows that proposed a NEW referent: listed now, counted by the host in pass C
          fk_l2[r] = scratch<int32_t>(ctx, cnt);
          if (!fk_l2[r]) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          hipLaunchKernelGGL(compact_new_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (size_t)cnt, draws, 1, c2ctr + r,
                             fk_l2[r], nullptr);
        }
        const int rcm = sf.mark();  // pass C (library's stream) follows this root's pass A, not the option lists
        if (rcm) return rcm;
      }
      sf.back();
      return PCLEAN_OK;
    };
    auto pass_leaf = [&]() -> int {
      for (size_t oi = 0; oi < leaf_roots.size(); ++oi) {
        const int r = leaf_roots[oi];
        const int root = roots[r];
        // (the reference slots keep their streams to themselves: theirs are the longest chains of a sub-batch)
        const int K = side_streams(ctx), n_fk_side = fk_first ? std::min((int)fk_roots.size(), std::max(K - 1, 0)) : 0;
        const int rcs = sf.use(K > n_fk_side ? n_fk_side + (int)oi % (K - n_fk_side) : (int)oi);
        if (rcs) return rcs;
        int32_t* draws = scratch<int32_t>(ctx, cnt);
        if (!draws) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        int rc = eval_node(ctx, block_id, root, il, nullptr, seed, sweep_idx, 1, nullptr, draws, nullptr, nullptr, false);
        if (rc) return rc;
        hipLaunchKernelGGL(scatter_vals_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, org, draws, nn, root,
                           d_vals);
      }
      sf.back();
      return PCLEAN_OK;
    };
    int rc = fk_first ? pass_fk() : pass_leaf();
    if (!rc) rc = fk_first ? pass_leaf() : pass_fk();
    if (rc) return rc;
    for (int r : fk_roots) {
      const int root = roots[r];
      if (!fk_l2[r]) continue;
      // referents proposed as NEW: sample their contents with the same evidence
      unsigned int c2 = 0;
      PCLEAN_READ_COUNT(ctx, c2ctr + r, &c2);
      if (c2) {
        int32_t* l2 = fk_l2[r];
        int32_t* row2 = scratch<int32_t>(ctx, c2);
        int32_t* cx2 = scratch<int32_t>(ctx, (size_t)c2 * PCLEAN_MAX_CTX);
        int32_t* part2 = scratch<int32_t>(ctx, c2);
        int32_t* org2 = scratch<int32_t>(ctx, c2);
        int32_t* evl2 = scratch<int32_t>(ctx, c2);
        int32_t* evh2 = scratch<int32_t>(ctx, c2);
        int32_t* rng2 = scratch<int32_t>(ctx, c2);
        int32_t* ex2 = scratch<int32_t>(ctx, c2);
        if (!row2 || !cx2 || !part2 || !org2 || !evl2 || !evh2 || !rng2 || !ex2)
          return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
        hipLaunchKernelGGL(sublist_items_kernel, grid1(c2), dim3(256), 0, ctx->stream, (int)c2, l2, il.row, il.ctx,
                           il.particle, il.origin, row2, cx2, part2, org2, il.ev_lo, il.ev_hi, il.rng_row, evl2, evh2,
                           rng2);
        hipLaunchKernelGGL(gather_i32_kernel, grid1(c2), dim3(256), 0, ctx->stream, (int)c2, l2, fk_ex[r], ex2);
        ItemList sub{(int)c2, nullptr, nullptr, part2, org2, evl2, evh2, d_evr, d_evc, rng2};
        rc = sample_children(ctx, block_id, root, sub, ex2, seed, sweep_idx, d_vals, nn);
        if (rc) return rc;
      }
    }
    const int rcj = sf.join();
    if (rcj) return rcj;
  }
  HIPCHK(ctx, hipMemcpyAsync(h_chosen, d_chosen, b_keys, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(h_vals, d_vals, b_vals, hipMemcpyDeviceToHost, ctx->stream));
  {
    const int rcq = queue_over_copy(ctx);  // the sync-free re-runs' counts ride on the call's one synchronisation
    if (rcq) return rcq;
  }
  PCLEAN_SYNC(ctx);
  memcpy(chosen, h_chosen, b_keys);
  memcpy(vals, h_vals, b_vals);
  apply_over_stats(ctx);
  s->lat_agg.clear();
  if (s->prof_on) prof_collect(ctx);
  return finish_call(ctx);
}

extern "C" int pclean_debug_root_flags(pclean_ctx* ctx, int32_t n_rows, int32_t* out) {
  if (!ctx || !out || n_rows <= 0) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_debug_root_flags: bad arguments");
  SweepState* s = st(ctx);
  if (!s->dbg_desc || s->dbg_items != n_rows)
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_debug_root_flags: the last call was not a pclean_sweep of %d rows "
                                              "through the compact-table root kernel", n_rows);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<int32_t> d;
  if (d.alloc(n_rows)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  int rc = pclean_launch_root_flags(ctx, s->dbg_groups, s->dbg_desc, s->dbg_grp_off, s->dbg_members, s->dbg_oflag, d.p);
  hipError_t e = rc ? hipSuccess : hipMemcpyAsync(out, d.p, (size_t)n_rows * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  d.release();
  if (rc) return rc;
  if (e != hipSuccess) return pclean_fail(ctx, PCLEAN_ERR_HIP, "pclean_debug_root_flags: %s", hipGetErrorString(e));
  return PCLEAN_OK;
}

extern "C" int pclean_debug_global_evidence_sort(pclean_ctx* ctx, int32_t on) {
  if (!ctx) return PCLEAN_ERR_ARG;
  ctx->no_item_agg = on != 0;
  return PCLEAN_OK;
}

extern "C" int pclean_debug_force_generic(pclean_ctx* ctx, int32_t on) {
  if (!ctx) return PCLEAN_ERR_ARG;
  ctx->force_generic = on != 0;
  return PCLEAN_OK;
}

extern "C" int pclean_set_row_offset(pclean_ctx* ctx, int64_t row_offset) {
  if (!ctx || row_offset < 0) return pclean_fail(ctx, PCLEAN_ERR_ARG, "bad row offset");
  st(ctx)->row_offset = row_offset;
  return PCLEAN_OK;
}

extern "C" int pclean_score_node(pclean_ctx* ctx, int32_t block_id, int32_t node_id, int32_t n_items,
                                 const int32_t* rows, const int32_t* ctxv, const int32_t* excl, const double* snew,
                                 uint64_t seed, uint32_t sweep, int32_t n_draws, double* lse, double* scores,
                                 int32_t* draws) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || n_items <= 0 || !rows ||
      n_draws < 0 || (n_draws > 0 && !draws))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_score_node: bad arguments");
  Block& b = ctx->block[block_id];
  if (node_id < 0 || node_id >= (int)b.nodes.size()) return pclean_fail(ctx, PCLEAN_ERR_ARG, "bad node id");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  {
    const int rcb = begin_call(ctx);
    if (rcb) return rcb;
  }
  const pclean_node& n = b.nodes[node_id];
  const CandTable& t = ctx->cand[n.table];
  const int nc = t.n_rows + (n.kind == PCLEAN_NODE_FK ? 1 : 0);
  int32_t* d_rows = scratch<int32_t>(ctx, n_items);
  int32_t* d_ctx = ctxv ? scratch<int32_t>(ctx, (size_t)n_items * PCLEAN_MAX_CTX) : nullptr;
  int32_t* d_excl = excl ? scratch<int32_t>(ctx, n_items) : nullptr;
  double* d_snew = snew ? scratch<double>(ctx, n_items) : nullptr;
  double* d_lse = scratch<double>(ctx, n_items);
  double* d_scores = scores ? scratch<double>(ctx, (size_t)n_items * nc) : nullptr;
  int32_t* d_draws = n_draws ? scratch<int32_t>(ctx, (size_t)n_items * n_draws) : nullptr;
  if (!d_rows || !d_lse || (ctxv && !d_ctx) || (excl && !d_excl) || (snew && !d_snew) || (scores && !d_scores) ||
      (n_draws && !d_draws))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  HIPCHK(ctx, hipMemcpyAsync(d_rows, rows, n_items * 4, hipMemcpyHostToDevice, ctx->stream));
  if (ctxv) HIPCHK(ctx, hipMemcpyAsync(d_ctx, ctxv, (size_t)n_items * PCLEAN_MAX_CTX * 4, hipMemcpyHostToDevice, ctx->stream));
  if (excl) HIPCHK(ctx, hipMemcpyAsync(d_excl, excl, n_items * 4, hipMemcpyHostToDevice, ctx->stream));
  if (snew) HIPCHK(ctx, hipMemcpyAsync(d_snew, snew, n_items * 8, hipMemcpyHostToDevice, ctx->stream));
  ItemList il{n_items, d_rows, d_ctx, nullptr, nullptr};
  // snew given: score this node alone; snew null on an FK node: evaluate its sub-tree
  int rc = eval_node(ctx, block_id, node_id, il, d_excl, seed, sweep, n_draws, d_lse, d_draws, d_scores,
                     (n.kind == PCLEAN_NODE_FK && snew) ? d_snew : nullptr, false);
  if (rc) return rc;
  if (lse) HIPCHK(ctx, hipMemcpyAsync(lse, d_lse, n_items * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (scores) HIPCHK(ctx, hipMemcpyAsync(scores, d_scores, (size_t)n_items * nc * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (n_draws) HIPCHK(ctx, hipMemcpyAsync(draws, d_draws, (size_t)n_items * n_draws * 4, hipMemcpyDeviceToHost, ctx->stream));
  PCLEAN_SYNC(ctx);
  return finish_call(ctx);
}

// pclean_score_node for EVIDENCE SETS: item t is a latent row scored against the observed rows
// ev_rows[ev_off[t] .. ev_off[t + 1]) (with their per-row ctx) — one plan node of a latent class's plan, as
// pclean_sweep_latent evaluates it (same aggregation, same kernels), with the per-candidate scores returned.
extern "C" int pclean_score_node_ev(pclean_ctx* ctx, int32_t block_id, int32_t node_id, int32_t n_items,
                                    const int32_t* keys, const int32_t* ev_off, const int32_t* ev_rows,
                                    const int32_t* ev_ctx, const int32_t* excl, uint64_t seed, uint32_t sweep,
                                    int32_t n_draws, double* lse, double* scores, int32_t* draws) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || n_items <= 0 || !keys ||
      !ev_off || n_draws < 0 || n_draws > 1 || (n_draws > 0 && !draws))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_score_node_ev: bad arguments");
  Block& b = ctx->block[block_id];
  if (node_id < 0 || node_id >= (int)b.nodes.size()) return pclean_fail(ctx, PCLEAN_ERR_ARG, "bad node id");
  const int n_ev = ev_off[n_items];
  if (n_ev < 0 || (n_ev > 0 && !ev_rows)) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_score_node_ev: evidence rows missing");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  {
    const int rcb = begin_call(ctx);
    if (rcb) return rcb;
  }
  SweepState* s = st(ctx);
  const pclean_node& n = b.nodes[node_id];
  const CandTable& t = ctx->cand[n.table];
  const int nc = t.n_rows + (n.kind == PCLEAN_NODE_FK ? 1 : 0);
  int32_t* d_keys = scratch<int32_t>(ctx, n_items);
  int32_t* d_off = scratch<int32_t>(ctx, (size_t)n_items + 1);
  int32_t* d_evr = scratch<int32_t>(ctx, std::max(n_ev, 1));
  int32_t* d_evc = ev_ctx ? scratch<int32_t>(ctx, (size_t)std::max(n_ev, 1) * PCLEAN_MAX_CTX) : nullptr;
  int32_t* d_iop = scratch<int32_t>(ctx, std::max(n_ev, 1));
  int32_t* d_excl = excl ? scratch<int32_t>(ctx, n_items) : nullptr;
  double* d_lse = scratch<double>(ctx, n_items);
  double* d_scores = scores ? scratch<double>(ctx, (size_t)n_items * nc) : nullptr;
  int32_t* d_draws = n_draws ? scratch<int32_t>(ctx, n_items) : nullptr;
  if (!d_keys || !d_off || !d_evr || (ev_ctx && !d_evc) || !d_iop || (excl && !d_excl) || !d_lse || (scores && !d_scores) ||
      (n_draws && !d_draws))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  HIPCHK(ctx, hipMemcpyAsync(d_keys, keys, (size_t)n_items * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_off, ev_off, ((size_t)n_items + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_ev) HIPCHK(ctx, hipMemcpyAsync(d_evr, ev_rows, (size_t)n_ev * 4, hipMemcpyHostToDevice, ctx->stream));
  if (ev_ctx && n_ev)
    HIPCHK(ctx, hipMemcpyAsync(d_evc, ev_ctx, (size_t)n_ev * PCLEAN_MAX_CTX * 4, hipMemcpyHostToDevice, ctx->stream));
  if (excl) HIPCHK(ctx, hipMemcpyAsync(d_excl, excl, (size_t)n_items * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_ev) hipLaunchKernelGGL(item_of_pos_kernel, grid1(n_ev), dim3(256), 0, ctx->stream, n_ev, n_items, d_off, d_iop);
  s->lat_off = d_off;
  s->lat_item_of_pos = d_iop;
  s->lat_items = n_items;
  s->lat_ev = n_ev;
  s->lat_max_ev = 0;
  for (int i = 0; i < n_items; ++i) s->lat_max_ev = std::max(s->lat_max_ev, ev_off[i + 1] - ev_off[i]);
  s->lat_agg.clear();
  ItemList il{n_items, nullptr, nullptr, nullptr, nullptr, d_off, d_off + 1, d_evr, d_evc, d_keys};
  int rc = eval_node(ctx, block_id, node_id, il, d_excl, seed, sweep, n_draws, d_lse, d_draws, d_scores, nullptr, false);
  s->lat_agg.clear();
  if (rc) return rc;
  if (lse) HIPCHK(ctx, hipMemcpyAsync(lse, d_lse, (size_t)n_items * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (scores) HIPCHK(ctx, hipMemcpyAsync(scores, d_scores, (size_t)n_items * nc * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (n_draws) HIPCHK(ctx, hipMemcpyAsync(draws, d_draws, (size_t)n_items * 4, hipMemcpyDeviceToHost, ctx->stream));
  PCLEAN_SYNC(ctx);
  return finish_call(ctx);
}

// Can a particle of this sweep draw the ProposalDummyValue of some option list of block bi?  Cacheable lists know it
// per observed value (ensure_leaf_cache: weight of the dummy option); any other list with a dummy is assumed to.
static int block_dummy_drawable(pclean_ctx* ctx, int bi, bool* out) {
  Block& b = ctx->block[bi];
  *out = false;
  for (int node = 0; node < (int)b.nodes.size(); ++node) {
    const pclean_node& n = b.nodes[node];
    if (n.kind != PCLEAN_NODE_LEAF || n.dummy_value == 0) continue;
    if (!n.cacheable || (node < (int)b.node_gauss.size() && b.node_gauss[node] >= 0)) {
      *out = true;
      return PCLEAN_OK;
    }
    const double* cache;
    const int32_t* ocol;
    int n_obs;
    int rc = ensure_leaf_cache(ctx, bi, node, &cache, &ocol, &n_obs);
    if (rc) return rc;
    if (b.leaf_drawable[node] != 0) {
      static const bool dbg = getenv("PCLEAN_DEBUG_DUMMY") != nullptr;
      if (dbg) fprintf(stderr, "[pclean] block %d node %d: its ProposalDummyValue can be drawn (flag %d)\n", bi, node, b.leaf_drawable[node]);
      *out = true;
      return PCLEAN_OK;
    }
  }
  return PCLEAN_OK;
}

// w[slot] += correction for every NEW slot of block bi whose sampled new row chose a ProposalDummyValue
static int apply_dummy_corrections(pclean_ctx* ctx, int bi, const int32_t* new_slots, const int32_t* vals, int n_new, int N,
                                   uint64_t seed, uint32_t sweep, double* w) {
  if (n_new <= 0) return PCLEAN_OK;
  Block& b = ctx->block[bi];
  SweepState* s = st(ctx);
  const int nn = (int)b.nodes.size();
  std::vector<DummyLeafDev> leaves;
  bool need_strings = false;
  int64_t slot_cells = 0;  // distance-matrix cells one NEW slot can need: every term of every dummy-bearing list
  for (int node = 0; node < nn; ++node) {
    const pclean_node& n = b.nodes[node];
    if (n.kind != PCLEAN_NODE_LEAF || n.dummy_value == 0) continue;
    const CandTable& t = ctx->cand[n.table];
    if (!t.valid || !t.is_options) return pclean_fail(ctx, PCLEAN_ERR_STATE, "node %d: option table not set", node);
    DummyLeafDev lf{};
    lf.node = node;
    lf.dummy_val = n.dummy_value - 1;
    lf.kind = n.dummy_spec & 0xff;
    lf.min_len = (n.dummy_spec >> 8) & 0xff;
    lf.max_len = (n.dummy_spec >> 16) & 0xff;
    lf.opt_vals = t.cols.p;
    lf.opt_logp = t.logc_full.p;
    if (lf.kind == PCLEAN_DUMMY_STRING_PRIOR)
      for (int ti = 0; ti < n.n_terms; ++ti) {
        const pclean_term& tm = b.terms[n.term_begin + ti];
        if (tm.dens_kind != PCLEAN_DENS_ADD_TYPOS || tm.ctx_slot >= 0) continue;  // JuliaNode terms keep the placeholder
        const PairTable& pt = ctx->pair[tm.pair_table];
        if (!pt.valid) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pair table %d not built", tm.pair_table);
        if (!pt.obs_ids.p) continue;  // host-computed table: no strings to compare a drawn value with
        if (lf.n_terms >= DUMMY_MAX_TERMS)
          return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "node %d: more than %d observations below a dummy-bearing choice", node,
                             DUMMY_MAX_TERMS);
        if (pt.max_obs_len > DUMMY_MAX_LEN)
          return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "node %d: observed strings longer than %d symbols below a dummy-bearing "
                                                       "choice", node, DUMMY_MAX_LEN);
        DummyTermDev& td = lf.t[lf.n_terms++];
        td.obs_col = ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows + ctx->active_begin;
        td.pair = pt.d.p;
        td.lat_len = pt.lat_len.p;
        td.obs_ids = pt.obs_ids.p;
        td.n_lat = pt.n_lat;
        td.elem_bytes = pt.elem_bytes;
        td.max_typos = tm.max_typos;
        td.dist_mode = pt.dist_mode;
        td.dp_off = slot_cells;
        slot_cells += (int64_t)(pt.max_obs_len + 2) * (lf.max_len + 2);
        need_strings = true;
      }
    leaves.push_back(lf);
  }
  if (leaves.empty()) return PCLEAN_OK;
  if (need_strings && !ctx->lm_valid)
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "a StringPrior dummy value has observations below it: pclean_set_lm_tables first");
  if (leaves.size() > DUMMY_MAX_LEAVES && need_strings)
    return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "block %d: more than %d dummy-bearing option lists", bi, DUMMY_MAX_LEAVES);
  if (s->dummy_ctr.alloc(2) || (need_strings && s->dummy_dp.alloc(DUMMY_DP_ARENA)))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (!s->dummy_used) HIPCHK(ctx, hipMemsetAsync(s->dummy_ctr.p, 0, 2 * sizeof(unsigned int), ctx->stream));
  s->dummy_used = true;
  ProfScope ps(ctx, "dummy_value_weights");
  // slices of NEW slots whose worst-case distance matrices fit the arena
  const int per_launch = slot_cells > 0 ? (int)std::max<int64_t>(1, (int64_t)DUMMY_DP_ARENA / slot_cells) : n_new;
  for (size_t l0 = 0; l0 < leaves.size(); l0 += DUMMY_MAX_LEAVES) {
    DummyPackDev dp{};
    dp.n_leaves = (int)std::min<size_t>(DUMMY_MAX_LEAVES, leaves.size() - l0);
    dp.site_block = bi;
    for (int i = 0; i < dp.n_leaves; ++i) dp.leaf[i] = leaves[l0 + i];
    dp.sym = ctx->sym.p;
    dp.off = ctx->off.p;
    dp.lm_init = ctx->lm_init.p;
    dp.lm_trans = ctx->lm_trans.p;
    dp.letter_sym = ctx->letter_sym.p;
    dp.nb = ctx->nb.p;
    dp.logl = ctx->logl.p;
    dp.nb_stride = ctx->max_d + 1;
    dp.dp = s->dummy_dp.p;
    dp.slot_cells = slot_cells;
    dp.dp_ctr = s->dummy_ctr.p;
    for (int j0 = 0; j0 < n_new; j0 += per_launch) {
      const int cnt = std::min(per_launch, n_new - j0);
      hipLaunchKernelGGL(dummy_correction_kernel, grid1(cnt), dim3(256), 0, ctx->stream, j0, cnt, N, new_slots, vals, nn, dp,
                         seed, sweep, s->row_offset + ctx->active_begin, w);
    }
  }
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}

// ---------------------------------------------------------------------------
__global__ void gather_moved_kernel(int n, const int32_t* list, const int32_t* choice, int32_t* out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) out[j] = choice[list[j]];
}

// use_dd_proposals = false (block_proposal.jl:168): reference slots are sampled from their CRP prior (68-84), the
// unobserved discrete choices of a new row from their prior proposals (42-56; q_cont and p cancel, a chosen
// ProposalDummyValue leaves -log(dummy mass) and gets random(dist), 58-60) and p accumulates the log-density of the
// observed choices given the sampled values (62-64): the particle's weight increment is the likelihood of its sampled
// sub-tree.  Implemented for plans whose likelihood terms are AddTypos observations (plain or through a JuliaNode).
static int prior_mode_supported(pclean_ctx* ctx, const Block& b, const char* who) {
  bool ok = b.valid && !b.is_score;
  for (const pclean_term& tm : b.terms) ok = ok && tm.dens_kind == PCLEAN_DENS_ADD_TYPOS;
  for (int g : b.node_gauss) ok = ok && g < 0;
  if (!ok)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "%s: use_dd_proposals = false (prior proposals, block_proposal.jl:168) is implemented "
                                            "for plans whose likelihood terms are AddTypos observations; this plan has equality "
                                            "constraints, MaybeSwap, Gaussian terms or a scoring block", who);
  return PCLEAN_OK;
}

// device copies of what prior_terms_kernel needs of block bi: every node with its full terms, the child lists
static int upload_plan_nodes(pclean_ctx* ctx, int bi, const NodeDev** nds, const int32_t** n_children,
                             const int32_t** child_begin, const int32_t** children) {
  Block& b = ctx->block[bi];
  const int nn = (int)b.nodes.size();
  std::vector<NodeDev> h(nn);
  std::vector<int32_t> nc(nn), cb(nn);
  for (int i = 0; i < nn; ++i) {
    int rc = build_node_dev(ctx, b, i, h[i]);
    if (rc) return rc;
    nc[i] = b.nodes[i].n_children;
    cb[i] = b.nodes[i].child_begin;
  }
  NodeDev* d = (NodeDev*)scratch<unsigned char>(ctx, sizeof(NodeDev) * nn);
  int32_t* dnc = scratch<int32_t>(ctx, nn);
  int32_t* dcb = scratch<int32_t>(ctx, nn);
  int32_t* dch = scratch<int32_t>(ctx, std::max<size_t>(b.children.size(), 1));
  if (!d || !dnc || !dcb || !dch) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  HIPCHK(ctx, hipMemcpyAsync(d, h.data(), sizeof(NodeDev) * nn, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dnc, nc.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dcb, cb.data(), nn * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!b.children.empty())
    HIPCHK(ctx, hipMemcpyAsync(dch, b.children.data(), b.children.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  PCLEAN_SYNC(ctx);  // the host vectors go out of scope
  *nds = d;
  *n_children = dnc;
  *child_begin = dcb;
  *children = dch;
  return PCLEAN_OK;
}

// per-(row, particle) context values [PCLEAN_MAX_CTX][NP]: the kernels that produce and group them touch the n_ctx
// slots the block uses; the others are zeroed here, once per shape (the few readers of whole tuples see zeros)
static int ensure_it_ctx(pclean_ctx* ctx, BlockRun& r, size_t NP, int n_ctx) {
  if (r.it_ctx.alloc(NP * PCLEAN_MAX_CTX)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (r.it_ctx_np != NP || r.it_ctx_used != n_ctx) {
    HIPCHK(ctx, hipMemsetAsync(r.it_ctx.p, 0, NP * PCLEAN_MAX_CTX * sizeof(int32_t), ctx->stream));
    r.it_ctx_np = NP;
    r.it_ctx_used = n_ctx;
  }
  return PCLEAN_OK;
}

extern "C" int pclean_sweep(pclean_ctx* ctx, const pclean_infer_config* cfg, uint64_t seed, uint32_t sweep_idx,
                            int32_t n_blocks, const int32_t* cur, int32_t* choice, int32_t* chosen_particle,
                            double* logml) {
  if (!ctx || !cfg || n_blocks <= 0 || n_blocks > PCLEAN_MAX_BLOCKS)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep: bad arguments");
  // cur == NULL: the current referents are the device-resident array of pclean_set_cur / the last pclean_commit_device
  const bool dev_cur = cur == nullptr;
  if (dev_cur && (!ctx->dev_cur_valid || ctx->dev_cur_blocks != n_blocks || !ctx->dev_cur.p))
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_sweep: cur == NULL needs pclean_set_cur (device-resident referents)");
  const bool defer = ctx->defer_outputs;
  if (defer && (choice || chosen_particle || logml))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep: deferred outputs (pclean_set_sweep_mode) take no per-row output buffers");
  const bool prior_mode = !cfg->use_dd_proposals;
  if (prior_mode)
    for (int b = 0; b < n_blocks; ++b) {
      const int rcp = prior_mode_supported(ctx, ctx->block[b], "pclean_sweep");
      if (rcp) return rcp;
    }
  const int N = ctx->active_count >= 0 ? ctx->active_count : ctx->n_rows;
  int P = cfg->num_particles;
  const int use_mh = cfg->use_mh_instead_of_pg != 0;
  if (use_mh) P = 2;  // infer_config.jl:11-13
  if (N <= 0) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_sweep: no observed rows loaded");
  if (P < 1 || P > MAXP) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep: num_particles must be in 1..%d", MAXP);
  for (int b = 0; b < n_blocks; ++b)
    if (!ctx->block[b].valid) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_sweep: block %d not loaded", b);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  SweepState* s = st(ctx);
  {
    const int rcb = begin_call(ctx);
    if (rcb) return rcb;
  }
  struct PriorModeGuard {  // no exit of this call (error returns included) leaves the context in prior mode
    pclean_ctx* c;
    ~PriorModeGuard() { c->prior_mode = false; }
  } prior_guard{ctx};
  if (!s->ev0) {
    HIPCHK(ctx, hipEventCreate(&s->ev0));
    HIPCHK(ctx, hipEventCreate(&s->ev1));
    HIPCHK(ctx, hipEventCreate(&s->evs));
    HIPCHK(ctx, hipEventCreate(&s->eve));
  }
  if (!s->h_counts) HIPCHK(ctx, hipHostMalloc((void**)&s->h_counts, 4 * PCLEAN_MAX_BLOCKS * sizeof(int32_t), hipHostMallocDefault));
  const size_t NP = (size_t)N * P;
  if (s->cur.alloc((size_t)N * n_blocks) || s->chosen.alloc(N) || s->ancestors.alloc(NP) || s->w.alloc(NP) ||
      s->log_total.alloc(N) || s->logml_inc.alloc(N) || s->logml_acc.alloc(N) || s->logml.alloc(N) ||
      s->counter.alloc(4) || s->arr_ptrs.alloc(2 * PCLEAN_MAX_BLOCKS) || s->did.alloc(N) ||
      s->tail_counts.alloc(2 * PCLEAN_MAX_BLOCKS))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  // cur_base + bi * cur_ld = current referents of block bi over the active window
  const int32_t* cur_base = s->cur.p;
  size_t cur_ld = (size_t)N;
  if (dev_cur) {
    cur_base = ctx->dev_cur.p + ctx->active_begin;
    cur_ld = (size_t)ctx->n_rows;
  } else if (ctx->cur_stride > 0 && ctx->cur_stride != N) {
    if (ctx->cur_stride < N) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_sweep: cur stride smaller than the active window");
    for (int b = 0; b < n_blocks; ++b)
      HIPCHK(ctx, hipMemcpyAsync(s->cur.p + (size_t)b * N, cur + (size_t)b * ctx->cur_stride, (size_t)N * 4,
                                 hipMemcpyHostToDevice, ctx->stream));
  } else {
    HIPCHK(ctx, hipMemcpyAsync(s->cur.p, cur, (size_t)N * n_blocks * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  s->last_cur_base = cur_base;
  s->last_cur_ld = cur_ld;
  s->last_N = N;
  s->last_blocks = n_blocks;
  s->last_dev_cur = dev_cur;
  (void)hipEventRecord(s->evs, ctx->stream);
  // particle weights start at +0.0: the first block's particle_update_kernel stores instead of accumulating (a sweep that
  // begins with a scoring block accumulates onto zeros)
  const bool w_by_first_block = !ctx->block[0].is_score;
  if (!w_by_first_block) HIPCHK(ctx, hipMemsetAsync(s->w.p, 0, NP * sizeof(double), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(s->logml_acc.p, 0, (size_t)N * sizeof(double), ctx->stream));
  ctx->timing = pclean_timing{};
  bool hot_timed = false;

  for (int bi = 0; bi < n_blocks; ++bi) {
    Block& b = ctx->block[bi];
    BlockRun& r = s->run[bi];
    if (b.is_score) {
      // pure scoring block: every particle's weight += sum of its observed choices' log-densities
      if (bi != n_blocks - 1) return pclean_fail(ctx, PCLEAN_ERR_ARG, "a scoring block must be the last block");
      ProfScope ps(ctx, "score_block");
      ScoreBlockDev sb{};
      auto make_src = [&](int blk, int col, SrcDev& out) -> int {
        if (blk < 0 || blk >= bi || ctx->block[blk].is_score) return pclean_fail(ctx, PCLEAN_ERR_ARG, "score block: bad source block");
        const CandTable& rt = ctx->cand[ctx->block[blk].nodes[0].table];
        if (col < 0 || col >= rt.n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "score block: bad source column");
        out = SrcDev{s->run[blk].pchoice.p, s->run[blk].pnewpos.p, s->run[blk].vals.p, rt.cols.p + (size_t)col * rt.n_rows,
                     (int)ctx->block[blk].nodes.size(), col, s->run[blk].plan};
        return 0;
      };
      const FnTable& pf = ctx->fn[b.prob_fn];
      if (!pf.valid || ctx->n_prob == 0) return pclean_fail(ctx, PCLEAN_ERR_STATE, "score block: prob fn / prob table not set");
      sb.n_terms = (int)b.score_terms.size();
      sb.prob_nb = pf.n_b;
      sb.prob_fn = pf.fn.p;
      sb.prob_same = ctx->prob_same.p;
      sb.prob_diff = ctx->prob_diff.p;
      sb.logn = ctx->logn.p;
      int rc2 = make_src(b.prob_a_block, b.prob_a_col, sb.pa);
      if (!rc2) rc2 = make_src(b.prob_b_block, b.prob_b_col, sb.pb);
      for (int k = 0; k < sb.n_terms && !rc2; ++k) {
        const ScoreTerm& st_ = b.score_terms[k];
        const PairTable& pt = ctx->pair[st_.pair_table];
        const FnTable& nf = ctx->fn[st_.nopt_fn];
        if (!pt.valid || !nf.valid || st_.obs_col < 0 || st_.obs_col >= ctx->n_cols)
          return pclean_fail(ctx, PCLEAN_ERR_ARG, "score block: term %d malformed", k);
        sb.t[k].obs_col = ctx->obs.p + (size_t)st_.obs_col * ctx->n_rows + ctx->active_begin;
        sb.t[k].pair = pt.d.p;
        sb.t[k].n_lat = pt.n_lat;
        sb.t[k].nopt_fn = nf.fn.p;
        sb.t[k].other_val = st_.other_val;
        rc2 = make_src(st_.val_block, st_.val_col, sb.t[k].val);
        if (!rc2) rc2 = make_src(st_.key_block, st_.key_col, sb.t[k].key);
      }
      if (rc2) return rc2;
      hipLaunchKernelGGL(score_block_kernel, grid1(NP), dim3(256), 0, ctx->stream, N, P, sb, s->w.p);
      continue;
    }
    const int nn = (int)b.nodes.size();
    const int32_t* cur_b = cur_base + (size_t)bi * cur_ld;
    if (r.pchoice.alloc(NP) || r.pnewpos.alloc(NP) || r.new_slots.alloc(NP) || r.choice.alloc(N) || r.chosen_newpos.alloc(N) ||
        r.moved_flag.alloc(N) || r.new_flag.alloc(N) || r.moved_list.alloc(N) || r.new_list.alloc(N))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    int rc = ensure_plan_dev(ctx, bi);
    if (rc) return rc;
    const bool has_ctx = b.n_ctx > 0;
    ItemList il;
    const int32_t* excl;
    if (prior_mode) {
      // every (row, particle) draws its referent from the CRP prior; the block's log marginal plays no part
      il = ItemList{N, nullptr, nullptr, nullptr, nullptr};
      excl = cur_b;
      if (r.draws.alloc(NP) || r.lse.alloc(N)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
      ctx->prior_mode = true;
      rc = eval_node(ctx, bi, 0, il, excl, seed, sweep_idx, P, nullptr, r.draws.p, nullptr, nullptr, false);
      if (rc) {
        ctx->prior_mode = false;
        return rc;
      }
      HIPCHK(ctx, hipMemsetAsync(r.lse.p, 0, (size_t)N * sizeof(double), ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
      hipLaunchKernelGGL(particle_update_kernel, dim3((N + PU_T - 1) / PU_T), dim3(PU_T), 0, ctx->stream, N, P, r.draws.p, r.lse.p,
                         (const int32_t*)nullptr, (const int32_t*)nullptr, (const double*)nullptr, cur_b, r.pchoice.p,
                         s->w.p, s->counter.p, r.new_slots.p, r.pnewpos.p, (w_by_first_block && bi == 0) ? 1 : 0);
      if (has_ctx) {  // the particles' contexts: read by the likelihood terms and handed to the new rows' items
        { const int rci = ensure_it_ctx(ctx, r, NP, b.n_ctx); if (rci) return rci; }
        CtxSrc cs{};
        cs.n_ctx = b.n_ctx;
        for (int c = 0; c < b.n_ctx; ++c) {
          const int sb = b.ctx_src_block[c];
          if (sb < 0 || sb >= bi) return pclean_fail(ctx, PCLEAN_ERR_ARG, "block %d: ctx source must be an earlier block", bi);
          const Block& src = ctx->block[sb];
          const CandTable& rt = ctx->cand[src.nodes[0].table];
          if (b.ctx_src_col[c] < 0 || b.ctx_src_col[c] >= rt.n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "ctx column out of range");
          cs.pchoice[c] = s->run[sb].pchoice.p;
          cs.pnewpos[c] = s->run[sb].pnewpos.p;
          cs.vals[c] = s->run[sb].vals.p;
          cs.n_nodes[c] = (int)src.nodes.size();
          cs.root_col[c] = rt.cols.p + (size_t)b.ctx_src_col[c] * rt.n_rows;
          cs.col[c] = b.ctx_src_col[c];
          cs.plan[c] = s->run[sb].plan;
        }
        hipLaunchKernelGGL(gather_ctx_kernel, grid1(NP), dim3(256), 0, ctx->stream, NP, cs, r.it_ctx.p);
      }
    } else if (!has_ctx) {
      il = ItemList{N, nullptr, nullptr, nullptr, nullptr};  // draws row-major [N][P]: one 80-byte store per row
      excl = cur_b;
      if (r.draws.alloc(NP) || r.lse.alloc(N)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
      rc = eval_node(ctx, bi, 0, il, excl, seed, sweep_idx, P, r.lse.p, r.draws.p, nullptr, nullptr, bi == 0);
      if (rc) return rc;
      if (bi == 0) {  // (the elapsed time of the launch is read at the end of the call: no synchronisation here)
        hot_timed = true;
        ctx->timing.hot_kernel_launches += 1;
      }
      ProfScope ps(ctx, "particle_update");
      HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
      hipLaunchKernelGGL(particle_update_kernel, dim3((N + PU_T - 1) / PU_T), dim3(PU_T), 0, ctx->stream, N, P, r.draws.p, r.lse.p,
                         (const int32_t*)nullptr, (const int32_t*)nullptr, (const double*)nullptr, cur_b, r.pchoice.p,
                         s->w.p, s->counter.p, r.new_slots.p, r.pnewpos.p, (w_by_first_block && bi == 0) ? 1 : 0);
    } else {
      { const int rci = ensure_it_ctx(ctx, r, NP, b.n_ctx); if (rci) return rci; }
      CtxSrc cs{};
      cs.n_ctx = b.n_ctx;
      for (int c = 0; c < b.n_ctx; ++c) {
        const int sb = b.ctx_src_block[c];
        if (sb < 0 || sb >= bi) return pclean_fail(ctx, PCLEAN_ERR_ARG, "block %d: ctx source must be an earlier block", bi);
        const Block& src = ctx->block[sb];
        const CandTable& rt = ctx->cand[src.nodes[0].table];
        if (b.ctx_src_col[c] < 0 || b.ctx_src_col[c] >= rt.n_cols) return pclean_fail(ctx, PCLEAN_ERR_ARG, "ctx column out of range");
        cs.pchoice[c] = s->run[sb].pchoice.p;
        cs.pnewpos[c] = s->run[sb].pnewpos.p;
        cs.vals[c] = s->run[sb].vals.p;
        cs.n_nodes[c] = (int)src.nodes.size();
        cs.root_col[c] = rt.cols.p + (size_t)b.ctx_src_col[c] * rt.n_rows;
        cs.col[c] = b.ctx_src_col[c];
        cs.plan[c] = s->run[sb].plan;
      }
      // One enumeration per distinct (row, context): particles whose earlier choices give the same
      // context share the candidate scores and differ only in their Philox draws.
      int32_t* rep = scratch<int32_t>(ctx, NP);
      int32_t* slot_item = scratch<int32_t>(ctx, NP);
      int32_t* n_distinct = scratch<int32_t>(ctx, (size_t)N + 1);
      int32_t* off = scratch<int32_t>(ctx, (size_t)N + 1);
      size_t tmp_scan = 0;
      HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, n_distinct, off, N + 1, ctx->stream));
      unsigned char* tmp = scratch<unsigned char>(ctx, tmp_scan);
      if (!rep || !slot_item || !n_distinct || !off || !tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      unsigned int n_items = 0;
      {
        ProfScope ps(ctx, "ctx_items");
        hipLaunchKernelGGL(gather_ctx_kernel, grid1(NP), dim3(256), 0, ctx->stream, NP, cs, r.it_ctx.p);
        HIPCHK(ctx, hipMemsetAsync(n_distinct + N, 0, sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(ctx_count_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, (int)b.n_ctx, r.it_ctx.p, rep, n_distinct);
        HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(tmp, tmp_scan, n_distinct, off, N + 1, ctx->stream));
        PCLEAN_READ_COUNT(ctx, off + N, &n_items);
      }
      int32_t* d_row = scratch<int32_t>(ctx, n_items);
      int32_t* d_ctx = scratch<int32_t>(ctx, (size_t)n_items * PCLEAN_MAX_CTX);
      int32_t* d_excl = scratch<int32_t>(ctx, n_items);
      double* lse_item = scratch<double>(ctx, n_items);
      int32_t* draws_item = scratch<int32_t>(ctx, (size_t)n_items * P);
      if (!d_row || !d_ctx || !d_excl || !lse_item || !draws_item)
        return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      hipLaunchKernelGGL(ctx_fill_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, (int)b.n_ctx, r.it_ctx.p, rep, off, cur_b, slot_item,
                         d_row, d_ctx, d_excl);
      il = ItemList{(int)n_items, d_row, d_ctx, nullptr, nullptr};
      rc = eval_node(ctx, bi, 0, il, d_excl, seed, sweep_idx, P, lse_item, draws_item, nullptr, nullptr, false);
      if (rc) return rc;
      ProfScope ps(ctx, "particle_update");
      HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
      hipLaunchKernelGGL(particle_update_kernel, dim3((N + PU_T - 1) / PU_T), dim3(PU_T), 0, ctx->stream, N, P, (const int32_t*)nullptr,
                         (const double*)nullptr, slot_item, draws_item, lse_item, cur_b, r.pchoice.p, s->w.p,
                         s->counter.p, r.new_slots.p, r.pnewpos.p, (w_by_first_block && bi == 0) ? 1 : 0);
    }
    // ---- particles that proposed a NEW referent: sample the new row's contents
    unsigned int n_new = 0;
    PCLEAN_READ_COUNT(ctx, s->counter.p, &n_new);
    r.n_new = (int)n_new;
    // The contents of a proposed new row matter (a) as context / scored values of LATER blocks — every particle's —
    // and (b) for the particle that is finally chosen.  For the last block only (b) is left: its sampling is
    // deferred until after the final choice and done for the chosen particles alone (same Philox counters, so
    // the values are the ones eager sampling would have produced; typically 20x fewer items).
    static const bool eager_all = getenv("PCLEAN_EAGER_NEW") != nullptr;
    // A chosen ProposalDummyValue changes its particle's weight (apply_dummy_corrections): where one can be drawn
    // every NEW slot is sampled before the final choice.
    bool drawable = prior_mode;  // (prior draws of a StringPrior choice are the dummy almost surely)
    if (!prior_mode) {
      rc = block_dummy_drawable(ctx, bi, &drawable);
      if (rc) return rc;
    }
    r.lazy_new = bi == n_blocks - 1 && !eager_all && !drawable;
    if (r.lazy_new) n_new = 0;  // nothing sampled now
    if (r.vals.alloc(std::max<size_t>((size_t)n_new * nn, 1))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    if (n_new) {
      ProfScope ps(ctx, "new_row_sampling");
      const int32_t* list = r.new_slots.p;
      hipLaunchKernelGGL(fill_i32_kernel, grid1((size_t)n_new * nn), dim3(256), 0, ctx->stream, r.vals.p,
                         (size_t)n_new * nn, -2);
      int32_t* row = scratch<int32_t>(ctx, n_new);
      int32_t* cx = scratch<int32_t>(ctx, (size_t)n_new * PCLEAN_MAX_CTX);
      int32_t* part = scratch<int32_t>(ctx, n_new);
      int32_t* org = scratch<int32_t>(ctx, n_new);
      int32_t* ex = scratch<int32_t>(ctx, n_new);
      if (!row || !cx || !part || !org || !ex) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      hipLaunchKernelGGL(rootlist_items_kernel, grid1(n_new), dim3(256), 0, ctx->stream, (int)n_new, N, NP, list,
                         has_ctx ? r.it_ctx.p : nullptr, cur_b, row, cx, part, org, ex);
      hipLaunchKernelGGL(set_col_kernel, grid1(n_new), dim3(256), 0, ctx->stream, (int)n_new, nn, 0,
                         (int32_t)PCLEAN_CHOICE_NEW, r.vals.p);
      ItemList sub{(int)n_new, row, cx, part, org};
      rc = sample_children(ctx, bi, 0, sub, ex, seed, sweep_idx, r.vals.p, nn);
      if (rc) return rc;
      if (prior_mode) {
        ctx->prior_mode = false;
        const NodeDev* nds;
        const int32_t *dnc, *dcb, *dch;
        rc = upload_plan_nodes(ctx, bi, &nds, &dnc, &dcb, &dch);
        if (!rc)
          rc = pclean_launch_prior_terms(ctx, NP, N, nn, nds, dnc, dcb, dch, r.pchoice.p, r.pnewpos.p, r.vals.p,
                                         has_ctx ? r.it_ctx.p : nullptr, s->w.p);
        if (rc) return rc;
      }
      if (drawable) {
        rc = apply_dummy_corrections(ctx, bi, r.new_slots.p, r.vals.p, (int)n_new, N, seed, sweep_idx, s->w.p);
        if (rc) return rc;
      }
    } else if (prior_mode) {  // nobody proposed a new referent: the likelihood of the chosen referents alone
      ctx->prior_mode = false;
      const NodeDev* nds;
      const int32_t *dnc, *dcb, *dch;
      rc = upload_plan_nodes(ctx, bi, &nds, &dnc, &dcb, &dch);
      if (!rc)
        rc = pclean_launch_prior_terms(ctx, NP, N, nn, nds, dnc, dcb, dch, r.pchoice.p, r.pnewpos.p, r.vals.p,
                                       has_ctx ? r.it_ctx.p : nullptr, s->w.p);
      if (rc) return rc;
    }
    ctx->prior_mode = false;
    // (pnewpos is only read where pchoice == NEW, so it needs no initialisation when nobody proposed one)

    // ---- resampling between blocks (row_inference.jl:152-155)
    const int grp_here = b.group >= 0 ? b.group : bi;
    const int grp_next = bi + 1 < n_blocks ? (ctx->block[bi + 1].group >= 0 ? ctx->block[bi + 1].group : bi + 1) : -2;
    if (!use_mh && bi < n_blocks - 1 && grp_here != grp_next) {  // (the slots of one model block: no resampling in between)
      ProfScope ps(ctx, "resample");
      DISPATCH_PMAX(P, hipLaunchKernelGGL(maybe_resample_kernel<PMAX>, grid1(N), dim3(256), 0, ctx->stream, N, P, s->w.p,
                                          (size_t)1, (size_t)N, 1, cur_b, seed, sweep_idx, (uint32_t)bi,
                                          s->row_offset + ctx->active_begin, s->ancestors.p, s->logml_inc.p,
                                          (double*)nullptr, s->did.p));
      AncestorArrays arrs{};
      int n_arr = 0;
      for (int k = 0; k <= bi; ++k) {
        if (ctx->block[k].is_score) continue;
        arrs.p[n_arr++] = s->run[k].pchoice.p;
        arrs.p[n_arr++] = s->run[k].pnewpos.p;
      }
      hipLaunchKernelGGL(apply_ancestors_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, s->ancestors.p, n_arr, arrs, s->w.p,
                         s->did.p, s->logml_inc.p, s->logml_acc.p);
    }
  }

  // ---- final choice + per-block outputs: one pass per block, ordered compaction of the rows that moved /
  // got a new referent (hipcub select keeps ascending row order), ONE read-back of the counts
  {
    ProfScope ps(ctx, "final_choice_and_outputs");
    DISPATCH_PMAX(P, hipLaunchKernelGGL(final_choice_kernel<PMAX>, grid1(N), dim3(256), 0, ctx->stream, N, P, s->w.p,
                                        (size_t)1, (size_t)N, use_mh, 1, cur_base, seed, sweep_idx,
                                        s->row_offset + ctx->active_begin, s->chosen.p, (double*)nullptr,
                                        s->logml_acc.p, s->logml.p));
    for (int bi = 0; bi < n_blocks; ++bi) {  // deferred new-row contents of the last block (chosen particles only)
      BlockRun& r = s->run[bi];
      Block& bb = ctx->block[bi];
      if (bb.is_score || !r.lazy_new || r.n_new == 0) continue;
      ProfScope ps2(ctx, "new_row_sampling_chosen");
      const int nn = (int)bb.nodes.size();
      const int32_t* cur_b = cur_base + (size_t)bi * cur_ld;
      HIPCHK(ctx, hipMemsetAsync(s->counter.p, 0, sizeof(unsigned int), ctx->stream));
      hipLaunchKernelGGL(chosen_new_kernel, grid1(N), dim3(256), 0, ctx->stream, N, s->chosen.p, r.pchoice.p, s->counter.p,
                         r.new_slots.p, r.pnewpos.p);
      unsigned int cnt = 0;
      PCLEAN_READ_COUNT(ctx, s->counter.p, &cnt);
      if (r.vals.alloc(std::max<size_t>((size_t)cnt * nn, 1))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
      if (!cnt) continue;
      hipLaunchKernelGGL(fill_i32_kernel, grid1((size_t)cnt * nn), dim3(256), 0, ctx->stream, r.vals.p, (size_t)cnt * nn, -2);
      int32_t* row = scratch<int32_t>(ctx, cnt);
      int32_t* cx = scratch<int32_t>(ctx, (size_t)cnt * PCLEAN_MAX_CTX);
      int32_t* part = scratch<int32_t>(ctx, cnt);
      int32_t* org = scratch<int32_t>(ctx, cnt);
      int32_t* ex = scratch<int32_t>(ctx, cnt);
      if (!row || !cx || !part || !org || !ex) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      hipLaunchKernelGGL(rootlist_items_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, N, NP, r.new_slots.p,
                         bb.n_ctx > 0 ? r.it_ctx.p : nullptr, cur_b, row, cx, part, org, ex);
      hipLaunchKernelGGL(set_col_kernel, grid1(cnt), dim3(256), 0, ctx->stream, (int)cnt, nn, 0, (int32_t)PCLEAN_CHOICE_NEW,
                         r.vals.p);
      ItemList sub{(int)cnt, row, cx, part, org};
      int rc = sample_children(ctx, bi, 0, sub, ex, seed, sweep_idx, r.vals.p, nn);
      if (rc) return rc;
    }
    size_t tmp_sel = 0;
    HIPCHK(ctx, hipcub::DeviceSelect::Flagged(nullptr, tmp_sel, hipcub::CountingInputIterator<int32_t>(0),
                                              (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, N,
                                              ctx->stream));
    unsigned char* tmp = scratch<unsigned char>(ctx, std::max<size_t>(tmp_sel, 16));
    if (!tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    for (int bi = 0; bi < n_blocks; ++bi) {
      BlockRun& r = s->run[bi];
      Block& bb = ctx->block[bi];
      bb.locals_host.clear();
      r.locals_rows = 0;
      if (bb.is_score) continue;
      const int32_t* cur_b = cur_base + (size_t)bi * cur_ld;
      CandTable& rt = ctx->cand[bb.nodes[0].table];
      HIPCHK(ctx, hipMemsetAsync(rt.stats.p, 0, (size_t)std::max(rt.n_rows, 1) * 8, ctx->stream));
      const int hist_rows = rt.n_rows <= 12288 ? rt.n_rows : 0;  // (48 KB of LDS per workgroup at most)
      hipLaunchKernelGGL(finalize_block_kernel, grid1(N), dim3(256), (size_t)hist_rows * sizeof(int32_t), ctx->stream, N,
                         s->chosen.p, r.pchoice.p, r.pnewpos.p, cur_b, r.choice.p, r.chosen_newpos.p,
                         (unsigned long long*)rt.stats.p, hist_rows, r.moved_flag.p, r.new_flag.p);
      HIPCHK(ctx, hipcub::DeviceSelect::Flagged(tmp, tmp_sel, hipcub::CountingInputIterator<int32_t>(0), r.moved_flag.p,
                                                r.moved_list.p, s->tail_counts.p + 2 * bi, N, ctx->stream));
      HIPCHK(ctx, hipcub::DeviceSelect::Flagged(tmp, tmp_sel, hipcub::CountingInputIterator<int32_t>(0), r.new_flag.p,
                                                r.new_list.p, s->tail_counts.p + 2 * bi + 1, N, ctx->stream));
      if (choice)
        HIPCHK(ctx, hipMemcpyAsync(choice + (size_t)bi * N, r.choice.p, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
      if (!bb.node_gauss.empty() && bb.node_gauss[0] >= 0 && bb.gauss[bb.node_gauss[0]].n_locals > 0) {
        GaussDev gd;
        int rc = build_gauss_dev(ctx, bb.gauss[bb.node_gauss[0]], &rt, gd);
        if (rc) return rc;
        if (r.locals.alloc((size_t)N * 2)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
        hipLaunchKernelGGL(locals_tail_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, gd, r.plan, s->chosen.p,
                           r.pchoice.p, r.pnewpos.p, r.vals.p, (int)bb.nodes.size(), seed, sweep_idx, (uint32_t)bi,
                           s->row_offset + ctx->active_begin, r.locals.p);
        r.locals_rows = N;
        if (!defer) {  // (deferred outputs: pclean_get_locals copies them when asked)
          bb.locals_host.resize((size_t)N * 2);
          HIPCHK(ctx, hipMemcpyAsync(bb.locals_host.data(), r.locals.p, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
      }
    }
    (void)hipEventRecord(s->eve, ctx->stream);
    if (chosen_particle) HIPCHK(ctx, hipMemcpyAsync(chosen_particle, s->chosen.p, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (logml) HIPCHK(ctx, hipMemcpyAsync(logml, s->logml.p, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  s->last_hot_timed = hot_timed;
  s->outputs_pending = true;
  s->lists_on_host = false;
  {
    // bytes(row) = sum_b [4 F_b + (K_b+1)(8 F_b + 4)] + 8 P (full enumeration); reported beside the
    // byte model of the implemented algorithm (bench.py)
    const Block& b0 = ctx->block[0];
    const double F = b0.nodes[0].n_terms, K = ctx->cand[b0.nodes[0].table].n_rows;
    ctx->timing.hot_kernel_alg_bytes = (double)N * (4.0 * F + (K + 1.0) * (8.0 * F + 4.0) + 8.0 * P);
  }
  if (defer) return PCLEAN_OK;  // nothing read back, no synchronisation: pclean_commit_device / pclean_sweep_fetch finish the call
  int rcf = pclean_sweep_finish_queue(ctx);
  if (rcf) return rcf;
  PCLEAN_SYNC(ctx);
  rcf = pclean_sweep_finish_synced(ctx);
  if (rcf) return rcf;
  if (choice)
    for (int bi = 0; bi < n_blocks; ++bi)
      if (ctx->block[bi].is_score)
        for (int i = 0; i < N; ++i) choice[(size_t)bi * N + i] = 0;
  return pclean_sweep_fetch_lists(ctx);
}

// ---- end of a sweep, in three steps so that a caller with more work for the stream (pclean_commit_device) pays ONE
// synchronisation for everything -----------------------------------------------------------------------------------
// 1. queue the small read-backs: moved / new-row counts, overflow statistics, the dummy-arena flag
int pclean_sweep_finish_queue(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  if (!s->outputs_pending) return PCLEAN_OK;
  HIPCHK(ctx, hipMemcpyAsync(s->h_counts, s->tail_counts.p, 2 * s->last_blocks * sizeof(int32_t), hipMemcpyDeviceToHost,
                             ctx->stream));
  const int rco = queue_over_copy(ctx);
  if (rco) return rco;
  s->h_counts[3 * PCLEAN_MAX_BLOCKS] = 0;
  if (s->dummy_used)
    HIPCHK(ctx, hipMemcpyAsync(s->h_counts + 3 * PCLEAN_MAX_BLOCKS, s->dummy_ctr.p + 1, sizeof(int32_t), hipMemcpyDeviceToHost,
                               ctx->stream));
  return PCLEAN_OK;
}
// 2. after the caller's stream synchronisation: statistics, timing, error flags
int pclean_sweep_finish_synced(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  if (!s->outputs_pending) return PCLEAN_OK;
  s->outputs_pending = false;
  apply_over_stats(ctx);
  if (s->prof_on) prof_collect(ctx);
  float tot = 0;
  HIPCHK(ctx, hipEventElapsedTime(&tot, s->evs, s->eve));
  ctx->timing.total_ms = tot;
  ctx->timing.hot_kernel_ms = 0.f;
  if (s->last_hot_timed) {
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, s->ev0, s->ev1));
    ctx->timing.hot_kernel_ms = ms;
  }
  if (s->h_counts[3 * PCLEAN_MAX_BLOCKS])
    return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "pclean_sweep: an observed string longer than %d symbols below a chosen "
                                                 "dummy value", DUMMY_MAX_LEN);
  return PCLEAN_OK;
}
// 3. rows whose referent changed and new-row records of the chosen particles -> host (ascending rows); one more
// synchronisation.  The host commit's input; the device-resident commit never needs it.
int pclean_sweep_fetch_lists(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  if (s->lists_on_host) return PCLEAN_OK;
  for (int bi = 0; bi < s->last_blocks; ++bi) {
    BlockRun& r = s->run[bi];
    Block& b = ctx->block[bi];
    const int nn = (int)b.nodes.size();
    b.new_rows_host.clear();
    b.new_vals_host.clear();
    b.moved_rows_host.clear();
    b.moved_choice_host.clear();
    if (b.is_score) continue;
    const int n_moved = s->h_counts[2 * bi], n_newrows = s->h_counts[2 * bi + 1];
    if (n_moved > 0) {
      int32_t* ch_d = scratch<int32_t>(ctx, n_moved);
      if (!ch_d) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      hipLaunchKernelGGL(gather_moved_kernel, grid1(n_moved), dim3(256), 0, ctx->stream, n_moved, r.moved_list.p,
                         r.choice.p, ch_d);
      b.moved_rows_host.resize(n_moved);
      b.moved_choice_host.resize(n_moved);
      HIPCHK(ctx, hipMemcpyAsync(b.moved_rows_host.data(), r.moved_list.p, (size_t)n_moved * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(b.moved_choice_host.data(), ch_d, (size_t)n_moved * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n_newrows > 0) {
      int32_t* rows_d = scratch<int32_t>(ctx, n_newrows);
      int32_t* vals_d = scratch<int32_t>(ctx, (size_t)n_newrows * nn);
      if (!rows_d || !vals_d) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      hipLaunchKernelGGL(gather_new_rows_kernel, grid1(n_newrows), dim3(256), 0, ctx->stream, n_newrows, r.new_list.p,
                         r.chosen_newpos.p, r.vals.p, nn, s->chosen.p, rows_d, vals_d);
      b.new_rows_host.resize(n_newrows);
      b.new_vals_host.resize((size_t)n_newrows * nn);
      HIPCHK(ctx, hipMemcpyAsync(b.new_rows_host.data(), rows_d, (size_t)n_newrows * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(b.new_vals_host.data(), vals_d, (size_t)n_newrows * nn * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  PCLEAN_SYNC(ctx);
  s->lists_on_host = true;
  return PCLEAN_OK;
}

// Finish a sweep run with deferred outputs (pclean_set_sweep_mode) the way a plain pclean_sweep call ends: counts,
// statistics and the moved-row / new-row lists on the host (pclean_get_moved / pclean_get_new_rows / pclean_get_stats).
extern "C" int pclean_sweep_fetch(pclean_ctx* ctx) {
  if (!ctx) return PCLEAN_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = pclean_sweep_finish_queue(ctx);
  if (rc) return rc;
  PCLEAN_SYNC(ctx);
  rc = pclean_sweep_finish_synced(ctx);
  if (rc) return rc;
  return pclean_sweep_fetch_lists(ctx);
}

// bit 0: deferred outputs — pclean_sweep reads nothing back and does not synchronise; pclean_commit_device (or
// pclean_sweep_fetch) finishes the call.
extern "C" int pclean_set_sweep_mode(pclean_ctx* ctx, int32_t flags) {
  if (!ctx || flags < 0 || flags > 1) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_sweep_mode: bad flags");
  ctx->defer_outputs = (flags & 1) != 0;
  return PCLEAN_OK;
}

extern "C" int pclean_get_new_rows(pclean_ctx* ctx, int32_t block_id, int32_t* n_out, int32_t* rows_out,
                                   int32_t* vals_out) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || !n_out)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_get_new_rows: bad arguments");
  const Block& b = ctx->block[block_id];
  *n_out = (int32_t)b.new_rows_host.size();
  if (rows_out && !b.new_rows_host.empty()) memcpy(rows_out, b.new_rows_host.data(), b.new_rows_host.size() * 4);
  if (vals_out && !b.new_vals_host.empty()) memcpy(vals_out, b.new_vals_host.data(), b.new_vals_host.size() * 4);
  return PCLEAN_OK;
}

extern "C" int pclean_get_moved(pclean_ctx* ctx, int32_t block_id, int32_t* n_out, int32_t* rows_out,
                                int32_t* choice_out) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || !n_out)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_get_moved: bad arguments");
  const Block& b = ctx->block[block_id];
  *n_out = (int32_t)b.moved_rows_host.size();
  if (rows_out && !b.moved_rows_host.empty()) memcpy(rows_out, b.moved_rows_host.data(), b.moved_rows_host.size() * 4);
  if (choice_out && !b.moved_choice_host.empty())
    memcpy(choice_out, b.moved_choice_host.data(), b.moved_choice_host.size() * 4);
  return PCLEAN_OK;
}

extern "C" int pclean_get_locals(pclean_ctx* ctx, int32_t block_id, int32_t* out) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid || !out)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_get_locals: bad arguments");
  Block& b = ctx->block[block_id];
  const int N = ctx->active_count >= 0 ? ctx->active_count : ctx->n_rows;
  {
    BlockRun& r = st(ctx)->run[block_id];
    if (b.locals_host.empty() && r.locals_rows == N && r.locals.p) {  // a sweep with deferred outputs left them on the device
      HIPCHK(ctx, hipSetDevice(ctx->device));
      b.locals_host.resize((size_t)N * 2);
      HIPCHK(ctx, hipMemcpyAsync(b.locals_host.data(), r.locals.p, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
      PCLEAN_SYNC(ctx);
    }
  }
  if (b.locals_host.size() == (size_t)N * 2)
    memcpy(out, b.locals_host.data(), (size_t)N * 8);
  else
    for (int i = 0; i < 2 * N; ++i) out[i] = -1;
  return PCLEAN_OK;
}

extern "C" int pclean_stats_device_ptr(pclean_ctx* ctx, int32_t table_id, void** dptr, int64_t* n) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || !ctx->cand[table_id].valid || !dptr || !n)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_stats_device_ptr: bad arguments");
  *dptr = ctx->cand[table_id].stats.p;
  *n = ctx->cand[table_id].n_rows;
  return PCLEAN_OK;
}

extern "C" int pclean_get_stats(pclean_ctx* ctx, int32_t table_id, int64_t* out) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || !ctx->cand[table_id].valid || !out)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_get_stats: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const CandTable& t = ctx->cand[table_id];
  if (t.n_rows) HIPCHK(ctx, hipMemcpy(out, t.stats.p, (size_t)t.n_rows * 8, hipMemcpyDeviceToHost));
  return PCLEAN_OK;
}

extern "C" int pclean_set_cur_stride(pclean_ctx* ctx, int64_t stride) {
  if (!ctx || stride < 0) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_cur_stride: bad stride");
  ctx->cur_stride = stride;
  return PCLEAN_OK;
}

extern "C" int pclean_get_root_stats(pclean_ctx* ctx, pclean_root_stats* out) {
  if (!ctx || !out) return PCLEAN_ERR_ARG;
  *out = ctx->root_stats;
  return PCLEAN_OK;
}

extern "C" int pclean_get_timing(pclean_ctx* ctx, pclean_timing* out) {
  if (!ctx || !out) return PCLEAN_ERR_ARG;
  *out = ctx->timing;
  return PCLEAN_OK;
}

// ---- particle primitives exposed for parity ------------------------------------
extern "C" int pclean_maybe_resample(pclean_ctx* ctx, int32_t n_rows, int32_t n_particles, const double* logw,
                                     int32_t retain_first, uint64_t seed, uint32_t sweep, uint32_t block,
                                     int32_t* ancestors, double* logml_inc, double* ess) {
  if (!ctx || n_rows <= 0 || n_particles <= 0 || n_particles > MAXP || !logw || !ancestors || !logml_inc)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_maybe_resample: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  SweepState* s = st(ctx);
  s->pool_used = 0;
  s->dbg_desc = nullptr;
  const size_t NP = (size_t)n_rows * n_particles;
  double* d_w = scratch<double>(ctx, NP);
  int32_t* d_a = scratch<int32_t>(ctx, NP);
  double* d_inc = scratch<double>(ctx, n_rows);
  double* d_ess = scratch<double>(ctx, n_rows);
  if (!d_w || !d_a || !d_inc || !d_ess) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  HIPCHK(ctx, hipMemcpyAsync(d_w, logw, NP * 8, hipMemcpyHostToDevice, ctx->stream));
  DISPATCH_PMAX(n_particles, hipLaunchKernelGGL(maybe_resample_kernel<PMAX>, grid1(n_rows), dim3(256), 0, ctx->stream,
                                                n_rows, n_particles, d_w, (size_t)n_particles, (size_t)1, retain_first,
                                                (const int32_t*)nullptr, seed, sweep, block, s->row_offset, d_a, d_inc,
                                                d_ess, (int32_t*)nullptr));
  HIPCHK(ctx, hipMemcpyAsync(ancestors, d_a, NP * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(logml_inc, d_inc, (size_t)n_rows * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (ess) HIPCHK(ctx, hipMemcpyAsync(ess, d_ess, (size_t)n_rows * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCLEAN_SYNC(ctx);
  return PCLEAN_OK;
}

extern "C" int pclean_final_choice(pclean_ctx* ctx, int32_t n_rows, int32_t n_particles, const double* logw,
                                   int32_t use_mh, int32_t is_csmc, uint64_t seed, uint32_t sweep, int32_t* chosen,
                                   double* log_total) {
  if (!ctx || n_rows <= 0 || n_particles <= 0 || n_particles > MAXP || !logw || !chosen)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_final_choice: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  SweepState* s = st(ctx);
  s->pool_used = 0;
  s->dbg_desc = nullptr;
  const size_t NP = (size_t)n_rows * n_particles;
  double* d_w = scratch<double>(ctx, NP);
  int32_t* d_c = scratch<int32_t>(ctx, n_rows);
  double* d_t = scratch<double>(ctx, n_rows);
  if (!d_w || !d_c || !d_t) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
  HIPCHK(ctx, hipMemcpyAsync(d_w, logw, NP * 8, hipMemcpyHostToDevice, ctx->stream));
  DISPATCH_PMAX(n_particles, hipLaunchKernelGGL(final_choice_kernel<PMAX>, grid1(n_rows), dim3(256), 0, ctx->stream,
                                                n_rows, n_particles, d_w, (size_t)n_particles, (size_t)1, use_mh, is_csmc,
                                                (const int32_t*)nullptr, seed, sweep, s->row_offset, d_c, d_t,
                                                (const double*)nullptr, (double*)nullptr));
  HIPCHK(ctx, hipMemcpyAsync(chosen, d_c, (size_t)n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (log_total) HIPCHK(ctx, hipMemcpyAsync(log_total, d_t, (size_t)n_rows * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCLEAN_SYNC(ctx);
  return PCLEAN_OK;
}

// ---- numeric-contract probes -------------------------------------------------
__global__ void debug_detmath_kernel(int n, const double* x, double* e, double* l, uint64_t* f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  e[i] = pclean_exp(x[i]);
  l[i] = pclean_log(x[i]);
  f[i] = pclean_fixw(x[i]);
}
__global__ void debug_rand64_kernel(int n, uint64_t seed, const uint32_t* rows, uint32_t site, uint32_t particle,
                                    uint32_t sweep, uint64_t* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pclean_rand64(seed, rows[i], site, particle, sweep);
}

extern "C" int pclean_debug_detmath(pclean_ctx* ctx, int32_t n, const double* x, double* exp_out, double* log_out,
                                    uint64_t* fixw_out) {
  if (!ctx || n <= 0 || !x || !exp_out || !log_out || !fixw_out) return pclean_fail(ctx, PCLEAN_ERR_ARG, "bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<double> dx, de, dl;
  DevBuf<uint64_t> df;
  if (dx.alloc(n) || de.alloc(n) || dl.alloc(n) || df.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "alloc");
  hipError_t e = hipMemcpy(dx.p, x, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(debug_detmath_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, dx.p, de.p, dl.p, df.p);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpy(exp_out, de.p, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(log_out, dl.p, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(fixw_out, df.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost);
  dx.release(); de.release(); dl.release(); df.release();
  if (e != hipSuccess) return pclean_fail(ctx, PCLEAN_ERR_HIP, "debug_detmath: %s", hipGetErrorString(e));
  return PCLEAN_OK;
}

extern "C" int pclean_debug_rand64(pclean_ctx* ctx, int32_t n, uint64_t seed, const uint32_t* rows, uint32_t site,
                                   uint32_t particle, uint32_t sweep, uint64_t* out) {
  if (!ctx || n <= 0 || !rows || !out) return pclean_fail(ctx, PCLEAN_ERR_ARG, "bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf<uint32_t> dr;
  DevBuf<uint64_t> d_out;
  if (dr.alloc(n) || d_out.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "alloc");
  hipError_t e = hipMemcpy(dr.p, rows, n * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(debug_rand64_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, seed, dr.p, site,
                       particle, sweep, d_out.p);
    e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost);
  dr.release(); d_out.release();
  if (e != hipSuccess) return pclean_fail(ctx, PCLEAN_ERR_HIP, "debug_rand64: %s", hipGetErrorString(e));
  return PCLEAN_OK;
}
