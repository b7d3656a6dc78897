// Shared by the translation units of the sweep orchestration (internal):
//   sweep.hip   the observed-class sweep (pclean_sweep), its particle kernels, the call services (scratch pool, count
//               read-backs, per-phase profile) and the small entry points;
//   eval.hip    evaluation of one plan node for a list of items: option lists, reference slots with their children, the
//               compact-table fast path, item grouping, memoised marginals, sampling of a new row's contents;
//   latent.hip  the latent-class sweep (pclean_sweep_latent), its evidence aggregation and pclean_score_node_ev.
// Small kernels used by more than one of them are `static` here (one copy per translation unit).
#pragma once
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
extern int g_pclean_sync_count;
int read_count(pclean_ctx* ctx, const void* dev, void* out, const char* func, int line);  // (sweep.hip)
#define PCLEAN_READ_COUNT(ctx, dev, out)                                  \
  do {                                                                    \
    const int rc_ = read_count(ctx, (dev), (out), __func__, __LINE__);    \
    if (rc_) return rc_;                                                  \
  } while (0)

#define PCLEAN_SYNC(ctx)                                                                                   \
  do {                                                                                                     \
    static const bool trace_ = getenv("PCLEAN_TRACE_SYNC") != nullptr;                                     \
    if (trace_) fprintf(stderr, "[pclean sync %d] %s:%d\n", ++g_pclean_sync_count, __func__, __LINE__);          \
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                                                        \
  } while (0)

// ---- host-side types -----------------------------------------------------------------------------------------------
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
struct ItemGroups {
  int n_groups = 0;              // 0: grouping not applicable / not worth it
  const int32_t* grp_off = nullptr;  // [n_groups + 1] into members
  const int32_t* members = nullptr;  // [n] item ids, groups contiguous
  const int32_t* head = nullptr;     // [n] 1 at the first member of each group (sorted order)
  const int32_t* uid = nullptr;      // [n] inclusive scan of head
  // hash-table grouping (eval.hip: make_item_groups_hash): inclusive scan over the slots, the slots' representatives, slot of
  // every item — all a caller needs that only asks which group an item is in (hg_unique_kernel)
  const uint64_t* hg_incl = nullptr;
  const int32_t* hg_rep = nullptr;
  const int32_t* hg_slot_of = nullptr;
};

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

// A zeroed device counter for one use (how many items need the next step): begin_call zeroes a bank of them with the
// memset it makes anyway — a hipMemsetAsync of four bytes per use was a dispatch per use.  Null-safe: when the bank is
// exhausted the last counter is re-zeroed the old way.
static unsigned int* fresh_counter(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  if (!s->over_ctr.p) return nullptr;
  unsigned int* bank = s->over_ctr.p + OVER_SLOTS + STAT_WORDS;
  // (PCLEAN_CTR_BANK: a smaller first bank, so that tests reach the growth path below)
  static const int first = [] {
    const char* e = getenv("PCLEAN_CTR_BANK");
    return e ? std::max(1, std::min(atoi(e), CTR_BANK)) : CTR_BANK;
  }();
  if (s->bank_used < first) return bank + s->bank_used++;
  // more than CTR_BANK counted launches in one call (hundreds of nested option lists x re-runs): further banks, allocated
  // once and zeroed at their first use in a call; a counter is never handed out twice within a call
  const int idx = s->bank_used - first, b = idx / CTR_BANK, o = idx % CTR_BANK;
  while ((int)s->more_banks.size() <= b) s->more_banks.emplace_back();
  if (s->more_banks[b].alloc(CTR_BANK)) return nullptr;  // (callers report the allocation failure)
  if (o == 0 && hipMemsetAsync(s->more_banks[b].p, 0, CTR_BANK * sizeof(unsigned int), ctx->stream) != hipSuccess) return nullptr;
  ++s->bank_used;
  return s->more_banks[b].p + o;
}

// ---- per-phase profile (pclean_set_profiling): HIP events on the library's stream around groups of launches
int prof_phase_id(SweepState* s, const char* name);
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

// ---- small kernels -------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// small kernels
static __global__ void iota_missing_kernel(int32_t* p, int n_obs) {  // [0..n_obs-1, -1]
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_obs) p[i] = i < n_obs ? i : -1;
}

static __global__ void fill_i32_kernel(int32_t* p, size_t n, int32_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

static __global__ void fill_f64_kernel(double* p, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// excl_child[t] = row of the child's table that loses a reference because the
// parent's excluded row is garbage-collected (dependency_tracking.jl:189-201)
static __global__ void derive_excl_kernel(int n, const int32_t* parent_excl, const int64_t* parent_counts,
                                   const int32_t* parent_fk_col, int32_t* out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int e = parent_excl ? parent_excl[t] : -1;
  out[t] = (e >= 0 && parent_counts[e] <= 1) ? parent_fk_col[e] : -1;
}

// compaction of NEW choices: pass 0 counts, pass 1 fills
// (block-aggregated: one global atomic per 256 elements instead of one per hit)
static __global__ __launch_bounds__(256) void compact_new_kernel(size_t n, const int32_t* choice, int fill,
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
static __global__ void sublist_items_kernel(int n, const int32_t* list, const int32_t* p_row, const int32_t* p_ctx,
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

static __global__ void scatter_vals_kernel(int n, const int32_t* origin, const int32_t* draws, int n_nodes, int node,
                                    int32_t* vals) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) vals[(size_t)origin[j] * n_nodes + node] = draws[j];
}

static __global__ void gather_i32_kernel(int n, const int32_t* list, const int32_t* src, int32_t* dst) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = src[list[j]];
}

// ---- functions defined in one translation unit and called from another ------------------------------------------------
int begin_call(pclean_ctx* ctx);
int queue_over_copy(pclean_ctx* ctx);
// Small device -> page-locked-host read-backs that ride on ONE synchronisation: queued, then written by ONE kernel through the
// host buffers' device mappings (d2h_flush) instead of one blit dispatch each (six 5-us copies back to back after a commit).
// A buffer that is not page-locked / mapped, an odd size or a full queue: hipMemcpyAsync right away.
// (declared in sweep_state.h: d2h_small(ctx, host, dev, bytes, host_base = the allocation `host` lies in), d2h_flush(ctx))
void apply_over_stats(pclean_ctx* ctx);
int finish_call(pclean_ctx* ctx);
void prof_collect(pclean_ctx* ctx);
int prior_mode_supported(pclean_ctx* ctx, const Block& b, const char* who);
int upload_plan_nodes(pclean_ctx* ctx, int bi, const NodeDev** nds, const int32_t** n_children,
                             const int32_t** child_begin, const int32_t** children);
int build_gauss_dev(pclean_ctx* ctx, const pclean_gauss& g, const CandTable* t, GaussDev& d);
int build_node_dev(pclean_ctx* ctx, const Block& b, int node_id, NodeDev& nd);
int ensure_leaf_cache(pclean_ctx* ctx, int block_id, int node_id, const double** out, const int32_t** obs_col,
                             int* n_obs);
int eval_node(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                     uint64_t seed, uint32_t sweep, int n_draws, double* lse_out, int32_t* draws_out,
                     double* scores_out, const double* snew_override, bool time_it);
int sample_children(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const int32_t* excl,
                           uint64_t seed, uint32_t sweep, int32_t* vals, int n_nodes);
int ensure_plan_dev(pclean_ctx* ctx, int block_id);
int prefetch_fast_root(pclean_ctx* ctx, int block_id, int node_id, hipStream_t side);
int ensure_agg(pclean_ctx* ctx, int block_id, int node_id, const ItemList& il, const AggDev** out);
