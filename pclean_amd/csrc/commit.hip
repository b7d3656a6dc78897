// Device-resident commit of an observed-class sweep (commit_core.h holds the algorithm and what it restates):
// the latent tables, their reference counts, free lists and the observed rows' current referents stay in HBM
// between sweeps; a sweep run with deferred outputs (pclean_set_sweep_mode) is followed by pclean_commit_device, and
// ONE stream synchronisation returns a small summary (how many rows moved, how every touched table looks) — no
// per-row data crosses PCIe.  Whatever the device commit cannot do (a created row would hold a ProposalDummyValue,
// a table would outgrow its capacity, more new-row records than the scratch holds) is detected BEFORE anything is
// modified and reported: the caller then commits on the host as before (pclean_sweep_fetch).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#define PCC_DEVICE
// phase clock of the commit kernel (PCLEAN_COMMIT_PROF=1 prints it): thread 0 stamps the constant-rate counter
__device__ long long pcc_prof_t[64];
__device__ int pcc_prof_n;
#define PCC_STAMP(name)                                                   \
  do {                                                                    \
    if (tid == 0 && blockIdx.x == 0 && pcc_prof_n < 64) pcc_prof_t[pcc_prof_n++] = (long long)wall_clock64(); \
  } while (0)
#define PCC_CUR_SEPARATE
#include "commit_core.h"
#include "sweep_state.h"

struct PccRefresh {  // one table slot of pcc_refresh_kernel
  const int64_t* counts;
  double* logc_full;
  double* logc_m1;
  const double* lut;
  int32_t lut_n, stride;
};
struct PccRefreshAll {
  PccRefresh t[PCC_MAX_SLOTS];
};
struct PccSums {
  unsigned long long total[PCC_MAX_SLOTS], live[PCC_MAX_SLOTS], maxc[PCC_MAX_SLOTS];
  int32_t lut_overflow, pad;
};

struct CommitSlot {
  int table_id = -1;
  int stride = 0;
  bool state_set = false;
  DevBuf<uint8_t> live;
  DevBuf<int32_t> free_stack, gflag, gscan, glist, origin, chg;
  DevBuf<double> lut;
  int lut_n = 0;
  double lut_discount = 0.0;
  bool lut_valid = false;
  int64_t max_count = 0;  // largest reference count known (last upload / last commit)
};

struct CommitState {
  bool enabled = false;
  int n_slots = 0, n_blocks = 0, n_plans = 0;
  int slot_of_table[PCLEAN_MAX_TABLES];
  int plan_block[PCC_MAX_BLOCKS];  // block id of plan p (the blocks with a reference slot, in block order)
  CommitSlot slot[PCC_MAX_SLOTS];
  PccTable h_tables[PCC_MAX_SLOTS];
  PccPlan h_plans[PCC_MAX_BLOCKS];
  DevBuf<PccTable> d_tables;
  DevBuf<PccPlan> d_plans;
  DevBuf<PccBlock> d_blocks;
  DevBuf<int32_t> d_colmap[PCC_MAX_BLOCKS];
  DevBuf<int32_t> d_states;  // [PCC_MAX_SLOTS][PCC_ST_WORDS]
  bool tables_dirty = true, plans_dirty = true;
  // per plan scratch
  int kcap[PCC_MAX_BLOCKS];
  DevBuf<int32_t> ht[PCC_MAX_BLOCKS], rep[PCC_MAX_BLOCKS], flags[PCC_MAX_BLOCKS], scan[PCC_MAX_BLOCKS], base[PCC_MAX_BLOCKS],
      newid[PCC_MAX_BLOCKS], recpos[PCC_MAX_BLOCKS];
  DevBuf<PccResult> d_res;
  DevBuf<PccSums> d_sums;
  DevBuf<int> d_bar;   // arrival counter of pcc_commit_mw_kernel's barriers (monotone) ...
  int bar_count = 0;   // ... and its value when the next launch starts
  // several ranks (pclean_commit_device_dist): segment capacities (the same on every rank: from the window's size, then
  // from the last commit's global totals), the all-gather buffers, the gathered lists and the gathered-form blocks
  int cap_m[PCC_MAX_BLOCKS] = {0}, cap_k[PCC_MAX_BLOCKS] = {0};
  DevBuf<int32_t> seg, seg_all, g_counts2;
  DevBuf<int32_t> g_moved[PCC_MAX_BLOCKS], g_choice[PCC_MAX_BLOCKS], g_new[PCC_MAX_BLOCKS], g_chosen[PCC_MAX_BLOCKS],
      g_vals[PCC_MAX_BLOCKS];
  DevBuf<PccBlock> d_blocks_g;
  PccBlock* h_blocks_g = nullptr;
  int32_t* h_g_counts2 = nullptr;
  // page-locked mirrors
  PccBlock* h_blocks = nullptr;
  PccResult* h_res = nullptr;
  PccSums* h_sums = nullptr;
  int32_t* h_states = nullptr;
};

static CommitState* cst(pclean_ctx* ctx) {
  if (!ctx->commit_state) ctx->commit_state = new CommitState();
  return (CommitState*)ctx->commit_state;
}

void pclean_commit_state_free(pclean_ctx* ctx) {
  if (!ctx->commit_state) return;
  CommitState* c = (CommitState*)ctx->commit_state;
  for (auto& s : c->slot) {
    s.live.release(); s.free_stack.release(); s.gflag.release(); s.gscan.release(); s.glist.release(); s.origin.release(); s.chg.release();
    s.lut.release();
  }
  c->d_tables.release(); c->d_plans.release(); c->d_blocks.release(); c->d_states.release(); c->d_res.release();
  c->d_sums.release();
  c->d_bar.release();
  for (int b = 0; b < PCC_MAX_BLOCKS; ++b) {
    c->d_colmap[b].release(); c->ht[b].release(); c->rep[b].release(); c->flags[b].release(); c->scan[b].release();
    c->base[b].release(); c->newid[b].release(); c->recpos[b].release();
  }
  if (c->h_blocks) (void)hipHostFree(c->h_blocks);
  if (c->h_res) (void)hipHostFree(c->h_res);
  if (c->h_sums) (void)hipHostFree(c->h_sums);
  if (c->h_states) (void)hipHostFree(c->h_states);
  if (c->h_blocks_g) (void)hipHostFree(c->h_blocks_g);
  if (c->h_g_counts2) (void)hipHostFree(c->h_g_counts2);
  c->seg.release(); c->seg_all.release(); c->g_counts2.release(); c->d_blocks_g.release();
  for (int b = 0; b < PCC_MAX_BLOCKS; ++b) {
    c->g_moved[b].release(); c->g_choice[b].release(); c->g_new[b].release(); c->g_chosen[b].release(); c->g_vals[b].release();
  }
  ctx->dev_cur.release();
  delete c;
  ctx->commit_state = nullptr;
}

// pclean_set_table replaced the table's device arrays: whatever pclean_commit_set_table_state said about the previous upload
// (live flags, free stack, high-water mark) is stale until it is called again (pclean_commit_device checks state_set)
void pclean_commit_table_reuploaded(pclean_ctx* ctx, int table_id) {
  if (!ctx->commit_state || table_id < 0 || table_id >= PCLEAN_MAX_TABLES) return;
  CommitState* c = (CommitState*)ctx->commit_state;
  if (!c->enabled) return;
  const int si = c->slot_of_table[table_id];
  if (si >= 0) c->slot[si].state_set = false;
}

// ---- kernels --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void pcc_commit_kernel(PccTable* tb, int n_slots, const PccPlan* plans, const PccBlock* blocks,
                                                          int n_blocks, PccResult* res, int gathered) {
  __shared__ int32_t part[1025];
  if (threadIdx.x == 0) pcc_prof_n = 0;
  if (threadIdx.x == 0 && !gathered) res->fallback_in = 0;  // (several ranks: pcc_merge_kernel wrote it; read by this thread)
  if (threadIdx.x == 0)
    for (int s = 0; s < n_slots; ++s) {
      tb[s].state[PCC_ST_COLS_CHANGED] = 0;
      tb[s].state[PCC_ST_CREATED] = 0;
      tb[s].state[PCC_ST_DELETED] = 0;
      tb[s].state[PCC_ST_NCHG] = 0;
    }
  pcc_commit(tb, n_slots, plans, blocks, n_blocks, res, part, (int)threadIdx.x, (int)blockDim.x);
}

// ---- one workgroup PER PLAN, when every plan's tables are its own (PccPlan::exclusive: hospital's Hospital / Measure slots,
// every program of this image).  The one-workgroup kernel above walks the plans one after the other (1M rows: 90 us of
// hashing the Measure slot's new-row records, then 65 us of reference counts + garbage collection of the Hospital plan,
// then the Measure plan's creation and collection); their phases touch disjoint tables and disjoint scratch, so they run
// side by side.  What they share — "is the commit refused?", decided before anything is modified — goes through a barrier
// across the workgroups: a monotone device counter (never reset: the host passes the value it starts from), every
// workgroup's thread 0 arrives and spins; the grid is one workgroup per plan (<= 16), always co-resident.
__device__ void pcc_mw_barrier(int* ctr, int target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(ctr, 1);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}
__global__ __launch_bounds__(1024) void pcc_commit_mw_kernel(PccTable* tb, int n_slots, const PccPlan* plans, const PccBlock* blocks,
                                                             int n_blocks, PccResult* res, int gathered, int* bar, int bar0) {
  __shared__ int32_t part[1025];
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x, bi = (int)blockIdx.x;
  if (bi == 0 && tid == 0) {  // what pcc_commit's first lines and the one-workgroup kernel's do
    pcc_prof_n = 0;
    if (!gathered) res->fallback_in = 0;
    for (int s = 0; s < n_slots; ++s) {
      tb[s].state[PCC_ST_COLS_CHANGED] = 0;
      tb[s].state[PCC_ST_CREATED] = 0;
      tb[s].state[PCC_ST_DELETED] = 0;
      tb[s].state[PCC_ST_NCHG] = 0;
    }
    res->fallback = res->fallback_in;
    res->n_changed = 0;
    for (int s = 0; s < PCC_MAX_SLOTS; ++s) res->alloc_upper[s] = 0;
    for (int b = 0; b < PCC_MAX_BLOCKS; ++b) res->n_records[b] = res->n_distinct[b] = res->n_nested[b] = 0;
  }
  pcc_mw_barrier(bar, bar0 + n_blocks);
  if (res->fallback) {  // (uniform: written before the barrier) — the second barrier's arrivals are still made: the host counts them
    if (tid == 0) atomicAdd(bar, 1);
    return;
  }
  pcc_prepare_block(tb, plans[bi], blocks[bi], bi, res, tid, nt);
  __syncthreads();
  if (tid == 0) {  // the capacity of this plan's own tables (pcc_commit's check; every table has one user)
    const PccPlan& pl = plans[bi];
    for (int u = 0; u < pl.n_used; ++u) {
      const int s = pl.used_slot[u];
      const int a = res->alloc_upper[s];
      if (a == 0) continue;
      if (tb[s].state[PCC_ST_NHW] + (a > tb[s].state[PCC_ST_NFREE] ? a - tb[s].state[PCC_ST_NFREE] : 0) > tb[s].stride)
        atomicOr(&res->fallback, PCC_FB_CAPACITY);
    }
  }
  pcc_mw_barrier(bar, bar0 + 2 * n_blocks);
  if (res->fallback) return;
  pcc_apply_block(tb, plans[bi], blocks[bi], bi, res, part, tid, nt);
}

// the moved rows' current referents, after the commit kernel decided the created rows' ids (grid.y = plan)
__global__ __launch_bounds__(256) void pcc_cur_kernel(const PccBlock* blocks, const PccResult* res) {
  if (res->fallback) return;
  pcc_update_cur(blocks[blockIdx.y], (int)(blockIdx.x * 256 + threadIdx.x), (int)(gridDim.x * 256));
}

// several ranks: this rank's lists -> its segment of the all-gather buffer; the gathered segments -> global lists
// (grid.y = plan; commit_core.h: pcc_pack / pcc_merge)
struct PccGathered {  // per plan: where pcc_merge puts the concatenated lists
  int32_t* g_moved[PCC_MAX_BLOCKS];
  int32_t* g_choice[PCC_MAX_BLOCKS];
  int32_t* g_new[PCC_MAX_BLOCKS];
  int32_t* g_chosen[PCC_MAX_BLOCKS];
  int32_t* g_vals[PCC_MAX_BLOCKS];
  int32_t cap_out_m[PCC_MAX_BLOCKS], cap_out_k[PCC_MAX_BLOCKS];
};
__global__ __launch_bounds__(256) void pcc_pack_kernel(PccSegLayout L, const PccBlock* blocks, int empty, int32_t* seg) {
  pcc_pack(L, (int)blockIdx.y, blocks[blockIdx.y], empty, seg, (int)(blockIdx.x * 256 + threadIdx.x), (int)(gridDim.x * 256));
}
__global__ __launch_bounds__(256) void pcc_merge_kernel(PccSegLayout L, int n_ranks, const int32_t* all, PccGathered g,
                                                        int32_t* counts2, PccResult* res) {
  const int p = blockIdx.y;
  pcc_merge(L, p, n_ranks, all, g.cap_out_m[p], g.cap_out_k[p], g.g_moved[p], g.g_choice[p], g.g_new[p], g.g_chosen[p], g.g_vals[p],
            counts2 + 2 * p, &res->fallback_in, (int)(blockIdx.x * 256 + threadIdx.x), (int)(gridDim.x * 256),
            blockIdx.x == 0 ? counts2 + 2 * PCC_MAX_BLOCKS + 2 * p : nullptr);
}

__global__ void pcc_live_kernel(int n, const int64_t* counts, uint8_t* live) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) live[r] = counts[r] > 0 ? 1 : 0;
}

// CRP prior pieces of every touched table from its counts: log(count - discount) through a table of host-libm
// logarithms (the values pclean_set_table computes: bit-identical whichever side commits), and the sums the host needs
// for the table's scalars.
__global__ __launch_bounds__(256) void pcc_refresh_kernel(PccRefreshAll a, const PccResult* res, PccSums* sums) {
  if (res->fallback) return;
  const PccRefresh& t = a.t[blockIdx.y];
  const int r = blockIdx.x * 256 + threadIdx.x;
  unsigned long long c = 0;
  if (r < t.stride) {
    const int64_t cc = t.counts[r];
    c = cc > 0 ? (unsigned long long)cc : 0ull;
    const bool over = c >= (unsigned long long)t.lut_n;
    if (over) atomicOr(&sums->lut_overflow, 1);
    t.logc_full[r] = (c > 0 && !over) ? t.lut[c] : -__builtin_inf();
    t.logc_m1[r] = (c > 1 && !over) ? t.lut[c - 1] : -__builtin_inf();
  }
  __shared__ unsigned long long s_tot[4], s_live[4], s_max[4];
  unsigned long long tot = c, lv = c > 0 ? 1ull : 0ull, mx = c;
  for (int o = 32; o > 0; o >>= 1) {
    tot += __shfl_xor(tot, o, 64);
    lv += __shfl_xor(lv, o, 64);
    const unsigned long long x = __shfl_xor(mx, o, 64);
    mx = x > mx ? x : mx;
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_tot[wave] = tot;
    s_live[wave] = lv;
    s_max[wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      s_tot[0] += s_tot[w];
      s_live[0] += s_live[w];
      s_max[0] = s_max[w] > s_max[0] ? s_max[w] : s_max[0];
    }
    if (s_tot[0]) atomicAdd(&sums->total[blockIdx.y], s_tot[0]);
    if (s_live[0]) atomicAdd(&sums->live[blockIdx.y], s_live[0]);
    if (s_max[0]) atomicMax(&sums->maxc[blockIdx.y], s_max[0]);
  }
}

// ---- set-up ---------------------------------------------------------------------------------------------------------
// Prepare the device-resident commit for blocks 0 .. n_blocks-1 of the loaded plan.  Returns PCLEAN_OK and *supported
// = 1, or *supported = 0 with the reason in pclean_last_error (plan shapes the device commit does not take: the caller
// keeps committing on the host).
extern "C" int pclean_commit_enable(pclean_ctx* ctx, int32_t n_blocks, int32_t* supported) {
  if (!ctx || n_blocks <= 0 || n_blocks > PCLEAN_MAX_BLOCKS || !supported)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_commit_enable: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  *supported = 0;
  CommitState* c = cst(ctx);
  c->enabled = false;
  c->n_blocks = n_blocks;
  for (auto& s : c->slot) {
    s.table_id = -1;
    s.state_set = false;
    s.lut_valid = false;
  }
  std::vector<std::vector<PccNodeIn>> nodes(n_blocks);
  std::vector<PccBlockIn> in(n_blocks);
  for (int bi = 0; bi < n_blocks; ++bi) {
    const Block& b = ctx->block[bi];
    if (!b.valid) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_enable: block %d not loaded", bi);
    for (const pclean_node& n : b.nodes) {
      nodes[bi].push_back(PccNodeIn{n.kind, n.table, n.child_begin, n.n_children, n.parent, n.parent_fk_col, n.colmap_begin,
                                    n.dummy_value});
      if (n.kind == PCLEAN_NODE_FK && (!ctx->cand[n.table].valid || ctx->cand[n.table].is_options))
        return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_enable: latent table %d of block %d not uploaded", n.table, bi);
    }
    in[bi] = PccBlockIn{nodes[bi].data(), (int32_t)nodes[bi].size(), b.children.data(), b.is_score ? 1 : 0};
  }
  static_assert(PCLEAN_MAX_TABLES <= 64, "PccSchema::slot_of_table");
  PccSchema* sc = new PccSchema();
  const char* why = pcc_build_schema(in.data(), n_blocks, *sc);
  if (!why)
    for (int s = 0; s < sc->n_slots && !why; ++s)
      if (ctx->cand[sc->slot_table[s]].n_cols > PCC_MAX_COLS) why = "a class with more than 96 flattened columns";
  if (why) {
    delete sc;
    (void)pclean_fail(ctx, PCLEAN_OK, "device-resident commit not available for this plan: %s", why);
    return PCLEAN_OK;
  }
  c->n_slots = sc->n_slots;
  c->n_plans = sc->n_plans;
  for (int t = 0; t < PCLEAN_MAX_TABLES; ++t) c->slot_of_table[t] = sc->slot_of_table[t];
  for (int s = 0; s < sc->n_slots; ++s) {
    c->slot[s].table_id = sc->slot_table[s];
    c->h_tables[s] = sc->tables[s];
  }
  for (int p = 0; p < sc->n_plans; ++p) {
    c->plan_block[p] = sc->plan_block[p];
    c->h_plans[p] = sc->plans[p];
    const Block& b = ctx->block[sc->plan_block[p]];
    if (c->d_colmap[p].alloc(std::max<size_t>(b.colmap.size(), 2))) {
      delete sc;
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    }
    if (!b.colmap.empty()) {
      const hipError_t e = hipMemcpy(c->d_colmap[p].p, b.colmap.data(), b.colmap.size() * 4, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        delete sc;
        return pclean_fail(ctx, PCLEAN_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
      }
    }
    c->h_plans[p].colmap = c->d_colmap[p].p;
    c->kcap[p] = 0;
  }
  delete sc;
  if (c->d_tables.alloc(PCC_MAX_SLOTS) || c->d_plans.alloc(PCC_MAX_BLOCKS) || c->d_blocks.alloc(PCC_MAX_BLOCKS) ||
      c->d_states.alloc(PCC_MAX_SLOTS * PCC_ST_WORDS) || c->d_res.alloc(1) || c->d_sums.alloc(1))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  if (!c->h_blocks) {
    HIPCHK(ctx, hipHostMalloc((void**)&c->h_blocks, sizeof(PccBlock) * PCC_MAX_BLOCKS, hipHostMallocDefault));
    HIPCHK(ctx, hipHostMalloc((void**)&c->h_res, sizeof(PccResult), hipHostMallocDefault));
    HIPCHK(ctx, hipHostMalloc((void**)&c->h_sums, sizeof(PccSums), hipHostMallocDefault));
    HIPCHK(ctx, hipHostMalloc((void**)&c->h_states, sizeof(int32_t) * PCC_MAX_SLOTS * PCC_ST_WORDS, hipHostMallocDefault));
  }
  c->tables_dirty = c->plans_dirty = true;
  c->enabled = true;
  *supported = 1;
  return PCLEAN_OK;
}

extern "C" int pclean_commit_n_slots(pclean_ctx* ctx, int32_t* n_slots, int32_t* table_ids) {
  if (!ctx || !n_slots) return PCLEAN_ERR_ARG;
  CommitState* c = cst(ctx);
  *n_slots = c->enabled ? c->n_slots : 0;
  if (table_ids)
    for (int s = 0; s < *n_slots; ++s) table_ids[s] = c->slot[s].table_id;
  return PCLEAN_OK;
}

// Allocation state of latent table `table_id` as the host trace holds it (LatentTable.n / .free), after the table itself
// was uploaded with pclean_set_table — rows [0, n_rows of the upload) are the table's device capacity.
extern "C" int pclean_commit_set_table_state(pclean_ctx* ctx, int32_t table_id, int32_t n_hw, int32_t n_free,
                                             const int32_t* free_stack) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || n_hw < 0 || n_free < 0 || (n_free > 0 && !free_stack))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_commit_set_table_state: bad arguments");
  CommitState* c = cst(ctx);
  if (!c->enabled || c->slot_of_table[table_id] < 0)
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_set_table_state: table %d is not part of an enabled commit", table_id);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int si = c->slot_of_table[table_id];
  CommitSlot& s = c->slot[si];
  CandTable& t = ctx->cand[table_id];
  if (!t.valid || n_hw > t.n_rows || n_free > n_hw)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_commit_set_table_state: table %d: %d rows in use, capacity %d", table_id, n_hw,
                       t.n_rows);
  const size_t st = std::max(t.n_rows, 1);
  if (s.live.alloc(st) || s.free_stack.alloc(st) || s.gflag.alloc(st) || s.gscan.alloc(st) || s.glist.alloc(st) ||
      s.origin.alloc(4 * st) || s.chg.alloc(st))
    return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  s.stride = t.n_rows;
  t.n_used = n_hw;  // rows beyond the high-water mark are spare capacity (ctx.h)
  HIPCHK(ctx, hipMemsetAsync(s.gflag.p, 0, st * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(s.origin.p, 0, st * 16, ctx->stream));
  hipLaunchKernelGGL(pcc_live_kernel, dim3((unsigned)((st + 255) / 256)), dim3(256), 0, ctx->stream, t.n_rows, t.counts.p, s.live.p);
  if (n_free) HIPCHK(ctx, hipMemcpyAsync(s.free_stack.p, free_stack, (size_t)n_free * 4, hipMemcpyHostToDevice, ctx->stream));
  int32_t words[PCC_ST_WORDS] = {n_hw, n_free, 0, 0, 0, 0, 0, 0};
  HIPCHK(ctx, hipMemcpyAsync(c->d_states.p + (size_t)si * PCC_ST_WORDS, words, sizeof words, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (host buffers of the caller / the stack)
  s.max_count = 0;
  for (int64_t v : t.h_counts) s.max_count = std::max(s.max_count, v);
  PccTable& pt = c->h_tables[si];
  pt.cols = t.cols.p;
  pt.counts = t.counts.p;
  pt.live = s.live.p;
  pt.free_stack = s.free_stack.p;
  pt.state = c->d_states.p + (size_t)si * PCC_ST_WORDS;
  pt.gflag = s.gflag.p;
  pt.gscan = s.gscan.p;
  pt.glist = s.glist.p;
  pt.origin = s.origin.p;
  pt.chg = s.chg.p;
  pt.stride = t.n_rows;
  pt.n_cols = t.n_cols;
  s.state_set = true;
  c->tables_dirty = true;
  return PCLEAN_OK;
}

extern "C" int pclean_set_cur(pclean_ctx* ctx, int32_t n_blocks, const int32_t* cur) {
  if (!ctx || n_blocks <= 0 || n_blocks > PCLEAN_MAX_BLOCKS || !cur || ctx->n_rows <= 0)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_cur: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)n_blocks * ctx->n_rows;
  if (ctx->dev_cur.alloc(n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpyAsync(ctx->dev_cur.p, cur, n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->dev_cur_blocks = n_blocks;
  ctx->dev_cur_valid = true;
  return PCLEAN_OK;
}

extern "C" int pclean_get_cur(pclean_ctx* ctx, int32_t n_blocks, int32_t* cur) {
  if (!ctx || !cur || !ctx->dev_cur_valid || n_blocks != ctx->dev_cur_blocks)
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_get_cur: no device-resident referents of that shape");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(cur, ctx->dev_cur.p, (size_t)n_blocks * ctx->n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PCLEAN_OK;
}

extern "C" int pclean_drop_cur(pclean_ctx* ctx) {
  if (!ctx) return PCLEAN_ERR_ARG;
  ctx->dev_cur_valid = false;
  return PCLEAN_OK;
}

// Device state of latent table `table_id` -> host (every array sized by the table's device capacity, pclean_table_shape):
// state8 = PCC_ST_* words, cols [n_cols][cap], counts [cap], live [cap], free_stack [cap] (state8[1] entries are valid),
// origin [cap][4] (marks are cleared on the device afterwards).
extern "C" int pclean_commit_pull_table(pclean_ctx* ctx, int32_t table_id, int32_t* state8, int32_t* cols, int64_t* counts,
                                        uint8_t* live, int32_t* free_stack, int32_t* origin) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || !state8)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_commit_pull_table: bad arguments");
  CommitState* c = cst(ctx);
  if (!c->enabled || c->slot_of_table[table_id] < 0 || !c->slot[c->slot_of_table[table_id]].state_set)
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_pull_table: table %d has no device-resident state", table_id);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int si = c->slot_of_table[table_id];
  CommitSlot& s = c->slot[si];
  CandTable& t = ctx->cand[table_id];
  const size_t st = (size_t)t.n_rows;
  HIPCHK(ctx, hipMemcpyAsync(state8, c->d_states.p + (size_t)si * PCC_ST_WORDS, PCC_ST_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (cols && st) HIPCHK(ctx, hipMemcpyAsync(cols, t.cols.p, st * t.n_cols * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (counts && st) HIPCHK(ctx, hipMemcpyAsync(counts, t.counts.p, st * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (live && st) HIPCHK(ctx, hipMemcpyAsync(live, s.live.p, st, hipMemcpyDeviceToHost, ctx->stream));
  if (free_stack && st) HIPCHK(ctx, hipMemcpyAsync(free_stack, s.free_stack.p, st * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (origin && st) {
    HIPCHK(ctx, hipMemcpyAsync(origin, s.origin.p, st * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(s.origin.p, 0, st * 16, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  // the host holds the device's columns again: the library's mirror with them, so that the host's next re-upload can say
  // which rows it changed (pclean_set_table: the chain of deltas the compact byte tables are refreshed along)
  if (cols && st && t.h_cols_stale && t.h_cols.size() == st * (size_t)t.n_cols) {
    memcpy(t.h_cols.data(), cols, st * (size_t)t.n_cols * 4);
    t.h_cols_stale = false;
  }
  return PCLEAN_OK;
}

static int ensure_lut(pclean_ctx* ctx, CommitSlot& s, int64_t need) {
  const CandTable& t = ctx->cand[s.table_id];
  if (s.lut_valid && s.lut_discount == t.discount && s.lut_n > need) return PCLEAN_OK;
  const int64_t n = std::max<int64_t>(4096, 2 * need + 1024);
  if (n > (int64_t)1 << 30) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "reference count beyond the log table");
  if (s.lut.alloc((size_t)n)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  std::vector<double> h((size_t)n);
  h[0] = -INFINITY;
  for (int64_t cc = 1; cc < n; ++cc) h[(size_t)cc] = std::log((double)cc - t.discount);  // as pclean_set_table
  HIPCHK(ctx, hipMemcpy(s.lut.p, h.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  s.lut_n = (int)n;
  s.lut_discount = t.discount;
  s.lut_valid = true;
  return PCLEAN_OK;
}

static int launch_refresh(pclean_ctx* ctx, CommitState* c) {
  PccRefreshAll ra{};
  int max_stride = 1;
  for (int si = 0; si < c->n_slots; ++si) {
    CommitSlot& s = c->slot[si];
    CandTable& t = ctx->cand[s.table_id];
    ra.t[si] = PccRefresh{t.counts.p, t.logc_full.p, t.logc_m1.p, s.lut.p, s.lut_n, t.n_rows};
    max_stride = std::max(max_stride, t.n_rows);
  }
  { const int rcz = dev_zero(ctx, c->d_sums.p, sizeof(PccSums)); if (rcz) return rcz; }
  hipLaunchKernelGGL(pcc_refresh_kernel, dim3((max_stride + 255) / 256, c->n_slots), dim3(256), 0, ctx->stream, ra, c->d_res.p,
                     c->d_sums.p);
  {
    const int rcs = d2h_small(ctx, c->h_sums, c->d_sums.p, sizeof(PccSums));  // (flushed by the caller's pclean_sweep_finish_queue / d2h_flush)
    if (rcs) return rcs;
  }
  return PCLEAN_OK;
}

// Commit the last pclean_sweep (run with cur == NULL and deferred outputs) on the device.  One stream synchronisation.
// out->fallback != 0: nothing was modified; finish the sweep with pclean_sweep_fetch and commit on the host.
static int commit_device_impl(pclean_ctx* ctx, int32_t n_blocks, uint32_t sweep_idx, bool dist, bool local_empty,
                              int32_t max_local_rows, pclean_commit_summary* out);
extern "C" int pclean_commit_device(pclean_ctx* ctx, int32_t n_blocks, uint32_t sweep_idx, pclean_commit_summary* out) {
  if (ctx && ctx->comm_ranks > 1)
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_device: several ranks commit through pclean_commit_device_dist "
                                              "(every rank's moved rows and new-row records are gathered on the device first)");
  return commit_device_impl(ctx, n_blocks, sweep_idx, false, false, 0, out);
}
// The commit of a sweep whose rows are sharded over the ranks of pclean_comm_init (collective: every rank calls it at
// the same point with the same n_blocks / sweep_idx / max_local_rows).  The delta reference counts are summed over the ranks
// (one all-reduce), every rank's moved rows and new-row records are all-gathered into every rank's HBM (one fixed-capacity
// all-gather, rank order = row order) and the SAME commit kernel runs over the concatenation on every rank: identical
// tables, free lists, counts and referents everywhere, nothing crosses PCIe.  local_empty != 0: this rank swept no row of
// the window (no pclean_sweep before the call).  max_local_rows: the largest shard of the window (bounds the first
// exchange's segments).  A one-rank communicator takes the same path (PCLEAN_FORCE_DIST runs).
extern "C" int pclean_commit_device_dist(pclean_ctx* ctx, int32_t n_blocks, uint32_t sweep_idx, int32_t local_empty,
                                         int32_t max_local_rows, pclean_commit_summary* out) {
  if (!ctx || !ctx->rccl_comm || ctx->comm_ranks < 1 || max_local_rows < 0)
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_device_dist: pclean_comm_init first");
  return commit_device_impl(ctx, n_blocks, sweep_idx, true, local_empty != 0, max_local_rows, out);
}
static int commit_device_impl(pclean_ctx* ctx, int32_t n_blocks, uint32_t sweep_idx, bool dist, bool local_empty,
                              int32_t max_local_rows, pclean_commit_summary* out) {
  if (!ctx || !out) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_commit_device: bad arguments");
  CommitState* c = cst(ctx);
  SweepState* s = st(ctx);
  if (!c->enabled || c->n_blocks != n_blocks) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_device: pclean_commit_enable first");
  if (!local_empty && (!s->outputs_pending || !s->last_dev_cur || s->last_blocks != n_blocks))
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_device: needs a pclean_sweep with device-resident referents "
                                              "(cur == NULL) and deferred outputs right before it");
  if (dist && (!ctx->dev_cur_valid || ctx->dev_cur_blocks != n_blocks))
    return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_device_dist: pclean_set_cur first");
  for (int si = 0; si < c->n_slots; ++si)
    if (!c->slot[si].state_set || c->slot[si].stride != ctx->cand[c->slot[si].table_id].n_rows)
      return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_commit_device: table %d was re-uploaded without "
                                                "pclean_commit_set_table_state", c->slot[si].table_id);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  memset(out, 0, sizeof *out);
  const int N = local_empty ? 0 : s->last_N;
  const int world = dist ? ctx->comm_ranks : 1;
  // option-value pointers / table pointers may have moved with a re-upload
  for (int p = 0; p < c->n_plans; ++p) {
    const Block& b = ctx->block[c->plan_block[p]];
    PccPlan& pl = c->h_plans[p];
    for (size_t i = 0; i < b.nodes.size(); ++i)
      if (b.nodes[i].kind == PCLEAN_NODE_LEAF) {
        const int32_t* ov = ctx->cand[b.nodes[i].table].cols.p;
        if (pl.opt_vals[i] != ov) {
          pl.opt_vals[i] = ov;
          c->plans_dirty = true;
        }
      }
  }
  for (int si = 0; si < c->n_slots; ++si) {
    CandTable& t = ctx->cand[c->slot[si].table_id];
    PccTable& pt = c->h_tables[si];
    if (pt.cols != t.cols.p || pt.counts != t.counts.p) {
      pt.cols = t.cols.p;
      pt.counts = t.counts.p;
      c->tables_dirty = true;
    }
    // a commit can raise a count by at most the rows of the window (+ the rows it creates)
    int rc = ensure_lut(ctx, c->slot[si], c->slot[si].max_count + 1);
    if (rc) return rc;
  }
  if (c->tables_dirty) {
    HIPCHK(ctx, hipMemcpyAsync(c->d_tables.p, c->h_tables, sizeof(PccTable) * PCC_MAX_SLOTS, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (pageable source)
    c->tables_dirty = false;
  }
  if (c->plans_dirty) {
    HIPCHK(ctx, hipMemcpyAsync(c->d_plans.p, c->h_plans, sizeof(PccPlan) * PCC_MAX_BLOCKS, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    c->plans_dirty = false;
  }
  for (int p = 0; p < c->n_plans; ++p) {
    const int bi = c->plan_block[p];
    const Block& b = ctx->block[bi];
    BlockRun& r = s->run[bi];
    const PccPlan& pl = c->h_plans[p];
    if (c->kcap[p] < 16384) c->kcap[p] = 16384;
    const int kcap = c->kcap[p];
    int hsz = 1;
    while (hsz < 4 * kcap) hsz <<= 1;
    if (c->ht[p].alloc(hsz) || c->rep[p].alloc(kcap) || c->flags[p].alloc(kcap) || c->scan[p].alloc(kcap) ||
        c->base[p].alloc((size_t)kcap * pl.n_used) || c->newid[p].alloc(kcap) || c->recpos[p].alloc(kcap))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    PccBlock& pb = c->h_blocks[p];
    pb.N = N;
    pb.nn = (int)b.nodes.size();
    pb.block_id = bi;
    pb.sweep_idx = (int32_t)sweep_idx;
    pb.row_lo = (int32_t)(s->row_offset + ctx->active_begin);
    pb.pad0 = 0;
    pb.choice = r.choice.p;
    pb.chosen = s->chosen.p;
    pb.chosen_newpos = r.chosen_newpos.p;
    pb.vals = r.vals.p;
    pb.moved_list = r.moved_list.p;
    pb.new_list = r.new_list.p;
    pb.counts2 = s->tail_counts.p + 2 * bi;
    pb.cur = local_empty ? nullptr : const_cast<int32_t*>(s->last_cur_base) + (size_t)bi * s->last_cur_ld;
    pb.moved_choice = nullptr;
    pb.rec_chosen = nullptr;
    pb.delta = ctx->cand[b.nodes[0].table].stats.p;
    pb.kcap = kcap;
    pb.hmask = hsz - 1;
    pb.ht = c->ht[p].p;
    pb.rep = c->rep[p].p;
    pb.flags = c->flags[p].p;
    pb.scan = c->scan[p].p;
    pb.base = c->base[p].p;
    pb.newid = c->newid[p].p;
    pb.recpos = c->recpos[p].p;
  }
  HIPCHK(ctx, hipMemcpyAsync(c->d_blocks.p, c->h_blocks, sizeof(PccBlock) * c->n_plans, hipMemcpyHostToDevice, ctx->stream));
  const PccBlock* commit_blocks = c->d_blocks.p;
  bool stats_reduced = false;
  int cur_rows = N;  // rows pcc_cur_kernel's grid is sized for
  if (dist) {
    // Everything that can fail on ONE rank alone (allocations, layout) comes first: once the first collective is queued a
    // rank that returned early would leave the others waiting in the next one.  (A failure after this point is an RCCL /
    // HIP error: the communicator is unusable on every rank anyway.)
    // 1. this rank's lists -> its segment (layout + buffers)
    PccSegLayout L{};
    L.n_plans = c->n_plans;
    int32_t off = 0;
    for (int p = 0; p < c->n_plans; ++p) {
      if (c->cap_m[p] <= 0) c->cap_m[p] = std::max(max_local_rows, 16);  // (first exchange / after a refusal: whatever a shard can hold)
      if (c->cap_k[p] <= 0) c->cap_k[p] = std::max(max_local_rows, 16);
      L.off[p] = off;
      L.cap_m[p] = c->cap_m[p];
      L.cap_k[p] = c->cap_k[p];
      L.nn[p] = (int)ctx->block[c->plan_block[p]].nodes.size();
      off += pcc_seg_words(L.cap_m[p], L.cap_k[p], L.nn[p]);
    }
    L.seg_words = off;
    if (c->seg.alloc((size_t)off) || c->seg_all.alloc((size_t)off * world) || c->g_counts2.alloc(4 * PCC_MAX_BLOCKS) ||
        c->d_blocks_g.alloc(PCC_MAX_BLOCKS))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    if (!c->h_blocks_g) {
      HIPCHK(ctx, hipHostMalloc((void**)&c->h_blocks_g, sizeof(PccBlock) * PCC_MAX_BLOCKS, hipHostMallocDefault));
      HIPCHK(ctx, hipHostMalloc((void**)&c->h_g_counts2, sizeof(int32_t) * 4 * PCC_MAX_BLOCKS, hipHostMallocDefault));
    }
    PccGathered G{};
    for (int p = 0; p < c->n_plans; ++p) {
      const size_t om = (size_t)L.cap_m[p] * world, ok = (size_t)L.cap_k[p] * world;
      if (c->g_moved[p].alloc(om) || c->g_choice[p].alloc(om) || c->g_new[p].alloc(ok) || c->g_chosen[p].alloc(ok) ||
          c->g_vals[p].alloc(ok * L.nn[p]))
        return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
      G.g_moved[p] = c->g_moved[p].p;
      G.g_choice[p] = c->g_choice[p].p;
      G.g_new[p] = c->g_new[p].p;
      G.g_chosen[p] = c->g_chosen[p].p;
      G.g_vals[p] = c->g_vals[p].p;
      G.cap_out_m[p] = (int32_t)std::min<size_t>(om, 0x7fffffff);
      G.cap_out_k[p] = (int32_t)std::min<size_t>(ok, 0x7fffffff);
      // scratch of the commit kernel: every gathered record
      if ((size_t)c->kcap[p] < ok) {
        c->kcap[p] = (int)ok;
        const int kcap = c->kcap[p];
        int hsz = 1;
        while (hsz < 4 * kcap) hsz <<= 1;
        const PccPlan& pl = c->h_plans[p];
        if (c->ht[p].alloc(hsz) || c->rep[p].alloc(kcap) || c->flags[p].alloc(kcap) || c->scan[p].alloc(kcap) ||
            c->base[p].alloc((size_t)kcap * pl.n_used) || c->newid[p].alloc(kcap) || c->recpos[p].alloc(kcap))
          return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
      }
    }
    for (int p = 0; p < c->n_plans; ++p) {  // the gathered-form blocks
      const int bi = c->plan_block[p];
      PccBlock& pb = c->h_blocks_g[p];
      pb = c->h_blocks[p];
      int hsz = 1;
      while (hsz < 4 * c->kcap[p]) hsz <<= 1;
      pb.kcap = c->kcap[p];
      pb.hmask = hsz - 1;
      pb.ht = c->ht[p].p;
      pb.rep = c->rep[p].p;
      pb.flags = c->flags[p].p;
      pb.scan = c->scan[p].p;
      pb.base = c->base[p].p;
      pb.newid = c->newid[p].p;
      pb.recpos = c->recpos[p].p;
      pb.N = ctx->n_rows;
      pb.row_lo = 0;
      pb.choice = nullptr;
      pb.chosen = nullptr;
      pb.chosen_newpos = nullptr;
      pb.vals = c->g_vals[p].p;
      pb.moved_list = c->g_moved[p].p;
      pb.new_list = c->g_new[p].p;
      pb.moved_choice = c->g_choice[p].p;
      pb.rec_chosen = c->g_chosen[p].p;
      pb.counts2 = c->g_counts2.p + 2 * p;
      pb.cur = ctx->dev_cur.p + (size_t)bi * ctx->n_rows;
    }
    HIPCHK(ctx, hipMemcpyAsync(c->d_blocks_g.p, c->h_blocks_g, sizeof(PccBlock) * c->n_plans, hipMemcpyHostToDevice, ctx->stream));
    { const int rcz = dev_zero(ctx, c->d_res.p, sizeof(PccResult)); if (rcz) return rcz; }
    // 2. the delta reference counts of the root tables, summed over the ranks (in place): from here on the device buffers
    //    hold the SUMS — a refused commit tells its caller so (summary.stats_reduced), the host exchange must not sum again
    int32_t tids[PCC_MAX_BLOCKS];
    int n_t = 0;
    for (int p = 0; p < c->n_plans; ++p) {
      const int t = ctx->block[c->plan_block[p]].nodes[0].table;
      bool seen = false;
      for (int k = 0; k < n_t; ++k) seen |= tids[k] == t;
      if (!seen) tids[n_t++] = t;
    }
    int rc = pclean_comm_allreduce_stats_queue(ctx, n_t, tids, local_empty ? 1 : 0);
    if (rc) return rc;
    stats_reduced = true;
    // 3. all-gather of the segments; the concatenated lists
    hipLaunchKernelGGL(pcc_pack_kernel, dim3(64, c->n_plans), dim3(256), 0, ctx->stream, L, c->d_blocks.p, local_empty ? 1 : 0,
                       c->seg.p);
    rc = pclean_comm_allgather_i32(ctx, c->seg.p, c->seg_all.p, (size_t)L.seg_words);
    if (rc) return rc;
    hipLaunchKernelGGL(pcc_merge_kernel, dim3(64, c->n_plans), dim3(256), 0, ctx->stream, L, world, c->seg_all.p, G, c->g_counts2.p,
                       c->d_res.p);
    HIPCHK(ctx, hipMemcpyAsync(c->h_g_counts2, c->g_counts2.p, sizeof(int32_t) * 4 * PCC_MAX_BLOCKS, hipMemcpyDeviceToHost, ctx->stream));
    commit_blocks = c->d_blocks_g.p;
    cur_rows = 0;
    for (int p = 0; p < c->n_plans; ++p) cur_rows = std::max<int64_t>(cur_rows, (int64_t)L.cap_m[p] * world);
  }
  // one workgroup per plan when no two plans share a table (pcc_commit_mw_kernel), else one workgroup for everything
  static const bool no_mw = getenv("PCLEAN_COMMIT_ONE_WG") != nullptr;
  bool all_exclusive = c->n_plans > 1 && !no_mw;
  for (int p = 0; p < c->n_plans; ++p) all_exclusive = all_exclusive && c->h_plans[p].exclusive != 0;
  if (all_exclusive && !c->d_bar.p) {
    if (c->d_bar.alloc(16)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    HIPCHK(ctx, hipMemsetAsync(c->d_bar.p, 0, 16 * sizeof(int), ctx->stream));
    c->bar_count = 0;
  }
  if (all_exclusive) {
    if (c->bar_count > (1 << 30)) {  // (never in practice: 2 x plans per commit)
      HIPCHK(ctx, hipMemsetAsync(c->d_bar.p, 0, 16 * sizeof(int), ctx->stream));
      c->bar_count = 0;
    }
    hipLaunchKernelGGL(pcc_commit_mw_kernel, dim3(c->n_plans), dim3(1024), 0, ctx->stream, c->d_tables.p, c->n_slots, c->d_plans.p,
                       commit_blocks, c->n_plans, c->d_res.p, dist ? 1 : 0, c->d_bar.p, c->bar_count);
    c->bar_count += 2 * c->n_plans;
  } else {
    hipLaunchKernelGGL(pcc_commit_kernel, dim3(1), dim3(1024), 0, ctx->stream, c->d_tables.p, c->n_slots, c->d_plans.p,
                       commit_blocks, c->n_plans, c->d_res.p, dist ? 1 : 0);
  }
  hipLaunchKernelGGL(pcc_cur_kernel, dim3(std::max(1, std::min(1024, (cur_rows + 255) / 256)), c->n_plans), dim3(256), 0, ctx->stream,
                     commit_blocks, c->d_res.p);
  int rc = launch_refresh(ctx, c);
  if (rc) return rc;
  rc = d2h_small(ctx, c->h_res, c->d_res.p, sizeof(PccResult));
  if (!rc) rc = d2h_small(ctx, c->h_states, c->d_states.p, sizeof(int32_t) * PCC_MAX_SLOTS * PCC_ST_WORDS);
  if (rc) return rc;
  rc = pclean_sweep_finish_queue(ctx);  // (its read-backs and the three above: one kernel)
  if (!rc) rc = d2h_flush(ctx);         // (no sweep before the call, a rank without rows: nothing of the sweep's was queued)
  if (rc) return rc;
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the ONE synchronisation of sweep + commit
  rc = pclean_sweep_finish_synced(ctx);
  if (rc) return rc;
  static const bool prof = getenv("PCLEAN_COMMIT_PROF") != nullptr;
  if (prof) {
    long long t[64];
    int n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(pcc_prof_n), sizeof n);
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(pcc_prof_t), sizeof t);
    fprintf(stderr, "[pclean] commit kernel phases (us since start; 100 MHz clock):");
    for (int i = 1; i < n && i < 64; ++i) fprintf(stderr, " %.1f", (double)(t[i] - t[0]) / 100.0);
    fprintf(stderr, "  (records %d %d, moved %d %d)\n", c->h_res->n_records[0], c->h_res->n_records[1], s->h_counts[0], s->h_counts[2]);
  }
  out->fallback = c->h_res->fallback;
  out->n_changed = c->h_res->n_changed;
  out->n_slots = c->n_slots;
  out->stats_reduced = stats_reduced ? 1 : 0;
  for (int p = 0; p < c->n_plans; ++p) {
    out->n_records[c->plan_block[p]] = c->h_res->n_records[p];
    out->n_distinct[c->plan_block[p]] = c->h_res->n_distinct[p];
    // scratch for the next sweep: twice what this one needed
    if (2 * c->h_res->n_records[p] > c->kcap[p]) c->kcap[p] = 2 * c->h_res->n_records[p];
    if (dist) {  // the next exchange's segments: four times the LARGEST rank's lists of this commit (every rank read the same
      // headers: the same capacities everywhere); a refused commit starts over from what a shard can hold
      c->cap_m[p] = out->fallback ? 0 : std::max(1024, 4 * c->h_g_counts2[2 * PCC_MAX_BLOCKS + 2 * p] + 256);
      c->cap_k[p] = out->fallback ? 0 : std::max(256, 4 * c->h_g_counts2[2 * PCC_MAX_BLOCKS + 2 * p + 1] + 256);
    }
  }
  if (dist && !out->fallback) {
    int64_t moved = 0;
    for (int p = 0; p < c->n_plans; ++p) moved += c->h_g_counts2[2 * p];
    out->n_changed = (int32_t)moved;
  }
  if (out->fallback) return PCLEAN_OK;
  for (int guard = 0; c->h_sums->lut_overflow && guard < 4; ++guard) {  // a count beyond the log table: rebuild it, redo the priors
    for (int si = 0; si < c->n_slots; ++si) {
      rc = ensure_lut(ctx, c->slot[si], (int64_t)c->h_sums->maxc[si] + 1);
      if (rc) return rc;
    }
    rc = launch_refresh(ctx, c);
    if (!rc) rc = d2h_flush(ctx);
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (c->h_sums->lut_overflow) return pclean_fail(ctx, PCLEAN_ERR_CAPACITY, "pclean_commit_device: log table overflow");
  for (int si = 0; si < c->n_slots; ++si) {
    CommitSlot& cs = c->slot[si];
    CandTable& t = ctx->cand[cs.table_id];
    const int32_t* w = c->h_states + (size_t)si * PCC_ST_WORDS;
    const double total = (double)c->h_sums->total[si], live = (double)c->h_sums->live[si];
    t.scal[0] = std::log(total + t.strength);  // as pclean_set_table
    t.scal[1] = std::log((total - 1.0) + t.strength);
    t.scal[2] = std::log(t.strength + t.discount * live);
    t.scal[3] = std::log(t.strength + t.discount * (live - 1.0));
    cs.max_count = (int64_t)c->h_sums->maxc[si];
    t.logc_max = cs.max_count > 0 ? std::log((double)cs.max_count - t.discount) : -INFINITY;
    t.h_mirror_stale = true;
    t.version = ++g_pclean_version;
    if (w[PCC_ST_COLS_CHANGED]) {
      // the rows whose columns were written: tables derived from the columns (root_wave.hip's candidate-compact byte
      // tables) refresh those rows alone when they were built from the version this commit started from
      t.cols_delta_base = t.cols_version;
      t.cols_delta_n = w[PCC_ST_NCHG];
      t.cols_delta_rows = cs.chg.p;
      t.commit_delta_base = t.cols_version;
      t.commit_delta_n = w[PCC_ST_NCHG];
      t.commit_delta_rows = cs.chg.p;
      t.cols_version = t.version;
      t.commit_delta_next = t.cols_version;
      t.h_cols_stale = true;  // (until the host pulls the table: pclean_commit_pull_table)
      t.delta_log.clear();
      t.union_n = -1;
    }
    pclean_commit_slot& o = out->slot[si];
    o.table_id = cs.table_id;
    o.n_hw = w[PCC_ST_NHW];
    t.n_used = w[PCC_ST_NHW];
    o.n_free = w[PCC_ST_NFREE];
    o.cols_changed = w[PCC_ST_COLS_CHANGED];
    o.created = w[PCC_ST_CREATED];
    o.deleted = w[PCC_ST_DELETED];
    o.total = (int64_t)c->h_sums->total[si];
    o.live = (int64_t)c->h_sums->live[si];
    o.max_count = cs.max_count;
  }
  return PCLEAN_OK;
}
