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
#include "sweep_internal.h"

// every blocking point of the orchestration goes through here: PCLEAN_TRACE_SYNC=1 lists them per call
int g_pclean_sync_count = 0;

// One 32-bit count from the device to the host in the middle of a call (how many items need the next step): instead of a
// copy + stream synchronisation (~25 us until the host thread is woken) a one-thread kernel publishes the value in
// page-locked memory and the host spins on a sequence number — the remaining host round trips of a sweep cost a few
// microseconds each.  PCLEAN_NO_POLL=1: the plain copy + synchronisation.
__global__ void publish_count_kernel(const unsigned int* __restrict__ src, volatile unsigned int* __restrict__ dst, unsigned int seq) {
  dst[0] = *src;
  __threadfence_system();
  dst[1] = seq;
}
int read_count(pclean_ctx* ctx, const void* dev, void* out, const char* func, int line) {
  SweepState* s = st(ctx);
  static const bool no_poll = getenv("PCLEAN_NO_POLL") != nullptr;
  static const bool trace = getenv("PCLEAN_TRACE_SYNC") != nullptr;
  if (trace) fprintf(stderr, "[pclean sync %d] %s:%d (count)\n", ++g_pclean_sync_count, func, line);
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
  if (s->on_first_wait) {  // (work for another stream that the host queues while the device is still behind it)
    std::function<int()> f;
    f.swap(s->on_first_wait);
    const int rcf = f();
    if (rcf) return rcf;
  }
  volatile unsigned int* h = s->h_poll;
  for (long spins = 0; h[1] != seq; ++spins)
    if (spins > 2000000) {  // (a failed launch would spin for ever: let the runtime report it)
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (h[1] != seq) return pclean_fail(ctx, PCLEAN_ERR_HIP, "count read-back: the publishing kernel did not run");
    }
  *(unsigned int*)out = h[0];
  return PCLEAN_OK;
}

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
// ---- the three kernels above in ONE pass (gather_ctx + ctx_count + scan + ctx_fill: 6 dispatches, 0.19 ms per 1M rows x 20
// particles).  One thread per row: the context of every particle (written to it_ctx as before), compared with particle 0's.
// Item i is row i's PRIMARY item (particle 0's context: nearly every row has no other); each further distinct context of a
// row becomes an EXTRA item N + k, k from one atomic per workgroup — no scan, no second pass.  The numbering of the extra
// items follows the workgroups' arrival: nothing downstream depends on item order (scores are per item, draws are keyed by
// (row, particle)).  Extra items beyond extra_cap are counted, not written: the host re-runs with room (read_count).
#define CI_PC 8  // particles of a row whose loads ctx_items_kernel has in flight together
__global__ __launch_bounds__(256) void ctx_items_kernel(int N, int P, CtxSrc cs, const int32_t* __restrict__ cur_b,
                                                        int32_t* __restrict__ it_ctx, int32_t* __restrict__ slot_item,
                                                        int32_t* __restrict__ row, int32_t* __restrict__ ctxv,
                                                        int32_t* __restrict__ excl, unsigned int* __restrict__ n_extra,
                                                        int extra_cap) {
  __shared__ unsigned int wcnt[4];
  __shared__ unsigned int bbase;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t NP = (size_t)N * P;
  int c0[PCLEAN_MAX_CTX] = {0, 0, 0, 0};
  static_assert(PCLEAN_MAX_CTX == 4, "c0 initialiser");
  unsigned int mine = 0;  // extra items of this row
  if (i < N) {
    bool all_same = true;
    // CI_PC particles at a time: their choices, then their referents' context values, each level loaded together and
    // unconditionally (slots beyond P re-read particle P - 1, a NEW choice reads row 0 and is resolved afterwards) — one
    // particle after the other was a chain of 2 P dependent round trips per row
    for (int p0 = 0; p0 < P; p0 += CI_PC) {
#pragma unroll
      for (int s = 0; s < PCLEAN_MAX_CTX; ++s) {
        if (s >= cs.n_ctx) break;
        int ch[CI_PC];
        int32_t v[CI_PC];
#pragma unroll
        for (int u = 0; u < CI_PC; ++u) ch[u] = cs.pchoice[s][(size_t)min(p0 + u, P - 1) * N + i];
#pragma unroll
        for (int u = 0; u < CI_PC; ++u) v[u] = cs.root_col[s][ch[u] >= 0 ? ch[u] : 0];
        bool any_new = false;
#pragma unroll
        for (int u = 0; u < CI_PC; ++u) any_new |= ch[u] < 0;
        if (any_new) {
#pragma unroll
          for (int u = 0; u < CI_PC; ++u)
            if (ch[u] < 0 && p0 + u < P)
              v[u] = resolve_new_value(cs.plan[s], 0, cs.col[s],
                                       cs.vals[s] + (size_t)cs.pnewpos[s][(size_t)(p0 + u) * N + i] * cs.n_nodes[s]);
        }
        if (p0 == 0) c0[s] = v[0];
#pragma unroll
        for (int u = 0; u < CI_PC; ++u)
          if (p0 + u < P) {
            it_ctx[(size_t)s * NP + (size_t)(p0 + u) * N + i] = v[u];
            all_same &= v[u] == c0[s];
          }
      }
    }
    row[i] = i;
    excl[i] = cur_b ? cur_b[i] : -1;
    for (int k = 0; k < PCLEAN_MAX_CTX; ++k) ctxv[(size_t)i * PCLEAN_MAX_CTX + k] = c0[k];
    if (all_same) {
      for (int p = 0; p < P; ++p) slot_item[(size_t)p * N + i] = i;
    } else {  // representatives: first particle with the same context tuple (this thread's own writes to it_ctx: cache hits)
      slot_item[i] = i;
      for (int p = 1; p < P; ++p) {
        const size_t sp = (size_t)p * N + i;
        int q = 0;
        for (; q < p; ++q) {
          bool same = true;
          for (int k = 0; k < cs.n_ctx; ++k) same &= it_ctx[(size_t)k * NP + (size_t)q * N + i] == it_ctx[(size_t)k * NP + sp];
          if (same) break;
        }
        if (q == p) {
          slot_item[sp] = -1 - (int)mine;  // the row's mine-th extra item: numbered below
          ++mine;
        } else {
          slot_item[sp] = slot_item[(size_t)q * N + i];
        }
      }
    }
  }
  const unsigned long long any = __ballot(mine != 0u);
  unsigned int incl = mine;
  if (any) {
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int x = __shfl_up(incl, o, 64);
      if (lane >= o) incl += x;
    }
  }
  if (lane == 63) wcnt[wave] = any ? incl : 0u;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    bbase = tot ? atomicAdd(n_extra, tot) : 0u;
  }
  __syncthreads();
  if (!mine) return;
  unsigned int base = bbase + incl - mine;
  for (int w = 0; w < wave; ++w) base += wcnt[w];
  for (int p = 1; p < P; ++p) {
    const size_t sp = (size_t)p * N + i;
    const int si = slot_item[sp];
    if (si >= 0) continue;
    // (a particle that shares an extra item reads the representative's entry: resolved when the representative was, since
    // representatives come first — q < p)
    int q = 0;
    for (; q < p; ++q) {
      bool same = true;
      for (int k = 0; k < cs.n_ctx; ++k) same &= it_ctx[(size_t)k * NP + (size_t)q * N + i] == it_ctx[(size_t)k * NP + sp];
      if (same) break;
    }
    if (q < p) {
      slot_item[sp] = slot_item[(size_t)q * N + i];
      continue;
    }
    const unsigned int e = base + (unsigned int)(-1 - si);
    const int j = N + (int)e;
    slot_item[sp] = j;
    if (e < (unsigned int)extra_cap) {
      row[j] = i;
      excl[j] = cur_b ? cur_b[i] : -1;
      for (int k = 0; k < PCLEAN_MAX_CTX; ++k) ctxv[(size_t)j * PCLEAN_MAX_CTX + k] = k < cs.n_ctx ? it_ctx[(size_t)k * NP + sp] : 0;
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
                                                              int32_t* __restrict__ pnewpos, int first,
                                                              const int32_t* __restrict__ emit_rows, int only_emit,
                                                              int stage_stride, const double* __restrict__ w_uni, int skip_w) {
  // w_uni: every particle of row i carries the weight 0.0 + w_uni[i] so far and w was never written (pclean_sweep: the first
  // block's log marginal is shared by a row's particles); skip_w: this launch leaves it that way (nothing stored: 8 P bytes
  // per row that the next block would only read back)
  __shared__ unsigned int wsum[PU_T / 64];
  __shared__ unsigned int bbase;
  extern __shared__ int32_t pu_stage[];  // stage_stride > 0: the workgroup's rows of draws_rm, stage_stride (odd) words apart
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The root kernels leave a row's P draws next to each other (draws_rm[i * P + p]: one or two lines per member item on the
  // writing side); a thread walking its own P x 4 bytes issues P loads a row apart from its neighbour's — P requests per
  // line.  The workgroup's rows are one contiguous piece: it is read once, coalesced, into LDS (rows an odd number of words
  // apart: no bank conflicts when every lane then reads its own row).
  if (stage_stride > 0) {
    const int r0 = blockIdx.x * PU_T;
    const int n_here = min(PU_T, N - r0);
    const int n_words = n_here * P;
    const int32_t* src = draws_rm + (size_t)r0 * P;
    for (int k = threadIdx.x; k < n_words; k += PU_T) {
      const int row = k / P, col = k - row * P;
      pu_stage[row * stage_stride + col] = src[k];
    }
    __syncthreads();
  }
  uint64_t newmask = 0;
  // (only_emit: the rows outside emit_rows are not this kernel's — particle_update_final_kernel takes them)
  if (i < N && !(only_emit && !emit_rows[i])) {
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
        d = stage_stride > 0 ? pu_stage[threadIdx.x * stage_stride + p] : draws_rm[(size_t)i * P + p];
        l = lse[i];
      }
      const int c = (p == 0 && keep >= 0) ? keep : d;
      pchoice[sp] = c;
      if (!skip_w) w[sp] = first ? 0.0 + l : (w_uni ? (0.0 + w_uni[i]) + l : w[sp] + l);  // (first block: the weights start at +0.0)
      if (c == PCLEAN_CHOICE_NEW) newmask |= 1ull << p;
      // a row without any possible candidate (log marginal -inf or NaN): bit 31 of the counter tells the host, which then
      // runs the resampling step it would otherwise know to be a no-op (pclean_sweep: equal_weights)
      if (p == 0 && !(l > -__builtin_inf())) atomicOr(n_new, 0x80000000u);
    }
    // (last block, dummy values drawable for a few rows only: the list holds those rows' NEW slots — the others'
    // contents are sampled after the final choice, for the chosen particle alone)
    if (emit_rows && !emit_rows[i]) newmask = 0;
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
    bbase = total ? (atomicAdd(n_new, total) & 0x7fffffffu) : 0u;  // (bit 31: the degenerate-row flag)
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

// the weights a launch of particle_update_kernel with skip_w did not store: w[p][i] = 0.0 + w_uni[i]
__global__ void materialise_w_kernel(int N, int P, const double* __restrict__ w_uni, double* __restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double v = 0.0 + w_uni[i];
  for (int p = 0; p < P; ++p) w[(size_t)p * N + i] = v;
}
__global__ void add_weight_kernel(size_t n, const double* lse, double* w) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) w[t] += lse[t];
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
  int64_t* pairs[PCLEAN_MAX_BLOCKS];  // the particles' own choices of prior proposals (gauss_prior_kernel: two int32 per slot)
  int n_pairs;
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
  for (int a = 0; a < arrays.n_pairs; ++a) {  // (oracle/sweep.h: plocals follow their particles)
    int64_t* arr = arrays.pairs[a] + i;
    int64_t t2[MAXP];
    for (int p = 0; p < P; ++p) t2[p] = arr[(size_t)ancestors[(size_t)p * n_rows + i] * n_rows];
    for (int p = 0; p < P; ++p) arr[(size_t)p * n_rows] = t2[p];
  }
  for (int p = 0; p < P; ++p) w[(size_t)p * n_rows + i] = 0.0;
}

// row_inference.jl:158-165 + return value 186 (logml = accumulated resampling increments + log mean weight)
template <int PMAX>
__global__ void final_choice_kernel(int n_rows, int P, const double* logw, size_t sr, size_t sp, int use_mh,
                                    int is_csmc, const int32_t* csmc_flag, uint64_t seed, uint32_t sweep,
                                    int64_t row_offset, int32_t* chosen, double* log_total, const double* logml_acc,
                                    double* logml, const int32_t* __restrict__ only_rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  if (only_rows && !only_rows[i]) return;  // (the other rows were finished by particle_update_final_kernel)
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


// The LAST block's particle update and the final choice in one pass (when nothing can touch individual particle weights
// in between: no drawable ProposalDummyValue, no prior-mode likelihood, no scoring block behind it): a thread has its row's
// P weights in registers anyway — they are never stored (160 MB written and read back by the two separate kernels at 1M
// rows x 20 particles), nor is the list of NEW slots built (their contents are sampled for the chosen particle alone).
// Same operations in the same order as particle_update_kernel followed by final_choice_kernel: bit-identical results.
template <int PMAX>
__global__ __launch_bounds__(256) void particle_update_final_kernel(
    int N, int P, const int32_t* __restrict__ draws_rm, const double* __restrict__ lse, const int32_t* __restrict__ slot_item,
    const int32_t* __restrict__ draws_item, const double* __restrict__ lse_item, const int32_t* __restrict__ cur_b,
    int32_t* __restrict__ pchoice, const double* __restrict__ w, int first, int use_mh, const int32_t* __restrict__ csmc_flag,
    uint64_t seed, uint32_t sweep, int64_t row_offset, int32_t* __restrict__ chosen, const double* __restrict__ logml_acc,
    double* __restrict__ logml, const int32_t* __restrict__ skip_rows, int lazy, const double* __restrict__ w_uni) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (skip_rows && skip_rows[i]) return;  // (a row that can draw a dummy value: the separate kernels, with the weight corrections between them)
  const int keep = cur_b ? cur_b[i] : -1;
  // lazy: the root kernels left lists, not draws (enum.h: RootExtra) — the chosen particle's referent is drawn afterwards
  // (lazy_draw_kernel); only the retained particle's is known here
  if (lazy && keep >= 0) pchoice[i] = keep;
  FixW<PMAX> f;
  double wv[PMAX];
  f.m = -__builtin_inf();
  // Every particle's loads of one level are issued together and unconditionally (the slots beyond P re-read particle P - 1;
  // their values are never used): a load under `p < P` is a branch with its own wait, and the row's 2 P dependent loads
  // (item of the slot, then the item's log marginal) were a serial chain of 2 P round trips per thread.
  int dv[PMAX];
  double lv[PMAX], wp[PMAX];
  const size_t Ns = (size_t)N;
  if (slot_item) {
    int itm[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) itm[p] = slot_item[(size_t)(p < P ? p : P - 1) * Ns + i];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) lv[p] = lse_item[itm[p]];
    if (!lazy) {
#pragma unroll
      for (int p = 0; p < PMAX; ++p) dv[p] = draws_item[(size_t)itm[p] * P + (p < P ? p : P - 1)];
    }
  } else {
    const double l0 = lse[i];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) lv[p] = l0;
    if (!lazy) {
#pragma unroll
      for (int p = 0; p < PMAX; ++p) dv[p] = draws_rm[(size_t)i * P + (p < P ? p : P - 1)];
    }
  }
  if (lazy) {
#pragma unroll
    for (int p = 0; p < PMAX; ++p) dv[p] = 0;
  }
  if (first) {
#pragma unroll
    for (int p = 0; p < PMAX; ++p) wp[p] = 0.0;
  } else if (w_uni) {  // (see particle_update_kernel)
    const double wu = 0.0 + w_uni[i];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) wp[p] = wu;
  } else {
#pragma unroll
    for (int p = 0; p < PMAX; ++p) wp[p] = w[(size_t)(p < P ? p : P - 1) * Ns + i];
  }
#pragma unroll
  for (int p = 0; p < PMAX; ++p) {
    wv[p] = 0.0;
    if (p < P) {
      if (!lazy) pchoice[(size_t)p * Ns + i] = (p == 0 && keep >= 0) ? keep : dv[p];
      wv[p] = wp[p] + lv[p];
      f.m = fmax(f.m, wv[p]);
    }
  }
  f.U = 0;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) {
    f.u[p] = (p < P && f.m != -__builtin_inf()) ? pclean_fixw(wv[p] - f.m) : 0ull;
    f.U += f.u[p];
  }
  const uint32_t rr = (uint32_t)((int64_t)i + row_offset);
  const bool csmc = csmc_flag[i] >= 0;
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
  logml[i] = logml_acc[i] + pclean_lse_from_fix(f.m, f.U) - pclean_log((double)P);
}

// per block after the final choice: the chosen particle's referent, its new-row record, the delta reference
// counts (the all-reduce payload) and the flags of moved rows / rows with a new referent.  Tables with few rows
// (hist_rows > 0: a handful of very popular referents, e.g. 28 measures for 1M records) accumulate the deltas in an
// LDS histogram per workgroup first — thousands of global atomics on the same few addresses would serialise.
__global__ __launch_bounds__(256) void finalize_block_kernel(int n_rows, const int32_t* chosen, const int32_t* pchoice,
                                                             const int32_t* pnewpos, const int32_t* cur_b,
                                                             int32_t* choice, int32_t* chosen_newpos,
                                                             unsigned long long* stats, int hist_rows,
                                                             int32_t* moved_flag, int32_t* new_flag,
                                                             unsigned int* __restrict__ wg_moved,
                                                             unsigned int* __restrict__ wg_new) {
  extern __shared__ int32_t hist[];
  for (int k = threadIdx.x; k < hist_rows; k += 256) hist[k] = 0;
  if (hist_rows) __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int is_moved = 0, is_new = 0;
  if (i < n_rows) {
    const size_t s = (size_t)chosen[i] * n_rows + i;
    const int c = pchoice[s];
    const int o = cur_b[i];
    choice[i] = c;
    const int np = c == PCLEAN_CHOICE_NEW ? pnewpos[s] : -1;
    chosen_newpos[i] = np;
    new_flag[i] = is_new = np >= 0 ? 1 : 0;
    moved_flag[i] = is_moved = c != o ? 1 : 0;
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
  // this workgroup's share of the two ordered lists (tail_scan_kernel / tail_scatter_kernel)
  const int n_moved = __syncthreads_count(is_moved), n_newr = __syncthreads_count(is_new);
  if (threadIdx.x == 0) {
    wg_moved[blockIdx.x] = (unsigned int)n_moved;
    wg_new[blockIdx.x] = (unsigned int)n_newr;
  }
}

// ---- ordered lists of the rows that moved / got a new referent, every block's at once: the per-workgroup counts
// finalize_block_kernel left -> offsets (one workgroup per list) -> the rows, ascending (one workgroup per 256-row tile and
// list).  Two dispatches per sweep where four hipcub::DeviceSelect calls were sixteen.
#define TAIL_MAX_LISTS (2 * PCLEAN_MAX_BLOCKS)
__global__ __launch_bounds__(1024) void tail_scan_kernel(int nb, unsigned int* __restrict__ wg_cnt, int32_t* __restrict__ totals) {
  __shared__ unsigned int wsum[16];
  unsigned int* c = wg_cnt + (size_t)blockIdx.x * nb;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int carry = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int j = base + threadIdx.x;
    const unsigned int v = j < nb ? c[j] : 0u;
    unsigned int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int x = __shfl_up(incl, o, 64);
      if (lane >= o) incl += x;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned int wbase = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
      if (k < wave) wbase += wsum[k];
      tot += wsum[k];
    }
    if (j < nb) c[j] = carry + wbase + incl - v;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = (int32_t)carry;
}
struct TailLists {
  const int32_t* flag[TAIL_MAX_LISTS];  // null: no such list (scoring block)
  int32_t* list[TAIL_MAX_LISTS];
};
__global__ __launch_bounds__(256) void tail_scatter_kernel(int n_rows, int nb, TailLists tl, const unsigned int* __restrict__ wg_off) {
  __shared__ unsigned int wcnt[4];
  const int l = blockIdx.y;
  const int32_t* flag = tl.flag[l];
  if (!flag) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool f = i < n_rows && flag[i] != 0;
  const unsigned long long mk = __ballot(f);
  if (lane == 0) wcnt[wave] = (unsigned int)__popcll(mk);
  __syncthreads();
  if (!f) return;
  unsigned int pos = wg_off[(size_t)l * nb + blockIdx.x] + (unsigned int)__popcll(mk & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += wcnt[w];
  tl.list[l][pos] = i;
}

// own enumerated choices (locals) of the chosen particle, drawn from their conditional given
// the chosen referent: the inner draws of the nested enumeration (proposal_compiler.jl:115-127)
// ---- prior proposals (use_dd_proposals = false) of a block whose slot carries a Gaussian term (experiments/rents/run.jl:19-25).
// The own choices the data-driven proposal enumerates inside the candidate branch are sampled from their priors by every
// particle (block_proposal.jl:42-56: the proposal's and the model's densities of a sampled choice cancel), an OBSERVED own
// choice is scored (62-64), the retained particle keeps the row's current ones (cur_locals), and the observed number is
// scored given the particle's referent and own choices.  One thread per (particle, row) slot; the oracle restates the order
// of the additions (oracle/sweep.h: gauss_prior_term): observed choices' densities, Normal log-density, - log |derivative|.
__global__ void gauss_prior_kernel(int n_rows, int P, GaussDev g, PlanDev plan, const int32_t* __restrict__ pchoice,
                                   const int32_t* __restrict__ pnewpos, const int32_t* __restrict__ vals, int n_nodes,
                                   const int32_t* __restrict__ cur_b, const int32_t* __restrict__ cur_locals, uint64_t seed,
                                   uint32_t sweep, uint32_t block, int64_t row_offset, double* __restrict__ w,
                                   int32_t* __restrict__ plocals) {
  const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= (size_t)n_rows * P) return;
  const int p = (int)(slot / (size_t)n_rows), i = (int)(slot % (size_t)n_rows);
  const int choice = pchoice[slot];
  const int32_t* v = choice >= 0 ? nullptr : vals + (size_t)pnewpos[slot] * n_nodes;
  const bool keep = p == 0 && cur_b && cur_b[i] >= 0 && cur_locals;
  const uint32_t rr = (uint32_t)((int64_t)i + row_offset);
  int l[2] = {0, 0}, out[2] = {-1, -1};
  double s = 0.0;
  for (int k = 0; k < g.n_locals; ++k) {
    const int o = g.local_obs[k] ? g.local_obs[k][i] : -1;
    if (o >= 0) {
      l[k] = o;
      s += g.local_logp[k];
    } else if (keep && cur_locals[2 * (size_t)i + k] >= 0) {
      l[k] = cur_locals[2 * (size_t)i + k];
    } else {
      l[k] = (int)pclean_mulhi64(pclean_rand64(seed, rr, PCLEAN_SITE_LOCALS(block), (uint32_t)p | ((uint32_t)(k + 1) << 16), sweep),
                                 (uint64_t)g.local_n[k]);
    }
    out[k] = l[k];
  }
  plocals[2 * slot] = out[0];
  plocals[2 * slot + 1] = out[1];
  const double xv = g.x[i];
  if (xv == xv) {
    int idx = 0;
    for (int d = 0; d < g.n_dims; ++d) {
      int val;
      if (g.src_kind[d] == PCLEAN_GSRC_LOCAL)
        val = l[g.src_slot[d]];
      else if (g.src_kind[d] == PCLEAN_GSRC_CAND)
        val = choice >= 0 ? g.src_ptr[d][choice] : resolve_new_value(plan, 0, g.src_slot[d], v);
      else
        val = g.src_ptr[d][i];  // PCLEAN_GSRC_OBS
      idx += g.stride[d] * val;
    }
    const int u = g.t_kind == PCLEAN_GSRC_LOCAL ? l[g.t_src] : 0;
    s += gauss_normal_logpdf(g.tx[u] ? g.tx[u][i] : xv * g.t_scale[u], g.mu[idx], g.sigma, g.log_sigma);
    s -= g.tl[u] ? g.tl[u][i] : g.t_lad[u];
  }
  w[slot] += s;
}
// ... and the chosen particle's own choices
__global__ void locals_pick_kernel(int n_rows, const int32_t* __restrict__ chosen, const int32_t* __restrict__ plocals,
                                   int32_t* __restrict__ locals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const size_t slot = (size_t)chosen[i] * n_rows + i;
  locals[2 * i] = plocals[2 * slot];
  locals[2 * i + 1] = plocals[2 * slot + 1];
}

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
  for (auto& b : s->more_banks) b.release();
  for (auto e : s->prof_ev) (void)hipEventDestroy(e);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->pre_stream) (void)hipStreamDestroy(s->pre_stream);
  if (s->pre_fork) (void)hipEventDestroy(s->pre_fork);
  if (s->pre_join) (void)hipEventDestroy(s->pre_join);
  if (s->evg0) (void)hipEventDestroy(s->evg0);
  if (s->evg1) (void)hipEventDestroy(s->evg1);
  if (s->evs) (void)hipEventDestroy(s->evs);
  if (s->eve) (void)hipEventDestroy(s->eve);
  delete s;
  ctx->sweep_state = nullptr;
}

// Start of an entry point that evaluates plan nodes: scratch pool rewound, overflow counters cleared.
int begin_call(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  s->pool_used = 0;
  s->dbg_desc = nullptr;
  s->dummy_used = false;
  ctx->prior_mode = false;
  s->over_rec.clear();
  if (s->over_ctr.alloc(OVER_SLOTS + STAT_WORDS + CTR_BANK)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  { const int rcz = dev_zero(ctx, s->over_ctr.p, (OVER_SLOTS + STAT_WORDS + CTR_BANK) * sizeof(unsigned int)); if (rcz) return rcz; }
  s->bank_used = 0;
  s->scan_stats_used = false;
  return PCLEAN_OK;
}
// End of such a call, after its last stream synchronisation has been queued: the overflow counts of the sync-free
// launches go into the statistics and the "does the pre-filter pay for this option list" heuristic.
__global__ void dev_zero_kernel(uint4* __restrict__ p16, size_t n16, unsigned char* __restrict__ tail, int n_tail) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p16[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) tail[threadIdx.x] = 0;
}
int dev_zero(pclean_ctx* ctx, void* p, size_t bytes) {
  static const bool off = getenv("PCLEAN_NO_ZERO_KERNEL") != nullptr;
  if (bytes == 0) return PCLEAN_OK;
  unsigned char* b = static_cast<unsigned char*>(p);
  const size_t head = (16 - ((uintptr_t)b & 15)) & 15;  // bytes before the first 16-byte boundary
  if (!off && bytes <= 256) {  // a counter or two, wherever they lie
    hipLaunchKernelGGL(dev_zero_kernel, dim3(1), dim3(256), 0, ctx->stream, (uint4*)nullptr, (size_t)0, b, (int)bytes);
    return PCLEAN_OK;
  }
  if (off || head > 0) {  // (every buffer the sweeps zero is hipMalloc'ed or 16-byte aligned inside one: the general case keeps the API)
    HIPCHK(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
    return PCLEAN_OK;
  }
  const size_t n16 = bytes >> 4;
  const int n_tail = (int)(bytes & 15);
  const unsigned int blocks = (unsigned int)std::min<size_t>(std::max<size_t>((n16 + 255) / 256, 1), 4096);
  hipLaunchKernelGGL(dev_zero_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<uint4*>(b), n16, b + (n16 << 4), n_tail);
  return PCLEAN_OK;
}
__global__ void publish_regions_kernel(const SweepState::PubRegions pr) {
  const int r = blockIdx.x;
  const uint32_t* __restrict__ src = pr.src[r];
  uint32_t* __restrict__ dst = pr.dst[r];
  for (uint32_t i = threadIdx.x; i < pr.words[r]; i += blockDim.x) dst[i] = src[i];
}
int d2h_small(pclean_ctx* ctx, void* host, const void* dev, size_t bytes, void* host_base) {
  SweepState* s = st(ctx);
  static const bool off = getenv("PCLEAN_NO_PUBLISH_REGIONS") != nullptr;
  if (bytes == 0) return PCLEAN_OK;
  void* mapped = nullptr;
  if (!off && (bytes & 3) == 0 && bytes <= (size_t)(1 << 20) && s->pub.n < 12) {
    void* base = host_base ? host_base : host;
    auto it = s->pub_map.find(base);
    if (it != s->pub_map.end()) {
      mapped = it->second;
    } else {
      if (hipHostGetDevicePointer(&mapped, base, 0) != hipSuccess) {
        (void)hipGetLastError();
        mapped = nullptr;
      }
      s->pub_map[base] = mapped;  // (null: not mappable, remembered)
    }
    if (mapped) mapped = static_cast<char*>(mapped) + (static_cast<char*>(host) - static_cast<char*>(base));
  }
  if (!mapped) {
    HIPCHK(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return PCLEAN_OK;
  }
  const int k = s->pub.n++;
  s->pub.src[k] = static_cast<const uint32_t*>(dev);
  s->pub.dst[k] = static_cast<uint32_t*>(mapped);
  s->pub.words[k] = (uint32_t)(bytes >> 2);
  return PCLEAN_OK;
}
int d2h_flush(pclean_ctx* ctx) {
  SweepState* s = st(ctx);
  if (s->pub.n == 0) return PCLEAN_OK;
  hipLaunchKernelGGL(publish_regions_kernel, dim3(s->pub.n), dim3(256), 0, ctx->stream, s->pub);
  s->pub.n = 0;
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
int queue_over_copy(pclean_ctx* ctx) {  // before a stream synchronisation of the caller
  SweepState* s = st(ctx);
  if (s->over_rec.empty() && !s->scan_stats_used) return PCLEAN_OK;
  if (!s->h_over) HIPCHK(ctx, hipHostMalloc((void**)&s->h_over, (OVER_SLOTS + STAT_WORDS) * sizeof(unsigned int), hipHostMallocDefault));
  int rc = PCLEAN_OK;
  if (!s->over_rec.empty()) rc = d2h_small(ctx, s->h_over, s->over_ctr.p, s->over_rec.size() * sizeof(unsigned int));
  if (!rc && s->scan_stats_used) rc = d2h_small(ctx, s->h_over + OVER_SLOTS, s->over_ctr.p + OVER_SLOTS, STAT_WORDS * sizeof(unsigned int), s->h_over);
  return rc;
}
void apply_over_stats(pclean_ctx* ctx) {  // after that synchronisation
  SweepState* s = st(ctx);
  for (size_t i = 0; i < s->over_rec.size(); ++i) {
    const SweepState::OverRec& r = s->over_rec[i];
    const unsigned int h = s->h_over[i];
    if (r.min_items < 0) {  // a work-list record (eval.hip): h = groups the settle kernel left of r.n_items
      if ((size_t)h * 2 > (size_t)r.n_items) s->fast[r.block * 64 + r.node].wl_off = 32;  // (most of them: no list for a while)
      continue;
    }
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
    unsigned long long tot[5] = {0, 0, 0, 0, 0};  // 64 slots a cache line apart (root_wave.hip: WAVE_STAT_SLOTS)
    for (int sl = 0; sl < 64; ++sl)
      for (int i = 0; i < 5; ++i) tot[i] += s->h_over[OVER_SLOTS + sl * 32 + i];
    ctx->root_stats.lazy_entries = (int32_t)tot[4];
    ctx->root_stats.full_scans = (int32_t)tot[0];
    ctx->root_stats.fine_blocks = (int32_t)tot[1];
    ctx->root_stats.scored_terms = (int32_t)tot[2];
    ctx->root_stats.resolved_groups = (int32_t)tot[3];
    s->scan_stats_used = false;
  }
}
int finish_call(pclean_ctx* ctx) {
  if (st(ctx)->over_rec.empty() && !st(ctx)->scan_stats_used) return PCLEAN_OK;
  int rc = queue_over_copy(ctx);
  if (!rc) rc = d2h_flush(ctx);
  if (rc) return rc;
  PCLEAN_SYNC(ctx);
  apply_over_stats(ctx);
  return PCLEAN_OK;
}


// ---- per-phase profile (pclean_set_profiling): HIP events on the library's stream around groups of launches
int prof_phase_id(SweepState* s, const char* name) {
  for (size_t i = 0; i < s->prof_names.size(); ++i)
    if (s->prof_names[i] == name) return (int)i;
  s->prof_names.push_back(name);
  s->prof_ms.push_back(0.f);
  s->prof_launches.push_back(0);
  return (int)s->prof_names.size() - 1;
}
void prof_collect(pclean_ctx* ctx) {
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


// ---------------------------------------------------------------------------
extern "C" int pclean_set_active_rows(pclean_ctx* ctx, int32_t begin, int32_t count) {
  if (!ctx || begin < 0 || count < -1 || (count >= 0 && (int64_t)begin + count > ctx->n_rows))
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_active_rows: window outside the loaded rows");
  ctx->active_begin = count < 0 ? 0 : begin;
  ctx->active_count = count;
  return PCLEAN_OK;
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


// Can a particle of this sweep draw the ProposalDummyValue of some option list of block bi?  Cacheable lists know it
// per observed value (ensure_leaf_cache: weight of the dummy option); any other list with a dummy is assumed to.
static int block_dummy_drawable(pclean_ctx* ctx, int bi, bool* out, bool* by_row = nullptr) {
  Block& b = ctx->block[bi];
  *out = false;
  if (by_row) *by_row = true;  // every list that can draw its dummy knows for which observed values (leaf_udummy)
  for (int node = 0; node < (int)b.nodes.size(); ++node) {
    const pclean_node& n = b.nodes[node];
    if (n.kind != PCLEAN_NODE_LEAF || n.dummy_value == 0) continue;
    if (!n.cacheable || (node < (int)b.node_gauss.size() && b.node_gauss[node] >= 0)) {
      *out = true;
      if (by_row) *by_row = false;
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
      if (b.leaf_drawable[node] < 0 || !b.leaf_udummy[node].p) {
        if (by_row) *by_row = false;
        return PCLEAN_OK;
      }
      if (!by_row) return PCLEAN_OK;
    }
  }
  return PCLEAN_OK;
}

// flag[i] = 1 when some cacheable option list of block bi can draw its ProposalDummyValue for row i of the active window:
// the fixed-point weight of the dummy option for the row's observed value (ensure_leaf_cache: leaf_udummy) is not zero.
// Rebuilt when a cache, the observations or the window change.
struct DummyFlagLeaf {
  const int32_t* obs_col;
  const uint64_t* udummy;
  int32_t n_obs, pad;
};
struct DummyFlagPack {
  int32_t n_leaves, pad;
  DummyFlagLeaf leaf[DUMMY_MAX_LEAVES];
};
__global__ void dummy_rows_kernel(int n, DummyFlagPack pk, int32_t* __restrict__ flag, unsigned int* __restrict__ n_flagged) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int f = 0;
  if (i < n) {
    for (int l = 0; l < pk.n_leaves; ++l) {
      const int o = pk.leaf[l].obs_col[i];
      f |= pk.leaf[l].udummy[o < 0 ? pk.leaf[l].n_obs : o] != 0ull ? 1 : 0;
    }
    flag[i] = f;
  }
  const unsigned long long mk = __ballot(f != 0);
  if ((threadIdx.x & 63) == 0 && mk) atomicAdd(n_flagged, (unsigned int)__popcll(mk));
}
// *out = null with *none = true: no row of the window can draw any dummy (nothing to sample before the final choice)
static int dummy_rows_flags(pclean_ctx* ctx, int bi, int N, const int32_t** out, bool* none) {
  *none = false;
  Block& b = ctx->block[bi];
  SweepState* s = st(ctx);
  SweepState::DummyRows& dr = s->dummy_rows[bi];
  DummyFlagPack pk{};
  uint64_t sig = (uint64_t)N * 0x9e3779b97f4a7c15ull + (uint64_t)ctx->active_begin * 0xd6e8feb86659fd93ull + ctx->obs_version + b.version * 7919ull;
  for (int node = 0; node < (int)b.nodes.size(); ++node) {
    const pclean_node& n = b.nodes[node];
    if (n.kind != PCLEAN_NODE_LEAF || n.dummy_value == 0 || b.leaf_drawable[node] == 0) continue;
    if (pk.n_leaves >= DUMMY_MAX_LEAVES) {
      *out = nullptr;  // (more lists than the pack holds: the caller samples every NEW slot)
      return PCLEAN_OK;
    }
    const pclean_term& tm = b.terms[n.term_begin];
    const PairTable& pt = ctx->pair[tm.pair_table];
    DummyFlagLeaf& lf = pk.leaf[pk.n_leaves++];
    lf.obs_col = ctx->obs.p + (size_t)tm.obs_col * ctx->n_rows + ctx->active_begin;
    lf.udummy = b.leaf_udummy[node].p;
    lf.n_obs = pt.n_obs;
    sig = sig * 1000003ull + s->leaf_version[bi * 256 + node] + (uint64_t)(uintptr_t)lf.udummy;
  }
  if (dr.sig != sig || dr.n != N || !dr.flag.p) {
    if (dr.flag.alloc(std::max(N, 1))) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    unsigned int* ctr = fresh_counter(ctx);
    if (!ctr) return pclean_fail(ctx, PCLEAN_ERR_STATE, "counter bank missing");
    hipLaunchKernelGGL(dummy_rows_kernel, grid1(N), dim3(256), 0, ctx->stream, N, pk, dr.flag.p, ctr);
    unsigned int nf = 0;
    PCLEAN_READ_COUNT(ctx, ctr, &nf);  // (once per rebuild of the caches: a read-back is affordable)
    dr.n_flagged = (int)nf;
    dr.sig = sig;
    dr.n = N;
  }
  *out = dr.n_flagged > 0 ? dr.flag.p : nullptr;
  *none = dr.n_flagged == 0;
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
// sub-tree: AddTypos observations (plain or through a JuliaNode), noise-free observations (equality constraints: 0 or
// -inf), the MaybeSwap observations of the referring rows of a latent class; a scoring block proposes nothing and scores as
// always (62-64); a Gaussian term on the slot is scored at own choices sampled from their priors (gauss_prior_kernel).
int prior_mode_supported(pclean_ctx* ctx, const Block& b, const char* who) {
  bool ok = b.valid;
  for (const pclean_term& tm : b.terms)
    ok = ok && (tm.dens_kind == PCLEAN_DENS_ADD_TYPOS || tm.dens_kind == PCLEAN_DENS_EQUAL || tm.dens_kind == PCLEAN_DENS_MAYBE_SWAP);
  // a Gaussian term: on the slot of an observed-class block (gauss_prior_kernel scores it once per particle at the sampled
  // own choices; its copies on the nodes of a new row's choices are there for the enumeration and are not looked at), or the
  // evidence of a latent class (nothing enumerated: a likelihood term like any other)
  const bool root_g = !b.node_gauss.empty() && b.node_gauss[0] >= 0;
  for (size_t i = 0; i < b.node_gauss.size() && ok; ++i) {
    if (b.node_gauss[i] < 0) continue;
    const pclean_gauss& g = b.gauss[b.node_gauss[i]];
    if (i != 0) {
      ok = !(g.n_locals > 0 && !root_g);
      continue;
    }
    bool plain = g.transform_src_kind != PCLEAN_GSRC_EVCTX;
    for (int d = 0; d < g.n_dims; ++d)
      plain = plain && (g.src_kind[d] == PCLEAN_GSRC_CAND || g.src_kind[d] == PCLEAN_GSRC_OBS || g.src_kind[d] == PCLEAN_GSRC_LOCAL);
    ok = plain || g.n_locals == 0;
  }
  if (!ok)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "%s: use_dd_proposals = false (prior proposals, block_proposal.jl:168) is implemented "
                                            "for AddTypos, equality and MaybeSwap observations, scoring blocks and a Gaussian term "
                                            "on the block's slot whose sources are the referent, observed columns and the block's "
                                            "own choices; this plan has a Gaussian term elsewhere or with context sources", who);
  return PCLEAN_OK;
}

// device copies of what prior_terms_kernel needs of block bi: every node with its full terms, the child lists
int upload_plan_nodes(pclean_ctx* ctx, int bi, const NodeDev** nds, const int32_t** n_children,
                             const int32_t** child_begin, const int32_t** children) {
  Block& b = ctx->block[bi];
  const int nn = (int)b.nodes.size();
  std::vector<NodeDev> h(nn);
  std::vector<int32_t> nc(nn), cb(nn);
  for (int i = 0; i < nn; ++i) {
    int rc = build_node_dev(ctx, b, i, h[i]);
    if (rc) return rc;
    if (getenv("PCLEAN_DEBUG_LATENT"))
      fprintf(stderr, "[plan node %d] terms %d gauss on %d dims %d locals %d x %p mu %p src %p %p %p %p kinds %d %d %d %d\n", i, h[i].n_terms,
              h[i].g.on, h[i].g.n_dims, h[i].g.n_locals, (const void*)h[i].g.x, (const void*)h[i].g.mu, (const void*)h[i].g.src_ptr[0],
              (const void*)h[i].g.src_ptr[1], (const void*)h[i].g.src_ptr[2], (const void*)h[i].g.src_ptr[3], h[i].g.src_kind[0],
              h[i].g.src_kind[1], h[i].g.src_kind[2], h[i].g.src_kind[3]);
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

// ---- several small buffers zeroed by ONE launch (a hipMemsetAsync apiece is a dispatch apiece) -------------------------
#define ZERO_LIST_MAX 16
struct ZeroList {
  int32_t n, pad;
  void* p[ZERO_LIST_MAX];
  uint32_t words[ZERO_LIST_MAX];  // 4-byte words
};
__global__ __launch_bounds__(256) void zero_list_kernel(ZeroList zl) {
  uint32_t* p = (uint32_t*)zl.p[blockIdx.y];
  const uint32_t nw = zl.words[blockIdx.y];
  for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < nw; k += gridDim.x * 256) p[k] = 0u;
}
static int zero_list_flush(pclean_ctx* ctx, ZeroList& zl) {
  if (zl.n == 0) return PCLEAN_OK;
  uint32_t mx = 0;
  for (int k = 0; k < zl.n; ++k) mx = std::max(mx, zl.words[k]);
  const unsigned gx = std::max(1u, std::min(1024u, (mx + 255u) / 256u));
  hipLaunchKernelGGL(zero_list_kernel, dim3(gx, zl.n), dim3(256), 0, ctx->stream, zl);
  zl.n = 0;
  HIPCHK(ctx, hipGetLastError());
  return PCLEAN_OK;
}
static void zero_list_add(pclean_ctx* ctx, ZeroList& zl, void* p, size_t bytes) {  // bytes: a multiple of 4
  if (!p || bytes == 0) return;
  if (zl.n == ZERO_LIST_MAX || bytes > (size_t)0xffffffffu * 4) {  // (full, or too large for the word count: the plain way)
    (void)hipMemsetAsync(p, 0, bytes, ctx->stream);
    return;
  }
  zl.p[zl.n] = p;
  zl.words[zl.n] = (uint32_t)(bytes / 4);
  ++zl.n;
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
    HIPCHK(ctx, hipEventCreate(&s->evg0));
    HIPCHK(ctx, hipEventCreate(&s->evg1));
  }
  s->gate_timed = false;
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
  if (!w_by_first_block) { const int rcz = dev_zero(ctx, s->w.p, NP * sizeof(double)); if (rcz) return rcz; }
  { const int rcz = dev_zero(ctx, s->logml_acc.p, (size_t)N * sizeof(double)); if (rcz) return rcz; }
  ctx->timing = pclean_timing{};
  // the later blocks' root tables refresh on a side stream while block 0 runs (eval.hip: prefetch_fast_root): enqueued
  // once block 0's kernels are (see the block loop), so that the host time of those launches is not block 0's
  static const bool no_prefetch = getenv("PCLEAN_NO_COMPACT_PREFETCH") != nullptr;
  bool prefetching = false, prefetch_started = false;
  if (!s->pre_stream) {
    HIPCHK(ctx, hipStreamCreateWithFlags(&s->pre_stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreateWithFlags(&s->pre_fork, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&s->pre_join, hipEventDisableTiming));
  }
  HIPCHK(ctx, hipEventRecord(s->pre_fork, ctx->stream));
  auto start_prefetch = [&]() -> int {
    prefetch_started = true;
    if (no_prefetch || prior_mode || ctx->force_generic || n_blocks <= 1) return PCLEAN_OK;
    HIPCHK(ctx, hipStreamWaitEvent(s->pre_stream, s->pre_fork, 0));  // (recorded at the START of the sweep: nothing of block 0)
    for (int bi = 1; bi < n_blocks; ++bi) {
      const Block& pb = ctx->block[bi];
      if (pb.is_score || pb.nodes.empty() || (!pb.node_gauss.empty() && pb.node_gauss[0] >= 0)) continue;
      const int rcp = prefetch_fast_root(ctx, bi, 0, s->pre_stream);
      if (rcp) return rcp;
    }
    HIPCHK(ctx, hipEventRecord(s->pre_join, s->pre_stream));
    prefetching = true;
    return PCLEAN_OK;
  };
  struct FirstWaitGuard {  // (the hook refers to this frame: no exit of the call leaves it behind; and an error exit between
    SweepState* st_;       // the fork and the join still joins the side stream: its work must not overlap the next call)
    pclean_ctx* c_;
    bool* prefetching_;
    ~FirstWaitGuard() {
      st_->on_first_wait = nullptr;
      if (*prefetching_) (void)hipStreamWaitEvent(c_->stream, st_->pre_join, 0);
    }
  } first_wait_guard{s, ctx, &prefetching};
  static const bool late_prefetch = getenv("PCLEAN_LATE_COMPACT_PREFETCH") != nullptr;
  if (!late_prefetch)
    s->on_first_wait = [&]() -> int { return prefetch_started ? PCLEAN_OK : start_prefetch(); };
  bool hot_timed = false;
  // particle_update_kernel stages the root kernels' row-major draws through LDS (rows of P words, an odd stride apart)
  static const bool no_pu_stage = getenv("PCLEAN_NO_PU_STAGE") != nullptr;
  // (P | 1) x PU_T words: 84 KB at 20 particles; beyond PU_STAGE_MAX_BYTES of LDS (P >= 37: 64 particles would ask for 266 KB
  // of the CU's 160) the kernel reads the draws directly, as it did before the staging existed
  constexpr size_t PU_STAGE_MAX_BYTES = 150 * 1024;
  int pu_stage_stride = no_pu_stage ? 0 : (P | 1);
  if ((size_t)pu_stage_stride * PU_T * sizeof(int32_t) > PU_STAGE_MAX_BYTES) pu_stage_stride = 0;
  const size_t pu_stage_bytes = (size_t)pu_stage_stride * PU_T * sizeof(int32_t);
  if (pu_stage_bytes > 48 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      HIPCHK(ctx, hipFuncSetAttribute((const void*)particle_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)PU_STAGE_MAX_BYTES));
      attr_set = true;
    }
  }
  // The first block's log marginal is shared by the particles of a row (no context yet to tell them apart): its launch of
  // particle_update_kernel stores no weights (w_uni = that block's lse; 160 MB written and read back per 1M rows x 20
  // particles otherwise) and the next block's update starts from it.  Whoever touches individual weights in between (dummy
  // corrections, a resampling step, a scoring block, the separate final choice) gets them materialised first.
  static const bool no_w_uni = getenv("PCLEAN_NO_UNIFORM_W") != nullptr;
  const double* w_uni = nullptr;
  auto materialise_w = [&]() {
    if (!w_uni) return;
    hipLaunchKernelGGL(materialise_w_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, w_uni, s->w.p);
    w_uni = nullptr;
  };
  bool final_fused = false;            // the last block's particle update made the final choice as well
  const int32_t* final_only_rows = nullptr;  // ... for the rows outside this flag array (the final choice kernel takes the flagged ones)
  const bool defer_final_off = false;  // (placeholder of a condition that would forbid it)

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
      materialise_w();
      hipLaunchKernelGGL(score_block_kernel, grid1(NP), dim3(256), 0, ctx->stream, N, P, sb, s->w.p);
      continue;
    }
    if (prefetching && bi >= 1) {  // (the refreshed tables of this and the later blocks: one join)
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->pre_join, 0));
      prefetching = false;
    }
    const int nn = (int)b.nodes.size();
    const int32_t* cur_b = cur_base + (size_t)bi * cur_ld;
    if (r.pchoice.alloc(NP) || r.pnewpos.alloc(NP) || r.new_slots.alloc(NP) || r.choice.alloc(N) || r.chosen_newpos.alloc(N) ||
        r.moved_flag.alloc(N) || r.new_flag.alloc(N) || r.moved_list.alloc(N) || r.new_list.alloc(N))
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
    int rc = ensure_plan_dev(ctx, bi);
    if (rc) return rc;
    const bool has_ctx = b.n_ctx > 0;
    // The contents of a proposed new row matter (a) as context / scored values of LATER blocks — every particle's —
    // and (b) for the particle that is finally chosen.  For the last block only (b) is left: its sampling is
    // deferred until after the final choice and done for the chosen particles alone (same Philox counters, so
    // the values are the ones eager sampling would have produced; typically 20x fewer items).
    // A chosen ProposalDummyValue changes its particle's weight (apply_dummy_corrections): where one can be drawn
    // the NEW slots are sampled before the final choice — all of them, or, when every list that can draw its dummy knows
    // for which observed values (cacheable lists: leaf_udummy), those of the few rows that hold such a value (emit_rows).
    static const bool eager_all = getenv("PCLEAN_EAGER_NEW") != nullptr;
    static const bool no_partial = getenv("PCLEAN_NO_PARTIAL_EAGER") != nullptr;
    bool drawable = prior_mode, dummy_by_row = false;  // (prior draws of a StringPrior choice are the dummy almost surely)
    if (!prior_mode) {
      rc = block_dummy_drawable(ctx, bi, &drawable, &dummy_by_row);
      if (rc) return rc;
    }
    const int32_t* emit_rows = nullptr;
    if (bi == n_blocks - 1 && !eager_all && !no_partial && !prior_mode && drawable && dummy_by_row) {
      bool none = false;
      rc = dummy_rows_flags(ctx, bi, N, &emit_rows, &none);
      if (rc) return rc;
      if (none) drawable = false;  // the dummy's fixed-point weight is 0 for every row of the window: it cannot be drawn
    }
    // last block and nothing between its particle update and the final choice: one fused kernel (particle_update_final_kernel)
    static const bool no_fuse_final = getenv("PCLEAN_NO_FUSED_FINAL") != nullptr;
    const bool last_plain = bi == n_blocks - 1 && !prior_mode && !eager_all && !no_fuse_final && !defer_final_off;
    const bool fuse_final = last_plain && !drawable;
    // ... and when only a few rows can draw one (emit_rows): those rows take the separate kernels, all the others the fused one
    const bool split_final = last_plain && drawable && emit_rows != nullptr;
    // prior proposals, Gaussian term on the slot: scored per particle at own choices sampled from their priors, after the
    // likelihood terms of the sampled sub-tree (pclean_launch_prior_terms) and before the dummy corrections
    r.plocals_on = false;
    auto gauss_prior = [&]() -> int {
      if (b.node_gauss.empty() || b.node_gauss[0] < 0) return PCLEAN_OK;
      GaussDev gd;
      const CandTable& rt = ctx->cand[b.nodes[0].table];
      int rcg = build_gauss_dev(ctx, b.gauss[b.node_gauss[0]], &rt, gd);
      if (rcg) return rcg;
      if (r.plocals.alloc((size_t)NP * 2)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
      const int32_t* cl = (b.cur_locals.p && b.cur_locals_rows == ctx->n_rows) ? b.cur_locals.p + (size_t)ctx->active_begin * 2 : nullptr;
      hipLaunchKernelGGL(gauss_prior_kernel, grid1(NP), dim3(256), 0, ctx->stream, N, P, gd, r.plan, r.pchoice.p, r.pnewpos.p,
                         r.vals.p, (int)b.nodes.size(), cur_b, cl, seed, sweep_idx, (uint32_t)bi,
                         s->row_offset + ctx->active_begin, s->w.p, r.plocals.p);
      r.plocals_on = true;
      return PCLEAN_OK;
    };
    ItemList il;
    const int32_t* excl;
    unsigned int* n_new_ctr = nullptr;  // particles of the block that proposed a NEW referent (fresh_counter)
    SweepState::LazyOut lazy;           // the root kernels left lists instead of draws (last block: enum.h RootExtra)
    auto lazy_draws = [&]() -> int {    // ... and the chosen particles' draws follow the fused final choice
      if (!lazy.valid) return PCLEAN_OK;
      ProfScope psl(ctx, "lazy_draws");
      lazy.args.n_rows = N;
      lazy.args.n_particles = P;
      lazy.args.chosen = s->chosen.p;
      lazy.args.cur_b = cur_b;
      lazy.args.pchoice = r.pchoice.p;
      return pclean_launch_lazy_draws(ctx, lazy.args, seed, sweep_idx, lazy.site);
    };
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
      { const int rcz = dev_zero(ctx, r.lse.p, (size_t)N * sizeof(double)); if (rcz) return rcz; }
      n_new_ctr = fresh_counter(ctx);
      if (!n_new_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "counter bank: device alloc failed");
      hipLaunchKernelGGL(particle_update_kernel, dim3((N + PU_T - 1) / PU_T), dim3(PU_T), pu_stage_bytes, ctx->stream, N, P, r.draws.p, r.lse.p,
                         (const int32_t*)nullptr, (const int32_t*)nullptr, (const double*)nullptr, cur_b, r.pchoice.p,
                         s->w.p, n_new_ctr, r.new_slots.p, r.pnewpos.p, (w_by_first_block && bi == 0) ? 1 : 0,
                         (const int32_t*)nullptr, 0, pu_stage_stride, (const double*)nullptr, 0);
      HIPCHK(ctx, hipGetLastError());
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
      if (fuse_final || split_final) s->lazy_req = SweepState::LazyReq{true, split_final ? emit_rows : nullptr};
      rc = eval_node(ctx, bi, 0, il, excl, seed, sweep_idx, P, r.lse.p, r.draws.p, nullptr, nullptr, bi == ctx->timed_block);
      s->lazy_req = SweepState::LazyReq();
      if (rc) return rc;
      lazy = s->lazy_out;
      s->lazy_out.valid = false;
      if (bi == ctx->timed_block) {  // (the elapsed time of the launch is read at the end of the call: no synchronisation here)
        hot_timed = true;
        ctx->timing.hot_kernel_launches += 1;
      }
      ProfScope ps(ctx, "particle_update");
      if (fuse_final) {
        DISPATCH_PMAX(P, hipLaunchKernelGGL(particle_update_final_kernel<PMAX>, grid1(N), dim3(256), 0, ctx->stream, N, P, r.draws.p,
                                            r.lse.p, (const int32_t*)nullptr, (const int32_t*)nullptr, (const double*)nullptr, cur_b,
                                            r.pchoice.p, s->w.p, (w_by_first_block && bi == 0) ? 1 : 0, use_mh, cur_base, seed,
                                            sweep_idx, s->row_offset + ctx->active_begin, s->chosen.p, s->logml_acc.p, s->logml.p,
                                            (const int32_t*)nullptr, lazy.valid ? 1 : 0, w_uni));
        final_fused = true;
        { const int rcl = lazy_draws(); if (rcl) return rcl; }
      } else {
      n_new_ctr = fresh_counter(ctx);
      if (!n_new_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "counter bank: device alloc failed");
      // (the first block of several: its weights are the block's log marginal for every particle of a row — not stored)
      const bool skip_w = bi == 0 && w_by_first_block && bi < n_blocks - 1 && !split_final && !emit_rows && !no_w_uni;
      hipLaunchKernelGGL(particle_update_kernel, dim3((N + PU_T - 1) / PU_T), dim3(PU_T), pu_stage_bytes, ctx->stream, N, P, r.draws.p, r.lse.p,
                         (const int32_t*)nullptr, (const int32_t*)nullptr, (const double*)nullptr, cur_b, r.pchoice.p,
                         s->w.p, n_new_ctr, r.new_slots.p, r.pnewpos.p, (w_by_first_block && bi == 0) ? 1 : 0, emit_rows, split_final ? 1 : 0,
                         pu_stage_stride, w_uni, skip_w ? 1 : 0);
      HIPCHK(ctx, hipGetLastError());
      if (skip_w)
        w_uni = r.lse.p;
      else if (!split_final)
        w_uni = nullptr;  // (every row's weights are in w now)
      if (split_final) {
        DISPATCH_PMAX(P, hipLaunchKernelGGL(particle_update_final_kernel<PMAX>, grid1(N), dim3(256), 0, ctx->stream, N, P, r.draws.p,
                                            r.lse.p, (const int32_t*)nullptr, (const int32_t*)nullptr, (const double*)nullptr, cur_b,
                                            r.pchoice.p, s->w.p, (w_by_first_block && bi == 0) ? 1 : 0, use_mh, cur_base, seed,
                                            sweep_idx, s->row_offset + ctx->active_begin, s->chosen.p, s->logml_acc.p, s->logml.p,
                                            emit_rows, lazy.valid ? 1 : 0, w_uni));
        final_only_rows = emit_rows;
        { const int rcl = lazy_draws(); if (rcl) return rcl; }
      }
      }
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
      int32_t* slot_item = scratch<int32_t>(ctx, NP);
      if (!slot_item) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      unsigned int n_items = 0;
      int32_t *d_row = nullptr, *d_ctx = nullptr, *d_excl = nullptr;
      static const bool no_fused_items = getenv("PCLEAN_NO_FUSED_CTX_ITEMS") != nullptr;
      if (!no_fused_items) {
        // one pass (ctx_items_kernel): item i = row i with particle 0's context, extra items behind them; room for the extra
        // items from the last sweep's number (a sweep that needs more runs the kernel again)
        ProfScope ps(ctx, "ctx_items");
        for (int attempt = 0;; ++attempt) {
          const int cap = std::max(r.ctx_extra_cap, 4096);
          d_row = scratch<int32_t>(ctx, (size_t)N + cap);
          d_ctx = scratch<int32_t>(ctx, ((size_t)N + cap) * PCLEAN_MAX_CTX);
          d_excl = scratch<int32_t>(ctx, (size_t)N + cap);
          unsigned int* extra_ctr = fresh_counter(ctx);
          if (!d_row || !d_ctx || !d_excl || !extra_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
          hipLaunchKernelGGL(ctx_items_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, cs, cur_b, r.it_ctx.p, slot_item, d_row, d_ctx,
                             d_excl, extra_ctr, cap);
          unsigned int n_extra = 0;
          PCLEAN_READ_COUNT(ctx, extra_ctr, &n_extra);
          r.ctx_extra_cap = (int)std::min<size_t>((size_t)n_extra * 2 + 4096, NP);
          if (n_extra <= (unsigned int)cap) {
            n_items = (unsigned int)N + n_extra;
            break;
          }
          if (attempt >= 2) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_sweep: the context items did not fit twice in a row");
        }
      } else {
      int32_t* rep = scratch<int32_t>(ctx, NP);
      int32_t* n_distinct = scratch<int32_t>(ctx, (size_t)N + 1);
      int32_t* off = scratch<int32_t>(ctx, (size_t)N + 1);
      size_t tmp_scan = 0;
      HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, n_distinct, off, N + 1, ctx->stream));
      unsigned char* tmp = scratch<unsigned char>(ctx, tmp_scan);
      if (!rep || !n_distinct || !off || !tmp) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      {
        ProfScope ps(ctx, "ctx_items");
        hipLaunchKernelGGL(gather_ctx_kernel, grid1(NP), dim3(256), 0, ctx->stream, NP, cs, r.it_ctx.p);
        HIPCHK(ctx, hipMemsetAsync(n_distinct + N, 0, sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(ctx_count_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, (int)b.n_ctx, r.it_ctx.p, rep, n_distinct);
        HIPCHK(ctx, hipcub::DeviceScan::ExclusiveSum(tmp, tmp_scan, n_distinct, off, N + 1, ctx->stream));
        PCLEAN_READ_COUNT(ctx, off + N, &n_items);
      }
      d_row = scratch<int32_t>(ctx, n_items);
      d_ctx = scratch<int32_t>(ctx, (size_t)n_items * PCLEAN_MAX_CTX);
      d_excl = scratch<int32_t>(ctx, n_items);
      if (!d_row || !d_ctx || !d_excl) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      hipLaunchKernelGGL(ctx_fill_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, (int)b.n_ctx, r.it_ctx.p, rep, off, cur_b, slot_item,
                         d_row, d_ctx, d_excl);
      }
      double* lse_item = scratch<double>(ctx, n_items);
      int32_t* draws_item = scratch<int32_t>(ctx, (size_t)n_items * P);
      if (!lse_item || !draws_item) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
      il = ItemList{(int)n_items, d_row, d_ctx, nullptr, nullptr};
      if (fuse_final || split_final) s->lazy_req = SweepState::LazyReq{true, split_final ? emit_rows : nullptr};
      rc = eval_node(ctx, bi, 0, il, d_excl, seed, sweep_idx, P, lse_item, draws_item, nullptr, nullptr, bi == ctx->timed_block);
      s->lazy_req = SweepState::LazyReq();
      if (rc) return rc;
      if (bi == ctx->timed_block) {
        hot_timed = true;
        ctx->timing.hot_kernel_launches += 1;
      }
      lazy = s->lazy_out;
      s->lazy_out.valid = false;
      lazy.args.slot_item = slot_item;
      ProfScope ps(ctx, "particle_update");
      if (fuse_final) {
        DISPATCH_PMAX(P, hipLaunchKernelGGL(particle_update_final_kernel<PMAX>, grid1(N), dim3(256), 0, ctx->stream, N, P,
                                            (const int32_t*)nullptr, (const double*)nullptr, slot_item, draws_item, lse_item, cur_b,
                                            r.pchoice.p, s->w.p, (w_by_first_block && bi == 0) ? 1 : 0, use_mh, cur_base, seed,
                                            sweep_idx, s->row_offset + ctx->active_begin, s->chosen.p, s->logml_acc.p, s->logml.p,
                                            (const int32_t*)nullptr, lazy.valid ? 1 : 0, w_uni));
        final_fused = true;
        { const int rcl = lazy_draws(); if (rcl) return rcl; }
      } else {
      n_new_ctr = fresh_counter(ctx);
      if (!n_new_ctr) return pclean_fail(ctx, PCLEAN_ERR_HIP, "counter bank: device alloc failed");
      hipLaunchKernelGGL(particle_update_kernel, dim3((N + PU_T - 1) / PU_T), dim3(PU_T), 0, ctx->stream, N, P, (const int32_t*)nullptr,
                         (const double*)nullptr, slot_item, draws_item, lse_item, cur_b, r.pchoice.p, s->w.p,
                         n_new_ctr, r.new_slots.p, r.pnewpos.p, (w_by_first_block && bi == 0) ? 1 : 0, emit_rows, split_final ? 1 : 0, 0,
                         w_uni, 0);
      if (!split_final) w_uni = nullptr;  // (every row's weights are in w now)
      if (split_final) {
        DISPATCH_PMAX(P, hipLaunchKernelGGL(particle_update_final_kernel<PMAX>, grid1(N), dim3(256), 0, ctx->stream, N, P,
                                            (const int32_t*)nullptr, (const double*)nullptr, slot_item, draws_item, lse_item, cur_b,
                                            r.pchoice.p, s->w.p, (w_by_first_block && bi == 0) ? 1 : 0, use_mh, cur_base, seed,
                                            sweep_idx, s->row_offset + ctx->active_begin, s->chosen.p, s->logml_acc.p, s->logml.p,
                                            emit_rows, lazy.valid ? 1 : 0, w_uni));
        final_only_rows = emit_rows;
        { const int rcl = lazy_draws(); if (rcl) return rcl; }
      }
      }
    }
    // ---- particles that proposed a NEW referent: sample the new row's contents
    if (!prefetch_started) {  // (this block's kernels are queued: the host has time for the later blocks' table refresh)
      const int rcp = start_prefetch();
      if (rcp) return rcp;
    }
    unsigned int n_new = 0;
    if (!final_fused) PCLEAN_READ_COUNT(ctx, n_new_ctr, &n_new);  // (fused with the final choice: nothing is sampled now, nobody asks)
    const bool degenerate_rows = (n_new & 0x80000000u) != 0u;  // some row's block marginal is -inf (particle_update_kernel)
    n_new &= 0x7fffffffu;
    r.n_new = final_fused ? -1 : (int)n_new;
    r.lazy_new = final_fused || (bi == n_blocks - 1 && !eager_all && (!drawable || emit_rows != nullptr));
    if (r.lazy_new && !emit_rows) n_new = 0;  // nothing sampled now
    if (emit_rows) r.n_new = -1;  // (the list held the flagged rows' slots only: whether anybody proposed a NEW referent is not known)
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
        if (!rc) rc = gauss_prior();
        if (rc) return rc;
      }
      if (drawable) {
        if (!split_final) materialise_w();  // (split: the flagged rows' weights were stored, nobody looks at the others')
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
      if (!rc) rc = gauss_prior();
      if (rc) return rc;
    }
    ctx->prior_mode = false;
    // (pnewpos is only read where pchoice == NEW, so it needs no initialisation when nobody proposed one)

    // ---- resampling between blocks (row_inference.jl:152-155)
    const int grp_here = b.group >= 0 ? b.group : bi;
    const int grp_next = bi + 1 < n_blocks ? (ctx->block[bi + 1].group >= 0 ? ctx->block[bi + 1].group : bi + 1) : -2;
    // After the FIRST block of a sweep every particle of a row carries the same weight — the block's log marginal, shared by
    // the particles (no context yet to tell them apart) — unless a dummy-value correction or a prior-mode likelihood touched
    // individual particles: the effective sample size is exactly P, maybe_resample (row_inference.jl:87-105) keeps every
    // particle and adds 0 to the log-ML estimate (a row whose marginal is -inf has total weight 0 and IS resampled: those
    // rows are reported through bit 31 of the NEW-slot counter).  Two kernels over all rows that can only ever do nothing are
    // not launched.
    const bool equal_weights = bi == 0 && w_by_first_block && !has_ctx && !prior_mode && !(drawable && n_new > 0) && !degenerate_rows;
    static const bool always_resample = getenv("PCLEAN_ALWAYS_RESAMPLE") != nullptr;
    if (!use_mh && bi < n_blocks - 1 && grp_here != grp_next && (!equal_weights || always_resample)) {  // (the slots of one model block: no resampling in between)
      ProfScope ps(ctx, "resample");
      materialise_w();
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
        if (s->run[k].plocals_on) arrs.pairs[arrs.n_pairs++] = reinterpret_cast<int64_t*>(s->run[k].plocals.p);
      }
      hipLaunchKernelGGL(apply_ancestors_kernel, grid1(N), dim3(256), 0, ctx->stream, N, P, s->ancestors.p, n_arr, arrs, s->w.p,
                         s->did.p, s->logml_inc.p, s->logml_acc.p);
    }
  }

  if (prefetching) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->pre_join, 0));  // (no later block took it)
  // ---- final choice + per-block outputs: one pass per block, ordered compaction of the rows that moved /
  // got a new referent (hipcub select keeps ascending row order), ONE read-back of the counts
  {
    ProfScope ps(ctx, "final_choice_and_outputs");
    if (!final_fused && !final_only_rows) materialise_w();  // (never pending here: a plain update of every row stored them)
    if (!final_fused)
    DISPATCH_PMAX(P, hipLaunchKernelGGL(final_choice_kernel<PMAX>, grid1(N), dim3(256), 0, ctx->stream, N, P, s->w.p,
                                        (size_t)1, (size_t)N, use_mh, 1, cur_base, seed, sweep_idx,
                                        s->row_offset + ctx->active_begin, s->chosen.p, (double*)nullptr,
                                        s->logml_acc.p, s->logml.p, final_only_rows));
    for (int bi = 0; bi < n_blocks; ++bi) {  // deferred new-row contents of the last block (chosen particles only)
      BlockRun& r = s->run[bi];
      Block& bb = ctx->block[bi];
      if (bb.is_score || !r.lazy_new || r.n_new == 0) continue;
      ProfScope ps2(ctx, "new_row_sampling_chosen");
      const int nn = (int)bb.nodes.size();
      const int32_t* cur_b = cur_base + (size_t)bi * cur_ld;
      unsigned int* cnt_ctr = fresh_counter(ctx);
      hipLaunchKernelGGL(chosen_new_kernel, grid1(N), dim3(256), 0, ctx->stream, N, s->chosen.p, r.pchoice.p, cnt_ctr,
                         r.new_slots.p, r.pnewpos.p);
      unsigned int cnt = 0;
      PCLEAN_READ_COUNT(ctx, cnt_ctr, &cnt);
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
    const int nb_wg = (N + 255) / 256;
    unsigned int* wg_cnt = scratch<unsigned int>(ctx, (size_t)2 * n_blocks * nb_wg);
    if (!wg_cnt) return pclean_fail(ctx, PCLEAN_ERR_HIP, "scratch alloc failed");
    TailLists tl{};
    // the tables' delta reference counts start at zero: one kernel over every root table instead of a memset apiece
    {
      ZeroList zl{};
      for (int bi = 0; bi < n_blocks; ++bi) {
        if (ctx->block[bi].is_score) continue;
        CandTable& rt = ctx->cand[ctx->block[bi].nodes[0].table];
        bool seen = false;
        for (int k = 0; k < zl.n; ++k) seen |= zl.p[k] == (void*)rt.stats.p;
        if (!seen) zero_list_add(ctx, zl, rt.stats.p, (size_t)std::max(rt.n_rows, 1) * 8);
      }
      const int rcz = zero_list_flush(ctx, zl);
      if (rcz) return rcz;
    }
    for (int bi = 0; bi < n_blocks; ++bi) {
      BlockRun& r = s->run[bi];
      Block& bb = ctx->block[bi];
      bb.locals_host.clear();
      r.locals_rows = 0;
      if (bb.is_score) continue;
      const int32_t* cur_b = cur_base + (size_t)bi * cur_ld;
      CandTable& rt = ctx->cand[bb.nodes[0].table];
      // (an LDS histogram per workgroup: a few rows take most of the moves — 28 true measures for 1M records.  It covers the
      // rows below the table's high-water mark only: nobody refers to, or chooses, the spare capacity behind it, and zeroing
      // + flushing the counters of 8 240 rows in each of 4 000 workgroups was 0.05 ms per sweep)
      const int rows_in_use = (rt.n_used > 0 && rt.n_used <= rt.n_rows) ? std::min(rt.n_rows, (rt.n_used + 63) & ~63) : rt.n_rows;
      static const bool no_hist = getenv("PCLEAN_NO_HIST") != nullptr;  // (measurement switch: plain atomics for every table)
      const int hist_rows = (rt.n_rows <= 12288 && !no_hist) ? rows_in_use : 0;  // (48 KB of LDS per workgroup at most; larger tables: plain atomics)
      hipLaunchKernelGGL(finalize_block_kernel, grid1(N), dim3(256), (size_t)hist_rows * sizeof(int32_t), ctx->stream, N,
                         s->chosen.p, r.pchoice.p, r.pnewpos.p, cur_b, r.choice.p, r.chosen_newpos.p,
                         (unsigned long long*)rt.stats.p, hist_rows, r.moved_flag.p, r.new_flag.p,
                         wg_cnt + (size_t)(2 * bi) * nb_wg, wg_cnt + (size_t)(2 * bi + 1) * nb_wg);
      tl.flag[2 * bi] = r.moved_flag.p;
      tl.list[2 * bi] = r.moved_list.p;
      tl.flag[2 * bi + 1] = r.new_flag.p;
      tl.list[2 * bi + 1] = r.new_list.p;
      if (choice)
        HIPCHK(ctx, hipMemcpyAsync(choice + (size_t)bi * N, r.choice.p, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
      if (!bb.node_gauss.empty() && bb.node_gauss[0] >= 0 && bb.gauss[bb.node_gauss[0]].n_locals > 0) {
        GaussDev gd;
        int rc = build_gauss_dev(ctx, bb.gauss[bb.node_gauss[0]], &rt, gd);
        if (rc) return rc;
        if (r.locals.alloc((size_t)N * 2)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
        if (r.plocals_on)  // prior proposals: what the chosen particle sampled (or kept)
          hipLaunchKernelGGL(locals_pick_kernel, grid1(N), dim3(256), 0, ctx->stream, N, s->chosen.p, r.plocals.p, r.locals.p);
        else
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
    {  // (a scoring block's two slots: their workgroup counts were never written — zero them so that the totals read 0)
      for (int bi = 0; bi < n_blocks; ++bi)
        if (ctx->block[bi].is_score)
          HIPCHK(ctx, hipMemsetAsync(wg_cnt + (size_t)(2 * bi) * nb_wg, 0, (size_t)2 * nb_wg * sizeof(unsigned int), ctx->stream));
      hipLaunchKernelGGL(tail_scan_kernel, dim3(2 * n_blocks), dim3(1024), 0, ctx->stream, nb_wg, wg_cnt, s->tail_counts.p);
      hipLaunchKernelGGL(tail_scatter_kernel, dim3(nb_wg, 2 * n_blocks), dim3(256), 0, ctx->stream, N, nb_wg, tl, wg_cnt);
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
    const Block& b0 = ctx->block[(ctx->timed_block >= 0 && ctx->timed_block < n_blocks && !ctx->block[ctx->timed_block].is_score) ? ctx->timed_block : 0];
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
  int rc = d2h_small(ctx, s->h_counts, s->tail_counts.p, 2 * s->last_blocks * sizeof(int32_t));
  if (rc) return rc;
  const int rco = queue_over_copy(ctx);
  if (rco) return rco;
  s->h_counts[3 * PCLEAN_MAX_BLOCKS] = 0;
  if (s->dummy_used) rc = d2h_small(ctx, s->h_counts + 3 * PCLEAN_MAX_BLOCKS, s->dummy_ctr.p + 1, sizeof(int32_t), s->h_counts);
  if (rc) return rc;
  return d2h_flush(ctx);  // (everything a caller queued before this call rides along)
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
    if (s->gate_timed) {  // the gate of the new-row branch on the root's groups: it computes what group_desc_kernel used to
      float mg = 0;
      HIPCHK(ctx, hipEventElapsedTime(&mg, s->evg0, s->evg1));
      ctx->timing.hot_kernel_ms += mg;
    }
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

extern "C" int pclean_set_cur_locals(pclean_ctx* ctx, int32_t block_id, const int32_t* locals, int32_t n_rows) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS || !ctx->block[block_id].valid)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_cur_locals: bad arguments");
  Block& b = ctx->block[block_id];
  if (!locals || n_rows <= 0) {
    b.cur_locals.release();
    b.cur_locals_rows = 0;
    return PCLEAN_OK;
  }
  if (n_rows != ctx->n_rows) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_cur_locals: one pair per observed row (%d), got %d", ctx->n_rows, n_rows);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (b.cur_locals.alloc((size_t)n_rows * 2)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  HIPCHK(ctx, hipMemcpyAsync(b.cur_locals.p, locals, (size_t)n_rows * 8, hipMemcpyHostToDevice, ctx->stream));
  PCLEAN_SYNC(ctx);  // (the caller's array may go away)
  b.cur_locals_rows = n_rows;
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

extern "C" int pclean_set_timed_block(pclean_ctx* ctx, int32_t block_id) {
  if (!ctx || block_id < 0 || block_id >= PCLEAN_MAX_BLOCKS) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_set_timed_block: bad block");
  ctx->timed_block = block_id;
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
                                                (const double*)nullptr, (double*)nullptr, (const int32_t*)nullptr));
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
