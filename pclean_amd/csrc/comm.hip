// Inter-GPU exchange of the CRP sufficient statistics through RCCL, for hosts that do not have
// torch.distributed (pclean_comm_* / pclean_allreduce_stats of include/pclean_hip.h).  RCCL is resolved
// with dlopen at the first call: a process that already carries an RCCL (PyTorch) keeps using that
// copy, and single-GPU users never load it.
#include <dlfcn.h>

#include <cstring>

#include "ctx.h"

namespace {

// the handful of RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE)
struct UniqueId {
  char internal[128];
};
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*all_gather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*comm_destroy_t)(void*);
typedef const char* (*get_error_string_t)(int);
const int kNcclInt64 = 4;  // ncclInt64
const int kNcclInt32 = 2;  // ncclInt32
const int kNcclSum = 0;    // ncclSum

struct Rccl {
  void* handle = nullptr;
  get_unique_id_t get_unique_id = nullptr;
  comm_init_rank_t comm_init_rank = nullptr;
  all_reduce_t all_reduce = nullptr;
  all_gather_t all_gather = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  get_error_string_t error_string = nullptr;
};

Rccl* rccl(pclean_ctx* ctx) {
  static Rccl r;
  if (r.handle) return &r;
  // Never bring a second RCCL into the process: if one is already loaded (PyTorch ships its own
  // librccl.so) ncclGetUniqueId is visible through the global scope — use that copy.
  if (dlsym(RTLD_DEFAULT, "ncclGetUniqueId")) {
    r.handle = dlopen(nullptr, RTLD_NOW);
  } else {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
  }
  if (!r.handle) {
    pclean_fail(ctx, PCLEAN_ERR_STATE, "RCCL not found (dlopen librccl.so): %s", dlerror());
    return nullptr;
  }
  r.get_unique_id = (get_unique_id_t)dlsym(r.handle, "ncclGetUniqueId");
  r.comm_init_rank = (comm_init_rank_t)dlsym(r.handle, "ncclCommInitRank");
  r.all_reduce = (all_reduce_t)dlsym(r.handle, "ncclAllReduce");
  r.all_gather = (all_gather_t)dlsym(r.handle, "ncclAllGather");
  r.comm_destroy = (comm_destroy_t)dlsym(r.handle, "ncclCommDestroy");
  r.error_string = (get_error_string_t)dlsym(r.handle, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
    pclean_fail(ctx, PCLEAN_ERR_STATE, "RCCL symbols missing in librccl.so");
    r.handle = nullptr;
    return nullptr;
  }
  return &r;
}

int rccl_fail(pclean_ctx* ctx, Rccl* r, const char* what, int rc) {
  return pclean_fail(ctx, PCLEAN_ERR_HIP, "%s failed: %s", what, r->error_string ? r->error_string(rc) : "rccl error");
}

}  // namespace

extern "C" int pclean_comm_unique_id(pclean_ctx* ctx, unsigned char id_out[128]) {
  if (!ctx || !id_out) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_comm_unique_id: bad arguments");
  Rccl* r = rccl(ctx);
  if (!r) return PCLEAN_ERR_STATE;
  UniqueId id;
  const int rc = r->get_unique_id(&id);
  if (rc) return rccl_fail(ctx, r, "ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, 128);
  return PCLEAN_OK;
}

extern "C" int pclean_comm_init(pclean_ctx* ctx, int32_t n_ranks, int32_t rank, const unsigned char id_in[128]) {
  if (!ctx || n_ranks < 1 || rank < 0 || rank >= n_ranks || !id_in)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_comm_init: bad arguments");
  if (ctx->rccl_comm) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_comm_init: communicator already initialised");
  Rccl* r = rccl(ctx);
  if (!r) return PCLEAN_ERR_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  UniqueId id;
  memcpy(id.internal, id_in, 128);
  void* comm = nullptr;
  const int rc = r->comm_init_rank(&comm, n_ranks, id, rank);
  if (rc) return rccl_fail(ctx, r, "ncclCommInitRank", rc);
  ctx->rccl_comm = comm;
  ctx->comm_ranks = n_ranks;
  ctx->comm_rank = rank;
  return PCLEAN_OK;
}

extern "C" int pclean_allreduce_stats(pclean_ctx* ctx, int32_t table_id, int64_t* out) {
  if (!ctx || table_id < 0 || table_id >= PCLEAN_MAX_TABLES || !ctx->cand[table_id].valid)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_allreduce_stats: bad arguments");
  if (!ctx->rccl_comm) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_allreduce_stats: call pclean_comm_init first");
  Rccl* r = rccl(ctx);
  if (!r) return PCLEAN_ERR_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  CandTable& t = ctx->cand[table_id];
  if (t.n_rows > 0) {
    const int rc = r->all_reduce(t.stats.p, t.stats.p, (size_t)t.n_rows, kNcclInt64, kNcclSum, ctx->rccl_comm, ctx->stream);
    if (rc) return rccl_fail(ctx, r, "ncclAllReduce", rc);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (out) HIPCHK(ctx, hipMemcpy(out, t.stats.p, (size_t)t.n_rows * 8, hipMemcpyDeviceToHost));
  }
  return PCLEAN_OK;
}

__global__ void stats_pack_kernel(int n, const int64_t* __restrict__ src, int64_t* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// The delta reference counts of several latent tables as ONE int64 vector: packed into a device buffer, summed over
// the ranks with a single RCCL all-reduce (the payload is ~80 KB: latency-bound, so one collective instead of one
// per table), unpacked into every table's stats buffer and copied to the host once.
static int allreduce_stats_queue(pclean_ctx* ctx, int32_t n_tables, const int32_t* table_ids, int32_t local_is_zero, int64_t* out,
                                 size_t* total_out);
extern "C" int pclean_allreduce_stats_fused(pclean_ctx* ctx, int32_t n_tables, const int32_t* table_ids,
                                            int32_t local_is_zero, int64_t* out) {
  size_t total = 0;
  const int rc = allreduce_stats_queue(ctx, n_tables, table_ids, local_is_zero, out, &total);
  if (rc || total == 0) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return PCLEAN_OK;
}
// the same, queued on the library's stream and left there (the device-resident commit of several ranks: commit.hip)
int pclean_comm_allreduce_stats_queue(pclean_ctx* ctx, int32_t n_tables, const int32_t* table_ids, int32_t local_is_zero) {
  size_t total = 0;
  return allreduce_stats_queue(ctx, n_tables, table_ids, local_is_zero, nullptr, &total);
}
// ---- what the collectives moved and how long they took on the device -------------------------------------------------
// which = 0: all-gather, 1: all-reduce.  fold: the pending pair's elapsed time goes into the total (the collective it brackets
// was queued a whole sweep ago when the next one arrives: the wait is not one)
static void comm_stats_fold(pclean_ctx* ctx, int which) {
  pclean_ctx::CommStats& cs = ctx->comm_stats;
  bool& pending = which == 0 ? cs.ag_pending : cs.ar_pending;
  if (!pending) return;
  float ms = 0.0f;
  if (hipEventSynchronize(cs.ev[2 * which + 1]) == hipSuccess &&
      hipEventElapsedTime(&ms, cs.ev[2 * which], cs.ev[2 * which + 1]) == hipSuccess)
    (which == 0 ? cs.ag_us : cs.ar_us) += 1e3 * (double)ms;
  pending = false;
}
static void comm_stats_begin(pclean_ctx* ctx, int which) {
  pclean_ctx::CommStats& cs = ctx->comm_stats;
  comm_stats_fold(ctx, which);
  for (int i = 0; i < 2; ++i)
    if (!cs.ev[2 * which + i] && hipEventCreate(&cs.ev[2 * which + i]) != hipSuccess) cs.ev[2 * which + i] = nullptr;
  if (cs.ev[2 * which] && cs.ev[2 * which + 1]) (void)hipEventRecord(cs.ev[2 * which], ctx->stream);
}
static void comm_stats_end(pclean_ctx* ctx, int which) {
  pclean_ctx::CommStats& cs = ctx->comm_stats;
  if (cs.ev[2 * which] && cs.ev[2 * which + 1] && hipEventRecord(cs.ev[2 * which + 1], ctx->stream) == hipSuccess)
    (which == 0 ? cs.ag_pending : cs.ar_pending) = true;
}
extern "C" int pclean_comm_get_stats(pclean_ctx* ctx, int64_t out[8]) {
  if (!ctx || !out) return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_comm_get_stats: bad arguments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  comm_stats_fold(ctx, 0);
  comm_stats_fold(ctx, 1);
  const pclean_ctx::CommStats& cs = ctx->comm_stats;
  out[0] = (int64_t)cs.ag_calls;
  out[1] = (int64_t)cs.ag_bytes_last;
  out[2] = (int64_t)(cs.ag_us * 1e3);  // ns
  out[3] = (int64_t)cs.ar_calls;
  out[4] = (int64_t)cs.ar_elems_last * 8;
  out[5] = (int64_t)(cs.ar_us * 1e3);
  out[6] = ctx->comm_ranks;
  out[7] = ctx->comm_rank;
  return PCLEAN_OK;
}

// all-gather of words_per_rank int32 words per rank (rank r's words land at recv + r * words_per_rank); queued, no sync
int pclean_comm_allgather_i32(pclean_ctx* ctx, const int32_t* send, int32_t* recv, size_t words_per_rank) {
  if (!ctx->rccl_comm) return pclean_fail(ctx, PCLEAN_ERR_STATE, "all-gather: call pclean_comm_init first");
  Rccl* r = rccl(ctx);
  if (!r) return PCLEAN_ERR_STATE;
  if (!r->all_gather) return pclean_fail(ctx, PCLEAN_ERR_STATE, "ncclAllGather missing in librccl.so");
  comm_stats_begin(ctx, 0);
  const int rc = r->all_gather(send, recv, words_per_rank, kNcclInt32, ctx->rccl_comm, ctx->stream);
  if (rc) return rccl_fail(ctx, r, "ncclAllGather", rc);
  comm_stats_end(ctx, 0);
  ctx->comm_stats.ag_calls += 1;
  ctx->comm_stats.ag_bytes_last = (uint64_t)words_per_rank * 4u;
  return PCLEAN_OK;
}
static int allreduce_stats_queue(pclean_ctx* ctx, int32_t n_tables, const int32_t* table_ids, int32_t local_is_zero, int64_t* out,
                                 size_t* total_out) {
  if (!ctx || n_tables <= 0 || n_tables > PCLEAN_MAX_TABLES || !table_ids)
    return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_allreduce_stats_fused: bad arguments");
  if (!ctx->rccl_comm) return pclean_fail(ctx, PCLEAN_ERR_STATE, "pclean_allreduce_stats_fused: call pclean_comm_init first");
  Rccl* r = rccl(ctx);
  if (!r) return PCLEAN_ERR_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t total = 0;
  for (int i = 0; i < n_tables; ++i) {
    if (table_ids[i] < 0 || table_ids[i] >= PCLEAN_MAX_TABLES || !ctx->cand[table_ids[i]].valid)
      return pclean_fail(ctx, PCLEAN_ERR_ARG, "pclean_allreduce_stats_fused: bad table id");
    total += (size_t)ctx->cand[table_ids[i]].n_rows;
  }
  *total_out = total;
  if (total == 0) return PCLEAN_OK;
  if (ctx->stats_pack.alloc(total)) return pclean_fail(ctx, PCLEAN_ERR_HIP, "device alloc failed");
  size_t off = 0;
  if (local_is_zero) HIPCHK(ctx, hipMemsetAsync(ctx->stats_pack.p, 0, total * sizeof(int64_t), ctx->stream));
  for (int i = 0; i < n_tables && !local_is_zero; ++i) {
    const CandTable& t = ctx->cand[table_ids[i]];
    if (t.n_rows)
      hipLaunchKernelGGL(stats_pack_kernel, dim3((t.n_rows + 255) / 256), dim3(256), 0, ctx->stream, t.n_rows, t.stats.p,
                         ctx->stats_pack.p + off);
    off += (size_t)t.n_rows;
  }
  comm_stats_begin(ctx, 1);
  const int rc = r->all_reduce(ctx->stats_pack.p, ctx->stats_pack.p, total, kNcclInt64, kNcclSum, ctx->rccl_comm, ctx->stream);
  if (rc) return rccl_fail(ctx, r, "ncclAllReduce", rc);
  comm_stats_end(ctx, 1);
  ctx->comm_stats.ar_calls += 1;
  ctx->comm_stats.ar_elems_last = (uint64_t)total;
  off = 0;
  for (int i = 0; i < n_tables; ++i) {
    CandTable& t = ctx->cand[table_ids[i]];
    if (t.n_rows)
      hipLaunchKernelGGL(stats_pack_kernel, dim3((t.n_rows + 255) / 256), dim3(256), 0, ctx->stream, t.n_rows,
                         ctx->stats_pack.p + off, t.stats.p);
    off += (size_t)t.n_rows;
  }
  if (out) HIPCHK(ctx, hipMemcpyAsync(out, ctx->stats_pack.p, total * 8, hipMemcpyDeviceToHost, ctx->stream));
  return PCLEAN_OK;
}

extern "C" int pclean_comm_destroy(pclean_ctx* ctx) {
  if (!ctx) return PCLEAN_ERR_ARG;
  if (!ctx->rccl_comm) return PCLEAN_OK;
  Rccl* r = rccl(ctx);
  if (r) (void)r->comm_destroy(ctx->rccl_comm);
  ctx->rccl_comm = nullptr;
  ctx->comm_ranks = 0;
  ctx->comm_rank = 0;
  for (hipEvent_t& e : ctx->comm_stats.ev)
    if (e) {
      (void)hipEventDestroy(e);
      e = nullptr;
    }
  ctx->comm_stats.ag_pending = ctx->comm_stats.ar_pending = false;
  return PCLEAN_OK;
}
