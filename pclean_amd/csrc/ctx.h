// Internal state of libpclean_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pclean_hip.h"

#define PCLEAN_MAX_TABLES 64
#define PCLEAN_MAX_BLOCKS 16

// RAII-less device buffer: freed by ctx destroy / reassign.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    if (count <= n && p) return 0;
    release();
    if (count == 0) return 0;
    if (hipMalloc((void**)&p, count * sizeof(T)) != hipSuccess) {
      p = nullptr;
      n = 0;
      return -1;
    }
    n = count;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// Page-locked staging area owned by the library (grow-only).  Host arrays of the caller are copied through it instead
// of being handed to hipMemcpy[Async]: the runtime page-locks a caller's pageable pages for the DMA and keeps that
// registration; when the caller later frees them (NumPy returns large arrays to the OS) the kernel driver's MMU
// notifier evicts the process's GPU queues to drop the mapping and restores them lazily — measured as 5-27 ms on
// the next GPU operation, whatever its size (a 48-byte table upload after a latent sweep had uploaded 16 MB of evidence).
struct HostStage {
  unsigned char* p = nullptr;
  size_t n = 0, used = 0;
  void rewind() { used = 0; }
  void* take(size_t bytes) {  // null when the area cannot hold it (grow() first)
    const size_t at = (used + 255) & ~(size_t)255;
    if (at + bytes > n) return nullptr;
    used = at + bytes;
    return p + at;
  }
  int grow(size_t bytes) {  // only between uses (nothing in flight reads the old area)
    if (bytes <= n) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc((void**)&p, want, hipHostMallocDefault) != hipSuccess) {
      p = nullptr;
      return -1;
    }
    n = want;
    return 0;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = used = 0;
  }
};

extern uint64_t g_pclean_version;  // bumped whenever a table is (re)uploaded; keys the leaf caches

struct PairTable {
  bool valid = false;
  uint64_t version = 0;
  int32_t n_obs = 0, n_lat = 0;
  int32_t elem_bytes = 1;     // 1: uint8 distances, 2: uint16
  DevBuf<uint8_t> d;          // [n_obs][n_lat] * elem_bytes
  DevBuf<uint16_t> lat_len;   // word length (characters) of each latent value
  DevBuf<int32_t> obs_ids;    // AddTypos tables built on the device: pool string of every observed value
  int32_t dist_mode = PCLEAN_DIST_DL;  // flavour the table was built with (dummy_dev.h scores drawn strings with it)
  int32_t max_lat_len = 0, max_obs_len = 0;
  double mean_lat_len = 0.0;  // AddTypos tables built on the device: mean length of the latent strings
};

struct CandTable {
  bool valid = false;
  uint64_t version = 0;       // any re-upload (counts / priors)
  uint64_t cols_version = 0;  // re-upload of the value columns
  bool is_options = false;
  int32_t n_rows = 0, n_cols = 0;
  // rows [n_used, n_rows) have never held a row (the spare capacity of the device-resident commit: count 0, weight
  // exactly 0): a scan need not look at them.  0: unknown (every row may be in use)
  int32_t n_used = 0;
  DevBuf<int32_t> cols;       // column-major [n_cols][n_rows]
  DevBuf<int64_t> counts;     // FK tables
  DevBuf<double> logc_full;   // FK: log(count-discount); options: logp
  DevBuf<double> logc_m1;     // FK only
  double scal[4] = {0, 0, 0, 0};  // logden_full, logden_m1, lognew_n, lognew_nm1
  std::vector<int32_t> h_vals;   // option tables: the values as last uploaded (pclean_set_options)
  bool is_options_1col = false;
  std::vector<int64_t> h_counts;
  std::vector<double> h_logc_full, h_logc_m1;
  DevBuf<int64_t> stats;      // delta reference counts of last sweep
  double strength = 1.0, discount = 0.0;  // Pitman-Yor parameters of the last upload
  double logc_max = 0.0;      // FK tables: max over rows of log(count - discount) (-inf for an empty table)
  // the last change of the value columns, when a device commit made it: rows [cols_delta_rows[0 .. cols_delta_n)) are what
  // differs between cols_version and cols_delta_base (-1: unknown, e.g. a re-upload)
  uint64_t cols_delta_base = 0;
  int32_t cols_delta_n = -1;
  const int32_t* cols_delta_rows = nullptr;
  // ... or a re-upload of the same shape that differs from the previous one in a few rows (pclean_set_table compares with
  // h_cols, the columns as last uploaded: a latent sub-batch's host commit creates / collects a handful of rows)
  std::vector<int32_t> h_cols;
  DevBuf<int32_t> upload_delta_rows;
  bool h_mirror_stale = false;  // the device arrays moved on without the host mirrors (pclean_commit_device)
  bool h_cols_stale = false;    // ... the value columns among them (a commit that created rows; pclean_commit_pull_table
                                // brings h_cols up to date again)
  // The chain of the host re-uploads' deltas: entry i leads from cols_version `base` to `next` by rewriting `rows`, entry
  // i + 1 starts where entry i ends, the last one ends at cols_version.  A consumer built from ANY version on the chain (the
  // observed class's compact byte tables, last refreshed before the 32 sub-batch uploads of a latent class sweep) refreshes
  // the union of the rows since then instead of rebuilding (cols_union, eval.hip: try_fast_root).
  // the device commit's own delta (rows it wrote, a device list that lives until the table's next commit): the chain's link
  // before the first host re-upload that follows it
  uint64_t commit_delta_base = 0, commit_delta_next = 0;
  int32_t commit_delta_n = -1;
  const int32_t* commit_delta_rows = nullptr;
  struct DeltaEntry {
    uint64_t base, next;
    std::vector<int32_t> rows;
    std::vector<uint64_t> masks;  // bit min(c, 63): column c of the row changed (a Place that moved to another of a hundred
                                  // identical County rows rewrites a reference column that no byte table is built from)
  };
  std::vector<DeltaEntry> delta_log;
  DevBuf<int32_t> union_rows;  // memo of the last union asked for: rows differing between union_base and union_head
  uint64_t union_base = 0, union_head = 0, union_mask = 0;
  int32_t union_n = -1;
  double h_lse = 0.0;         // options: log-sum of logp (+1e-9), valid for version h_lse_ver (eval.hip: subtree_ub)
  uint64_t h_lse_ver = 0;
};

struct FnTable {
  bool valid = false;
  int32_t n_a = 0, n_b = 0;
  DevBuf<int32_t> fn;
};

struct MeanTable {
  bool valid = false;
  int32_t n = 0;
  DevBuf<double> v;
};

struct ScoreTerm {
  int32_t obs_col, pair_table, val_block, val_col, key_block, key_col, nopt_fn, other_val;
};

struct Block {
  bool valid = false;
  uint64_t version = 0;              // bumped by every pclean_load_block of this block
  int32_t group = -1;                // pclean_set_block_group (-1: its own group)
  bool is_score = false;             // no reference slot: only scores observed choices (flights Obs block 3)
  std::vector<ScoreTerm> score_terms;
  int32_t prob_fn = -1, prob_a_block = -1, prob_a_col = -1, prob_b_block = -1, prob_b_col = -1;
  std::vector<pclean_gauss> gauss;   // Gaussian terms of this block's nodes
  std::vector<int32_t> node_gauss;   // per node: index into gauss, -1 none
  std::vector<pclean_node> nodes;
  std::vector<pclean_term> terms;
  std::vector<int32_t> children;
  std::vector<int32_t> colmap;
  int32_t n_ctx = 0;
  int32_t ctx_src_block[PCLEAN_MAX_CTX] = {-1, -1};
  int32_t ctx_src_col[PCLEAN_MAX_CTX] = {-1, -1};
  DevBuf<pclean_term> d_terms;
  // per-sweep work buffers live in Sweep state (sweep.hip)
  std::vector<DevBuf<double>> leaf_cache;  // per node: marginal per unique observed value
  // cacheable option lists (enum_kernels.hip: leaf_coarse_*): per unique observed value the maximum, the fixed-point
  // total and the inclusive prefix at every 256th option
  std::vector<DevBuf<double>> leaf_m;
  std::vector<DevBuf<uint64_t>> leaf_U, leaf_coarse;
  std::vector<DevBuf<uint64_t>> leaf_udummy;  // fixed-point weight of the ProposalDummyValue option per observed value
  std::vector<int> leaf_drawable;  // per node: -1 unknown, 0 the dummy cannot be drawn for any loaded row, 1 it can
  std::vector<int32_t> new_rows_host, new_vals_host, locals_host;
  DevBuf<int32_t> cur_locals;  // [n_rows][2] current own choices of every observed row (pclean_set_cur_locals), null: none
  int cur_locals_rows = 0;
  std::vector<int32_t> moved_rows_host, moved_choice_host;  // rows whose referent changed in the last sweep
};

struct pclean_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // string pool
  int32_t n_strings = 0;
  DevBuf<uint16_t> sym;
  DevBuf<int64_t> off;
  std::vector<int64_t> h_off;
  int32_t n_symbols = 0;

  // observed columns
  int32_t n_rows = 0, n_cols = 0;
  DevBuf<int32_t> obs;  // [n_cols][n_rows]
  uint64_t obs_version = 0;  // bumped by every pclean_load_columns (keys the static per-row tuple ids, eval.hip)
  std::vector<char> col_has_missing;  // per observed column: some row holds an explicitly missing value (-1)
  DevBuf<int32_t> iota; // identity column for per-unique-value leaf caches
  int32_t n_xcols = 0;
  DevBuf<double> xnum;  // numeric observed columns [n_xcols][n_rows]
  MeanTable mean[PCLEAN_MAX_TABLES];
  int32_t n_prob = 0;
  DevBuf<double> prob_same, prob_diff, logn;  // log1p(-p), log(p), log(n)
  std::vector<double> h_prob_same, h_prob_diff, h_logn;

  // density tables
  int32_t max_r = -1, max_d = -1, max_len = -1;
  std::vector<double> h_nb, h_logl;
  DevBuf<double> nb, logl;
  // letter model of random(StringPrior) + pool symbol of every letter (pclean_set_lm_tables)
  bool lm_valid = false;
  DevBuf<double> lm_init, lm_trans;
  DevBuf<uint16_t> letter_sym;
  DevBuf<double> atd;  // [max_len + 1][max_d + 1]: nb[(L+4)/5][d] - logl[L] d - log(26)/2 d in that fp64 order (add_typos.jl:61-63)

  PairTable pair[PCLEAN_MAX_TABLES];
  CandTable cand[PCLEAN_MAX_TABLES];
  FnTable fn[PCLEAN_MAX_TABLES];
  Block block[PCLEAN_MAX_BLOCKS];

  pclean_timing timing = {};
  pclean_root_stats root_stats = {};
  int64_t cur_stride = 0;  // pclean_set_cur_stride
  // device-resident current referents [n_blocks][n_rows] (pclean_set_cur; kept up to date by pclean_commit_device)
  DevBuf<int32_t> dev_cur;
  int32_t dev_cur_blocks = 0;
  bool dev_cur_valid = false;
  // evidence CSR of a latent class built on the device (pclean_build_evidence): the observed rows ordered by the latent row
  // they refer to and their per-row context values, kept for the sub-batches of the class sweep (pclean_sweep_latent_resident)
  DevBuf<int32_t> ev_res_rows, ev_res_ctx;
  int32_t ev_res_n = 0;         // rows held (0: nothing resident)
  bool ev_res_has_ctx = false;
  bool defer_outputs = false;   // pclean_set_sweep_mode bit 0
  void* commit_state = nullptr;  // owned by commit.hip
  HostStage stage;               // page-locked staging of caller arrays (table uploads, latent-sweep inputs / outputs)
  HostStage ustage;              // ... of the row unions of CandTable::delta_log (asked for in the middle of a sweep call,
                                 // while `stage` holds the call's inputs and outputs)
  const int32_t* obs_override = nullptr;  // eval.hip: ensure_leaf_cache scores "item t observes value t"
  int32_t active_begin = 0, active_count = -1;  // pclean_set_active_rows window (-1 = all rows)
  int32_t timed_block = 0;  // the block whose root launch group pclean_timing / pclean_root_stats describe (pclean_set_timed_block)
  bool prior_mode = false;     // sweep.hip / latent.hip: the running sweep proposes from the priors (use_dd_proposals = false)
  bool force_generic = false;  // debug: never take the compact-table root kernel
  bool no_item_agg = false;    // debug: aggregate evidence with the global sort + run-length encoding only
  void* sweep_state = nullptr;  // owned by sweep.hip
  void* rccl_comm = nullptr;    // ncclComm_t of pclean_comm_init (comm.hip)
  DevBuf<int64_t> stats_pack;   // pclean_allreduce_stats_fused: the tables' delta counts as one vector
  int32_t comm_ranks = 0, comm_rank = 0;
  // what the collectives of this context moved (pclean_comm_get_stats): HIP events around the last all-gather / all-reduce on
  // the library's stream, folded into the totals when the next one is queued or the statistics are read
  struct CommStats {
    uint64_t ag_calls = 0, ag_bytes_last = 0, ar_calls = 0, ar_elems_last = 0;
    double ag_us = 0.0, ar_us = 0.0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // all-gather begin / end, all-reduce begin / end
    bool ag_pending = false, ar_pending = false;
  } comm_stats;
};

inline int pclean_fail(pclean_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx->err = buf;
  }
  return code;
}

#define HIPCHK(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return pclean_fail(ctx, PCLEAN_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                  \
  } while (0)

// dist_kernels.hip
int pclean_launch_dist(pclean_ctx* ctx, PairTable& pt, const int32_t* d_obs_ids, const int32_t* d_lat_ids,
                       int dist_mode, const int32_t* h_lat_ids);  // h_lat_ids: the latent string ids on the host
// density tables (api.hip)
int pclean_ensure_density(pclean_ctx* ctx, int max_len);
// sweep.hip
void pclean_sweep_state_free(pclean_ctx* ctx);
// commit.hip
void pclean_commit_state_free(pclean_ctx* ctx);
void pclean_commit_table_reuploaded(pclean_ctx* ctx, int table_id);
// comm.hip: collectives queued on the library's stream (no synchronisation)
int pclean_comm_allreduce_stats_queue(pclean_ctx* ctx, int32_t n_tables, const int32_t* table_ids, int32_t local_is_zero);
int pclean_comm_allgather_i32(pclean_ctx* ctx, const int32_t* send, int32_t* recv, size_t words_per_rank);
