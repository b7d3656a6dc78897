// Per-context state of the sweep entry points (sweep.hip, eval.hip, latent.hip) that other translation units of the library read
// (commit.hip: the device-resident commit consumes a sweep's device-side outputs).  Not part of the ABI.
#pragma once
#include <functional>
#include <map>
#include <string>
#include <deque>
#include <vector>

#include "ctx.h"
#include "enum.h"

// ---------------------------------------------------------------------------
// device-side plan description for resolving values of freshly sampled rows
struct PlanDev {
  int32_t n_nodes;
  const int32_t* kind;          // [n_nodes]
  const int32_t* const* cols;   // [n_nodes] base pointer of the node's table columns
  const int32_t* n_rows;        // [n_nodes] column stride
  const int32_t* colmap_begin;  // [n_nodes]
  const int32_t* colmap;        // pairs
};


struct BlockRun {  // per-block device state of one sweep
  DevBuf<int32_t> pchoice, pnewpos, draws, it_ctx, choice, chosen_newpos, vals, locals, moved_flag, new_flag, moved_list,
      new_list, new_slots;
  DevBuf<double> lse;
  int n_new = 0;  // particles of the block that proposed a NEW referent
  int locals_rows = 0;  // rows of `locals` the last sweep filled (0: none)
  DevBuf<int32_t> plocals;  // prior proposals with a Gaussian term: every particle's own choices [P][N][2]
  bool plocals_on = false;
  size_t it_ctx_np = 0;  // shape it_ctx was last zeroed for (sweep.hip: ensure_it_ctx)
  int it_ctx_used = -1;
  bool lazy_new = false;  // their contents are sampled after the final choice, for the chosen particles only
  int ctx_extra_cap = 0;  // room for the extra context items of the next sweep (sweep.hip: ctx_items_kernel)
  DevBuf<int32_t> plan_kind, plan_nrows, plan_cmb, plan_colmap;
  DevBuf<const int32_t*> plan_cols;
  PlanDev plan{};
  bool plan_ready = false;
  std::vector<int32_t> plan_sig_nrows, plan_sig_kind, plan_sig_cmb;  // what the device arrays were built from
  std::vector<const int32_t*> plan_sig_cols;
  size_t plan_sig_colmap = 0;
  uint64_t plan_sig_block = 0;
};

struct FastRoot {  // candidate-compact tables of a reference slot (root_wave.hip)
  std::vector<DevBuf<uint8_t>> comp, clen, cblk;  // cblk: block minima of comp (one byte per 64 candidates)
  std::vector<uint64_t> ver;
  DevBuf<double> prior_e, prior_n;
  DevBuf<uint16_t> alive;
  DevBuf<uint8_t> zero_row;  // kpad zero bytes (byte row of a missing observation)
  int disabled = 0;  // > 0: the pre-filter does not pay for this node (most items overflowed): that many evaluations use the generic kernel
  int backoff = 64;  // length of the next disabled period (doubles every time the retry overflows again)
  uint64_t cmin_key = 0;  // (lmax, dmax, density-table stride) the cached c_min belongs to
  double cmin = 0.0;
  uint64_t prior_ver = 0;
  int kpad = 0;
  double logc_max = 0.0;  // max over candidates of log(count - discount)
  int wl_off = 0;  // > 0: the settle kernel left most groups of this node unsettled: that many launches run without the work list
  // the node's observed values ROW-MAJOR ([n_rows][PCLEAN_MAX_TERMS] int32: the 64 bytes a group's descriptor wants of its
  // row in one piece instead of one line per term column); the observations never change during a run: built once
  DevBuf<int32_t> obs_rm;
  uint64_t obs_rm_key = 0;
};

struct SweepState {
  FastRoot fast[PCLEAN_MAX_BLOCKS * 64];  // [block * 64 + node]
  std::vector<DevBuf<unsigned char>> pool;  // scratch buffers, recycled per sweep
  size_t pool_used = 0;
  BlockRun run[PCLEAN_MAX_BLOCKS];
  DevBuf<int32_t> cur, chosen, ancestors, csmc_flag, did;
  // the last pclean_sweep call, for the calls that finish it (pclean_sweep_finish_*, pclean_commit_device)
  const int32_t* last_cur_base = nullptr;  // current referents of block bi: last_cur_base + bi * last_cur_ld
  size_t last_cur_ld = 0;
  int last_N = 0, last_blocks = 0;
  bool last_dev_cur = false, last_hot_timed = false;
  volatile unsigned int* h_poll = nullptr;  // page-locked (value, sequence number) of read_count (sweep.hip)
  unsigned int* d_poll = nullptr;
  unsigned int poll_seq = 0;
  bool outputs_pending = false;  // counts / statistics of the last sweep not yet read back
  bool lists_on_host = false;    // moved-row / new-row lists of the last sweep already copied
  DevBuf<double> w, log_total, logml_inc, logml_acc, logml;
  DevBuf<unsigned int> counter;
  DevBuf<int32_t*> arr_ptrs;
  std::map<int, DevBuf<int32_t>> leaf_iota;  // key = block*256+node
  std::map<int, uint64_t> leaf_version;
  int64_t row_offset = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evs = nullptr, eve = nullptr;
  hipEvent_t evg0 = nullptr, evg1 = nullptr;  // around group_gate_kernel of the timed root (part of its launch group)
  hipStream_t pre_stream = nullptr;           // compact-table refresh of the later blocks' roots, beside block 0 (pclean_sweep)
  // ... queued by the host the first time it WAITS for the device in the running sweep (read_count): the enqueue costs no
  // critical-path time, and the refresh runs beside the first block's grouping instead of beside its root scan
  std::function<int()> on_first_wait;
  hipEvent_t pre_fork = nullptr, pre_join = nullptr;
  bool gate_timed = false;
  // memo tables of option-list marginals (leaf_memo_*): key = block * 64 + node
  struct LeafMemo {
    DevBuf<uint64_t> keys;   // [cap][3]
    DevBuf<double> vals;     // [cap]
    DevBuf<unsigned int> count;
    uint64_t ver = 0;
    int cap = 0;
  };
  std::map<int, LeafMemo> memo;
  struct TupleIds {  // ensure_tuple_ids: key = block * 64 + node
    DevBuf<int32_t> id;
    DevBuf<uint32_t> pre;
    uint64_t sig = 0;
  };
  std::map<int, TupleIds> tuple_ids;
  // gate of the new-row branch (eval.hip): a node whose gate let EVERY item through three evaluations in a row (the
  // Measure slot of the 1M-row table: the new-row candidate is never 28.5 nats behind) skips it for a while — the gate
  // only ever removes work whose weight is exactly 0, so results do not depend on it.  key = block * 64 + node
  struct GateStat { int all_need_run = 0, skip = 0; };
  std::map<int, GateStat> gate_stat;
  // rows for which some cacheable option list of the block can draw its ProposalDummyValue (sweep.hip: dummy_rows_flags)
  struct DummyRows { DevBuf<int32_t> flag; uint64_t sig = 0; int n = 0, n_flagged = 0; };
  std::map<int, DummyRows> dummy_rows;  // key = block
  // evidence of the running pclean_sweep_latent call (ensure_agg)
  const int32_t* lat_off = nullptr;      // [lat_items + 1] CSR offsets of the original items into the evidence list
  const int32_t* lat_item_of_pos = nullptr;  // [lat_ev]
  int lat_items = 0, lat_ev = 0, lat_max_ev = 0;  // (largest evidence set of the call)
  std::map<int, const AggDev*> lat_agg;  // node -> device array [n_terms]
  DevBuf<int32_t> tail_counts;      // [2 * PCLEAN_MAX_BLOCKS] number of moved rows / rows with a new referent
  int32_t* h_counts = nullptr;      // page-locked mirror of tail_counts (+ scratch words)
  // per-phase HIP-event profile of a sweep (pclean_set_profiling): (phase, start, stop) records
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;
  std::vector<int> prof_phase;      // phase id of record r (events 2r, 2r+1)
  size_t prof_used = 0;
  std::vector<std::string> prof_names;
  std::vector<float> prof_ms;
  std::vector<int32_t> prof_launches;
  // overflow counters of the compact-table launches of the running call whose re-run needs no read-back
  // (overflow_lds_kernel in list mode): counted into the statistics / heuristics at the end of the call
  bool scan_stats_used = false;
  int bank_used = 0;               // counters of the bank (the tail of over_ctr) handed out by the running call
  DevBuf<unsigned int> over_ctr;   // [OVER_SLOTS + STAT_WORDS]: the tail = scan statistics of the timed root launch
  std::deque<DevBuf<unsigned int>> more_banks;
  // queued small read-backs (d2h_small / d2h_flush, sweep.hip) and the device mappings of the host buffers seen so far
  struct PubRegions { const uint32_t* src[12]; uint32_t* dst[12]; uint32_t words[12]; int n; } pub = {};
  std::map<void*, void*> pub_map;  // further counter banks of a call that used up the first (fresh_counter)
  struct OverRec { int block, node, n_items; bool time_it, leaf; int min_items; };  // min_items: from how many items on the "does the pre-filter pay" rule applies
  std::vector<OverRec> over_rec;
  // pclean_sweep_latent: the option lists of a latent row are independent given its evidence — each is evaluated on one
  // of these streams (forked from / joined into the library's stream by events), so that the few-hundred-workgroup
  // launches of a sub-batch overlap on the chip instead of running one behind the other
  static const int MAX_SIDE = 8;
  hipStream_t side[MAX_SIDE] = {};
  hipEvent_t side_fork = nullptr, side_join[MAX_SIDE] = {}, side_mid[MAX_SIDE] = {};
  int n_side = -1;  // -1: not created yet
  unsigned int* h_over = nullptr;  // page-locked copy of over_ctr
  // dummy_correction_kernel: distance matrices of the strings drawn for chosen ProposalDummyValues
  DevBuf<int16_t> dummy_dp;
  DevBuf<unsigned int> dummy_ctr;  // [0] matrices handed out by the running launch, [1] set when they ran out
  bool dummy_used = false;
  // lazy draws of a sweep's last block (enum.h: RootExtra).  pclean_sweep sets lazy_req before the block's root evaluation;
  // eval_node takes it (nested evaluations never see it) and, when the compact-table kernels honoured it, leaves lazy_out for
  // pclean_launch_lazy_draws (the sweep fills in what only it knows: chosen particles, slot_item, pchoice, ...)
  struct LazyReq { bool on = false; const int32_t* eager_rows = nullptr; } lazy_req;
  struct LazyOut { bool valid = false; LazyDrawArgs args{}; uint32_t site = 0; } lazy_out;
  // block 0's root scan of the last pclean_sweep (pclean_debug_root_flags; the scratch stays valid until the next call)
  const int32_t* dbg_desc = nullptr;
  const int32_t* dbg_grp_off = nullptr;
  const int32_t* dbg_members = nullptr;
  const int32_t* dbg_oflag = nullptr;
  int dbg_groups = 0, dbg_items = 0;
};

#define OVER_SLOTS 256
#define STAT_WORDS (64 * 32)  // scan statistics of the timed root launch: 64 slots, 128 bytes apart
#define CTR_BANK 2048         // zeroed 32-bit counters handed out one after the other within a call (fresh_counter)
static SweepState* st(pclean_ctx* ctx) {
  if (!ctx->sweep_state) ctx->sweep_state = new SweepState();
  return (SweepState*)ctx->sweep_state;
}


// sweep.hip: the three steps that end a sweep (see there)
int pclean_sweep_finish_queue(pclean_ctx* ctx);
// small read-backs riding on one synchronisation (sweep.hip; see sweep_internal.h)
int d2h_small(pclean_ctx* ctx, void* host, const void* dev, size_t bytes, void* host_base = nullptr);
int d2h_flush(pclean_ctx* ctx);
// zero bytes of device memory with a plain kernel on the library's stream (hipMemsetAsync costs ~2x the host time of a launch:
// pointer look-ups before its blit kernel is queued; a sweep zeroes nine small buffers)
int dev_zero(pclean_ctx* ctx, void* p, size_t bytes);
int pclean_sweep_finish_synced(pclean_ctx* ctx);
int pclean_sweep_fetch_lists(pclean_ctx* ctx);
