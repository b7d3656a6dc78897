// Device-resident commit of one batched sweep of the observed class: what the tail of run_smc!
// (src/inference/row_inference.jl:169-185) and dependency_tracking.jl:6-258 do for one row — move the row's
// reference, create the latent rows a chosen particle proposed (refer_to_row!, 205-236), collect the rows nobody
// refers to any more (unrefer_to_row!, 162-201) — done for a whole sweep at once, on the device, with the batched
// semantics of pclean_amd/parallel.py:exchange_and_commit + pclean_amd/trace.py (materialise_bulk, insert_rows_bulk,
// delete_rows_bulk): identical new-row proposals of a sweep become ONE latent row (first occurrence), a proposing
// row whose old referent just lost its last reference and holds exactly the proposed values keeps it, row ids come
// from the table's free list (last freed first) and then from its high-water mark — the same ids, the same free
// list, the same counts as the host commit, bit for bit.
//
// ONE code path for two builds:
//   * the GPU (commit.hip, PCC_DEVICE): one workgroup; every loop is strided over its threads, phases are
//     separated by workgroup barriers, control flow is uniform (every branch condition is read from memory
//     written before the preceding barrier);
//   * the host harness of the CPU test-suite (tests/commit_host): tid = 0, nt = 1, barriers are no-ops.
// Nothing here is a CPU fallback of the product: the product path only ever runs the GPU build.
#pragma once
#include <stdint.h>

#ifdef PCC_DEVICE
#define PCC_FN __device__
#define PCC_BARRIER() __syncthreads()
#define PCC_ADD64(p, v) atomicAdd((unsigned long long*)(p), (unsigned long long)(long long)(v))
#define PCC_ADD32(p, v) atomicAdd((int*)(p), (int)(v))
#define PCC_FETCH_ADD32(p, v) atomicAdd((int*)(p), (int)(v))
#define PCC_OR32(p, v) atomicOr((int*)(p), (int)(v))
#define PCC_MIN32(p, v) atomicMin((int*)(p), (int)(v))
#define PCC_CAS32(p, cmp, v) atomicCAS((int*)(p), (int)(cmp), (int)(v))
#else
#define PCC_FN static inline
#define PCC_BARRIER() ((void)0)
#define PCC_ADD64(p, v) (*(p) += (int64_t)(v))
#define PCC_ADD32(p, v) (*(p) += (int32_t)(v))
static inline int32_t pcc_fetch_add32_host(int32_t* p, int32_t v) {
  const int32_t old = *p;
  *p += v;
  return old;
}
#define PCC_FETCH_ADD32(p, v) pcc_fetch_add32_host((p), (v))
#define PCC_OR32(p, v) (*(p) |= (int32_t)(v))
#define PCC_MIN32(p, v) (*(p) = (*(p) < (int32_t)(v) ? *(p) : (int32_t)(v)))
static inline int32_t pcc_cas32_host(int32_t* p, int32_t cmp, int32_t v) {
  const int32_t old = *p;
  if (old == cmp) *p = v;
  return old;
}
#define PCC_CAS32(p, cmp, v) pcc_cas32_host((p), (cmp), (v))
#endif

#ifndef PCC_STAMP
#define PCC_STAMP(name) ((void)0)
#endif

#define PCC_MAX_SLOTS 16   // latent tables one commit can touch (the classes reachable from the blocks' roots)
#define PCC_MAX_NODES 64   // nodes of one block's plan
#define PCC_MAX_FK 8       // direct reference slots of one class
#define PCC_MAX_COLS 96    // flattened columns of one class
#define PCC_MAX_BLOCKS 16
#define PCC_MAX_DEPTH 12   // reference chain length (garbage-collection cascade)

// words of PccTable::state
#define PCC_ST_NHW 0           // high-water mark of the table (LatentTable.n)
#define PCC_ST_NFREE 1         // entries of the free stack (LatentTable.free)
#define PCC_ST_COLS_CHANGED 2  // a row was written since the flag was last cleared
#define PCC_ST_CREATED 3       // rows created / deleted since the counters were last cleared
#define PCC_ST_DELETED 4
#define PCC_ST_NCHG 5           // entries of PccTable::chg: the rows written since the counter was last cleared
#define PCC_ST_WORDS 8

// reasons a commit is refused (PccResult::fallback): nothing has been modified, the caller commits on the host
#define PCC_FB_RECORDS 1   // more new-row records than the scratch holds
#define PCC_FB_DUMMY 2     // a created row would hold a ProposalDummyValue (the host draws its value, block_proposal.jl:58-60)
#define PCC_FB_CAPACITY 4  // a table would outgrow its device capacity

struct PccTable {
  int32_t* cols;        // [n_cols][stride] column-major flattened rows
  int64_t* counts;      // [stride] reference counts
  uint8_t* live;        // [stride]
  int32_t* free_stack;  // [stride] ids of collected rows, reused last-in first-out
  int32_t* state;       // [PCC_ST_WORDS]
  int32_t* gflag;       // [stride] scratch, all zero between uses
  int32_t* gscan;       // [stride] scratch
  int32_t* glist;       // [stride] scratch: the rows being deleted
  int32_t* chg;         // [stride] ids of the rows this commit wrote (any order): what derived tables have to refresh
  int32_t* origin;      // [stride][4]: (mark, creating observed row, chosen particle, sweep index); mark: 0 untouched since
                        //   the last pull, 1 + block = recorded, -1 = cleared (Trace.row_origin)
  int32_t stride, n_cols, n_fk, n_blocks_using;  // n_blocks_using: blocks of the commit whose plan can create rows here
  int32_t fk_col[PCC_MAX_FK];   // direct reference slots: column ...
  int32_t fk_slot[PCC_MAX_FK];  // ... and table slot of the target class
};

struct PccPlan {  // static description of one block's enumeration plan (include/pclean_hip.h: pclean_node)
  int32_t n_nodes, n_fk_nodes, n_used, track;  // track: the block has option lists with a ProposalDummyValue
  int32_t exclusive, pad1;               // no other block of the commit can touch the tables of this plan
  int32_t kind[PCC_MAX_NODES];
  int32_t slot[PCC_MAX_NODES];           // FK nodes: table slot; leaves: -1
  int32_t parent[PCC_MAX_NODES];
  int32_t parent_fk_col[PCC_MAX_NODES];
  int32_t cmb[PCC_MAX_NODES];            // colmap_begin
  int32_t dummy_val[PCC_MAX_NODES];      // leaves: 1 + latent value id of the ProposalDummyValue, 0 none
  int32_t node_used[PCC_MAX_NODES];      // FK nodes: index into used_slot
  int32_t fk_post[PCC_MAX_NODES];        // FK nodes in creation order: a node after all its descendants, the root last
  int32_t used_slot[PCC_MAX_SLOTS];      // distinct table slots of the FK nodes; used_slot[0] = the root's
  const int32_t* opt_vals[PCC_MAX_NODES];  // leaves: option index -> latent value id
  const int32_t* colmap;                 // pairs (child node, child column), see pclean_node::colmap_begin
};

struct PccBlock {  // one block of one sweep: the sweep's outputs (device-resident) + scratch
  int32_t N, nn, block_id, sweep_idx;
  int32_t row_lo, pad0;          // global index of the window's first row (Trace.row_origin holds global rows)
  const int32_t* choice;         // [N] chosen referent, PCLEAN_CHOICE_NEW = -1
  const int32_t* chosen;         // [N] chosen particle
  const int32_t* chosen_newpos;  // [N] record of the chosen particle's new row in vals
  const int32_t* vals;           // [.][nn] node choices of proposed new rows
  const int32_t* moved_list;     // rows whose referent changed, ascending
  const int32_t* new_list;       // rows whose chosen particle proposed a NEW referent, ascending
  const int32_t* counts2;        // [0] number of moved rows, [1] number of new-row records
  int32_t* cur;                  // [N] current referents (updated)
  const int64_t* delta;          // [root stride] delta reference counts of the sweep, summed over all ranks
  int32_t kcap, hmask;           // scratch capacity in records; hash table size - 1
  int32_t* ht;                   // [hmask + 1]
  int32_t* rep;                  // [kcap] first record with the same node choices
  int32_t* flags;                // [kcap]
  int32_t* scan;                 // [kcap]
  int32_t* base;                 // [kcap][n_used] first allocation index of the record in every used table
  int32_t* newid;                // [kcap] id of the record's root row
  int32_t* recpos;               // [kcap] chosen_newpos[new_list[j]] (filled by phase A)
  // GATHERED form (several ranks: every rank's lists concatenated in rank order = row order, pcc_pack / pcc_merge; the
  // per-row arrays choice / chosen / chosen_newpos are null, N = all rows, row_lo = 0, moved_list / new_list hold GLOBAL
  // rows, record j of new_list is vals + j * nn and cur covers every observed row):
  const int32_t* moved_choice;   // [n_moved] new referent of moved_list[m]
  const int32_t* rec_chosen;     // [k] chosen particle of the proposing row of record j
};

struct PccResult {
  int32_t fallback;              // PCC_FB_* bits; non-zero: nothing was modified
  int32_t n_changed;             // rows whose referent changed (all blocks)
  int32_t n_records[PCC_MAX_BLOCKS];
  int32_t n_distinct[PCC_MAX_BLOCKS];
  int32_t n_nested[PCC_MAX_BLOCKS];   // distinct proposals with a nested NEW referent
  int32_t alloc_upper[PCC_MAX_SLOTS];  // rows the commit may create per table (before reuse)
  int32_t fallback_in, pad;      // set by the caller before the commit (pcc_merge: PCC_FB_RECORDS), 0 otherwise
};

#define PCC_F_FIRST 1
#define PCC_F_SIMPLE 2
#define PCC_F_REUSE 4

// part[0..nt) -> exclusive prefix sums, part[nt] = total
#ifdef PCC_DEVICE
__device__ void pcc_scan_partials(int32_t* part, int nt, int tid) {  // nt: a multiple of 64, at most 1024
  __shared__ int32_t wave_total[16];
  const int lane = tid & 63, w = tid >> 6;
  const int32_t x = part[tid];
  int32_t incl = x;
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t y = __shfl_up(incl, o, 64);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wave_total[w] = incl;
  __syncthreads();
  if (w == 0) {
    const int nw = nt >> 6;
    const int32_t v = lane < nw ? wave_total[lane] : 0;
    int32_t inc = v;
    for (int o = 1; o < 16; o <<= 1) {
      const int32_t y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
    }
    if (lane < nw) wave_total[lane] = inc - v;
    if (lane == nw - 1) part[nt] = inc;
  }
  __syncthreads();
  part[tid] = wave_total[w] + incl - x;
}
#else
static inline void pcc_scan_partials(int32_t* part, int nt, int tid) {
  int32_t run = 0;
  for (int t = 0; t < nt; ++t) {
    const int32_t x = part[t];
    part[t] = run;
    run += x;
  }
  part[nt] = run;
}
#endif

// ---- workgroup-wide exclusive scan of a[0..n) in place; returns the total (uniform) ------------------------------
// part: nt + 1 words shared by the workgroup.  Each thread owns a contiguous chunk (n is a few thousand).
PCC_FN int32_t pcc_excl_scan(int32_t* a, int n, int32_t* part, int tid, int nt) {
  const int chunk = (n + nt - 1) / nt;
  const int lo = tid * chunk < n ? tid * chunk : n;
  const int hi = lo + chunk < n ? lo + chunk : n;
  int32_t s = 0;
  for (int i = lo; i < hi; ++i) s += a[i];
  PCC_BARRIER();  // (part may still be read by the previous scan's callers)
  part[tid] = s;
  PCC_BARRIER();
  pcc_scan_partials(part, nt, tid);
  PCC_BARRIER();
  int32_t run = part[tid];
  for (int i = lo; i < hi; ++i) {
    const int32_t x = a[i];
    a[i] = run;
    run += x;
  }
  const int32_t total = part[nt];
  PCC_BARRIER();
  return total;
}

PCC_FN const int32_t* pcc_record(const PccBlock& b, int j) { return b.vals + (size_t)b.recpos[j] * b.nn; }
// node choice k of a record; entry 0 (the root) is NEW by definition (the library stores other things there)
PCC_FN int32_t pcc_val(const int32_t* v, int k) { return k == 0 ? -1 : v[k]; }

PCC_FN uint32_t pcc_hash(const PccBlock& b, const int32_t* v) {
  uint64_t h = 0x2545f4914f6cdd1dull;
  for (int k = 0; k < b.nn; ++k) {
    h ^= (uint64_t)(uint32_t)pcc_val(v, k) + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 32;
  }
  return (uint32_t)(h >> 16);
}
PCC_FN bool pcc_same(const PccBlock& b, const int32_t* x, const int32_t* y) {
  for (int k = 1; k < b.nn; ++k)
    if (x[k] != y[k]) return false;
  return true;
}

// FK nodes of a record that are reached through NEW parents and are NEW themselves (bit f of the mask);
// row[f] = the existing referent of every reached FK node that is not NEW
PCC_FN uint64_t pcc_new_nodes(const PccPlan& pl, const int32_t* v, int32_t* row) {
  uint64_t mask = 0;
  for (int p = pl.n_fk_nodes - 1; p >= 0; --p) {  // parents before children
    const int f = pl.fk_post[p];
    if (f == 0) {
      mask |= 1ull;
      continue;
    }
    if (!((mask >> pl.parent[f]) & 1ull)) continue;
    if (pcc_val(v, f) < 0)
      mask |= 1ull << f;
    else
      row[f] = v[f];
  }
  return mask;
}

// value of column c of the row a record creates at FK node f (children of f already have their rows)
PCC_FN int32_t pcc_col_value(const PccTable* tb, const PccPlan& pl, const int32_t* v, const int32_t* row, int f, int c) {
  const int cn = pl.colmap[2 * (pl.cmb[f] + c)], cc = pl.colmap[2 * (pl.cmb[f] + c) + 1];
  if (cn >= 0) {
    if (pl.kind[cn] == 1) return pl.opt_vals[cn][v[cn]];
    const PccTable& tc = tb[pl.slot[cn]];
    return tc.cols[(size_t)cc * tc.stride + row[cn]];
  }
  for (int p = 0; p < pl.n_fk_nodes; ++p) {  // the reference slot itself: its child node's row
    const int g = pl.fk_post[p];
    if (g != 0 && pl.parent[g] == f && pl.parent_fk_col[g] == c) return row[g];
  }
  return -1;
}

// ---- garbage collection: delete the flagged rows of table slot s (ascending), return their number (uniform) ------
PCC_FN int32_t pcc_delete_flagged(PccTable* tb, int s, int32_t* part, int tid, int nt) {
  PccTable& t = tb[s];
  const int n = t.state[PCC_ST_NHW];
  const int top0 = t.state[PCC_ST_NFREE];
  for (int r = tid; r < n; r += nt) t.gscan[r] = t.gflag[r];
  PCC_BARRIER();
  const int32_t L = pcc_excl_scan(t.gscan, n, part, tid, nt);
  for (int r = tid; r < n; r += nt)
    if (t.gflag[r]) {
      const int pos = t.gscan[r];
      t.glist[pos] = r;
      t.live[r] = 0;
      t.free_stack[top0 + pos] = r;
      t.gflag[r] = 0;
    }
  PCC_BARRIER();
  if (tid == 0 && L) {
    t.state[PCC_ST_NFREE] = top0 + L;
    t.state[PCC_ST_DELETED] += L;
  }
  PCC_BARRIER();
  return L;
}

// delete_rows_bulk (trace.py) of every unreferenced live row of table slot s0, cascading depth-first through the
// reference slots in column order exactly as the host recursion does
PCC_FN void pcc_collect(PccTable* tb, int s0, int32_t* part, int tid, int nt) {
  {
    PccTable& t = tb[s0];
    const int n = t.state[PCC_ST_NHW];
    for (int r = tid; r < n; r += nt) t.gflag[r] = (t.counts[r] == 0 && t.live[r]) ? 1 : 0;
    PCC_BARRIER();
  }
  int32_t fr_slot[PCC_MAX_DEPTH], fr_n[PCC_MAX_DEPTH], fr_next[PCC_MAX_DEPTH];
  int depth = 0;
  const int32_t L0 = pcc_delete_flagged(tb, s0, part, tid, nt);
  if (L0 > 0) {
    fr_slot[0] = s0;
    fr_n[0] = L0;
    fr_next[0] = 0;
    depth = 1;
  }
  while (depth > 0) {
    const int s = fr_slot[depth - 1];
    PccTable& t = tb[s];
    if (fr_next[depth - 1] >= t.n_fk) {
      --depth;
      continue;
    }
    const int q = fr_next[depth - 1]++;
    const int L = fr_n[depth - 1];
    PccTable& g = tb[t.fk_slot[q]];
    const int32_t* refcol = t.cols + (size_t)t.fk_col[q] * t.stride;
    for (int i = tid; i < L; i += nt) PCC_ADD64(&g.counts[refcol[t.glist[i]]], -1);
    PCC_BARRIER();
    for (int i = tid; i < L; i += nt) {
      const int ref = refcol[t.glist[i]];
      if (g.counts[ref] == 0 && g.live[ref]) g.gflag[ref] = 1;
    }
    PCC_BARRIER();
    const int32_t L2 = pcc_delete_flagged(tb, t.fk_slot[q], part, tid, nt);
    if (L2 > 0 && depth < PCC_MAX_DEPTH) {
      fr_slot[depth] = t.fk_slot[q];
      fr_n[depth] = L2;
      fr_next[depth] = 0;
      ++depth;
    }
  }
}

// Proposals without a nested NEW referent: the proposing row keeps its old referent where that row just lost its last
// reference and holds exactly the proposed values (Trace.materialise_bulk: reuse).  pending_delta: the sweep's delta
// counts are not applied yet (phase A of a plan whose tables no other block touches: what it sees is what its own
// apply step will see).  Sets PCC_F_REUSE + newid; scan[j] = 1 for the simple first records that need a fresh row.
PCC_FN void pcc_mark_reuse(const PccTable* tb, const PccPlan& pl, const PccBlock& b, int k, bool pending_delta, int tid, int nt) {
  const PccTable& root = tb[pl.used_slot[0]];
  for (int j = tid; j < k; j += nt) {
    int32_t fl = b.flags[j];
    int32_t fresh = 0;
    if ((fl & PCC_F_FIRST) && (fl & PCC_F_SIMPLE)) {
      const int32_t* v = pcc_record(b, j);
      int32_t row[PCC_MAX_NODES];
      (void)pcc_new_nodes(pl, v, row);
      const int old = b.cur[b.new_list[j]];
      bool same = old >= 0 && root.live[old] && root.counts[old] + (pending_delta ? b.delta[old] : 0) == 0;
      for (int c = 0; c < root.n_cols && same; ++c)
        same = root.cols[(size_t)c * root.stride + old] == pcc_col_value(tb, pl, v, row, 0, c);
      if (same) {
        b.newid[j] = old;
        b.flags[j] = fl | PCC_F_REUSE;
      } else {
        fresh = 1;
      }
    }
    b.scan[j] = fresh;
  }
  PCC_BARRIER();
}

// ---- phase A of one block: group identical records, check what the commit would need; modifies scratch only -------
PCC_FN void pcc_prepare_block(const PccTable* tb, const PccPlan& pl, const PccBlock& b, int bi, PccResult* res, int tid, int nt) {
  const int k = b.counts2[1];
  if (tid == 0) {
    res->n_records[bi] = k;
    res->n_distinct[bi] = 0;
    res->n_nested[bi] = 0;
  }
  if (k > b.kcap) {
    if (tid == 0) PCC_OR32(&res->fallback, PCC_FB_RECORDS);
    return;
  }
  if (k == 0) return;
  for (int j = tid; j < k; j += nt) b.recpos[j] = b.chosen_newpos ? b.chosen_newpos[b.new_list[j]] : j;
  int hm = b.hmask < 63 ? b.hmask : 63;  // hash table of this sweep: the smallest power of two >= 4 k (at most the scratch's)
  while (hm + 1 < 4 * k && hm < b.hmask) hm = 2 * hm + 1;
  const uint32_t hmask = (uint32_t)hm;
  for (int i = tid; i <= hm; i += nt) b.ht[i] = -1;
  PCC_BARRIER();
  // insert: the slot of a class of identical records ends up holding its smallest record index
  for (int j = tid; j < k; j += nt) {
    const int32_t* v = pcc_record(b, j);
    uint32_t s = pcc_hash(b, v) & hmask;
    for (;;) {
      int32_t c = b.ht[s];
      if (c < 0) {
        c = PCC_CAS32(&b.ht[s], -1, j);
        if (c < 0) break;  // claimed
      }
      if (pcc_same(b, pcc_record(b, c), v)) {
        PCC_MIN32(&b.ht[s], j);
        break;
      }
      s = (s + 1) & hmask;
    }
  }
  PCC_BARRIER();
  for (int j = tid; j < k; j += nt) {
    const int32_t* v = pcc_record(b, j);
    uint32_t s = pcc_hash(b, v) & hmask;
    for (;;) {
      const int32_t c = b.ht[s];
      if (pcc_same(b, pcc_record(b, c), v)) {
        b.rep[j] = c;
        break;
      }
      s = (s + 1) & hmask;
    }
    int32_t fl = 0;
    if (b.rep[j] == j) {
      fl = PCC_F_FIRST | PCC_F_SIMPLE;
      int32_t row[PCC_MAX_NODES];
      const uint64_t mask = pcc_new_nodes(pl, v, row);
      if (mask != 1ull) {  // a nested reference slot proposes a NEW row as well
        fl &= ~PCC_F_SIMPLE;
        PCC_ADD32(&res->n_nested[bi], 1);
      }
      for (int p = 0; p < pl.n_fk_nodes; ++p) {
        const int f = pl.fk_post[p];
        if (((mask >> f) & 1ull) && !(f == 0 && pl.exclusive)) PCC_ADD32(&res->alloc_upper[pl.slot[f]], 1);
      }
      for (int l = 0; l < pl.n_nodes; ++l)  // would a created row hold a ProposalDummyValue?
        if (pl.kind[l] == 1 && pl.dummy_val[l] != 0 && ((mask >> pl.parent[l]) & 1ull) && v[l] >= 0 &&
            pl.opt_vals[l][v[l]] == pl.dummy_val[l] - 1)
          PCC_OR32(&res->fallback, PCC_FB_DUMMY);
      PCC_ADD32(&res->n_distinct[bi], 1);
    }
    b.flags[j] = fl;
  }
  PCC_BARRIER();
  if (pl.exclusive) {  // the root rows this block really creates: first records that do not keep their old referent
    pcc_mark_reuse(tb, pl, b, k, true, tid, nt);
    for (int j = tid; j < k; j += nt)
      if ((b.flags[j] & PCC_F_FIRST) && !(b.flags[j] & PCC_F_REUSE)) PCC_ADD32(&res->alloc_upper[pl.used_slot[0]], 1);
    PCC_BARRIER();
  }
}

// current referents of the rows that moved: the chosen existing referent, or the row created for the chosen proposal
PCC_FN void pcc_update_cur(const PccBlock& b, int tid, int nt) {
  const int k = b.counts2[1], n_moved = b.counts2[0];
  for (int m = tid; m < n_moved; m += nt) {
    const int r = b.moved_list[m];
    const int c = b.moved_choice ? b.moved_choice[m] : b.choice[r];
    if (c >= 0) b.cur[r] = c;
  }
  for (int j = tid; j < k; j += nt) b.cur[b.new_list[j]] = b.newid[j];
}

// ---- phase B of one block: apply ------------------------------------------------------------------------------------
PCC_FN void pcc_apply_block(PccTable* tb, const PccPlan& pl, const PccBlock& b, int bi, PccResult* res, int32_t* part, int tid,
                            int nt) {
  const int k = b.counts2[1], n_moved = b.counts2[0];
  const int U = pl.n_used;
  PccTable& root = tb[pl.used_slot[0]];
  // 1. delta reference counts of the rows that moved between existing referents
  {
    const int n0 = root.state[PCC_ST_NHW];
    for (int r = tid; r < n0; r += nt) root.counts[r] += b.delta[r];
    PCC_BARRIER();
  }
  PCC_STAMP("delta");
  if (k > 0) {
  // 2. proposals without a nested NEW referent that keep the proposing row's old referent (pcc_mark_reuse; a plan with
  //    tables of its own decided in phase A, which also left scan[j] = needs a fresh root row)
  if (!pl.exclusive) pcc_mark_reuse(tb, pl, b, k, false, tid, nt);
  const int32_t n_simple = pcc_excl_scan(b.scan, k, part, tid, nt);
  for (int j = tid; j < k; j += nt) b.base[(size_t)j * U] = b.scan[j];
  PCC_BARRIER();
  PCC_STAMP("reuse+scan");
  // 3. proposals with nested NEW referents are created one after the other, after all the others: allocation index
  //    of every record in every table = exclusive scan of the rows it creates there
  int32_t total[PCC_MAX_SLOTS];
  for (int u = 0; u < U; ++u) total[u] = u == 0 ? n_simple : 0;
  if (res->n_nested[bi] > 0)
  for (int u = 0; u < U; ++u) {
    for (int j = tid; j < k; j += nt) {
      int32_t c = 0;
      const int32_t fl = b.flags[j];
      if ((fl & PCC_F_FIRST) && !(fl & PCC_F_SIMPLE)) {
        int32_t row[PCC_MAX_NODES];
        const uint64_t mask = pcc_new_nodes(pl, pcc_record(b, j), row);
        for (int p = 0; p < pl.n_fk_nodes; ++p) {
          const int f = pl.fk_post[p];
          if (((mask >> f) & 1ull) && pl.node_used[f] == u) ++c;
        }
      }
      b.scan[j] = c;
    }
    PCC_BARRIER();
    const int32_t tot = pcc_excl_scan(b.scan, k, part, tid, nt);
    for (int j = tid; j < k; j += nt) {
      const int32_t fl = b.flags[j];
      if ((fl & PCC_F_FIRST) && !(fl & PCC_F_SIMPLE)) b.base[(size_t)j * U + u] = b.scan[j] + (u == 0 ? n_simple : 0);
    }
    total[u] = tot + (u == 0 ? n_simple : 0);
    PCC_BARRIER();
  }
  PCC_STAMP("nested scans");
  // 4. create the rows: ids from the free stack (last freed first), then from the high-water mark
  int32_t top0[PCC_MAX_SLOTS], hw0[PCC_MAX_SLOTS];
  for (int u = 0; u < U; ++u) {
    top0[u] = tb[pl.used_slot[u]].state[PCC_ST_NFREE];
    hw0[u] = tb[pl.used_slot[u]].state[PCC_ST_NHW];
  }
  PCC_BARRIER();
  for (int j = tid; j < k; j += nt) {
    const int32_t fl = b.flags[j];
    if (!(fl & PCC_F_FIRST) || (fl & PCC_F_REUSE)) continue;
    const int32_t* v = pcc_record(b, j);
    int32_t row[PCC_MAX_NODES];
    const uint64_t mask = pcc_new_nodes(pl, v, row);
    int32_t used_n[PCC_MAX_SLOTS];
    for (int u = 0; u < U; ++u) used_n[u] = 0;
    const int obs_row = b.new_list[j];
    for (int p = 0; p < pl.n_fk_nodes; ++p) {
      const int f = pl.fk_post[p];
      if (!((mask >> f) & 1ull)) continue;
      const int u = pl.node_used[f];
      PccTable& t = tb[pl.used_slot[u]];
      const int a = b.base[(size_t)j * U + u] + used_n[u]++;
      const int id = a < top0[u] ? t.free_stack[top0[u] - 1 - a] : hw0[u] + (a - top0[u]);
      for (int c = 0; c < t.n_cols; ++c) t.cols[(size_t)c * t.stride + id] = pcc_col_value(tb, pl, v, row, f, c);
      t.counts[id] = 0;
      t.live[id] = 1;
      for (int q = 0; q < t.n_fk; ++q)  // refer_to_row! of the new row's own reference slots
        PCC_ADD64(&tb[t.fk_slot[q]].counts[t.cols[(size_t)t.fk_col[q] * t.stride + id]], 1);
      if (pl.track) {
        t.origin[4 * (size_t)id] = 1 + b.block_id;
        t.origin[4 * (size_t)id + 1] = b.row_lo + obs_row;
        t.origin[4 * (size_t)id + 2] = b.rec_chosen ? b.rec_chosen[j] : b.chosen[obs_row];
        t.origin[4 * (size_t)id + 3] = b.sweep_idx;
      } else if (!(fl & PCC_F_SIMPLE)) {
        t.origin[4 * (size_t)id] = -1;
      }
      t.state[PCC_ST_COLS_CHANGED] = 1;
      t.chg[PCC_FETCH_ADD32(&t.state[PCC_ST_NCHG], 1)] = id;
      row[f] = id;
    }
    b.newid[j] = row[0];
  }
  PCC_BARRIER();
  if (tid == 0)
    for (int u = 0; u < U; ++u) {
      PccTable& t = tb[pl.used_slot[u]];
      const int a = total[u];
      t.state[PCC_ST_NFREE] = a < top0[u] ? top0[u] - a : 0;
      t.state[PCC_ST_NHW] = hw0[u] + (a > top0[u] ? a - top0[u] : 0);
      t.state[PCC_ST_CREATED] += a;
    }
  PCC_BARRIER();
  PCC_STAMP("create");
  // 5. every proposing row refers to its (group's) new row
  for (int j = tid; j < k; j += nt) {
    const int id = b.newid[b.rep[j]];
    if (b.rep[j] != j) b.newid[j] = id;
    PCC_ADD64(&root.counts[id], 1);
  }
  PCC_BARRIER();
  }  // k > 0
  PCC_STAMP("refer");
  // 6. current referents (the GPU build runs pcc_update_cur as a wide kernel of its own after this one: nothing below
  //    reads them)
#ifndef PCC_CUR_SEPARATE
  pcc_update_cur(b, tid, nt);
#endif
  if (tid == 0) PCC_ADD32(&res->n_changed, n_moved);  // (atomic: the plans may run on a workgroup each)
  PCC_BARRIER();
  PCC_STAMP("cur");
  // 7. garbage collection
  pcc_collect(tb, pl.used_slot[0], part, tid, nt);
  PCC_STAMP("collect");
}

// ---- several ranks: every rank's moved rows and new-row records, concatenated in rank order ----------------------------
// Rows are block-partitioned over the ranks (contiguous shards, ascending), so rank order IS row order: the concatenation of
// the ranks' ascending lists is the ascending global list the one-rank commit works on (SURVEY §8e "second exchange").
// A rank's SEGMENT of the all-gather buffer holds, per plan p (PccSegLayout): a 4-word header (moved rows, new-row records
// of the rank, 0, 0), moved rows [cap_m] (GLOBAL row ids), their new referents [cap_m], proposing rows [cap_k] (global),
// their chosen particles [cap_k], the records [cap_k][nn].  A rank with more entries than a capacity still reports its true
// counts: pcc_merge then refuses the commit (PCC_FB_RECORDS) on every rank alike, before anything is modified.
struct PccSegLayout {
  int32_t n_plans, seg_words;            // words of one rank's segment
  int32_t off[PCC_MAX_BLOCKS];           // first word of plan p inside a segment
  int32_t cap_m[PCC_MAX_BLOCKS], cap_k[PCC_MAX_BLOCKS], nn[PCC_MAX_BLOCKS];
};
static inline int32_t pcc_seg_words(int cap_m, int cap_k, int nn) { return 4 + 2 * cap_m + 2 * cap_k + cap_k * nn; }  // (host side)

// this rank's lists of plan p (per-row form of the block: choice / chosen / chosen_newpos given) -> its segment.
// empty != 0: the rank swept no row of the window (its sweep buffers hold an older sweep): header zero.
PCC_FN void pcc_pack(const PccSegLayout& L, int p, const PccBlock& b, int empty, int32_t* seg, int tid, int nt) {
  int32_t* o = seg + L.off[p];
  const int cm = L.cap_m[p], ck = L.cap_k[p], nn = L.nn[p];
  const int n_moved = empty ? 0 : b.counts2[0], k = empty ? 0 : b.counts2[1];
  if (tid == 0) {
    o[0] = n_moved;
    o[1] = k;
    o[2] = o[3] = 0;
  }
  int32_t* mrow = o + 4;
  int32_t* mch = mrow + cm;
  int32_t* nrow = mch + cm;
  int32_t* nchs = nrow + ck;
  int32_t* rec = nchs + ck;
  const int nm = n_moved < cm ? n_moved : cm, nk = k < ck ? k : ck;
  for (int m = tid; m < nm; m += nt) {
    const int r = b.moved_list[m];
    mrow[m] = b.row_lo + r;
    mch[m] = b.choice[r];
  }
  for (int j = tid; j < nk; j += nt) {
    const int r = b.new_list[j];
    nrow[j] = b.row_lo + r;
    nchs[j] = b.chosen[r];
  }
  for (int x = tid; x < nk * nn; x += nt) {
    const int j = x / nn, c = x - j * nn;
    rec[x] = b.vals[(size_t)b.chosen_newpos[b.new_list[j]] * nn + c];
  }
}

// the gathered lists of plan p from the n_ranks segments of `all` (rank r's segment at all + r * L.seg_words):
// g_moved / g_choice [sum of moved], g_new / g_chosen [sum of records], g_vals [.][nn], counts2[2] = the totals;
// *fallback |= PCC_FB_RECORDS when a rank's lists did not fit its segment (or the totals do not fit cap_out_*)
PCC_FN void pcc_merge(const PccSegLayout& L, int p, int n_ranks, const int32_t* all, int cap_out_m, int cap_out_k,
                      int32_t* g_moved, int32_t* g_choice, int32_t* g_new, int32_t* g_chosen, int32_t* g_vals, int32_t* counts2,
                      int32_t* fallback, int tid, int nt, int32_t* rank_max = nullptr) {
  const int cm = L.cap_m[p], ck = L.cap_k[p], nn = L.nn[p];
  int tot_m = 0, tot_k = 0;
  bool bad = false;
  for (int r = 0; r < n_ranks; ++r) {
    const int32_t* o = all + (size_t)r * L.seg_words + L.off[p];
    bad |= o[0] > cm || o[1] > ck || o[0] < 0 || o[1] < 0;
    tot_m += o[0];
    tot_k += o[1];
  }
  bad |= tot_m > cap_out_m || tot_k > cap_out_k;
  if (tid == 0) {
    counts2[0] = bad ? 0 : tot_m;
    counts2[1] = bad ? 0 : tot_k;
    if (bad) PCC_OR32(fallback, PCC_FB_RECORDS);
    if (rank_max) {  // the largest single rank's lists: what the next exchange's segments are sized by
      int mm = 0, mk = 0;
      for (int r = 0; r < n_ranks; ++r) {
        const int32_t* o = all + (size_t)r * L.seg_words + L.off[p];
        mm = o[0] > mm ? o[0] : mm;
        mk = o[1] > mk ? o[1] : mk;
      }
      rank_max[0] = mm;
      rank_max[1] = mk;
    }
  }
  if (bad) return;
  int base_m = 0, base_k = 0;
  for (int r = 0; r < n_ranks; ++r) {
    const int32_t* o = all + (size_t)r * L.seg_words + L.off[p];
    const int32_t* mrow = o + 4;
    const int32_t* mch = mrow + cm;
    const int32_t* nrow = mch + cm;
    const int32_t* nchs = nrow + ck;
    const int32_t* rec = nchs + ck;
    const int nm = o[0], nk = o[1];
    for (int m = tid; m < nm; m += nt) {
      g_moved[base_m + m] = mrow[m];
      g_choice[base_m + m] = mch[m];
    }
    for (int j = tid; j < nk; j += nt) {
      g_new[base_k + j] = nrow[j];
      g_chosen[base_k + j] = nchs[j];
    }
    for (int x = tid; x < nk * nn; x += nt) g_vals[(size_t)base_k * nn + x] = rec[x];
    base_m += nm;
    base_k += nk;
  }
}

// ---- the whole commit ------------------------------------------------------------------------------------------------
PCC_FN void pcc_commit(PccTable* tb, int n_slots, const PccPlan* plans, const PccBlock* blocks, int n_blocks, PccResult* res,
                       int32_t* part, int tid, int nt) {
  if (tid == 0) {
    res->fallback = res->fallback_in;  // (what pcc_merge found wrong with the gathered lists; 0 on one rank)
    res->n_changed = 0;
    for (int s = 0; s < PCC_MAX_SLOTS; ++s) res->alloc_upper[s] = 0;
    for (int bi = 0; bi < PCC_MAX_BLOCKS; ++bi) res->n_records[bi] = res->n_distinct[bi] = res->n_nested[bi] = 0;
  }
  PCC_BARRIER();
  PCC_STAMP("start");
  if (res->fallback) return;
  for (int bi = 0; bi < n_blocks; ++bi) {
    pcc_prepare_block(tb, plans[bi], blocks[bi], bi, res, tid, nt);
    PCC_STAMP("prepare");
  }
  PCC_BARRIER();
  if (tid == 0)
    for (int s = 0; s < n_slots; ++s) {
      const int a = res->alloc_upper[s];
      if (a == 0) continue;
      // a table only one block can create rows in sees its free stack exactly; otherwise count on the high-water mark alone
      const int from_free = tb[s].n_blocks_using <= 1 ? tb[s].state[PCC_ST_NFREE] : 0;
      if (tb[s].state[PCC_ST_NHW] + (a > from_free ? a - from_free : 0) > tb[s].stride) res->fallback |= PCC_FB_CAPACITY;
    }
  PCC_BARRIER();
  if (res->fallback) return;
  for (int bi = 0; bi < n_blocks; ++bi) pcc_apply_block(tb, plans[bi], blocks[bi], bi, res, part, tid, nt);
}

// ---- host-side construction of a PccPlan from a block's plan arrays (shared by commit.hip and the test harness) ---
#ifndef PCC_DEVICE_ONLY
struct PccNodeIn {  // the fields of pclean_node the commit needs
  int32_t kind, table, child_begin, n_children, parent, parent_fk_col, colmap_begin, dummy_value;
};
static inline void pcc_post_order(const PccNodeIn* nodes, const int32_t* children, int f, PccPlan& pl) {
  for (int c = 0; c < nodes[f].n_children; ++c) {
    const int g = children[nodes[f].child_begin + c];
    if (nodes[g].kind == 0) pcc_post_order(nodes, children, g, pl);
  }
  pl.fk_post[pl.n_fk_nodes++] = f;
}
// slot_of_table[table id] -> table slot (-1: not a latent table of the commit).  Returns 0, or a negative code:
// -1 too many nodes, -2 a class refers to a table outside the slots, -3 too many tables.
static inline int pcc_build_plan(const PccNodeIn* nodes, int n_nodes, const int32_t* children, const int32_t* slot_of_table,
                                 PccPlan& pl) {
  if (n_nodes > PCC_MAX_NODES || n_nodes < 1 || nodes[0].kind != 0) return -1;
  pl.n_nodes = n_nodes;
  pl.n_fk_nodes = 0;
  pl.n_used = 0;
  pl.track = 0;
  for (int i = 0; i < n_nodes; ++i) {
    pl.kind[i] = nodes[i].kind;
    pl.parent[i] = nodes[i].parent;
    pl.parent_fk_col[i] = nodes[i].parent_fk_col;
    pl.cmb[i] = nodes[i].colmap_begin;
    pl.dummy_val[i] = nodes[i].kind == 1 ? nodes[i].dummy_value : 0;
    pl.slot[i] = -1;
    pl.node_used[i] = -1;
    pl.opt_vals[i] = nullptr;
    if (nodes[i].kind == 1 && nodes[i].dummy_value != 0) pl.track = 1;
    if (nodes[i].kind == 0) {
      const int s = slot_of_table[nodes[i].table];
      if (s < 0) return -2;
      pl.slot[i] = s;
    }
  }
  pcc_post_order(nodes, children, 0, pl);
  // used slots: the root's first
  pl.used_slot[pl.n_used++] = pl.slot[0];
  for (int p = 0; p < pl.n_fk_nodes; ++p) {
    const int f = pl.fk_post[p];
    int u = -1;
    for (int x = 0; x < pl.n_used; ++x)
      if (pl.used_slot[x] == pl.slot[f]) u = x;
    if (u < 0) {
      if (pl.n_used >= PCC_MAX_SLOTS) return -3;
      u = pl.n_used;
      pl.used_slot[pl.n_used++] = pl.slot[f];
    }
    pl.node_used[f] = u;
  }
  return 0;
}

// ---- table slots, reference-slot schema and plans of a commit from the loaded blocks --------------------------------
struct PccBlockIn {
  const PccNodeIn* nodes;
  int32_t n_nodes;
  const int32_t* children;
  int32_t is_score;  // a block without a reference slot: nothing to commit
};
struct PccSchema {
  int32_t n_slots, n_plans;
  int32_t slot_of_table[64];        // candidate-table id -> slot, -1
  int32_t slot_table[PCC_MAX_SLOTS];  // slot -> candidate-table id
  int32_t plan_block[PCC_MAX_BLOCKS];  // plan -> block id
  PccTable tables[PCC_MAX_SLOTS];   // n_fk / fk_col / fk_slot / n_blocks_using filled; pointers and shapes are the caller's
  PccPlan plans[PCC_MAX_BLOCKS];    // opt_vals / colmap are the caller's
};
static inline bool pcc_reaches(const PccSchema& sc, int from, int target, int depth) {
  if (depth > PCC_MAX_SLOTS) return true;
  for (int q = 0; q < sc.tables[from].n_fk; ++q) {
    const int g = sc.tables[from].fk_slot[q];
    if (g == target || pcc_reaches(sc, g, target, depth + 1)) return true;
  }
  return false;
}
// tables on the longest reference chain that starts at slot `from` (the schema is acyclic here)
static inline int pcc_chain_depth(const PccSchema& sc, int from) {
  int d = 0;
  for (int q = 0; q < sc.tables[from].n_fk; ++q) {
    const int x = pcc_chain_depth(sc, sc.tables[from].fk_slot[q]);
    d = x > d ? x : d;
  }
  return 1 + d;
}
// Returns nullptr, or why these blocks cannot be committed on the device (the caller keeps the host commit).
static inline const char* pcc_build_schema(const PccBlockIn* blocks, int n_blocks, PccSchema& sc) {
  sc.n_slots = sc.n_plans = 0;
  for (int t = 0; t < 64; ++t) sc.slot_of_table[t] = -1;
  for (int s = 0; s < PCC_MAX_SLOTS; ++s) {
    sc.tables[s] = PccTable();
    sc.slot_table[s] = -1;
  }
  if (n_blocks > PCC_MAX_BLOCKS) return "too many blocks";
  for (int bi = 0; bi < n_blocks; ++bi) {  // table slots: every latent table an FK node enumerates
    const PccBlockIn& b = blocks[bi];
    if (b.is_score) continue;
    if (b.n_nodes > PCC_MAX_NODES) return "a block with more than 64 plan nodes";
    for (int i = 0; i < b.n_nodes; ++i) {
      if (b.nodes[i].kind != 0) continue;
      const int t = b.nodes[i].table;
      if (t < 0 || t >= 64) return "table id out of range";
      if (sc.slot_of_table[t] < 0) {
        if (sc.n_slots >= PCC_MAX_SLOTS) return "more than 16 latent tables";
        sc.slot_of_table[t] = sc.n_slots;
        sc.slot_table[sc.n_slots++] = t;
      }
    }
  }
  for (int bi = 0; bi < n_blocks; ++bi) {  // direct reference slots of every class: an FK child of an FK node
    const PccBlockIn& b = blocks[bi];
    if (b.is_score) continue;
    bool seen[PCC_MAX_SLOTS] = {false};
    for (int i = 0; i < b.n_nodes; ++i) {
      const PccNodeIn& n = b.nodes[i];
      if (n.kind != 0) continue;
      const int s = sc.slot_of_table[n.table];
      if (!seen[s]) {
        seen[s] = true;
        ++sc.tables[s].n_blocks_using;
      }
      if (n.parent < 0) continue;
      const PccNodeIn& par = b.nodes[n.parent];
      if (par.kind != 0 || n.parent_fk_col < 0) return "a reference slot below an option list";
      PccTable& pt = sc.tables[sc.slot_of_table[par.table]];
      int q = 0;
      for (; q < pt.n_fk; ++q)
        if (pt.fk_col[q] == n.parent_fk_col) break;
      if (q == pt.n_fk) {
        if (pt.n_fk >= PCC_MAX_FK) return "a class with more than 8 reference slots";
        pt.fk_col[q] = n.parent_fk_col;
        pt.fk_slot[q] = s;
        ++pt.n_fk;
      } else if (pt.fk_slot[q] != s) {
        return "inconsistent reference slots";
      }
    }
  }
  for (int s = 0; s < sc.n_slots; ++s) {  // reference slots in column order (the order delete_rows_bulk walks them)
    PccTable& t = sc.tables[s];
    for (int x = 0; x < t.n_fk; ++x)
      for (int y = x + 1; y < t.n_fk; ++y)
        if (t.fk_col[y] < t.fk_col[x]) {
          const int c = t.fk_col[x], g = t.fk_slot[x];
          t.fk_col[x] = t.fk_col[y];
          t.fk_slot[x] = t.fk_slot[y];
          t.fk_col[y] = c;
          t.fk_slot[y] = g;
        }
  }
  for (int s = 0; s < sc.n_slots; ++s)
    if (pcc_reaches(sc, s, s, 0)) return "a class that refers to itself";
  for (int s = 0; s < sc.n_slots; ++s)  // the garbage-collection cascade keeps one frame per table of a reference chain
    if (pcc_chain_depth(sc, s) > PCC_MAX_DEPTH) return "a reference chain deeper than 12 classes";
  int root_seen[PCC_MAX_SLOTS] = {0};
  for (int bi = 0; bi < n_blocks; ++bi) {
    const PccBlockIn& b = blocks[bi];
    if (b.is_score) continue;
    const int p = sc.n_plans++;
    sc.plan_block[p] = bi;
    PccPlan& pl = sc.plans[p];
    pl = PccPlan();
    if (pcc_build_plan(b.nodes, b.n_nodes, b.children, sc.slot_of_table, pl)) return "plan shape";
    if (root_seen[pl.used_slot[0]]++) return "two blocks with the same root class";
    pl.exclusive = 1;
    for (int u = 0; u < pl.n_used; ++u)
      if (sc.tables[pl.used_slot[u]].n_blocks_using > 1) pl.exclusive = 0;
  }
  if (sc.n_plans == 0) return "no block with a reference slot";
  return nullptr;
}
#endif
