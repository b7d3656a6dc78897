// TEST INFRASTRUCTURE: host build of pclean_amd/csrc/commit_core.h (the algorithm of the device-resident commit) with
// tid = 0 / nt = 1, so that the CPU suite can hold it against the product's host commit (pclean_amd/parallel.py +
// trace.py) on real sweeps without a GPU.  The product never loads this library: its commit runs as a HIP kernel
// (pclean_amd/csrc/commit.hip) built from the same header.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../pclean_amd/csrc/commit_core.h"

struct Harness {
  std::vector<std::vector<PccNodeIn>> nodes;
  std::vector<std::vector<int32_t>> children, colmap;
  std::vector<std::vector<const int32_t*>> opt_vals;
  std::vector<int> is_score;
  PccSchema sc;
  bool built = false;
  std::vector<int32_t> gflag[PCC_MAX_SLOTS], gscan[PCC_MAX_SLOTS], glist[PCC_MAX_SLOTS], chg[PCC_MAX_SLOTS];
  const char* why = nullptr;
};

extern "C" {
void* pcch_create() { return new Harness(); }
void pcch_destroy(void* h) { delete (Harness*)h; }

// nodes: int32 [n_nodes][12] in the layout of pclean_node (include/pclean_hip.h); opt_vals[i]: option values of leaf i
int pcch_add_block(void* hv, int n_nodes, const int32_t* nodes12, int n_children, const int32_t* children, int n_colmap,
                   const int32_t* colmap, const int64_t* opt_vals, int is_score) {
  Harness* h = (Harness*)hv;
  std::vector<PccNodeIn> in;
  for (int i = 0; i < n_nodes; ++i) {
    const int32_t* n = nodes12 + 12 * i;
    // kind, table, term_begin, n_terms, child_begin, n_children, parent, parent_fk_col, cacheable, colmap_begin, dummy_value, dummy_spec
    in.push_back(PccNodeIn{n[0], n[1], n[4], n[5], n[6], n[7], n[9], n[10]});
  }
  h->nodes.push_back(in);
  h->children.emplace_back(children, children + n_children);
  h->colmap.emplace_back(colmap, colmap + n_colmap);
  std::vector<const int32_t*> ov(n_nodes, nullptr);
  for (int i = 0; i < n_nodes; ++i) ov[i] = (const int32_t*)(intptr_t)opt_vals[i];
  h->opt_vals.push_back(ov);
  h->is_score.push_back(is_score);
  return (int)h->nodes.size() - 1;
}

// returns 0, or 1 with *why set when the device commit does not take these blocks
int pcch_build(void* hv, const char** why) {
  Harness* h = (Harness*)hv;
  std::vector<PccBlockIn> in;
  for (size_t b = 0; b < h->nodes.size(); ++b)
    in.push_back(PccBlockIn{h->nodes[b].data(), (int32_t)h->nodes[b].size(), h->children[b].data(), h->is_score[b]});
  h->why = pcc_build_schema(in.data(), (int)in.size(), h->sc);
  if (why) *why = h->why;
  if (h->why) return 1;
  for (int p = 0; p < h->sc.n_plans; ++p) {
    const int bi = h->sc.plan_block[p];
    h->sc.plans[p].colmap = h->colmap[bi].data();
    for (size_t i = 0; i < h->nodes[bi].size(); ++i) h->sc.plans[p].opt_vals[i] = h->opt_vals[bi][i];
  }
  h->built = true;
  return 0;
}
int pcch_n_slots(void* hv) { return ((Harness*)hv)->sc.n_slots; }
int pcch_slot_table(void* hv, int s) { return ((Harness*)hv)->sc.slot_table[s]; }
int pcch_n_plans(void* hv) { return ((Harness*)hv)->sc.n_plans; }
int pcch_plan_block(void* hv, int p) { return ((Harness*)hv)->sc.plan_block[p]; }
int pcch_plan_n_used(void* hv, int p) { return ((Harness*)hv)->sc.plans[p].n_used; }

// arrays of table slot s (owned by the caller, modified in place by pcch_commit)
int pcch_set_table(void* hv, int s, int stride, int n_cols, int32_t* cols, int64_t* counts, uint8_t* live, int32_t* free_stack,
                   int32_t* state8, int32_t* origin) {
  Harness* h = (Harness*)hv;
  if (!h->built || s < 0 || s >= h->sc.n_slots) return -1;
  PccTable& t = h->sc.tables[s];
  t.cols = cols;
  t.counts = counts;
  t.live = live;
  t.free_stack = free_stack;
  t.state = state8;
  t.origin = origin;
  t.stride = stride;
  t.n_cols = n_cols;
  h->gflag[s].assign(stride > 0 ? stride : 1, 0);
  h->gscan[s].assign(stride > 0 ? stride : 1, 0);
  h->glist[s].assign(stride > 0 ? stride : 1, 0);
  t.gflag = h->gflag[s].data();
  t.gscan = h->gscan[s].data();
  t.glist = h->glist[s].data();
  h->chg[s].assign(stride > 0 ? stride : 1, 0);
  t.chg = h->chg[s].data();
  return 0;
}

// One commit.  Per plan p (arrays of n_plans pointers passed as int64): choice, chosen, chosen_newpos, vals, moved_list,
// new_list, counts2, cur, delta.  res_out: fallback, n_changed, then n_records[16], n_distinct[16].
int pcch_commit(void* hv, int N, int sweep_idx, int row_lo, int kcap, const int32_t* nn, const int64_t* choice,
                const int64_t* chosen, const int64_t* chosen_newpos, const int64_t* vals, const int64_t* moved_list,
                const int64_t* new_list, const int64_t* counts2, const int64_t* cur, const int64_t* delta, int32_t* res_out) {
  Harness* h = (Harness*)hv;
  if (!h->built) return -1;
  const int P = h->sc.n_plans;
  std::vector<PccBlock> blocks(P);
  std::vector<std::vector<int32_t>> ht(P), rep(P), flags(P), scan(P), base(P), newid(P), recpos(P);
  int hsz = 1;
  while (hsz < 4 * kcap) hsz <<= 1;
  for (int p = 0; p < P; ++p) {
    PccBlock& b = blocks[p];
    memset(&b, 0, sizeof b);
    b.N = N;
    b.nn = nn[p];
    b.block_id = h->sc.plan_block[p];
    b.sweep_idx = sweep_idx;
    b.row_lo = row_lo;
    b.choice = (const int32_t*)(intptr_t)choice[p];
    b.chosen = (const int32_t*)(intptr_t)chosen[p];
    b.chosen_newpos = (const int32_t*)(intptr_t)chosen_newpos[p];
    b.vals = (const int32_t*)(intptr_t)vals[p];
    b.moved_list = (const int32_t*)(intptr_t)moved_list[p];
    b.new_list = (const int32_t*)(intptr_t)new_list[p];
    b.counts2 = (const int32_t*)(intptr_t)counts2[p];
    b.cur = (int32_t*)(intptr_t)cur[p];
    b.delta = (const int64_t*)(intptr_t)delta[p];
    b.kcap = kcap;
    b.hmask = hsz - 1;
    ht[p].assign(hsz, -1);
    rep[p].assign(kcap, 0);
    flags[p].assign(kcap, 0);
    scan[p].assign(kcap, 0);
    base[p].assign((size_t)kcap * h->sc.plans[p].n_used, 0);
    newid[p].assign(kcap, 0);
    recpos[p].assign(kcap, 0);
    b.ht = ht[p].data();
    b.rep = rep[p].data();
    b.flags = flags[p].data();
    b.scan = scan[p].data();
    b.base = base[p].data();
    b.newid = newid[p].data();
    b.recpos = recpos[p].data();
  }
  for (int s = 0; s < h->sc.n_slots; ++s) {
    h->sc.tables[s].state[PCC_ST_COLS_CHANGED] = 0;
    h->sc.tables[s].state[PCC_ST_CREATED] = 0;
    h->sc.tables[s].state[PCC_ST_DELETED] = 0;
    h->sc.tables[s].state[PCC_ST_NCHG] = 0;
  }
  PccResult res;
  memset(&res, 0, sizeof res);
  int32_t part[2];
  pcc_commit(h->sc.tables, h->sc.n_slots, h->sc.plans, blocks.data(), P, &res, part, 0, 1);
  res_out[0] = res.fallback;
  res_out[1] = res.n_changed;
  for (int i = 0; i < 16; ++i) {
    res_out[2 + i] = res.n_records[i];
    res_out[18 + i] = res.n_distinct[i];
  }
  return 0;
}

// The commit of a sweep whose rows were sharded over n_ranks ranks, the way pclean_commit_device_dist does it: every rank's
// lists packed into its segment (pcc_pack), the segments side by side (what the all-gather leaves in every rank's HBM),
// the concatenated lists (pcc_merge), then the commit in its gathered form.  Per (rank r, plan p) at index r * P + p:
// choice, chosen_newpos, vals, moved_list, new_list, counts2 (rows relative to the rank's shard); per rank: chosen, N,
// row_lo; per plan: cur (ALL observed rows), delta (summed over the ranks), cap_m, cap_k.
int pcch_commit_gathered(void* hv, int n_ranks, int n_rows_total, int sweep_idx, const int32_t* N_r, const int32_t* row_lo_r,
                         const int32_t* empty_r, const int32_t* nn, const int32_t* cap_m, const int32_t* cap_k,
                         const int64_t* choice, const int64_t* chosen, const int64_t* chosen_newpos, const int64_t* vals,
                         const int64_t* moved_list, const int64_t* new_list, const int64_t* counts2, const int64_t* cur,
                         const int64_t* delta, int32_t* res_out) {
  Harness* h = (Harness*)hv;
  if (!h->built) return -1;
  const int P = h->sc.n_plans;
  PccSegLayout L;
  memset(&L, 0, sizeof L);
  L.n_plans = P;
  int off = 0;
  for (int p = 0; p < P; ++p) {
    L.off[p] = off;
    L.cap_m[p] = cap_m[p];
    L.cap_k[p] = cap_k[p];
    L.nn[p] = nn[p];
    off += pcc_seg_words(cap_m[p], cap_k[p], nn[p]);
  }
  L.seg_words = off;
  std::vector<int32_t> all((size_t)off * n_ranks, -7);
  for (int r = 0; r < n_ranks; ++r)
    for (int p = 0; p < P; ++p) {
      PccBlock b;
      memset(&b, 0, sizeof b);
      b.N = N_r[r];
      b.nn = nn[p];
      b.row_lo = row_lo_r[r];
      b.choice = (const int32_t*)(intptr_t)choice[r * P + p];
      b.chosen = (const int32_t*)(intptr_t)chosen[r];
      b.chosen_newpos = (const int32_t*)(intptr_t)chosen_newpos[r * P + p];
      b.vals = (const int32_t*)(intptr_t)vals[r * P + p];
      b.moved_list = (const int32_t*)(intptr_t)moved_list[r * P + p];
      b.new_list = (const int32_t*)(intptr_t)new_list[r * P + p];
      b.counts2 = (const int32_t*)(intptr_t)counts2[r * P + p];
      pcc_pack(L, p, b, empty_r[r], all.data() + (size_t)r * off, 0, 1);
    }
  PccResult res;
  memset(&res, 0, sizeof res);
  std::vector<PccBlock> blocks(P);
  std::vector<std::vector<int32_t>> ht(P), rep(P), flags(P), scan(P), base(P), newid(P), recpos(P), g_moved(P), g_choice(P), g_new(P),
      g_chosen(P), g_vals(P);
  std::vector<int32_t> g_counts2(2 * P, 0);
  for (int p = 0; p < P; ++p) {
    const int om = cap_m[p] * n_ranks, ok = cap_k[p] * n_ranks, kcap = ok > 16 ? ok : 16;
    g_moved[p].assign(om + 1, 0);
    g_choice[p].assign(om + 1, 0);
    g_new[p].assign(ok + 1, 0);
    g_chosen[p].assign(ok + 1, 0);
    g_vals[p].assign((size_t)ok * nn[p] + 1, 0);
    pcc_merge(L, p, n_ranks, all.data(), om, ok, g_moved[p].data(), g_choice[p].data(), g_new[p].data(), g_chosen[p].data(),
              g_vals[p].data(), g_counts2.data() + 2 * p, &res.fallback_in, 0, 1);
    int hsz = 1;
    while (hsz < 4 * kcap) hsz <<= 1;
    PccBlock& b = blocks[p];
    memset(&b, 0, sizeof b);
    b.N = n_rows_total;
    b.nn = nn[p];
    b.block_id = h->sc.plan_block[p];
    b.sweep_idx = sweep_idx;
    b.row_lo = 0;
    b.vals = g_vals[p].data();
    b.moved_list = g_moved[p].data();
    b.new_list = g_new[p].data();
    b.moved_choice = g_choice[p].data();
    b.rec_chosen = g_chosen[p].data();
    b.counts2 = g_counts2.data() + 2 * p;
    b.cur = (int32_t*)(intptr_t)cur[p];
    b.delta = (const int64_t*)(intptr_t)delta[p];
    b.kcap = kcap;
    b.hmask = hsz - 1;
    ht[p].assign(hsz, -1);
    rep[p].assign(kcap, 0);
    flags[p].assign(kcap, 0);
    scan[p].assign(kcap, 0);
    base[p].assign((size_t)kcap * h->sc.plans[p].n_used, 0);
    newid[p].assign(kcap, 0);
    recpos[p].assign(kcap, 0);
    b.ht = ht[p].data();
    b.rep = rep[p].data();
    b.flags = flags[p].data();
    b.scan = scan[p].data();
    b.base = base[p].data();
    b.newid = newid[p].data();
    b.recpos = recpos[p].data();
  }
  for (int s = 0; s < h->sc.n_slots; ++s) {
    h->sc.tables[s].state[PCC_ST_COLS_CHANGED] = 0;
    h->sc.tables[s].state[PCC_ST_CREATED] = 0;
    h->sc.tables[s].state[PCC_ST_DELETED] = 0;
    h->sc.tables[s].state[PCC_ST_NCHG] = 0;
  }
  int32_t part[2];
  pcc_commit(h->sc.tables, h->sc.n_slots, h->sc.plans, blocks.data(), P, &res, part, 0, 1);
  res_out[0] = res.fallback;
  res_out[1] = res.n_changed;
  for (int i = 0; i < 16; ++i) {
    res_out[2 + i] = res.n_records[i];
    res_out[18 + i] = res.n_distinct[i];
  }
  return 0;
}
}
