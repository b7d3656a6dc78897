// Device-side views used by the enumeration kernels (internal).
#pragma once
#include "ctx.h"

#define PCLEAN_MAX_CHILDREN 16
#define PCLEAN_MAX_TERMS 16

struct TermDev {
  const int32_t* obs_col;   // [n_rows] observed value index, -1 = missing
  const int32_t* cand_col;  // [n_cand] latent value index of each candidate
  const uint8_t* pair;      // D[obs][lat], elem_bytes wide
  const uint16_t* lat_len;  // [n_lat]
  const int32_t* fn;        // ctx lookup table or nullptr
  int32_t n_lat, elem_bytes, dens_kind, max_typos, ctx_slot, fn_nb;
  int32_t ctx_mode, pad;    // 0: ctx of the item; 1: ctx of the evidence row, fn[ctx][cand]; 2: fn[cand][ctx]
  const int32_t* aux_col;   // MAYBE_SWAP: [n_cand] number of options of the candidate's key group
  int32_t other_val, pad2;  // MAYBE_SWAP: first latent value that is "not one of the options" (the dummy; strings drawn for a chosen dummy follow it)
};

// Gaussian observation with enumerated locals (pclean_gauss resolved to device pointers)
struct GaussDev {
  int32_t on, n_dims, n_locals, t_kind;
  const double* x;    // [n_rows]
  const double* mu;   // mean table
  int32_t src_kind[4];
  const int32_t* src_ptr[4];
  int32_t src_slot[4], stride[4];
  int32_t local_n[2];
  const int32_t* local_obs[2];
  double local_logp[2];
  int32_t t_src, pad;
  double t_scale[4], t_lad[4];
  const double* tx[4];  // non-linear transformation u: backward_u(x) / log|deriv_u(backward_u(x))| of every row (indexed like x);
  const double* tl[4];  // null: x * t_scale[u] / t_lad[u]
  double sigma, log_sigma;
};

struct NodeDev {
  int32_t kind, n_cand, n_terms, pad;
  GaussDev g;
  const int64_t* counts;
  const double* logc_full;
  const double* logc_m1;
  double scal[4];
  TermDev terms[PCLEAN_MAX_TERMS];
};

struct DensDev {
  const double* nb;
  const double* logl;
  int32_t nb_stride, pad;
  const double* prob_same;  // log1p(-p_i)
  const double* prob_diff;  // log(p_i)
  const double* logn;       // log(n)
};

// Aggregated evidence of one term of a latent-class node (latent.hip: ensure_agg): for original latent item i the
// distinct (ctx value, observed value) pairs among its evidence rows, ascending, with multiplicities:
// entries [off[i], off[i+1]) of key / cnt; key = item << 40 | ctx << 24 | (observed value index + 1) (0 = missing).
struct AggDev {
  const uint64_t* key;
  const int32_t* cnt;
  const int32_t* off;  // runs of original item oi: [off[oi], end ? end[oi] : off[oi + 1])
  const int32_t* end;  // null: the runs of consecutive items are contiguous
};

// Work items of one enumeration launch. Item t scores evidence row row[t]
// (identity when null) under ctx[t][.], with candidate excl[t] having lost one
// reference; draws use particle id particle[t] (n_draws==1) or 0..n_draws-1.
struct ItemsDev {
  int32_t n, pad;
  const int32_t* row;
  const int32_t* ctx;
  const int32_t* excl;
  const int32_t* particle;
  int64_t row_offset;  // global id of local row 0 (RNG counter), multi-GPU sharding
  const int32_t* out_pos;  // where item t writes its outputs (identity when null)
  // Evidence sets (latent-class rows scored against all observed rows that refer to them,
  // ExternalLikelihoodNodes of proposal_compiler.jl:306-350): item t sums its terms over
  // observed rows ev_rows[ev_lo[t] .. ev_hi[t]); ev_ctx[e*PCLEAN_MAX_CTX + s] = ctx of evidence row e.
  const int32_t* ev_lo;   // [n] first evidence position of item t (CSR: offsets)
  const int32_t* ev_hi;   // [n] one past the last (CSR: offsets + 1)
  const int32_t* ev_rows;
  const int32_t* ev_ctx;
  const int32_t* rng_row;  // RNG row id of item t (defaults to the evidence row)
  // Grouped launch (items with identical score vectors): workgroup g scores item members[grp_off[g]]
  // once and emits the log marginal and the draws of every member item members[grp_off[g] ..
  // grp_off[g+1]) — each with its own RNG row / particle.  Null: one item per workgroup, n = items.
  const int32_t* grp_off;
  const int32_t* members;
  // draw j of item t is written to draws_out[t * draw_is + j * draw_ds]; (0, 0) = item-major (n_draws, 1).
  // The sweep keeps its particle arrays particle-major ([P][N]: draw_is = 1, draw_ds = N) so that the
  // one-thread-per-row particle kernels read them coalesced.
  int32_t draw_is, draw_ds;
  // evidence sets: per term of the node the aggregated evidence (device array [n_terms]) and the original
  // latent item each list item stands for (identity when null)
  const AggDev* agg;
  const int32_t* ev_item;
  // Indirect launch (the sync-free re-run of the items an evidence-set scan flagged): workgroup b of the generic kernels
  // takes item sel[b] and retires when b >= *sel_n — the list and its length never leave the device.  Null: item b.
  const int32_t* sel;
  const unsigned int* sel_n;
  // grouped launch: group of member position mi, + 1 (the grouping's uid; null: not at hand) — per-member outputs of a
  // launch whose groups were not cut into pieces (lazy draws) go by position, not by a walk over a group's members
  const int32_t* grp_uid;
};

// Log-marginals of the children of a "new row": either one value per item, or a
// per-unique-observed-value cache indexed through an observed column.
struct ChildrenDev {
  int32_t n, pad;
  const double* arr[PCLEAN_MAX_CHILDREN];
  const int32_t* obs_col[PCLEAN_MAX_CHILDREN];
  int32_t n_obs[PCLEAN_MAX_CHILDREN];
};

// Upper bounds of the children's log-marginals for the gate of the "new row" branch (gate_new_kernel)
struct GateDev {
  int32_t n, pad;
  const double* cache[PCLEAN_MAX_CHILDREN];   // cacheable leaf: exact per-unique-observed-value marginal, else null
  const int32_t* obs_col[PCLEAN_MAX_CHILDREN];
  int32_t n_obs[PCLEAN_MAX_CHILDREN];
  double ub[PCLEAN_MAX_CHILDREN];             // otherwise: a constant upper bound
};
int pclean_launch_gate(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const GateDev& gt, int32_t* flag);

// Fast path for a reference slot with many candidates (root_wave.hip, the dominant kernel): per term a
// candidate-compact byte table comp[o][k] = D[o][value of candidate k] (rebuilt only when
// the latent table's columns change) so a work item streams F contiguous byte rows instead
// of gathering, plus the candidates' word lengths clen[k].
struct FastTermDev {
  const uint8_t* comp;     // [n_obs][kpad] byte distances saturated at 42 (root_wave.hip: PRE_CLAMP); null for a ctx term
  const uint8_t* clen;     // [kpad]
  const uint8_t* cmin;     // [n_obs][cstride] smallest byte of comp[o] within each block of 64 candidates (255 padding)
  const int32_t* obs_col;  // [n_rows]
  int32_t max_typos, ctx_slot;  // ctx_slot >= 0: the latent value goes through fn[ctx][value] first (a
                                // JuliaNode of an earlier block's choice); scored by gathering, never pre-filtered
  const uint8_t* pair;       // [n_obs][n_lat] byte distances (ctx terms; true distance behind a saturated byte)
  const uint16_t* lat_len;   // [n_lat]
  const int32_t* cand_col;   // [n_cand]
  const int32_t* fn;         // [n_ctx][fn_nb]                         (ctx terms only)
  int32_t n_lat, fn_nb;
};
struct FastRootDev {
  int32_t n_cand, kpad, n_terms, lmax, dstride;
  int32_t kscan;          // candidates [kscan, kpad) are spare capacity that never held a row (CandTable::n_used): the scans stop
                          // there; a multiple of 64 (or kpad); kpad stays the stride of the byte rows
  int32_t is_leaf;        // 1: option list of a LEAF node (no exclusion, no "new row" candidate; prior_e / counts / logc_m1 null)
  const uint16_t* alive;  // [kpad / 16] bit e of word q: candidate 16 q + e is a live row / an option with a finite prior
  const double* prior_e;  // [kpad] log(count-discount) - logden_m1, -inf for free slots / padding
  const double* prior_n;  // [kpad] same with logden_full (no exclusion)
  const double* logc_m1;
  const int64_t* counts;
  double scal[4];
  // integer pre-filter (root_fast.hip): terms whose byte rows are summed, 1 / (smallest cost of one
  // edit), and the largest prior with / without an excluded reference
  int32_t n_pre, pre[3];
  int32_t cstride, atd_stride;  // cstride: bytes per row of the block-minimum tables (0: none)
  double inv_c, prior_max_e, prior_max_n;
  const double* atd;         // [max_len + 1][atd_stride] AddTypos log-density by (latent length, distance) (ctx->atd)
  const uint8_t* zero_row;   // kpad zero bytes: stands in for the byte row of a missing observation
  const int32_t* obs_rm;     // [rows][PCLEAN_MAX_TERMS] the terms' observed values row-major (null: gather obs_col[f][row])
  FastTermDev terms[PCLEAN_MAX_TERMS];
};

// Optional extra of a compact-table root launch (root_wave.hip): LAZY draws.  The last block of a sweep needs the draw of
// the finally chosen particle alone (the final choice looks at the weights, i.e. the log marginals, only): instead of n_draws
// draws per member item the kernel leaves every group's survivor list and fixed-point prefix (lz_k / lz_p, ROOT_LZ_CAP entries
// per group; lz_ns[g] = their number, -1: the group's draws were all written to draws_out after all — settled, overflowed, or
// a row of eager_rows) and pclean_launch_lazy_draws draws once per row after the final choice, from the same Philox counter.
#define ROOT_LZ_CAP 256
struct RootExtra {
  int32_t* lz_k;
  uint64_t* lz_p;
  int32_t* lz_ns;
  const int32_t* eager_rows;  // [rows] != 0: every draw of this row's items is wanted now (null: of no row)
  // filled by the launch for pclean_launch_lazy_draws: per-group fixed-point totals (device, inside desc_scratch)
  const uint64_t* g_U;
};
int pclean_launch_root_fast(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const ChildrenDev& ch,
                            uint64_t seed, uint32_t sweep, uint32_t site, int n_draws, double* lse_out,
                            int32_t* draws_out, int32_t* overflow_flag, unsigned int* overflow_count,
                            int32_t* desc_scratch, int32_t* overflow_list,
                            unsigned int* scan_stats = nullptr, int n_items = 0, const double* pre_score = nullptr,
                            bool want_worklist = true, unsigned int* wl_stat = nullptr, const int32_t* pre_obs = nullptr,
                            RootExtra* extra = nullptr);
// one draw per row, after the final choice, from the lists a lazy root launch left (see RootExtra): thread per member
// position mi of the launch's grouping; uid[mi] - 1 = its group.  slot_item: [P][N] item of (particle, row), null: item = row;
// the retained particle of a row with a current referent (cur_b[row] >= 0, particle 0) is not this kernel's.
struct LazyDrawArgs {
  int32_t n_pos, n_rows, n_particles, pad;
  const int32_t* members;    // [n_pos] item of position mi (null: mi)
  const int32_t* uid;        // [n_pos] group of position mi, + 1 (null: mi + 1)
  const int32_t* item_row;   // [items] row of an item (null: the item)
  const int32_t* slot_item;
  const int32_t* chosen;     // [rows] the chosen particle
  const int32_t* cur_b;      // [rows]
  const int32_t* eager_rows;
  const int32_t* draws_item; // [items][P] draws of the groups with lz_ns < 0
  const int32_t* lz_k;
  const uint64_t* lz_p;
  const int32_t* lz_ns;
  const uint64_t* g_U;
  int32_t* pchoice;          // [P][N]: pchoice[chosen * N + row] = the draw
  int64_t row_offset;
  int32_t res_new, pad2;
};
int pclean_launch_lazy_draws(pclean_ctx* ctx, const LazyDrawArgs& a, uint64_t seed, uint32_t sweep, uint32_t site);
// gate of the new-row branch per GROUP of `it` (grouped view), with the exact score of every group's current referent and
// the observed values of its row (obs_out: PCLEAN_MAX_TERMS words per group, what group_desc_kernel would gather again)
int pclean_launch_group_gate(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const GateDev& gt, int32_t* flag,
                             double* score_out, int32_t* obs_out);
int pclean_launch_overflow_fast(pclean_ctx* ctx, const FastRootDev& fr, const ItemsDev& it, const ChildrenDev& ch,
                                uint64_t seed, uint32_t sweep, uint32_t site, int n_draws, double* lse_out,
                                int32_t* draws_out, const int32_t* over_list, const unsigned int* over_count);
int pclean_overflow_fast_ok(const FastRootDev& fr, const ItemsDev& it);
int pclean_launch_root_flags(pclean_ctx* ctx, int n_groups, const int32_t* gd, const int32_t* grp_off,
                             const int32_t* members, const int32_t* oflag, int32_t* out);
size_t pclean_fast_desc_words(int n_groups);  // int32 words of desc_scratch for n_groups groups
// fault hunting (PCLEAN_DEBUG_LATENT): range checks of what the Gaussian evidence term of `node` would dereference
int pclean_debug_gauss_ev_probe(pclean_ctx* ctx, int n_items, int P, int n_nodes, const NodeDev* nds, int node, const ItemsDev& it,
                                const int32_t* vals, int n_mean, int n_ev);
// out[row][f] = obs_col[f][row] (f < n_terms, else -1), rows [0, n_rows): the row-major copy FastRootDev::obs_rm points into
int pclean_build_obs_rowmajor(pclean_ctx* ctx, const int32_t* const* obs_cols, int n_terms, int n_rows, int32_t* out);
int pclean_build_compact(pclean_ctx* ctx, const uint8_t* pair, int n_obs, int n_lat, const int32_t* cand_col,
                         const uint16_t* lat_len, int n_cand, int kpad, uint8_t* comp, uint8_t* clen);
// block minima of a compact table: cmin[o][kb] = min of comp[o][64 kb .. 64 kb + 63] (root_wave.hip: the coarse level
// of the pre-filter scan)
int pclean_build_compact_min(pclean_ctx* ctx, const uint8_t* comp, int n_obs, int kpad, int cstride, uint8_t* cmin);
// ... of the blocks holding the candidates rows[0 .. n_rows) (after pclean_update_compact), or of every block when the rows are many
int pclean_update_compact_min(pclean_ctx* ctx, const uint8_t* comp, int n_obs, int kpad, int cstride, const int32_t* rows,
                              int n_rows, uint8_t* cmin);
int pclean_update_compact(pclean_ctx* ctx, const uint8_t* pair, int n_obs, int n_lat, const int32_t* cand_col,
                          const uint16_t* lat_len, const int32_t* rows, int n_rows, int kpad, uint8_t* comp, uint8_t* clen);
int pclean_build_priors(pclean_ctx* ctx, const int64_t* counts, const double* logc_full, int n_cand, int kpad,
                        double logden_e, double logden_n, double* prior_e, double* prior_n, uint16_t* alive);

int pclean_launch_enum(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const ChildrenDev& ch, uint64_t seed,
                       uint32_t sweep, uint32_t site, int n_draws, double* lse_out, double* scores_out,
                       int32_t* draws_out, double* scores_tmp = nullptr);
// doubles of scratch (scores_tmp) with which the launch above computes its scores one candidate per thread first; 0: not used
size_t pclean_enum_split_scores(const NodeDev& nd, const ItemsDev& it);
// cacheable option lists: per-observed-value (maximum, total, coarse prefix) and draws through them (enum_kernels.hip)
int pclean_leaf_coarse_blocks(int n_options);
int pclean_launch_leaf_coarse_build(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, int n_blocks, double* lse_out,
                                    double* m_out, uint64_t* U_out, uint64_t* coarse, int dummy_k, uint64_t* udummy_out);
int pclean_launch_leaf_coarse_draw(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const int32_t* obs_col, int n_obs,
                                   int n_blocks, const double* lse_c, const double* m_c, const uint64_t* U_c,
                                   const uint64_t* coarse, uint64_t seed, uint32_t sweep, uint32_t site, int n_draws,
                                   double* lse_out, int32_t* draws_out);
// prior proposals (use_dd_proposals = false): w[slot] += likelihood of the slot's sampled sub-tree (enum_kernels.hip)
int pclean_launch_prior_terms(pclean_ctx* ctx, size_t n_slots, int N, int n_nodes, const NodeDev* nds,
                              const int32_t* n_children, const int32_t* child_begin, const int32_t* children,
                              const int32_t* pchoice, const int32_t* pnewpos, const int32_t* vals, const int32_t* it_ctx,
                              double* w);
int pclean_launch_prior_terms_ev(pclean_ctx* ctx, int n_items, int P, int n_nodes, const NodeDev* nds, const AggDev* const* aggs,
                                 const int32_t* n_children, const int32_t* child_begin, const int32_t* children, int n_roots,
                                 const int32_t* roots, const ItemsDev& it, const int32_t* vals, double* w);
// option list of a LEAF node scored against evidence sets (enum_kernels.hip: ev_leaf_block_kernel)
// ... or a reference slot (FK node; ch = the marginals of its new-row branch's children, null for an option list)
int pclean_launch_ev_leaf(pclean_ctx* ctx, const NodeDev& nd, const ItemsDev& it, const FastRootDev& fr, uint64_t seed,
                          uint32_t sweep, uint32_t site, int n_draws, double* lse_out, int32_t* draws_out,
                          int32_t* overflow_flag, unsigned int* overflow_count, int32_t* overflow_list = nullptr,
                          const ChildrenDev* ch = nullptr, uint32_t* dsum = nullptr);  // dsum: [it.n][kpad] zeroed scratch: the
                                                                                        // weighted sums by a chip-wide kernel first
