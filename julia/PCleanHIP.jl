# PCleanHIP.jl — the Julia side of the drop-in: `ccall` bindings of include/pclean_hip.h, the lowering of a
# PCleanModel + Query into the static plan IR, the trace <-> flat-table conversion and the commits.
#
# STATUS: written against the reference's sources (/root/reference/src, cited per function) and against the tested
# Python host (pclean_amd/: _lib.py = bindings, model.py = lowering, trace.py / parallel.py / inference.py = commits).
# The build image has no Julia, so this file has NEVER been executed.  What pins it: tests/golden/plans_*.json hold the
# plan arrays the Python lowering produces for the three experiment programs (generator scripts/make_plan_goldens.py), and
# section 3 (`lower`) is stated a second time in Python — tests/julia_lowering.py, function by function over the reference's
# own model structures — where tests/test_julia_lowering.py runs it against those goldens for hospital, flights AND rents.
# `lower(model, query, data, columns)` below must produce the same arrays (dump with `plan_json(lw)` and diff).  Every
# function the pgibbs_sweep! patch at the bottom calls is defined in this file.
#
# Include after `using PClean` from the package's own module scope:  include("PCleanHIP.jl")
module PCleanHIP

using ..PClean: PCleanModel, PCleanClass, PCleanTrace, TableTrace, Query, ObservedDataset, InferenceConfig,
                ForeignKeyNode, RandomChoiceNode, JuliaNode, SubmodelNode, ParameterNode, ExternalLikelihoodNode,
                PCleanNode, VertexID, ClassID, Key, Plan, Step, ProposalDummyValue, AddTypos, StringPrior, TimePrior,
                ChooseUniformly, ChooseProportionally, MaybeSwap, TransformedGaussian, Transformation, Unmodeled, strip_subnodes, has_discrete_proposal, discrete_proposal,
                discrete_proposal_dummy_value, incorporate_row!, unincorporate_row!, pclean_gensym!,
                resample_value!, resample_py_params!

const lib = "libpclean_hip"            # pclean_amd/libpclean_hip.so on the loader path

# =============================================================================================== 1. C ABI
mutable struct Ctx; h::Ptr{Cvoid}; end
function Ctx(device::Integer=0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:pclean_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r)
    rc == 0 || error("pclean_ctx_create failed ($rc): no gfx950 device visible; there is no CPU fallback")
    c = Ctx(r[]); finalizer(x -> ccall((:pclean_ctx_destroy, lib), Cint, (Ptr{Cvoid},), x.h), c); c
end
check(c::Ctx, rc) = rc == 0 || error(unsafe_string(ccall((:pclean_last_error, lib), Cstring, (Ptr{Cvoid},), c.h)))

# isbits mirrors of pclean_node / pclean_term / pclean_infer_config (include/pclean_hip.h)
struct CNode; kind::Int32; table::Int32; term_begin::Int32; n_terms::Int32; child_begin::Int32; n_children::Int32
              parent::Int32; parent_fk_col::Int32; cacheable::Int32; colmap_begin::Int32; dummy_value::Int32; dummy_spec::Int32; end
struct CTerm; obs_col::Int32; cand_col::Int32; pair_table::Int32; dens_kind::Int32; max_typos::Int32
              ctx_slot::Int32; fn_table::Int32; ctx_mode::Int32; end
struct CConfig; num_iters::Int32; num_particles::Int32; dd::Int32; lo::Int32; mh::Int32; rejuv::Int32; report::Int32; end
CConfig(cfg::InferenceConfig) = CConfig(cfg.num_iters, cfg.num_particles, cfg.use_dd_proposals, cfg.use_lo_sweeps,
                                        cfg.use_mh_instead_of_pg, cfg.rejuv_frequency, cfg.reporting_frequency)
const NODE_FK, NODE_LEAF = Int32(0), Int32(1)
const DENS_ADD_TYPOS, DENS_EQUAL = Int32(0), Int32(1)
const CHOICE_NEW = Int32(-1)
const MAX_CTX = 4

load_strings(c, sym::Vector{UInt16}, off::Vector{Int64}) = GC.@preserve sym off check(c,
    ccall((:pclean_load_strings, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{UInt16}, Ptr{Int64}), c.h, length(off) - 1, sym, off))
load_columns(c, obs::Matrix{Int32}) = GC.@preserve obs check(c,           # n_rows x n_cols (column-major = [col][row] in C)
    ccall((:pclean_load_columns, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}), c.h, size(obs, 1), size(obs, 2), obs))
build_pair_table(c, id, obs_ids::Vector{Int32}, lat_ids::Vector{Int32}, mode=1) = GC.@preserve obs_ids lat_ids check(c,
    ccall((:pclean_build_pair_table, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Int32),
          c.h, id, length(obs_ids), obs_ids, length(lat_ids), lat_ids, mode))
set_table(c, id, cols::Matrix{Int32}, counts::Vector{Int64}, py) = GC.@preserve cols counts check(c,  # n_rows x n_cols
    ccall((:pclean_set_table, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}, Ptr{Int64}, Cdouble, Cdouble),
          c.h, id, size(cols, 1), size(cols, 2), cols, counts, py.strength, py.discount))
set_options(c, id, values::Vector{Int32}, logp::Vector{Float64}) = GC.@preserve values logp check(c,
    ccall((:pclean_set_options, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Float64}), c.h, id, length(values), values, logp))
set_fn_table(c, id, fn::Matrix{Int32}) = GC.@preserve fn check(c,         # n_b x n_a in Julia = fn[a][b] in C
    ccall((:pclean_set_fn_table, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}), c.h, id, size(fn, 2), size(fn, 1), fn))
set_lm_tables(c, init_p::Vector{Float64}, trans_p::Matrix{Float64}, letter_sym::Vector{UInt16}) = GC.@preserve init_p trans_p letter_sym check(c,
    ccall((:pclean_set_lm_tables, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{UInt16}), c.h, init_p, trans_p, letter_sym))
set_block_group(c, block, group) = check(c, ccall((:pclean_set_block_group, lib), Cint, (Ptr{Cvoid}, Int32, Int32), c.h, block, group))
set_active_rows(c, first0, count) = check(c, ccall((:pclean_set_active_rows, lib), Cint, (Ptr{Cvoid}, Int32, Int32), c.h, first0, count))
function load_block(c, id, nodes::Vector{CNode}, terms::Vector{CTerm}, children::Vector{Int32}, colmap::Vector{Int32},
                    ctx_block::Vector{Int32}=Int32[], ctx_col::Vector{Int32}=Int32[])
    GC.@preserve nodes terms children colmap ctx_block ctx_col check(c, ccall((:pclean_load_block, lib), Cint,
        (Ptr{Cvoid}, Int32, Int32, Ptr{CNode}, Int32, Ptr{CTerm}, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Int32}),
        c.h, id, length(nodes), nodes, length(terms), terms, length(children), children, length(colmap), colmap,
        length(ctx_block), ctx_block, ctx_col))
end
function string_prior_scores(c, lm::Vector{UInt8}, off::Vector{Int64}, lo, hi, init_logp::Vector{Float64}, trans_logp::Matrix{Float64})
    out = Vector{Float64}(undef, length(off) - 1)
    GC.@preserve lm off init_logp trans_logp out check(c, ccall((:pclean_string_prior_scores, lib), Cint,
        (Ptr{Cvoid}, Int32, Ptr{UInt8}, Ptr{Int64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, length(out), lm, off, lo, hi, init_logp, trans_logp, out))
    out
end
function sweep!(c, cfg::InferenceConfig, seed, sweep_idx, cur::Matrix{Int32})   # cur: n_rows x n_blocks, 0-based slots, -1 none
    cc = Ref(CConfig(cfg))
    GC.@preserve cur check(c, ccall((:pclean_sweep, lib), Cint,               # per-row outputs NULL: read moved / new rows
        (Ptr{Cvoid}, Ref{CConfig}, UInt64, UInt32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}),
        c.h, cc, seed, sweep_idx, size(cur, 2), cur, C_NULL, C_NULL, C_NULL))
end
function moved(c, block)
    n = Ref{Int32}(0)
    check(c, ccall((:pclean_get_moved, lib), Cint, (Ptr{Cvoid}, Int32, Ref{Int32}, Ptr{Int32}, Ptr{Int32}), c.h, block, n, C_NULL, C_NULL))
    rows = Vector{Int32}(undef, n[]); ch = Vector{Int32}(undef, n[])
    GC.@preserve rows ch check(c, ccall((:pclean_get_moved, lib), Cint, (Ptr{Cvoid}, Int32, Ref{Int32}, Ptr{Int32}, Ptr{Int32}), c.h, block, n, rows, ch))
    rows, ch
end
function get_new_rows(c, block, n_nodes)            # vals[node, j]; vals[1, j] = -1 - chosen particle
    n = Ref{Int32}(0)
    check(c, ccall((:pclean_get_new_rows, lib), Cint, (Ptr{Cvoid}, Int32, Ref{Int32}, Ptr{Int32}, Ptr{Int32}), c.h, block, n, C_NULL, C_NULL))
    rows = Vector{Int32}(undef, n[]); vals = Matrix{Int32}(undef, n_nodes, n[])
    GC.@preserve rows vals check(c, ccall((:pclean_get_new_rows, lib), Cint, (Ptr{Cvoid}, Int32, Ref{Int32}, Ptr{Int32}, Ptr{Int32}), c.h, block, n, rows, vals))
    rows, vals
end
function sweep_latent!(c, cfg, seed, sweep_idx, block, roots::Vector{Int32}, keys::Vector{Int32}, ev_off::Vector{Int32},
                       ev_rows::Vector{Int32}, ev_ctx::Union{Nothing,Matrix{Int32}}, excl::Matrix{Int32}, n_nodes)
    n = length(keys); chosen = zeros(Int32, n); vals = fill(Int32(-2), n_nodes, n); cc = Ref(CConfig(cfg))
    ctxp = ev_ctx === nothing ? Ptr{Int32}(C_NULL) : pointer(ev_ctx)          # ev_ctx: MAX_CTX x n_evidence
    GC.@preserve roots keys ev_off ev_rows ev_ctx excl chosen vals check(c, ccall((:pclean_sweep_latent, lib), Cint,
        (Ptr{Cvoid}, Ref{CConfig}, UInt64, UInt32, Int32, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32},
         Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
        c.h, cc, seed, sweep_idx, block, length(roots), roots, n, keys, ev_off, ev_rows, ctxp, excl, chosen, vals))
    chosen, vals
end
# per-candidate scores of ONE plan node of a latent plan against evidence sets (parity checks: what the generated proposal
# of proposal_compiler.jl:306-350 accumulates for every candidate); scores[:, i] over the node's candidates (+ new row)
function score_node_ev(c, block, node, keys::Vector{Int32}, ev_off::Vector{Int32}, ev_rows::Vector{Int32},
                       ev_ctx::Union{Nothing,Matrix{Int32}}, excl::Union{Nothing,Vector{Int32}}, n_cand)
    n = length(keys); lse = zeros(Float64, n); scores = zeros(Float64, n_cand, n)
    ctxp = ev_ctx === nothing ? Ptr{Int32}(C_NULL) : pointer(ev_ctx)
    exp_ = excl === nothing ? Ptr{Int32}(C_NULL) : pointer(excl)
    GC.@preserve keys ev_off ev_rows ev_ctx excl lse scores check(c, ccall((:pclean_score_node_ev, lib), Cint,
        (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, UInt64, UInt32, Int32,
         Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
        c.h, block, node, n, keys, ev_off, ev_rows, ctxp, exp_, 0, 0, 0, lse, scores, C_NULL))
    lse, scores
end
function random_string_prior_at(c, seeds::Vector{UInt64}, elems::Vector{UInt32}, lo, hi, init_p, trans_p)
    n = length(seeds); out = zeros(UInt8, hi, n); len = Vector{Int32}(undef, n)
    GC.@preserve seeds elems init_p trans_p out len check(c, ccall((:pclean_random_string_prior_at, lib), Cint,
        (Ptr{Cvoid}, Int32, Ptr{UInt64}, Ptr{UInt32}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, UInt32, Int32, Ptr{UInt8}, Ptr{Int32}),
        c.h, n, seeds, elems, lo, hi, init_p, trans_p, 0, hi, out, len))
    [String([ALPHABET[out[k, i] + 1] for k in 1:len[i]]) for i in 1:n]
end
const ALPHABET = [collect('a':'z')..., ' ', '.']
# pclean_dummy_seed (include/pclean_philox.h)
function dummy_seed(seed::UInt64, site::UInt32, particle::UInt32, sweep::UInt32)
    x = seed ⊻ ((UInt64(site) << 32) | UInt64(sweep))
    x = (x ⊻ (x >> 30)) * 0xbf58476d1ce4e5b9
    x ⊻= UInt64(particle + 0x1) * 0x94d049bb133111eb
    x = (x ⊻ (x >> 27)) * 0x94d049bb133111eb
    x ⊻ (x >> 31)
end
# ---- what flights (a scoring block of MaybeSwap observations) and rents (a Gaussian observation with own enumerated choices)
# load besides reference-slot blocks; argument meaning in include/pclean_hip.h, tested Python twins in pclean_amd/_lib.py ----
set_options_cols(c, id, cols::Matrix{Int32}, logp::Vector{Float64}) = GC.@preserve cols logp check(c,    # cols: n_options x n_cols
    ccall((:pclean_set_options_cols, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}, Ptr{Float64}),
          c.h, id, size(cols, 1), size(cols, 2), cols, logp))
set_pair_table(c, id, d::Matrix{UInt8}) = GC.@preserve d check(c,                                        # d: n_lat x n_obs (row-major [obs][lat])
    ccall((:pclean_set_pair_table, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{UInt8}), c.h, id, size(d, 2), size(d, 1), d))
set_prob_table(c, p::Vector{Float64}) = GC.@preserve p check(c,                                          # ProbParameter values, maybe_swap.jl:36-52
    ccall((:pclean_set_prob_table, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), c.h, length(p), p))
# a block without a reference slot (flights Obs block 3; block_proposal.jl:62-64): per term the observed column, its 0/1
# same-string table, (block, root column) of the latent value and of the key whose option count is nopt_fn[key]
function load_score_block(c, block, obs_col::Vector{Int32}, pair_table::Vector{Int32}, val_src::Matrix{Int32},
                          key_src::Matrix{Int32}, nopt_fn::Vector{Int32}, other_val::Vector{Int32}, prob_fn,
                          prob_a_src::Vector{Int32}, prob_b_src::Vector{Int32})                           # *_src: 2 x n_terms / length 2
    GC.@preserve obs_col pair_table val_src key_src nopt_fn other_val prob_a_src prob_b_src check(c,
        ccall((:pclean_load_score_block, lib), Cint,
              (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Int32}),
              c.h, block, length(obs_col), obs_col, pair_table, val_src, key_src, nopt_fn, other_val, prob_fn, prob_a_src, prob_b_src))
end
# pclean_gauss (add_noise.jl:1-7, transformed_gaussian.jl:3-17: Normal(mean[index], sigma) on backward(x))
struct CGauss
    x_col::Int32; mean_table::Int32; n_dims::Int32
    src_kind::NTuple{4,Int32}; src::NTuple{4,Int32}; stride::NTuple{4,Int32}
    n_locals::Int32; local_n::NTuple{2,Int32}; local_obs_col::NTuple{2,Int32}
    transform_src_kind::Int32; transform_src::Int32; fixed_locals::Int32; pad::Int32
    t_scale::NTuple{4,Float64}; t_logabsderiv::NTuple{4,Float64}; sigma::Float64
    t_x_col::NTuple{4,Int32}; t_lad_col::NTuple{4,Int32}                      # non-linear Transformations: numeric columns of backward(x) / log|deriv|, -1 = the linear form
end
load_numeric_columns(c, x::Matrix{Float64}) = GC.@preserve x check(c,                                    # x: n_rows x n_cols, NaN = missing
    ccall((:pclean_load_numeric_columns, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}), c.h, size(x, 1), size(x, 2), x))
set_mean_table(c, id, mean::Vector{Float64}) = GC.@preserve mean check(c,
    ccall((:pclean_set_mean_table, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}), c.h, id, length(mean), mean))
set_node_gauss(c, block, node, g::CGauss) = check(c,
    ccall((:pclean_set_node_gauss, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ref{CGauss}), c.h, block, node, Ref(g)))
function get_locals(c, block, n_rows)                                                                    # own choices of the chosen particles
    out = Matrix{Int32}(undef, 2, n_rows)
    GC.@preserve out check(c, ccall((:pclean_get_locals, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}), c.h, block, out)); out
end
# current own choices of every observed row (2 x n_rows, 0-based option indices, -1 = none): what the retained particle of a
# sweep with use_dd_proposals = false keeps (row_inference.jl:143-145; block_proposal.jl:42-56)
set_cur_locals(c, block, locals::Matrix{Int32}) = GC.@preserve locals check(c,
    ccall((:pclean_set_cur_locals, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Int32), c.h, block, locals, size(locals, 2)))

# ---- the device-resident commit (csrc/commit.hip; row_inference.jl:169-185 + dependency_tracking.jl:26-236 for a whole
# sweep): tables uploaded once with spare rows, their allocation state and the rows' referents stay in HBM ----------------
struct CCommitSlot; table_id::Int32; n_hw::Int32; n_free::Int32; cols_changed::Int32; created::Int32; deleted::Int32
                    total::Int64; live::Int64; max_count::Int64; end
struct CCommitSummary; fallback::Int32; n_changed::Int32; n_slots::Int32; stats_reduced::Int32
                       n_records::NTuple{16,Int32}; n_distinct::NTuple{16,Int32}; slot::NTuple{16,CCommitSlot}; end
prepare(c, ev_blocks::Integer=0) = check(c, ccall((:pclean_prepare, lib), Cint, (Ptr{Cvoid}, UInt32), c.h, ev_blocks))
function commit_enable(c, n_blocks)                                                                      # false: this plan commits on the host
    ok = Ref{Int32}(0); check(c, ccall((:pclean_commit_enable, lib), Cint, (Ptr{Cvoid}, Int32, Ref{Int32}), c.h, n_blocks, ok)); ok[] != 0
end
commit_set_table_state(c, table, n_hw, free::Vector{Int32}) = GC.@preserve free check(c,
    ccall((:pclean_commit_set_table_state, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int32}), c.h, table, n_hw, length(free), free))
set_cur(c, cur::Matrix{Int32}) = GC.@preserve cur check(c,                                               # cur: n_rows x n_blocks, 0-based, -1 none
    ccall((:pclean_set_cur, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}), c.h, size(cur, 2), cur))
get_cur!(c, cur::Matrix{Int32}) = GC.@preserve cur check(c,
    ccall((:pclean_get_cur, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}), c.h, size(cur, 2), cur))
set_sweep_mode(c, deferred::Bool) = check(c, ccall((:pclean_set_sweep_mode, lib), Cint, (Ptr{Cvoid}, Int32), c.h, deferred ? 1 : 0))
sweep_fetch(c) = check(c, ccall((:pclean_sweep_fetch, lib), Cint, (Ptr{Cvoid},), c.h))
# one sweep on the device-resident referents (cur == NULL) + its commit; summary.fallback != 0: nothing was modified,
# sweep_fetch(c) and commit on the host (moved / get_new_rows)
function sweep_commit_device!(c, cfg::InferenceConfig, seed, sweep_idx, n_blocks)
    set_sweep_mode(c, true)
    out = Ref{CCommitSummary}()
    try
        check(c, ccall((:pclean_sweep, lib), Cint, (Ptr{Cvoid}, Ref{CConfig}, UInt64, UInt32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}),
                       c.h, Ref(CConfig(cfg)), seed, sweep_idx, n_blocks, C_NULL, C_NULL, C_NULL, C_NULL))
        check(c, ccall((:pclean_commit_device, lib), Cint, (Ptr{Cvoid}, Int32, UInt32, Ref{CCommitSummary}), c.h, n_blocks, sweep_idx, out))
    finally
        set_sweep_mode(c, false)
    end
    out[]
end
# several ranks (one Julia process per GPU; rendezvous id from rank 0 by whatever transport the host program has):
comm_unique_id(c) = (id = zeros(UInt8, 128); check(c, ccall((:pclean_comm_unique_id, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}), c.h, id)); id)
comm_init(c, n_ranks, rank, id::Vector{UInt8}) = GC.@preserve id check(c,
    ccall((:pclean_comm_init, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), c.h, n_ranks, rank, id))
# collective: every rank after its shard's sweep (local_empty: this rank owns no row of the window, no sweep before the call)
function commit_device_dist!(c, n_blocks, sweep_idx, local_empty::Bool, max_local_rows)
    out = Ref{CCommitSummary}()
    check(c, ccall((:pclean_commit_device_dist, lib), Cint, (Ptr{Cvoid}, Int32, UInt32, Int32, Int32, Ref{CCommitSummary}),
                   c.h, n_blocks, sweep_idx, local_empty ? 1 : 0, max_local_rows, out)); out[]
end
function commit_pull_table(c, table, cap, n_cols)                                                        # device state of a latent table -> host
    state = zeros(Int32, 8); cols = Matrix{Int32}(undef, cap, n_cols); counts = Vector{Int64}(undef, cap)
    live = Vector{UInt8}(undef, cap); free = Vector{Int32}(undef, cap); origin = Matrix{Int32}(undef, 4, cap)
    GC.@preserve state cols counts live free origin check(c, ccall((:pclean_commit_pull_table, lib), Cint,
        (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}, Ptr{UInt8}, Ptr{Int32}, Ptr{Int32}), c.h, table, state, cols, counts, live, free, origin))
    state, cols, counts, live, free[1:state[2]], origin
end
# stable argsort of small ids on the device (0-based permutation; the sort behind evidence_csr at 10^6 rows)
function argsort_ids(c, ids::Vector{Int32}, id_max)
    out = Vector{Int32}(undef, length(ids))
    GC.@preserve ids out check(c, ccall((:pclean_argsort_ids, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Int32, Ptr{Int32}),
                                        c.h, length(ids), ids, id_max, out)); out
end
# Evidence sets of a latent class built on the device (pclean_build_evidence): needs the referents on the device (set_cur /
# the device-resident commit) and the tables as uploaded.  steps = [(table id, reference-slot column)] from block cur_block's
# root table down to the class, sources = [(block, table id, value column)] of the per-row ctx values.  Returns the offsets
# (n_target_rows + 1, 0-based positions into the ordered rows that stay on the device).
function build_evidence!(c, cur_block, steps::Vector{Tuple{Int32,Int32}}, n_target_rows, sources::Vector{Tuple{Int32,Int32,Int32}})
    st = Int32[a for (a, _) in steps]; sc = Int32[b for (_, b) in steps]
    sb = Int32[a for (a, _, _) in sources]; stb = Int32[b for (_, b, _) in sources]; scl = Int32[d for (_, _, d) in sources]
    off = Vector{Int32}(undef, n_target_rows + 1)
    GC.@preserve st sc sb stb scl off check(c, ccall((:pclean_build_evidence, lib), Cint,
        (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
        c.h, cur_block, length(st), st, sc, n_target_rows, length(sb), sb, stb, scl, off)); off
end
function get_evidence(c, ev_begin, n)                                                                    # resident rows -> host (checks)
    rows = Vector{Int32}(undef, n); cx = Matrix{Int32}(undef, MAX_CTX, n)
    GC.@preserve rows cx check(c, ccall((:pclean_get_evidence, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Int32}),
                                        c.h, ev_begin, n, rows, cx)); rows, cx
end
# sweep_latent! over rows ev_begin + ev_off[t] .. of the resident evidence (ev_off[1] == 0)
function sweep_latent_resident!(c, cfg, seed, sweep_idx, block, roots::Vector{Int32}, keys::Vector{Int32}, ev_off::Vector{Int32},
                                ev_begin, excl::Matrix{Int32}, n_nodes)
    n = length(keys); chosen = zeros(Int32, n); vals = fill(Int32(-2), n_nodes, n); cc = Ref(CConfig(cfg))
    GC.@preserve roots keys ev_off excl chosen vals check(c, ccall((:pclean_sweep_latent_resident, lib), Cint,
        (Ptr{Cvoid}, Ref{CConfig}, UInt64, UInt32, Int32, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Int32}, Int32,
         Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
        c.h, cc, seed, sweep_idx, block, length(roots), roots, n, keys, ev_off, ev_begin, excl, chosen, vals))
    chosen, vals
end
# the remaining entry points (pclean_allreduce_stats_fused, pclean_random_*, the debug probes) bind the same way; signatures in
# include/pclean_hip.h, tested Python bindings in pclean_amd/_lib.py.

# =============================================================================================== 2. dictionary encoding
mutable struct Pool; index::Dict{String,Int32}; strings::Vector{String}; end
Pool() = Pool(Dict{String,Int32}(), String[])
id!(p::Pool, s::AbstractString) = get!(p.index, String(s)) do; push!(p.strings, String(s)); Int32(length(p.strings) - 1) end
function pool_arrays(p::Pool)                       # dense symbol ids of code points + offsets (pclean_load_strings)
    symid = Dict{Char,UInt16}(); sym = UInt16[]; off = Int64[0]
    for s in p.strings; for ch in s; push!(sym, get!(symid, ch, UInt16(length(symid)))); end; push!(off, length(sym)); end
    sym, off, symid
end
mutable struct Domain; ids::Vector{Int32}; pos::Dict{Int32,Int32}; n_base::Int; end   # value index (0-based) <-> pool id
Domain() = Domain(Int32[], Dict{Int32,Int32}(), 0)
function add!(d::Domain, pool::Pool, s; extra=false)
    pid = id!(pool, s)
    (!extra && haskey(d.pos, pid)) && return d.pos[pid]
    push!(d.ids, pid); j = Int32(length(d.ids) - 1)
    extra || (d.pos[pid] = j; d.n_base = length(d.ids))                       # extras (strings drawn for chosen dummies) are
    j                                                                         # never options: ids after the dummy's
end
value_of(d::Domain, pool::Pool, s) = d.pos[pool.index[String(s)]]

# =============================================================================================== 3. lowering
# PCleanModel + Query + data -> the static plan IR of include/pclean_hip.h, for all three experiment programs (hospital:
# reference slots, AddTypos, JuliaNodes across slots; flights: keyed TimePrior atoms, a scoring block of MaybeSwap
# observations with an indexed ProbParameter; rents: Unmodeled key, keyed StringPrior atoms, a TransformedGaussian with own
# enumerated choices).  Reference structures read: PCleanClass.nodes / blocks / names (model/model.jl:78-118),
# ForeignKeyNode.vmap (169-174), SubmodelNode (176-180), the query's obsmap / cleanmap (dsl/query.jl:1-13).  The reference
# flattens a referenced class INTO the referring class's vertices (dsl/builder.jl:123-175): vertex vmap[i] of class C holds
# the value of vertex i of its slot's target — exactly the flattened columns of a latent table here — and files EVERY copy,
# nested ones included, under the one slot (SubmodelNode(v, i, copy_node(node, v))): the nesting is in `.subnode`, and is
# followed through the TARGET classes' own vertices (resolve, path_below).
#
# THE ALGORITHM BELOW IS PINNED WITHOUT JULIA: tests/julia_lowering.py is this section walked in Python — same functions,
# same names, same order of steps — on the reference's structures as dsl/builder.jl builds them (tests/julia_refmodel.py),
# and tests/test_julia_lowering.py holds its output against tests/golden/plans_{hospital,flights,rents}.json array by array
# and against the product lowering's table contents.  A change here is made there too, function by function.
# Vertex ids are Julia's (1-based); table / column / node / option indices are 0-based as the C ABI wants them.

const DENS_MAYBE_SWAP = Int32(2)
const DUMMY_STRING_PRIOR, DUMMY_TIME_PRIOR = Int32(1), Int32(2)
const GSRC = Dict(:cand => Int32(0), :obs => Int32(1), :local => Int32(2), :itemctx => Int32(3), :evctx => Int32(4))
const DomKey = Tuple{ClassID,VertexID}

"value-carrying vertices of a class in vertex order: own choices, reference slots, their flattened copies"
value_vertices(cm::PCleanClass) = VertexID[v for (v, n) in enumerate(cm.nodes)
                                           if strip_subnodes(n) isa Union{RandomChoiceNode,ForeignKeyNode} && !(n isa ExternalLikelihoodNode)]

"constant arguments of a choice node: JuliaNodes without arguments (the parser wraps literals, dsl/builder.jl:96-100)"
const_args(cm::PCleanClass, node::RandomChoiceNode) = Any[cm.nodes[a].f() for a in node.arg_node_ids
                                                          if cm.nodes[a] isa JuliaNode && isempty(cm.nodes[a].arg_node_ids)]

struct Term; obs::VertexID; path::Vector{VertexID}; pair::Int32; dens::Int32; max_typos::Int32; ctx::Union{Nothing,Tuple{Int32,Int32}}; end
mutable struct LBlock
    root_class::Union{Nothing,ClassID}; root_vertex::Union{Nothing,VertexID}; group::Int32; score::Bool
    nodes::Vector{CNode}; terms::Vector{CTerm}; children::Vector{Int32}; colmap::Vector{Int32}
    ctx_block::Vector{Int32}; ctx_col::Vector{Int32}
    node_class::Vector{ClassID}; node_vertex::Vector{VertexID}                # per node: class + own vertex of the choice / slot
    node_path::Vector{Vector{VertexID}}                                       # per node: slot chain below the block's root
end
LBlock(rc, rv, group) = LBlock(rc, rv, Int32(group), false, CNode[], CTerm[], Int32[], Int32[], Int32[], Int32[], ClassID[], VertexID[], Vector{VertexID}[])
struct CrossTerm; obs::VertexID; pair::Int32; max_typos::Int32; fn::Int32; ctx_block::Int; ctx_path::Vector{VertexID}
                  local_block::Int; local_path::Vector{VertexID}; end
struct ScoreTerm; obs::Int32; pair::Int32; val::Tuple{Int32,Int32}; key::Tuple{Int32,Int32}; nopt_fn::Int32; other::Int32
                  vertex::VertexID; val_vertex::VertexID; end
struct ProbSpec; fn::Int32; a::Tuple{Int32,Int32}; b::Tuple{Int32,Int32}; consts::Vector{Float64}; keys::Vector{Any}; param::VertexID; end
struct GaussSpec
    x_col::Int32; param::VertexID; n_mean::Int; strides::Vector{Int32}; n_locals::Int32; local_n::Vector{Int32}; local_obs::Vector{Int32}
    t_local::Int32; sigma::Float64; t_scale::Vector{Float64}; t_lad::Vector{Float64}
    kinds::Vector{Tuple{Symbol,Int32}}; transform::Tuple{Symbol,Int32}
    mean_keys::Vector{Any}                                                    # mean-table index (strides) => the key the program's own expression asks the parameter for
end
mutable struct Lowered
    model::PCleanModel; query::Query; columns::Vector{Symbol}; pool::Pool
    layout::Dict{ClassID,Vector{VertexID}}                # class => value vertices (= table columns, 0-based index = position - 1)
    col_of::Dict{ClassID,Dict{VertexID,Int32}}
    table_id::Dict{ClassID,Int32}
    dom_keys::Vector{DomKey}; latent_dom::Dict{DomKey,Domain}                 # latent domains in creation order = numbering of the option tables
    option_id::Dict{DomKey,Int32}; option_values::Dict{DomKey,Vector{Int32}}
    option_keycol::Dict{DomKey,Vector{Int32}}; option_ncol::Dict{DomKey,Vector{Int32}}
    keyed::Dict{DomKey,VertexID}                          # a choice whose atoms depend on a key => the key's vertex (same class)
    keyed_order::Vector{DomKey}
    obs_dom::Dict{VertexID,Domain}; obs_col::Dict{VertexID,Int32}; obs_vertices::Vector{VertexID}
    direct_obs::Dict{VertexID,DomKey}                     # noise-free observations: observed vertex => the latent value it shows
    numeric_obs::Dict{VertexID,Int32}; num_cols::Vector{Symbol}; never_missing::Set{VertexID}
    pair_keys::Vector{Tuple{VertexID,Any}}
    pair_id::Dict{Tuple{VertexID,Any},Tuple{Int32,Domain,Vector{Int32}}}      # (obs vertex, latent key) => (id, obs dom, latent pool ids)
    eq_pairs::Dict{DomKey,Tuple{Int32,Int}}; same_pairs::Dict{Int32,Tuple{Domain,Domain}}; next_pair::Int32
    fn_tables::Vector{Matrix{Int32}}                      # stored [column index, row index] = C's fn[row][col]
    blocks::Vector{LBlock}; block_group::Vector{Int32}
    score_blocks::Dict{Int,Tuple{Vector{ScoreTerm},Union{Nothing,ProbSpec}}}; prob_spec::Union{Nothing,ProbSpec}
    gauss::Dict{Tuple{Int32,Int32},GaussSpec}; locals::Dict{Int,Vector{VertexID}}
    cross_terms::Vector{CrossTerm}
    plan_keys::Vector{ClassID}; latent_plans::Dict{ClassID,Any}
    latent_ev_locals::Dict{ClassID,Int}; latent_ev_prob::Dict{ClassID,Int}
    extra_latent::Dict{DomKey,Vector{String}}
end
Lowered(model, query, columns, extra_latent) = Lowered(model, query, columns, Pool(), Dict(), Dict(), Dict(), DomKey[], Dict(), Dict(), Dict(), Dict(), Dict(),
    Dict(), DomKey[], Dict(), Dict(), VertexID[], Dict(), Dict(), Symbol[], Set{VertexID}(), Tuple{VertexID,Any}[], Dict(), Dict(), Dict(), Int32(0),
    Matrix{Int32}[], LBlock[], Int32[], Dict(), nothing, Dict(), Dict(), CrossTerm[], ClassID[], Dict(), Dict(), Dict(), extra_latent)
string_of(lw::Lowered, dom::Domain, j) = lw.pool.strings[dom.ids[j+1]+1]                                  # j: 0-based value index

# ---- helpers over the reference's structures ---------------------------------------------------------------------------
"the class and own vertex a (possibly nested) SubmodelNode vertex `v` of class `cls` stands for"
function resolve(model::PCleanModel, cls::ClassID, v::VertexID)
    n = model.classes[cls].nodes[v]
    while n isa SubmodelNode
        fk = strip_subnodes(model.classes[cls].nodes[n.foreign_key_node_id])::ForeignKeyNode
        v = n.subnode_id; cls = fk.target_class
        n = model.classes[cls].nodes[v]
    end
    cls, v
end
"the reference slot of the observed class whose (possibly nested) copy vertex v is (builder.jl:140-150: every copy is filed under the one slot)"
slot_of_vertex(ocm::PCleanClass, v::VertexID) = (n = ocm.nodes[v]; n isa SubmodelNode ? n.foreign_key_node_id : v)
"vertex ids from slot `fk` down to the value copy vertex v holds: the chain of nested slot vertices (each in ITS class), then the value's own vertex"
function path_below(model::PCleanModel, ocm::PCleanClass, fk::VertexID, v::VertexID)
    cls = (ocm.nodes[fk]::ForeignKeyNode).target_class; u = (ocm.nodes[v]::SubmodelNode).subnode_id; chain = VertexID[]
    while true
        n = model.classes[cls].nodes[u]
        n isa SubmodelNode || (push!(chain, u); return chain)
        push!(chain, n.foreign_key_node_id)
        cls = (model.classes[cls].nodes[n.foreign_key_node_id]::ForeignKeyNode).target_class; u = n.subnode_id
    end
end
"vertex of class cls that holds the value reached by following `path` (nested slot vertices, then the own vertex)"
function flat_vertex(model::PCleanModel, cls::ClassID, path::Vector{VertexID})
    length(path) == 1 && return path[1]
    fk = model.classes[cls].nodes[path[1]]::ForeignKeyNode
    fk.vmap[flat_vertex(model, fk.target_class, path[2:end])]
end
"slot-copy vertices an observed-class vertex ultimately depends on (through AddTypos / JuliaNode arguments)"
function leaf_args(ocm::PCleanClass, v::VertexID)
    n = ocm.nodes[v]
    n isa SubmodelNode && return VertexID[v]
    n isa Union{RandomChoiceNode,JuliaNode} || return VertexID[]
    reduce(vcat, [leaf_args(ocm, a) for a in n.arg_node_ids]; init=VertexID[])
end
"the argument node of a StringPrior / TimePrior choice that computes its atoms from ANOTHER vertex of the class
(`possibilities[countykey]`, `times_for_flight[\"\$flight_id-...\"]`), or nothing when the atoms are a constant"
function keyed_atoms_node(cm::PCleanClass, n::RandomChoiceNode)
    a = cm.nodes[n.arg_node_ids[n.dist isa StringPrior ? 3 : 1]]
    (a isa JuliaNode && !isempty(a.arg_node_ids)) ? a : nothing
end
function dummy_value(n::RandomChoiceNode, cm::PCleanClass)
    n.dist isa TimePrior && return "**:** p.m."                              # time_prior.jl:17-19
    lo, hi = const_args(cm, n)[1:2]; "*"^((lo + hi) ÷ 2)                      # string_prior.jl:24-26
end
"atoms listed under key k, or nothing when nothing is (the key's own dummy value: the program's lookup throws)"
atoms_of_key(anode::JuliaNode, k) = try collect(anode.f(k)) catch; nothing end

# ---- domains --------------------------------------------------------------------------------------------------------------
new_domain!(lw::Lowered, key::DomKey) = (lw.latent_dom[key] = Domain(); push!(lw.dom_keys, key); lw.latent_dom[key])

function build_domains!(lw::Lowered, data)
    m = lw.model
    for cls in m.class_order, (v, n) in enumerate(m.classes[cls].nodes)       # the observed class included: its own discrete choices
        n isa RandomChoiceNode || continue
        cm = m.classes[cls]
        if n.dist isa Union{StringPrior,TimePrior}
            dom = new_domain!(lw, (cls, v)); anode = keyed_atoms_node(cm, n)
            if anode !== nothing
                lw.keyed[(cls, v)] = anode.arg_node_ids[1]; push!(lw.keyed_order, (cls, v))   # filled below, once the key's domain exists
            else
                foreach(s -> add!(dom, lw.pool, s), const_args(cm, n)[3]); add!(dom, lw.pool, dummy_value(n, cm))
            end
        elseif n.dist isa ChooseProportionally || (n.dist isa ChooseUniformly && all(o -> o isa AbstractString, const_args(cm, n)[1]))
            dom = new_domain!(lw, (cls, v)); foreach(s -> add!(dom, lw.pool, s), const_args(cm, n)[1])
        end
    end
    ocm = m.classes[lw.query.class]
    for col in lw.columns                                                     # in the @query's order (the Dict forgets it)
        v = lw.query.obsmap[col]; n = ocm.nodes[v]; column = data[!, col]
        if n isa RandomChoiceNode && n.dist isa Union{AddTypos,MaybeSwap}
            dom = Domain(); foreach(x -> ismissing(x) || add!(dom, lw.pool, string(x)), column); lw.obs_dom[v] = dom
        elseif n isa RandomChoiceNode && n.dist isa TransformedGaussian
            lw.numeric_obs[v] = Int32(length(lw.num_cols)); push!(lw.num_cols, col); continue
        else  # a latent value (or an own discrete choice) observed without noise: the observed domain IS the latent domain
            key = n isa RandomChoiceNode ? (lw.query.class, v) : resolve(m, lw.query.class, v)
            if !haskey(lw.latent_dom, key)                                    # Unmodeled: its values are whatever is observed
                dom = new_domain!(lw, key); foreach(x -> ismissing(x) || add!(dom, lw.pool, string(x)), column)
            end
            lw.obs_dom[v] = lw.latent_dom[key]; lw.direct_obs[v] = key
        end
        lw.obs_col[v] = Int32(length(lw.obs_vertices)); push!(lw.obs_vertices, v)
        any(ismissing, column) || push!(lw.never_missing, v)
    end
    for (cls, v) in lw.keyed_order                                            # keyed atoms: every key's atoms in key order, the dummy last
        cm = m.classes[cls]; n = cm.nodes[v]; dom = lw.latent_dom[(cls, v)]; kdom = lw.latent_dom[(cls, lw.keyed[(cls, v)])]
        anode = keyed_atoms_node(cm, n)
        for j in 0:length(kdom.ids)-1
            atoms = atoms_of_key(anode, string_of(lw, kdom, j))
            atoms === nothing || foreach(s -> add!(dom, lw.pool, s), atoms)
        end
        add!(dom, lw.pool, dummy_value(n, cm))
    end
    for key in lw.dom_keys, s in get(lw.extra_latent, key, String[]); add!(lw.latent_dom[key], lw.pool, s; extra=true); end
end

function build_layouts!(lw::Lowered)
    m = lw.model; next = Int32(0)
    for cls in m.class_order
        cls == lw.query.class && continue
        lw.layout[cls] = value_vertices(m.classes[cls])
        lw.col_of[cls] = Dict(v => Int32(j - 1) for (j, v) in enumerate(lw.layout[cls]))
        lw.table_id[cls] = next; next += 1
    end
    for key in lw.dom_keys
        dom = lw.latent_dom[key]; lw.option_id[key] = next; next += 1
        lw.option_values[key] = Int32.(0:dom.n_base-1)                       # option k of discrete_proposal = value k; the dummy last
        haskey(lw.keyed, key) || continue
        # options = for every key: its atoms, then one dummy option; column 1 = the key, column 2 = its atom count
        cls, v = key; cm = m.classes[cls]; n = cm.nodes[v]; kdom = lw.latent_dom[(cls, lw.keyed[key])]; anode = keyed_atoms_node(cm, n)
        dummy = value_of(dom, lw.pool, dummy_value(n, cm)); vals = Int32[]; keys_ = Int32[]; ncol = Int32[]
        for j in 0:length(kdom.ids)-1
            atoms = atoms_of_key(anode, string_of(lw, kdom, j)); atoms === nothing && continue
            for s in atoms; push!(vals, value_of(dom, lw.pool, s)); push!(keys_, j); end
            push!(vals, dummy); push!(keys_, j); append!(ncol, fill(Int32(length(atoms)), length(atoms) + 1))
        end
        lw.option_values[key] = vals; lw.option_keycol[key] = keys_; lw.option_ncol[key] = ncol
    end
end

# ---- blocks ---------------------------------------------------------------------------------------------------------------
function pair_for!(lw::Lowered, obs_v, key, lat_ids::Vector{Int32})
    if !haskey(lw.pair_id, (obs_v, key))
        lw.pair_id[(obs_v, key)] = (lw.next_pair, lw.obs_dom[obs_v], lat_ids); push!(lw.pair_keys, (obs_v, key)); lw.next_pair += 1
    end
    lw.pair_id[(obs_v, key)][1]
end
"0/1 identity table over a shared domain (the observed value must equal the latent value)"
function eq_pair_for!(lw::Lowered, key::DomKey)
    haskey(lw.eq_pairs, key) || (lw.eq_pairs[key] = (lw.next_pair, length(lw.latent_dom[key].ids)); lw.next_pair += 1)
    lw.eq_pairs[key][1]
end
cterm(lw, t::Term, cand_col) = CTerm(lw.obs_col[t.obs], cand_col, t.pair, t.dens, t.max_typos,
                                     t.ctx === nothing ? -1 : t.ctx[1], t.ctx === nothing ? -1 : t.ctx[2], 0)

"observation terms of one engine block: every AddTypos observation whose latent argument lies below slot `fk`, then the
noise-free observations of values below it (equality constraints)"
function block_terms!(lw::Lowered, ocm::PCleanClass, bi::Int, fk::VertexID, names::Vector{VertexID}, fk_block, blk::LBlock)
    m = lw.model; terms = Term[]
    below(v) = ocm.nodes[v] isa SubmodelNode && slot_of_vertex(ocm, v) == fk
    for v in names
        n = ocm.nodes[v]
        (n isa RandomChoiceNode && n.dist isa AddTypos && haskey(lw.obs_col, v)) || continue
        word = n.arg_node_ids[1]                                             # AddTypos(word[, max_typos]) (add_typos.jl:50)
        mt = length(n.arg_node_ids) > 1 ? Int32(ocm.nodes[n.arg_node_ids[2]].f()) : Int32(-1)
        if below(word)                                                       # obs ~ AddTypos(slot.path)
            cls, own = resolve(m, lw.query.class, word)
            pid = pair_for!(lw, v, (cls, own), lw.latent_dom[(cls, own)].ids)
            push!(terms, Term(v, path_below(m, ocm, fk, word), pid, DENS_ADD_TYPOS, mt, nothing)); continue
        end
        j = ocm.nodes[word]::JuliaNode                                       # obs ~ AddTypos(f(args...))
        locals = [a for a in j.arg_node_ids if below(a)]; others = [a for a in j.arg_node_ids if !below(a)]
        (length(locals) == 1 && length(others) <= 1) || error("JuliaNode under AddTypos: one value of this slot, at most one of an earlier slot")
        lc, lown = resolve(m, lw.query.class, locals[1]); ldom = lw.latent_dom[(lc, lown)]
        if isempty(others)                                                   # f(value): a pair table over the strings f(v)
            ids = Int32[id!(lw.pool, string(j.f(lw.pool.strings[pid+1]))) for pid in ldom.ids]
            push!(terms, Term(v, path_below(m, ocm, fk, locals[1]), pair_for!(lw, v, (:julia, word), ids), DENS_ADD_TYPOS, mt, nothing)); continue
        end
        oc, oown = resolve(m, lw.query.class, others[1]); odom = lw.latent_dom[(oc, oown)]   # f(an earlier slot's value, value)
        ofk = slot_of_vertex(ocm, others[1]); sb = fk_block[ofk]
        sb < bi || error("context must come from an earlier slot")
        src = (Int32(sb - 1), lw.col_of[lw.blocks[sb].root_class][(ocm.nodes[others[1]]::SubmodelNode).subnode_id])
        slot = findfirst(==(src), collect(zip(blk.ctx_block, blk.ctx_col)))  # two JuliaNodes reading the same earlier value share its slot
        if slot === nothing
            length(blk.ctx_block) < MAX_CTX || error("more than $MAX_CTX context values in one block")
            push!(blk.ctx_block, src[1]); push!(blk.ctx_col, src[2]); slot = length(blk.ctx_block)
        end
        order = (findfirst(==(others[1]), j.arg_node_ids), findfirst(==(locals[1]), j.arg_node_ids))
        jdom = Domain(); fn = Matrix{Int32}(undef, length(ldom.ids), length(odom.ids))   # [local, other] = fn[other][local] in C
        for x in 1:length(odom.ids), y in 1:length(ldom.ids)                 # x outer, y inner: the value ids are numbered in this order
            argv = Vector{Any}(undef, 2); argv[order[1]] = lw.pool.strings[odom.ids[x]+1]; argv[order[2]] = lw.pool.strings[ldom.ids[y]+1]
            fn[y, x] = add!(jdom, lw.pool, string(j.f(argv...)))
        end
        push!(lw.fn_tables, fn); fid = Int32(length(lw.fn_tables) - 1)
        pid = pair_for!(lw, v, (:julia, word), jdom.ids); lpath = path_below(m, ocm, fk, locals[1])
        push!(terms, Term(v, lpath, pid, DENS_ADD_TYPOS, mt, (Int32(slot - 1), fid)))
        # the same observation also constrains the OTHER argument's class (external likelihood of e.g. County.state through
        # Record.stateavg_obs): the latent plans of the classes below the earlier slot pick it up (copy_subtree!)
        push!(lw.cross_terms, CrossTerm(v, pid, mt, fid, sb, path_below(m, ocm, ofk, others[1]), bi, lpath))
    end
    for v in lw.obs_vertices                                                 # noise-free observations of values below the root slot
        (haskey(lw.direct_obs, v) && below(v)) || continue
        push!(terms, Term(v, path_below(m, ocm, fk, v), eq_pair_for!(lw, lw.direct_obs[v]), DENS_EQUAL, Int32(-1), nothing))
    end
    terms
end

"emit the node of class `cls` (a reference slot) with the terms whose value lives in its sub-tree; returns its node id"
function emit_fk_node!(lw::Lowered, blk::LBlock, cls::ClassID, slot_vertex, prefix::Vector{VertexID}, terms::Vector{Term}, parent::Int32, parent_fk_col::Int32)
    cm = lw.model.classes[cls]
    nid = Int32(length(blk.nodes)); push!(blk.nodes, CNode(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
    push!(blk.node_class, cls); push!(blk.node_vertex, slot_vertex); push!(blk.node_path, copy(prefix))
    tb = Int32(length(blk.terms))
    for t in terms                                                           # candidate column = the flattened column of the value
        push!(blk.terms, cterm(lw, t, lw.col_of[cls][flat_vertex(lw.model, cls, t.path)]))
    end
    nt = Int32(length(blk.terms)) - tb
    kids = Int32[]; colsrc = Dict{VertexID,Tuple{Int32,Int32}}()
    for (v, n) in enumerate(cm.nodes)                                        # own attributes in declaration (vertex) order
        if n isa ForeignKeyNode
            sub = [Term(t.obs, t.path[2:end], t.pair, t.dens, t.max_typos, t.ctx) for t in terms if length(t.path) > 1 && t.path[1] == v]
            cid = emit_fk_node!(lw, blk, n.target_class, v, vcat(prefix, [v]), sub, nid, lw.col_of[cls][v])
            push!(kids, cid); colsrc[v] = (Int32(-1), Int32(-1))
            for (i, vv) in n.vmap                                            # flattened copies come from the child's columns
                (haskey(lw.col_of[cls], vv) && haskey(lw.col_of[n.target_class], i)) && (colsrc[vv] = (cid, lw.col_of[n.target_class][i]))
            end
        elseif n isa RandomChoiceNode && haskey(lw.latent_dom, (cls, v))
            sub = [t for t in terms if t.path == [v]]
            cid = Int32(length(blk.nodes)); ltb = Int32(length(blk.terms))
            foreach(t -> push!(blk.terms, cterm(lw, t, Int32(0))), sub)
            n_leaf_terms = length(sub)
            if haskey(lw.keyed, (cls, v))  # atoms belong to the key they were listed under: the option's key must equal the row's
                kt = [t for t in terms if t.path == [lw.keyed[(cls, v)]] && t.dens == DENS_EQUAL]
                length(kt) == 1 || error("keyed atoms need their key attribute observed directly")
                push!(blk.terms, cterm(lw, kt[1], Int32(1))); n_leaf_terms += 1
            end
            cacheable = Int32(n_leaf_terms == 1 && length(sub) == 1 && sub[1].ctx === nothing)
            dval, dspec = Int32(0), Int32(0)
            if n.dist isa Union{StringPrior,TimePrior}
                dval = value_of(lw.latent_dom[(cls, v)], lw.pool, dummy_value(n, cm)) + Int32(1)
                dspec = n.dist isa TimePrior ? DUMMY_TIME_PRIOR : (args = const_args(cm, n); Int32(DUMMY_STRING_PRIOR | (args[1] << 8) | (args[2] << 16)))
            end
            push!(blk.nodes, CNode(NODE_LEAF, lw.option_id[(cls, v)], ltb, n_leaf_terms, 0, 0, nid, -1, cacheable, 0, dval, dspec))
            push!(blk.node_class, cls); push!(blk.node_vertex, v); push!(blk.node_path, vcat(prefix, [v]))
            push!(kids, cid); colsrc[v] = (cid, Int32(0))
        end
    end
    cb = Int32(length(blk.children)); append!(blk.children, kids)
    cmb = Int32(length(blk.colmap) ÷ 2)
    for v in lw.layout[cls]; s = get(colsrc, v, (Int32(-1), Int32(-1))); push!(blk.colmap, s[1], s[2]); end
    blk.nodes[nid+1] = CNode(NODE_FK, lw.table_id[cls], tb, nt, cb, length(kids), parent, parent_fk_col, 0, cmb, 0, 0)
    nid
end

"(engine block, column of that block's root table), both 0-based, of the latent value a slot-copy vertex v of the observed class holds"
function value_source(lw::Lowered, ocm::PCleanClass, fk_block, v::VertexID)
    fk = slot_of_vertex(ocm, v)
    (Int32(fk_block[fk] - 1), lw.col_of[(ocm.nodes[fk]::ForeignKeyNode).target_class][(ocm.nodes[v]::SubmodelNode).subnode_id])
end

"what an indexed `@learned` parameter hands back to the lowering's probe: the key that was asked for"
struct KeyRef; key::Any; end
struct KeyProbe end
Base.getindex(::KeyProbe, k) = KeyRef(k)
Base.getindex(::KeyProbe, k...) = KeyRef(k)

"a block without a reference slot: MaybeSwap observations of values chosen in earlier blocks (flights: the last block)"
function lower_score_block!(lw::Lowered, bi::Int, ocm::PCleanClass, names::Vector{VertexID}, fk_block)
    m = lw.model; terms = ScoreTerm[]; prob_spec = nothing
    for v in names
        n = ocm.nodes[v]; n isa RandomChoiceNode || continue
        n.dist isa MaybeSwap || error("a block without a reference slot may only hold MaybeSwap observations")
        val_v, opt_v, prob_v = n.arg_node_ids                                # MaybeSwap(val, options, prob) (maybe_swap.jl:13)
        onode = ocm.nodes[opt_v]::JuliaNode; key_v = onode.arg_node_ids[1]   # options = f(key): a JuliaNode of one slot-copy vertex
        vcls, vown = resolve(m, lw.query.class, val_v); kcls, kown = resolve(m, lw.query.class, key_v)
        vdom, kdom = lw.latent_dom[(vcls, vown)], lw.latent_dom[(kcls, kown)]
        pid = lw.next_pair; lw.next_pair += 1                                # 0/1 "same string" table: observed values x latent domain
        lw.same_pairs[pid] = (lw.obs_dom[v], vdom)
        nopt = ones(Int32, 1, length(kdom.ids))                              # number of options under every key (MaybeSwap's length(options))
        for j in 0:length(kdom.ids)-1
            opts = atoms_of_key(onode, string_of(lw, kdom, j)); opts === nothing || (nopt[1, j+1] = length(opts))
        end
        push!(lw.fn_tables, nopt); fid = Int32(length(lw.fn_tables) - 1)
        vn = m.classes[vcls].nodes[vown]
        push!(terms, ScoreTerm(lw.obs_col[v], pid, value_source(lw, ocm, fk_block, val_v), value_source(lw, ocm, fk_block, key_v), fid,
                               value_of(vdom, lw.pool, dummy_value(vn, m.classes[vcls])), v, val_v))
        prob_spec === nothing || continue
        pn = ocm.nodes[prob_v]                                               # the error probability: a JuliaNode of two values and the parameter
        pn isa JuliaNode || error("MaybeSwap's probability: a JuliaNode of two latent values and an indexed parameter")
        vals = [a for a in pn.arg_node_ids if !(ocm.nodes[a] isa ParameterNode)]; par = [a for a in pn.arg_node_ids if ocm.nodes[a] isa ParameterNode]
        (length(vals) == 2 && length(par) == 1) || error("MaybeSwap's probability: a JuliaNode of two latent values and an indexed parameter")
        acls, aown = resolve(m, lw.query.class, vals[1]); bcls, bown = resolve(m, lw.query.class, vals[2])
        adom, bdom = lw.latent_dom[(acls, aown)], lw.latent_dom[(bcls, bown)]
        keys_ = Any[]; consts = Float64[]; pf = Matrix{Int32}(undef, length(bdom.ids), length(adom.ids))    # [b, a] = fn[a][b] in C
        for x in 0:length(adom.ids)-1, y in 0:length(bdom.ids)-1             # x outer, y inner: constants / keys are numbered in this order
            argv = Dict{VertexID,Any}(vals[1] => string_of(lw, adom, x), vals[2] => string_of(lw, bdom, y), par[1] => KeyProbe())
            r = pn.f((argv[a] for a in pn.arg_node_ids)...)
            if r isa KeyRef
                k = findfirst(==(r.key), keys_); k === nothing && (push!(keys_, r.key); k = length(keys_)); pf[y+1, x+1] = -k
            else
                k = findfirst(==(Float64(r)), consts); k === nothing && (push!(consts, Float64(r)); k = length(consts)); pf[y+1, x+1] = k - 1
            end
        end
        pf = map(e -> e < 0 ? Int32(length(consts) + (-e - 1)) : e, pf)      # prob table: the constants first, then one entry per key
        push!(lw.fn_tables, pf)
        prob_spec = ProbSpec(Int32(length(lw.fn_tables) - 1), value_source(lw, ocm, fk_block, vals[1]), value_source(lw, ocm, fk_block, vals[2]), consts, keys_, par[1])
    end
    lw.score_blocks[bi] = (terms, prob_spec); lw.prob_spec = prob_spec
end

"x ~ TransformedGaussian(param[f(root values, own choices)], std, unit) with own ChooseUniformly choices (experiments/rents/run.jl:19-25)
-> the pclean_gauss specs of the block's root and of the open leaf of its new-row branch"
function lower_gaussian!(lw::Lowered, bi::Int, blk::LBlock, ocm::PCleanClass, names::Vector{VertexID}, fk::VertexID)
    m = lw.model
    ga = [v for v in names if ocm.nodes[v] isa RandomChoiceNode && ocm.nodes[v].dist isa TransformedGaussian]
    isempty(ga) && return
    length(ga) == 1 || error("one Gaussian observation per block")
    g = ocm.nodes[ga[1]]; mean_v, std_v, unit_v = g.arg_node_ids              # TransformedGaussian(mean, std, t) (transformed_gaussian.jl:13)
    look = ocm.nodes[mean_v]
    look isa JuliaNode || error("TransformedGaussian's mean: a JuliaNode indexing ONE learned parameter")
    par = [a for a in look.arg_node_ids if ocm.nodes[a] isa ParameterNode]
    length(par) == 1 || error("TransformedGaussian's mean: a JuliaNode indexing ONE learned parameter")
    units = const_args(ocm, ocm.nodes[unit_v])[1]; t_scale = Float64[]; t_lad = Float64[]
    for u in units  # the kernels evaluate backward(x) as x * backward(1) and log|g'| as a constant: linear Transformations only
        b1 = Float64(u.backward(1.0)); d1 = Float64(u.deriv(b1))
        lin = abs(Float64(u.backward(0.0))) <= 1e-12 && all(x -> abs(u.backward(x) - x * b1) <= 1e-9 * max(1.0, abs(x * b1)) &&
              abs(u.deriv(u.backward(x)) - d1) <= 1e-9 * max(1.0, abs(d1)), (0.5, 2.0, -3.0, 1267.0))
        # (the Python host, model.py: _lower_gaussian, also takes non-linear ones: backward(x) and log|deriv| of every row as two more
        #  numeric columns, pclean_gauss.t_x_col / t_lad_col — not transcribed here)
        lin || error("TransformedGaussian: only linear Transformations (backward(x) = c x)")
        push!(t_scale, b1); push!(t_lad, log(abs(d1)))
    end
    locs = VertexID[]; dims = Tuple{Symbol,Int,Int}[]                        # (kind, payload, n values) per index argument
    for a in look.arg_node_ids
        a in par && continue
        n = ocm.nodes[a]
        if n isa SubmodelNode
            slot_of_vertex(ocm, a) == fk || error("index values come from the block's own slot")
            cn, own = resolve(m, lw.query.class, a); push!(dims, (:cand, a, length(lw.latent_dom[(cn, own)].ids)))
        else
            (n isa RandomChoiceNode && n.dist isa ChooseUniformly) || error("own index arguments must be ChooseUniformly choices")
            a in locs || push!(locs, a)
            push!(dims, (:local, findfirst(==(a), locs) - 1, length(const_args(ocm, n)[1])))
        end
    end
    unit_v in locs || push!(locs, unit_v)
    length(locs) <= 2 || error("at most two enumerated own choices")
    strides = Int32[]; acc = 1
    for d in reverse(dims); pushfirst!(strides, acc); acc *= d[3]; end
    lw.locals[bi] = locs
    # mean-table index => the key the program's own expression asks the parameter for (the host fills pclean_set_mean_table in this order)
    mean_keys = Vector{Any}(undef, acc)
    for idx in 0:acc-1
        argv = Dict{VertexID,Any}(par[1] => KeyProbe())
        for (d, s) in zip(dims, strides)
            j = (idx ÷ s) % d[3]
            a = d[1] == :cand ? d[2] : locs[d[2]+1]
            argv[a] = d[1] == :cand ? string_of(lw, lw.latent_dom[resolve(m, lw.query.class, a)], j) : const_args(ocm, ocm.nodes[a])[1][j+1]
        end
        r = look.f((argv[a] for a in look.arg_node_ids)...); mean_keys[idx+1] = r isa KeyRef ? r.key : r
    end
    t_local = Int32(findfirst(==(unit_v), locs) - 1)
    spec(kinds) = GaussSpec(lw.numeric_obs[ga[1]], par[1], acc, strides, length(locs), Int32[length(const_args(ocm, ocm.nodes[l])[1]) for l in locs],
                            Int32[haskey(lw.direct_obs, l) ? lw.obs_col[l] : -1 for l in locs], t_local, Float64(ocm.nodes[std_v].f()), t_scale, t_lad,
                            kinds, (:local, t_local), mean_keys)
    rc = (ocm.nodes[fk]::ForeignKeyNode).target_class
    # (a) the block's root: candidate-side index values come from the candidate's columns
    lw.gauss[(Int32(bi - 1), Int32(0))] = spec([d[1] == :cand ? (:cand, lw.col_of[rc][(ocm.nodes[d[2]]::SubmodelNode).subnode_id]) : (:local, Int32(d[2])) for d in dims])
    # (b) the new-row branch: the leaf of the ONE candidate-side value that is not always observed carries the term, the others
    #     are read from their direct observations
    open_dims = [d for d in dims if d[1] == :cand && !(haskey(lw.direct_obs, d[2]) && d[2] in lw.never_missing)]
    length(open_dims) == 1 || error("exactly one candidate-side index value may be unobserved")
    open_path = path_below(m, ocm, fk, open_dims[1][2])
    for nid in 1:length(blk.nodes)
        (blk.nodes[nid].kind == NODE_LEAF && blk.node_path[nid] == open_path) || continue
        kinds = [d[1] == :local ? (:local, Int32(d[2])) : d === open_dims[1] ? (:cand, Int32(0)) : (:obs, lw.obs_col[d[2]]) for d in dims]
        lw.gauss[(Int32(bi - 1), Int32(nid - 1))] = spec(kinds)
        n = blk.nodes[nid]                                                   # not cacheable any more
        blk.nodes[nid] = CNode(n.kind, n.table, n.term_begin, n.n_terms, n.child_begin, n.n_children, n.parent, n.parent_fk_col, 0, n.colmap_begin, n.dummy_value, n.dummy_spec)
    end
end

function build_blocks!(lw::Lowered)
    ocm = lw.model.classes[lw.query.class]; fk_block = Dict{VertexID,Int}()
    eblocks = Tuple{Int,Vector{VertexID}}[]                                   # (model block, vertices) per ENGINE block
    for (ub, names) in enumerate(ocm.blocks)
        fks = [v for v in names if ocm.nodes[v] isa ForeignKeyNode]
        if length(fks) <= 1; push!(eblocks, (ub, collect(names))); continue; end
        # several slots in one block: one engine block per slot, an observation goes to the LAST slot it mentions
        groups = [VertexID[f] for f in fks]
        for v in names
            ocm.nodes[v] isa Union{ForeignKeyNode,SubmodelNode} && continue
            hs = [findfirst(==(slot_of_vertex(ocm, a)), fks) for a in leaf_args(ocm, v)]
            hs = [h for h in hs if h !== nothing]
            push!(groups[isempty(hs) ? length(fks) : maximum(hs)], v)
        end
        foreach(g -> push!(eblocks, (ub, g)), groups)
    end
    lw.block_group = Int32[ub - 1 for (ub, _) in eblocks]
    for (bi, (ub, names)) in enumerate(eblocks)                               # slots first: a scoring block reads values of every slot
        fks = [v for v in names if ocm.nodes[v] isa ForeignKeyNode]; isempty(fks) || (fk_block[fks[1]] = bi)
    end
    for (bi, (ub, names)) in enumerate(eblocks)
        fks = [v for v in names if ocm.nodes[v] isa ForeignKeyNode]
        if isempty(fks)
            lower_score_block!(lw, bi, ocm, names, fk_block)
            blk = LBlock(nothing, nothing, ub - 1); blk.score = true; push!(lw.blocks, blk); continue
        end
        fk = fks[1]
        blk = LBlock((ocm.nodes[fk]::ForeignKeyNode).target_class, fk, ub - 1); push!(lw.blocks, blk)
        terms = block_terms!(lw, ocm, bi, fk, names, fk_block, blk)
        emit_fk_node!(lw, blk, blk.root_class, fk, VertexID[], terms, Int32(-1), Int32(-1))
        lower_gaussian!(lw, bi, blk, ocm, names, fk)
    end
end

# ---- latent-class plans ---------------------------------------------------------------------------------------------------
"for every latent class T: the sub-plans of its own attributes, scored against all observed rows that (transitively) refer to a
row of T — the children of T's node in the observed plan, re-rooted"
function build_latent_plans!(lw::Lowered)
    next = Int32(length(lw.blocks))
    for (bi, blk) in enumerate(lw.blocks)
        blk.score && continue
        for (nid, n) in enumerate(blk.nodes)
            cls = blk.node_class[nid]
            (n.kind == NODE_FK && !haskey(lw.latent_plans, cls)) || continue
            plan = (block_id=next, src_block=Int32(bi - 1), src_node=Int32(nid - 1), cls=cls, path=copy(blk.node_path[nid]), nodes=CNode[], terms=CTerm[],
                    children=Int32[], colmap=Int32[], roots=Int32[], root_vertex=VertexID[],
                    ctx_sources=Tuple{Int32,Int32}[collect(zip(blk.ctx_block, blk.ctx_col))...], node_class=ClassID[], node_vertex=VertexID[])
            for k in n.child_begin+1:n.child_begin+n.n_children
                child = blk.children[k] + 1
                push!(plan.roots, copy_subtree!(lw, blk, child, plan, Int32(-1), bi)); push!(plan.root_vertex, blk.node_vertex[child])
            end
            lw.latent_plans[cls] = plan; push!(lw.plan_keys, cls); next += 1
        end
    end
end
function copy_subtree!(lw::Lowered, blk::LBlock, nid::Int, plan, parent::Int32, bi::Int)
    n = blk.nodes[nid]; new_id = Int32(length(plan.nodes)); push!(plan.nodes, n)
    push!(plan.node_class, blk.node_class[nid]); push!(plan.node_vertex, blk.node_vertex[nid])
    tb = Int32(length(plan.terms))
    for t in blk.terms[n.term_begin+1:n.term_begin+n.n_terms]               # the context now comes from the evidence row: fn[ctx][candidate]
        push!(plan.terms, t.ctx_slot >= 0 ? CTerm(t.obs_col, t.cand_col, t.pair_table, t.dens_kind, t.max_typos, t.ctx_slot, t.fn_table, 1) : t)
    end
    # cross-block JuliaNode observations whose OTHER argument lives in this sub-tree: fn[candidate][ctx of the evidence row]
    p = blk.node_path[nid]; ncls = blk.node_class[nid]
    for ct in lw.cross_terms
        ct.ctx_block == bi || continue
        q = ct.ctx_path
        col = if n.kind == NODE_LEAF && p == q; Int32(0)
              elseif n.kind == NODE_FK && length(q) > length(p) && q[1:length(p)] == p; lw.col_of[ncls][flat_vertex(lw.model, ncls, q[length(p)+1:end])]
              else; continue; end
        lb = ct.local_block; lrc = lw.blocks[lb].root_class
        src = (Int32(lb - 1), lw.col_of[lrc][flat_vertex(lw.model, lrc, ct.local_path)])
        if !(src in plan.ctx_sources)
            length(plan.ctx_sources) < MAX_CTX || error("more than $MAX_CTX per-evidence-row context values")
            push!(plan.ctx_sources, src)
        end
        push!(plan.terms, CTerm(lw.obs_col[ct.obs], col, ct.pair, DENS_ADD_TYPOS, ct.max_typos, findfirst(==(src), plan.ctx_sources) - 1, ct.fn, 2))
    end
    # MaybeSwap observations (scoring blocks) of this value: the external likelihood of the referring rows, each with its own
    # error probability (evidence ctx slot 0 = index into the prob table)
    ocm = lw.model.classes[lw.query.class]
    for sbi in sort(collect(keys(lw.score_blocks))), t in lw.score_blocks[sbi][1]
        (n.kind == NODE_LEAF && t.val[1] == bi - 1 && path_below(lw.model, ocm, lw.blocks[bi].root_vertex, t.val_vertex) == p) || continue
        push!(plan.terms, CTerm(t.obs, 0, t.pair, DENS_MAYBE_SWAP, 2, 0, t.other, 1)); lw.latent_ev_prob[plan.cls] = sbi
    end
    nt = Int32(length(plan.terms)) - tb
    if n.kind == NODE_LEAF && haskey(lw.gauss, (Int32(bi - 1), Int32(nid - 1)))
        # the latent sweep of the class owning this value: the referring rows' Gaussian observations, their own choices held at
        # their current values (evidence ctx)
        s = lw.gauss[(Int32(bi - 1), Int32(nid - 1))]
        lw.gauss[(plan.block_id, new_id)] = GaussSpec(s.x_col, s.param, s.n_mean, s.strides, 0, s.local_n, s.local_obs, s.t_local, s.sigma, s.t_scale, s.t_lad,
            [k[1] == :local ? (:evctx, k[2]) : k for k in s.kinds], (:evctx, s.t_local), s.mean_keys)
        lw.latent_ev_locals[plan.cls] = bi
    end
    if n.kind == NODE_FK
        remap = Dict{Int32,Int32}(); kids = Int32[]
        for k in n.child_begin+1:n.child_begin+n.n_children
            c = blk.children[k]; remap[c] = copy_subtree!(lw, blk, c + 1, plan, new_id, bi); push!(kids, remap[c])
        end
        cb = Int32(length(plan.children)); append!(plan.children, kids); cmb = Int32(length(plan.colmap) ÷ 2)
        for j in 0:length(lw.layout[ncls])-1
            cn, cc = blk.colmap[2*(n.colmap_begin+j)+1], blk.colmap[2*(n.colmap_begin+j)+2]
            push!(plan.colmap, cn >= 0 ? remap[cn] : Int32(-1), cc)
        end
        plan.nodes[new_id+1] = CNode(n.kind, n.table, tb, nt, cb, length(kids), parent, n.parent_fk_col, 0, cmb, 0, 0)
    else
        plan.nodes[new_id+1] = CNode(n.kind, n.table, tb, nt, 0, 0, parent, -1, 0, 0, 0, 0)
    end
    new_id
end

"PCleanModel + Query + DataFrame -> Lowered.  `columns`: the @query's column symbols in the order they were written (the
reference's Dicts forget it; the numbering of observed columns, Unmodeled domains and pair tables follows it)."
function lower(model::PCleanModel, query::Query, data, columns::Vector{Symbol}=sort(collect(keys(query.obsmap)));
               extra_latent=Dict{DomKey,Vector{String}}())
    lw = Lowered(model, query, columns, extra_latent)
    build_domains!(lw, data); build_layouts!(lw); build_blocks!(lw); build_latent_plans!(lw)
    lw
end
"arguments of pclean_load_score_block for scoring block bi (1-based engine block index)"
function score_block_args(lw::Lowered, bi::Int)
    t, pr = lw.score_blocks[bi]
    (Int32[x.obs for x in t], Int32[x.pair for x in t], Int32[x.val[i] for i in 1:2, x in t], Int32[x.key[i] for i in 1:2, x in t],
     Int32[x.nopt_fn for x in t], Int32[x.other for x in t], pr.fn, Int32[pr.a...], Int32[pr.b...])
end
"pclean_gauss of a lowered Gaussian spec (engine.py: make_gauss)"
function cgauss(s::GaussSpec; mean_table=0)
    pad4(x, fill_) = ntuple(i -> i <= length(x) ? x[i] : fill_, 4); pad2(x, fill_) = ntuple(i -> i <= length(x) ? x[i] : fill_, 2)
    CGauss(s.x_col, mean_table, length(s.kinds), pad4(Int32[GSRC[k[1]] for k in s.kinds], Int32(0)), pad4(Int32[k[2] for k in s.kinds], Int32(0)),
           pad4(s.strides, Int32(0)), s.n_locals, pad2(s.local_n, Int32(1)), pad2(s.local_obs, Int32(-1)), GSRC[s.transform[1]], s.transform[2], 0, 0,
           pad4(s.t_scale, 1.0), pad4(s.t_lad, 0.0), s.sigma, ntuple(_ -> Int32(-1), 4), ntuple(_ -> Int32(-1), 4))
end
"observed columns as the library wants them: n_rows x n_cols, value index in the column's observed domain, -1 missing"
function encode_observations(lw::Lowered, data)
    obs = fill(Int32(-1), size(data, 1), length(lw.obs_vertices))
    for col in lw.columns
        v = lw.query.obsmap[col]; haskey(lw.obs_col, v) || continue
        for (i, x) in enumerate(data[!, col]); ismissing(x) || (obs[i, lw.obs_col[v]+1] = value_of(lw.obs_dom[v], lw.pool, string(x))); end
    end
    obs
end
"static upload: strings, observed columns, pair / equality / same-string tables, value-function tables, numeric columns, letter
model, plans (engine.py:_upload_static).  Option priors that depend on nothing but the program are uploaded by upload_tables!."
function upload_static!(c::Ctx, lw::Lowered, obs::Matrix{Int32}, data, init_p, trans_p; dist_mode=1)
    sym, off, symid = pool_arrays(lw.pool)
    load_strings(c, sym, off); load_columns(c, obs)
    set_lm_tables(c, init_p, trans_p, UInt16[get(symid, ch, 0xFFFF) for ch in ALPHABET])
    for k in lw.pair_keys; pid, odom, lat_ids = lw.pair_id[k]; build_pair_table(c, pid, odom.ids, lat_ids, dist_mode); end
    for (fid, fn) in enumerate(lw.fn_tables); set_fn_table(c, fid - 1, fn); end
    for (_, (pid, n)) in lw.eq_pairs; set_pair_table(c, pid, UInt8[i == j ? 0 : 1 for i in 1:n, j in 1:n]); end          # 0 on the diagonal
    for (pid, (odom, vdom)) in lw.same_pairs; set_pair_table(c, pid, UInt8[o == l ? 0 : 1 for l in vdom.ids, o in odom.ids]); end   # [lat, obs]
    isempty(lw.num_cols) || load_numeric_columns(c, Float64[ismissing(data[i, col]) ? NaN : Float64(data[i, col]) for i in 1:size(data, 1), col in lw.num_cols])
    for (bi, b) in enumerate(lw.blocks)
        b.score ? load_score_block(c, bi - 1, score_block_args(lw, bi)...) : load_block(c, bi - 1, b.nodes, b.terms, b.children, b.colmap, b.ctx_block, b.ctx_col)
    end
    for cls in lw.plan_keys
        pl = lw.latent_plans[cls]                                            # latent-mode ctx terms read the evidence row's ctx slots
        nctx = max(haskey(lw.latent_ev_locals, cls) ? 2 : 1, length(pl.ctx_sources))
        load_block(c, pl.block_id, pl.nodes, pl.terms, pl.children, pl.colmap, zeros(Int32, nctx), zeros(Int32, nctx))
    end
    length(unique(lw.block_group)) < length(lw.block_group) && foreach(bi -> set_block_group(c, bi - 1, lw.block_group[bi]), eachindex(lw.block_group))
end
"after the first pclean_set_mean_table: the Gaussian specs of the blocks and latent plans (engine.py:_upload_gauss)"
upload_gauss!(c::Ctx, lw::Lowered) = for ((bid, nid), s) in lw.gauss; set_node_gauss(c, bid, nid, cgauss(s)); end
"the plan arrays as nested Dicts/Vectors, for diffing against tests/golden/plans_*.json"
function plan_json(lw::Lowered)
    nodes(ns) = [collect(Int, (n.kind, n.table, n.term_begin, n.n_terms, n.child_begin, n.n_children, n.parent, n.parent_fk_col, n.cacheable,
                               n.colmap_begin, n.dummy_value, n.dummy_spec)) for n in ns]
    terms(ts) = [collect(Int, (t.obs_col, t.cand_col, t.pair_table, t.dens_kind, t.max_typos, t.ctx_slot, t.fn_table, t.ctx_mode)) for t in ts]
    Dict("blocks" => [b.score ? Dict("score" => true, "args" => [x isa Integer ? Int(x) : Int.(vec(x)) for x in score_block_args(lw, bi)]) :
                      Dict("root_class" => String(b.root_class), "nodes" => nodes(b.nodes), "terms" => terms(b.terms), "children" => Int.(b.children),
                           "colmap" => Int.(b.colmap), "ctx_src_block" => Int.(b.ctx_block), "ctx_src_col" => Int.(b.ctx_col)) for (bi, b) in enumerate(lw.blocks)],
         "latent_plans" => Dict(String(cls) => (pl = lw.latent_plans[cls]; Dict("block_id" => Int(pl.block_id), "src_block" => Int(pl.src_block),
                           "roots" => Int.(pl.roots), "nodes" => nodes(pl.nodes), "terms" => terms(pl.terms), "children" => Int.(pl.children),
                           "colmap" => Int.(pl.colmap))) for cls in lw.plan_keys),
         "table_id" => Dict(String(k) => Int(v) for (k, v) in lw.table_id),
         "fn_tables" => Dict(string(i - 1) => [size(f, 2), size(f, 1)] for (i, f) in enumerate(lw.fn_tables)))
end

# =============================================================================================== 4. trace <-> flat tables
# Latent rows get dense 0-based slots per class (`slot_of[class][key]`, `key_of[class][slot+1]`, `free[class]`).
mutable struct Slots; slot_of::Dict{ClassID,Dict{Key,Int32}}; key_of::Dict{ClassID,Vector{Union{Key,Nothing}}}; free::Dict{ClassID,Vector{Int32}}; end
Slots(lw::Lowered) = Slots(Dict(c => Dict{Key,Int32}() for c in keys(lw.table_id)), Dict(c => Union{Key,Nothing}[] for c in keys(lw.table_id)),
                           Dict(c => Int32[] for c in keys(lw.table_id)))
function slot!(s::Slots, cls::ClassID, key::Key)
    get!(s.slot_of[cls], key) do
        if !isempty(s.free[cls]); j = pop!(s.free[cls]); s.key_of[cls][j+1] = key; j
        else; push!(s.key_of[cls], key); Int32(length(s.key_of[cls]) - 1); end
    end
end
function release_dead!(s::Slots, trace::PCleanTrace)
    for (cls, m) in s.slot_of, (key, j) in collect(m)
        haskey(trace.tables[cls].rows, key) || (delete!(m, key); s.key_of[cls][j+1] = nothing; push!(s.free[cls], j))
    end
end
"TableTrace -> pclean_set_table / pclean_set_options (engine.py:upload_trace)"
function upload_tables!(c::Ctx, lw::Lowered, trace::PCleanTrace, s::Slots)
    release_dead!(s, trace)
    for cls in lw.model.class_order
        haskey(lw.table_id, cls) || continue
        table = trace.tables[cls]; foreach(k -> slot!(s, cls, k), keys(table.rows))
        n = length(s.key_of[cls]); cols = zeros(Int32, n, length(lw.layout[cls])); counts = zeros(Int64, n)
        for (key, j) in s.slot_of[cls]
            row = table.rows[key]; counts[j+1] = get(table.reference_counts, key, 0)
            for (jj, v) in enumerate(lw.layout[cls])
                node = strip_subnodes(lw.model.classes[cls].nodes[v])
                if node isa ForeignKeyNode
                    cols[j+1, jj] = slot!(s, node.target_class, row[v])
                else
                    oc, ov = resolve(lw.model, cls, v); cols[j+1, jj] = latent_value!(lw, oc, ov, row[v])
                end
            end
        end
        set_table(c, lw.table_id[cls], cols, counts, table.pitman_yor_params)
    end
    for key in lw.dom_keys                                                    # option priors: discrete_proposal(dist, args...) (distributions.jl:16)
        cls, v = key; cm = lw.model.classes[cls]; node = cm.nodes[v]; oid = lw.option_id[key]; vals = lw.option_values[key]
        if haskey(lw.keyed, key)  # per key the proposal over ITS atoms (string_prior.jl:16-22, time_prior.jl:8-22): atom scores, then that key's dummy mass
            anode = keyed_atoms_node(cm, node); kdom = lw.latent_dom[(cls, lw.keyed[key])]; logp = Float64[]
            for j in 0:length(kdom.ids)-1
                atoms = atoms_of_key(anode, string_of(lw, kdom, j)); atoms === nothing && continue
                args = node.dist isa StringPrior ? Any[const_args(cm, node)[1:2]..., atoms] : Any[atoms]
                _, lps = discrete_proposal(node.dist, args...); append!(logp, Float64.(lps))
            end
            cols = node.dist isa TimePrior ? hcat(vals, lw.option_keycol[key], lw.option_ncol[key]) : hcat(vals, lw.option_keycol[key])
            set_options_cols(c, oid, cols, logp)
        elseif node.dist isa Unmodeled                                        # logdensity 0 (unmodeled.jl:7-10)
            cls == lw.query.class || set_options(c, oid, vals, zeros(length(vals)))
        else
            args = const_args(cm, node)
            node.dist isa ChooseProportionally && (args = Any[args[1], trace.tables[cls].parameters[node.arg_node_ids[2]].current_value])
            _, lps = discrete_proposal(node.dist, args...)
            set_options(c, oid, vals, Float64.(lps))
        end
    end
    if lw.prob_spec !== nothing                                               # MaybeSwap's error probabilities: the program's constants, then one entry per key
        p = trace.tables[lw.query.class].parameters[lw.prob_spec.param]       # IndexedParameter: p[key] is the key's ProbParameter (distributions.jl:50-55)
        set_prob_table(c, Float64[lw.prob_spec.consts..., (p[k].current_value for k in lw.prob_spec.keys)...])
    end
    if !isempty(lw.gauss)                                                     # the Gaussian's means in mean-table order (lower_gaussian!: mean_keys)
        g = first(values(lw.gauss)); p = trace.tables[lw.query.class].parameters[g.param]
        set_mean_table(c, 0, Float64[p[k].current_value for k in g.mean_keys]); upload_gauss!(c, lw)
    end
end
"value index of a string held by the trace; a string that is no proposal atom (drawn for a chosen dummy) is registered as an extra"
function latent_value!(lw::Lowered, cls, v, s)
    dom = lw.latent_dom[(cls, v)]; pid = id!(lw.pool, string(s))
    haskey(dom.pos, pid) && return dom.pos[pid]
    j = findfirst(==(pid), dom.ids)                                           # an extra registered earlier
    j !== nothing ? Int32(j - 1) : error("value $(s) of $(cls) is not in the lowered domain: grow_domain! first")
end
"n_rows x n_blocks matrix of the rows' current referents (0-based slots, -1 = none yet)"
current_referents(trace, lw::Lowered, s::Slots) =
    Int32[haskey(trace.tables[lw.query.class].rows, i) ? s.slot_of[b.root_class][trace.tables[lw.query.class].rows[i][b.root_vertex]] : -1
          for i in 1:length(trace.tables[lw.query.class].observations), b in lw.blocks]

# =============================================================================================== 5. commits
"RowTrace entries (vertex => value) of observed row `row` below slot block b, for existing referent `key`"
function splice_referent!(row, lw::Lowered, trace, b::LBlock, key::Key)
    ocm = lw.model.classes[lw.query.class]; fk = ocm.nodes[b.root_vertex]::ForeignKeyNode
    row[b.root_vertex] = key
    for (i, v) in fk.vmap                                                    # flattened copies (dependency_tracking.jl:88-96)
        haskey(trace.tables[b.root_class].rows[key], i) && (row[v] = trace.tables[b.root_class].rows[key][i])
    end
end
"build the latent rows a new-row record describes (top-down over the block's nodes) and return the root's key"
function materialise!(trace, lw::Lowered, b::LBlock, node::Int, vals::AbstractVector{Int32}, s::Slots)
    cls = b.node_class[node]; cm = lw.model.classes[cls]; key = pclean_gensym!("row"); row = Dict{VertexID,Any}()
    n = b.nodes[node]
    for k in n.child_begin+1:n.child_begin+n.n_children
        cid = b.children[k] + 1; cn = b.nodes[cid]; v = b.node_vertex[cid]; choice = vals[cid]
        if cn.kind == NODE_LEAF                                              # option index -> string (the dummy stays a placeholder here)
            dom = lw.latent_dom[(cls, v)]; row[v] = lw.pool.strings[dom.ids[lw.option_values[(cls, v)][choice+1]+1]+1]
        else                                                                 # nested slot: existing row or a new one
            tcls = b.node_class[cid]
            ckey = choice >= 0 ? s.key_of[tcls][choice+1] : materialise!(trace, lw, b, cid, vals, s)
            row[v] = ckey
            for (i, vv) in (cm.nodes[v]::ForeignKeyNode).vmap
                haskey(trace.tables[tcls].rows[ckey], i) && (row[vv] = trace.tables[tcls].rows[ckey][i])
            end
        end
    end
    trace.tables[cls].rows[key] = row                                        # incorporate_row! of the REFERRING row bumps the counts
    slot!(s, cls, key); key
end
"apply one observed-class sweep: only the moved rows are touched (run_smc! tail, row_inference.jl:169-185)"
function commit!(trace, lw::Lowered, s::Slots, block0::Int, rows, ch, new_rows, new_vals)
    cls = lw.query.class; b = lw.blocks[block0+1]; newpos = Dict(r => j for (j, r) in enumerate(new_rows))
    origin = Dict{Key,Tuple{Int,Int}}()                                      # new latent row => (creating row, chosen particle)
    for (r, c0) in zip(rows, ch)
        i = Int(r) + 1
        haskey(trace.tables[cls].rows, i) && unincorporate_row!(trace, cls, i)   # drops the old references (GC of orphans)
        row = get(trace.tables[cls].rows, i, copy(trace.tables[cls].observations[i]))
        if c0 >= 0
            splice_referent!(row, lw, trace, b, s.key_of[b.root_class][c0+1])
        else
            vals = view(new_vals, :, newpos[r]); key = materialise!(trace, lw, b, 1, vals, s)
            origin[key] = (Int(r), Int(-1 - vals[1])); splice_referent!(row, lw, trace, b, key)
        end
        trace.tables[cls].rows[i] = row; incorporate_row!(trace, cls, i)
    end
    origin
end
"apply a latent-class sweep: rows whose chosen particle is fresh take the sampled values (inference.py:commit_latent)"
function commit_latent!(trace, lw::Lowered, s::Slots, cls::ClassID, live::Vector{Int32}, chosen, vals)
    pl = lw.latent_plans[cls]
    for (t, c) in enumerate(chosen)
        c > 0 || continue
        key = s.key_of[cls][live[t]+1]; row = trace.tables[cls].rows[key]
        for (r, root) in enumerate(pl.roots)
            v = pl.root_vertex[r]; n = pl.nodes[root+1]; choice = vals[root+1, t]
            if n.kind == NODE_LEAF
                dom = lw.latent_dom[(cls, v)]; row[v] = lw.pool.strings[dom.ids[lw.option_values[(cls, v)][choice+1]+1]+1]
            else
                tcls = pl.node_class[root+1]
                row[v] = choice >= 0 ? s.key_of[tcls][choice+1] : materialise!(trace, lw, LBlock(cls, v, 0, pl.nodes, pl.terms, pl.children, pl.colmap, Int32[], Int32[], pl.node_class, pl.node_vertex), root + 1, view(vals, :, t), s)
            end
        end
        # reference counts, hashed keys and the flattened copies held by referring rows: the reference's own routines
        PClean.update_referring_rows_with_new_values_for_updated_row!(trace, cls, key)   # dependency_tracking.jl:239-257
    end
end
"chosen ProposalDummyValues (block_proposal.jl:58-60): the placeholder of a freshly created row becomes the string the sweep weighed"
function resample_dummies!(c::Ctx, trace, lw::Lowered, origin, block0::Int, seed::UInt64, sweep_idx::UInt32, init_p, trans_p)
    b = lw.blocks[block0+1]; changed = false
    for (node, n) in enumerate(b.nodes)
        (n.kind == NODE_LEAF && n.dummy_value != 0 && (n.dummy_spec & 0xff) == 1) || continue
        cls, v = b.node_class[node], b.node_vertex[node]
        ph = lw.pool.strings[lw.latent_dom[(cls, v)].ids[n.dummy_value]+1]
        holders = [(key, o) for (key, o) in origin if haskey(trace.tables[cls].rows, key) && trace.tables[cls].rows[key][v] == ph]
        isempty(holders) && continue
        site = UInt32(((block0) << 16) | (node - 1))
        seeds = UInt64[dummy_seed(seed, site, UInt32(o[2]), sweep_idx) for (_, o) in holders]
        strs = random_string_prior_at(c, seeds, UInt32[o[1] for (_, o) in holders], (n.dummy_spec >> 8) & 0xff, (n.dummy_spec >> 16) & 0xff, init_p, trans_p)
        for ((key, _), s_) in zip(holders, strs)
            trace.tables[cls].rows[key][v] = s_; add!(lw.latent_dom[(cls, v)], lw.pool, s_; extra=true); changed = true
            push!(get!(lw.extra_latent, (cls, v), String[]), s_)
        end
    end
    changed                                                                   # true: re-run lower(...; extra_latent) + upload_static! (Engine.reload)
end
"evidence CSR of a latent class: observed rows referring (transitively) to each live latent row (inference.py:build_evidence)"
function evidence_csr(trace, lw::Lowered, s::Slots, cls::ClassID)
    pl = lw.latent_plans[cls]; ocls = lw.query.class; b = lw.blocks[pl.src_block+1]
    n = length(trace.tables[ocls].rows); keyslot = Vector{Int32}(undef, n)
    for i in 1:n                                                             # follow the slots from the block's root down to cls
        keyslot[i] = follow(trace, lw, s, b, trace.tables[ocls].rows[i][b.root_vertex], cls)
    end
    live = sort(Int32[j for (_, j) in s.slot_of[cls]]); order = sortperm(keyslot)
    counts = zeros(Int32, length(s.key_of[cls])); foreach(k -> counts[k+1] += 1, keyslot)
    live, Int32[0; cumsum(counts[live.+1])], Int32.(order .- 1)
end
function follow(trace, lw, s, b::LBlock, key, cls)
    c = b.root_class
    while c != cls                                                           # descend through the slot that leads to cls
        fkv = first(v for (v, n) in enumerate(lw.model.classes[c].nodes) if n isa ForeignKeyNode && reaches(lw.model, n.target_class, cls))
        key = trace.tables[c].rows[key][fkv]; c = (lw.model.classes[c].nodes[fkv]::ForeignKeyNode).target_class
    end
    s.slot_of[cls][key]
end
reaches(model, from, to) = from == to || any(n isa ForeignKeyNode && reaches(model, n.target_class, to) for n in model.classes[from].nodes)

# =============================================================================================== 6. pgibbs_sweep! with the HIP path
"drop-in for PClean.pgibbs_sweep! (inference/inference.jl:60-81): same class order, parameters every rejuv_frequency rows"
function pgibbs_sweep_hip!(c::Ctx, trace::PCleanTrace, lw::Lowered, s::Slots, config::InferenceConfig, seed::UInt64, iter::Int, init_p, trans_p)
    for cls in trace.model.class_order
        upload_tables!(c, lw, trace, s)
        if cls == lw.query.class
            sweep!(c, config, seed, UInt32(iter), current_referents(trace, lw, s))
            for b0 in 0:length(lw.blocks)-1
                new_rows, new_vals = get_new_rows(c, b0, length(lw.blocks[b0+1].nodes)); rows, ch = moved(c, b0)
                origin = commit!(trace, lw, s, b0, rows, ch, new_rows, new_vals)
                resample_dummies!(c, trace, lw, origin, b0, seed, UInt32(iter), init_p, trans_p) &&
                    error("latent domains grew: re-lower with lw.extra_latent and upload_static! (pclean_amd/engine.py: Engine.reload)")
            end
        elseif haskey(lw.latent_plans, cls)
            pl = lw.latent_plans[cls]; live, ev_off, ev_rows = evidence_csr(trace, lw, s, cls)
            excl = Int32[pl.nodes[r+1].kind == NODE_FK ? s.slot_of[pl.node_class[r+1]][trace.tables[cls].rows[s.key_of[cls][k+1]][pl.root_vertex[i]]] : -1
                         for k in live, (i, r) in enumerate(pl.roots)]
            chosen, vals = sweep_latent!(c, config, seed, UInt32(iter), pl.block_id, pl.roots, live, ev_off, ev_rows, nothing, excl, length(pl.nodes))
            commit_latent!(trace, lw, s, cls, live, chosen, vals)
        end
        for (_, param) in trace.tables[cls].parameters; resample_value!(param); end   # inference.jl:72-77, once per class sweep here
        resample_py_params!(trace.tables[cls])
    end
end
end # module
