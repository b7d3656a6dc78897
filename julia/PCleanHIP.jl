# PCleanHIP.jl — the Julia side of the drop-in: `ccall` bindings of include/pclean_hip.h, the lowering of a
# PCleanModel + Query into the static plan IR, the trace <-> flat-table conversion and the commits.
#
# STATUS: written against the reference's sources (/root/reference/src, cited per function) and against the tested
# Python host (pclean_amd/: _lib.py = bindings, model.py = lowering, trace.py / parallel.py / inference.py = commits).
# The build image has no Julia, so this file has NEVER been executed.  What pins it: tests/golden/plans_*.json hold the
# plan arrays the Python lowering produces for the three experiment programs (generator scripts/make_plan_goldens.py);
# `lower(model, query, data)` below must produce the same arrays up to the numbering of domains (dump with
# `plan_json(lw)` and diff).  Every function the pgibbs_sweep! patch at the bottom calls is defined in this file.
#
# Include after `using PClean` from the package's own module scope:  include("PCleanHIP.jl")
module PCleanHIP

using ..PClean: PCleanModel, PCleanClass, PCleanTrace, TableTrace, Query, ObservedDataset, InferenceConfig,
                ForeignKeyNode, RandomChoiceNode, JuliaNode, SubmodelNode, ParameterNode, ExternalLikelihoodNode,
                PCleanNode, VertexID, ClassID, Key, Plan, Step, ProposalDummyValue, AddTypos, StringPrior, TimePrior,
                ChooseUniformly, ChooseProportionally, strip_subnodes, has_discrete_proposal, discrete_proposal,
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
struct CCommitSummary; fallback::Int32; n_changed::Int32; n_slots::Int32; pad::Int32
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
# Mirrors pclean_amd/model.py: LoweredModel.  Reference structures read: PCleanClass.nodes / blocks / names
# (model/model.jl:78-118), ForeignKeyNode.vmap (169-174), SubmodelNode (176-180), the query's obsmap / cleanmap
# (dsl/query.jl:1-13).  The reference flattens a referenced class INTO the referring class's vertices
# (dsl/builder.jl:123-175): vertex vmap[i] of class C holds the value of vertex i of its slot's target — exactly
# the flattened columns of a latent table here.

"value-carrying vertices of a class in vertex order: own choices, reference slots, their flattened copies"
value_vertices(cm::PCleanClass) = [v for (v, n) in enumerate(cm.nodes)
                                   if strip_subnodes(n) isa Union{RandomChoiceNode,ForeignKeyNode} && !(n isa ExternalLikelihoodNode)]

"constant arguments of a choice node: JuliaNodes without arguments (the parser wraps literals, dsl/syntax.jl:130-134)"
const_args(cm::PCleanClass, node::RandomChoiceNode) = [cm.nodes[a].f() for a in node.arg_node_ids
                                                       if cm.nodes[a] isa JuliaNode && isempty(cm.nodes[a].arg_node_ids)]

struct Term; obs::VertexID; path::Vector{VertexID}; pair::Int32; max_typos::Int32; ctx::Union{Nothing,Tuple{Int32,Int32}}; end
mutable struct LBlock
    root_class::ClassID; root_vertex::VertexID; group::Int32
    nodes::Vector{CNode}; terms::Vector{CTerm}; children::Vector{Int32}; colmap::Vector{Int32}
    ctx_block::Vector{Int32}; ctx_col::Vector{Int32}
    node_class::Vector{ClassID}; node_vertex::Vector{VertexID}                # per node: class + own vertex of the choice / slot
end
mutable struct Lowered
    model::PCleanModel; query::Query; pool::Pool
    layout::Dict{ClassID,Vector{VertexID}}                # class => value vertices (= table columns, 0-based index = position - 1)
    col_of::Dict{ClassID,Dict{VertexID,Int32}}
    table_id::Dict{ClassID,Int32}; option_id::Dict{Tuple{ClassID,VertexID},Int32}
    latent_dom::Dict{Tuple{ClassID,VertexID},Domain}      # own choice vertex of a latent class => its value domain
    option_values::Dict{Tuple{ClassID,VertexID},Vector{Int32}}
    obs_dom::Dict{VertexID,Domain}; obs_col::Dict{VertexID,Int32}; obs_vertices::Vector{VertexID}
    pair_id::Dict{Tuple{VertexID,Any},Tuple{Int32,Domain,Vector{Int32}}}      # (obs vertex, latent key) => (id, obs dom, latent pool ids)
    fn_tables::Vector{Matrix{Int32}}
    blocks::Vector{LBlock}; latent_plans::Dict{ClassID,Any}
    extra_latent::Dict{Tuple{ClassID,VertexID},Vector{String}}
end

"the class and own vertex a (possibly nested) SubmodelNode vertex `v` of class `cls` stands for"
function resolve(model::PCleanModel, cls::ClassID, v::VertexID)
    n = model.classes[cls].nodes[v]
    while n isa SubmodelNode
        fk = model.classes[cls].nodes[n.foreign_key_node_id]
        fk = fk isa SubmodelNode ? strip_subnodes(fk) : fk
        v = n.subnode_id; cls = fk.target_class
        n = model.classes[cls].nodes[v]
    end
    cls, v
end

function build_domains!(lw::Lowered, data)
    m = lw.model
    for cls in m.class_order, (v, n) in enumerate(m.classes[cls].nodes)
        (n isa RandomChoiceNode && has_discrete_proposal(n.dist)) || continue
        cls == lw.query.class && continue
        args = const_args(m.classes[cls], n)
        dom = Domain()
        if n.dist isa StringPrior
            foreach(s -> add!(dom, lw.pool, s), args[3]); add!(dom, lw.pool, discrete_proposal_dummy_value(n.dist, args...))
        elseif n.dist isa Union{ChooseUniformly,ChooseProportionally}
            foreach(s -> add!(dom, lw.pool, string(s)), args[1])
        else
            opts, _ = discrete_proposal(n.dist, args...)                      # TimePrior and user distributions: atoms + dummy
            foreach(o -> add!(dom, lw.pool, o isa ProposalDummyValue ? discrete_proposal_dummy_value(n.dist, args...) : string(o)), opts)
        end
        foreach(s -> add!(dom, lw.pool, s; extra=true), get(lw.extra_latent, (cls, v), String[]))
        lw.latent_dom[(cls, v)] = dom
    end
    for (col, v) in lw.query.obsmap                                           # observed columns: unique non-missing values
        dom = Domain(); foreach(x -> ismissing(x) || add!(dom, lw.pool, string(x)), data[!, col])
        lw.obs_dom[v] = dom; lw.obs_col[v] = Int32(length(lw.obs_vertices)); push!(lw.obs_vertices, v)
    end
end

function build_layouts!(lw::Lowered)
    next = Int32(0)
    for cls in lw.model.class_order
        cls == lw.query.class && continue
        lw.layout[cls] = value_vertices(lw.model.classes[cls])
        lw.col_of[cls] = Dict(v => Int32(j - 1) for (j, v) in enumerate(lw.layout[cls]))
        lw.table_id[cls] = next; next += 1
    end
    for ((cls, v), dom) in sort(collect(lw.latent_dom); by=x -> (findfirst(==(x[1][1]), lw.model.class_order), x[1][2]))
        lw.option_id[(cls, v)] = next; next += 1
        lw.option_values[(cls, v)] = Int32.(0:dom.n_base-1)                  # option k of discrete_proposal = value k; dummy last
    end
end

"observation terms of one engine block: every observed AddTypos vertex whose latent argument lies below slot `fk`"
function block_terms!(lw::Lowered, ocm::PCleanClass, bi::Int, fk::VertexID, names::Vector{VertexID}, fk_block, blk::LBlock)
    m = lw.model; terms = Term[]
    below(v) = (n = ocm.nodes[v]; n isa SubmodelNode && slot_of_vertex(ocm, v) == fk)
    for v in names
        n = ocm.nodes[v]
        (n isa RandomChoiceNode && n.dist isa AddTypos && haskey(lw.obs_col, v)) || continue
        word = n.arg_node_ids[1]                                             # AddTypos(word[, max_typos]) (add_typos.jl:50)
        mt = length(n.arg_node_ids) > 1 ? Int32(ocm.nodes[n.arg_node_ids[2]].f()) : Int32(-1)
        if below(word)                                                       # obs ~ AddTypos(slot.path)
            cls, own = resolve(m, lw.query.class, word)
            pid = pair_for!(lw, v, (cls, own), lw.latent_dom[(cls, own)].ids)
            push!(terms, Term(v, path_below(ocm, fk, word), pid, mt, nothing))
        else                                                                 # obs ~ AddTypos(f(args...)): a JuliaNode
            j = ocm.nodes[word]::JuliaNode
            locals = [a for a in j.arg_node_ids if below(a)]; others = [a for a in j.arg_node_ids if !below(a)]
            (length(locals) == 1 && length(others) <= 1) || error("JuliaNode under AddTypos: one value of this slot, at most one of an earlier slot")
            lc, lown = resolve(m, lw.query.class, locals[1]); ldom = lw.latent_dom[(lc, lown)]
            if isempty(others)                                               # f(value): pair table over the strings f(v)
                ids = Int32[id!(lw.pool, string(j.f(lw.pool.strings[pid+1]))) for pid in ldom.ids]
                push!(terms, Term(v, path_below(ocm, fk, locals[1]), pair_for!(lw, v, (:julia, word), ids), mt, nothing))
            else                                                             # f(earlier slot's value, value): ctx + fn table
                oc, oown = resolve(m, lw.query.class, others[1]); odom = lw.latent_dom[(oc, oown)]
                sb = fk_block[slot_of_vertex(ocm, others[1])]; sb < bi || error("context must come from an earlier slot")
                src = (Int32(sb - 1), lw.col_of[lw.blocks[sb].root_class][vertex_in_target(ocm, others[1])])
                slot = findfirst(==(src), collect(zip(blk.ctx_block, blk.ctx_col)))
                if slot === nothing
                    push!(blk.ctx_block, src[1]); push!(blk.ctx_col, src[2]); slot = length(blk.ctx_block)
                    slot <= MAX_CTX || error("more than $MAX_CTX context values in one block")
                end
                jdom = Domain(); fn = Matrix{Int32}(undef, length(ldom.ids), length(odom.ids))   # [local, other] = fn[other][local] in C
                order = (findfirst(==(others[1]), j.arg_node_ids), findfirst(==(locals[1]), j.arg_node_ids))
                for (x, xo) in enumerate(odom.ids), (y, yl) in enumerate(ldom.ids)
                    argv = Vector{Any}(undef, 2); argv[order[1]] = lw.pool.strings[xo+1]; argv[order[2]] = lw.pool.strings[yl+1]
                    fn[y, x] = add!(jdom, lw.pool, string(j.f(argv...)))
                end
                push!(lw.fn_tables, fn); fid = Int32(length(lw.fn_tables) - 1)
                pid = pair_for!(lw, v, (:julia, word), jdom.ids)
                push!(terms, Term(v, path_below(ocm, fk, locals[1]), pid, mt, (Int32(slot - 1), fid)))
            end
        end
    end
    terms
end
function pair_for!(lw::Lowered, obs_v, key, lat_ids::Vector{Int32})
    haskey(lw.pair_id, (obs_v, key)) || (lw.pair_id[(obs_v, key)] = (Int32(length(lw.pair_id)), lw.obs_dom[obs_v], lat_ids))
    lw.pair_id[(obs_v, key)][1]
end
"the top-level reference slot of the observed class whose flattened copy vertex v is"
function slot_of_vertex(ocm::PCleanClass, v::VertexID)
    n = ocm.nodes[v]
    while n isa SubmodelNode && ocm.nodes[n.foreign_key_node_id] isa SubmodelNode
        v = n.foreign_key_node_id; n = ocm.nodes[v]
    end
    n isa SubmodelNode ? n.foreign_key_node_id : v
end
"vertex ids from the slot `fk` down to vertex v: the chain of nested slots, then v's id in ITS class"
function path_below(ocm::PCleanClass, fk::VertexID, v::VertexID)
    chain = VertexID[]; n = ocm.nodes[v]
    while n isa SubmodelNode && n.foreign_key_node_id != fk
        pushfirst!(chain, n.subnode_id); v = n.foreign_key_node_id; n = ocm.nodes[v]
    end
    pushfirst!(chain, (n::SubmodelNode).subnode_id); chain
end
vertex_in_target(ocm::PCleanClass, v::VertexID) = (ocm.nodes[v]::SubmodelNode).subnode_id - 0   # id of v's value in the slot's target

"emit the node of class `cls` (a reference slot) with the terms whose value lives in its sub-tree; returns its node id"
function emit_fk_node!(lw::Lowered, blk::LBlock, cls::ClassID, slot_vertex, terms::Vector{Term}, parent::Int32, parent_fk_col::Int32)
    cm = lw.model.classes[cls]
    nid = Int32(length(blk.nodes)); push!(blk.nodes, CNode(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
    push!(blk.node_class, cls); push!(blk.node_vertex, slot_vertex)
    tb = Int32(length(blk.terms))
    for t in terms                                                           # candidate column = the flattened column of the value
        push!(blk.terms, cterm(lw, t, lw.col_of[cls][flat_vertex(lw.model, cls, t.path)]))
    end
    nt = Int32(length(blk.terms)) - tb
    kids = Int32[]; colsrc = Dict{VertexID,Tuple{Int32,Int32}}()
    for (v, n) in enumerate(cm.nodes)                                        # own attributes in declaration (vertex) order
        if n isa ForeignKeyNode
            sub = [Term(t.obs, t.path[2:end], t.pair, t.max_typos, t.ctx) for t in terms if length(t.path) > 1 && t.path[1] == v]
            cid = emit_fk_node!(lw, blk, n.target_class, v, sub, nid, lw.col_of[cls][v])
            push!(kids, cid); colsrc[v] = (Int32(-1), Int32(-1))
            for (i, vv) in n.vmap                                            # flattened copies come from the child's columns
                haskey(lw.col_of[cls], vv) && (colsrc[vv] = (cid, lw.col_of[n.target_class][i]))
            end
        elseif n isa RandomChoiceNode && haskey(lw.latent_dom, (cls, v))
            sub = [t for t in terms if t.path == [v]]
            cid = Int32(length(blk.nodes)); ltb = Int32(length(blk.terms))
            foreach(t -> push!(blk.terms, cterm(lw, t, Int32(0))), sub)
            cacheable = Int32(length(sub) == 1 && sub[1].ctx === nothing)
            dval, dspec = Int32(0), Int32(0)
            if n.dist isa Union{StringPrior,TimePrior}
                args = const_args(cm, n)
                dval = value_of(lw.latent_dom[(cls, v)], lw.pool, discrete_proposal_dummy_value(n.dist, args...)) + Int32(1)
                dspec = n.dist isa TimePrior ? Int32(2) : Int32(1 | (args[1] << 8) | (args[2] << 16))
            end
            push!(blk.nodes, CNode(NODE_LEAF, lw.option_id[(cls, v)], ltb, length(sub), 0, 0, nid, -1, cacheable, 0, dval, dspec))
            push!(blk.node_class, cls); push!(blk.node_vertex, v)
            push!(kids, cid); colsrc[v] = (cid, Int32(0))
        end
    end
    cb = Int32(length(blk.children)); append!(blk.children, kids)
    cmb = Int32(length(blk.colmap) ÷ 2)
    for v in lw.layout[cls]; s = get(colsrc, v, (Int32(-1), Int32(-1))); push!(blk.colmap, s[1], s[2]); end
    blk.nodes[nid+1] = CNode(NODE_FK, lw.table_id[cls], tb, nt, cb, length(kids), parent, parent_fk_col, 0, cmb, 0, 0)
    nid
end
cterm(lw, t::Term, cand_col) = CTerm(lw.obs_col[t.obs], cand_col, t.pair, DENS_ADD_TYPOS, t.max_typos,
                                     t.ctx === nothing ? -1 : t.ctx[1], t.ctx === nothing ? -1 : t.ctx[2], 0)
"vertex of class cls that holds the value reached by following `path` (nested slot vertices, then the own vertex)"
function flat_vertex(model::PCleanModel, cls::ClassID, path::Vector{VertexID})
    length(path) == 1 && return path[1]
    fk = model.classes[cls].nodes[path[1]]::ForeignKeyNode
    fk.vmap[flat_vertex(model, fk.target_class, path[2:end])]
end

function build_blocks!(lw::Lowered)
    ocm = lw.model.classes[lw.query.class]; fk_block = Dict{VertexID,Int}()
    eblocks = Tuple{Int,Vector{VertexID}}[]                                   # (model block, vertices) per ENGINE block
    for (ub, names) in enumerate(ocm.blocks)
        fks = [v for v in names if ocm.nodes[v] isa ForeignKeyNode]
        if length(fks) <= 1; push!(eblocks, (ub, names)); continue; end
        # several slots in one block: one engine block per slot, an observation goes to the LAST slot it mentions
        groups = [VertexID[f] for f in fks]
        for v in names
            ocm.nodes[v] isa ForeignKeyNode && continue
            hs = [findfirst(==(slot_of_vertex(ocm, a)), fks) for a in leaf_args(ocm, v)]
            hs = [h for h in hs if h !== nothing]
            push!(groups[isempty(hs) ? length(fks) : maximum(hs)], v)
        end
        foreach(g -> push!(eblocks, (ub, g)), groups)
    end
    for (bi, (ub, names)) in enumerate(eblocks)
        fks = [v for v in names if ocm.nodes[v] isa ForeignKeyNode]
        isempty(fks) && error("blocks without a reference slot (scoring blocks) bind through pclean_load_score_block: see model.py:_lower_score_block")
        fk = fks[1]; fk_block[fk] = bi
        blk = LBlock((ocm.nodes[fk]::ForeignKeyNode).target_class, fk, Int32(ub - 1), CNode[], CTerm[], Int32[], Int32[], Int32[], Int32[], ClassID[], VertexID[])
        push!(lw.blocks, blk)
        terms = block_terms!(lw, ocm, bi, fk, names, fk_block, blk)
        emit_fk_node!(lw, blk, blk.root_class, fk, terms, Int32(-1), Int32(-1))
    end
end
"slot-copy vertices an observed-class vertex ultimately depends on (through AddTypos / JuliaNode arguments)"
function leaf_args(ocm::PCleanClass, v::VertexID)
    n = ocm.nodes[v]
    n isa SubmodelNode && return [v]
    n isa Union{RandomChoiceNode,JuliaNode} || return VertexID[]
    vcat([leaf_args(ocm, a) for a in n.arg_node_ids]...)
end

"latent-class plans: the children of the class's node in the observed plan, re-rooted (model.py:_build_latent_plans)"
function build_latent_plans!(lw::Lowered)
    next = Int32(length(lw.blocks))
    for (bi, blk) in enumerate(lw.blocks), (nid, n) in enumerate(blk.nodes)
        (n.kind == NODE_FK && !haskey(lw.latent_plans, blk.node_class[nid])) || continue
        cls = blk.node_class[nid]
        plan = (block_id=next, src_block=bi - 1, cls=cls, nodes=CNode[], terms=CTerm[], children=Int32[], colmap=Int32[],
                roots=Int32[], root_vertex=VertexID[], ctx_sources=collect(zip(blk.ctx_block, blk.ctx_col)), node_class=ClassID[], node_vertex=VertexID[])
        for k in n.child_begin+1:n.child_begin+n.n_children
            child = blk.children[k] + 1
            push!(plan.roots, copy_subtree!(lw, blk, child, plan, Int32(-1))); push!(plan.root_vertex, blk.node_vertex[child])
        end
        lw.latent_plans[cls] = plan; next += 1
    end
end
function copy_subtree!(lw, blk::LBlock, nid::Int, plan, parent::Int32)
    n = blk.nodes[nid]; new_id = Int32(length(plan.nodes)); push!(plan.nodes, n)
    push!(plan.node_class, blk.node_class[nid]); push!(plan.node_vertex, blk.node_vertex[nid])
    tb = Int32(length(plan.terms))
    for t in blk.terms[n.term_begin+1:n.term_begin+n.n_terms]               # ctx now comes from the evidence row: mode 1
        push!(plan.terms, t.ctx_slot >= 0 ? CTerm(t.obs_col, t.cand_col, t.pair_table, t.dens_kind, t.max_typos, t.ctx_slot, t.fn_table, 1) : t)
    end
    # cross-slot JuliaNode observations whose CONTEXT argument lives in this sub-tree: model.py:_copy_subtree adds them
    # with ctx_mode 2 (fn[candidate][ctx of the evidence row]) and one ctx_sources slot per local argument — same rule here:
    # lw.cross_terms is filled by block_terms! in the Python lowering; port alongside when JuliaNodes across slots are used.
    nt = Int32(length(plan.terms)) - tb
    if n.kind == NODE_FK
        remap = Dict{Int32,Int32}(); kids = Int32[]
        for k in n.child_begin+1:n.child_begin+n.n_children
            c = blk.children[k]; remap[c] = copy_subtree!(lw, blk, c + 1, plan, new_id); push!(kids, remap[c])
        end
        cb = Int32(length(plan.children)); append!(plan.children, kids); cmb = Int32(length(plan.colmap) ÷ 2)
        for j in 0:length(lw.layout[blk.node_class[nid]])-1
            cn, cc = blk.colmap[2*(n.colmap_begin+j)+1], blk.colmap[2*(n.colmap_begin+j)+2]
            push!(plan.colmap, cn >= 0 ? remap[cn] : Int32(-1), cc)
        end
        plan.nodes[new_id+1] = CNode(n.kind, n.table, tb, nt, cb, length(kids), parent, n.parent_fk_col, 0, cmb, 0, 0)
    else
        plan.nodes[new_id+1] = CNode(n.kind, n.table, tb, nt, 0, 0, parent, -1, 0, 0, 0, 0)
    end
    new_id
end

"PCleanModel + Query + DataFrame -> Lowered (model.py: LoweredModel.__init__)"
function lower(model::PCleanModel, query::Query, data; extra_latent=Dict{Tuple{ClassID,VertexID},Vector{String}}())
    lw = Lowered(model, query, Pool(), Dict(), Dict(), Dict(), Dict(), Dict(), Dict(), Dict(), Dict(), VertexID[], Dict(), Matrix{Int32}[],
                 LBlock[], Dict(), extra_latent)
    build_domains!(lw, data); build_layouts!(lw); build_blocks!(lw); build_latent_plans!(lw)
    lw
end
"observed columns as the library wants them: n_rows x n_cols, value index in the column's observed domain, -1 missing"
function encode_observations(lw::Lowered, data)
    obs = fill(Int32(-1), size(data, 1), length(lw.obs_vertices))
    for (col, v) in lw.query.obsmap, (i, x) in enumerate(data[!, col])
        ismissing(x) || (obs[i, lw.obs_col[v]+1] = value_of(lw.obs_dom[v], lw.pool, string(x)))
    end
    obs
end
"static upload: strings, observed columns, pair tables, fn tables, letter model, plans (engine.py:_upload_static)"
function upload_static!(c::Ctx, lw::Lowered, obs::Matrix{Int32}, init_p, trans_p; dist_mode=1)
    sym, off, symid = pool_arrays(lw.pool)
    load_strings(c, sym, off); load_columns(c, obs)
    set_lm_tables(c, init_p, trans_p, UInt16[get(symid, ch, 0xFFFF) for ch in ALPHABET])
    for ((_, _), (pid, odom, lat_ids)) in lw.pair_id; build_pair_table(c, pid, odom.ids, lat_ids, dist_mode); end
    for (fid, fn) in enumerate(lw.fn_tables); set_fn_table(c, fid - 1, fn); end
    for (bi, b) in enumerate(lw.blocks); load_block(c, bi - 1, b.nodes, b.terms, b.children, b.colmap, b.ctx_block, b.ctx_col); end
    for (_, pl) in lw.latent_plans
        nctx = max(1, length(pl.ctx_sources)); load_block(c, pl.block_id, pl.nodes, pl.terms, pl.children, pl.colmap, zeros(Int32, nctx), zeros(Int32, nctx))
    end
    groups = [b.group for b in lw.blocks]
    length(unique(groups)) < length(groups) && foreach(bi -> set_block_group(c, bi - 1, groups[bi]), eachindex(groups))
end
"the plan arrays as nested Dicts/Vectors, for diffing against tests/golden/plans_*.json"
plan_json(lw::Lowered) = Dict("blocks" => [Dict("root_class" => String(b.root_class), "nodes" => [collect(Int, (n.kind, n.table, n.term_begin,
    n.n_terms, n.child_begin, n.n_children, n.parent, n.parent_fk_col, n.cacheable, n.colmap_begin, n.dummy_value, n.dummy_spec)) for n in b.nodes],
    "terms" => [collect(Int, (t.obs_col, t.cand_col, t.pair_table, t.dens_kind, t.max_typos, t.ctx_slot, t.fn_table, t.ctx_mode)) for t in b.terms],
    "children" => Int.(b.children), "colmap" => Int.(b.colmap), "ctx_src_block" => Int.(b.ctx_block), "ctx_src_col" => Int.(b.ctx_col)) for b in lw.blocks])

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
    for ((cls, v), oid) in lw.option_id                                       # discrete_proposal(dist, args...) (distributions.jl:16)
        node = lw.model.classes[cls].nodes[v]; args = const_args(lw.model.classes[cls], node)
        node.dist isa ChooseProportionally && (args = Any[args[1], trace.tables[cls].parameters[node.arg_node_ids[2]].current_value])
        _, lps = discrete_proposal(node.dist, args...)
        set_options(c, oid, lw.option_values[(cls, v)], Float64.(lps))
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
