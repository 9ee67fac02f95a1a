# MultiModalMuSigHIP.jl -- drop-in Julia front-end over libmmmusig_hip.so (include/mmmusig.h).
#
# Same exported names and call signatures as MultiModalMuSig.jl (src/MultiModalMuSig.jl:9):
#     IMMCTM, MMCTM, LDA, fit!, format_counts_lda, format_counts_ctm, format_counts_mmctm
# and the same struct field names; every hot-path function is one `ccall`.  Model state lives in HBM inside the
# library handle; after `fit!` the fields the reference exposes (ϕ, θ, γ, λ, μ, Σ, props, elbo, ll, converged, ...) are
# downloaded into ordinary Julia arrays of the reference's shapes.
#
# MUTATION IN PLACE.  Upstream the Julia arrays ARE the model: a caller may assign any field and then call any function --
# scripts/run_mmctm.jl:124-131 overwrites `model.γ[m]`, `model.Elnϕ[m]`, `model.ϕ[m]` and calls `fit!`.  Every entry point here
# therefore uploads, before its `ccall`, the fields the reference function READS BEFORE IT WRITES THEM (`FIT_READS` for `fit!`, every
# field for the stage functions) and downloads afterwards what it writes.  tests/test_julia_shim_cpu.py holds the read / write sets of
# every reference function and checks them against this file.
#
# The un-exported functions the reference's own test-suite drives the path through (test/lda.jl, ilda.jl, mmctm.jl, immctm.jl,
# common.jl: `MultiModalMuSig.update_ϕ!(model)`, `update_ζ!(model, d)`, `calculate_sumθ(model, d)`, `λ_objective(...)`,
# `calculate_modality_loglikelihood(...)`, `calculate_ElnPβ(model)`, ...) are all defined here under the same names and
# signatures -- the STAGE API at the end of each model's section and the FREE FUNCTIONS section.  Between stage calls the Julia
# arrays ARE the model, exactly as upstream: the tests assign fields in place (`model.Elnθ .= Elnθ`, `model.θ[1][1] = ...`,
# `model.ζ = ...`) and then call one function, so every stage call uploads the fields (`upload!`), runs ONE entry point of
# the library and copies back, in place, the fields the reference function writes.  tests/golden/reference_test_api.json lists
# every `MultiModalMuSig.<name>` the reference's tests use; tests/test_julia_shim_cpu.py checks each is defined here.
#
# STATUS: Julia is not installed in the build container or on the GPU test boxes, so this file has NOT been executed;
# it is a 1:1 mechanical mapping onto the C ABI (each ccall's argument list is the corresponding prototype of
# include/mmmusig.h), which itself is exercised end-to-end by the Python host mirror and the test-suite.
#
# Random initialisation is drawn here with `rand(1:100, ...)` in the same order as the reference constructors
# (LDA.jl:36; MMCTM.jl:60-63; IMMCTM.jl:59-65), so `Random.seed!` behaves as upstream.
module MultiModalMuSigHIP

using DataFrames
using Random

export IMMCTM, MMCTM, ILDA, LDA, fit!, format_counts_lda, format_counts_ctm, format_counts_mmctm

const LIB = get(ENV, "MMM_LIB_PATH", joinpath(@__DIR__, "..", "lib", "libmmmusig_hip.so"))

# ---- context ------------------------------------------------------------------------------------------------------
mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer=0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:mmm_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, out)
        rc == 0 || error("mmm_ctx_create: " * unsafe_string(ccall((:mmm_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
        ctx = new(out[])
        finalizer(c -> ccall((:mmm_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), c.h), ctx)
        return ctx
    end
end

# mmm_tuning_opts (include/mmmusig.h): the caller's choices for the handles created on a context from now on -- E-step build
# (0 auto, 1 sparse, 2 dense, 3 wide), a pinned launch geometry (`geometry_cus`: same bits on any gfx950 device), the side stream, ...
struct TuningOpts
    lda_build::Cint; ctm_build::Cint; geometry_cus::Cint; grid_blocks::Cint; waves_per_block::Cint; moment_blocks::Cint
    side_stream::Cint; resident_cap::Cint; disable::Cuint; solve_lanes::Cint; solve_waves::Cint; reserved::NTuple{5,Cint}
end
function set_tuning!(ctx::Context; lda_build=0, ctm_build=0, geometry_cus=0, grid_blocks=0, waves_per_block=0, moment_blocks=0, side_stream=0,
                     resident_cap=0, disable=0, solve_lanes=0, solve_waves=0)
    t = Ref(TuningOpts(lda_build, ctm_build, geometry_cus, grid_blocks, waves_per_block, moment_blocks, side_stream, resident_cap, disable, solve_lanes, solve_waves,
                       ntuple(i -> Cint(0), 5)))
    rc = ccall((:mmm_ctx_set_tuning, LIB), Cint, (Ptr{Cvoid}, Ref{TuningOpts}), ctx.h, t)
    rc == 0 || error("mmm_ctx_set_tuning: " * unsafe_string(ccall((:mmm_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx.h)))
    return ctx
end

const DEFAULT_CTX = Ref{Union{Nothing,Context}}(nothing)
default_context() = (DEFAULT_CTX[] === nothing && (DEFAULT_CTX[] = Context(0)); DEFAULT_CTX[])

function check(rc::Cint, ctx::Context, what::String)
    rc == 0 || error(what * ": " * unsafe_string(ccall((:mmm_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx.h)))
    nothing
end

# ---- count formatting: same results as src/utils.jl:1-36 (a document = W x 2 matrix [1-based term id, count > 0]) ----------
# sparse view of one count column: rows (term, count) for the terms that occur
function make_count_matrix(counts)
    terms = [t for t in eachindex(counts) if counts[t] > 0]
    return Int[terms counts[terms]]
end
column(df::DataFrame, c::Symbol) = collect(df[!, c])
format_counts_lda(df::DataFrame, cols::Vector{Symbol}) = Matrix{Int}[make_count_matrix(column(df, c)) for c in cols]
format_counts_mmctm(dfs::Vector{DataFrame}, cols::Vector{Symbol}) =
    Vector{Matrix{Int}}[Matrix{Int}[make_count_matrix(column(df, c)) for df in dfs] for c in cols]
format_counts_ctm(df::DataFrame, cols::Vector{Symbol}) = format_counts_mmctm([df], cols)

# CSR packing expected by the ABI: 0-based Int32 terms, Int32 counts, Int64 offsets
function pack_lda(X::Vector{Matrix{Int}})
    D = length(X)
    doc_ptr = zeros(Int64, D + 1)
    for d in 1:D doc_ptr[d + 1] = doc_ptr[d] + size(X[d], 1) end
    term = Vector{Int32}(undef, doc_ptr[end]); count = Vector{Int32}(undef, doc_ptr[end])
    for d in 1:D, w in 1:size(X[d], 1)
        term[doc_ptr[d] + w] = X[d][w, 1] - 1
        count[doc_ptr[d] + w] = X[d][w, 2]
    end
    return doc_ptr, term, count
end

function pack_mm(X::Vector{Vector{Matrix{Int}}}, M::Int)
    D = length(X)
    doc_ptr = zeros(Int64, M * (D + 1)); term = Int32[]; count = Int32[]
    base = 0
    for m in 1:M
        dp, t, c = pack_lda(Matrix{Int}[X[d][m] for d in 1:D])
        doc_ptr[(m - 1) * (D + 1) + 1:m * (D + 1)] = dp .+ base
        base += dp[end]; append!(term, t); append!(count, c)
    end
    return doc_ptr, term, count
end

# ---- LDA (struct fields as LDA.jl:1-22) ---------------------------------------------------------------------------------
mutable struct LDA
    K::Int; D::Int; N::Vector{Int}; V::Int
    η::Float64; λ::Matrix{Float64}; β::Matrix{Float64}; Elnβ::Matrix{Float64}
    α::Float64; γ::Matrix{Float64}; θ::Matrix{Float64}; Elnθ::Matrix{Float64}
    ϕ::Vector{Matrix{Float64}}
    X::Vector{Matrix{Int}}
    converged::Bool; elbo::Float64; ll::Float64
    ctx::Context; h::Ptr{Cvoid}; doc_ptr::Vector{Int64}

    function LDA(k::Int, α::Float64, η::Float64, V::Int, X::Vector{Matrix{Int}}; ctx::Context=default_context())
        model = new()
        model.K = k; model.α = α; model.η = η; model.X = X; model.D = length(X); model.V = V
        model.N = [sum(X[d][:, 2]) for d in 1:model.D]
        model.λ = rand(1:100, V, k)                                   # LDA.jl:36
        doc_ptr, term, count = pack_lda(X)
        model.doc_ptr = doc_ptr; model.ctx = ctx
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mmm_lda_create, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Cint, Cint, Cdouble, Cdouble, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
                    ctx.h, model.D, V, k, α, η, doc_ptr, term, count, model.λ, out), ctx, "mmm_lda_create")
        model.h = out[]
        finalizer(m -> ccall((:mmm_lda_destroy, LIB), Cint, (Ptr{Cvoid},), m.h), model)
        model.converged = false
        download!(model)
        return model
    end
end

function LDA(k::Int, α::Float64, η::Float64, X::Vector{Matrix{Int}}; kw...)     # LDA.jl:57-67
    V = 0
    for d in 1:length(X)
        size(X[d], 1) > 0 && (V = max(V, maximum(X[d][:, 1])))
    end
    return LDA(k, α, η, V, X; kw...)
end

function lda_get(model, field::Int, n::Int)
    buf = Vector{Float64}(undef, n)
    check(ccall((:mmm_lda_get, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Csize_t), model.h, field, buf, n), model.ctx, "mmm_lda_get")
    return buf
end

# field ids of mmm_lda_get / mmm_lda_set (include/mmmusig.h): 0 λ, 1 Elnβ, 2 β, 3 γ, 4 Elnθ, 5 θ, 6 ϕ; ILDA factors 7 λ, 8 Elnβ, 9 β
const LDA_FIELDS = (λ=0, Elnβ=1, β=2, γ=3, Elnθ=4, θ=5, ϕ=6)

# ϕ[d] (K x W_d) of every document out of / into the flat K x nnz buffer of the library
function unflatten_ϕ!(model, flat::Vector{Float64})
    K = model.K
    ϕ = Vector{Matrix{Float64}}(undef, model.D)
    for d in 1:model.D
        ϕ[d] = reshape(flat[K * model.doc_ptr[d] + 1:K * model.doc_ptr[d + 1]], K, :)
    end
    model.ϕ = ϕ
end
flatten_ϕ(model) = reduce(vcat, [vec(Matrix{Float64}(p)) for p in model.ϕ]; init=Float64[])

# the reference's functions mutate the model's arrays in place (`.=`): keep the array when the shape allows, replace it otherwise
function assign!(model, f::Symbol, new::Array{Float64})
    if isdefined(model, f) && size(getfield(model, f)) == size(new)
        copyto!(getfield(model, f), new)
    else
        setfield!(model, f, new)
    end
    return model
end

# one field device -> Julia array
function download_field!(model::LDA, f::Symbol)
    V, K, D = model.V, model.K, model.D
    if f == :ϕ
        unflatten_ϕ!(model, lda_get(model, 6, K * model.doc_ptr[end]))
    elseif f in (:λ, :Elnβ, :β)
        assign!(model, f, reshape(lda_get(model, LDA_FIELDS[f], V * K), V, K))
    else
        assign!(model, f, reshape(lda_get(model, LDA_FIELDS[f], K * D), K, D))
    end
    return model
end

function download!(model::LDA)
    for f in (:λ, :Elnβ, :β, :γ, :Elnθ, :θ, :ϕ) download_field!(model, f) end
    return model
end

# one field Julia array -> device.  :α / :η are the hyper-parameters (plain mutable fields upstream: `model.α = 0.5; fit!(model)`)
function upload_field!(model::LDA, f::Symbol)
    if f == :ϕ
        lda_set(model, 6, flatten_ϕ(model))
    elseif f == :α || f == :η
        lda_set_hyper(model, model.α, Float64[model.η])
    else
        lda_set(model, LDA_FIELDS[f], vec(Matrix{Float64}(getfield(model, f))))
    end
    return model
end

const LDA_ALL_FIELDS = (:α, :η, :λ, :Elnβ, :β, :γ, :Elnθ, :θ, :ϕ)
# Julia arrays -> device, every field (see the header comment: the arrays are the model between stage calls)
function upload!(model::LDA)
    for f in LDA_ALL_FIELDS upload_field!(model, f) end
    return model
end

# What `fit!` reads before it writes it.  LDA.jl:198-224 / ILDA.jl:246-272: update_γ! reads ϕ and α (LDA.jl:83-87), update_ϕ! reads Elnβ
# (:72; its Elnθ has just been written by update_γ!), update_λ! reads η (:101).  γ, Elnθ, θ, λ, β are written before anything reads them.
# MMCTM.jl:457-494 / IMMCTM.jl:437-466: fitdoc! reads λ, ν (update_ζ! :172-181, and the start points of both LD_MMA solves), Elnϕ (update_θ!
# :190), μ and invΣ (the objectives, common.jl:11-36); update_γ! reads α (:226); with updateΣ = false Σ and invΣ are never written and the
# ELBO reads both; γ and ϕ are written (update_γ!, update_ϕ!) before they are read but belong to the topics a caller seeds together with
# Elnϕ (run_mmctm.jl:126-128), so they go up with it.  ζ, θ, props are written before they are read: not uploaded.
const FIT_READS = (LDA = (:α, :η, :Elnβ, :ϕ), ILDA = (:α, :η, :Elnβ, :ϕ),
                   MMCTM = (:α, :μ, :Σ, :invΣ, :γ, :Elnϕ, :ϕ, :λ, :ν), IMMCTM = (:α, :μ, :Σ, :invΣ, :γ, :Elnϕ, :λ, :ν))
function upload_fit_reads!(model)
    for f in FIT_READS[nameof(typeof(model))] upload_field!(model, f) end
    return model
end

# fit!(model; maxiter, tol, verbose) -- LDA.jl:198-224.  `resident = true` (not upstream) skips the upload: the caller states that no array
# of the model has been assigned since the last call returned (a 10k x 96-term ϕ is 77 MB).
function fit!(model::LDA; maxiter=1000, tol=1e-4, verbose=true, resident=false)
    resident || upload_fit_reads!(model)
    ll = Vector{Float64}(undef, maxiter); n = Ref{Cint}(0); cv = Ref{Cint}(0); elbo = Ref{Cdouble}(0.0)
    check(ccall((:mmm_lda_fit, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}, Ref{Cdouble}),
                model.h, maxiter, tol, ll, n, cv, elbo), model.ctx, "mmm_lda_fit")
    resize!(ll, n[])
    if verbose
        for (iter, v) in enumerate(ll) println("$iter\tLog-likelihood: ", v) end
    end
    model.converged = cv[] != 0; model.elbo = elbo[]; model.ll = ll[end]
    download!(model)
    return ll
end

function lda_set_hyper(model, α::Float64, η::Vector{Float64})
    check(ccall((:mmm_lda_set_hyper, LIB), Cint, (Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Cint), model.h, α, η, length(η)), model.ctx, "mmm_lda_set_hyper")
end

function lda_set(model, field::Int, v::Vector{Float64})
    check(ccall((:mmm_lda_set, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Csize_t), model.h, field, v, length(v)), model.ctx, "mmm_lda_set")
end

# ---- LDA / ILDA stage API: update_ϕ!, update_γ! (+ update_Elnθ!), update_λ! (+ update_Elnβ!), update_β!, update_θ! -- LDA.jl:69-112,
# ILDA.jl:65-130 -- the calls of test/lda.jl:38-103 and test/ilda.jl:52-158.  `written`: the fields the reference function writes.
function lda_stage!(model, rc_of_call::Function, what::String, written)
    upload!(model)
    check(rc_of_call(), model.ctx, what)
    for f in written download_field!(model, f) end
    return nothing
end

# ---- LDA frozen-topic inference (LDA.jl:226-295) ------------------------------------------------------------------------
function lda_infer!(model, unsmoothed::Bool, maxiter::Int, tol::Float64, verbose::Bool)
    ll = Vector{Float64}(undef, maxiter); n = Ref{Cint}(0); cv = Ref{Cint}(0)
    check(ccall((:mmm_lda_infer, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cdouble, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}),
                model.h, unsmoothed ? 1 : 0, maxiter, tol, ll, n, cv), model.ctx, "mmm_lda_infer")
    resize!(ll, n[])
    if verbose
        for (iter, v) in enumerate(ll) println("$iter\tLog-likelihood: ", v) end
    end
    model.converged = cv[] != 0
    return ll
end

function lda_elbo(model)
    e = Ref{Cdouble}(0.0)
    check(ccall((:mmm_lda_elbo, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ptr{Cdouble}), model.h, e, C_NULL), model.ctx, "mmm_lda_elbo")
    return e[]
end

# transform(model, X) -- LDA.jl:233-263: θ (K x D) of new documents under the trained β
function transform(model::LDA, X::Vector{Matrix{Int}}; maxiter=1000, tol=1e-4, verbose=false)
    newmodel = LDA(model.K, model.α, model.η, model.V, X; ctx=model.ctx)
    lda_set(newmodel, 2, vec(model.β))                                    # :237
    lda_infer!(newmodel, true, maxiter, Float64(tol), verbose)
    newmodel.converged || @warn "transform did not converge"              # :258-260 (`warn` upstream)
    download!(newmodel)
    return newmodel.θ
end

# fit_heldout(Xheldout, model) -- LDA.jl:265-295
function fit_heldout(Xheldout::Vector{Matrix{Int}}, model::LDA; maxiter=100, verbose=false)
    heldout_model = LDA(model.K, model.α, model.η, model.V, Xheldout; ctx=model.ctx)
    lda_set(heldout_model, 0, vec(model.λ)); lda_set(heldout_model, 2, vec(model.β)); lda_set(heldout_model, 1, vec(model.Elnβ))   # :269-271
    ll = lda_infer!(heldout_model, false, maxiter, 1e-4, verbose)
    heldout_model.elbo = lda_elbo(heldout_model)                          # :291
    heldout_model.ll = ll[end]
    download!(heldout_model)
    return heldout_model
end

# ---- ILDA (struct fields as ILDA.jl:1-23; same C handle type as LDA) ---------------------------------------------------------
mutable struct ILDA
    K::Int; D::Int; I::Int; J::Vector{Int}
    η::Vector{Float64}; λ::Vector{Matrix{Float64}}; β::Vector{Matrix{Float64}}; Elnβ::Vector{Matrix{Float64}}
    α::Float64; γ::Matrix{Float64}; θ::Matrix{Float64}; Elnθ::Matrix{Float64}
    ϕ::Vector{Matrix{Float64}}
    features::Matrix{Int}; X::Vector{Matrix{Int}}
    converged::Bool; elbo::Float64; ll::Float64
    ctx::Context; h::Ptr{Cvoid}; doc_ptr::Vector{Int64}

    function ILDA(k::Int, α::Float64, η::Vector{Float64}, features::Matrix{Int}, X::Vector{Matrix{Int}}; ctx::Context=default_context())
        model = new()
        model.K = k; model.α = α; model.η = copy(η); model.X = X; model.D = length(X)
        model.I = size(features, 2); model.J = vec(maximum(features, dims=1)); model.features = features
        V = size(features, 1)
        model.λ = [Float64.(rand(1:100, model.J[i], k)) for i in 1:model.I]               # ILDA.jl:36
        lambda0 = vcat([vec(model.λ[i]) for i in 1:model.I]...)                           # J_i x K column-major, feature after feature
        featflat = Int32.(vec(features .- 1))                                             # [i*V + v], 0-based values
        doc_ptr, term, count = pack_lda(X)
        model.doc_ptr = doc_ptr; model.ctx = ctx
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mmm_ilda_create, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Cint, Cint, Cdouble, Cint, Ptr{Cint}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32},
                     Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
                    ctx.h, model.D, V, k, α, model.I, Cint.(model.J), model.η, featflat, doc_ptr, term, count, lambda0, out), ctx, "mmm_ilda_create")
        model.h = out[]
        finalizer(m -> ccall((:mmm_lda_destroy, LIB), Cint, (Ptr{Cvoid},), m.h), model)
        model.converged = false
        download!(model)
        return model
    end
end

ILDA(k::Int, α::Float64, η::Float64, features::Matrix{Int}, X::Vector{Matrix{Int}}; kw...) =      # ILDA.jl:58-63
    ILDA(k, α, fill(η, size(features, 2)), features, X; kw...)

const ILDA_FACTOR_FIELDS = (λ=7, Elnβ=8, β=9)

function download_field!(model::ILDA, f::Symbol)
    K, D = model.K, model.D
    if f == :ϕ
        unflatten_ϕ!(model, lda_get(model, 6, K * model.doc_ptr[end]))
    elseif f in (:λ, :Elnβ, :β)                     # J_i x K column-major, feature after feature
        off = cumsum([0; model.J .* K])
        flat = lda_get(model, ILDA_FACTOR_FIELDS[f], off[end])
        setfield!(model, f, Matrix{Float64}[reshape(flat[off[i] + 1:off[i + 1]], model.J[i], K) for i in 1:model.I])
    else
        assign!(model, f, reshape(lda_get(model, LDA_FIELDS[f], K * D), K, D))
    end
    return model
end

function download!(model::ILDA)
    for f in (:λ, :Elnβ, :β, :γ, :Elnθ, :θ, :ϕ) download_field!(model, f) end
    return model
end

packed_factors(fs) = reduce(vcat, [vec(Matrix{Float64}(f)) for f in fs]; init=Float64[])

function upload_field!(model::ILDA, f::Symbol)
    if f == :ϕ
        lda_set(model, 6, flatten_ϕ(model))
    elseif f == :α || f == :η
        lda_set_hyper(model, model.α, Vector{Float64}(model.η))
    elseif f in (:λ, :Elnβ, :β)      # the factor arrays; the library derives its effective V x K tables from the uploaded Elnβ[i] / β[i]
        lda_set(model, ILDA_FACTOR_FIELDS[f], packed_factors(getfield(model, f)))
    else
        lda_set(model, LDA_FIELDS[f], vec(Matrix{Float64}(getfield(model, f))))
    end
    return model
end

function upload!(model::ILDA)
    for f in LDA_ALL_FIELDS upload_field!(model, f) end
    return model
end

# stage functions of LDA and ILDA (same C handle type, same entry points)
const TopicModel = Union{LDA,ILDA}
update_ϕ!(model::TopicModel) = lda_stage!(model, () -> ccall((:mmm_lda_update_phi, LIB), Cint, (Ptr{Cvoid},), model.h), "update_ϕ!", (:ϕ,))
update_Elnθ!(model::TopicModel) = lda_stage!(model, () -> ccall((:mmm_lda_update_Elntheta, LIB), Cint, (Ptr{Cvoid},), model.h), "update_Elnθ!", (:Elnθ,))
update_γ!(model::TopicModel) = lda_stage!(model, () -> ccall((:mmm_lda_update_gamma, LIB), Cint, (Ptr{Cvoid},), model.h), "update_γ!", (:γ, :Elnθ))
update_θ!(model::TopicModel) = lda_stage!(model, () -> ccall((:mmm_lda_update_theta, LIB), Cint, (Ptr{Cvoid},), model.h), "update_θ!", (:θ,))
update_Elnβ!(model::TopicModel) = lda_stage!(model, () -> ccall((:mmm_lda_update_Elnbeta, LIB), Cint, (Ptr{Cvoid},), model.h), "update_Elnβ!", (:Elnβ,))
update_λ!(model::TopicModel) = lda_stage!(model, () -> ccall((:mmm_lda_update_lambda, LIB), Cint, (Ptr{Cvoid},), model.h), "update_λ!", (:λ, :Elnβ))
update_β!(model::TopicModel) = lda_stage!(model, () -> ccall((:mmm_lda_update_beta, LIB), Cint, (Ptr{Cvoid},), model.h), "update_β!", (:β,))

# calculate_elbo and its seven terms (LDA.jl:114-172; ILDA.jl:132-201): one launch sequence, terms[7] =
# (ElnPβ, ElnPθ, ElnPZ, ElnPX, ElnQβ, ElnQθ, ElnQZ); elbo = the first four minus the last three
function elbo_terms(model::TopicModel)
    upload!(model)
    e = Ref{Cdouble}(0.0); t = Vector{Float64}(undef, 7)
    check(ccall((:mmm_lda_elbo, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ptr{Cdouble}), model.h, e, t), model.ctx, "mmm_lda_elbo")
    return e[], t
end
calculate_elbo(model::TopicModel) = elbo_terms(model)[1]
calculate_ElnPβ(model::TopicModel) = elbo_terms(model)[2][1]
calculate_ElnPθ(model::TopicModel) = elbo_terms(model)[2][2]
calculate_ElnPZ(model::TopicModel) = elbo_terms(model)[2][3]
calculate_ElnPX(model::TopicModel) = elbo_terms(model)[2][4]
calculate_ElnQβ(model::TopicModel) = elbo_terms(model)[2][5]
calculate_ElnQθ(model::TopicModel) = elbo_terms(model)[2][6]
calculate_ElnQZ(model::TopicModel) = elbo_terms(model)[2][7]

# calculate_loglikelihood -- LDA.jl:174-196 (X, θ, β) and ILDA.jl:203-239 (X, features, θ, β): free functions over the caller's arrays
function calculate_loglikelihood(X::Vector{Matrix{Int}}, θ::Matrix{Float64}, β::Matrix{Float64}; ctx::Context=default_context())
    doc_ptr, term, count = pack_lda(X)
    ll = Ref{Cdouble}(0.0)
    K, V = size(θ, 1), size(β, 1)
    check(ccall((:mmm_mixture_loglik, LIB), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
                ctx.h, length(X), K, V, doc_ptr, term, count, vec(θ), vec(β), ll), ctx, "mmm_mixture_loglik")
    return ll[]
end
calculate_loglikelihood(X::Vector{Matrix{Int}}, model::LDA) = calculate_loglikelihood(X, model.θ, model.β; ctx=model.ctx)
calculate_loglikelihood(model::LDA) = calculate_loglikelihood(model.X, model.θ, model.β; ctx=model.ctx)

function calculate_loglikelihood(X::Vector{Matrix{Int}}, features::Matrix{Int}, θ::Matrix{Float64}, β::Vector{Matrix{Float64}};
                                 ctx::Context=default_context())
    doc_ptr, term, count = pack_lda(X)
    ll = Ref{Cdouble}(0.0)
    K, V, I = size(θ, 1), size(features, 1), size(features, 2)
    J = Cint[size(β[i], 1) for i in 1:I]
    ϕflat = reduce(vcat, [reduce(vcat, [β[i][:, k] for i in 1:I]) for k in 1:K])      # [k][i][j] = β[i][j, k]
    check(ccall((:mmm_mixture_loglik_features, LIB), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Cint}, Ptr{Int32}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ref{Cdouble}),
                ctx.h, length(X), K, V, I, J, Int32.(vec(features .- 1)), doc_ptr, term, count, vec(θ), 0, ϕflat, ll), ctx, "mmm_mixture_loglik_features")
    return ll[]
end
calculate_loglikelihood(X::Vector{Matrix{Int}}, model::ILDA) = calculate_loglikelihood(X, model.features, model.θ, model.β; ctx=model.ctx)
calculate_loglikelihood(model::ILDA) = calculate_loglikelihood(model.X, model.features, model.θ, model.β; ctx=model.ctx)

# fit!(model; maxiter, tol, verbose) -- ILDA.jl:246-272
function fit!(model::ILDA; maxiter=1000, tol=1e-4, verbose=true, resident=false)
    resident || upload_fit_reads!(model)
    ll = Vector{Float64}(undef, maxiter); n = Ref{Cint}(0); cv = Ref{Cint}(0); elbo = Ref{Cdouble}(0.0)
    check(ccall((:mmm_lda_fit, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}, Ref{Cdouble}),
                model.h, maxiter, tol, ll, n, cv, elbo), model.ctx, "mmm_lda_fit")
    resize!(ll, n[])
    if verbose
        for (iter, v) in enumerate(ll) println("$iter\tLog-likelihood: ", v) end
    end
    model.converged = cv[] != 0; model.elbo = elbo[]; model.ll = ll[end]
    download!(model)
    return ll
end

# fit_heldout(Xheldout, model) -- ILDA.jl:320-353.  (`transform(::ILDA)` is a MethodError upstream, ILDA.jl:293: not provided.)
function fit_heldout(Xheldout::Vector{Matrix{Int}}, model::ILDA; maxiter=100, verbose=false)
    heldout_model = ILDA(model.K, model.α, model.η, model.features, Xheldout; ctx=model.ctx)
    packed(fs) = vcat([vec(f) for f in fs]...)
    lda_set(heldout_model, 7, packed(model.λ)); lda_set(heldout_model, 9, packed(model.β)); lda_set(heldout_model, 8, packed(model.Elnβ))
    ll = lda_infer!(heldout_model, false, maxiter, 1e-4, verbose)
    heldout_model.elbo = lda_elbo(heldout_model)
    heldout_model.ll = ll[end]
    download!(heldout_model)
    return heldout_model
end

# ---- MMCTM / IMMCTM (struct fields as MMCTM.jl:1-27 / IMMCTM.jl:1-27) -----------------------------------------------------
struct SolverOpts
    xtol_rel::Cdouble; xtol_abs::Cdouble; nu_lower::Cdouble; xtol_rule::Cint; max_eval::Cint
end

mutable struct MMCTM
    K::Vector{Int}; D::Int; N::Vector{Vector{Int}}; M::Int; V::Vector{Int}
    μ::Vector{Float64}; Σ::Matrix{Float64}; invΣ::Matrix{Float64}
    props::Vector{Vector{Vector{Float64}}}; α::Vector{Float64}; ϕ::Vector{Vector{Vector{Float64}}}
    ζ::Vector{Vector{Float64}}; θ::Vector{Vector{Matrix{Float64}}}; λ::Vector{Vector{Float64}}; ν::Vector{Vector{Float64}}
    γ::Vector{Vector{Vector{Float64}}}; Elnϕ::Vector{Vector{Vector{Float64}}}
    X::Vector{Vector{Matrix{Int}}}
    converged::Bool; elbo::Float64; ll::Vector{Float64}
    ctx::Context; h::Ptr{Cvoid}; doc_ptr::Vector{Int64}

    function MMCTM(k::Vector{Int}, α::Vector{Float64}, V::Vector{Int}, X::Vector{Vector{Matrix{Int}}};
                   init=:random, ctx::Context=default_context())
        init == :random || error("init must be either :random or :document")      # MMCTM.jl:76
        model = new()
        model.K = copy(k); model.α = copy(α); model.X = X; model.D = length(X); model.M = length(k); model.V = copy(V)
        model.N = [[sum(X[d][m][:, 2]) for m in 1:model.M] for d in 1:model.D]
        model.γ = [[Float64.(rand(1:100, model.V[m])) for kk in 1:model.K[m]] for m in 1:model.M]   # MMCTM.jl:60-63
        gamma0 = vcat([vcat(model.γ[m]...) for m in 1:model.M]...)
        doc_ptr, term, count = pack_mm(X, model.M)
        model.doc_ptr = doc_ptr; model.ctx = ctx
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mmm_ctm_create, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32},
                     Ptr{Cint}, Ptr{Cint}, Ptr{Int32}, Ptr{Cdouble}, Ptr{SolverOpts}, Ref{Ptr{Cvoid}}),
                    ctx.h, model.D, model.M, Cint.(k), Cint.(V), α, doc_ptr, term, count, C_NULL, C_NULL, C_NULL, gamma0, C_NULL, out),
              ctx, "mmm_ctm_create")
        model.h = out[]
        finalizer(m -> ccall((:mmm_ctm_destroy, LIB), Cint, (Ptr{Cvoid},), m.h), model)
        model.converged = false
        download!(model)
        return model
    end
end

function MMCTM(k::Vector{Int}, α::Vector{Float64}, X::Vector{Vector{Matrix{Int}}}; kw...)    # MMCTM.jl:94-108
    M = length(k); V = zeros(Int, M)
    for d in 1:length(X), m in 1:M
        size(X[d][m], 1) > 0 && (V[m] = max(V[m], maximum(X[d][m][:, 1])))
    end
    return MMCTM(k, α, V, X; kw...)
end

function ctm_get(model, field::Int, n::Int)
    buf = Vector{Float64}(undef, n)
    check(ccall((:mmm_ctm_get, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Csize_t), model.h, field, buf, n), model.ctx, "mmm_ctm_get")
    return buf
end

# μ, Σ, invΣ and the per-document λ, ν, ζ, θ: same flat layouts for MMCTM and IMMCTM handles
function download_docs!(model)
    M, D, K = model.M, model.D, model.K
    MK = sum(K)
    model.μ = ctm_get(model, 0, MK); model.Σ = reshape(ctm_get(model, 1, MK * MK), MK, MK); model.invΣ = reshape(ctm_get(model, 2, MK * MK), MK, MK)
    lam = reshape(ctm_get(model, 6, D * MK), MK, D); nu = reshape(ctm_get(model, 7, D * MK), MK, D)
    model.λ = [lam[:, d] for d in 1:D]; model.ν = [nu[:, d] for d in 1:D]
    z = reshape(ctm_get(model, 8, D * M), M, D); model.ζ = [z[:, d] for d in 1:D]
    dp = model.doc_ptr
    estart = [dp[(m - 1) * (D + 1) + 1] for m in 1:M]
    nnz = [dp[m * (D + 1)] - estart[m] for m in 1:M]
    toff = cumsum([0; nnz .* K])
    th = ctm_get(model, 10, toff[end])
    model.θ = [[reshape(th[toff[m] + (dp[(m - 1) * (D + 1) + d] - estart[m]) * K[m] + 1:toff[m] + (dp[(m - 1) * (D + 1) + d + 1] - estart[m]) * K[m]], K[m], :)
                for m in 1:M] for d in 1:D]
    return model
end

function download!(model::MMCTM)
    M, D, K, V = model.M, model.D, model.K, model.V
    MK = sum(K); koff = cumsum([0; K]); goff = cumsum([0; K .* V])
    download_docs!(model)
    nest(flat) = [[flat[goff[m] + (kk - 1) * V[m] + 1:goff[m] + kk * V[m]] for kk in 1:K[m]] for m in 1:M]
    model.γ = nest(ctm_get(model, 3, goff[end])); model.Elnϕ = nest(ctm_get(model, 4, goff[end])); model.ϕ = nest(ctm_get(model, 5, goff[end]))
    pr = reshape(ctm_get(model, 9, D * MK), MK, D)
    model.props = [[pr[koff[m] + 1:koff[m + 1], d] for m in 1:M] for d in 1:D]
    return model
end

# fit!(model; maxiter, tol, verbose, autoα, updateΣ) -- MMCTM.jl:457-494
function fit!(model::MMCTM; maxiter=100, tol=1e-4, verbose=true, autoα=false, updateΣ=true, resident=false)
    resident || upload_fit_reads!(model)      # scripts/run_mmctm.jl:124-131 assigns γ[m] / Elnϕ[m] / ϕ[m] and then calls fit!
    M = model.M
    ll = Vector{Float64}(undef, maxiter * M); n = Ref{Cint}(0); cv = Ref{Cint}(0); elbo = Ref{Cdouble}(0.0)
    check(ccall((:mmm_ctm_fit, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}, Ref{Cdouble}),
                model.h, maxiter, tol, (updateΣ ? 1 : 0) | (autoα ? 2 : 0),      # MMM_FIT_UPDATE_SIGMA | MMM_FIT_AUTO_ALPHA
                ll, n, cv, elbo), model.ctx, "mmm_ctm_fit")
    hist = [ll[(i - 1) * M + 1:i * M] for i in 1:n[]]
    if verbose
        for (iter, v) in enumerate(hist) println("$iter\tLog-likelihoods: ", join(v, ", ")) end
    end
    model.converged = cv[] != 0; model.elbo = elbo[]; model.ll = hist[end]
    autoα && (model.α = ctm_get(model, 11, M))
    download!(model)
    return hist
end

# ---- restart batch: what scripts/run_mmctm.jl:97-109 gets from `pmap(fit_restart, seeds)`, in one handle -----------
# γ0s[r][m][k] = rand(1:100, V[m]) of restart r (MMCTM.jl:60-63).  Returns (handle, ll[r][iter][m], converged[r], elbo[r]);
# `select_restart!(model, h, r)` then downloads restart r into `model`.
function fit_restarts(k::Vector{Int}, α::Vector{Float64}, V::Vector{Int}, X::Vector{Vector{Matrix{Int}}}, γ0s; maxiter=1000, tol=1e-4,
                      ctx::Context=default_context())
    R = length(γ0s); M = length(k); D = length(X)
    doc_ptr, term, count = pack_mm(X, M)
    g0 = Float64[]
    for r in 1:R, m in 1:M, kk in 1:k[m] append!(g0, Float64.(γ0s[r][m][kk])) end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mmm_ctm_create_batch, LIB), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Cint}, Ptr{Cint}, Ptr{Int32},
                 Ptr{Cdouble}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                ctx.h, R, D, M, Cint.(k), Cint.(V), α, doc_ptr, term, count, C_NULL, C_NULL, C_NULL, g0, C_NULL, h), ctx, "mmm_ctm_create_batch")
    ll = Vector{Float64}(undef, R * maxiter * M); n = Vector{Cint}(undef, R); cv = Vector{Cint}(undef, R); elbo = Vector{Cdouble}(undef, R)
    check(ccall((:mmm_ctm_fit_batch, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}),
                h[], maxiter, tol, 1, ll, n, cv, elbo), ctx, "mmm_ctm_fit_batch")
    hist = [[ll[((r - 1) * maxiter + (i - 1)) * M + 1:((r - 1) * maxiter + i) * M] for i in 1:n[r]] for r in 1:R]
    return h[], hist, cv .!= 0, elbo
end

function select_restart!(model::MMCTM, h::Ptr{Cvoid}, r::Int)
    check(ccall((:mmm_ctm_select, LIB), Cint, (Ptr{Cvoid}, Cint), h, r - 1), model.ctx, "mmm_ctm_select")
    old = model.h; model.h = h
    download!(model)
    model.h = old
    return model
end

# ---- the restart driver of scripts/run_mmctm.jl:77-182 under its own names -----------------------------------------------------------
# `fit_model(counts, K, α, V, restarts, verbose, seed, progress)` is what the script's `main` calls (:270).  Stage 1 (`fit_seed_models`,
# :97-109: one `fit_restart` per seed under `pmap`) is ONE restart batch here; the γ₀ of restart i is what `MMCTM(K, α, V, counts)` draws
# after `Random.seed!(seeds[i])` (:78-81), so a seed means what it means upstream.  `progress` is accepted and ignored (no worker processes).
draw_γ0(K::Vector{Int}, V::Vector{Int}) = [[Float64.(rand(1:100, V[m])) for kk in 1:K[m]] for m in 1:length(K)]      # MMCTM.jl:60-63

function fit_restart(seed, K, α, V, counts; ctx::Context=default_context())                       # run_mmctm.jl:77-84
    Random.seed!(seed)
    model = MMCTM(K, α, V, counts; ctx=ctx)
    fit!(model, maxiter=1000, tol=1e-4, verbose=false, resident=true)
    return model
end

# run_mmctm.jl:86-97 on the [restart, modality] matrix of final log-likelihoods: the best restart of every modality
pick_optimal_modality_restarts(ll::Matrix{Float64}) = vec([x[1] for x in findmax(ll; dims=1)[2]])
pick_optimal_modality_models(models) = models[pick_optimal_modality_restarts(permutedims(reduce(hcat, [m.ll for m in models])))]

function fit_seed_models(counts, K, α, V, seeds; progress=false, ctx::Context=default_context())      # run_mmctm.jl:99-111
    γ0s = map(seeds) do seed
        Random.seed!(seed)
        draw_γ0(K, V)
    end
    h, hist, converged, elbo = fit_restarts(K, α, V, counts, γ0s; maxiter=1000, tol=1e-4, ctx=ctx)
    ll = permutedims(reduce(hcat, [hist[r][end] for r in 1:length(seeds)]))                         # [restart, modality]
    opt = pick_optimal_modality_restarts(ll)
    opt_models = map(1:length(K)) do m
        Random.seed!(seeds[opt[m]])
        model = MMCTM(K, α, V, counts; ctx=ctx)
        select_restart!(model, h, opt[m])
        model.ll = hist[opt[m]][end]; model.converged = converged[opt[m]]; model.elbo = elbo[opt[m]]
        model
    end
    check(ccall((:mmm_ctm_destroy, LIB), Cint, (Ptr{Cvoid},), h), ctx, "mmm_ctm_destroy")
    return opt_models
end

function seed_and_fit_restart(seed, opt_models; verbose=false, ctx::Context=opt_models[1].ctx)      # run_mmctm.jl:113-134
    Random.seed!(seed)
    K = opt_models[1].K; α = opt_models[1].α; V = opt_models[1].V; counts = opt_models[1].X
    model = MMCTM(K, α, V, counts; ctx=ctx)
    for m in 1:length(K)
        model.γ[m] = deepcopy(opt_models[m].γ[m])
        model.Elnϕ[m] = deepcopy(opt_models[m].Elnϕ[m])
        model.ϕ[m] = deepcopy(opt_models[m].ϕ[m])
    end
    fit!(model, maxiter=1000, tol=1e-5, verbose=verbose)      # uploads γ, Elnϕ, ϕ (FIT_READS) before the first pass
    return model
end

# StatsBase.denserank
function denserank(x)
    u = sort(unique(x))
    return [searchsortedfirst(u, v) for v in x]
end

function pick_optimal_model(models)                                                                 # run_mmctm.jl:136-147
    ll = permutedims(reduce(hcat, [m.ll for m in models]))
    ranks = reduce(hcat, [Float64.(denserank(abs.(ll[:, i]))) for i in 1:size(ll, 2)])
    return models[findmin(vec(sum(ranks, dims=2)) ./ size(ll, 2))[2]]
end

# run_mmctm.jl:149-161 fits one seeded model per seed and ranks them.  Every one of those fits starts from the same γ / Elnϕ / ϕ -- the seed
# only feeds the constructor's random γ, which is overwritten, and `NLopt.srand`, which the deterministic LD_MMA never reads -- so they are the
# same fit: it is run once.
seed_and_fit_model(opt_models, seeds; progress=false) = pick_optimal_model([seed_and_fit_restart(seeds[1], opt_models)])

function fit_model(counts, K, α, V, restarts, verbose, seed, progress; ctx::Context=default_context())   # run_mmctm.jl:163-182
    Random.seed!(seed)
    seeds = rand(1:typemax(Int), restarts)
    seed_models = fit_seed_models(counts, K, α, V, seeds; progress=progress, ctx=ctx)
    if verbose
        println("Modality optimal model log-likelihoods:")
        for m in 1:length(K) println(m, ": ", seed_models[m].ll) end
    end
    model = seed_and_fit_model(seed_models, seeds; progress=progress)
    if verbose
        println("Seeded model log-likelihoods:")
        println(model.ll)
    end
    return model
end

# ---- frozen-topic inference (MMCTM.jl:496-586): fresh model on X, copied globals, passes on the GPU --------------------
function ctm_set(model, field::Int, v::Vector{Float64})
    check(ccall((:mmm_ctm_set, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Csize_t), model.h, field, v, length(v)), model.ctx, "mmm_ctm_set")
end

function ctm_infer!(model, flags::Int, maxiter::Int, tol::Float64)
    M = model.M
    ll = Vector{Float64}(undef, maxiter * M); n = Ref{Cint}(0); cv = Ref{Cint}(0)
    check(ccall((:mmm_ctm_infer, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cdouble, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}),
                model.h, flags, maxiter, tol, ll, n, cv), model.ctx, "mmm_ctm_infer")
    model.converged = cv[] != 0
    return [ll[(i - 1) * M + 1:i * M] for i in 1:n[]]
end

flat(nested) = vcat((vcat(x...) for x in nested)...)

function transform(model::MMCTM, X::Vector{Vector{Matrix{Int}}}; maxiter=1000, tol=1e4, fit_gaussian=false, verbose=false)   # MMCTM.jl:511-552
    newmodel = MMCTM(model.K, model.α, model.V, X; ctx=model.ctx)
    ctm_set(newmodel, 5, flat(model.ϕ))                                   # :516
    if !fit_gaussian
        ctm_set(newmodel, 0, model.μ); ctm_set(newmodel, 1, vec(model.Σ))   # :518-521 (invΣ stays I, as upstream)
    end
    ll = ctm_infer!(newmodel, 1 | (fit_gaussian ? 2 : 0), maxiter, Float64(tol))
    newmodel.ll = ll[end]
    download!(newmodel)
    return newmodel
end

function fit_heldout(Xheldout::Vector{Vector{Matrix{Int}}}, model::MMCTM; maxiter=100, verbose=false)                     # MMCTM.jl:554-586
    heldout_model = MMCTM(model.K, model.α, model.V, Xheldout; ctx=model.ctx)
    ctm_set(heldout_model, 0, model.μ); ctm_set(heldout_model, 1, vec(model.Σ)); ctm_set(heldout_model, 2, vec(model.invΣ))
    ctm_set(heldout_model, 3, flat(model.γ)); ctm_set(heldout_model, 4, flat(model.Elnϕ)); ctm_set(heldout_model, 5, flat(model.ϕ))
    ctm_infer!(heldout_model, 0, maxiter, 1e-4)
    download!(heldout_model)
    return heldout_model
end

# IMMCTM(k, α, features, X) -- IMMCTM.jl:29-88: same handle type on the C side (mmm_ctm_create with n_feat/J/features);
# the field download mirrors MMCTM with γ/Elnϕ nested one level deeper ([m][k][i][j]) and no props/ϕ fields.
mutable struct IMMCTM
    K::Vector{Int}; D::Int; N::Vector{Vector{Int}}; M::Int; I::Vector{Int}; J::Vector{Vector{Int}}; V::Vector{Int}
    μ::Vector{Float64}; Σ::Matrix{Float64}; invΣ::Matrix{Float64}; α::Vector{Vector{Float64}}
    ζ::Vector{Vector{Float64}}; θ::Vector{Vector{Matrix{Float64}}}; λ::Vector{Vector{Float64}}; ν::Vector{Vector{Float64}}
    γ::Vector{Vector{Vector{Vector{Float64}}}}; Elnϕ::Vector{Vector{Vector{Vector{Float64}}}}
    features::Vector{Matrix{Int}}; X::Vector{Vector{Matrix{Int}}}
    converged::Bool; elbo::Float64; ll::Vector{Float64}
    ctx::Context; h::Ptr{Cvoid}; doc_ptr::Vector{Int64}

    function IMMCTM(k::Vector{Int}, α::Vector{Vector{Float64}}, features::Vector{Matrix{Int}}, X::Vector{Vector{Matrix{Int}}};
                    ctx::Context=default_context())
        model = new()
        model.K = copy(k); model.α = deepcopy(α); model.features = deepcopy(features); model.X = X
        model.D = length(X); model.M = length(features)
        model.I = [size(features[m])[2] for m in 1:model.M]
        model.J = [vec(maximum(features[m], dims=1)) for m in 1:model.M]
        model.V = [size(features[m])[1] for m in 1:model.M]
        model.N = [[sum(X[d][m][:, 2]) for m in 1:model.M] for d in 1:model.D]
        model.γ = [[[Float64.(rand(1:100, model.J[m][i])) for i in 1:model.I[m]] for kk in 1:model.K[m]] for m in 1:model.M]  # IMMCTM.jl:59-65
        gamma0 = vcat([vcat([vcat(model.γ[m][kk]...) for kk in 1:model.K[m]]...) for m in 1:model.M]...)
        featflat = Int32.(vcat([vec(features[m] .- 1) for m in 1:model.M]...))       # [foff[m] + (i-1)*V + v], 0-based values
        doc_ptr, term, count = pack_mm(X, model.M)
        model.doc_ptr = doc_ptr; model.ctx = ctx
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mmm_ctm_create, LIB), Cint,
                    (Ptr{Cvoid}, Cint, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32},
                     Ptr{Cint}, Ptr{Cint}, Ptr{Int32}, Ptr{Cdouble}, Ptr{SolverOpts}, Ref{Ptr{Cvoid}}),
                    ctx.h, model.D, model.M, Cint.(k), Cint.(model.V), vcat(α...), doc_ptr, term, count,
                    Cint.(model.I), Cint.(vcat(model.J...)), featflat, gamma0, C_NULL, out), ctx, "mmm_ctm_create")
        model.h = out[]
        finalizer(m -> ccall((:mmm_ctm_destroy, LIB), Cint, (Ptr{Cvoid},), m.h), model)
        model.converged = false
        download!(model)
        return model
    end
end

function IMMCTM(k::Vector{Int}, α::Vector{Float64}, features::Vector{Matrix{Int}}, X::Vector{Vector{Matrix{Int}}}; kw...)   # IMMCTM.jl:81-88
    I = [size(features[m])[2] for m in 1:length(features)]
    return IMMCTM(k, Vector{Float64}[fill(α[m], I[m]) for m in 1:length(features)], features, X; kw...)
end

function fit!(model::IMMCTM; maxiter=100, tol=1e-4, verbose=true, autoα=false, resident=false)    # IMMCTM.jl:437-466
    resident || upload_fit_reads!(model)
    M = model.M
    ll = Vector{Float64}(undef, maxiter * M); n = Ref{Cint}(0); cv = Ref{Cint}(0); elbo = Ref{Cdouble}(0.0)
    check(ccall((:mmm_ctm_fit, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}, Ref{Cdouble}),
                model.h, maxiter, tol, 1 | (autoα ? 2 : 0), ll, n, cv, elbo), model.ctx, "mmm_ctm_fit")
    hist = [ll[(i - 1) * M + 1:i * M] for i in 1:n[]]
    if verbose
        for (iter, v) in enumerate(hist) println("$iter\tLog-likelihoods: ", join(v, ", ")) end
    end
    model.converged = cv[] != 0; model.elbo = elbo[]; model.ll = hist[end]
    if autoα
        a = ctm_get(model, 11, sum(model.I)); off = cumsum([0; model.I])
        model.α = [a[off[m] + 1:off[m + 1]] for m in 1:M]
    end
    download!(model)
    return hist
end

function download!(model::IMMCTM)
    M = model.M
    download_docs!(model)
    SJ = [sum(model.J[m]) for m in 1:M]; mgoff = cumsum([0; model.K .* SJ])
    gflat = ctm_get(model, 3, mgoff[end]); eflat = ctm_get(model, 4, mgoff[end])
    nest(flat) = [[[flat[mgoff[m] + (kk - 1) * SJ[m] + sum(model.J[m][1:i - 1]) + 1:mgoff[m] + (kk - 1) * SJ[m] + sum(model.J[m][1:i])]
                    for i in 1:model.I[m]] for kk in 1:model.K[m]] for m in 1:M]
    model.γ = nest(gflat); model.Elnϕ = nest(eflat)
    return model
end

flat3(nested) = vcat((flat(x) for x in nested)...)         # [m][k][i][j] -> flat, the library's layout

# fit_heldout(Xheldout, model) -- IMMCTM.jl:468-497
function fit_heldout(Xheldout::Vector{Vector{Matrix{Int}}}, model::IMMCTM; maxiter=100, verbose=false)
    heldout_model = IMMCTM(model.K, model.α, model.features, Xheldout; ctx=model.ctx)
    ctm_set(heldout_model, 0, model.μ); ctm_set(heldout_model, 1, vec(model.Σ)); ctm_set(heldout_model, 2, vec(model.invΣ))
    ctm_set(heldout_model, 3, flat3(model.γ)); ctm_set(heldout_model, 4, flat3(model.Elnϕ))
    hist = ctm_infer!(heldout_model, 0, maxiter, 1e-4)
    isempty(hist) || (heldout_model.ll = hist[end])
    download!(heldout_model)
    return heldout_model
end

# predict_modality_η(Xobs, m, model) -- MMCTM.jl:588-634 / IMMCTM.jl:499-545: E[η of the unobserved modality m | the other
# modalities' counts].  The frozen-topic passes run on the GPU; the Gaussian conditioning is MK x MK host algebra.
function predict_modality_η(Xobs::Vector{Vector{Matrix{Int}}}, m::Int, model::Union{MMCTM,IMMCTM}; maxiter=100, verbose=false)
    obsM = setdiff(1:model.M, m)
    koff = cumsum([0; model.K])
    unobs = collect(koff[m] + 1:koff[m + 1]); obs = setdiff(1:koff[end], unobs)
    if model isa IMMCTM
        obsmodel = IMMCTM(model.K[obsM], model.α[obsM], model.features[obsM], Xobs; ctx=model.ctx)
        ctm_set(obsmodel, 3, flat3(model.γ[obsM])); ctm_set(obsmodel, 4, flat3(model.Elnϕ[obsM]))
    else
        obsmodel = MMCTM(model.K[obsM], model.α[obsM], model.V[obsM], Xobs; ctx=model.ctx)
        ctm_set(obsmodel, 3, flat(model.γ[obsM])); ctm_set(obsmodel, 4, flat(model.Elnϕ[obsM])); ctm_set(obsmodel, 5, flat(model.ϕ[obsM]))
    end
    ctm_set(obsmodel, 0, model.μ[obs]); ctm_set(obsmodel, 1, vec(model.Σ[obs, obs])); ctm_set(obsmodel, 2, vec(model.invΣ[obs, obs]))   # :597-599
    ctm_infer!(obsmodel, 0, maxiter, 1e-4)
    obsmodel.converged || @warn "model not converged."                    # :623-625 (`warn` upstream)
    download_docs!(obsmodel)
    A = model.Σ[unobs, obs] * model.invΣ[obs, obs]                         # :627-633
    return [model.μ[unobs] .+ A * (obsmodel.λ[d] .- model.μ[obs]) for d in 1:obsmodel.D]
end


# ---- MMCTM / IMMCTM stage API -- MMCTM.jl:110-269, IMMCTM.jl:90-244: the calls of test/mmctm.jl:59-293 and test/immctm.jl:80-294 -----
# field ids of mmm_ctm_get / mmm_ctm_set: 0 μ, 1 Σ, 2 invΣ, 3 γ, 4 Elnϕ, 5 ϕ, 6 λ, 7 ν, 8 ζ, 9 props, 10 θ, 11 α
const CTM = Union{MMCTM,IMMCTM}

flat_docs(xs) = reduce(vcat, [Vector{Float64}(x) for x in xs]; init=Float64[])           # [d][i] -> [i + n (d-1)]
# θ[d][m] (K_m x W_dm) -> the library's modality-major buffer: all documents of modality 1, then modality 2, ...
flat_θ(model) = reduce(vcat, [reduce(vcat, [vec(Matrix{Float64}(model.θ[d][m])) for d in 1:model.D]; init=Float64[]) for m in 1:model.M]; init=Float64[])
topic_flat(model::MMCTM, nested) = Vector{Float64}(flat(nested))
topic_flat(model::IMMCTM, nested) = Vector{Float64}(flat3(nested))

# one field Julia array -> device
function upload_field!(model::CTM, f::Symbol)
    if f == :μ
        ctm_set(model, 0, Vector{Float64}(model.μ))
    elseif f == :Σ
        ctm_set(model, 1, vec(Matrix{Float64}(model.Σ)))
    elseif f == :invΣ
        ctm_set(model, 2, vec(Matrix{Float64}(model.invΣ)))
    elseif f == :γ
        ctm_set(model, 3, topic_flat(model, model.γ))
    elseif f == :Elnϕ
        ctm_set(model, 4, topic_flat(model, model.Elnϕ))
    elseif f == :ϕ
        ctm_set(model, 5, topic_flat(model, model.ϕ))
    elseif f == :λ
        ctm_set(model, 6, flat_docs(model.λ))
    elseif f == :ν
        ctm_set(model, 7, flat_docs(model.ν))
    elseif f == :ζ
        ctm_set(model, 8, flat_docs(model.ζ))
    elseif f == :props
        ctm_set(model, 9, reduce(vcat, [reduce(vcat, model.props[d]) for d in 1:model.D]; init=Float64[]))
    elseif f == :θ
        ctm_set(model, 10, flat_θ(model))
    elseif f == :α
        ctm_set(model, 11, model isa MMCTM ? Vector{Float64}(model.α) : Vector{Float64}(reduce(vcat, model.α)))
    else
        error("unknown field $f")
    end
    return model
end

const MMCTM_ALL_FIELDS = (:α, :μ, :Σ, :invΣ, :γ, :Elnϕ, :ϕ, :λ, :ν, :ζ, :props, :θ)
const IMMCTM_ALL_FIELDS = (:α, :μ, :Σ, :invΣ, :γ, :Elnϕ, :λ, :ν, :ζ, :θ)          # no ϕ / props fields (IMMCTM.jl:1-27)
# Julia arrays -> device, every field (the arrays are the model between stage calls)
function upload!(model::CTM)
    for f in (model isa MMCTM ? MMCTM_ALL_FIELDS : IMMCTM_ALL_FIELDS) upload_field!(model, f) end
    return model
end

# device -> Julia arrays, one field of the reference struct
function download_field!(model::CTM, f::Symbol)
    M, D, K = model.M, model.D, model.K
    MK = sum(K); koff = cumsum([0; K])
    if f == :μ
        model.μ = ctm_get(model, 0, MK)
    elseif f == :Σ
        model.Σ = reshape(ctm_get(model, 1, MK * MK), MK, MK)
    elseif f == :invΣ
        model.invΣ = reshape(ctm_get(model, 2, MK * MK), MK, MK)
    elseif f == :λ
        lam = reshape(ctm_get(model, 6, D * MK), MK, D); model.λ = [lam[:, d] for d in 1:D]
    elseif f == :ν
        nu = reshape(ctm_get(model, 7, D * MK), MK, D); model.ν = [nu[:, d] for d in 1:D]
    elseif f == :ζ
        z = reshape(ctm_get(model, 8, D * M), M, D); model.ζ = [z[:, d] for d in 1:D]
    elseif f == :props
        pr = reshape(ctm_get(model, 9, D * MK), MK, D)
        model.props = [[pr[koff[m] + 1:koff[m + 1], d] for m in 1:M] for d in 1:D]
    elseif f == :α
        a = ctm_get(model, 11, model isa MMCTM ? M : sum(model.I))
        if model isa MMCTM
            model.α = a
        else
            off = cumsum([0; model.I]); model.α = [a[off[m] + 1:off[m + 1]] for m in 1:M]
        end
    elseif f == :θ
        dp = model.doc_ptr
        estart = [dp[(m - 1) * (D + 1) + 1] for m in 1:M]
        nnz = [dp[m * (D + 1)] - estart[m] for m in 1:M]
        toff = cumsum([0; nnz .* K])
        th = ctm_get(model, 10, toff[end])
        model.θ = [[reshape(th[toff[m] + (dp[(m - 1) * (D + 1) + d] - estart[m]) * K[m] + 1:toff[m] + (dp[(m - 1) * (D + 1) + d + 1] - estart[m]) * K[m]], K[m], :)
                    for m in 1:M] for d in 1:D]
    elseif f in (:γ, :Elnϕ, :ϕ)
        id = f == :γ ? 3 : (f == :Elnϕ ? 4 : 5)
        if model isa MMCTM
            V = model.V; goff = cumsum([0; K .* V])
            fl = ctm_get(model, id, goff[end])
            setfield!(model, f, [[fl[goff[m] + (kk - 1) * V[m] + 1:goff[m] + kk * V[m]] for kk in 1:K[m]] for m in 1:M])
        else
            SJ = [sum(model.J[m]) for m in 1:M]; mgoff = cumsum([0; K .* SJ])
            fl = ctm_get(model, id, mgoff[end])
            setfield!(model, f, [[[fl[mgoff[m] + (kk - 1) * SJ[m] + sum(model.J[m][1:i - 1]) + 1:mgoff[m] + (kk - 1) * SJ[m] + sum(model.J[m][1:i])]
                                   for i in 1:model.I[m]] for kk in 1:K[m]] for m in 1:M])
        end
    else
        error("unknown field $f")
    end
    return model
end

# one document of a per-document field: the other documents keep their Julia arrays (update_ζ!(model, d) etc. touch index d only)
function download_doc!(model::CTM, f::Symbol, d::Int)
    old = getfield(model, f)
    download_field!(model, f)
    new = getfield(model, f)
    for dd in 1:model.D
        dd == d || (new[dd] = old[dd])
    end
    return model
end

function ctm_stage!(model::CTM, rc_of_call::Function, what::String, written)
    upload!(model)
    check(rc_of_call(), model.ctx, what)
    for f in written download_field!(model, f) end
    return nothing
end

function ctm_doc_stage!(model::CTM, stage::Int, d::Int, what::String, f::Symbol)
    upload!(model)
    check(ccall((:mmm_ctm_update_doc, LIB), Cint, (Ptr{Cvoid}, Cint, Cint), model.h, stage, d - 1), model.ctx, what)
    download_doc!(model, f, d)
    return nothing
end

# per-document functions (MMM_STAGE_ZETA = 0, _THETA = 1, _NU = 2, _LAMBDA = 3)
update_ζ!(model::CTM, d::Int) = ctm_doc_stage!(model, 0, d, "update_ζ!", :ζ)        # MMCTM.jl:172-181
update_θ!(model::CTM, d::Int) = ctm_doc_stage!(model, 1, d, "update_θ!", :θ)        # MMCTM.jl:183-198 / IMMCTM.jl:152-172
update_ν!(model::CTM, d::Int) = ctm_doc_stage!(model, 2, d, "update_ν!", :ν)        # MMCTM.jl:156-170 (LD_MMA on the device)
update_λ!(model::CTM, d::Int) = ctm_doc_stage!(model, 3, d, "update_λ!", :λ)        # MMCTM.jl:127-143 (LD_MMA on the device)
function fitdoc!(model::CTM, d::Int)                                                   # MMCTM.jl:450-455
    update_ζ!(model, d); update_θ!(model, d); update_ν!(model, d); update_λ!(model, d)
end

# M-step functions
update_μ!(model::CTM) = ctm_stage!(model, () -> ccall((:mmm_ctm_update_mu, LIB), Cint, (Ptr{Cvoid},), model.h), "update_μ!", (:μ,))                   # MMCTM.jl:200-202
update_Σ!(model::CTM) = ctm_stage!(model, () -> ccall((:mmm_ctm_update_Sigma, LIB), Cint, (Ptr{Cvoid},), model.h), "update_Σ!", (:Σ, :invΣ))         # MMCTM.jl:204-212
update_Elnϕ!(model::CTM) = ctm_stage!(model, () -> ccall((:mmm_ctm_update_Elnphi, LIB), Cint, (Ptr{Cvoid},), model.h), "update_Elnϕ!", (:Elnϕ,))     # MMCTM.jl:214-222
update_γ!(model::CTM) = ctm_stage!(model, () -> ccall((:mmm_ctm_update_gamma, LIB), Cint, (Ptr{Cvoid},), model.h), "update_γ!", (:γ, :Elnϕ))         # MMCTM.jl:224-242
update_α!(model::CTM) = ctm_stage!(model, () -> ccall((:mmm_ctm_update_alpha, LIB), Cint, (Ptr{Cvoid},), model.h), "update_α!", (:α,))               # MMCTM.jl:252-269
update_ϕ!(model::MMCTM) = ctm_stage!(model, () -> ccall((:mmm_ctm_update_phi, LIB), Cint, (Ptr{Cvoid},), model.h), "update_ϕ!", (:ϕ,))               # MMCTM.jl:244-250
update_props!(model::MMCTM) = ctm_stage!(model, () -> ccall((:mmm_ctm_update_props, LIB), Cint, (Ptr{Cvoid},), model.h), "update_props!", (:props,)) # MMCTM.jl:145-154

# calculate_sumθ(model, d), calculate_Ndivζ(model, d) -- MMCTM.jl:110-125
function calculate_sumθ(model::CTM, d::Int)
    upload!(model)
    out = Vector{Float64}(undef, sum(model.K))
    check(ccall((:mmm_ctm_doc_sums, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}), model.h, d - 1, out, C_NULL), model.ctx, "mmm_ctm_doc_sums")
    return out
end
function calculate_Ndivζ(model::CTM, d::Int)
    upload!(model)
    out = Vector{Float64}(undef, sum(model.K))
    check(ccall((:mmm_ctm_doc_sums, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}), model.h, d - 1, C_NULL, out), model.ctx, "mmm_ctm_doc_sums")
    return out
end

# calculate_elbo and its seven terms (MMCTM.jl:271-382; IMMCTM.jl:247-360): terms[7] = (ElnPϕ, ElnPη, ElnPZ, ElnPX, ElnQϕ, ElnQη, ElnQZ)
function elbo_terms(model::CTM)
    upload!(model)
    e = Ref{Cdouble}(0.0); t = Vector{Float64}(undef, 7)
    check(ccall((:mmm_ctm_elbo, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ptr{Cdouble}), model.h, e, t), model.ctx, "mmm_ctm_elbo")
    return e[], t
end
calculate_elbo(model::CTM) = elbo_terms(model)[1]
calculate_ElnPϕ(model::CTM) = elbo_terms(model)[2][1]
calculate_ElnPη(model::CTM) = elbo_terms(model)[2][2]
calculate_ElnPZ(model::CTM) = elbo_terms(model)[2][3]
calculate_ElnPX(model::CTM) = elbo_terms(model)[2][4]
calculate_ElnQϕ(model::CTM) = elbo_terms(model)[2][5]
calculate_ElnQη(model::CTM) = elbo_terms(model)[2][6]
calculate_ElnQZ(model::CTM) = elbo_terms(model)[2][7]

# Non-fatal events, counted where upstream silently carries on (NLopt's return code is dropped, MMCTM.jl:141,168; nothing checks for NaN):
# (solves of the last E-step that hit the evaluation cap, solves of the last E-step that met a non-finite objective value, non-finite
# values in the log-likelihood history) -- mmm_ctm_events / mmm_lda_events.  Read-only: nothing is uploaded.
function events(model::CTM)
    out = zeros(Int64, 4)
    check(ccall((:mmm_ctm_events, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), model.h, out), model.ctx, "mmm_ctm_events")
    return (n_capped=out[1], n_nonfinite=out[2], n_nonfinite_ll=out[3])
end
function events(model::TopicModel)
    out = zeros(Int64, 4)
    check(ccall((:mmm_lda_events, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), model.h, out), model.ctx, "mmm_lda_events")
    return (n_capped=out[1], n_nonfinite=out[2], n_nonfinite_ll=out[3])
end

# ---- FREE FUNCTIONS of src/common.jl and the log-likelihood helpers: caller arrays in, one `ccall` each ---------------------------
# λ_objective(λ, ∇λ, ν, Ndivζ, sumθ, μ, invΣ) -- common.jl:11-23 (∇λ is filled in place when it has elements, as upstream)
function λ_objective(λ::Vector{Float64}, ∇λ::Vector{Float64}, ν::Vector{Float64}, Ndivζ::Vector{Float64}, sumθ::Vector{Float64},
                     μ::Vector{Float64}, invΣ::Matrix{Float64}; ctx::Context=default_context())
    val = Ref{Cdouble}(0.0)
    check(ccall((:mmm_lambda_objective, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}),
                ctx.h, length(λ), λ, ν, Ndivζ, sumθ, μ, vec(invΣ), val, length(∇λ) > 0 ? pointer(∇λ) : Ptr{Cdouble}(C_NULL)), ctx, "mmm_lambda_objective")
    return val[]
end

# ν_objective(ν, ∇ν, λ, Ndivζ, μ, invΣ) -- common.jl:25-36
function ν_objective(ν::Vector{Float64}, ∇ν::Vector{Float64}, λ::Vector{Float64}, Ndivζ::Vector{Float64}, μ::Vector{Float64},
                     invΣ::Matrix{Float64}; ctx::Context=default_context())
    val = Ref{Cdouble}(0.0)
    check(ccall((:mmm_nu_objective, LIB), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}),
                ctx.h, length(ν), ν, λ, Ndivζ, μ, vec(invΣ), val, length(∇ν) > 0 ? pointer(∇ν) : Ptr{Cdouble}(C_NULL)), ctx, "mmm_nu_objective")
    return val[]
end

# α_objective(α, ∇α, sum_Elnϕ, K, V) -- common.jl:38-46
function α_objective(α::Vector{Float64}, ∇α::Vector{Float64}, sum_Elnϕ::Float64, K::Int, V::Int; ctx::Context=default_context())
    val = Ref{Cdouble}(0.0); g = Ref{Cdouble}(0.0)
    check(ccall((:mmm_alpha_objective, LIB), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cint, Cint, Ref{Cdouble}, Ref{Cdouble}),
                ctx.h, α[1], sum_Elnϕ, K, V, val, g), ctx, "mmm_alpha_objective")
    length(∇α) > 0 && (∇α[1] = g[])
    return val[]
end

# calculate_modality_loglikelihood(X, props, ϕ) -- MMCTM.jl:402-418; the per-document form (:384-400) is the D = 1 case
function calculate_modality_loglikelihood(X::Vector{Matrix{Int}}, props::Vector{Vector{Float64}}, ϕ::Vector{Vector{Float64}};
                                          ctx::Context=default_context())
    doc_ptr, term, count = pack_lda(X)
    ll = Ref{Cdouble}(0.0)
    K, V = length(ϕ), length(ϕ[1])
    check(ccall((:mmm_mixture_loglik, LIB), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
                ctx.h, length(X), K, V, doc_ptr, term, count, reduce(vcat, props; init=Float64[]), reduce(vcat, ϕ), ll), ctx, "mmm_mixture_loglik")
    return ll[]
end
calculate_docmodality_loglikelihood(X::Matrix{Int}, props::Vector{Float64}, ϕ::Vector{Vector{Float64}}; kw...) =
    calculate_modality_loglikelihood(Matrix{Int}[X], Vector{Float64}[props], ϕ; kw...)

function calculate_loglikelihoods(X::Vector{Vector{Matrix{Int}}}, props::Vector{Vector{Vector{Float64}}}, ϕ::Vector{Vector{Vector{Float64}}}; kw...)   # MMCTM.jl:420-440
    D = length(X); M = length(ϕ)
    return [calculate_modality_loglikelihood(Matrix{Int}[X[d][m] for d in 1:D], Vector{Float64}[props[d][m] for d in 1:D], ϕ[m]; kw...) for m in 1:M]
end
calculate_loglikelihoods(X::Vector{Vector{Matrix{Int}}}, model::MMCTM) = calculate_loglikelihoods(X, model.props, model.ϕ; ctx=model.ctx)     # MMCTM.jl:442-444
calculate_loglikelihoods(model::MMCTM) = calculate_loglikelihoods(model.X, model.props, model.ϕ; ctx=model.ctx)                                  # MMCTM.jl:446-448

# calculate_modality_loglikelihood(X, η, ϕ, features) -- IMMCTM.jl:387-407 (ϕ[k][i]: J_i probabilities; props = softmax(η[d]) on the device)
function calculate_modality_loglikelihood(X::Vector{Matrix{Int}}, η::Vector{Vector{Float64}}, ϕ::Vector{Vector{Vector{Float64}}},
                                          features::Matrix{Int}; ctx::Context=default_context())
    doc_ptr, term, count = pack_lda(X)
    ll = Ref{Cdouble}(0.0)
    K, V, I = length(ϕ), size(features, 1), size(features, 2)
    J = Cint[length(ϕ[1][i]) for i in 1:I]
    check(ccall((:mmm_mixture_loglik_features, LIB), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Cint}, Ptr{Int32}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ref{Cdouble}),
                ctx.h, length(X), K, V, I, J, Int32.(vec(features .- 1)), doc_ptr, term, count, reduce(vcat, η; init=Float64[]), 1, flat(ϕ), ll),
          ctx, "mmm_mixture_loglik_features")
    return ll[]
end
calculate_docmodality_loglikelihood(X::Matrix{Int}, η::Vector{Float64}, ϕ::Vector{Vector{Vector{Float64}}}, features::Matrix{Int}; kw...) =
    calculate_modality_loglikelihood(Matrix{Int}[X], Vector{Float64}[η], ϕ, features; kw...)                                                  # IMMCTM.jl:362-385

function calculate_loglikelihoods(X::Vector{Vector{Matrix{Int}}}, model::IMMCTM)                                                              # IMMCTM.jl:408-428
    koff = cumsum([0; model.K])
    return [calculate_modality_loglikelihood(Matrix{Int}[X[d][m] for d in 1:model.D], Vector{Float64}[model.λ[d][koff[m] + 1:koff[m + 1]] for d in 1:model.D],
                                             [[model.γ[m][k][i] ./ sum(model.γ[m][k][i]) for i in 1:model.I[m]] for k in 1:model.K[m]], model.features[m];
                                             ctx=model.ctx) for m in 1:model.M]
end
calculate_loglikelihoods(model::IMMCTM) = calculate_loglikelihoods(model.X, model)

end # module
