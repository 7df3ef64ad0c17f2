# UDECoreMI355.jl -- the reference-side binding of libudecore.so (include/udecore.h).
#
# NOT EXECUTED IN THE BUILD IMAGE (no Julia toolchain there).  It is the `ccall` shim a maintainer of
# ChrisRackauckas/universal_differential_equations would add next to the scripts; the same entry points are
# exercised by the ctypes binding (universal_differential_equations_amd/_lib.py) in tests/.
#
# The reference's extension mechanism is multiple dispatch on the algorithm / sensealg / ensemble-algorithm type
# (SURVEY.md 8(b)); the shim adds
#   MI355Tsit5(), MI355Vern7()            algorithm types        (replace Tsit5()/Vern7() at scenario_1.jl:84,191)
#   EnsembleMI355()                       ensemble algorithm     (SciMLBase.__solve(::EnsembleProblem, alg, ::EnsembleMI355))
#   UDEModel(desc)                        the declarative right-hand side (a Julia closure cannot cross the C ABI)
# and an rrule so that Zygote (AutoZygote, scenario_1.jl:111) differentiates `Array(solve(...))` through
# ude_vjp_ensemble, i.e. InterpolatingAdjoint semantics (seir_exposure.jl:138-140, Fisher-KPP-CNN.jl:136).
module UDECoreMI355

using SciMLBase, DiffEqBase, ChainRulesCore

const libudecore = get(ENV, "UDECORE_LIB", "libudecore.so")
const UDE_MAX_LAYERS = 8

# ---- mirrors of the C structs (field order = include/udecore.h) -------------------------------------------------
struct ModelDesc
    kind::Int32; dtype::Int32; n_state::Int32; n_param::Int32; n_layers::Int32
    dims::NTuple{9,Int32}; act::NTuple{8,Int32}
    nn_offset::Int32; lin_idx::NTuple{2,Int32}; stencil_offset::Int32; d0_offset::Int32; reserved::Int32
    lin_sign::NTuple{2,Float64}; lin_const::NTuple{2,Float64}; consts::NTuple{16,Float64}
end
struct SolveOpts
    alg::Int32; maxiters::Int32
    abstol::Float64; reltol::Float64; dtmax::Float64; dt0::Float64
    qmin::Float64; qmax::Float64; gamma::Float64; qoldinit::Float64; beta1::Float64; beta2::Float64
    sensealg::Int32        # 0 InterpolatingAdjoint, 1 discretise-then-optimise (ForwardDiffSensitivity)
    per_trajectory::Int32  # UDE_PT_TSPAN = 1 (tspan is 2 x N), UDE_PT_SAVEAT = 2 (saveat is ns x N)
end

const KIND_LV_TRUE, KIND_LV_UDE, KIND_SEIR_TRUE, KIND_SEIR_UDE, KIND_KPP_TRUE, KIND_KPP_UDE, KIND_SEIR_NODE = Int32.(0:6)
const Real32or64 = Union{Float32,Float64}
dtypecode(::Type{Float64}) = Int32(0)
dtypecode(::Type{Float32}) = Int32(1)   # every real-valued array of a call is then Float32 (include/udecore.h: ude_model_desc.dtype)
const ACT = Dict(identity => Int32(0), tanh => Int32(1), :rbf => Int32(2), :relu => Int32(3))
pad(t, n, z) = ntuple(i -> i <= length(t) ? oftype(z, t[i]) : z, n)

"RHS descriptor standing in for `ude_dynamics!` / `dudt_` / `nn_ode`; callable so `ODEProblem(f, u0, tspan, p)` accepts it."
struct UDEModel
    desc::ModelDesc
end
(m::UDEModel)(du, u, p, t) = error("UDEModel is evaluated on the MI355X by libudecore; solve with MI355Tsit5()/MI355Vern7()")

"`ude_dynamics!` of scenario_1.jl:69-73 for `U = Lux.Chain(Dense(2,5,rbf),Dense(5,5,rbf),Dense(5,5,rbf),Dense(5,2))`"
function lv_ude(; dims = (2, 5, 5, 5, 2), acts = (:rbf, :rbf, :rbf, identity), p_true = (1.3, 0.9, 0.8, 1.8))
    np = sum(dims[i] * dims[i+1] + dims[i+1] for i in 1:length(dims)-1)
    UDEModel(ModelDesc(KIND_LV_UDE, 0, 2, np, length(dims) - 1, pad(dims, 9, Int32(0)), pad(map(a -> ACT[a], acts), 8, Int32(0)),
                       0, (Int32(-1), Int32(-1)), 0, 0, 0, (1.0, 1.0), (p_true[1], -p_true[4]), pad((), 16, 0.0)))
end
"`dudt_` of seir_exposure.jl:117-130 with `ann = FastChain(FastDense(3,64,tanh),FastDense(64,64,tanh),FastDense(64,1))`"
seir_ude(p_) = UDEModel(ModelDesc(KIND_SEIR_UDE, 0, 7, 4481, 3, pad((3, 64, 64, 1), 9, Int32(0)), pad((1, 1, 0), 8, Int32(0)),
                                  0, (Int32(-1), Int32(-1)), 0, 0, 0, (1.0, 1.0), (0.0, 0.0), pad(Tuple(p_), 16, 0.0)))
"`dudt_node` of seir_exposure.jl:53-66: the pure neural ODE, `ann_node = FastChain(FastDense(7,64,tanh), FastDense(64,64,tanh), FastDense(64,64,tanh), FastDense(64,7))`"
seir_node(p_) = UDEModel(ModelDesc(KIND_SEIR_NODE, 0, 7, 9287, 4, pad((7, 64, 64, 64, 7), 9, Int32(0)), pad((1, 1, 1, 0), 8, Int32(0)),
                                   0, (Int32(-1), Int32(-1)), 0, 0, 0, (1.0, 1.0), (0.0, 0.0), pad(Tuple(p_), 16, 0.0)))
"`nn_ode` of Fisher-KPP-CNN.jl:111-126 (theta = [rx_nn; w1 w2 w3; conv bias; D0])"
kpp_ude(Nx) = UDEModel(ModelDesc(KIND_KPP_UDE, 0, Nx, 466, 4, pad((1, 10, 20, 10, 1), 9, Int32(0)), pad((1, 1, 1, 0), 8, Int32(0)),
                                 0, (Int32(-1), Int32(-1)), 461, 465, 0, (1.0, 1.0), (0.0, 0.0), pad((), 16, 0.0)))

"`nn_ode` of Fisher-KPP-CNN-Small.jl:89-124 (15 parameters: rx_nn 1-3-1 tanh; w1 w2 w3; conv bias; D0)"
kpp_small_ude(Nx) = UDEModel(ModelDesc(KIND_KPP_UDE, 0, Nx, 15, 2, pad((1, 3, 1), 9, Int32(0)), pad((1, 0), 8, Int32(0)),
                                       0, (Int32(-1), Int32(-1)), 10, 14, 0, (1.0, 1.0), (0.0, 0.0), pad((), 16, 0.0)))
"`ude_dynamics!` of scenario_2.jl:90-95: theta = [delta; ude(87)], du2 = -delta u2 + NN2"
lv_ude_s2(; alpha = 1.3) = UDEModel(ModelDesc(KIND_LV_UDE, 0, 2, 88, 4, pad((2, 5, 5, 5, 2), 9, Int32(0)), pad((2, 2, 2, 0), 8, Int32(0)),
                                              1, (Int32(-1), Int32(0)), 0, 0, 0, (1.0, -1.0), (alpha, 0.0), pad((), 16, 0.0)))
"`ude_dynamics!` of hudson_bay.jl:85-91: theta = [p1; p2; FastChain(87)], third layer tanh (the script is Float32: `lv_ude_hudson(Float32)`)"
lv_ude_hudson(::Type{T} = Float64) where {T<:Real32or64} = UDEModel(ModelDesc(KIND_LV_UDE, dtypecode(T), 2, 89, 4, pad((2, 5, 5, 5, 2), 9, Int32(0)), pad((2, 2, 1, 0), 8, Int32(0)),
                                     2, (Int32(0), Int32(1)), 0, 0, 0, (1.0, -1.0), (0.0, 0.0), pad((), 16, 0.0)))

"`nn_ode` of scenario_3.jl:103-114 (Float32; ude 1-5-5-5-1 rbf, theta = [ude(76); p2s(4); D0])"
kpp_s3_ude(::Type{T} = Float32) where {T<:Real32or64} = UDEModel(ModelDesc(KIND_KPP_UDE, dtypecode(T), 26, 81, 4, pad((1, 5, 5, 5, 1), 9, Int32(0)),
    pad((2, 2, 2, 0), 8, Int32(0)), 0, (Int32(-1), Int32(-1)), 76, 80, 0, (1.0, 1.0), (0.0, 0.0), pad((), 16, 0.0)))
"`rc_ode` of scenario_3.jl:43-53 / Fisher-KPP-CNN.jl:51-63: consts = the entries of D*lap and r as the script forms them in `T`"
function kpp_true(Nx, D, r, dx, ::Type{T} = Float64) where {T<:Real32or64}
    dx2 = T(dx) * T(dx)
    off, dia = T(1.0 / Float64(dx2)), T(-2.0 / Float64(dx2))
    UDEModel(ModelDesc(KIND_KPP_TRUE, dtypecode(T), Nx, 0, 0, pad((), 9, Int32(0)), pad((), 8, Int32(0)), 0, (Int32(-1), Int32(-1)), 0, 0, 0,
                       (1.0, 1.0), (0.0, 0.0), pad((Float64(T(D) * off), Float64(T(D) * dia), Float64(T(r))), 16, 0.0)))
end

# ---- context ----------------------------------------------------------------------------------------------------
const CTXS = Dict{Int,Ptr{Cvoid}}()      # one context per device; a device is driven by one host task at a time
const TABLE_LOCK = ReentrantLock()       # CTXS / COMMS are plain Dicts: the per-device tasks of __solve look contexts up concurrently
function ctx(dev::Integer = 0)
    lock(TABLE_LOCK) do
        get!(CTXS, dev) do
            h = Ref{Ptr{Cvoid}}(C_NULL)
            rc = ccall((:ude_create, libudecore), Cint, (Int32, Ptr{Ptr{Cvoid}}), dev, h)
            rc == 0 || error("ude_create(device = $dev) failed ($rc)")
            h[]
        end
    end
end
# UDE_ERR_TRAJECTORY (-5): a member stopped early; like upstream the solution RETURNS and its retcode says why, but a
# gradient over a partial ensemble is an error unless the caller opted in (the loss is +Inf in that case)
function check(rc, dev = 0; allow_failures = false)
    (rc == 0 || (rc == -5 && allow_failures)) && return rc
    error(unsafe_string(ccall((:ude_last_error, libudecore), Cstring, (Ptr{Cvoid},), ctx(dev))))
end

# ---- algorithm types ---------------------------------------------------------------------------------------------
struct MI355Tsit5 <: SciMLBase.AbstractODEAlgorithm end
struct MI355Vern7 <: SciMLBase.AbstractODEAlgorithm end
"ensemble algorithm: trajectories run as lane groups of the fused HIP kernels, sharded in contiguous blocks over `devices`"
struct EnsembleMI355 <: SciMLBase.EnsembleAlgorithm
    devices::Vector{Int}
end
EnsembleMI355(; devices = [0]) = EnsembleMI355(collect(Int, devices))
algcode(::MI355Tsit5) = Int32(0)
algcode(::MI355Vern7) = Int32(1)
# sensealg codes (include/udecore.h): 0 InterpolatingAdjoint (default), 1 discretise-then-optimise (ForwardDiffSensitivity), 2 `fast`
# (interpolating adjoint with lambda-only error control: opt-in, not upstream's step sequence), 3 InterpolatingAdjoint(checkpointing = true)
# (store u only, recompute the stages: results identical to 0, an eighth of the dense store; Fisher-KPP UDEs with Tsit5)
opts(alg; abstol = 0.0, reltol = 0.0, dtmax = 0.0, dt = 0.0, maxiters = 0, discrete = false, fast = false, checkpointing = false,
     per_trajectory = 0, kw...) =
    SolveOpts(algcode(alg), maxiters, abstol, reltol, dtmax, dt, 0, 0, 0, 0, 0, 0, discrete ? 1 : (fast ? 2 : (checkpointing ? 3 : 0)), per_trajectory)
function grid(saveat::Number, tspan)      # SciML: save_end = true for a Number saveat
    ts = collect(tspan[1]:saveat:tspan[2])
    ts[end] < tspan[2] && push!(ts, tspan[2])
    ts
end
grid(saveat, tspan) = collect(Float64, saveat)

"u0s: n x N matrix (column j = trajectory j); returns (u::Array{Float64,3} n x ns x N, stats 8 x N, retcode N)"
"`f.(eachcol(U), Ref(θ))` on the device: the right-hand side closure evaluated once per state (columns of `U`)."
function rhs_ensemble(m::UDEModel, U::Matrix{T}, θ::Vector{T}) where {T<:Real32or64}
    m.desc.dtype == dtypecode(T) || error("descriptor dtype and array element type differ")
    n, N = size(U); dU = similar(U); d = Ref(m.desc)
    GC.@preserve U θ dU check(ccall((:ude_rhs_ensemble, libudecore), Cint,
        (Ptr{Cvoid}, Ref{ModelDesc}, Int64, Ptr{T}, Ptr{T}, Ptr{T}), ctx(), d, N, U, θ, dU))
    dU
end

"`tspan` a pair or a 2 x N matrix, `ts` a vector or an ns x N matrix (per-member spans / save grids: the UDE_PT_* flags)"
function solve_ensemble(m::UDEModel, alg, u0s::Matrix{T}, tspan, θ::Vector{T}, ts::AbstractVecOrMat{T}; dev = 0, kw...) where {T<:Real32or64}
    m.desc.dtype == dtypecode(T) || error("descriptor dtype and array element type differ")
    n, N = size(u0s); ns = size(ts, 1)
    flags = (tspan isa AbstractMatrix ? 1 : 0) | (ts isa AbstractMatrix ? 2 : 0)
    out = Array{T}(undef, n, ns, N); stats = zeros(Int64, 8, N); rc = zeros(Int32, N)
    d = Ref(m.desc); o = Ref(opts(alg; per_trajectory = flags, kw...))
    tsp = collect(Float64, vec(tspan isa Tuple ? [tspan...] : tspan)); tsa = collect(T, ts)     # tspan: always host doubles
    GC.@preserve u0s θ tsa out stats rc tsp check(ccall((:ude_solve_ensemble, libudecore), Cint,
        (Ptr{Cvoid}, Ref{ModelDesc}, Ref{SolveOpts}, Int64, Ptr{T}, Ptr{Float64}, Ptr{T}, Ptr{T}, Int32,
         Ptr{T}, Ptr{Int64}, Ptr{Int32}), ctx(dev), d, o, N, u0s, tsp, θ, tsa, ns, out, stats, rc), dev; allow_failures = true)
    out, stats, rc
end

"loss(θ) = sum(abs2, data[rows, :] .- Array(solve(...))[rows, :]) over the ensemble and its gradient in ONE call
(seir_exposure.jl:144-147 with `rows` = 2:4; Fisher-KPP-CNN.jl:140-143; scenario_1.jl:91-94).  `tspans` (2 x N) /
`tss` (ns x N) give every member its own span and save grid (the segments of scenario_2.jl:104-124)."
function loss_grad_ensemble(m::UDEModel, alg, u0s::Matrix{T}, tspan, θ::Vector{T}, ts::AbstractVecOrMat{T},
                            data::Array{T,3}; rows = nothing, dev = 0, kw...) where {T<:Real32or64}
    m.desc.dtype == dtypecode(T) || error("descriptor dtype and array element type differ")
    n, N = size(u0s); ns = size(ts, 1)
    flags = (tspan isa AbstractMatrix ? 1 : 0) | (ts isa AbstractMatrix ? 2 : 0)
    mask = rows === nothing ? C_NULL : UInt8[i in rows for i in 1:n]
    loss = Ref(zero(T)); lpt = zeros(T, N); gθ = zeros(T, length(θ)); gu0 = zeros(T, n, N); u = Array{T}(undef, n, ns, N)
    stats = zeros(Int64, 8, N); rc = zeros(Int32, N)
    d = Ref(m.desc); o = Ref(opts(alg; per_trajectory = flags, kw...)); tsp = collect(Float64, vec(tspan isa Tuple ? [tspan...] : tspan)); tsa = collect(T, ts)
    GC.@preserve u0s θ tsa data mask lpt gθ gu0 u stats rc tsp check(ccall((:ude_loss_grad_ensemble, libudecore), Cint,
        (Ptr{Cvoid}, Ref{ModelDesc}, Ref{SolveOpts}, Int64, Ptr{T}, Ptr{Float64}, Ptr{T}, Ptr{T}, Int32,
         Ptr{T}, Ptr{UInt8}, Ref{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{Int64}, Ptr{Int32}),
        ctx(dev), d, o, N, u0s, tsp, θ, tsa, ns, data, mask, loss, lpt, gθ, gu0, u, stats, rc), dev)
    loss[], gθ, gu0, u, stats, rc
end

# ---- multi-GPU (SURVEY.md 8(e)): contiguous blocks of trajectories per device, ONE all-reduce of [grad; loss; counters] ----
"`_dev` entry points take HBM pointers; the shim keeps each device's buffers in plain hipMalloc'ed memory owned by
the caller's GPU array package (AMDGPU.jl `ROCArray` pointers) -- shown here with `Ptr{Float64}` arguments."
const COMMS = Dict{Vector{Int},Vector{Ptr{Cvoid}}}()     # communicators are created once per device set and kept
function comms_for(devs::Vector{Int})
    lock(TABLE_LOCK) do   # (reentrant: ctx() takes it again)
        get!(COMMS, devs) do
            ctxs = [ctx(dv) for dv in devs]
            comms = Vector{Ptr{Cvoid}}(undef, length(devs))
            check(ccall((:ude_comm_create_local, libudecore), Cint, (Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}), length(devs), ctxs, comms), devs[1])
            comms
        end
    end
end
"every device's `payload` = double[np + 4] = [grad(np); loss; Σnf; Σnaccept; Σnreject] (SURVEY.md 8(e), DESIGN 7): gradient and loss are
written by the device's own ude_loss_grad_ensemble_dev, the three counters by ude_pack_counters_dev from the device's `stats` array
(forward + backward, exact integer sums as doubles), and ONE all-reduce sums the np + 4 values over the devices -- the same payload
python's `pack_payload` builds"
function loss_grad_multi(m::UDEModel, alg, devs::Vector{Int}, dptr::Vector{<:NamedTuple}, tspan, ts::Vector{Float64}; p2p = true, kw...)
    np = Int(m.desc.n_param); nd = length(devs)
    ctxs = [ctx(dv) for dv in devs]
    comms = comms_for(devs)
    d = Ref(m.desc); o = Ref(opts(alg; kw...)); tsp = Float64[tspan[1], tspan[2]]
    for (k, dv) in enumerate(devs)       # enqueue on every device; nothing blocks
        b = dptr[k]                      # (N, u0, theta, saveat, data, mask, payload = double[np + 4], gu0, u, stats, rc)
        check(ccall((:ude_loss_grad_ensemble_dev, libudecore), Cint,
            (Ptr{Cvoid}, Ref{ModelDesc}, Ref{SolveOpts}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32,
             Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}),
            ctxs[k], d, o, b.N, b.u0, tsp, b.theta, b.saveat, length(ts), b.data, b.mask, b.payload + 8np, C_NULL, b.payload, b.gu0, b.u,
            b.stats, b.rc), dv)
        check(ccall((:ude_pack_counters_dev, libudecore), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}, Int32), ctxs[k], b.N, b.stats, b.payload, np), dv)
    end
    bufs = [b.payload for b in dptr]
    f = p2p ? :ude_allreduce_grad_p2p : :ude_allreduce_grad_local      # fixed-rank-order peer reads, or RCCL (grouped)
    check(ccall((f, libudecore), Cint, (Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Float64}}, Int64), nd, comms, bufs, np + 4), devs[1])
    nothing                              # every device's payload now holds the ensemble-wide [grad; loss; Σnf; Σnaccept; Σnreject]
end

# ---- highdim_pde/lambaem.jl: NNPDENS + LambaEM (SURVEY.md 8(f) N1) ---------------------------------------------------
struct HjbDesc
    d::Int32; hls::Int32; adaptive::Int32; maxiters::Int32; max_steps::Int32; reserved::Int32; seed::UInt64
    lambda::Float64; sigma::Float64; t0::Float64; t1::Float64; abstol::Float64; reltol::Float64; dt::Float64
    qmin::Float64; qmax::Float64; gamma::Float64; qoldinit::Float64; beta1::Float64; beta2::Float64; dtmax::Float64
end
"one evaluation of loss_n_sde() and its gradient w.r.t. Flux.params(u0, σᵀ∇u) (flattened) for `trajectories` LambaEM solves
(lambaem.jl:33-34); `iter` = the training iteration (fresh Philox noise per iteration)"
function hjb_loss_grad(x0::Vector{Float32}, θ::Vector{Float32}, trajectories::Integer, iter::Integer; d = 100, hls = 110, λ = 1.0,
                       tspan = (0.0, 1.0), abstol = 1e-4, reltol = 1e-4, seed = 0, adaptive = true, dt = 0.0, max_steps = 0)
    # max_steps = 0: the accepted-step store grows on demand (ude_hjb_loss_grad re-runs a call that outgrew it)
    D = Ref(HjbDesc(d, hls, adaptive, 0, max_steps, 0, seed, λ, sqrt(2.0f0), tspan[1], tspan[2], abstol, reltol, dt, 0, 0, 0, 0, 0, 0, 0))
    loss = Ref(0.0); g = zeros(Float32, length(θ)); u0 = Ref(0.0f0); M = Int64(trajectories)
    GC.@preserve x0 θ g check(ccall((:ude_hjb_loss_grad, libudecore), Cint,
        (Ptr{Cvoid}, Ref{HjbDesc}, Int64, Ptr{Float32}, Ptr{Float32}, UInt32, Ref{Float64}, Ptr{Float32}, Ref{Float32}, Ptr{Float32},
         Ptr{Float32}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}), ctx(), D, M, x0, θ, iter, loss, g, u0, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
    loss[], g, u0[]
end

"cotangent Δ (n x ns x N) of the saved states -> (dθ summed over the ensemble, du0 n x N)"
function vjp_ensemble(m::UDEModel, alg, u0s, tspan, θ, ts, Δ::Array{Float64,3}; kw...)
    n, N = size(u0s); ns = length(ts)
    gθ = zeros(length(θ)); gu0 = zeros(n, N); stats = zeros(Int64, 8, N); rc = zeros(Int32, N)
    d = Ref(m.desc); o = Ref(opts(alg; kw...)); tsp = Float64[tspan[1], tspan[2]]
    GC.@preserve u0s θ ts Δ gθ gu0 stats rc tsp check(ccall((:ude_vjp_ensemble, libudecore), Cint,
        (Ptr{Cvoid}, Ref{ModelDesc}, Ref{SolveOpts}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}),
        ctx(), d, o, N, u0s, tsp, θ, ts, ns, Δ, C_NULL, gθ, gu0, stats, rc))
    gθ, gu0
end

# ---- SciMLBase surface: solve(prob, MI355Vern7(); saveat, abstol, reltol) ----------------------------------------
function DiffEqBase.__solve(prob::SciMLBase.AbstractODEProblem, alg::Union{MI355Tsit5,MI355Vern7};
                            saveat = get(prob.kwargs, :saveat, nothing), sensealg = nothing, kw...)
    m = prob.f.f::UDEModel
    ts = grid(saveat, prob.tspan)
    u, stats, rc = solve_ensemble(m, alg, reshape(Vector{Float64}(prob.u0), :, 1), prob.tspan, Vector{Float64}(prob.p), ts; kw...)
    retcode = rc[1] == 0 ? :Success : rc[1] == 1 ? :MaxIters : rc[1] == 2 ? :DtLessThanMin : :Unstable
    destats = DiffEqBase.DEStats(0)      # (a mutable struct in DiffEqBase 6.94 / SciMLBase 1.45, LotkaVolterra/Manifest.toml:390,1723)
    destats.nf = stats[1, 1]; destats.naccept = stats[2, 1]; destats.nreject = stats[3, 1]
    # dense = false: like upstream's saveat-only solutions, sol(t, Val{1}) interpolates linearly (scenario_1.jl:46)
    DiffEqBase.build_solution(prob, alg, ts, [u[:, i, 1] for i in 1:length(ts)]; retcode = retcode, destats = destats, dense = false)
end

# Zygote: Array(solve(remake(prob; u0, p = θ), alg; saveat, ...)) differentiated by the interpolating adjoint on the GPU
function ChainRulesCore.rrule(::typeof(DiffEqBase.solve_up), prob, sensealg, u0, p, alg::Union{MI355Tsit5,MI355Vern7}; saveat, kw...)
    sol = DiffEqBase.__solve(remake(prob; u0 = u0, p = p), alg; saveat = saveat, kw...)
    function pullback(Δ)
        ts = grid(saveat, prob.tspan)
        Δa = reshape(Array{Float64}(Δ isa AbstractArray ? Δ : Δ.u), length(u0), length(ts), 1)
        gθ, gu0 = vjp_ensemble(prob.f.f, alg, reshape(Vector{Float64}(u0), :, 1), prob.tspan, Vector{Float64}(p), ts, Δa; kw...)
        (NoTangent(), NoTangent(), NoTangent(), vec(gu0), gθ, NoTangent())
    end
    sol, pullback
end

# ensembles: solve(EnsembleProblem(prob; prob_func = (prob,i,_) -> remake(prob; u0 = u0s[:, i])), alg, EnsembleMI355(); trajectories = N)
# Every member keeps ITS OWN tspan and save grid (prob_func may remake both: the segments of scenario_2.jl:104-124 are
# remake(prob; u0, tspan = (T[1], T[end])) solved with saveat = T): they are handed over as 2 x N / ns x N arrays whenever
# they differ between members.  Members are sharded in contiguous blocks over `ea.devices` (one host task per device).
function SciMLBase.__solve(ens::SciMLBase.AbstractEnsembleProblem, alg::Union{MI355Tsit5,MI355Vern7}, ea::EnsembleMI355;
                           trajectories, saveat = nothing, kw...)
    probs = [ens.prob_func(ens.prob, i, 1) for i in 1:trajectories]
    u0s = reduce(hcat, [Vector{Float64}(p.u0) for p in probs])
    grids = [grid(something(saveat, get(p.kwargs, :saveat, nothing)), p.tspan) for p in probs]
    all(length(g) == length(grids[1]) for g in grids) || error("EnsembleMI355: every member needs the same NUMBER of save points")
    same_span = all(p.tspan == probs[1].tspan for p in probs); same_grid = all(g == grids[1] for g in grids)
    tspans = same_span ? probs[1].tspan : reduce(hcat, [Float64[p.tspan[1], p.tspan[2]] for p in probs])
    tss = same_grid ? grids[1] : reduce(hcat, grids)
    all(p.p == probs[1].p for p in probs) || error("EnsembleMI355: the members of an ensemble share the parameter vector")
    θ = Vector{Float64}(probs[1].p); N = trajectories; nd = length(ea.devices)
    u = Array{Float64}(undef, size(u0s, 1), length(grids[1]), N); rc = zeros(Int32, N)
    bounds = [(div((k - 1) * N, nd) + 1, div(k * N, nd)) for k in 1:nd]
    foreach(ctx, ea.devices)   # contexts exist before the per-device tasks start (and the tables are locked anyway)
    @sync for (k, dv) in enumerate(ea.devices)
        lo, hi = bounds[k]; hi >= lo || continue
        Threads.@spawn begin
            uk, _, rck = solve_ensemble(ens.prob.f.f, alg, u0s[:, lo:hi], same_span ? tspans : tspans[:, lo:hi], θ,
                                        same_grid ? tss : tss[:, lo:hi]; dev = dv, kw...)
            u[:, :, lo:hi] .= uk; rc[lo:hi] .= rck
        end
    end
    SciMLBase.EnsembleSolution([DiffEqBase.build_solution(probs[j], alg, grids[j], [u[:, i, j] for i in 1:length(grids[j])];
                                                          retcode = rc[j] == 0 ? :Success : :Failure) for j in 1:trajectories], 0.0, true)
end

end # module
