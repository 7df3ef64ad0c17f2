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
    sensealg::Int32; reserved::Int32   # 0 InterpolatingAdjoint, 1 discretise-then-optimise (ForwardDiffSensitivity)
end

const KIND_LV_TRUE, KIND_LV_UDE, KIND_SEIR_TRUE, KIND_SEIR_UDE, KIND_KPP_TRUE, KIND_KPP_UDE = Int32.(0:5)
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
"`nn_ode` of Fisher-KPP-CNN.jl:111-126 (theta = [rx_nn; w1 w2 w3; conv bias; D0])"
kpp_ude(Nx) = UDEModel(ModelDesc(KIND_KPP_UDE, 0, Nx, 466, 4, pad((1, 10, 20, 10, 1), 9, Int32(0)), pad((1, 1, 1, 0), 8, Int32(0)),
                                 0, (Int32(-1), Int32(-1)), 461, 465, 0, (1.0, 1.0), (0.0, 0.0), pad((), 16, 0.0)))

# ---- context ----------------------------------------------------------------------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
function ctx()
    if CTX[] == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:ude_create, libudecore), Cint, (Int32, Ptr{Ptr{Cvoid}}), 0, h)
        rc == 0 || error("ude_create failed ($rc)")
        CTX[] = h[]
    end
    CTX[]
end
check(rc) = (rc == 0 || rc == -5) ? rc : error(unsafe_string(ccall((:ude_last_error, libudecore), Cstring, (Ptr{Cvoid},), ctx())))

# ---- algorithm types ---------------------------------------------------------------------------------------------
struct MI355Tsit5 <: SciMLBase.AbstractODEAlgorithm end
struct MI355Vern7 <: SciMLBase.AbstractODEAlgorithm end
struct EnsembleMI355 <: SciMLBase.EnsembleAlgorithm end
algcode(::MI355Tsit5) = Int32(0)
algcode(::MI355Vern7) = Int32(1)
opts(alg; abstol = 0.0, reltol = 0.0, dtmax = 0.0, dt = 0.0, maxiters = 0, discrete = false, kw...) =
    SolveOpts(algcode(alg), maxiters, abstol, reltol, dtmax, dt, 0, 0, 0, 0, 0, 0, discrete ? 1 : 0, 0)
grid(saveat::Number, tspan) = collect(tspan[1]:saveat:tspan[2])
grid(saveat, tspan) = collect(Float64, saveat)

"u0s: n x N matrix (column j = trajectory j); returns (u::Array{Float64,3} n x ns x N, stats 8 x N, retcode N)"
"`f.(eachcol(U), Ref(θ))` on the device: the right-hand side closure evaluated once per state (columns of `U`)."
function rhs_ensemble(m::UDEModel, U::Matrix{Float64}, θ::Vector{Float64})
    n, N = size(U); dU = similar(U); d = Ref(m.desc)
    GC.@preserve U θ dU check(ccall((:ude_rhs_ensemble, libudecore), Cint,
        (Ptr{Cvoid}, Ref{ModelDesc}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx(), d, N, U, θ, dU))
    dU
end

function solve_ensemble(m::UDEModel, alg, u0s::Matrix{Float64}, tspan, θ::Vector{Float64}, ts::Vector{Float64}; kw...)
    n, N = size(u0s); ns = length(ts)
    out = Array{Float64}(undef, n, ns, N); stats = zeros(Int64, 8, N); rc = zeros(Int32, N)
    d = Ref(m.desc); o = Ref(opts(alg; kw...)); tsp = Float64[tspan[1], tspan[2]]
    GC.@preserve u0s θ ts out stats rc tsp check(ccall((:ude_solve_ensemble, libudecore), Cint,
        (Ptr{Cvoid}, Ref{ModelDesc}, Ref{SolveOpts}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32,
         Ptr{Float64}, Ptr{Int64}, Ptr{Int32}), ctx(), d, o, N, u0s, tsp, θ, ts, ns, out, stats, rc))
    out, stats, rc
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
    sol = DiffEqBase.build_solution(prob, alg, ts, [u[:, i, 1] for i in 1:length(ts)]; retcode = retcode)
    sol.destats.nf = stats[1, 1]; sol.destats.naccept = stats[2, 1]; sol.destats.nreject = stats[3, 1]
    sol
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
function SciMLBase.__solve(ens::SciMLBase.AbstractEnsembleProblem, alg::Union{MI355Tsit5,MI355Vern7}, ::EnsembleMI355;
                           trajectories, saveat, kw...)
    probs = [ens.prob_func(ens.prob, i, 1) for i in 1:trajectories]
    u0s = reduce(hcat, [Vector{Float64}(p.u0) for p in probs])
    ts = grid(saveat, ens.prob.tspan)
    u, stats, rc = solve_ensemble(ens.prob.f.f, alg, u0s, ens.prob.tspan, Vector{Float64}(ens.prob.p), ts; kw...)
    SciMLBase.EnsembleSolution([DiffEqBase.build_solution(probs[j], alg, ts, [u[:, i, j] for i in 1:length(ts)]) for j in 1:trajectories], 0.0, true)
end

end # module
