/*
 * udecore.h -- C ABI of the MI355X-native UDE training core (libudecore.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference has no
 * FFI of its own: its extension mechanism is Julia multiple dispatch on the algorithm / sensealg /
 * ensemble-algorithm types.  Each entry point below names the reference interface it replaces; the
 * Julia `ccall` shim that a maintainer would add is in INTEGRATION.md / julia/UDECoreMI355.jl.
 *
 * Conventions: plain pointers and sizes, no torch types; every function returns 0 on success or a
 * negative UDE_ERR_* (message via ude_last_error); the caller owns every buffer.  Arrays are
 * column-major as Julia hands them over: u0 is n x N (trajectory j = column j), u_out / data /
 * cotangent are n x ns x N, theta is the flat parameter vector ([leading scalars; per Dense layer
 * vec(W) (out x in, column-major); b]).  Device-resident (`_dev`) entry points take HBM pointers of
 * the same layouts and are asynchronous on the context's stream; the host-buffer entry points copy
 * H<->D and block.  One context per device and host thread.
 */
#ifndef UDECORE_H
#define UDECORE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UDE_VERSION 100 /* 0.1.0 */
#define UDE_MAX_LAYERS 8
#define UDE_NSTATS 8

/* right-hand-side families ("declarative RHS descriptor": Julia closures cannot cross a C ABI) */
enum {
    UDE_KIND_LV_TRUE = 0,   /* lotka!        LotkaVolterra/scenario_1.jl:30-34 */
    UDE_KIND_LV_UDE = 1,    /* ude_dynamics! scenario_1.jl:69-73, scenario_2.jl:90-95, hudson_bay.jl:85-91 */
    UDE_KIND_SEIR_TRUE = 2, /* corona!       SEIR_exposure/seir_exposure.jl:16-30 */
    UDE_KIND_SEIR_UDE = 3,  /* dudt_         seir_exposure.jl:117-130 */
    UDE_KIND_KPP_TRUE = 4,  /* rc_ode        FisherKPP/Fisher-KPP-CNN.jl:51-63, LotkaVolterra/scenario_3.jl:43-53 */
    UDE_KIND_KPP_UDE = 5,   /* nn_ode        Fisher-KPP-CNN.jl:111-126, scenario_3.jl:103-114 */
    UDE_KIND_SEIR_NODE = 6  /* dudt_node     seir_exposure.jl:53-66: pure neural ODE 7-64-64-64-7 tanh, first five outputs = dS,dE,dI,dR,dD */
};
enum { UDE_ACT_IDENTITY = 0, UDE_ACT_TANH = 1, UDE_ACT_RBF = 2 /* scenario_1.jl:59 */, UDE_ACT_RELU = 3 };
/* sensealg: InterpolatingAdjoint(autojacvec=ReverseDiffVJP()) seir_exposure.jl:140, Fisher-KPP-CNN.jl:136;
 * discretise-then-optimise = the frozen-step derivative that ForwardDiffSensitivity() requests, scenario_1.jl:86 */
enum { UDE_SENSE_INTERPOLATING_ADJOINT = 0, UDE_SENSE_DISCRETE = 1,
       /* SURVEY.md 8(b) `fast` mode: the interpolating adjoint with ONLY lambda under error control.  The parameter cotangent
        * is carried as a quadrature on the adjoint's accepted steps (mu += dt * sum_s b_s g_s): no per-parameter error
        * estimate, no candidate/commit (cheaper steps, not fewer).  Not the step sequence of upstream's InterpolatingAdjoint
        * (which keeps mu in the error norm) -- an opt-in; gradients agree with mode 0 to the solver tolerance.  Shared time
        * grids only (per_trajectory = 0).  On the SEIR exposure UDE / neural ODE at the default lanes_per_traj this mode is the block-level
        * matrix-core accumulation (sixteen trajectories share one accumulator): a trajectory whose BACKWARD solve stops early cannot be
        * taken out of the sum again, so a call in which one does returns grad_theta = NaN (every entry; beside UDE_ERR_TRAJECTORY and
        * the infinite loss) -- never a gradient polluted by partial adjoints.  Forward failures are skipped before they contribute. */
       UDE_SENSE_INTERPOLATING_ADJOINT_FAST = 2,
       /* InterpolatingAdjoint(checkpointing = true), as store-u-only + recompute (SURVEY.md 8(b) `store dense|recompute`): the forward
        * pass keeps (t, t_end, dt, u) of every accepted step instead of u and all stage derivatives -- 1/(1 + stages) of the dense store
        * (Fisher-KPP 1024 points, Tsit5: 8.2 KB instead of 65.6 KB per step) -- and the adjoint kernel re-runs a step's stages when it
        * enters its interval.  Same operation sequence on the same inputs: every result is bit-identical to mode 0; the price is
        * the stages (+ Vern7's six lazy dense-output stages) once more per forward interval.  Every model and both algorithms on a shared
        * time grid (the lock-step SEIR / neural-ODE kernels hand over to the wavefront-per-trajectory ones), except a distributed state
        * whose recomputed stage vectors do not fit the registers (1024-point Fisher-KPP with Vern7): UDE_ERR_UNSUPPORTED */
       UDE_SENSE_INTERPOLATING_ADJOINT_CHECKPOINTED = 3 };
enum { UDE_ALG_TSIT5 = 0 /* Tsit5() scenario_1.jl:191 */, UDE_ALG_VERN7 = 1 /* Vern7() scenario_1.jl:84 */ };
/* per-trajectory return codes mirror the SciML retcodes stored in the reference's artifacts */
enum { UDE_RET_SUCCESS = 0, UDE_RET_MAXITERS = 1, UDE_RET_DTLESSTHANMIN = 2, UDE_RET_UNSTABLE = 3,
       UDE_RET_DENSE_OVERFLOW = 4 /* forward pass exceeded opts.max_dense_steps (adjoint only) */ };
enum { UDE_OK = 0, UDE_ERR_INVALID = -1, UDE_ERR_UNSUPPORTED = -2, UDE_ERR_HIP = -3, UDE_ERR_NOMEM = -4,
       UDE_ERR_TRAJECTORY = -5 /* at least one trajectory has retcode != Success */,
       UDE_ERR_TIMEOUT = -6 /* a cross-process all-reduce gave up waiting for a peer: the communicator is dead (sticky), make a new one */ };

/* replaces: the RHS closure + Lux/FastChain/Flux model captured by ODEProblem(f, u0, tspan, p)
 * (scenario_1.jl:62-78, seir_exposure.jl:114-131, Fisher-KPP-CNN.jl:92-131) */
typedef struct {
    int32_t kind;
    int32_t dtype;                     /* 0 = Float64; 1 = Float32 (scenario_3.jl:26-57,121-126; hudson_bay.jl:77-104): EVERY real-valued
                                          array argument of the call (u0, theta, saveat, u_out, data, cotangent, grad_*, loss,
                                          loss_per_traj) is then float behind the same pointers; tspan stays a pair of host doubles.
                                          (the array parameters are ude_real* = void* for that reason)
                                          Compiled Float32 instances: LV UDE 2-5-5-5-2 rbf/rbf/tanh (hudson), Fisher-KPP true and
                                          its 1-5-5-5-1 rbf UDE on <= 32 points; anything else: UDE_ERR_UNSUPPORTED */
    int32_t n_state;
    int32_t n_param;                   /* length of theta */
    int32_t n_layers;                  /* number of Dense layers (0 for mechanistic kinds) */
    int32_t dims[UDE_MAX_LAYERS + 1];  /* dims[0] = in ... dims[n_layers] = out */
    int32_t act[UDE_MAX_LAYERS];
    int32_t nn_offset;                 /* theta index of the first NN parameter */
    int32_t lin_idx[2];                /* LV_UDE: theta index of a trainable diagonal coefficient, or -1 */
    int32_t stencil_offset;            /* KPP_UDE: theta index of w1 (w1, w2, w3, one unused slot) */
    int32_t d0_offset;                 /* KPP_UDE: theta index of D0 */
    int32_t reserved;
    double lin_sign[2];                /* LV_UDE: du_i = (lin_idx<0 ? lin_const : lin_sign*theta[lin_idx])*u_i + NN_i(u) */
    double lin_const[2];
    double consts[16];                 /* SEIR: p_[0..8]; KPP_TRUE: D/dx^2, -2D/dx^2, r */
} ude_model_desc;

/* replaces: the keyword arguments of solve(prob, alg; saveat, abstol, reltol, ...) actually used by the
 * scripts (scenario_1.jl:84-87, seir_exposure.jl:138-140, Fisher-KPP-CNN.jl:136); zero = SciML default */
typedef struct {
    int32_t alg;
    int32_t maxiters;   /* <=0 -> 100000 */
    double abstol;      /* <=0 -> 1e-6 */
    double reltol;      /* <=0 -> 1e-3 */
    double dtmax;       /* <=0 -> |tf - t0| */
    double dt0;         /* >0 -> initial dt instead of the Hairer heuristic */
    double qmin, qmax, gamma, qoldinit; /* <=0 -> 0.2, 10, 0.9, 1e-4 */
    double beta1, beta2;                /* <=0 -> 7/(10 order), 2/(5 order) */
    int32_t sensealg;   /* UDE_SENSE_*: how ude_vjp / ude_loss_grad differentiate */
    int32_t per_trajectory; /* UDE_PT_* flags: every ensemble member has its own tspan (2 x N) and / or save grid (ns x N,
                               strictly increasing inside its tspan) -- remake(prob; tspan = (T[1], T[end])) + saveat = T per
                               segment, scenario_2.jl:104-124 -- and / or its own parameter vector (UDE_PT_THETA); 0 = shared */
} ude_solve_opts;
enum { UDE_PT_TSPAN = 1, UDE_PT_SAVEAT = 2,
       /* per-member PARAMETERS: theta is np x N (column j = member j's parameters) and grad_theta comes back np x N, one gradient per
        * member, nothing summed -- LotkaVolterra/run_loops.jl:55-62 runs 500 independent recoveries, each with its own data AND its own
        * network; `loss` is still the sum, loss_per_traj the members' own.  The compiled LV-kind instances (scenario_1 / scenario_2 /
        * hudson_bay's chains, Float64 and Float32: the member's weights live in registers; the 2-32-2 net: read from the member's
        * column at every use) and the Fisher-KPP kinds (FisherKPP/Fisher-KPP-CNN-Small.jl:311-391: five trainings from five initial
        * networks), interpolating adjoint and the discrete sweep; UDE_ERR_UNSUPPORTED elsewhere.  May be combined with the two
        * flags above. */
       UDE_PT_THETA = 4 };

/* launch/tuning knobs of the HIP back end (not part of the reference surface) */
typedef struct {
    int32_t lanes_per_traj;   /* 0 = auto; lanes cooperating on one trajectory.  Compiled variants: LV 2-5-5-5-2: 1, 4, 5 (default:
                                 12 trajectories per wavefront), 8; LV 2-32-2: 8, 32 (default); SEIR UDE: 64 (wavefront per
                                 trajectory), 256 (four wavefronts), 16 (default for the interpolating adjoint: lock-step backward
                                 kernel, 16 trajectories per block as columns of FP64 matrix-core products; plain solves and the forward pass of a gradient
                                 run on the same architecture); SEIR neural ODE 7-64-64-64-7: 64 (wavefront per trajectory), 16 (default for the
                                 interpolating adjoint: lock-step backward kernel); Fisher-KPP UDE <= 32 points: 32; 1024 points: 256 (default,
                                 four wavefronts per PDE in the adjoint, eight in the forward solve) or 64.  Every variant returns identical bits. */
    int32_t block_threads;    /* 0 = auto (64) */
    int32_t max_dense_steps;  /* capacity of the dense forward store per trajectory; 0 = automatic: starts at 256 and is
                                 re-run with 4x the capacity on UDE_RET_DENSE_OVERFLOW by the host-buffer entry points
                                 (ude_last_failures does the same for the asynchronous _dev entry points) */
    int32_t waves_per_simd;   /* kernel variant of the adjoint kernel: 0/1 = default (the only compiled one; the 2-waves-per-SIMD
                                 builds of the 4- and 8-lane LV kernels spill to scratch and were measured 5x slower) */
} ude_launch_opts;

/* per-trajectory stats, int64[UDE_NSTATS]:
 * 0 nf (= upstream destats.nf) 1 naccept 2 nreject 3 nf_lazy (Vern7 lazy dense-output evals)
 * 4 nf_bwd (augmented adjoint RHS evals) 5 naccept_bwd 6 nreject_bwd 7 nf_fwd_lazy_for_adjoint */

typedef struct ude_ctx ude_ctx;

int ude_version(void);
/* one context per (device, host thread); replaces nothing (the reference is single-process CPU Julia) */
int ude_create(int32_t device_id, ude_ctx** out);
void ude_destroy(ude_ctx* ctx);
const char* ude_last_error(ude_ctx* ctx);
/* stream on which the _dev entry points enqueue (a hipStream_t); NULL = the null stream.  Rebinding orders the new stream behind the
 * work already enqueued on the old one (an event; if the old stream no longer exists: a device synchronise).
 * hipGraph capture: the _dev entry points may be captured on the bound stream (the per-kernel timing events are left out).  A captured
 * graph bakes in the context's workspaces: while it is alive, use the context only with the SAME problem sizes (nothing may make the
 * workspaces grow -- a larger ensemble, an automatic dense-store regrowth after UDE_RET_DENSE_OVERFLOW) and do not run ordinary calls of
 * the same context on another stream concurrently with a replay: the library keeps no ordering between them. */
int ude_set_stream(ude_ctx* ctx, void* hip_stream);
int ude_set_launch_opts(ude_ctx* ctx, const ude_launch_opts* lo);
/* 0: this (model, alg) runs on a compiled fast instance; 1: it runs on a runtime-shape fallback kernel -- any chain of <= 8 Dense
 * layers of width <= 64, activations identity / tanh / rbf / relu, for UDE_KIND_LV_UDE 2 -> 2 (Float64 and Float32),
 * UDE_KIND_SEIR_UDE 3 -> 1 and UDE_KIND_SEIR_NODE 7 -> 7 (Float64); any pointwise reaction chain 1 -> ... -> 1 of <= 4 layers of
 * width <= 32 (<= 768 parameters) for UDE_KIND_KPP_UDE on <= 32 grid points (Float64); every sensealg; slower, bit-identical to the
 * oracle all the same; UDE_ERR_UNSUPPORTED otherwise */
int ude_model_supported(ude_ctx* ctx, const ude_model_desc* m, const ude_solve_opts* o, int32_t need_adjoint);

/* Real-valued arrays: `ude_real` is void -- the element type is the problem's scalar type, double for
 * ude_model_desc.dtype == 0 and float for dtype == 1 (u0, theta, saveat, u_out, data, cotangent, grad_*, loss, loss_per_traj).
 * tspan is always a pair (or 2 x N) of HOST doubles. */
typedef void ude_real;

/* replaces: Array(solve(remake(prob; u0, tspan, p = theta), alg; saveat, abstol, reltol)) for every
 * member of an ensemble sharing theta (predict: scenario_1.jl:82-88; scenario_2.jl:113-124 segments;
 * EnsembleProblem slot: SciMLBase.__solve(::EnsembleProblem, alg, ::EnsembleAlgorithm)). */
int ude_solve_ensemble(ude_ctx* ctx, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                       const ude_real* u0, const double* tspan /* 2, or 2 x N with UDE_PT_TSPAN; always a HOST pointer */,
                       const ude_real* theta, const ude_real* saveat /* ns, or ns x N with UDE_PT_SAVEAT */, int32_t ns,
                       ude_real* u_out, int64_t* stats, int32_t* retcode);
int ude_solve_ensemble_dev(ude_ctx* ctx, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                           const ude_real* u0, const double* tspan, const ude_real* theta,
                           const ude_real* saveat, int32_t ns, ude_real* u_out, int64_t* stats, int32_t* retcode);

/* replaces: the Zygote pullback of concrete_solve(prob, alg, u0, theta; saveat,
 * sensealg = InterpolatingAdjoint(autojacvec = ReverseDiffVJP())) (seir_exposure.jl:138-140,
 * Fisher-KPP-CNN.jl:136): cotangent (n x ns x N) -> grad_theta (np, summed over the ensemble),
 * grad_u0 (n x N, may be NULL).  u_out (may be NULL) receives the primal sol(saveat). */
int ude_vjp_ensemble(ude_ctx* ctx, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                     const ude_real* u0, const double* tspan, const ude_real* theta, const ude_real* saveat,
                     int32_t ns, const ude_real* cotangent, ude_real* u_out, ude_real* grad_theta,
                     ude_real* grad_u0, int64_t* stats, int32_t* retcode);
int ude_vjp_ensemble_dev(ude_ctx* ctx, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                         const ude_real* u0, const double* tspan, const ude_real* theta, const ude_real* saveat,
                         int32_t ns, const ude_real* cotangent, ude_real* u_out, ude_real* grad_theta,
                         ude_real* grad_u0, int64_t* stats, int32_t* retcode);

/* replaces: loss(theta) = sum(abs2, data[rows,:] .- predict(theta)[rows,:]) and its gradient
 * (seir_exposure.jl:144-147 rows 2:4; Fisher-KPP-CNN.jl:140-143 without the host-side penalty term;
 * scenario_1.jl:91-94), summed over the ensemble.  row_mask: n bytes (NULL = all rows).
 * loss: one double.  loss_per_traj (N doubles) may be NULL. */
int ude_loss_grad_ensemble(ude_ctx* ctx, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                           const ude_real* u0, const double* tspan, const ude_real* theta,
                           const ude_real* saveat, int32_t ns, const ude_real* data, const uint8_t* row_mask,
                           ude_real* loss, ude_real* loss_per_traj, ude_real* grad_theta, ude_real* grad_u0,
                           ude_real* u_out, int64_t* stats, int32_t* retcode);
int ude_loss_grad_ensemble_dev(ude_ctx* ctx, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                               const ude_real* u0, const double* tspan, const ude_real* theta,
                               const ude_real* saveat, int32_t ns, const ude_real* data, const uint8_t* row_mask,
                               ude_real* loss, ude_real* loss_per_traj, ude_real* grad_theta, ude_real* grad_u0,
                               ude_real* u_out, int64_t* stats, int32_t* retcode);

/* replaces: one evaluation of the right-hand side closure, `f(u, p, t)` / `ude_dynamics!(du, u, p, t)` for a batch of states --
 * what the scripts do with the trained UDE on the saved states before SINDy (`U(X_hat, p_trained, st)`,
 * scenario_1.jl:152-160, applied to the whole right-hand side here).  u, du: n x N (one state per column). */
int ude_rhs_ensemble(ude_ctx* ctx, const ude_model_desc* model, int64_t N, const ude_real* u_host, const ude_real* theta_host, ude_real* du_host);
int ude_rhs_ensemble_dev(ude_ctx* ctx, const ude_model_desc* model, int64_t N, const ude_real* u, const ude_real* theta, ude_real* du);

/* ---------------------------------------------------------------------------------------------------------------
 * SURVEY.md 8(f) N1 / BASELINE configs[4]: highdim_pde/lambaem.jl -- the deep-BSDE solver NNPDENS (NeuralNetDiffEq
 * 1.1.0) for a 100-dimensional Hamilton-Jacobi-Bellman equation: an ensemble of `trajectories` adaptive
 * Euler-Maruyama (StochasticDiffEq `LambaEM()`) solves of the (d+1)-dimensional SDE
 *   dX = sigma dW,   du = lambda |z|^2 dt + z . dW,   z = sigma^T grad u net([X; t]),   u(0) = u0 net(x0),
 * loss = mean_j (g(X_T) - u_T)^2 with g(X) = log(0.5 + 0.5 |X|^2), differentiated through the stepper (Tracker
 * upstream; here the reverse sweep over the recorded accepted steps).  Float32 throughout (lambaem.jl:9-10).
 * ude_hjb_desc replaces: TerminalPDEProblem(g, f, mu, sigma, x0, tspan) + NNPDENS(u0, sigma^T grad u) + the solve
 * keywords alg = LambaEM(), abstol, reltol (lambaem.jl:14-34) -- a declarative descriptor of THIS problem family
 * (g, f, mu = 0 and the constant diagonal sigma of lambaem.jl:14-17); the chains are Flux.Chain(Dense(d,hls,relu),
 * Dense(hls,hls,relu), Dense(hls,1)) and Chain(Dense(d+1,hls,relu), Dense(hls,hls,relu), Dense(hls,hls,relu),
 * Dense(hls,d)) (lambaem.jl:23-30), theta = Flux.params order [u0 chain; sigma^T grad u chain], per Dense layer W
 * (out x in, column-major) then b.  Compiled instance: d = 100, hls = 110.
 * Random numbers: counter-based Philox4x32-10, key = seed, counter = (chunk, draw event, trajectory, iteration) --
 * Julia's MersenneTwister stream (Random.seed!(0), lambaem.jl:7) is not reproducible outside Julia. */
typedef struct {
    int32_t d;          /* 100 (lambaem.jl:8) */
    int32_t hls;        /* 10 + d (lambaem.jl:20) */
    int32_t adaptive;   /* 1 = LambaEM with error control (lambaem.jl:33); 0 = fixed-step Euler-Maruyama with `dt` */
    int32_t maxiters;   /* step attempts per trajectory, <= 0 -> 1000000 */
    int32_t max_steps;  /* capacity of the accepted-step store per trajectory (2.2 KB per step); <= 0 = automatic: starts at 512; a trajectory
                           that outgrows it keeps stepping unrecorded and reports its true count, the capacity becomes the largest
                           count of the call (+ 1/8) and the call is repeated ONCE (ude_hjb_loss_grad does that by itself,
                           ude_hjb_last_failures for the asynchronous _dev entry point); loss-only calls record nothing */
    int32_t reserved;
    uint64_t seed;      /* Philox key */
    double lambda;      /* lambaem.jl:12 */
    double sigma;       /* diagonal of sigma: sqrt(2f0) (lambaem.jl:17) */
    double t0, t1;      /* tspan (lambaem.jl:10) */
    double abstol, reltol; /* lambaem.jl:34 */
    double dt;          /* adaptive = 0: the step; adaptive = 1: > 0 overrides the initial-dt heuristic */
    double qmin, qmax, gamma, qoldinit, beta1, beta2, dtmax; /* <= 0 -> StochasticDiffEq defaults 1/5, 9/8, 9/10, 1e-4, 7/10, 2/5, t1 - t0 */
} ude_hjb_desc;
enum { UDE_HJB_NSTATS = 4 }; /* per trajectory: 0 network evaluations, 1 naccept, 2 nreject, 3 random draw events */
enum { UDE_RET_STORE_OVERFLOW = 4 /* more accepted steps than ude_hjb_desc.max_steps (gradient calls only) */,
       UDE_RET_STACK_OVERFLOW = 5 /* more than 32 unconsumed rejected increments (RSwM stack) */ };

int ude_hjb_num_params(int32_t d, int32_t hls, int32_t* np_u0, int32_t* np_sg);
/* replaces: one evaluation of loss_n_sde() and its Tracker gradient inside Flux.train!(loss_n_sde, ps, data, opt)
 * of solve(prob::TerminalPDEProblem, pdealg::NNPDENS; trajectories = M, alg = LambaEM(), abstol, reltol)
 * (lambaem.jl:33-34).  iter = the training iteration (a Philox counter word: fresh noise per iteration).
 * Device pointers: x0 (d), theta (np_u0 + np_sg), loss (1 double), grad (np floats, or NULL = loss only),
 * u0_out (1 float = u0 net(x0), the PDE solution estimate the script returns, or NULL), uT (M or NULL), XT (d x M or NULL),
 * loss_traj (M doubles or NULL), stats (UDE_HJB_NSTATS x M int64 or NULL), retcode (M int32 or NULL).
 * Failed trajectories are left out of the gradient and make the loss +Inf.  Asynchronous on the context's stream. */
int ude_hjb_loss_grad_dev(ude_ctx* ctx, const ude_hjb_desc* D, int64_t M, const float* x0, const float* theta, uint32_t iter,
                          double* loss, float* grad, float* u0_out, float* uT, float* XT, double* loss_traj,
                          int64_t* stats, int32_t* retcode);
/* the same with host buffers (what a Julia ccall binds); blocks; UDE_ERR_TRAJECTORY if a trajectory failed */
int ude_hjb_loss_grad(ude_ctx* ctx, const ude_hjb_desc* D, int64_t M, const float* x0, const float* theta, uint32_t iter,
                      double* loss, float* grad, float* u0_out, float* uT, float* XT, double* loss_traj,
                      int64_t* stats, int32_t* retcode);
/* parity aids: the d standard normals of one draw event as the kernels generate them (host out, d doubles);
 * one evaluation of the sigma^T grad u chain on the FP32 matrix cores for n input columns (x_in: (d+1) x n, z: d x n, host) */
int ude_hjb_normals(ude_ctx* ctx, uint64_t seed, uint32_t iter, uint32_t traj, uint32_t event, int32_t d, double* out_host);
int ude_hjb_net(ude_ctx* ctx, int32_t d, int32_t hls, const float* theta_sg_host, int64_t n, const float* x_in_host, float* z_host);
/* debugging aid for parity work: raw float copy out of a workspace of the most recent call (0: [u0, initial dt, ...],
 * 1: the accepted-step records [trajectory][step][104] = X_n (100), t_n; 5: [trajectory][step][100] = 2 lambda dt z + dW) */
int ude_hjb_debug_read(ude_ctx* ctx, int32_t which, int64_t offset_floats, int64_t n_floats, float* out_host);
/* failure accounting of the most recent ude_hjb_loss_grad_dev call on this context (blocks on its stream): *nfail = trajectories
 * whose retcode is not Success (retcode_dev = the array passed to the call, or NULL for the context's own); if one of them outgrew the
 * AUTOMATIC accepted-step store, its capacity is raised to the largest accepted-step count of that call (+ 1/8) and *grown = 1: the caller repeats the call */
int ude_hjb_last_failures(ude_ctx* ctx, const int32_t* retcode_dev, int64_t M, int32_t* nfail, int32_t* grown);
/* device time (ms) of the forward and backward kernels of the most recent ude_hjb_loss_grad* call (HIP events on the stream) */
int ude_hjb_last_kernel_ms(ude_ctx* ctx, float* fwd_ms, float* bwd_ms);

/* ---------------------------------------------------------------------------------------------------------------
 * Multi-GPU: trajectories shard in contiguous blocks, theta is replicated, and the ONE exchange per
 * gradient is all-reduce(sum) of double[np + 4] = gradient (+) loss (+) (sum nf, sum naccept, sum nreject).  The
 * reference is single-process CPU Julia: these entry points replace nothing, they are what an
 * `EnsembleMI355(devices = 0:7)` ensemble algorithm calls after ude_loss_grad_ensemble_dev on every device.
 *   one process per GPU:   ude_comm_unique_id on rank 0 -> share the 128 bytes -> ude_comm_create on every rank
 *   one process, N GPUs:   ude_create per device -> ude_comm_create_local -> ude_allreduce_grad_local (RCCL, grouped)
 *                          or ude_allreduce_grad_p2p (one-shot peer reads over xGMI, fixed rank order: deterministic
 *                          fp64 sum, identical bits on every device)
 * All calls enqueue on the contexts' streams. */
typedef struct ude_comm ude_comm;
int ude_comm_unique_id(char id[128]);
int ude_comm_create(ude_ctx* ctx, int32_t nranks, int32_t rank, const char id[128], ude_comm** out);
int ude_comm_create_local(int32_t ndev, ude_ctx* const* ctxs, ude_comm** out /* ndev */);
void ude_comm_destroy(ude_comm* comm);
/* the tail of the double[np + 4] payload for a host without a device array library: payload_dev[np + 1 .. np + 3] = (sum nf, sum naccept,
 * sum nreject) over the N trajectories of stats_dev (8 x N int64 as the gradient call left them; forward + backward), as doubles; enqueued */
int ude_pack_counters_dev(ude_ctx* ctx, int64_t N, const int64_t* stats_dev, double* payload_dev, int32_t n_param);
int ude_allreduce_grad(ude_comm* comm, double* buf_dev, int64_t n);
int ude_allreduce_grad_local(int32_t ndev, ude_comm* const* comms, double* const* bufs_dev, int64_t n);
int ude_allreduce_grad_p2p(int32_t ndev, ude_comm* const* comms, double* const* bufs_dev, int64_t n);
/* The same one-shot reducer with ONE PROCESS PER GPU (the layout `bench.py --gpus N` and a multi-process Julia host use): every rank
 * creates a communicator with an exchange window in its own HBM and gets a 64-byte IPC handle; the host all-gathers the handles
 * (torch.distributed / MPI / Distributed.jl -- 64 bytes per rank, once) and connects; per gradient ONE kernel per rank: publish the
 * payload, add all ranks' payloads in rank order (peer reads over xGMI, system-scope flags; identical bits on every rank), write the
 * sum back.  n <= n_max <= 16384 doubles.  A peer that never arrives: the result is NaN after UDE_P2P_TIMEOUT_MS (default 5000) and
 * ude_comm_p2p_status counts it -- never a hung GPU.  A timeout is STICKY: the rank stops publishing (its peers fail fast instead of
 * summing a payload of another call), every later call of that communicator yields NaN on the device and, once the host has seen the
 * count through ude_comm_p2p_status, UDE_ERR_TIMEOUT -- re-create the communicator.  Needs no librccl; needs fine-grained device memory
 * for the window (UDE_ERR_UNSUPPORTED otherwise -- there is no coarse-grained fallback).
 * Teardown: ude_comm_p2p_disconnect = stream-ordered handshake (every rank stores a "closed" word into every peer's window, then waits on
 * its own window for all of them: no peer is still reading this rank's window, this rank reads no peer's) + unmap the peers' windows;
 * ude_comm_destroy then frees the window (and runs the disconnect itself if the host did not).  A host with a barrier of its own calls
 * disconnect on every rank, its barrier, then destroy, so that every importer has unmapped before any exporter frees. */
int ude_comm_create_p2p(ude_ctx* ctx, int32_t nranks, int32_t rank, int64_t n_max, char handle_out[64], ude_comm** out);
int ude_comm_p2p_connect(ude_comm* comm, const char* handles /* nranks x 64 bytes, rank order */);
int ude_allreduce_grad_p2p_mp(ude_comm* comm, double* buf_dev, int64_t n);
int ude_comm_p2p_status(ude_comm* comm, int32_t* timeouts);
int ude_comm_p2p_disconnect(ude_comm* comm);

/* Failure accounting of the most recent gradient call on this context (blocks on the context's stream): the number
 * of trajectories whose retcode is not Success.  Such trajectories contribute nothing to the gradient (one exception, loud: in the
 * block-level UDE_SENSE_INTERPOLATING_ADJOINT_FAST mode a BACKWARD failure makes grad_theta NaN, see there) and the
 * ensemble loss is +Inf (the reference's solve would abort / return an Inf loss), so a training loop can never
 * silently optimise a partial objective.  If any of them is a UDE_RET_DENSE_OVERFLOW and max_dense_steps is
 * automatic, *grown is set to 1 and the capacity is multiplied by 4: the caller repeats the call. */
int ude_last_failures(ude_ctx* ctx, const int32_t* retcode_dev, int64_t N, int32_t* nfail, int32_t* grown);

/* device time (ms) of the forward and backward kernels of the most recent call on this context,
 * measured with HIP events on the context's stream (valid after the stream has been synchronised) */
int ude_last_kernel_ms(ude_ctx* ctx, float* fwd_ms, float* bwd_ms);

/* debugging aid for parity work: record (t, dt, EEst, q, accepted) of every step attempt of trajectory
 * `traj` (forward rows [0,cap), backward rows [cap,2cap), 5 doubles each); traj < 0 switches it off */
int ude_set_trace(ude_ctx* ctx, int64_t traj, int32_t cap);
int ude_get_trace(ude_ctx* ctx, double* out_host /* 2*cap*5 */);

/* ARITH-SPEC primitives as evaluated on the device (parity tests): op 0 fastpow(x,y), 1 exp, 2 tanh,
 * 3 log10, 4 10^x, 5 sqrt, 6 x/y, 7 fma(x,y,x), 8 log, 9 x^y,
 * 10 v_mfma_f64_16x16x4 probe: per group of 64 values lane l supplies A[l%16][l/16] = x, B[l/16][l%16] = y, C = 0 and
 *    receives D[l/16][l%16] (n must be a multiple of 64) */
int ude_math_dev(ude_ctx* ctx, int32_t op, int64_t n, const double* x_host, const double* y_host, double* out_host);

/* DiffEqBase.fastpow as evaluated on the device (one thread), for parity tests of the controller */
int ude_fastpow_dev(ude_ctx* ctx, int64_t n, const double* x_host, const double* y_host, double* out_host);

#ifdef __cplusplus
}
#endif
#endif
