/*
 * ude_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the algorithms that execute the reference's hot path.  The reference
 * (ChrisRackauckas/universal_differential_equations) is Julia scripts; the arithmetic lives in
 * un-vendored upstream packages pinned by LotkaVolterra/Manifest.toml:
 *   OrdinaryDiffEq 6.19.2 (:1434)  Tsit5/Vern7 perform_step!, PI controller, initial dt, saveat
 *   DiffEqBase 6.94.4 (:390)       fastpow, default norm, calculate_residuals
 *   DiffEqSensitivity 6.79.0 (:420) InterpolatingAdjoint
 *   Lux 0.4.11 (:1187) / DiffEqFlux FastChain / Flux destructure: Dense layers, parameter layout
 * None of that source is under /root/reference, so this file restates the published algorithms
 *  and is PINNED against the golden data the reference ships in
 * LotkaVolterra/results/ (jld2 files decoded to tests/golden/ (json) by tools/make_golden.py): DEStats triples,
 * saved states, last-step integrator caches, tableaux, loss known-answers and the ADAM loss
 * trajectory (tests/test_oracle_golden.py).  The InterpolatingAdjoint backward pass has NO golden
 * in the reference ("parity unpinned" for step counts of the backward solve); it is pinned only
 * through gradients (ADAM loss history + finite differences).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef UDE_ORACLE_H
#define UDE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UDEO_MAX_LAYERS 8

/* model kinds: reference file:line each one restates */
enum {
    UDEO_KIND_LV_TRUE = 0,   /* lotka!            LotkaVolterra/scenario_1.jl:30-34      theta = (alpha,beta,gamma,delta) */
    UDEO_KIND_LV_UDE = 1,    /* ude_dynamics!     scenario_1.jl:69-73, scenario_2.jl:90-95, hudson_bay.jl:85-91 */
    UDEO_KIND_SEIR_TRUE = 2, /* corona!           SEIR_exposure/seir_exposure.jl:16-30    consts = p_[9] */
    UDEO_KIND_SEIR_UDE = 3,  /* dudt_             seir_exposure.jl:117-130 */
    UDEO_KIND_KPP_TRUE = 4,  /* rc_ode            FisherKPP/Fisher-KPP-CNN.jl:51-63, scenario_3.jl:43-53  consts = D/dx^2, -2D/dx^2, r */
    UDEO_KIND_KPP_UDE = 5,   /* nn_ode            Fisher-KPP-CNN.jl:111-126, scenario_3.jl:103-114 */
    UDEO_KIND_SEIR_NODE = 6  /* dudt_node         SEIR_exposure/seir_exposure.jl:53-66   consts as SEIR (mu, sigma used) */
};
enum { UDEO_ACT_IDENTITY = 0, UDEO_ACT_TANH = 1, UDEO_ACT_RBF = 2, UDEO_ACT_RELU = 3 };
enum { UDEO_ALG_TSIT5 = 0, UDEO_ALG_VERN7 = 1 };
enum { UDEO_SENSE_INTERPOLATING_ADJOINT = 0, UDEO_SENSE_DISCRETE = 1,
       UDEO_SENSE_FAST = 2 /* interpolating adjoint with lambda-only error control (SURVEY 8(b) `fast`): the parameter
                              cotangent rides along as a quadrature on the adjoint's own steps; not an upstream step sequence */,
       UDEO_SENSE_FAST_MM = 4 /* the SAME lambda solve as FAST (identical step counts, dL/du0); only the ASSOCIATION of the parameter
                              cotangent differs: the one the device's block-level matrix-core accumulation executes
                              (csrc/ude_seir_lsf.h) -- every network parameter is one fused chain
                                  mu = fma(-((dt b_s) delta), a, mu)
                              over the stage evaluations in the order they are made, a rejected attempt being taken back by the
                              same chain with the weights negated (its "replay") before the step is repeated.  Kinds whose
                              parameters are all network parameters (SEIR exposure UDE, the neural ODE) */ };
enum { UDEO_RET_SUCCESS = 0, UDEO_RET_MAXITERS = 1, UDEO_RET_DTLESSTHANMIN = 2, UDEO_RET_UNSTABLE = 3 };

/* Same field layout as include/udecore.h:ude_model_desc so one ctypes.Structure serves both. */
typedef struct {
    int32_t kind;
    int32_t dtype;                      /* 0 = f64, 1 = f32 (informational for the oracle: the function suffix decides) */
    int32_t n_state;
    int32_t n_param;                    /* length of theta */
    int32_t n_layers;                   /* number of Dense layers (0 for mechanistic kinds) */
    int32_t dims[UDEO_MAX_LAYERS + 1];  /* dims[0]=in ... dims[n_layers]=out */
    int32_t act[UDEO_MAX_LAYERS];       /* activation of each Dense layer */
    int32_t nn_offset;                  /* theta index of the first NN parameter; per layer [vec(W) col-major (out x in); b] */
    int32_t lin_idx[2];                 /* LV_UDE: theta index of a trainable diagonal coefficient, or -1 */
    int32_t stencil_offset;             /* KPP_UDE: theta index of w1 (w1,w2,w3, then one unused slot) */
    int32_t d0_offset;                  /* KPP_UDE: theta index of D0 */
    int32_t reserved;
    double lin_sign[2];                 /* LV_UDE: du_i = (lin_idx<0 ? lin_const : lin_sign*theta[lin_idx]) * u_i + NN_i(u) */
    double lin_const[2];
    double consts[16];                  /* SEIR: p_[0..8] = F,beta0,alpha,kappa,mu,sigma,gamma,d,lambda; KPP_TRUE: D/dx^2, -2D/dx^2, r */
} udeo_model_desc;

typedef struct {
    int32_t alg;        /* UDEO_ALG_* */
    int32_t maxiters;   /* <=0 -> 100000 (OrdinaryDiffEq default) */
    double abstol;      /* <=0 -> 1e-6 */
    double reltol;      /* <=0 -> 1e-3 */
    double dtmax;       /* <=0 -> |tf - t0| */
    double dt0;         /* >0 -> use as initial dt instead of the Hairer heuristic */
    double qmin, qmax, gamma, qoldinit; /* <=0 -> 0.2, 10, 0.9, 1e-4 */
    double beta1, beta2;                /* <=0 -> 7/(10 order), 2/(5 order) */
    int32_t sensealg;   /* UDEO_SENSE_*: 0 InterpolatingAdjoint (a10), 1 discretise-then-optimise (a9 / N2), 2 fast */
    int32_t reserved;
} udeo_solve_opts;

/* stats layout per trajectory (int64[8]):
 * 0 nf (upstream destats.nf), 1 naccept, 2 nreject, 3 nf_lazy (Vern7 lazy-interpolation evals, not in nf),
 * 4 nf_bwd (augmented adjoint RHS evals), 5 naccept_bwd, 6 nreject_bwd, 7 nf_bwd_lazy (forward lazy stages built for the adjoint) */
#define UDEO_NSTATS 8

/* ---- scalar helpers exposed for unit tests ---- */
float udeo_fastlog2(float x);
float udeo_exp2f(float x);
double udeo_fastpow(double x, double y);
double udeo_exp(double x);   /* ARITH-SPEC deterministic elementary functions (see ude_oracle.c) */
double udeo_tanh(double x);
double udeo_log10(double x);
double udeo_pow10(double y);
double udeo_log(double x);
double udeo_pow(double x, double y);
int udeo_num_params(const udeo_model_desc* m); /* NN parameter count implied by dims */

/* ---- f64 API ---- */
void udeo_rhs_f64(const udeo_model_desc* m, const double* theta, const double* u, double t, double* du);
/* dlam = (df/du)^T lam ; dtheta += (df/dtheta)^T lam */
int udeo_rhs_vjp_f64(const udeo_model_desc* m, const double* theta, const double* u, double t,
                     const double* lam, double* dlam, double* dtheta);

/* forward solve of N trajectories sharing theta.  u0: n x N (column j = trajectory j); tspan: 2 (shared);
 * saveat: ns ascending times in [t0,tf] (shared); u_out: n x ns x N; stats: UDEO_NSTATS x N; retcode: N.
 * nthreads<=1 -> serial. */
int udeo_solve_ensemble_f64(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                            const double* u0, const double* tspan, const double* theta,
                            const double* saveat, int32_t ns, double* u_out, int64_t* stats,
                            int32_t* retcode, int32_t nthreads);

/* interpolating-adjoint VJP: cotangent n x ns x N (dL/du at the save times) -> grad_theta (np, summed over N),
 * grad_u0 (n x N or NULL).  Re-solves forward densely, then integrates [lambda; mu] backward. u_out may be NULL. */
int udeo_vjp_ensemble_f64(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                          const double* u0, const double* tspan, const double* theta,
                          const double* saveat, int32_t ns, const double* cotangent,
                          double* u_out, double* grad_theta, double* grad_u0, int64_t* stats,
                          int32_t* retcode, int32_t nthreads);

/* loss = sum_j sum_i sum_c row_mask[c]*(pred - data)^2 with its adjoint gradient. data: n x ns x N.
 * loss_per_traj may be NULL. */
int udeo_loss_grad_ensemble_f64(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                                const double* u0, const double* tspan, const double* theta,
                                const double* saveat, int32_t ns, const double* data,
                                const uint8_t* row_mask, double* loss, double* loss_per_traj,
                                double* grad_theta, double* grad_u0, double* u_out, int64_t* stats,
                                int32_t* retcode, int32_t nthreads);

/* dense forward solve of ONE trajectory, returning every accepted step (for tests):
 * t_steps: cap+1, u_steps: n x (cap+1), k_steps: n x nk x cap with nk = 7 (Tsit5) / 16 (Vern7).
 * returns number of accepted steps or <0 on error. */
int udeo_solve_dense_f64(const udeo_model_desc* m, const udeo_solve_opts* o, const double* u0,
                         const double* tspan, const double* theta, int32_t cap, double* t_steps,
                         double* u_steps, double* k_steps, int64_t* stats);

/* ---- f32 API (Float32 problems: scenario_3.jl, hudson_bay.jl) ---- */
void udeo_rhs_f32(const udeo_model_desc* m, const float* theta, const float* u, float t, float* du);
int udeo_solve_ensemble_f32(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                            const float* u0, const float* tspan, const float* theta,
                            const float* saveat, int32_t ns, float* u_out, int64_t* stats,
                            int32_t* retcode, int32_t nthreads);

/* Float32 gradients (scenario_3.jl:121-134 and hudson_bay.jl:98-123 train in Float32 with ForwardDiffSensitivity: the
 * discrete sweep; the interpolating adjoint is instantiated too) */
int udeo_vjp_ensemble_f32(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                          const float* u0, const float* tspan, const float* theta,
                          const float* saveat, int32_t ns, const float* cotangent,
                          float* u_out, float* grad_theta, float* grad_u0, int64_t* stats,
                          int32_t* retcode, int32_t nthreads);
int udeo_loss_grad_ensemble_f32(const udeo_model_desc* m, const udeo_solve_opts* o, int64_t N,
                                const float* u0, const float* tspan, const float* theta,
                                const float* saveat, int32_t ns, const float* data,
                                const uint8_t* row_mask, float* loss, float* loss_per_traj,
                                float* grad_theta, float* grad_u0, float* u_out, int64_t* stats,
                                int32_t* retcode, int32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif
