// ude_kernels.h -- fused adaptive Runge-Kutta kernels (a7/a8 forward, a10 interpolating adjoint).
//
// One kernel = the whole solve of a trajectory: initial-dt heuristic, every RK stage with the NN
// right-hand side, scaled error norm, PI step controller (Float32 fastpow), accept/reject, save-point
// interpolation, dense-output store -- no launch or HBM round trip per step.  The same driver
// integrates the forward ODE and, time-reversed, the augmented adjoint system [lambda; mu]:
//   * "replicated" components (the ODE state / lambda, NR of them) live in registers of all G lanes
//   * "slot" components (mu = parameter cotangent, NSL per lane) are dealt to lanes with their neuron;
//     their derivative never depends on their value, so only running b- and btilde-weighted sums of the
//     stage derivatives are kept (no per-stage mu storage).
// Restated from upstream OrdinaryDiffEq 6.19.2 / DiffEqSensitivity 6.79.0 as pinned by the oracle
// (oracle/ude_oracle_impl.h); reference call sites: scenario_1.jl:84,206,
// seir_exposure.jl:138-140, Fisher-KPP-CNN.jl:136.
#pragma once
#include "ude_models.h"
#include "ude_tableaux.h"

namespace ude {

enum { RET_SUCCESS = 0, RET_MAXITERS = 1, RET_DTLESSTHANMIN = 2, RET_UNSTABLE = 3, RET_DENSE_OVERFLOW = 4 };

struct Opts {
    double abstol, reltol, dtmax, dt0, qmin, qmax, gamma, qoldinit, beta1, beta2;
    int32_t maxiters;
};

// the solve options in the kernel's scalar type (oracle: resolve_opts casts every option to REAL)
struct OptsR {
    real abstol, reltol, dtmax, dt0, qmin, qmax, gamma, qoldinit, beta1, beta2;
    int32_t maxiters;
    __device__ OptsR(const Opts& o)
        : abstol((real)o.abstol), reltol((real)o.reltol), dtmax((real)o.dtmax), dt0((real)o.dt0), qmin((real)o.qmin),
          qmax((real)o.qmax), gamma((real)o.gamma), qoldinit((real)o.qoldinit), beta1((real)o.beta1), beta2((real)o.beta2),
          maxiters(o.maxiters) {}
};

template <class T> struct TabDevT;
struct KParams {
    int64_t N;        // trajectories
    int64_t Npad;     // stride of the SoA workspaces
    int32_t ns, cap, n_state, n_param;
    double t0, tf;
    Opts o;
    ModelConsts mc;
    const real* u0;      // n x N
    const real* theta;   // np
    const real* saveat;  // ns
    real* u_out;         // n x ns x N or null
    int64_t* stats;        // 8 x N or null
    int32_t* retcode;      // N
    // dense forward store (replicated states: SoA, field-major [(step*NF + field)*Npad + traj]; distributed states: one contiguous record
    // per (traj, step) -- dense_rec / dense_fs below); null for plain solves
    real* dense;
    int32_t* dense_n;
    // loss / cotangent
    const real* data;       // n x ns x N or null
    const uint8_t* row_mask;  // n or null
    const real* cot_in;     // n x ns x N user cotangent or null
    real* cot;              // SoA [(i*n + c)*Npad + traj] (distributed states: [(traj*ns + i)*n + c]) written by the forward kernel when data != null
    real* loss_traj;        // N
    // backward outputs
    real* grad_part;  // [nwaves_total][np] per-wave partial gradients
    real* slot_glob;  // SLOTS_GLOBAL models: slot state mu in HBM, element c of thread g at slot_glob[c * nthreads + g]
    real* grad_u0;    // n x N or null
    // debugging: per-iteration trace (t, dt, EEst, q, accept) of one trajectory; fwd rows first, then bwd
    real* trace;      // [2][trace_cap][5] or null
    int64_t trace_traj;
    int32_t trace_cap;
    const TabDevT<real>* tab;  // tableau of the algorithm (device memory, in the kernel's scalar type)
    // per-trajectory time grids (kernels instantiated with PT = true): tspan_pt = 2 x N, saveat = ns x N when saveat_pt
    const double* tspan_pt;   // (2 x N doubles, as the host holds them)
    int32_t saveat_pt, dtmax_auto;
    // checkpointed adjoint (UDE_SENSE_INTERPOLATING_ADJOINT_CHECKPOINTED): the forward store keeps (t, t_end, dt, u) of every accepted
    // step and NOT its stage derivatives; the adjoint kernel recomputes them when it enters an interval (AdjSys::RECOMPUTE)
    int32_t ckpt;
    // per-member parameters (UDE_PT_THETA, LotkaVolterra/run_loops.jl:55-62: every ensemble member is its own recovery with its own
    // theta): theta is np x N, member j reads theta + j * theta_pm, and the gradient is returned per member (grad_part = the
    // caller's np x N array, row j written by trajectory j's lanes; no sum over trajectories).  0 = one shared theta.
    int32_t theta_pm;
    // cost-ordered launch of a multi-round ensemble (round 6, SURVEY.md 7 "sort / bucket trajectories by expected cost"): lane group g of
    // the adjoint launch works on trajectory perm[g] (the forward launch is always the identity); the trajectories then write ONE
    // GRADIENT ROW EACH (grad_part = N x np), which the finish kernel adds in TRAJECTORY order -- the sum does not depend on the
    // permutation, two runs give the same bits whatever order the sort's atomics produced.  null: identity, one partial row per wavefront.
    const int32_t* perm;
};

// Workspace layouts.  Replicated states -- the lanes of a wavefront belong to different trajectories -- keep the dense store and the
// cotangent rows as SoA, field-major: field f of step s of trajectory j at (s nf + f) Npad + j (adjacent trajectories adjacent).
// Distributed states -- a PDE's grid dealt over the lanes of its wavefronts -- and component-per-lane models (CPL: one wavefront per
// trajectory, lane c = component c; the lock-step SEIR / neural-ODE kernels fetch a record with the sixteen lanes of a slot's row)
// keep every record contiguous instead: record (j, s) at (j cap + s) nf, field stride 1, so the lanes' components are adjacent
// words (round 4: with the SoA layout every 8-byte component sat in a cache line of its own -- the 6x read amplification of the
// Fisher-KPP adjoint in profiles/r03_pmc_kpp.md; FETCH 6.2 -> 0.44 GB per launch)
template <bool DIST>
__device__ __forceinline__ size_t dense_fs(const KParams& p) { return DIST ? (size_t)1 : (size_t)p.Npad; }
template <bool DIST>
__device__ __forceinline__ real* dense_rec(const KParams& p, int s, int nf, int64_t j) {
    return DIST ? p.dense + ((size_t)j * p.cap + s) * nf : p.dense + ((size_t)s * nf) * p.Npad + j;
}



// ---------------------------------------------------------------------------------------------
// Tableau in memory: the stage loop is a RUNTIME loop (one inlined copy of the right-hand side instead
// of one per stage -- the fully unrolled kernels were ~100 KB of code, beyond the instruction cache), so
// coefficients are read with a wave-uniform index from this table (scalar loads).
// Rows 0..S-1: a_sj; rows S..S+NEXTRA-1: the lazy dense-output stages.
// ARITH-SPEC: every weighted sum is an fma chain in ascending stage order started by the product with the
// (always nonzero) first coefficient; zero coefficients contribute fma(0, k, acc) == acc exactly, so the
// value equals the oracle's skip-zeros chain (oracle/ude_oracle_impl.h: combine()).
// ---------------------------------------------------------------------------------------------
template <class T>
struct TabDevT {
    T A[16][16];
    T B[16], BT[16], C[16];
    T R[16][8];  // dense-output weights b_q(theta): 7-slot Horner tables (Tab::R), lane q of a CPL wavefront loads row q
};
using TabDev = TabDevT<real>;  // (the host uploads both the double and the float table; coefficients rounded to T)

template <class Tab, class T = double>
inline TabDevT<T> make_tabdev() {
    TabDevT<T> t{};
    for (int s = 0; s < Tab::S; ++s) {
        for (int j = 0; j < s; ++j) t.A[s][j] = (T)Tab::A(s, j);
        t.B[s] = (T)Tab::B(s);
        t.BT[s] = (T)Tab::BT(s);
        t.C[s] = (T)Tab::C(s);
    }
    for (int e = 0; e < Tab::NEXTRA; ++e) {
        for (int j = 0; j < Tab::S + e; ++j) t.A[Tab::S + e][j] = (T)Tab::AE(e, j);
        t.C[Tab::S + e] = (T)Tab::CE(e);
    }
    for (int q = 0; q < Tab::NK; ++q)
        for (int i = 0; i < 7; ++i) t.R[q][i] = (T)Tab::R(q, i);
    return t;
}

template <class Tab>
struct RowDense { static constexpr real at(int q) { return Tab::dense_uses(q) ? 1.0 : 0.0; } };

// sum_j v1(j) * v2(j) over the j with Row::at(j) != 0, j ascending (compile-time indices)
template <class Row, int N, class V1, class V2>
__device__ __forceinline__ real chain2(V1 v1, V2 v2) {
    real acc = 0.0;
    bool first = true;
    static_for<0, N>([&](auto j) {
        constexpr int jj = decltype(j)::value;
        if constexpr (Row::at(jj) != real(0)) {
            acc = first ? v1(j) * v2(j) : rfma(v1(j), v2(j), acc);
            first = false;
        }
    });
    return acc;
}

// ---------------------------------------------------------------------------------------------
// generic driver
// ---------------------------------------------------------------------------------------------
// Stage storage of a REPLICATED state: one LDS word per (stage, component) and lane group -- or per wavefront when
// a trajectory spans several (every wavefront keeps a private copy: no cross-wave ordering needed)
// Component-per-lane (CPL) systems keep ONE component per lane (lane c <-> component c, c < KCP - 1; the other lanes
// share the spare column KCP - 1 and only ever write zeros there): stage j of a wavefront at [wave][j][KCP].
constexpr int KCP = 8;
template <bool DIST, int G, int BLOCK, bool CPL = false>
constexpr int k_stride() { return CPL ? KCP : DIST ? BLOCK : (G > 64 ? BLOCK / 64 : BLOCK / G); }
template <bool DIST, int G, bool CPL = false>
__device__ __forceinline__ int k_offset(int nk = 0) {
    if constexpr (CPL) {
        const int lane = threadIdx.x & 63;
        return (threadIdx.x >> 6) * nk * KCP + (lane < KCP - 1 ? lane : KCP - 1);
    } else {
        return DIST ? threadIdx.x : (G > 64 ? threadIdx.x / 64 : threadIdx.x / G);
    }
}
// row of the partial-gradient matrix this thread's wavefront (or multi-wave trajectory) reports into
template <int G, int BLOCK>
__device__ __forceinline__ int64_t part_row() {
    if constexpr (G > 64) return (int64_t)blockIdx.x;
    else return BLOCK >= 64 ? ((int64_t)blockIdx.x * BLOCK + threadIdx.x) / 64 : (int64_t)blockIdx.x;
}

// (see Driver::run: a system whose tableau loads go through the constant address space)
template <class S, class = void> struct sys_tab_scalar { static constexpr bool v = false; };
template <class S> struct sys_tab_scalar<S, std::void_t<decltype(S::TAB_SCALAR)>> { static constexpr bool v = S::TAB_SCALAR; };

template <class Tab, class Sys, int G, int BLOCK>
struct Driver {
    static constexpr int NR = Sys::NR, NSL = Sys::NSL, NSLA = NSL > 0 ? NSL : 1;
    static constexpr int S = Tab::S, NK = Tab::NK;
    static constexpr bool USE_FSAL = Tab::FSAL && !Sys::ALWAYS_K0;
    static constexpr bool SLOT_FSAL = USE_FSAL && NSL > 0;  // stage-0 slot derivative handed over in LDS
    // deferred slots: the system keeps per-stage factors and forms the slot sums / error norm / candidate at step end
    static constexpr bool DEFER = Sys::DEFERRED;
    // `fast` adjoint mode (UDE_SENSE_FAST): only the replicated state (lambda) is under error control; the slot state (the
    // parameter cotangent) is a quadrature carried along on the accepted steps -- no error accumulators, no divisions
    static constexpr bool FAST = Sys::FAST;
    static_assert(!DEFER || (NSL == 0 && !USE_FSAL), "deferred slots: the system owns the slot state");
    // stage derivatives of a REPLICATED state are stored once per group (all lanes read/write the same word)
    // CPL: stage derivatives stored one component per lane (lane c <-> component c): the weighted stage sums are
    // formed by that lane alone and broadcast with v_readlane -- NR times fewer LDS reads and fma's per lane
    static constexpr bool CPL = Sys::CPL;
    static_assert(!CPL || (G == 64 && !Sys::STATE_DISTRIBUTED && NR <= 64), "component-per-lane: one wavefront per trajectory");
    static constexpr int KSTRIDE = k_stride<Sys::STATE_DISTRIBUTED, G, BLOCK, CPL>();
    // CPL systems run one wavefront per trajectory with the step-size state replicated in every lane: all 64 lanes take the
    // same branches by construction.  Saying so -- a scalar condition -- keeps the compiler from lowering the control flow
    // to EXEC-masked regions: cheaper (scalar branches, no mask bookkeeping in spilled SGPR pairs), and the masked lowering is
    // where, under the register pressure of the neural-ODE kernels, a compiler-inserted VGPR->AGPR copy ended up in front of
    // the EXEC restore of a join block and kept stale lanes
    static constexpr bool UNI = CPL || G >= 64;  // (G >= 64: a wavefront never holds lanes of two trajectories)
    static __device__ __forceinline__ bool uni(bool b) {
        if constexpr (UNI) return __builtin_amdgcn_readfirstlane((int)b) != 0; else return b;
    }

    struct Stats {
        int64_t nf = 0, nacc = 0, nrej = 0, nlazy = 0;
    };

    // Per-thread LDS arrays (element i of this thread at base[i * BLOCK]: conflict-free):
    //   kl: stage derivatives of the replicated part, k(j, c) = kl[(j*NR + c) * BLOCK]
    //   mu: slot state (touched once per step)
    // z: replicated state (registers).  Integrates from t0 along tdir through sys' tstops.
    static __device__ __forceinline__ int run(Sys& sys, const Opts& oin, const TabDev* __restrict__ tab, real (&z)[NR],
                                              real* kl, real* mu, real t0, real tdir, real ntot, Stats& st,
                                              real* gtmp = nullptr, real* gtmp2 = nullptr, int mustride = BLOCK) {
        const int MS = Sys::SLOTS_GLOBAL ? mustride : BLOCK;  // element c of this thread's slot column at mu[c * MS]
        // gtmp: per-thread LDS row (element c at gtmp[c * BLOCK]) holding the slot derivative of stage 0: parked there
        // by the initial-dt heuristic, and -- FSAL tableaux -- handed over from the last stage of an accepted step
        // (gtmp2 receives the last stage's slot derivative; the two rows swap on acceptance)
        const OptsR o(oin);
        // Systems that ask for it (Sys::TAB_SCALAR: the adjoint of the LV models with register-resident weights) read the tableau through the
        // CONSTANT address space: uniform loads of it are then scalar loads into SGPRs (s_load, operands of the vector instructions as they are)
        // instead of vector loads of one address by 64 lanes -- the kernels store to global memory, so the compiler cannot prove a plain global
        // load invariant.  The table is written by the host before the launch and never by a kernel.  Measured per kernel family (round 6,
        // profiles/r06_probes.md): configs[1] adj_kernel -1.5 %, 40 000 members -2.7 %, the 8-lane run-time shapes -3 %; the 2-32-2 net on 16 lanes
        // +2.4 %, the forward kernels and Fisher-KPP unchanged -- so it is a property of the system, not of the driver.
        typedef std::conditional_t<sys_tab_scalar<Sys>::v, const __attribute__((address_space(4))) TabDev, const TabDev> CTabDev;
        CTabDev* const ct = (CTabDev*)tab;
        real accb[NSLA], acce[NSLA];
        const real dtmax = sys.dtmax(o);  // (per-trajectory when the time grids are)
        real t = t0, dt, qold = o.qoldinit, q11 = real(1);
        bool accept = true, done = false;
        int iter = 0, ret = RET_SUCCESS;
        real tstop = sys.first_tstop();
        auto K = [&](int j, int c) -> real& { return kl[(j * NR + c) * KSTRIDE]; };
        auto K1 = [&](int j) -> real& { return kl[j * KCP]; };  // CPL: this lane's component of stage j

        // ---- initial dt (ode_determine_initdt), 2 evals ----
        if (o.dt0 > real(0)) {
            dt = tdir * o.dt0;
            if constexpr (USE_FSAL) {
                real kr[NR], gs[NSLA];
                sys.eval(t, z, kr, gs);
                if constexpr (CPL) K1(0) = own_of(kr);
                else static_for<0, NR>([&](auto c) { K(0, c) = kr[c]; });
                if constexpr (SLOT_FSAL) static_for<0, NSL>([&](auto c) { gtmp[c * BLOCK] = gs[c]; });
            }
            if constexpr (Tab::FSAL) st.nf += 1;
        } else {
            real f0[NR], gs0[NSLA], f1[NR], gs1[NSLA], z1[NR];
            if constexpr (DEFER) sys.eval_store(t, z, f0, 0);
            else sys.eval(t, z, f0, gs0);
            if constexpr (CPL) K1(0) = own_of(f0);
            else static_for<0, NR>([&](auto c) { K(0, c) = f0[c]; });
            // norms in real-real: slots first (lane-parallel), then the replicated components once
            real h0 = 0.0, l0 = 0.0, h1 = 0.0, l1 = 0.0;
            if constexpr (DEFER && !FAST) {
                sys.slot_init01(o, h0, l0, h1, l1);
                group_dd_sum<G>(h0, l0);
                group_dd_sum<G>(h1, l1);
            } else if constexpr (NSL > 0 && FAST) {
                static_for<0, NSL>([&](auto c) { gtmp[c * BLOCK] = gs0[c]; });  // (FSAL hand-over only)
                asm volatile("" ::: "memory");
            } else if constexpr (NSL > 0) {
                static_for<0, NSL>([&](auto c) {
                    const real m = mu[c * BLOCK];
                    const real sk = rfma(rabs(m), o.reltol, o.abstol);
                    const real q0 = m / sk, q1 = gs0[c] / sk;
                    dd_acc(h0, l0, q0 * q0);
                    dd_acc(h1, l1, q1 * q1);
                    gtmp[c * BLOCK] = gs0[c];
                });
                group_dd_sum<G>(h0, l0);
                group_dd_sum<G>(h1, l1);
                asm volatile("" ::: "memory");
            }
            if constexpr (Sys::STATE_DISTRIBUTED) {
                real hs0 = 0.0, ls0 = 0.0, hs1 = 0.0, ls1 = 0.0;
                static_for<0, NR>([&](auto c) {
                    const real on = sys.state_on(c);
                    const real sk = rfma(rabs(z[c]), o.reltol, o.abstol);
                    const real q0 = on * (z[c] / sk), q1 = on * (f0[c] / sk);
                    dd_acc(hs0, ls0, q0 * q0);
                    dd_acc(hs1, ls1, q1 * q1);
                });
                group_dd_sum<G>(hs0, ls0);
                group_dd_sum<G>(hs1, ls1);
                dd_acc(h0, l0, hs0); dd_acc(h0, l0, ls0);
                dd_acc(h1, l1, hs1); dd_acc(h1, l1, ls1);
            } else {
                static_for<0, NR>([&](auto c) {
                    const real sk = rfma(rabs(z[c]), o.reltol, o.abstol);
                    const real q0 = z[c] / sk, q1 = f0[c] / sk;
                    dd_acc(h0, l0, q0 * q0);
                    dd_acc(h1, l1, q1 * q1);
                });
            }
            const real s0 = h0 + l0, s1 = h1 + l1;
            const real d0 = rsqrt_ieee(s0 / ntot), d1 = rsqrt_ieee(s1 / ntot);
            if (uni(d1 != d1)) {
                ret = RET_UNSTABLE;
                done = true;
            }
            real dt0 = (d0 < real(1e-5) || d1 < real(1e-5)) ? real(1e-6) : (d0 / d1) / real(100);
            if (dt0 > dtmax) dt0 = dtmax;
            if (uni(dt0 < real(10) * REAL_EPS)) {
                dt = tdir * real(1e-6);
            } else {
                const real dt0t = tdir * dt0;
                static_for<0, NR>([&](auto c) { z1[c] = rfma(dt0t, f0[c], z[c]); });
                // (the slot part of u1 does not enter f: mu' is independent of mu)
                if constexpr (DEFER) sys.eval_store(t + dt0t, z1, f1, 1);
                else sys.eval(t + dt0t, z1, f1, gs1);
                real h2 = 0.0, l2 = 0.0;
                if constexpr (DEFER && !FAST) {
                    sys.slot_init2(o, h2, l2);
                    group_dd_sum<G>(h2, l2);
                } else if constexpr (NSL > 0 && !FAST) {
                    static_for<0, NSL>([&](auto c) {
                        const real sk = rfma(rabs(mu[c * BLOCK]), o.reltol, o.abstol);
                        const real q = (gs1[c] - gtmp[c * BLOCK]) / sk;
                        dd_acc(h2, l2, q * q);
                    });
                    group_dd_sum<G>(h2, l2);
                }
                if constexpr (Sys::STATE_DISTRIBUTED) {
                    real hs = 0.0, ls = 0.0;
                    static_for<0, NR>([&](auto c) {
                        const real sk = rfma(rabs(z[c]), o.reltol, o.abstol);
                        const real q = sys.state_on(c) * ((f1[c] - f0[c]) / sk);
                        dd_acc(hs, ls, q * q);
                    });
                    group_dd_sum<G>(hs, ls);
                    dd_acc(h2, l2, hs); dd_acc(h2, l2, ls);
                } else {
                    static_for<0, NR>([&](auto c) {
                        const real sk = rfma(rabs(z[c]), o.reltol, o.abstol);
                        const real q = (f1[c] - f0[c]) / sk;
                        dd_acc(h2, l2, q * q);
                    });
                }
                const real s2 = h2 + l2;
                const real d2 = rsqrt_ieee(s2 / ntot) / dt0;
                const real mx = d1 > d2 ? d1 : d2;
                real dt1;
                if (uni(mx <= real(1e-15))) {
                    dt1 = dt0 * real(1e-3);
                    if (dt1 < real(1e-6)) dt1 = real(1e-6);
                } else {
                    const real ex = -(real(2) + rlog10(mx)) / (real)Tab::ORDER;
                    dt1 = rpow10(ex);
                }
                real d = real(100) * dt0;
                if (dt1 < d) d = dt1;
                if (dtmax < d) d = dtmax;
                dt = tdir * d;
            }
            st.nf += 2;
            if constexpr (Tab::FSAL) st.nf += 1;  // initialize!: fsalfirst = f(u0) (same value, reused)
        }

        while (uni(!done)) {
            // ---- loopheader! ----
            if (uni(iter > 0 && !accept)) {  // step_reject_controller!
                real den = q11 / o.gamma;
                const real iq = real(1) / o.qmin;
                if (iq < den) den = iq;
                dt = dt / den;
            }
            iter += 1;
            if (rabs(dt) > dtmax) dt = tdir * dtmax;
            {
                const real rem = rabs(tstop - t);  // modify_dt_for_tstops!
                if (rabs(dt) > rem) dt = tdir * rem;
            }
            {   // the three ways a solve stops here, behind ONE test (checked in upstream's order)
                const bool b_it = iter > o.maxiters, b_nan = dt != dt;
                const bool b_min = rabs(dt) <= REAL_EPS * rabs(t) && rabs(dt) < rabs(tstop - t);
                if (uni(b_it || b_nan || b_min)) { ret = b_it ? RET_MAXITERS : b_nan ? RET_UNSTABLE : RET_DTLESSTHANMIN; break; }
            }

            // ---- perform_step!: runtime stage loop (wave-uniform s) ----
            real znew[NR];
            [[maybe_unused]] const real zo = CPL ? own_of(z) : real(0);  // this lane's component of z
            if constexpr (SLOT_FSAL) {
                const real bs = ct->B[0], es = ct->BT[0];
                static_for<0, NSL>([&](auto c) {
                    const real g0 = gtmp[c * BLOCK];
                    accb[c] = bs * g0;
                    if constexpr (!FAST) acce[c] = es * g0;
                });
            }
            for (int s = USE_FSAL ? 1 : 0; s < S; ++s) {
                real zs[NR], kr[NR], gs[NSLA];
                if (s == 0) {
                    static_for<0, NR>([&](auto c) { zs[c] = z[c]; });
                } else if constexpr (CPL) {
                    real acc = ct->A[s][0] * K1(0);
#pragma unroll 4
                    for (int j = 1; j < s; ++j) acc = rfma(ct->A[s][j], K1(j), acc);  // (a_sj = 0 for j >= s)
                    bcast_all(rfma(dt, acc, zo), zs);
                } else {
                    static_for<0, NR>([&](auto c) {
                        // all S-1 possible terms, unrolled: the coefficients of stages >= s are zero in the table and the
                        // (zero-initialised, always finite) k storage makes fma(0, k, acc) == acc exact -- one batch of
                        // scalar + LDS loads and one wait per stage instead of a load-wait-fma round trip per term
                        real acc = ct->A[s][0] * K(0, c);
                        static_for<1, S - 1>([&](auto j) { acc = rfma(ct->A[s][j], K(j, c), acc); });
                        zs[c] = rfma(dt, acc, z[c]);
                    });
                }
                if (Tab::FSAL && s == S - 1) static_for<0, NR>([&](auto c) { znew[c] = zs[c]; });
                if constexpr (Sys::ACT_CACHE) sys.store_hint = (s + 1 < S) && (ct->C[s + 1] == ct->C[s]);   // (false again after the last stage)
                if constexpr (DEFER) sys.eval_store(t + ct->C[s] * dt, zs, kr, s);
                else sys.eval(t + ct->C[s] * dt, zs, kr, gs);
                if constexpr (CPL) K1(s) = own_of(kr);
                else static_for<0, NR>([&](auto c) { K(s, c) = kr[c]; });
                if constexpr (NSL > 0) {
                    const real bs = ct->B[s], es = ct->BT[s];
                    if (s == 0) {
                        static_for<0, NSL>([&](auto c) {
                            accb[c] = bs * gs[c];
                            if constexpr (!FAST) acce[c] = es * gs[c];
                        });
                    } else {
                        static_for<0, NSL>([&](auto c) {
                            accb[c] = rfma(bs, gs[c], accb[c]);
                            if constexpr (!FAST) acce[c] = rfma(es, gs[c], acce[c]);
                        });
                    }
                    if constexpr (SLOT_FSAL) {
                        if (s == S - 1) static_for<0, NSL>([&](auto c) { gtmp2[c * BLOCK] = gs[c]; });
                    }
                }
            }
            st.nf += Tab::FSAL ? S - 1 : S;
            if constexpr (!Tab::FSAL && CPL) {
                real acc = ct->B[0] * K1(0);
#pragma unroll 3
                for (int j = 1; j < S; ++j) acc = rfma(ct->B[j], K1(j), acc);
                bcast_all(rfma(dt, acc, zo), znew);
            } else if constexpr (!Tab::FSAL) {
                static_for<0, NR>([&](auto c) {
                    real acc = ct->B[0] * K(0, c);
                    for (int j = 1; j < S; ++j) acc = rfma(ct->B[j], K(j, c), acc);
                    znew[c] = rfma(dt, acc, z[c]);
                });
            }
            // calculate_residuals + ODE_DEFAULT_NORM
            acc_t ss = 0.0;  // (ude_real.h: Float64 accumulation for both scalar types)
            if constexpr (CPL) {
                real acc = ct->BT[0] * K1(0);
#pragma unroll 3
                for (int j = 1; j < S; ++j) acc = rfma(ct->BT[j], K1(j), acc);
                const real a0 = rabs(zo), a1 = rabs(own_of(znew));
                real res[NR];
                bcast_all((dt * acc) / rfma((a0 > a1 ? a0 : a1), o.reltol, o.abstol), res);
                static_for<0, NR>([&](auto c) { ss = afma(res[c], res[c], ss); });
            } else
            static_for<0, NR>([&](auto c) {
                real acc = ct->BT[0] * K(0, c);
                for (int j = 1; j < S; ++j) acc = rfma(ct->BT[j], K(j, c), acc);
                const real a0 = rabs(z[c]), a1 = rabs(znew[c]);
                real res = (dt * acc) / rfma((a0 > a1 ? a0 : a1), o.reltol, o.abstol);
                if constexpr (Sys::STATE_DISTRIBUTED) res *= sys.state_on(c);
                ss = afma(res, res, ss);
            });
            if constexpr (Sys::STATE_DISTRIBUTED) ss = group_sum<G>(ss);
            if constexpr (DEFER && !FAST) ss += group_sum<G>((acc_t)sys.slot_step(dt, tab, o));
            if constexpr (NSL > 0 && !FAST) {
                acc_t ps = 0.0;
                static_for<0, NSL>([&](auto c) {
                    const real m0 = mu[c * MS];
                    const real m1 = rfma(dt, accb[c], m0);
                    accb[c] = m1;  // candidate new value
                    const real a0 = rabs(m0), a1 = rabs(m1);
                    const real res = (dt * acce[c]) / rfma((a0 > a1 ? a0 : a1), o.reltol, o.abstol);
                    ps = afma(res, res, ps);
                });
                ss += group_sum<G>(ps);
            }
            const real EEst = rsqrt_ieee((real)ss / ntot);

            // ---- loopfooter!: PIController ----
            real q;
            {   // (branch-free: EEst == 0 selects 1 / qmax and leaves q11 alone; fastpow has no special-case branches and runs for every
                //  argument -- in the lane-group kernels an `if` here is an EXEC-masked region per step attempt)
                const bool ez = EEst == real(0);
                const real q11n = (real)fastpow((double)EEst, (double)o.beta1);
                real qn = q11n / (real)fastpow((double)qold, (double)o.beta2);
                qn = qn / o.gamma;
                const real lo = real(1) / o.qmax, hi = real(1) / o.qmin;
                if (qn > hi) qn = hi;
                if (qn < lo) qn = lo;
                q = ez ? lo : qn;
                q11 = ez ? q11 : q11n;
            }
            accept = uni(EEst <= real(1));
            sys.trace(iter, t, dt, EEst, q, accept);
            if (accept) {
                st.nacc += 1;
                qold = EEst > o.qoldinit ? EEst : o.qoldinit;
                real dtnew = dt / q;
                const real tprev = t;
                const real ttmp = t + dt;
                {
                    const real mxt = t > tstop ? t : tstop;
                    t = rabs(ttmp - tstop) < real(100) * ulp_of(mxt) ? tstop : ttmp;
                }
                if (rabs(dtnew) > dtmax) dtnew = tdir * dtmax;
                // hook: saveat interpolation / dense store (forward); may build the lazy stages
                {
                    bool lazy_done = false;
                    auto lazy = [&]() {
                        if constexpr (Tab::NEXTRA > 0) {
                            if (!lazy_done) {
                                for (int e = 0; e < Tab::NEXTRA; ++e) {
                                    real zs[NR], kr[NR], gs[NSLA];
                                    const int row = S + e;
                                    if constexpr (CPL) {
                                        real acc = ct->A[row][0] * K1(0);
#pragma unroll 3
                                        for (int j = 1; j < row; ++j) acc = rfma(ct->A[row][j], K1(j), acc);
                                        bcast_all(rfma(dt, acc, zo), zs);
                                    } else
                                    static_for<0, NR>([&](auto c) {
                                        real acc = ct->A[row][0] * K(0, c);
                                        for (int j = 1; j < row; ++j) acc = rfma(ct->A[row][j], K(j, c), acc);
                                        zs[c] = rfma(dt, acc, z[c]);
                                    });
                                    sys.eval(tprev + ct->C[row] * dt, zs, kr, gs);
                                    if constexpr (CPL) K1(row) = own_of(kr);
                                    else static_for<0, NR>([&](auto c) { K(row, c) = kr[c]; });
                                }
                                lazy_done = true;
                                st.nlazy += Tab::NEXTRA;
                            }
                        }
                    };
                    const int hr = sys.accepted(tprev, t, dt, z, znew, kl, lazy);
                    if (uni(hr != RET_SUCCESS)) { ret = hr; done = true; }
                }
                if constexpr (FAST) {  // (with the step size this step USED)
                    static_for<0, NSL>([&](auto c) { mu[c * MS] = rfma(dt, accb[c], mu[c * MS]); });  // the same fma as the candidate of the parity mode
                    if constexpr (DEFER) sys.slot_commit(dt, tab);
                }
                dt = dtnew;
                bool bad = false;
                static_for<0, NR>([&](auto c) {
                    z[c] = znew[c];
                    bad = bad || (znew[c] != znew[c]);
                });
                if constexpr (Sys::STATE_DISTRIBUTED) {  // same decision on every lane of the trajectory
                    if constexpr (G > 64) bad = __syncthreads_or(bad); else bad = __any(bad);
                }
                if constexpr (!FAST) {
                    static_for<0, NSL>([&](auto c) { mu[c * MS] = accb[c]; });
                    if constexpr (DEFER) sys.slot_accept();
                }
                if constexpr (USE_FSAL && CPL) K1(0) = K1(S - 1);
                else if constexpr (USE_FSAL) static_for<0, NR>([&](auto c) { K(0, c) = K(S - 1, c); });
                if constexpr (SLOT_FSAL) { real* tsw = gtmp; gtmp = gtmp2; gtmp2 = tsw; }
                if (uni(bad)) { ret = RET_UNSTABLE; done = true; }
                if (uni(t == tstop)) {  // handle_tstop! + callbacks
                    const bool modified = uni(sys.at_tstop(t, z));
                    bool more = uni(sys.next_tstop(tstop));
                    if (!more) done = true;
                    else if (modified) {
                        if constexpr (Tab::FSAL) st.nf += 1;  // reset_fsal! after u_modified!
                        if constexpr (USE_FSAL) {
                            real kr[NR], gs[NSLA];
                            sys.eval(t, z, kr, gs);
                            if constexpr (CPL) K1(0) = own_of(kr);
                            else static_for<0, NR>([&](auto c) { K(0, c) = kr[c]; });
                            if constexpr (SLOT_FSAL) static_for<0, NSL>([&](auto c) { gtmp[c * BLOCK] = gs[c]; });
                        }
                    }
                }
            } else {
                st.nrej += 1;
                if (uni(EEst != EEst)) { ret = RET_UNSTABLE; done = true; }
            }
        }
        return ret;
    }
};

// ---------------------------------------------------------------------------------------------
// forward system: model RHS + saveat (savevalues!) + dense store + loss/cotangent
// dense field layout per step: 0 t_start, 1 t_end, 2 dt (the step size used), 3..3+n u_start, then k[q][c]
// ---------------------------------------------------------------------------------------------
// time grid of a trajectory: shared (kernel parameters, wave-uniform scalars) or its own (PT: registers)
template <bool PT>
struct TimeGrid {
    real t0_, tf_, dtmax_;
    const real* sv_;
    __device__ __forceinline__ void init(const KParams& p, int64_t j) {
        if constexpr (PT) {
            t0_ = p.tspan_pt ? (real)p.tspan_pt[2 * j] : (real)p.t0;   // (tspan pairs arrive as doubles from the host)
            tf_ = p.tspan_pt ? (real)p.tspan_pt[2 * j + 1] : (real)p.tf;
            dtmax_ = p.dtmax_auto ? tf_ - t0_ : (real)p.o.dtmax;
            sv_ = p.saveat + (p.saveat_pt ? (size_t)j * p.ns : 0);
        }
    }
    __device__ __forceinline__ real T0(const KParams& p) const { if constexpr (PT) return t0_; else return (real)p.t0; }
    __device__ __forceinline__ real TF(const KParams& p) const { if constexpr (PT) return tf_; else return (real)p.tf; }
    __device__ __forceinline__ real SV(const KParams& p, int i) const { if constexpr (PT) return sv_[i]; else return p.saveat[i]; }
    __device__ __forceinline__ real DTMAX(const OptsR& o) const { if constexpr (PT) return dtmax_; else return o.dtmax; }
};

template <bool SORTED>
__device__ __forceinline__ int64_t member_of(const KParams& p, int64_t gslot) {
    if constexpr (SORTED) return (int64_t)p.perm[gslot]; else return gslot;
}

// SORTED (cost-ordered launch, KParams::perm): `j` is the MEMBER the lane group works on (inputs and outputs are indexed by it), `jw` the
// column of the internal workspaces (dense store, cotangent rows, step counts) = the lane group's own position: adjacent groups, adjacent words
// Models that can hand the activations of one adjoint evaluation to the next one AT THE SAME TIME (round 6): the network input of an adjoint
// evaluation is the interpolated forward state u(t) -- a function of t alone -- and both tableaux end with two stages at t + dt (Tsit5: c6 = c7 = 1,
// Vern7: c9 = c10 = 1; the evaluation after a save-time jump is at that time once more).  A model with ACT_CACHE_WORDS (words per thread of an HBM
// row it owns: KParams::slot_glob, unused by register-slot models) is told when to store (the next stage has the same c) and when to load
// (t equals the stored evaluation's t BIT FOR BIT: same t, same interval, same interpolant, same activations -- nothing is approximated).
// ... and models whose activations fit the registers keep the LAST evaluation's there (Model::ActCache, Model::vjp_c): an evaluation at the same
// time as the one before it (per lane group) skips its forward pass (LvUde with register-resident weights, ude_models.h).
template <class M, class = void> struct act_reg { static constexpr bool v = false; struct type {}; };
template <class M> struct act_reg<M, std::void_t<typename M::ActCache>> { static constexpr bool v = true; using type = typename M::ActCache; };
template <class M, class = void> struct act_cache { static constexpr int v = 0; };
template <class M> struct act_cache<M, std::void_t<decltype(M::ACT_CACHE_WORDS)>> { static constexpr int v = M::ACT_CACHE_WORDS; };
// (the modes ACT_NONE / ACT_STORE / ACT_LOAD: ude_model_kpp_vec.h, in front of the model that implements them)

template <class Model, class Tab, int G, int BLOCKDIM, bool PT = false, bool SORTED = false>
struct FwdSys {
    static constexpr bool ACT_CACHE = false;
    TimeGrid<PT> tg;
    __device__ __forceinline__ real dtmax(const OptsR& o) const { return tg.DTMAX(o); }
    static constexpr int NR = Model::NS, NSL = 0;
    static constexpr bool ALWAYS_K0 = false, FAST = false, STATE_DISTRIBUTED = Model::STATE_DISTRIBUTED;
    static constexpr bool SLOTS_GLOBAL = false, CPL = Model::CPL, DEFERRED = false;
    static __device__ __forceinline__ bool uni(bool b) {  // (Driver::uni)
        if constexpr (CPL || G >= 64) return __builtin_amdgcn_readfirstlane((int)b) != 0; else return b;
    }
    typename Model::Ctx mctx;
    const KParams* p;
    int64_t j;      // trajectory
    int64_t jw;     // (SORTED) its workspace column
    __device__ __forceinline__ int64_t wcol() const { if constexpr (SORTED) return jw; else return j; }
    bool writer;    // lane 0 of the group
    int si, nsteps, r, n;
    real loss;
    // component c of this lane is state index comp(c); replicated states: every lane holds all of them
    __device__ __forceinline__ int comp(int c) const {
        if constexpr (STATE_DISTRIBUTED) return Model::point(c, r); else return c;
    }
    __device__ __forceinline__ bool cvalid(int c) const { return comp(c) < n; }
    __device__ __forceinline__ bool cwrite(int c) const { return STATE_DISTRIBUTED ? cvalid(c) : writer; }
    __device__ __forceinline__ real state_on(int c) const { return cvalid(c) ? 1.0 : 0.0; }

    __device__ __forceinline__ real first_tstop() const { return tg.TF(*p); }
    __device__ __forceinline__ bool next_tstop(real&) const { return false; }
    __device__ __forceinline__ bool at_tstop(real, real*) const { return false; }
    __device__ __forceinline__ void eval(real, const real* z, real* kr, real*) {
        asm volatile("" ::: "memory");  // keep the LDS-staged weights in LDS (no hoisting into registers)
        Model::rhs(mctx, z, kr);
    }
    __device__ __forceinline__ void trace(int iter, real t, real dt, real e, real q, bool acc) const {
        if (p->trace && writer && j == p->trace_traj && iter <= p->trace_cap) {
            real* row = p->trace + (size_t)(iter - 1) * 5;
            row[0] = t; row[1] = dt; row[2] = e; row[3] = q; row[4] = acc ? 1.0 : 0.0;
        }
    }

    __device__ __forceinline__ void save_point(int i, const real* v) {
        if (p->u_out) {
            real* dst = p->u_out + ((size_t)j * p->ns + i) * n;
            static_for<0, NR>([&](auto c) { if (cwrite(c)) dst[comp(c)] = v[c]; });
        }
        if (p->data) {
            const real* d = p->data + ((size_t)j * p->ns + i) * n;
            static_for<0, NR>([&](auto c) {
                if (cvalid(c)) {
                    const int ci = comp(c);
                    // masked-out rows are ignored entirely (a select, not a product: NaN/Inf data there must not
                    // reach the loss -- the reference slices those rows away, seir_exposure.jl:146)
                    const real e = (p->row_mask && !p->row_mask[ci]) ? real(0) : (v[c] - d[ci]);
                    loss = rfma(e, e, loss);
                    if (cwrite(c)) p->cot[STATE_DISTRIBUTED ? ((size_t)j * p->ns + i) * n + ci : ((size_t)i * n + ci) * p->Npad + wcol()] = real(2) * e;
                }
            });
        }
    }

    template <class Lazy>
    __device__ __forceinline__ int accepted(real tprev, real t, real dt, const real* z, const real* znew,
                                            const real* kl, Lazy& lazy) {
        auto k = [&](int q, int c) { return kl[(q * NR + c) * k_stride<STATE_DISTRIBUTED, G, BLOCKDIM>()]; };
        auto k1 = [&](int q) { return kl[q * KCP]; };  // CPL: this lane's component
        while (uni(si < p->ns && tg.SV(*p, si) <= t)) {
            const real curt = tg.SV(*p, si);
            if (uni(curt != t)) {
                lazy();
                const real th = (curt - tprev) / dt;
                real b[Tab::NK], y[NR];
                Tab::bth(th, b);
                if constexpr (CPL) {
                    const real acc = chain2<RowDense<Tab>, Tab::NK>([&](auto q) { return k1(q); }, [&](auto q) { return b[q]; });
                    bcast_all(rfma(dt, acc, own_of<NR>(reinterpret_cast<const real(&)[NR]>(*z))), y);
                } else
                static_for<0, NR>([&](auto c) {
                    const real acc = chain2<RowDense<Tab>, Tab::NK>([&](auto q) { return k(q, c); }, [&](auto q) { return b[q]; });
                    y[c] = rfma(dt, acc, z[c]);
                });
                save_point(si, y);
            } else {
                save_point(si, znew);
            }
            si += 1;
        }
        if (p->dense) {
            if (uni(nsteps >= p->cap)) return RET_DENSE_OVERFLOW;
            if (!p->ckpt) lazy();
            {
                const int nf = p->ckpt ? 3 + n : 3 + n + Tab::NK * n;
                const size_t DFS = dense_fs<STATE_DISTRIBUTED || CPL>(*p);
                real* base = dense_rec<STATE_DISTRIBUTED || CPL>(*p, nsteps, nf, wcol());
                if (writer) {
                    base[0] = tprev;
                    base[(size_t)1 * DFS] = t;
                    base[(size_t)2 * DFS] = dt;
                }
                if constexpr (CPL) {
                    // lane c stores component c of u and of every stage
                    const real zo = own_of<NR>(reinterpret_cast<const real(&)[NR]>(*z));
                    if (r < n) {
                        base[(size_t)(3 + r) * DFS] = zo;
                        if (!p->ckpt) static_for<0, Tab::NK>([&](auto q) { base[(size_t)(3 + n + q * n + r) * DFS] = k1(q); });
                    }
                } else {
                static_for<0, NR>([&](auto c) { if (cwrite(c)) base[(size_t)(3 + comp(c)) * DFS] = z[c]; });
                if (!p->ckpt)
                static_for<0, Tab::NK>([&](auto q) {
                    // every stage is stored (the discrete adjoint needs k2, k3, k10 too, not only the dense-output ones)
                    static_for<0, NR>([&](auto c) {
                        if (cwrite(c)) base[(size_t)(3 + n + q * n + comp(c)) * DFS] = k(q, c);
                    });
                });
                }
            }
            nsteps += 1;
        }
        return RET_SUCCESS;
    }
};

// blocks per CU a model's forward kernel is compiled for (Model::FWD_BLOCKS; register budget 512 / (waves per SIMD))
template <class M, class = void> struct fwd_blocks { static constexpr int v = 1; };
template <class M> struct fwd_blocks<M, std::void_t<decltype(M::FWD_BLOCKS)>> { static constexpr int v = M::FWD_BLOCKS; };
// model scratch of the forward / rhs kernels: Model::SCRATCH_FWD where a model declares one (its adjoint-only part left out)
template <class M, class = void> struct scratch_fwd { static constexpr int v = M::SCRATCH; };
template <class M> struct scratch_fwd<M, std::void_t<decltype(M::SCRATCH_FWD)>> { static constexpr int v = M::SCRATCH_FWD; };
// LDS layout of a block: [theta copy | model scratch | stage derivatives k | slot state]
template <class Model, class Tab, int G, int BLOCK, bool CPL = Model::CPL>
struct Layout {
    static constexpr int KSTRIDE = k_stride<Model::STATE_DISTRIBUTED, G, BLOCK, CPL>();
    static_assert(!CPL || Model::NS < KCP, "component-per-lane: state must fit the KCP columns");
    static constexpr int K_DOUBLES = CPL ? (BLOCK / 64) * Tab::NK * KCP : Tab::NK * Model::NS * KSTRIDE;
    // the adjoint solve never builds the lazy dense-output stages (it has no save points of its own): a distributed state with a
    // tableau of more interpolation stages than steps (Vern7: 16 against 10) keeps only the S step stages -- what lets the
    // 1024-point Fisher-KPP adjoint fit the LDS with Vern7 (82 KB instead of 131 KB of stage storage)
    static constexpr int K_DOUBLES_ADJ = (!CPL && Model::STATE_DISTRIBUTED && Tab::NK > Tab::S) ? Tab::S * Model::NS * KSTRIDE : K_DOUBLES;
    // group-shared forward-interval cache of the adjoint kernel (see AdjSys::IC_LDS)
    static constexpr bool IC_LDS = (G >= 5) && !Model::STATE_DISTRIBUTED && !Model::CPL;
    static constexpr int IC_DOUBLES = IC_LDS ? (Model::NS + Tab::NK * Model::NS) * (BLOCK / G) : 0;
    static __host__ __device__ constexpr int np_pad(int np) { return (np + 1) & ~1; }
};

// RTag: the translation unit's scalar type in the kernel's NAME (the Float32 and Float64 builds of one instance are different symbols)
template <class Model, class Tab, int G, int BLOCK, bool PT = false, class RTag = real, bool SORTED = false>
__global__ void __launch_bounds__(BLOCK, fwd_blocks<Model>::v) fwd_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using L = Layout<Model, Tab, G, BLOCK>;
    real* th = reinterpret_cast<real*>(smem_raw);
    real* scratch = th + Model::theta_lds(p.n_param);
    real* kbase = scratch + scratch_fwd<Model>::v;
    Model::stage_theta(th, p.theta, p.n_param, threadIdx.x, BLOCK);
    for (int i = threadIdx.x; i < L::K_DOUBLES; i += BLOCK) kbase[i] = 0.0;  // stage storage must always be finite
    __syncthreads();

    constexpr int GROUPS = BLOCK / G;  // trajectories per block (lanes beyond GROUPS*G idle when G is not a power of two)
    const int64_t gslot = (int64_t)blockIdx.x * GROUPS + threadIdx.x / G;
    const int r = threadIdx.x % G;
    if (gslot >= p.N || (int)threadIdx.x >= GROUPS * G) return;  // whole groups leave together
    const int64_t gid = member_of<SORTED>(p, gslot);   // (cost-ordered launch: the member this lane group works on)
    using Sys = FwdSys<Model, Tab, G, BLOCK, PT, SORTED>;
    using Drv = Driver<Tab, Sys, G, BLOCK>;
    Sys sys;
    Model::init(sys.mctx, theta_of<Model, PT>(p, th, gid), scratch, nullptr, 0, p.mc, r, p.theta);
    sys.p = &p;
    sys.tg.init(p, gid);
    sys.j = gid;
    sys.jw = gslot;
    sys.writer = (r == 0);
    sys.si = 0;
    sys.nsteps = 0;
    sys.loss = 0.0;
    sys.r = r;
    sys.n = p.n_state;
    real z[Sys::NR];
    real* kl = kbase + k_offset<Model::STATE_DISTRIBUTED, G, Model::CPL>(Tab::NK);
    real* mu = nullptr;  // no slot state in the forward pass
    static_for<0, Sys::NR>([&](auto c) { z[c] = sys.cvalid(c) ? p.u0[(size_t)gid * p.n_state + sys.comp(c)] : 0.0; });
    while (sys.si < p.ns && sys.tg.SV(p, sys.si) <= sys.tg.T0(p)) {  // save_start
        sys.save_point(sys.si, z);
        sys.si += 1;
    }
    typename Drv::Stats st;
    const int ret = Drv::run(sys, p.o, p.tab, z, kl, mu, sys.tg.T0(p), real(1), (real)p.n_state, st);
    if constexpr (Model::STATE_DISTRIBUTED) sys.loss = group_sum<G>(sys.loss);  // per-lane partial sums of the loss
    if (sys.writer) {
        if (p.stats) {
            int64_t* s = p.stats + (size_t)gid * 8;
            s[0] = st.nf; s[1] = st.nacc; s[2] = st.nrej;
            if (p.dense) { s[3] = 0; s[7] = st.nlazy; } else { s[3] = st.nlazy; s[7] = 0; }
            s[4] = 0; s[5] = 0; s[6] = 0;
        }
        p.retcode[gid] = ret;
        if (p.dense_n) p.dense_n[sys.wcol()] = sys.nsteps;
        if (p.loss_traj) p.loss_traj[gid] = ret == RET_SUCCESS ? sys.loss : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// One evaluation of the right-hand side for a batch of states (the trained UDE on saved states: what the SINDy stage
// of the scripts consumes, scenario_1.jl:152-160): du = f(u, theta), same lane-group layout as the solver kernels.
// ---------------------------------------------------------------------------------------------
template <class Model, class Tab, int G, int BLOCK, class RTag = real>
__global__ void __launch_bounds__(BLOCK) rhs_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    real* th = reinterpret_cast<real*>(smem_raw);
    real* scratch = th + Model::theta_lds(p.n_param);
    Model::stage_theta(th, p.theta, p.n_param, threadIdx.x, BLOCK);
    __syncthreads();
    constexpr int GROUPS = BLOCK / G;
    const int64_t gid = (int64_t)blockIdx.x * GROUPS + threadIdx.x / G;
    const int r = threadIdx.x % G;
    if (gid >= p.N || (int)threadIdx.x >= GROUPS * G) return;  // whole groups leave together
    using Sys = FwdSys<Model, Tab, G, BLOCK>;
    Sys sys;
    Model::init(sys.mctx, Model::THETA_GLOBAL ? const_cast<real*>(p.theta) : th, scratch, nullptr, 0, p.mc, r, p.theta);
    sys.p = &p; sys.j = gid; sys.writer = (r == 0); sys.r = r; sys.n = p.n_state;
    real z[Sys::NR], kr[Sys::NR];
    static_for<0, Sys::NR>([&](auto c) { z[c] = sys.cvalid(c) ? p.u0[(size_t)gid * p.n_state + sys.comp(c)] : 0.0; });
    Model::rhs(sys.mctx, z, kr);
    static_for<0, Sys::NR>([&](auto c) { if (sys.cwrite(c)) p.u_out[(size_t)gid * p.n_state + sys.comp(c)] = kr[c]; });
}

// ---------------------------------------------------------------------------------------------
// adjoint system (InterpolatingAdjoint): z = lambda (replicated), slots = mu
//   lambda' = -(df/du)^T lambda, mu' = -(df/dtheta)^T lambda at y(t) = forward dense interpolant;
//   save times are tstops with lambda += dL/du(t_i)            
// ---------------------------------------------------------------------------------------------
// VAR: kernel variant of the instance (build.py's waves column).  It is a template parameter so that variants of one
// (model, algorithm, lanes) are DIFFERENT kernels: a macro-only difference gives the same mangled name in several
// translation units and the linker keeps one of them.  VAR == 9: timing experiment, lambda only (no mu work at all).
template <class M, class = void> struct ks_stream_always { static constexpr bool v = false; };
template <class M> struct ks_stream_always<M, std::void_t<decltype(M::KS_STREAM_ALWAYS)>> { static constexpr bool v = M::KS_STREAM_ALWAYS; };

template <class Model, class Tab, int G, bool PT = false, int VAR = 1>
struct AdjSys {
    TimeGrid<PT> tg;
    __device__ __forceinline__ real dtmax(const OptsR& o) const { return tg.DTMAX(o); }
    static constexpr bool DEFERRED = Model::DEFERRED;
    static constexpr bool FAST = (VAR == 3);  // UDE_SENSE_FAST: lambda-only error control
    static constexpr bool ACT_CACHE = act_cache<Model>::v > 0 && !Model::DEFERRED;   // (see act_cache above)
    real cache_t;       // time of the evaluation whose activations the model's HBM row holds (NaN: none)
    static constexpr bool ACT_REG = act_reg<Model>::v && !Model::DEFERRED;
    static constexpr bool TAB_SCALAR = ACT_REG;   // (the models with register-resident weights: Driver::run, where the tableau is read)
    typename act_reg<Model>::type areg;   // (empty unless ACT_REG)
    bool store_hint;    // Driver: the NEXT evaluation is at the same time as this one (two stages with the same c)
    // VAR == 5: InterpolatingAdjoint(checkpointing = true) in its store-u-only form.  The forward store holds (t, t_end, dt, u)
    // per accepted step; entering an interval the kernel re-runs that step's stages from u with the stored dt -- the SAME
    // operation sequence as Driver::run's perform_step (and, for Vern7, its six lazy dense-output stages) on the same inputs, so
    // the recomputed k are the forward pass's k bit for bit and every result equals the dense-store mode's.  Round 3: the
    // Fisher-KPP UDEs with Tsit5; round 4: every model and both algorithms (replicated states keep the interval in registers
    // instead of the prefetched LDS row; component-per-lane systems recompute on the replicated point and keep their own
    // component), except distributed states whose NK stage vectors do not fit the registers (1024-point Fisher-KPP with Vern7).
    static constexpr bool RECOMPUTE = (VAR == 5);
    static_assert(!RECOMPUTE || !Model::STATE_DISTRIBUTED || Tab::NK * Model::NS <= 32 || (Tab::FSAL && Tab::NK == Tab::S),
                  "recompute mode: a distributed state keeps its NK stage vectors in registers");
    static constexpr int NR = Model::NS, NSL = (DEFERRED || VAR == 9) ? 0 : Model::NSL;
    static constexpr bool STATE_DISTRIBUTED = Model::STATE_DISTRIBUTED;
    // LDS-slot models re-evaluate stage 0 every step (its parameter cotangent is folded straight into the shared
    // accumulators); register-slot models hand k_S -> k_0 AND its slot derivative over (FSAL, as upstream)
    static constexpr bool SLOTS_GLOBAL = Model::SLOTS_GLOBAL, CPL = Model::CPL;
    static constexpr bool ALWAYS_K0 = DEFERRED;
    // (Driver::uni: component-per-lane systems are wave-uniform in everything that steers control flow)
    static __device__ __forceinline__ bool uni(bool b) {
        if constexpr (CPL || G >= 64) return __builtin_amdgcn_readfirstlane((int)b) != 0; else return b;
    }
    typename Model::Ctx mctx;
    const KParams* p;
    int64_t j;
    int64_t jw;   // (VAR == 6, cost-ordered launch) workspace column of member j: the lane group's own position
    __device__ __forceinline__ int64_t wcol() const { if constexpr (VAR == 6) return jw; else return j; }
    int nsteps, sf, cur, n;
    __device__ __forceinline__ int comp(int c) const {
        if constexpr (STATE_DISTRIBUTED) return Model::point(c, mctx.r); else return c;
    }
    __device__ __forceinline__ bool cvalid(int c) const { return comp(c) < n; }
    __device__ __forceinline__ bool cwrite(int c) const { return STATE_DISTRIBUTED ? cvalid(c) : mctx.r == 0; }
    __device__ __forceinline__ real state_on(int c) const { return cvalid(c) ? 1.0 : 0.0; }
    // cached forward interval: t_start/t_end in registers; u_start and the k's either in registers (IC_LDS = false)
    // or in a group-shared LDS row (IC_LDS: saves 2*NR*(NK+1) VGPRs per lane; reads are broadcasts inside the group)
    // CPL: lane c caches component c only (u_start and the k's of the interval: 1 + NK registers)
    static constexpr bool IC_LDS = (G >= 5) && !STATE_DISTRIBUTED && !CPL && !RECOMPUTE;
    static constexpr int IC_FIELDS = NR + Tab::NK * NR;
    static constexpr int IC_NR = (IC_LDS || CPL) ? 1 : NR;
    // KS_STREAM: a distributed state with more than 8 interpolation stages (Fisher-KPP with Vern7: 16 x 4 doubles per lane) does not
    // cache the interval's k in registers at all: every evaluation reads them from the dense store (coalesced, L2-resident)
    // (round 6: a model may ask for it whatever the stage count -- Model::KS_STREAM_ALWAYS: the Fisher-KPP vector kernel needs the 2 x NK x NR
    //  registers of the cached interval for its activations and deltas)
    static constexpr bool KS_STREAM = STATE_DISTRIBUTED && !CPL && ((Tab::NK > 8 && Tab::NK * NR > 32) || ks_stream_always<Model>::v) && VAR != 5;
    real ts, te, us[IC_NR], ks[(IC_LDS || KS_STREAM) ? 1 : Tab::NK][IC_NR];
    const real* kstore;   // KS_STREAM: field 0 of the current interval's record (this trajectory's column)
    real* ic;      // LDS: field f of this group at ic[f * icstride]
    int icstride;
    __device__ __forceinline__ real US(int c) const { if constexpr (IC_LDS) return ic[c * icstride]; else return us[c]; }
    __device__ __forceinline__ real KS(int q, int c) const {
        if constexpr (IC_LDS) return ic[(NR + q * NR + c) * icstride];
        else if constexpr (KS_STREAM) return cvalid(c) ? kstore[(size_t)(3 + n + q * n + comp(c)) * dense_fs<STATE_DISTRIBUTED || CPL>(*p)] : real(0);
        else return ks[q][c];
    }
    // cotangent access
    const real* cot;
    size_t cot_si, cot_sc;  // strides of save index / component

    // PREFETCH (IC_LDS groups): the backward solve walks the stored steps downwards, so while interval s is in use the
    // fields of interval s - 1 are already on their way from HBM into registers (pf_*); the switch to s - 1 is then an LDS
    // write of data that has long arrived instead of a dependent HBM round trip (~1.5 k cycles with nothing to hide it)
    static constexpr bool IC_PREFETCH = IC_LDS && (G <= 64);
    static constexpr int PF_N = IC_PREFETCH ? (IC_FIELDS + G - 1) / G : 1;
    real pf_ts, pf_f[PF_N];
    int pf_s = -1;
    __device__ __forceinline__ void prefetch_interval(int s) {
        if constexpr (IC_PREFETCH) {
            pf_s = s;
            if (s >= 0) {
                const int nf = 3 + n + Tab::NK * n;
                const real* base = dense_rec<STATE_DISTRIBUTED || CPL>(*p, s, nf, wcol());
                pf_ts = base[0];
                static_for<0, PF_N>([&](auto q) {
                    const int f = mctx.r + (int)decltype(q)::value * G;
                    pf_f[q] = base[(size_t)(3 + (f < IC_FIELDS ? f : 0)) * p->Npad];
                });
            }
        }
    }
    __device__ __forceinline__ void load_interval(int s) {
        if constexpr (IC_PREFETCH) {
            if (s == pf_s && s >= 0) {  // (group-uniform: every lane of a trajectory prefetched the same interval)
                sf = s;
                te = ts;                // stored steps are contiguous: t_end(s) == t_start(s + 1), the interval being left
                ts = pf_ts;
                asm volatile("" ::: "memory");
                static_for<0, PF_N>([&](auto q) {
                    const int f = mctx.r + (int)decltype(q)::value * G;
                    if (f < IC_FIELDS) ic[f * icstride] = pf_f[q];
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                prefetch_interval(s - 1);
                return;
            }
        }
        sf = s;
        const int nf = RECOMPUTE ? 3 + n : 3 + n + Tab::NK * n;
        const size_t DFS = dense_fs<STATE_DISTRIBUTED || CPL>(*p);
        const real* base = dense_rec<STATE_DISTRIBUTED || CPL>(*p, s, nf, wcol());
        ts = base[0];
        te = base[(size_t)1 * DFS];
        if constexpr (RECOMPUTE) {
            const real dtf = base[(size_t)2 * DFS];   // the step size the forward pass used (t_end may be a snapped tstop)
            const TabDevT<real>* tab = p->tab;
            if constexpr (CPL) {
                // component-per-lane: lane c keeps component c of u and of every k; a stage point is formed by that lane
                // (Driver::run's CPL chain: terms j < s only) and broadcast, the right-hand side runs on the replicated point
                const bool on = mctx.r < n;
                us[0] = on ? base[(size_t)(3 + (on ? mctx.r : 0)) * DFS] : 0.0;
                static_for<0, Tab::NK>([&](auto q) { ks[q][0] = 0.0; });
                static_for<0, Tab::NK>([&](auto sc) {
                    constexpr int st = decltype(sc)::value;
                    real zs[NR], kr[NR];
                    if constexpr (st == 0) {
                        bcast_all(us[0], zs);
                    } else {
                        real acc = tab->A[st][0] * ks[0][0];
                        static_for<1, st>([&](auto jc) { acc = rfma(tab->A[st][decltype(jc)::value], ks[decltype(jc)::value][0], acc); });
                        bcast_all(rfma(dtf, acc, us[0]), zs);
                    }
                    asm volatile("" ::: "memory");
                    Model::rhs(mctx, zs, kr);
                    ks[st][0] = own_of(kr);
                });
            } else {
            static_for<0, NR>([&](auto c) { us[c] = cvalid(c) ? base[(size_t)(3 + comp(c)) * DFS] : 0.0; });
            static_for<0, Tab::NK>([&](auto q) { static_for<0, NR>([&](auto c) { ks[q][c] = 0.0; }); });
            static_for<0, Tab::NK>([&](auto sc) {
                constexpr int st = decltype(sc)::value;
                real zs[NR], kr[NR];
                if constexpr (st == 0) {
                    static_for<0, NR>([&](auto c) { zs[c] = us[c]; });
                } else if constexpr (st < Tab::S) {
                    static_for<0, NR>([&](auto c) {   // Driver::run: all S - 1 terms, zero coefficients (and zero k) contribute exactly nothing
                        real acc = tab->A[st][0] * ks[0][c];
                        static_for<1, Tab::S - 1>([&](auto jc) { acc = rfma(tab->A[st][decltype(jc)::value], ks[decltype(jc)::value][c], acc); });
                        zs[c] = rfma(dtf, acc, us[c]);
                    });
                } else {
                    static_for<0, NR>([&](auto c) {   // Driver::run's lazy(): the dense-output stages, terms j < row
                        real acc = tab->A[st][0] * ks[0][c];
                        static_for<1, st>([&](auto jc) { acc = rfma(tab->A[st][decltype(jc)::value], ks[decltype(jc)::value][c], acc); });
                        zs[c] = rfma(dtf, acc, us[c]);
                    });
                }
                asm volatile("" ::: "memory");
                Model::rhs(mctx, zs, kr);
                static_for<0, NR>([&](auto c) { ks[st][c] = kr[c]; });
            });
            }
        } else if constexpr (IC_LDS) {
            // the G lanes of the group fetch the fields round-robin and publish them in the group's LDS row
            asm volatile("" ::: "memory");
            if constexpr (G > 64) __syncthreads();  // (block-uniform: t is replicated) readers of the old row are done
            for (int f = mctx.r; f < IC_FIELDS; f += G) ic[f * icstride] = base[(size_t)(3 + f) * DFS];
            if constexpr (G > 64) __syncthreads();
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            prefetch_interval(s - 1);
        } else if constexpr (CPL) {
            const bool on = mctx.r < n;
            const int rc = on ? mctx.r : 0;
            us[0] = on ? base[(size_t)(3 + rc) * DFS] : 0.0;
            static_for<0, Tab::NK>([&](auto q) {
                if constexpr (Tab::dense_uses(q)) ks[q][0] = on ? base[(size_t)(3 + n + q * n + rc) * DFS] : 0.0;
            });
        } else {
            static_for<0, NR>([&](auto c) { us[c] = cvalid(c) ? base[(size_t)(3 + comp(c)) * DFS] : 0.0; });
            if constexpr (KS_STREAM) kstore = base;
            else
            static_for<0, Tab::NK>([&](auto q) {
                if constexpr (Tab::dense_uses(q))
                    static_for<0, NR>([&](auto c) {
                        ks[q][c] = cvalid(c) ? base[(size_t)(3 + n + q * n + comp(c)) * DFS] : 0.0;
                    });
            });
        }
    }
    // sol(t, continuity = :right): interval [s, s+1] with t_s <= t, clamped to the stored range
    __device__ __forceinline__ void locate(real t) {
        if (uni(t < ts || t >= te)) {   // (ONE test on the common path: the evaluation time lies inside the cached interval)
            while (uni(t < ts && sf > 0)) load_interval(sf - 1);
            while (uni(t >= te && sf < nsteps - 1)) load_interval(sf + 1);
        }
    }
    // ---- deferred slots: slot state in HBM, two columns (current / candidate) that swap on acceptance ----
    real *mu_cur, *mu_new;
    int ms;
    real rq[7];  // CPL: lane q's Horner table of b_q(theta)
    __device__ __forceinline__ void load_bth_table() {
        const int q = mctx.r < Tab::NK ? mctx.r : 0;
        static_for<0, 7>([&](auto i) { rq[i] = p->tab->R[q][i]; });
    }
    // b_q(theta) evaluated by lane q alone (bit-identical to Tab::bth), handed to the component lanes as scalars
    __device__ __forceinline__ void bth_lanes(real th, real* b) const {
        real h = rq[0];
        static_for<1, 7>([&](auto i) { h = rfma(th, h, rq[i]); });
        const real bq = (mctx.r == 0 ? th : th * th) * h;
        static_for<0, Tab::NK>([&](auto q) {
            if constexpr (Tab::dense_uses(q)) b[q] = readlane_real(bq, decltype(q)::value);
        });
    }
    __device__ __forceinline__ void eval_store(real t, const real* lam, real* klam, int s) {
        if constexpr (DEFERRED) {
            asm volatile("" ::: "memory");
            locate(t);
            const real dtf = te - ts;
            const real th = (t - ts) / dtf;
            real b[Tab::NK], y[NR], dl[NR];
            bth_lanes(th, b);
            static_assert(!DEFERRED || CPL, "deferred slots come with component-per-lane interval caches");
            const real acc = chain2<RowDense<Tab>, Tab::NK>([&](auto q) { return ks[q][0]; }, [&](auto q) { return b[q]; });
            bcast_all(rfma(dtf, acc, us[0]), y);
            int slot = s;  // where the stage's factors go: models that keep only the stages the tableau weights use get the compacted slot
            if constexpr (Model::COMPACT_STAGES) slot = __builtin_popcount(stage_mask() & ((1u << s) - 1u));
            Model::vjp_store(mctx, y, lam, dl, slot);
            static_for<0, NR>([&](auto c) { klam[c] = -dl[c]; });
        }
    }
    __device__ __forceinline__ void slot_init01(const OptsR& o, real& h0, real& l0, real& h1, real& l1) {
        if constexpr (DEFERRED) Model::init_norm01(mctx, o.abstol, o.reltol, mu_cur, ms, h0, l0, h1, l1);
    }
    __device__ __forceinline__ void slot_init2(const OptsR& o, real& h2, real& l2) {
        if constexpr (DEFERRED) Model::init_norm2(mctx, o.abstol, o.reltol, mu_cur, ms, h2, l2);
    }
    // stages whose B and BT weights are both zero (Vern7: stages 2, 3) drop out of the deferred sums at compile time
    static constexpr unsigned stage_mask() {
        unsigned m = 0;
        for (int s = 0; s < Tab::S; ++s)
            if (Tab::B(s) != 0.0 || Tab::BT(s) != 0.0) m |= 1u << s;
        return m;
    }
    __device__ __forceinline__ acc_t slot_step(real dt, const TabDev* tab, const OptsR& o) {
        if constexpr (DEFERRED)
            return (acc_t)Model::template step_slots<Tab::S, stage_mask()>(mctx, tab->B, tab->BT, dt, o.abstol, o.reltol, mu_cur, mu_new, ms);
        else return 0.0;
    }
    // fast mode: mu += dt * sum_s B_s g_s for every slot, in place, on ACCEPTED steps only
    __device__ __forceinline__ void slot_commit(real dt, const TabDev* tab) {
        if constexpr (DEFERRED) Model::template commit_slots<Tab::S, stage_mask()>(mctx, tab->B, dt, mu_cur, ms);
    }
    __device__ __forceinline__ void slot_accept() {
        real* t = mu_cur; mu_cur = mu_new; mu_new = t;
    }
    __device__ __forceinline__ void eval(real t, const real* lam, real* klam, real* g) {
      if constexpr (!DEFERRED) {
        asm volatile("" ::: "memory");  // keep the LDS-staged weights in LDS (no hoisting into registers)
        locate(t);
        const real dtf = te - ts;
        const real th = (t - ts) / dtf;
        real b[Tab::NK], y[NR], dl[NR];
        Tab::bth(th, b);
        static_for<0, NR>([&](auto c) {
            const real acc = chain2<RowDense<Tab>, Tab::NK>([&](auto q) { return KS(q, c); }, [&](auto q) { return b[q]; });
            y[c] = rfma(dtf, acc, US(c));
        });
        if constexpr (ACT_REG) {
            const bool same = t == cache_t;   // (per lane group: its lanes hold the same t)
            cache_t = t;
            Model::template vjp_c<(NSL > 0)>(mctx, y, lam, dl, g, areg, same);
        } else if constexpr (ACT_CACHE) {
            const int mode = uni(t == cache_t) ? ACT_LOAD : (store_hint ? ACT_STORE : ACT_NONE);
            if (mode == ACT_STORE) cache_t = t;
            Model::template vjp<(NSL > 0)>(mctx, y, lam, dl, g, mode);
        } else
        Model::template vjp<(NSL > 0)>(mctx, y, lam, dl, g);
        static_for<0, NR>([&](auto c) { klam[c] = -dl[c]; });
        static_for<0, NSL>([&](auto c) { g[c] = -g[c]; });
      }
    }
    __device__ __forceinline__ void trace(int iter, real t, real dt, real e, real q, bool acc) const {
        if (p->trace && mctx.r == 0 && j == p->trace_traj && iter <= p->trace_cap) {
            real* row = p->trace + ((size_t)p->trace_cap + (iter - 1)) * 5;
            row[0] = t; row[1] = dt; row[2] = e; row[3] = q; row[4] = acc ? 1.0 : 0.0;
        }
    }

    __device__ __forceinline__ real tstop_from_cur() const {
        // next save time strictly inside (t0, t) in descending order, else t0
        return (cur >= 0 && tg.SV(*p, cur) > tg.T0(*p)) ? tg.SV(*p, cur) : tg.T0(*p);
    }
    __device__ __forceinline__ real first_tstop() const { return tstop_from_cur(); }
    __device__ __forceinline__ bool at_tstop(real t, real* lam) {
        bool mod = false;
        while (uni(cur >= 0 && tg.SV(*p, cur) >= t)) {
            if (uni(tg.SV(*p, cur) == t)) {
                static_for<0, NR>([&](auto c) {
                    if (cvalid(c)) lam[c] += cot[(size_t)cur * cot_si + (size_t)comp(c) * cot_sc];
                });
                mod = true;
            }
            cur -= 1;
        }
        return mod;
    }
    __device__ __forceinline__ bool next_tstop(real& tstop) {
        if (uni(tstop == tg.T0(*p))) return false;
        tstop = tstop_from_cur();
        return true;
    }
    template <class Lazy>
    __device__ __forceinline__ int accepted(real, real, real, const real*, const real*, const real*, Lazy&) {
        return RET_SUCCESS;
    }
};

// models whose discrete sweep defers the parameter cotangent (Model::DADJ_DEFERRED: the runtime-shape model -- hundreds of slots per lane)
template <class M, class = void> struct dadj_deferred { static constexpr bool v = false; };
template <class M> struct dadj_deferred<M, std::void_t<decltype(M::DADJ_DEFERRED)>> { static constexpr bool v = M::DADJ_DEFERRED; };
// models that can run with per-member parameters (Model::PER_MEMBER_THETA: theta is only read while the context is set up --
// weights and coefficients live in registers afterwards --, so a member's own HBM column can stand in for the block's LDS copy)
template <class M, class = void> struct per_member { static constexpr bool v = false; };
template <class M> struct per_member<M, std::void_t<decltype(M::PER_MEMBER_THETA)>> { static constexpr bool v = M::PER_MEMBER_THETA; };
// the parameter vector a trajectory's context is built from: the block's LDS copy (or theta in HBM for THETA_GLOBAL models), or --
// per-trajectory kernel variants of per-member models with UDE_PT_THETA -- the member's own column
template <class Model, bool PT>
__device__ __forceinline__ real* theta_of(const KParams& p, real* th_lds, int64_t gid) {
    if constexpr (PT && per_member<Model>::v) {
        if (p.theta_pm) return const_cast<real*>(p.theta) + (size_t)gid * p.theta_pm;
    }
    return Model::THETA_GLOBAL ? const_cast<real*>(p.theta) : th_lds;
}
template <class M, class = void> struct model_gfac { static constexpr int v = 0; };
template <class M> struct model_gfac<M, std::void_t<decltype(M::GFAC)>> { static constexpr int v = M::GFAC; };

template <class Model, class Tab, int G, int BLOCK, bool PT = false, int VAR = 1, class RTag = real>
__global__ void __launch_bounds__(BLOCK, (VAR == 2 ? 2 : 1)) adj_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using L = Layout<Model, Tab, G, BLOCK>;
    real* th = reinterpret_cast<real*>(smem_raw);
    real* scratch = th + Model::theta_lds(p.n_param);
    real* kbase = scratch + Model::SCRATCH;
    real* slots = kbase + L::K_DOUBLES_ADJ;
    const int np_pad = L::np_pad(p.n_param);
#ifdef UDE_EXP_LDS_FILL  // debugging experiment: poison (or zero) the whole dynamic LDS before anything is staged
    {
        const int tot = Model::theta_lds(p.n_param) + Model::SCRATCH + L::K_DOUBLES_ADJ;
        for (int i = threadIdx.x; i < tot; i += BLOCK) th[i] = UDE_EXP_LDS_FILL;
        __syncthreads();
    }
#endif
    Model::stage_theta(th, p.theta, p.n_param, threadIdx.x, BLOCK);
    for (int i = threadIdx.x; i < L::K_DOUBLES_ADJ; i += BLOCK) kbase[i] = 0.0;  // stage storage must always be finite
    __syncthreads();

    constexpr int GROUPS = BLOCK / G;
    // VAR == 6: the cost-ordered launch of a multi-round ensemble (KParams::perm; udecore.hip) -- lane group g works on member perm[g] and writes
    // that member's own gradient row.  A kernel of its own: the default kernel's code is what it was (as a run-time branch the two extra
    // loads moved the headline kernel's register allocation and cost it 4 %).
    constexpr bool SORTED = (VAR == 6);
    const int64_t gslot = (int64_t)blockIdx.x * GROUPS + threadIdx.x / G;
    int64_t gid = gslot;
    if constexpr (SORTED) gid = gslot < p.N ? (int64_t)p.perm[gslot] : gslot;
    const int r = threadIdx.x % G;
    using Sys = AdjSys<Model, Tab, G, PT, VAR>;
    using Drv = Driver<Tab, Sys, G, BLOCK>;
    constexpr int NSL = Sys::NSL;
    constexpr int NSLA = NSL > 0 ? NSL : 1;
    real* kl = kbase + k_offset<Model::STATE_DISTRIBUTED, G, Model::CPL>(Tab::NK);
    // register-slot mode: slot state mu of thread tid, element c at mu_lds[c * BLOCK]
    constexpr bool SG = Model::SLOTS_GLOBAL;  // mu in HBM (fused-accumulation models: nothing else needs a column)
    const int MS = SG ? (int)(gridDim.x * BLOCK) : BLOCK;
    real* mu_lds = SG ? p.slot_glob + (size_t)blockIdx.x * BLOCK + threadIdx.x : slots + threadIdx.x;
    real* gtmp = slots + (size_t)NSLA * BLOCK + threadIdx.x;        // initial-dt scratch, element c at gtmp[c * BLOCK]
    real* gtmp2 = slots + (size_t)2 * NSLA * BLOCK + threadIdx.x;    // last-stage slot derivative (FSAL hand-over)
    // (the third column only exists for FSAL tableaux: at 4 blocks per CU every LDS kilobyte counts -- the C2 ensemble
    // must fit the chip in ONE round of blocks)
    real* icbase = slots + (SG ? (size_t)0 : (size_t)(Tab::FSAL ? 3 : 2) * NSLA * BLOCK);  // interval cache rows (IC_LDS)
    real lam[Sys::NR];
    static_for<0, Sys::NR>([&](auto c) { lam[c] = 0.0; });
    constexpr int NSLOT = Model::DEFERRED ? Model::NSL : NSL;  // slots this thread reports (deferred: system-owned)
    if constexpr (NSLOT > 160) {  // (runtime-shape model: hundreds of slots -- a loop, not unrolled stores)
#pragma unroll 4
        for (int c = 0; c < NSLOT; ++c) mu_lds[(size_t)c * MS] = 0.0;
    } else static_for<0, NSLOT>([&](auto c) { mu_lds[(size_t)c * MS] = 0.0; });
    real* mu_final = mu_lds;
    const bool in_range = gid < p.N && (int)threadIdx.x < GROUPS * G;
    bool ok = in_range && p.retcode[in_range ? gid : 0] == RET_SUCCESS;
    if constexpr (Model::CPL || G >= 64) ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;  // one wavefront per trajectory: a scalar condition (Driver::uni)
    if (ok) {
        Sys sys;
        Model::init(sys.mctx, theta_of<Model, PT>(p, th, gid), scratch, slots, np_pad, p.mc, r, p.theta);
        sys.p = &p;
        sys.tg.init(p, gid);
        sys.j = gid;
        sys.jw = gslot;
        sys.n = p.n_state;
        if constexpr (Sys::ACT_REG) sys.cache_t = __builtin_nan("");
        if constexpr (Sys::ACT_CACHE) {
            sys.cache_t = __builtin_nan("");
            sys.store_hint = false;
            sys.mctx.acache = p.slot_glob + (size_t)blockIdx.x * (size_t)act_cache<Model>::v * BLOCK;   // this block's row
        }
        if constexpr (Model::DEFERRED) sys.load_bth_table();
        if constexpr (model_gfac<Model>::v > 0) {   // stage factors a model keeps in HBM: this thread's words behind its two mu columns
            sys.mctx.gfac = mu_lds + (size_t)(2 * Model::NSL) * MS;
            sys.mctx.gms = MS;
        }
        sys.mu_cur = mu_lds;
        sys.mu_new = mu_lds + (size_t)(Model::DEFERRED ? Model::NSL : 0) * MS;
        sys.ms = MS;
        sys.ic = icbase + threadIdx.x / G;
        sys.icstride = BLOCK / G;
        sys.nsteps = p.dense_n[sys.wcol()];
        if (p.cot_in || Model::STATE_DISTRIBUTED) {   // (distributed states keep a trajectory's cotangent rows contiguous)
            sys.cot = (p.cot_in ? p.cot_in : p.cot) + (size_t)gid * p.ns * p.n_state;
            sys.cot_si = p.n_state;
            sys.cot_sc = 1;
        } else {
            sys.cot = p.cot + sys.wcol();
            sys.cot_si = (size_t)p.n_state * p.Npad;
            sys.cot_sc = p.Npad;
        }
        sys.cur = p.ns - 1;
        sys.load_interval(sys.nsteps - 1);
        sys.at_tstop(sys.tg.TF(p), lam);  // init_cb: the jump at t = tf precedes the first step
        typename Drv::Stats st;
        const int ret = Drv::run(sys, p.o, p.tab, lam, kl, mu_lds, sys.tg.TF(p), real(-1), (real)(Sys::FAST ? p.n_state : p.n_state + p.n_param), st, gtmp, gtmp2, MS);
        if constexpr (Model::DEFERRED) mu_final = sys.mu_cur;
        if (r == 0) {
            if (p.stats) {
                int64_t* s = p.stats + (size_t)gid * 8;
                s[4] = st.nf; s[5] = st.nacc; s[6] = st.nrej;
            }
            if (ret != RET_SUCCESS) p.retcode[gid] = ret;
        }
        if (p.grad_u0)
            static_for<0, Sys::NR>([&](auto c) {
                if (sys.cwrite(c)) p.grad_u0[(size_t)gid * p.n_state + sys.comp(c)] = lam[c];
            });
        if (ret != RET_SUCCESS) {  // never poison the batch gradient
            if constexpr (NSLOT > 160) {
                for (int c = 0; c < NSLOT; ++c) mu_final[(size_t)c * MS] = 0.0;
            } else static_for<0, NSLOT>([&](auto c) { mu_final[(size_t)c * MS] = 0.0; });
        }
    }
    if constexpr (PT && per_member<Model>::v && !SG) {
        if (p.theta_pm) {   // per-member parameters: trajectory gid's lanes write ITS gradient row, nothing is summed
            if (in_range) {
                real* row = p.grad_part + (size_t)gid * p.n_param;
                for (int s = 0; s < NSL; ++s) {
                    const int idx = Model::slot_index(p.mc, r, s);
                    if (idx >= 0) row[idx] = mu_lds[(size_t)s * BLOCK];
                }
            }
            return;
        }
    }
    if constexpr (SORTED) {   // one gradient row per member, written by its own lanes; summed in member order afterwards (rows_chunk_sum_kernel)
        static_assert(!SORTED || (!SG && !Model::DEFERRED), "cost-ordered launch: register-slot models");
        if (in_range) {
            real* row = p.grad_part + (size_t)gid * p.n_param;
            for (int s = 0; s < NSL; ++s) {
                const int idx = Model::slot_index(p.mc, r, s);
                if (idx >= 0) row[idx] = mu_lds[(size_t)s * BLOCK];
            }
        }
        return;
    }
    if constexpr (!pow2_group<G>()) {
        // mu already sits in LDS ([slot][thread]): lane r of group 0 adds the r-th lanes of all groups in ascending
        // group order and writes the wave's partial row (runtime loops: this tail must not inflate the register peak)
        __syncthreads();
        if ((int)threadIdx.x < G) {
            const int64_t wave = part_row<G, BLOCK>();
            real* row = p.grad_part + (size_t)wave * p.n_param;
            const real* base = slots;
            for (int s = 0; s < NSL; ++s) {
                const int idx = Model::slot_index(p.mc, (int)threadIdx.x, s);
                if (idx >= 0) {
                    real acc = 0.0;
                    for (int gq = 0; gq < GROUPS; ++gq) acc += base[s * BLOCK + gq * G + (int)threadIdx.x];
                    row[idx] = acc;
                }
            }
        }
    }
    if constexpr (SG) {
        // one wavefront per trajectory, mu in HBM: the wave's partial row is its own mu, slot by slot
        static_assert(G == 64, "HBM slot state: one wavefront per trajectory");
        real* row = p.grad_part + (size_t)part_row<G, BLOCK>() * p.n_param;
        // (a runtime loop: unrolled, the compiler keeps all NSLOT loads in flight -- 2 x 146 registers for the neural ODE)
#ifdef UDE_EXP_TAIL_UNROLL
        static_for<0, NSLOT>([&](auto c) {
            const int idx = Model::slot_index(p.mc, r, c);
            if (idx >= 0) row[idx] = mu_final[(size_t)c * MS];
        });
#else
#pragma unroll 2
        for (int c = 0; c < NSLOT; ++c) {
            const int idx = Model::slot_index(p.mc, r, c);
            if (idx >= 0) row[idx] = mu_final[(size_t)c * MS];
        }
#endif
    }
    if constexpr (!SG && pow2_group<G>()) {
    real mu[NSLA];
    static_for<0, NSL>([&](auto c) { mu[c] = mu_lds[c * BLOCK]; });
    // ---- deterministic reduction: groups of a wave (xor butterfly), then one partial row per wave ----
    static_for<0, NSL>([&](auto c) {
        real v = mu[c];
#pragma unroll
        for (int m = G; m < (BLOCK < 64 ? BLOCK : 64); m <<= 1) v += __shfl_xor(v, m, 64);
        mu[c] = v;
    });
    const int lane = G > 64 ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    if (lane < G) {
        const int64_t wave = part_row<G, BLOCK>();
        real* row = p.grad_part + (size_t)wave * p.n_param;
        for (int s = 0; s < NSL; ++s) {
            const int idx = Model::slot_index(p.mc, lane, s);
            if (idx >= 0) {
                real v = 0.0;
                static_for<0, NSL>([&](auto c) { v = (s == c) ? mu[c] : v; });
                row[idx] = v;
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// a9 / SURVEY 8(f) N2: discretise-then-optimise gradient -- what `sensealg = ForwardDiffSensitivity()`
// (scenario_1.jl:86, scenario_2.jl:108, scenario_3.jl:124, hudson_bay.jl:102) differentiates: the discrete RK map
// with the step sequence and the save-point interpolation weights frozen.  Reverse sweep over the stored steps,
// one VJP per stage (+ the lazy Vern7 stages of steps that contain a save point); no error control, no
// divisions.  Accumulation order = oracle/ude_oracle_impl.h: discrete_sweep (ARITH-SPEC), so per-trajectory
// results are bit-identical to the oracle.
// ---------------------------------------------------------------------------------------------
template <class Model, class Tab, int G, int BLOCK, bool PT = false, class RTag = real>
__global__ void __launch_bounds__(BLOCK) dadj_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using L = Layout<Model, Tab, G, BLOCK, false>;  // (the reverse sweep keeps the replicated stage layout)
    constexpr int NR = Model::NS, NSL = Model::NSL, NSLA = NSL > 0 ? NSL : 1;
    constexpr int S = Tab::S, NK = Tab::NK, KSTRIDE = L::KSTRIDE, GROUPS = BLOCK / G;
    constexpr bool DIST = Model::STATE_DISTRIBUTED;
    real* th = reinterpret_cast<real*>(smem_raw);
    real* scratch = th + Model::theta_lds(p.n_param);
    // k of the current step: an LDS copy -- or, for models whose LDS is spoken for (DADJ_K_FROM_DENSE), read straight
    // from the dense store in HBM (21 L2-resident loads per component and step)
    constexpr bool KD = Model::DADJ_K_FROM_DENSE;
    real* kbase = scratch + Model::SCRATCH;
    real* kbbase = kbase + (KD ? 0 : L::K_DOUBLES);  // kbar
    real* slots = kbbase + L::K_DOUBLES;
    const int np_pad = L::np_pad(p.n_param);
    Model::stage_theta(th, p.theta, p.n_param, threadIdx.x, BLOCK);
    constexpr bool SG = Model::SLOTS_GLOBAL;
    constexpr bool DDEF = dadj_deferred<Model>::v;  // accumulators in this thread's HBM column, factors of the VJPs in the stage storage
    real* acc_lds = slots + threadIdx.x;  // register-slot models: accumulator row, element c at acc_lds[c*BLOCK]
    if constexpr (!SG) static_for<0, NSL>([&](auto c) { acc_lds[c * BLOCK] = 0.0; });
    const int MSG = (int)(gridDim.x * BLOCK);
    [[maybe_unused]] real* accg = DDEF ? p.slot_glob + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr;  // element c at accg[c * MSG]
    if constexpr (DDEF) {
#pragma unroll 4
        for (int c = 0; c < NSL; ++c) accg[(size_t)c * MSG] = 0.0;
    }
    __syncthreads();

    const int64_t gid = (int64_t)blockIdx.x * GROUPS + threadIdx.x / G;
    const int r = threadIdx.x % G;
    const bool in_range = gid < p.N && (int)threadIdx.x < GROUPS * G;
    const bool ok = in_range && p.retcode[in_range ? gid : 0] == RET_SUCCESS;
    if (ok) {
        typename Model::Ctx mctx;
        Model::init(mctx, theta_of<Model, PT>(p, th, gid), scratch, slots, np_pad, p.mc, r, p.theta);
        const int n = p.n_state;
        auto comp = [&](int c) {
            if constexpr (DIST) return Model::point(c, r); else return c;
        };
        auto cvalid = [&](int c) { return comp(c) < n; };
        auto cwrite = [&](int c) { return DIST ? cvalid(c) : r == 0; };
        const int koff = k_offset<DIST, G>();
        TimeGrid<PT> tg;
        tg.init(p, gid);
        const real* kdense = nullptr;  // first stage field of the current step in the dense store (KD)
        auto K = [&](int j, int c) -> real {
            if constexpr (KD) return cvalid(c) ? kdense[(size_t)(j * n + comp(c)) * dense_fs<DIST || Model::CPL>(p)] : 0.0;
            else return kbase[(j * NR + c) * KSTRIDE + koff];
        };
        auto KB = [&](int j, int c) -> real& { return kbbase[(j * NR + c) * KSTRIDE + koff]; };
        const TabDev* tab = p.tab;
        const real* cot;
        size_t cot_si, cot_sc;
        if (p.cot_in || DIST) {
            cot = (p.cot_in ? p.cot_in : p.cot) + (size_t)gid * p.ns * n;
            cot_si = n;
            cot_sc = 1;
        } else {
            cot = p.cot + gid;
            cot_si = (size_t)n * p.Npad;
            cot_sc = p.Npad;
        }
        auto COT = [&](int i, int c) { return cot[(size_t)i * cot_si + (size_t)comp(c) * cot_sc]; };
        const int nsteps = p.dense_n[gid];
        const int nf = 3 + n + NK * n;
        real ubar[NR], un[NR], carry[NR], acc[DDEF ? 1 : NSLA];
        static_for<0, NR>([&](auto c) { ubar[c] = 0.0; carry[c] = 0.0; });
        if constexpr (!DDEF) static_for<0, NSL>([&](auto c) { acc[c] = 0.0; });
        [[maybe_unused]] int nst = 0;  // DDEF: VJPs whose factors wait in the stage storage
        int si = p.ns - 1;
        int64_t nvjp = 0;
        // one VJP at stage input g with stage cotangent kbrow: w = (df/du)^T kbrow; parameter part into acc
        auto stage_vjp = [&](const real* g, const real* kbrow, real* w) {
            asm volatile("" ::: "memory");
            if constexpr (DDEF) {
                if (nst == Model::NSTC) {  // (wave-uniform) the stage storage is full: add its VJPs to the accumulators, in order
                    Model::dadj_flush(mctx, nst, accg, MSG);
                    nst = 0;
                    asm volatile("" ::: "memory");
                }
                Model::vjp_store(mctx, g, kbrow, w, nst);
                nst += 1;
            } else if constexpr (Model::FUSED_ACC) {
                Model::template vjp_acc<false>(mctx, g, kbrow, w, acc, acc, real(-1), real(0));  // acc += (df/dtheta)^T kbar
            } else {
                real gs[NSLA];
                Model::template vjp<true>(mctx, g, kbrow, w, gs);
                static_for<0, NSL>([&](auto c) { acc[c] += gs[c]; });
            }
            static_for<0, NR>([&](auto c) { un[c] += w[c]; });
            nvjp += 1;
        };
        // small states: the whole record of step st - 1 (times, u_n, every stage derivative) is requested while step st is
        // being swept, and consumed one iteration later -- no HBM round trip on the critical path of the sweep
        constexpr bool PREF = !KD && NK * NR <= 32;
        struct StepRec {
            real tn, tn1, dt, u[NR], k[PREF ? NK : 1][NR];
        };
        auto fetch_step = [&](int st, StepRec& rec) {
            const size_t DFS = dense_fs<DIST || Model::CPL>(p);
            const real* base = dense_rec<DIST || Model::CPL>(p, st, nf, gid);
            rec.tn = base[0]; rec.tn1 = base[(size_t)1 * DFS]; rec.dt = base[(size_t)2 * DFS];
            static_for<0, NR>([&](auto c) { rec.u[c] = cvalid(c) ? base[(size_t)(3 + comp(c)) * DFS] : real(0); });
            if constexpr (PREF)
                static_for<0, NK>([&](auto q) {
                    static_for<0, NR>([&](auto c) { rec.k[q][c] = cvalid(c) ? base[(size_t)(3 + n + (int)decltype(q)::value * n + comp(c)) * DFS] : real(0); });
                });
        };
        StepRec nxt;
        if constexpr (PREF) { if (nsteps > 0) fetch_step(nsteps - 1, nxt); }
        for (int st = nsteps - 1; st >= 0; --st) {
            const size_t DFS = dense_fs<DIST || Model::CPL>(p);
            const real* base = dense_rec<DIST || Model::CPL>(p, st, nf, gid);
            StepRec cur;
            if constexpr (PREF) {
                cur = nxt;
                if (st > 0) fetch_step(st - 1, nxt);
            } else {
                fetch_step(st, cur);
            }
            const real tn = cur.tn, tn1 = cur.tn1, dt = cur.dt;
            real u_n[NR];
            static_for<0, NR>([&](auto c) { u_n[c] = cur.u[c]; });
            kdense = base + (size_t)(3 + n) * DFS;
            if constexpr (PREF) {
                static_for<0, NK>([&](auto q) {
                    static_for<0, NR>([&](auto c) { kbase[((int)decltype(q)::value * NR + c) * KSTRIDE + koff] = cur.k[q][c]; });
                });
            } else {
            for (int q = 0; q < NK; ++q)
                static_for<0, NR>([&](auto c) {
                    if constexpr (!KD) kbase[(q * NR + c) * KSTRIDE + koff] = cvalid(c) ? base[(size_t)(3 + n + q * n + comp(c)) * DFS] : 0.0;
                });
            }
            // (1) saves exactly at the step end feed the cotangent of u_{n+1}
            while (si >= 0 && tg.SV(p, si) >= tn1) {
                if (tg.SV(p, si) == tn1) static_for<0, NR>([&](auto c) { if (cvalid(c)) ubar[c] += COT(si, c); });
                si -= 1;
            }
            // (2) u_{n+1} = u_n + dt*sum B_j k_j
            static_for<0, NR>([&](auto c) { un[c] = ubar[c]; });
            for (int j = 0; j < NK; ++j) {
                const real bj = j < S ? tab->B[j] : real(0);
                static_for<0, NR>([&](auto c) { KB(j, c) = bj != real(0) ? (dt * bj) * ubar[c] : real(0); });
            }
            if constexpr (Tab::FSAL) static_for<0, NR>([&](auto c) { KB(S - 1, c) += carry[c]; });
            // (3) saves strictly inside the step, descending: y = u_n + dt*sum b_j(theta) k_j
            bool interior = false;
            while (si >= 0 && tg.SV(p, si) > tn) {
                const real thv = (tg.SV(p, si) - tn) / dt;
                real bw[NK];
                Tab::bth(thv, bw);
                static_for<0, NR>([&](auto c) {
                    const real dl = cvalid(c) ? COT(si, c) : 0.0;
                    un[c] += dl;
                    static_for<0, NK>([&](auto j) {
                        if constexpr (Tab::dense_uses(j)) KB(j, c) = rfma(dt * bw[j], dl, KB(j, c));
                    });
                });
                interior = true;
                si -= 1;
            }
            if constexpr (DIST) interior = __any(interior);  // wave-uniform (the save grid is shared anyway)
            // (4)+(5) lazy dense-output stages (only if a save point used them), then the main stages S-1 .. 1
            for (int row = (Tab::NEXTRA > 0 && interior) ? S + Tab::NEXTRA - 1 : S - 1; row >= 1; --row) {
                real g[NR], kbrow[NR], w[NR];
                static_for<0, NR>([&](auto c) {
                    real a = tab->A[row][0] * K(0, c);
                    for (int j = 1; j < row; ++j) a = rfma(tab->A[row][j], K(j, c), a);
                    g[c] = rfma(dt, a, u_n[c]);
                    kbrow[c] = KB(row, c);
                });
                stage_vjp(g, kbrow, w);
                static_for<0, NR>([&](auto c) {
                    for (int j = 0; j < row; ++j) KB(j, c) = rfma(dt * tab->A[row][j], w[c], KB(j, c));
                });
            }
            // (6) stage 0: k_0 = f(u_n); FSAL: it is the previous step's last stage -- hand kbar_0 over
            if (Tab::FSAL && st > 0) {
                static_for<0, NR>([&](auto c) { carry[c] = KB(0, c); });
            } else {
                real kbrow[NR], w[NR];
                static_for<0, NR>([&](auto c) { kbrow[c] = KB(0, c); });
                stage_vjp(u_n, kbrow, w);
            }
            static_for<0, NR>([&](auto c) { ubar[c] = un[c]; });
        }
        while (si >= 0) {  // saves at t0 (save_start)
            if (tg.SV(p, si) == tg.T0(p)) static_for<0, NR>([&](auto c) { if (cvalid(c)) ubar[c] += COT(si, c); });
            si -= 1;
        }
        if (r == 0 && p.stats) p.stats[(size_t)gid * 8 + 4] = nvjp;
        if (p.grad_u0)
            static_for<0, NR>([&](auto c) { if (cwrite(c)) p.grad_u0[(size_t)gid * n + comp(c)] = ubar[c]; });
        if constexpr (DDEF) {
            asm volatile("" ::: "memory");
            Model::dadj_flush(mctx, nst, accg, MSG);
            real* row = p.grad_part + (size_t)part_row<G, BLOCK>() * p.n_param;
#pragma unroll 2
            for (int c = 0; c < NSL; ++c) {
                const int idx = Model::slot_index(p.mc, r, c);
                if (idx >= 0) row[idx] = accg[(size_t)c * MSG];
            }
        } else if constexpr (SG) {
            real* row = p.grad_part + (size_t)part_row<G, BLOCK>() * p.n_param;
            static_for<0, NSL>([&](auto c) {
                const int idx = Model::slot_index(p.mc, r, c);
                if (idx >= 0) row[idx] = acc[c];
            });
        } else {
            static_for<0, NSL>([&](auto c) { acc_lds[c * BLOCK] = acc[c]; });
        }
    }
    if constexpr (PT && per_member<Model>::v && !SG) {
        if (p.theta_pm) {   // per-member parameters: one gradient row per trajectory
            if (in_range) {
                real* rowm = p.grad_part + (size_t)gid * p.n_param;
                for (int s = 0; s < NSL; ++s) {
                    const int idx = Model::slot_index(p.mc, r, s);
                    if (idx >= 0) rowm[idx] = ok ? acc_lds[(size_t)s * BLOCK] : real(0);
                }
            }
            return;
        }
    }
    // ---- per-wave partial gradient row (fixed order) ----
    __syncthreads();
    const int64_t wave = part_row<G, BLOCK>();
    real* row = p.grad_part + (size_t)wave * p.n_param;
    if constexpr (!SG) {
        if ((int)threadIdx.x < G) {
            for (int s = 0; s < NSL; ++s) {
                const int idx = Model::slot_index(p.mc, (int)threadIdx.x, s);
                if (idx >= 0) {
                    real a = 0.0;
                    for (int gq = 0; gq < GROUPS; ++gq) a += slots[s * BLOCK + gq * G + (int)threadIdx.x];
                    row[idx] = a;
                }
            }
        }
    }
}

}  // namespace ude
