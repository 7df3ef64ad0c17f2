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

struct KParams {
    int64_t N;        // trajectories
    int64_t Npad;     // stride of the SoA workspaces
    int32_t ns, cap, n_state, n_param;
    double t0, tf;
    Opts o;
    ModelConsts mc;
    const double* u0;      // n x N
    const double* theta;   // np
    const double* saveat;  // ns
    double* u_out;         // n x ns x N or null
    int64_t* stats;        // 8 x N or null
    int32_t* retcode;      // N
    // dense forward store (SoA, field-major: [(step*NF + field)*Npad + traj]); null for plain solves
    double* dense;
    int32_t* dense_n;
    // loss / cotangent
    const double* data;       // n x ns x N or null
    const uint8_t* row_mask;  // n or null
    const double* cot_in;     // n x ns x N user cotangent or null
    double* cot;              // SoA [(i*n + c)*Npad + traj] written by the forward kernel when data != null
    double* loss_traj;        // N
    // backward outputs
    double* grad_part;  // [nwaves_total][np] per-wave partial gradients
    double* grad_u0;    // n x N or null
    // debugging: per-iteration trace (t, dt, EEst, q, accept) of one trajectory; fwd rows first, then bwd
    double* trace;      // [2][trace_cap][5] or null
    int64_t trace_traj;
    int32_t trace_cap;
};

__device__ __forceinline__ double ulp_of(double x) {
    x = fabs(x);
    return __longlong_as_double(__double_as_longlong(x) + 1) - x;
}

// ---------------------------------------------------------------------------------------------
// generic driver
// ---------------------------------------------------------------------------------------------
template <class Tab, class Sys, int G>
struct Driver {
    static constexpr int NR = Sys::NR, NSL = Sys::NSL, NSLA = NSL > 0 ? NSL : 1;
    static constexpr int S = Tab::S, NK = Tab::NK;
    static constexpr bool USE_FSAL = Tab::FSAL && !Sys::ALWAYS_K0;

    struct Stats {
        int64_t nf = 0, nacc = 0, nrej = 0, nlazy = 0;
    };

    // z: replicated state; mu: slot state.  Integrates from t0 along tdir through sys' tstops.
    static __device__ __forceinline__ int run(Sys& sys, const Opts& o, double (&z)[NR], double (&mu)[NSLA],
                                              double t0, double tdir, double inv_ntot, Stats& st) {
        double k[NK][NR];
        double accb[NSLA], acce[NSLA];
        double t = t0, dt, qold = o.qoldinit, q11 = 1.0;
        bool accept = true, done = false;
        int iter = 0, ret = RET_SUCCESS;
        double tstop = sys.first_tstop();

        // ---- initial dt (ode_determine_initdt), 2 evals ----
        if (o.dt0 > 0.0) {
            dt = tdir * o.dt0;
            if constexpr (USE_FSAL) {
                double gs[NSLA];
                sys.eval(t, z, k[0], gs);
            }
            if constexpr (Tab::FSAL) st.nf += 1;
        } else {
            double gs0[NSLA], f1[NR], gs1[NSLA], z1[NR];
            sys.eval(t, z, k[0], gs0);
            double s0 = 0.0, s1 = 0.0;
            static_for<0, NR>([&](auto c) {
                const double sk = o.abstol + fabs(z[c]) * o.reltol;
                const double q0 = z[c] / sk, q1 = k[0][c] / sk;
                s0 += q0 * q0;
                s1 += q1 * q1;
            });
            if constexpr (NSL > 0) {
                double p0 = 0.0, p1 = 0.0;
                static_for<0, NSL>([&](auto c) {
                    const double sk = o.abstol + fabs(mu[c]) * o.reltol;
                    const double q0 = mu[c] / sk, q1 = gs0[c] / sk;
                    p0 += q0 * q0;
                    p1 += q1 * q1;
                });
                s0 += group_sum<G>(p0);
                s1 += group_sum<G>(p1);
            }
            const double d0 = sqrt(s0 * inv_ntot), d1 = sqrt(s1 * inv_ntot);
            if (d1 != d1) {
                ret = RET_UNSTABLE;
                done = true;
            }
            double dt0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : (d0 / d1) / 100.0;
            dt0 = fmin(dt0, o.dtmax);
            if (dt0 < 10.0 * 2.220446049250313e-16) {
                dt = tdir * 1e-6;
            } else {
                const double dt0t = tdir * dt0;
                static_for<0, NR>([&](auto c) { z1[c] = z[c] + dt0t * k[0][c]; });
                // (the slot part of u1 does not enter f: mu' is independent of mu)
                sys.eval(t + dt0t, z1, f1, gs1);
                double s2 = 0.0;
                static_for<0, NR>([&](auto c) {
                    const double sk = o.abstol + fabs(z[c]) * o.reltol;
                    const double q = (f1[c] - k[0][c]) / sk;
                    s2 += q * q;
                });
                if constexpr (NSL > 0) {
                    double p2 = 0.0;
                    static_for<0, NSL>([&](auto c) {
                        const double sk = o.abstol + fabs(mu[c]) * o.reltol;
                        const double q = (gs1[c] - gs0[c]) / sk;
                        p2 += q * q;
                    });
                    s2 += group_sum<G>(p2);
                }
                const double d2 = sqrt(s2 * inv_ntot) / dt0;
                const double mx = fmax(d1, d2);
                double dt1;
                if (mx <= 1e-15) dt1 = fmax(1e-6, dt0 * 1e-3);
                else dt1 = pow(10.0, -(2.0 + log10(mx)) / (double)Tab::ORDER);
                dt = tdir * fmin(fmin(100.0 * dt0, dt1), o.dtmax);
            }
            st.nf += 2;
            if constexpr (Tab::FSAL) st.nf += 1;  // initialize!: fsalfirst = f(u0) (same value, reused)
        }

        while (!done) {
            // ---- loopheader! ----
            if (iter > 0 && !accept) dt = dt / fmin(1.0 / o.qmin, q11 / o.gamma);  // step_reject_controller!
            iter += 1;
            if (fabs(dt) > o.dtmax) dt = tdir * o.dtmax;
            {
                const double rem = fabs(tstop - t);  // modify_dt_for_tstops!
                if (fabs(dt) > rem) dt = tdir * rem;
            }
            if (iter > o.maxiters) { ret = RET_MAXITERS; break; }
            if (dt != dt) { ret = RET_UNSTABLE; break; }
            if (fabs(dt) <= 2.220446049250313e-16 * fabs(t) && fabs(dt) < fabs(tstop - t)) { ret = RET_DTLESSTHANMIN; break; }

            // ---- perform_step! ----
            {
                double gs[NSLA];
                if constexpr (!USE_FSAL) sys.eval(t, z, k[0], gs);
                else if constexpr (NSL > 0) sys.fsal_slots(gs);
                static_for<0, NSL>([&](auto c) {
                    accb[c] = (dt * Tab::B(0)) * gs[c];
                    acce[c] = (dt * Tab::BT(0)) * gs[c];
                });
            }
            double znew[NR];
            static_for<1, S>([&](auto sc) {
                constexpr int s = sc;
                double zs[NR], gs[NSLA];
                static_for<0, NR>([&](auto c) {
                    double acc = 0.0;
                    static_for<0, s>([&](auto j) {
                        if constexpr (Tab::A(s, j) != 0.0) acc += Tab::A(s, j) * k[j][c];
                    });
                    zs[c] = z[c] + dt * acc;
                });
                if constexpr (Tab::FSAL && s == S - 1) static_for<0, NR>([&](auto c) { znew[c] = zs[c]; });
                sys.eval(t + Tab::C(s) * dt, zs, k[s], gs);
                static_for<0, NSL>([&](auto c) {
                    if constexpr (Tab::B(s) != 0.0) accb[c] = __builtin_fma(dt * Tab::B(s), gs[c], accb[c]);
                    if constexpr (Tab::BT(s) != 0.0) acce[c] = __builtin_fma(dt * Tab::BT(s), gs[c], acce[c]);
                });
                if constexpr (USE_FSAL && NSL > 0 && s == S - 1) sys.store_fsal_slots(gs);
            });
            st.nf += Tab::FSAL ? S - 1 : S;
            if constexpr (!Tab::FSAL) {
                static_for<0, NR>([&](auto c) {
                    double acc = 0.0;
                    static_for<0, S>([&](auto j) {
                        if constexpr (Tab::B(j) != 0.0) acc += Tab::B(j) * k[j][c];
                    });
                    znew[c] = z[c] + dt * acc;
                });
            }
            // calculate_residuals + ODE_DEFAULT_NORM
            double ss = 0.0;
            static_for<0, NR>([&](auto c) {
                double acc = 0.0;
                static_for<0, S>([&](auto j) {
                    if constexpr (Tab::BT(j) != 0.0) acc += Tab::BT(j) * k[j][c];
                });
                const double res = (dt * acc) / (o.abstol + fmax(fabs(z[c]), fabs(znew[c])) * o.reltol);
                ss += res * res;
            });
            if constexpr (NSL > 0) {
                double ps = 0.0;
                static_for<0, NSL>([&](auto c) {
                    const double res = acce[c] / (o.abstol + fmax(fabs(mu[c]), fabs(mu[c] + accb[c])) * o.reltol);
                    ps += res * res;
                });
                ss += group_sum<G>(ps);
            }
            const double EEst = sqrt(ss * inv_ntot);

            // ---- loopfooter!: PIController ----
            double q;
            if (EEst == 0.0) {
                q = 1.0 / o.qmax;
            } else {
                q11 = fastpow(EEst, o.beta1);
                q = q11 / fastpow(qold, o.beta2);
                q = q / o.gamma;
                q = fmin(q, 1.0 / o.qmin);  // NaN-safe ordering not needed: NaN EEst is rejected below
                q = fmax(q, 1.0 / o.qmax);
            }
            accept = (EEst <= 1.0);
            sys.trace(iter, t, dt, EEst, q, accept);
            if (accept) {
                st.nacc += 1;
                qold = fmax(EEst, o.qoldinit);
                double dtnew = dt / q;
                const double tprev = t;
                const double ttmp = t + dt;
                t = fabs(ttmp - tstop) < 100.0 * ulp_of(fmax(t, tstop)) ? tstop : ttmp;
                if (fabs(dtnew) > o.dtmax) dtnew = tdir * o.dtmax;
                // hook: saveat interpolation / dense store (forward); may build the lazy stages
                {
                    bool lazy_done = false;
                    auto lazy = [&]() {
                        if constexpr (Tab::NEXTRA > 0) {
                            if (!lazy_done) {
                                static_for<0, Tab::NEXTRA>([&](auto ec) {
                                    constexpr int e = ec;
                                    double zs[NR], gs[NSLA];
                                    static_for<0, NR>([&](auto c) {
                                        double acc = 0.0;
                                        static_for<0, S + e>([&](auto j) {
                                            if constexpr (Tab::AE(e, j) != 0.0) acc += Tab::AE(e, j) * k[j][c];
                                        });
                                        zs[c] = z[c] + dt * acc;
                                    });
                                    sys.eval(tprev + Tab::CE(e) * dt, zs, k[S + e], gs);
                                });
                                lazy_done = true;
                                st.nlazy += Tab::NEXTRA;
                            }
                        }
                    };
                    const int hr = sys.accepted(tprev, t, dt, z, znew, k, lazy);
                    if (hr != RET_SUCCESS) { ret = hr; done = true; }
                }
                dt = dtnew;
                bool bad = false;
                static_for<0, NR>([&](auto c) {
                    z[c] = znew[c];
                    bad = bad || (znew[c] != znew[c]);
                });
                static_for<0, NSL>([&](auto c) { mu[c] += accb[c]; });
                if constexpr (USE_FSAL) static_for<0, NR>([&](auto c) { k[0][c] = k[S - 1][c]; });
                if (bad) { ret = RET_UNSTABLE; done = true; }
                if (t == tstop) {  // handle_tstop! + callbacks
                    const bool modified = sys.at_tstop(t, z);
                    bool more = sys.next_tstop(tstop);
                    if (!more) done = true;
                    else if (modified) {
                        if constexpr (Tab::FSAL) st.nf += 1;  // reset_fsal! after u_modified!
                        if constexpr (USE_FSAL) {
                            double gs[NSLA];
                            sys.eval(t, z, k[0], gs);
                            if constexpr (NSL > 0) sys.store_fsal_slots(gs);
                        }
                    }
                }
            } else {
                st.nrej += 1;
                if (EEst != EEst) { ret = RET_UNSTABLE; done = true; }
            }
        }
        return ret;
    }
};

// ---------------------------------------------------------------------------------------------
// forward system: model RHS + saveat (savevalues!) + dense store + loss/cotangent
// dense field layout per step: 0 t_start, 1 t_end, 2..2+NS u_start, then k[q][c]
// ---------------------------------------------------------------------------------------------
template <class Model, class Tab, int G>
struct FwdSys {
    static constexpr int NR = Model::NS, NSL = 0;
    static constexpr bool ALWAYS_K0 = false;
    static constexpr int NF = 2 + NR + Tab::NK * NR;
    typename Model::Ctx mctx;
    const KParams* p;
    int64_t j;      // trajectory (clamped)
    bool writer;    // lane 0 of an in-range group
    int si, nsteps;
    double loss;

    __device__ __forceinline__ double first_tstop() const { return p->tf; }
    __device__ __forceinline__ bool next_tstop(double&) const { return false; }
    __device__ __forceinline__ bool at_tstop(double, double*) const { return false; }
    __device__ __forceinline__ void eval(double, const double* z, double* kr, double*) { Model::rhs(mctx, z, kr); }
    __device__ __forceinline__ void trace(int iter, double t, double dt, double e, double q, bool acc) const {
        if (p->trace && writer && j == p->trace_traj && iter <= p->trace_cap) {
            double* row = p->trace + (size_t)(iter - 1) * 5;
            row[0] = t; row[1] = dt; row[2] = e; row[3] = q; row[4] = acc ? 1.0 : 0.0;
        }
    }
    __device__ __forceinline__ void fsal_slots(double*) {}
    __device__ __forceinline__ void store_fsal_slots(const double*) {}

    __device__ __forceinline__ void save_point(int i, const double* v) {
        const int n = NR;
        if (p->u_out && writer) {
            double* dst = p->u_out + ((size_t)j * p->ns + i) * n;
            static_for<0, NR>([&](auto c) { dst[c] = v[c]; });
        }
        if (p->data) {
            const double* d = p->data + ((size_t)j * p->ns + i) * n;
            static_for<0, NR>([&](auto c) {
                const double on = (p->row_mask && !p->row_mask[c]) ? 0.0 : 1.0;
                const double e = on * (v[c] - d[c]);
                loss += e * e;
                if (writer) p->cot[((size_t)i * n + c) * p->Npad + j] = 2.0 * e;
            });
        }
    }

    template <class Lazy>
    __device__ __forceinline__ int accepted(double tprev, double t, double dt, const double* z, const double* znew,
                                            double (&k)[Tab::NK][NR], Lazy& lazy) {
        while (si < p->ns && p->saveat[si] <= t) {
            const double curt = p->saveat[si];
            if (curt != t) {
                lazy();
                const double th = (curt - tprev) / dt;
                double b[Tab::NK], y[NR];
                Tab::bth(th, b);
                static_for<0, NR>([&](auto c) {
                    double acc = 0.0;
                    static_for<0, Tab::NK>([&](auto q) {
                        if constexpr (Tab::dense_uses(q)) acc += k[q][c] * b[q];
                    });
                    y[c] = z[c] + dt * acc;
                });
                save_point(si, y);
            } else {
                save_point(si, znew);
            }
            si += 1;
        }
        if (p->dense) {
            if (nsteps >= p->cap) return RET_DENSE_OVERFLOW;
            lazy();
            if (writer) {
                double* base = p->dense + ((size_t)nsteps * NF) * p->Npad + j;
                base[0] = tprev;
                base[(size_t)1 * p->Npad] = t;
                static_for<0, NR>([&](auto c) { base[(size_t)(2 + c) * p->Npad] = z[c]; });
                static_for<0, Tab::NK>([&](auto q) {
                    if constexpr (Tab::dense_uses(q))
                        static_for<0, NR>([&](auto c) { base[(size_t)(2 + NR + q * NR + c) * p->Npad] = k[q][c]; });
                });
            }
            nsteps += 1;
        }
        return RET_SUCCESS;
    }
};

template <class Model, class Tab, int G, int BLOCK>
__global__ void __launch_bounds__(BLOCK) fwd_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* th = reinterpret_cast<double*>(smem_raw);
    for (int i = threadIdx.x; i < p.n_param; i += BLOCK) th[i] = p.theta[i];
    __syncthreads();

    const int64_t gid = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) / G;
    const int r = threadIdx.x % G;
    if (gid >= p.N) return;  // whole groups leave together
    using Sys = FwdSys<Model, Tab, G>;
    using Drv = Driver<Tab, Sys, G>;
    Sys sys;
    Model::init(sys.mctx, th, p.mc, r);
    sys.p = &p;
    sys.j = gid;
    sys.writer = (r == 0);
    sys.si = 0;
    sys.nsteps = 0;
    sys.loss = 0.0;
    double z[Sys::NR], mu[1] = {0.0};
    static_for<0, Sys::NR>([&](auto c) { z[c] = p.u0[(size_t)gid * Sys::NR + c]; });
    while (sys.si < p.ns && p.saveat[sys.si] <= p.t0) {  // save_start
        sys.save_point(sys.si, z);
        sys.si += 1;
    }
    typename Drv::Stats st;
    const int ret = Drv::run(sys, p.o, z, mu, p.t0, 1.0, 1.0 / (double)Sys::NR, st);
    if (sys.writer) {
        if (p.stats) {
            int64_t* s = p.stats + (size_t)gid * 8;
            s[0] = st.nf; s[1] = st.nacc; s[2] = st.nrej;
            if (p.dense) { s[3] = 0; s[7] = st.nlazy; } else { s[3] = st.nlazy; s[7] = 0; }
            s[4] = 0; s[5] = 0; s[6] = 0;
        }
        p.retcode[gid] = ret;
        if (p.dense_n) p.dense_n[gid] = sys.nsteps;
        if (p.loss_traj) p.loss_traj[gid] = ret == RET_SUCCESS ? sys.loss : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// adjoint system (InterpolatingAdjoint): z = lambda (replicated), slots = mu
//   lambda' = -(df/du)^T lambda, mu' = -(df/dtheta)^T lambda at y(t) = forward dense interpolant;
//   save times are tstops with lambda += dL/du(t_i)            
// ---------------------------------------------------------------------------------------------
template <class Model, class Tab, int G>
struct AdjSys {
    static constexpr int NR = Model::NS, NSL = Model::NSL;
    static constexpr bool ALWAYS_K0 = true;  // stage 0 re-evaluated every step: no FSAL slot storage, uniform flow
    static constexpr int NF = 2 + NR + Tab::NK * NR;
    typename Model::Ctx mctx;
    const KParams* p;
    int64_t j;
    int nsteps, sf, cur;
    // cached forward interval
    double ts, te, us[NR], ks[Tab::NK][NR];
    // cotangent access
    const double* cot;
    size_t cot_si, cot_sc;  // strides of save index / component

    __device__ __forceinline__ void load_interval(int s) {
        sf = s;
        const double* base = p->dense + ((size_t)s * NF) * p->Npad + j;
        ts = base[0];
        te = base[(size_t)1 * p->Npad];
        static_for<0, NR>([&](auto c) { us[c] = base[(size_t)(2 + c) * p->Npad]; });
        static_for<0, Tab::NK>([&](auto q) {
            if constexpr (Tab::dense_uses(q))
                static_for<0, NR>([&](auto c) { ks[q][c] = base[(size_t)(2 + NR + q * NR + c) * p->Npad]; });
        });
    }
    // sol(t, continuity = :right): interval [s, s+1] with t_s <= t, clamped to the stored range
    __device__ __forceinline__ void locate(double t) {
        while (t < ts && sf > 0) load_interval(sf - 1);
        while (t >= te && sf < nsteps - 1) load_interval(sf + 1);
    }
    __device__ __forceinline__ void eval(double t, const double* lam, double* klam, double* g) {
        locate(t);
        const double dtf = te - ts;
        const double th = (t - ts) / dtf;
        double b[Tab::NK], y[NR], dl[NR];
        Tab::bth(th, b);
        static_for<0, NR>([&](auto c) {
            double acc = 0.0;
            static_for<0, Tab::NK>([&](auto q) {
                if constexpr (Tab::dense_uses(q)) acc += ks[q][c] * b[q];
            });
            y[c] = us[c] + dtf * acc;
        });
        Model::template vjp<true>(mctx, y, lam, dl, g);
        static_for<0, NR>([&](auto c) { klam[c] = -dl[c]; });
        static_for<0, NSL>([&](auto c) { g[c] = -g[c]; });
    }
    __device__ __forceinline__ void fsal_slots(double*) {}
    __device__ __forceinline__ void store_fsal_slots(const double*) {}
    __device__ __forceinline__ void trace(int iter, double t, double dt, double e, double q, bool acc) const {
        if (p->trace && mctx.r == 0 && j == p->trace_traj && iter <= p->trace_cap) {
            double* row = p->trace + ((size_t)p->trace_cap + (iter - 1)) * 5;
            row[0] = t; row[1] = dt; row[2] = e; row[3] = q; row[4] = acc ? 1.0 : 0.0;
        }
    }

    __device__ __forceinline__ double tstop_from_cur() const {
        // next save time strictly inside (t0, t) in descending order, else t0
        return (cur >= 0 && p->saveat[cur] > p->t0) ? p->saveat[cur] : p->t0;
    }
    __device__ __forceinline__ double first_tstop() const { return tstop_from_cur(); }
    __device__ __forceinline__ bool at_tstop(double t, double* lam) {
        bool mod = false;
        while (cur >= 0 && p->saveat[cur] >= t) {
            if (p->saveat[cur] == t) {
                static_for<0, NR>([&](auto c) { lam[c] += cot[(size_t)cur * cot_si + (size_t)c * cot_sc]; });
                mod = true;
            }
            cur -= 1;
        }
        return mod;
    }
    __device__ __forceinline__ bool next_tstop(double& tstop) {
        if (tstop == p->t0) return false;
        tstop = tstop_from_cur();
        return true;
    }
    template <class Lazy>
    __device__ __forceinline__ int accepted(double, double, double, const double*, const double*,
                                            double (&)[Tab::NK][NR], Lazy&) {
        return RET_SUCCESS;
    }
};

template <class Model, class Tab, int G, int BLOCK>
__global__ void __launch_bounds__(BLOCK) adj_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* th = reinterpret_cast<double*>(smem_raw);
    for (int i = threadIdx.x; i < p.n_param; i += BLOCK) th[i] = p.theta[i];
    __syncthreads();

    const int64_t gid = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) / G;
    const int r = threadIdx.x % G;
    using Sys = AdjSys<Model, Tab, G>;
    using Drv = Driver<Tab, Sys, G>;
    constexpr int NSL = Sys::NSL;
    double lam[Sys::NR], mu[NSL];
    static_for<0, Sys::NR>([&](auto c) { lam[c] = 0.0; });
    static_for<0, NSL>([&](auto c) { mu[c] = 0.0; });
    const bool in_range = gid < p.N;
    bool ok = in_range && p.retcode[in_range ? gid : 0] == RET_SUCCESS;
    if (ok) {
        Sys sys;
        Model::init(sys.mctx, th, p.mc, r);
        sys.p = &p;
        sys.j = gid;
        sys.nsteps = p.dense_n[gid];
        if (p.cot_in) {
            sys.cot = p.cot_in + (size_t)gid * p.ns * Sys::NR;
            sys.cot_si = Sys::NR;
            sys.cot_sc = 1;
        } else {
            sys.cot = p.cot + gid;
            sys.cot_si = (size_t)Sys::NR * p.Npad;
            sys.cot_sc = p.Npad;
        }
        sys.cur = p.ns - 1;
        sys.load_interval(sys.nsteps - 1);
        sys.at_tstop(p.tf, lam);  // init_cb: the jump at t = tf precedes the first step
        typename Drv::Stats st;
        const int ret = Drv::run(sys, p.o, lam, mu, p.tf, -1.0, 1.0 / (double)(Sys::NR + p.n_param), st);
        if (r == 0) {
            if (p.stats) {
                int64_t* s = p.stats + (size_t)gid * 8;
                s[4] = st.nf; s[5] = st.nacc; s[6] = st.nrej;
            }
            if (ret != RET_SUCCESS) p.retcode[gid] = ret;
            if (p.grad_u0) static_for<0, Sys::NR>([&](auto c) { p.grad_u0[(size_t)gid * Sys::NR + c] = lam[c]; });
        }
        if (ret != RET_SUCCESS) static_for<0, NSL>([&](auto c) { mu[c] = 0.0; });  // never poison the batch gradient
    }
    // ---- deterministic reduction: groups of a wave (xor butterfly), then one partial row per wave ----
    static_for<0, NSL>([&](auto c) {
        double v = mu[c];
#pragma unroll
        for (int m = G; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
        mu[c] = v;
    });
    const int lane = threadIdx.x & 63;
    if (lane < G) {
        const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) / 64;
        double* row = p.grad_part + (size_t)wave * p.n_param;
        for (int s = 0; s < NSL; ++s) {
            const int idx = Model::slot_index(p.mc, lane, s);
            if (idx >= 0) {
                double v = 0.0;
                static_for<0, NSL>([&](auto c) { v = (s == c) ? mu[c] : v; });
                row[idx] = v;
            }
        }
    }
}

}  // namespace ude
