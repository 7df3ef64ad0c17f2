// ude_seir_ls_fwd.h -- the FORWARD solve of the SEIR exposure UDE (dudt_, SEIR_exposure/seir_exposure.jl:114-147) on the
// architecture of ude_seir_ls.h: sixteen trajectory slots of a block step in lock-step as the columns of v_mfma_f64_16x16x4.
//
// fwd_kernel<SeirUde<64>> spends a wavefront per trajectory (64 dependent fma's per lane behind 64 LDS broadcast reads for the
// hidden layer).  Here one trip of a persistent block is ONE right-hand side of every busy slot, whatever it is for: the two
// evaluations of the initial-dt heuristic, a stage of a step attempt, or one of Vern7's six lazy dense-output stages of an
// accepted step.  The network runs on the matrix cores exactly as in the backward kernel (first layer: one MFMA whose fourth
// k-step adds the bias; hidden layer: four 16-term chains added left to right = the wide-dot rule; output layer: rounded products
// w3 a2 reduced by the adjacent-pair tree on the slot's 16-lane row), everything else -- stage combinations, error norm, PI
// controller, save-point interpolation, the dense record the adjoint reads -- is the Driver's / FwdSys's own sequence on the
// slot's row, component c on lane c.  Per trajectory every number (saved states, step counts, dense store, loss and cotangent
// rows) is bit-identical to fwd_kernel<SeirUde<64>> and to the oracle.  Float64, shared time grid.
#pragma once
#ifndef UDE_LS_FWD_PER_CU
#define UDE_LS_FWD_PER_CU 1   // resident blocks of the forward lock-step kernel per compute unit (launch bounds AND grid size)
#endif
#include "ude_seir_ls.h"

namespace ude {
namespace seirls {

enum { FPH_IDLE = -4, FPH_FSAL0 = -3, FPH_INIT0 = -2, FPH_INIT1 = -1 };   // >= 0: stage s; s >= S: lazy dense-output stage s - S

template <class Tab>
constexpr int fwd_lds_doubles() { return H * TLD + 4 * 16 + NSLOTS * PLD + TABL + NSLOTS * Tab::NK * 16 + 16 + H; }

// GEN = true (round 5): the RUNTIME-SHAPE instance -- any exposure-UDE chain 3 -> H1 -> H2 -> 1 (tanh, tanh, identity), H1, H2 <= 64 (the set
// udecore.hip's seir_gen_ls_fwd_shape admits: no 32- / 64-term product with fewer than 16 results), weights zero-padded to 64 x 64, every product in the ORACLE'S association for its length (ude_seir_ls2.h): the hidden
// product four 16-term chains for H1 == 64 and ONE ascending chain otherwise, the output layer the adjacent-pair tree of rounded products
// for H2 == 64 and one ascending fma chain over the units otherwise
template <class Tab, bool GEN = false>
__global__ void __launch_bounds__(BLOCKT, UDE_LS_FWD_PER_CU) seir_ls_fwd_kernel(const KParams p, int* __restrict__ queue) {
    constexpr int S = Tab::S, NK = Tab::NK, NX = Tab::NEXTRA;
    constexpr int FIRST = Tab::FSAL ? 1 : 0;   // first stage an attempt evaluates (FSAL: stage 0 is handed over)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* T_A1 = sm;                       // [64][17]
    double* XIN = T_A1 + H * TLD;            // [4][16]: x0 x1 x2 1
    double* PG = XIN + 4 * 16;               // [16][65]: w3[i] a2[i] of slot (column) and hidden row i
    double* TB = PG + NSLOTS * PLD;          // tableau: A[16][16], B, BT, C
    double* KSL = TB + TABL;                 // [16 slots][NK][16]: stage derivatives, component c of the slot on lane c of its row
    double* W3L = KSL + NSLOTS * NK * 16 + 16;   // [64] w3 (GEN with H2 < 64: the output layer's chain reads it)

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int kq = l >> 4, jc = l & 15;      // matrix view
    const int rr = l >> 4, lm = l & 15;      // scalar view: slot 4w + rr, lane lm of its row
    const int slot = 4 * w + rr;
    const double* __restrict__ th = p.theta;
    const TabDev* __restrict__ tab = p.tab;
    const int n = NC;

    const int H1 = GEN ? p.mc.dims[1] : H, H2 = GEN ? p.mc.dims[2] : H;
    const int oW1 = 0, oB1 = 3 * H1, oW2 = oB1 + H1, oB2 = oW2 + H1 * H2, oW3 = oB2 + H2, oB3 = oW3 + H2;
    const bool blk_hid = H1 == H, tree_out = H2 == H;
    double W2A[16];
    {
        const int row = 16 * w + jc;
        static_for<0, 16>([&](auto sc) {
            const int col = 4 * decltype(sc)::value + kq;
            W2A[sc] = (row < H2 && col < H1) ? th[oW2 + row + col * H2] : 0.0;
        });
    }
    const double W1A = (16 * w + jc) < H1 ? (kq < 3 ? th[oW1 + (16 * w + jc) + kq * H1] : th[oB1 + 16 * w + jc]) : 0.0;
    double b2r[4], w3r[4];
    static_for<0, 4>([&](auto r) {
        const int row = 16 * w + kq + 4 * decltype(r)::value;
        b2r[r] = row < H2 ? th[oB2 + row] : 0.0;
        w3r[r] = row < H2 ? th[oW3 + row] : 0.0;
    });
    const double b3c = th[oB3];
    if (tid < H) W3L[tid] = tid < H2 ? th[oW3 + tid] : 0.0;
    const double Fc = p.mc.consts[0], b0c = p.mc.consts[1], muc = p.mc.consts[4], sgc = p.mc.consts[5], gac = p.mc.consts[6],
                 dc = p.mc.consts[7], lac = p.mc.consts[8];
    if (tid < 16) XIN[3 * 16 + tid] = 1.0;
    for (int i = tid; i < 16 * 16; i += BLOCKT) TB[i] = tab->A[i >> 4][i & 15];
    if (tid < 16) { TB[256 + tid] = tab->B[tid]; TB[272 + tid] = tab->BT[tid]; TB[288 + tid] = tab->C[tid]; }

    const OptsR o(p.o);
    const double T0 = p.t0, TF = p.tf, tdir = 1.0;
    const double dtmax = o.dtmax;
    const double ntot = (double)p.n_state;
    const bool user_dt = o.dt0 > 0.0;
    int ph = FPH_IDLE, ret = RET_SUCCESS, iter = 0, si = 0, nsteps = 0;
    long long gid = 0;
    bool accept = true, exhausted = false, fresh = false;
    double t = T0, dt = 0.0, dt0 = 0.0, d1n = 0.0, qold = o.qoldinit, q11 = 1.0, tprev = T0, dtnew = 0.0, loss = 0.0;
    int nfc = 0, nacc = 0, nrej = 0, nlazy = 0;
    double zo = 0.0, zn = 0.0;   // this lane's component of the state and of the candidate state (zn: f0 during the initial-dt heuristic)
    double* const K = KSL + (size_t)(4 * w + (l >> 4)) * NK * 16 + (l & 15);   // K[16 q]: stage q (LDS: sixteen doubles per lane less in registers)
    static_for<0, NK>([&](auto q) { K[16 * decltype(q)::value] = 0.0; });

    auto own = [&](const double (&v)[NC]) {
        double r = 0.0;
        static_for<0, NC>([&](auto c) { r = (lm == (int)decltype(c)::value) ? v[c] : r; });
        return r;
    };
    auto bcast = [&](double ownv, double (&out)[NC]) { static_for<0, NC>([&](auto c) { out[c] = rshfl(ownv, decltype(c)::value); }); };
    auto SV = [&](int i) { return p.saveat[i]; };
    // FwdSys::save_point: a saved state, its loss term and the cotangent row the adjoint will jump by
    auto save_point = [&](int i, const double (&v)[NC]) {
        if (p.u_out && lm < NC) p.u_out[((size_t)gid * p.ns + i) * n + lm] = own(v);
        if (p.data) {
            const double* d = p.data + ((size_t)gid * p.ns + i) * n;
            static_for<0, NC>([&](auto c) {
                constexpr int ci = decltype(c)::value;
                const double e = (p.row_mask && !p.row_mask[ci]) ? 0.0 : (v[c] - d[ci]);
                loss = __builtin_fma(e, e, loss);
                if (lm == ci) p.cot[((size_t)i * n + ci) * p.Npad + gid] = 2.0 * e;
            });
        }
    };
    // the end of a trajectory: counters, return code, number of dense records, its loss
    auto finish = [&]() {
        if (lm == 0) {
            if (p.stats) {
                int64_t* s = p.stats + (size_t)gid * 8;
                s[0] = nfc; s[1] = nacc; s[2] = nrej;
                if (p.dense) { s[3] = 0; s[7] = nlazy; } else { s[3] = nlazy; s[7] = 0; }
                s[4] = 0; s[5] = 0; s[6] = 0;
            }
            p.retcode[gid] = ret;
            if (p.dense_n) p.dense_n[gid] = nsteps;
            if (p.loss_traj) p.loss_traj[gid] = ret == RET_SUCCESS ? loss : 0.0;
        }
        ph = FPH_IDLE;
    };
    __syncthreads();

    for (;;) {
        // ---- A. an idle slot takes the next trajectory ----
        if (ph == FPH_IDLE && !exhausted) {
            int g = 0;
            if (lm == 0) g = atomicAdd(queue, 1);
            g = __shfl(g, 0, 16);
            if (g >= p.N) exhausted = true;
            else {
                gid = g;
                zo = lm < NC ? p.u0[(size_t)gid * n + lm] : 0.0;
                zn = 0.0;
                static_for<0, NK>([&](auto q) { K[16 * decltype(q)::value] = 0.0; });
                t = T0; qold = o.qoldinit; q11 = 1.0; accept = true; iter = 0; ret = RET_SUCCESS;
                nfc = 0; nacc = 0; nrej = 0; nlazy = 0; si = 0; nsteps = 0; loss = 0.0;
                while (si < p.ns && SV(si) <= T0) {   // save_start
                    double y[NC];
                    bcast(zo, y);
                    save_point(si, y);
                    si += 1;
                }
                if (user_dt) {
                    dt = tdir * o.dt0;
                    if constexpr (Tab::FSAL) ph = FPH_FSAL0;
                    else { ph = 0; fresh = true; }
                } else ph = FPH_INIT0;
            }
        }

        // ---- B. the state this slot's right-hand side is evaluated at ----
        bool ev = false;
        double zs[NC], kr[NC];
        static_for<0, NC>([&](auto c) { kr[c] = 0.0; });
        double zsrc = zo;   // this lane's component of the state the right-hand side is evaluated at
        if (ph == FPH_INIT0 || ph == FPH_FSAL0) {
            ev = true;
        } else if (ph == FPH_INIT1) {
            ev = true;
            const double dt0t = tdir * dt0;
            zsrc = __builtin_fma(dt0t, zn, zo);   // (zn holds f0 during the heuristic)
        } else if (ph >= 0) {
            bool go = true;
            if (fresh) {   // loopheader!
                fresh = false;
                if (iter > 0 && !accept) {
                    double den = q11 / o.gamma;
                    const double iq = 1.0 / o.qmin;
                    if (iq < den) den = iq;
                    dt = dt / den;
                }
                iter += 1;
                if (fabs(dt) > dtmax) dt = tdir * dtmax;
                {
                    const double rem = fabs(TF - t);
                    if (fabs(dt) > rem) dt = tdir * rem;
                }
                if (iter > o.maxiters) { ret = RET_MAXITERS; go = false; }
                else if (dt != dt) { ret = RET_UNSTABLE; go = false; }
                else if (fabs(dt) <= REAL_EPS * fabs(t) && fabs(dt) < fabs(TF - t)) { ret = RET_DTLESSTHANMIN; go = false; }
            }
            if (go) {
                ev = true;
                if (ph > 0) {
                    // all NK - 1 possible terms: coefficients beyond the row's own are zero in the table, the k storage is always
                    // finite, fma(0, k, acc) == acc exactly
                    const double* Ar = TB + ph * 16;
                    double acc = Ar[0] * K[0];
                    static_for<1, NK>([&](auto j) { acc = __builtin_fma(Ar[decltype(j)::value], K[16 * decltype(j)::value], acc); });
                    zsrc = __builtin_fma(dt, acc, zo);
                }
            } else {
                finish();
            }
        }
        bcast(zsrc, zs);
        if (ev && lm == 0) {
            XIN[0 * 16 + slot] = zs[0] / zs[4];
            XIN[1 * 16 + slot] = zs[2];
            XIN[2 * 16 + slot] = zs[5] / zs[4];
        }
        if (!__syncthreads_or(ph != FPH_IDLE)) break;   // all slots idle and the queue empty: done

        // ---- C. the network for all 16 slots ----
        {
            v4d z = __builtin_amdgcn_mfma_f64_16x16x4f64(W1A, XIN[kq * 16 + jc], v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                T_A1[(16 * w + kq + 4 * r) * TLD + jc] = dtanh(z[r]);
            });
            __syncthreads();
            v4d acc[4];
            if (!GEN || blk_hid) {
                static_for<0, 4>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
                    static_for<0, 4>([&](auto q) {
                        constexpr int s = 4 * b + decltype(q)::value;
                        acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2A[s], T_A1[(4 * s + kq) * TLD + jc], acc[b], 0, 0, 0);
                    });
                });
            } else {   // fewer than 64 inputs: one ascending chain
                acc[0] = v4d{0.0, 0.0, 0.0, 0.0};
                static_for<0, 16>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2A[s], T_A1[(4 * s + kq) * TLD + jc], acc[0], 0, 0, 0);
                });
            }
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const double z2 = ((!GEN || blk_hid) ? (((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r]) : acc[0][r]) + b2r[r];
                const double a2 = dtanh(z2);
                PG[jc * PLD + 16 * w + kq + 4 * r] = (!GEN || tree_out) ? w3r[r] * a2 : a2;   // (chain case: the activation itself)
            });
        }
        __syncthreads();

        // ---- D. the slot's row: right-hand side of dudt_ ----
        if (ev) {
            const double* pr = PG + slot * PLD + 4 * lm;
            double zn;
            if (!GEN || tree_out) zn = row_tree4(pr[0], pr[1], pr[2], pr[3]) + b3c;
            else {   // an output layer of fewer than 64 terms: one ascending fma chain over the units (every lane of the row the same chain)
                double acc = 0.0;
                const double* a2s = PG + slot * PLD;
                for (int u = 0; u < H2; ++u) acc = __builtin_fma(W3L[u], a2s[u], acc);
                zn = acc + b3c;
            }
            const double Sv = zs[0], Ev = zs[1], Iv = zs[2], Rv = zs[3], Nv = zs[4], Dv = zs[5];
            kr[0] = -b0c * Sv * Fc / Nv - zn - muc * Sv;
            kr[1] = b0c * Sv * Fc / Nv + zn - (sgc + muc) * Ev;
            kr[2] = sgc * Ev - (gac + muc) * Iv;
            kr[3] = gac * Iv - muc * Rv;
            kr[4] = -muc * Nv;
            kr[5] = dc * gac * Iv - lac * Dv;
            kr[6] = sgc * Ev;
        }

        // ---- F. the slot's state machine ----
        bool finalize = false;
        if (ph == FPH_FSAL0 && ev) {
            K[0] = own(kr);
            nfc += 1;
            ph = FIRST; fresh = true;
        } else if (ph == FPH_INIT0 && ev) {
            K[0] = own(kr);
            zn = own(kr);   // f0
            double h0 = 0.0, l0 = 0.0, h1 = 0.0, l1 = 0.0;
            static_for<0, NC>([&](auto c) {
                const double sk = __builtin_fma(fabs(zs[c]), o.reltol, o.abstol);   // (zs is the state itself in this phase)
                const double q0 = zs[c] / sk, q1 = kr[c] / sk;
                dd_acc(h0, l0, q0 * q0);
                dd_acc(h1, l1, q1 * q1);
            });
            const double s0 = h0 + l0, s1 = h1 + l1;
            const double d0 = __builtin_sqrt(s0 / ntot);
            d1n = __builtin_sqrt(s1 / ntot);
            if (d1n != d1n) ret = RET_UNSTABLE;
            dt0 = (d0 < 1e-5 || d1n < 1e-5) ? 1e-6 : (d0 / d1n) / 100.0;
            if (dt0 > dtmax) dt0 = dtmax;
            if (dt0 < 10.0 * REAL_EPS) {
                dt = tdir * 1e-6;
                nfc += 2;
                if constexpr (Tab::FSAL) nfc += 1;
                if (ret != RET_SUCCESS) finish();
                else { ph = FIRST; fresh = true; }
            } else {
                ph = FPH_INIT1;
            }
        } else if (ph == FPH_INIT1 && ev) {
            double h2 = 0.0, l2 = 0.0;
            double f0[NC], uu[NC];
            bcast(zn, f0);
            bcast(zo, uu);
            static_for<0, NC>([&](auto c) {
                const double sk = __builtin_fma(fabs(uu[c]), o.reltol, o.abstol);
                const double q = (kr[c] - f0[c]) / sk;
                dd_acc(h2, l2, q * q);
            });
            const double s2 = h2 + l2;
            const double d2 = __builtin_sqrt(s2 / ntot) / dt0;
            const double mx = d1n > d2 ? d1n : d2;
            double dt1;
            if (mx <= 1e-15) {
                dt1 = dt0 * 1e-3;
                if (dt1 < 1e-6) dt1 = 1e-6;
            } else {
                const double ex = -(2.0 + rlog10(mx)) / (double)Tab::ORDER;
                dt1 = rpow10(ex);
            }
            double d = 100.0 * dt0;
            if (dt1 < d) d = dt1;
            if (dtmax < d) d = dtmax;
            dt = tdir * d;
            nfc += 2;
            if constexpr (Tab::FSAL) nfc += 1;
            if (ret != RET_SUCCESS) finish();
            else { ph = FIRST; fresh = true; }
        } else if (ph >= 0 && ev) {
            const double ko = own(kr);
            K[16 * ph] = ko;
            if (ph < S - 1) {
                ph += 1;
            } else if (ph == S - 1) {
                // perform_step! is complete: new state, error estimate, controller
                nfc += Tab::FSAL ? S - 1 : S;
                if constexpr (Tab::FSAL) zn = own(zs);
                else {
                    double acc = TB[256] * K[0];
                    static_for<1, S>([&](auto j) { acc = __builtin_fma(TB[256 + decltype(j)::value], K[16 * decltype(j)::value], acc); });
                    zn = __builtin_fma(dt, acc, zo);
                }
                double acc = TB[272] * K[0];
                static_for<1, S>([&](auto j) { acc = __builtin_fma(TB[272 + decltype(j)::value], K[16 * decltype(j)::value], acc); });
                const double a0 = fabs(zo), a1 = fabs(zn);
                double res[NC];
                bcast((dt * acc) / __builtin_fma((a0 > a1 ? a0 : a1), o.reltol, o.abstol), res);
                double ss = 0.0;
                static_for<0, NC>([&](auto c) { ss = __builtin_fma(res[c], res[c], ss); });
                const double EEst = __builtin_sqrt(ss / ntot);
                double q;
                if (EEst == 0.0) {
                    q = 1.0 / o.qmax;
                } else {
                    q11 = fastpow(EEst, o.beta1);
                    q = q11 / fastpow(qold, o.beta2);
                    q = q / o.gamma;
                    const double lo = 1.0 / o.qmax, hi = 1.0 / o.qmin;
                    if (q > hi) q = hi;
                    if (q < lo) q = lo;
                }
                accept = EEst <= 1.0;
                if (p.trace && lm == 0 && gid == p.trace_traj && iter <= p.trace_cap) {
                    double* row = p.trace + (size_t)(iter - 1) * 5;
                    row[0] = t; row[1] = dt; row[2] = EEst; row[3] = q; row[4] = accept ? 1.0 : 0.0;
                }
                if (accept) {
                    nacc += 1;
                    qold = EEst > o.qoldinit ? EEst : o.qoldinit;
                    dtnew = dt / q;
                    tprev = t;
                    const double ttmp = t + dt;
                    {
                        const double mxt = t > TF ? t : TF;
                        t = fabs(ttmp - TF) < 100.0 * ulp_of(mxt) ? TF : ttmp;
                    }
                    if (fabs(dtnew) > dtmax) dtnew = tdir * dtmax;
                    // the lazy dense-output stages are needed by a save point strictly inside the step and by the dense record
                    bool need = false;
                    if constexpr (NX > 0) {
                        need = p.dense != nullptr && nsteps < p.cap;
                        for (int i = si; i < p.ns && SV(i) <= t; ++i) need = need || (SV(i) != t);
                    }
                    if (need) ph = S;
                    else finalize = true;
                } else {
                    nrej += 1;
                    if (EEst != EEst) { ret = RET_UNSTABLE; finish(); }
                    else { ph = FIRST; fresh = true; }
                }
            } else {
                // a lazy stage is in place
                if (ph < S + NX - 1) ph += 1;
                else { nlazy += NX; finalize = true; }
            }
        }
        if (finalize) {
            // FwdSys::accepted: save points inside (tprev, t], the dense record
            bool fin = false;
            while (si < p.ns && SV(si) <= t) {
                const double curt = SV(si);
                if (curt != t) {
                    const double thv = (curt - tprev) / dt;
                    // b_q(theta) by lane q of the row from the tableau's Horner table (bit-identical to Tab::bth; as compile-time
                    // constants the 16 x 7 coefficients of Vern7 would sit in registers for the whole solve)
                    double b[NK], y[NC];
                    {
                        const double* R = tab->R[lm < NK ? lm : 0];
                        double h = R[0];
                        static_for<1, 7>([&](auto i) { h = __builtin_fma(thv, h, R[decltype(i)::value]); });
                        const double bq = (lm == 0 ? thv : thv * thv) * h;
                        static_for<0, NK>([&](auto q) { b[q] = Tab::dense_uses(decltype(q)::value) ? rshfl(bq, decltype(q)::value) : 0.0; });
                    }
                    const double acc = chain2<RowDense<Tab>, NK>([&](auto q) { return K[16 * decltype(q)::value]; }, [&](auto q) { return b[q]; });
                    bcast(__builtin_fma(dt, acc, zo), y);
                    save_point(si, y);
                } else {
                    double y[NC];
                    bcast(zn, y);
                    save_point(si, y);
                }
                si += 1;
            }
            if (p.dense) {
                if (nsteps >= p.cap) { ret = RET_DENSE_OVERFLOW; fin = true; }
                else {
                    const int nf = 3 + n + NK * n;
                    double* base = dense_rec<true>(p, nsteps, nf, gid);   // (record-major: ude_kernels.h)
                    if (lm == 0) {
                        base[0] = tprev;
                        base[1] = t;
                        base[2] = dt;
                    }
                    if (lm < n) {
                        base[3 + lm] = zo;
                        static_for<0, NK>([&](auto q) { base[3 + n + (int)decltype(q)::value * n + lm] = K[16 * decltype(q)::value]; });
                    }
                    nsteps += 1;
                }
            }
            // (Driver::run goes on after a dense overflow: the state moves, a NaN in it overrides the return code)
            dt = dtnew;
            bool bad = false;
            zo = zn;
            bad = ((__ballot(zn != zn) >> (16 * rr)) & 0x7Full) != 0;   // (a NaN in any of the row's seven components)
            if constexpr (Tab::FSAL) K[0] = K[16 * (S - 1)];
            if (bad) { ret = RET_UNSTABLE; fin = true; }
            else if (t == TF) fin = true;   // the one tstop of a forward solve
            if (fin) finish();
            else { ph = FIRST; fresh = true; }
        }
    }
}

}  // namespace seirls
}  // namespace ude
