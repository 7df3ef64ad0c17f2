// ude_node_ls2.h -- second generation of the lock-step PARITY-mode adjoint of the SEIR script's neural ODE (ude_node_ls.h: same
// arithmetic, same results bit for bit, same host interface), built like ude_seir_ls2.h: the trip of the fast-mode kernel
// ude_node_lsf.h (hidden units permuted over the registers: the seven input-cotangent sums need two lane exchanges each instead of
// eight; a slot's row keeps only its component of lambda, K in LDS; the state machine moves on in front of the E barrier where no
// parameter pass lies in between) around the UNCHANGED step-end pass of ude_node_ls.h (slot_pass, mu in HBM, the six factor rows of
// a stage through the workspace: a1 a2 delta3 delta2 transposed through their tiles, a3 and delta1 -- four consecutive units per lane
// in the permuted layout -- straight from the registers).
#pragma once
#include "ude_node_ls.h"

namespace ude {
namespace nodels2 {

using namespace nodels;
using seirls::kst;

template <class Tab>
constexpr int lds_doubles() {
    constexpr int NSTC = popc(stage_mask<Tab>());
    return 2 * H * LDW + 4 * H * TLD + 8 * 16 + 8 * 16 + NSTC * NSLOTS * XFW + TABL + 6 * NSLOTS +
           NSLOTS * 4 * 2 + NSLOTS * kst<Tab>() + 16 * 8 + 4 * 2 * NSTC * QW + NIN * H + NSLOTS * Tab::S * 8 + 2 * H + 2 * NSLOTS;
}

template <class Tab>
__global__ void __launch_bounds__(BLOCKT, 1) node_ls2_adj_kernel(const KParams p, double* __restrict__ facws, int* __restrict__ queue) {
    constexpr int S = Tab::S, NK = Tab::NK;
    constexpr unsigned MASK = stage_mask<Tab>();
    constexpr int NSTC = popc(MASK);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* WL2 = sm;                          // [64][65]: W2[i][k] at WL2[i + k * LDW]
    double* WL3 = WL2 + H * LDW;
    double* T_A1 = WL3 + H * LDW;              // [64][17] tiles [unit][slot]
    double* T_A2 = T_A1 + H * TLD;
    double* T_D3 = T_A2 + H * TLD;
    double* T_D2 = T_D3 + H * TLD;
    double* XIN = T_D2 + H * TLD;              // [8][16]: x0..x6, 1
    double* D4S = XIN + 8 * 16;                // [8][16]: delta4_0..6, 0
    double* XF = D4S + 8 * 16;                 // [NSTC][16][14]
    double* TB = XF + NSTC * NSLOTS * XFW;     // tableau
    double* RDT = TB + TABL;
    long long* RG = reinterpret_cast<long long*>(RDT + NSLOTS);
    int* REQI = reinterpret_cast<int*>(RG + NSLOTS);
    int* REQZ = REQI + NSLOTS;
    int* RCOL = REQZ + NSLOTS;
    int* ROK = RCOL + NSLOTS;
    double* SUMW = RDT + 6 * NSLOTS;           // [16][4][2]
    double* KSL = SUMW + NSLOTS * 4 * 2;       // [16 slots][KST]
    double* RQL = KSL + NSLOTS * kst<Tab>();   // [16][8]
    double* ASTG = RQL + 16 * 8;               // [4 wavefronts][2][NSTC][16]: staging of the step-end pass (E) ...
    // ... whose space serves, outside E, the interpolation weights / states of phase C and the input-cotangent partial sums of the
    // matrix phase (written and consumed in front of the E barrier): 6.6 KB that the block does not have otherwise (160 KB per CU)
    double* GXP = ASTG;                        // [7][16][4]
    double* BQ = GXP + NIN * NSLOTS * 4;       // [16][16]
    double* YS = BQ + NSLOTS * 16;             // [16][8]
    static_assert(NIN * NSLOTS * 4 + NSLOTS * 16 + NSLOTS * 8 <= 4 * 2 * NSTC * QW, "phase-C / matrix-phase scratch inside the E staging");
    double* W1L = ASTG + 4 * 2 * NSTC * QW;    // [7][64]
    double* KL = W1L + NIN * H;                // [16 slots][S][8]
    double* B2L = KL + NSLOTS * S * 8;         // [64] b2, [64] b3
    double* B3L = B2L + H;
    int* RCS = reinterpret_cast<int*>(B3L + H);   // [16] compacted stage of the slot's evaluation, [16] whether it evaluates (matrix view reads them)
    int* REV = RCS + NSLOTS;

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int kq = l >> 4, jc = l & 15;
    const int rr = l >> 4, lm = l & 15;
    const int slot = 4 * w + rr;
    const double* __restrict__ th = p.theta;
    const TabDev* __restrict__ tab = p.tab;
    const int n = NC;
    const int nfld = 3 + n + NK * n;

    for (int i = tid; i < H * H; i += BLOCKT) { WL2[(i % H) + (i / H) * LDW] = th[OFF_W2 + i]; WL3[(i % H) + (i / H) * LDW] = th[OFF_W3 + i]; }
    const int urow = 16 * w + 4 * (jc & 3) + (jc >> 2);
    const int u0r = 16 * w + 4 * kq;
    double W1A[2], W4T[2];
    static_for<0, 2>([&](auto sc) {
        const int k = 4 * decltype(sc)::value + kq;
        W1A[sc] = k < NIN ? th[OFF_W1 + urow + k * H] : th[OFF_B1 + urow];
        W4T[sc] = k < NOUT ? th[OFF_W4 + k + urow * NOUT] : 0.0;
    });
    for (int i = tid; i < NIN * H; i += BLOCKT) W1L[i] = th[OFF_W1 + i];
    if (tid < H) { B2L[tid] = th[OFF_B2 + tid]; B3L[tid] = th[OFF_B3 + tid]; }
    const double muc = p.mc.consts[4], sgc = p.mc.consts[5];
    for (int i = tid; i < 8 * 16; i += BLOCKT) { XIN[i] = 1.0; D4S[i] = 0.0; }
    for (int i = tid; i < 16 * 16; i += BLOCKT) TB[i] = tab->A[i >> 4][i & 15];
    if (tid < 16) { TB[256 + tid] = tab->B[tid]; TB[272 + tid] = tab->BT[tid]; TB[288 + tid] = tab->C[tid]; }
    for (int i = tid; i < NSLOTS * 8; i += BLOCKT) SUMW[i] = 0.0;
    for (int i = tid; i < NSLOTS * S * 8; i += BLOCKT) KL[i] = 0.0;
    for (int i = tid; i < 16 * 8; i += BLOCKT) RQL[i] = ((i >> 3) < NK && (i & 7) < 7) ? tab->R[i >> 3][i & 7] : 0.0;

    const OptsR o(p.o);
    const double T0 = p.t0, TF = p.tf, tdir = -1.0;
    const double dtmax = o.dtmax;
    const double ntot = (double)(p.n_state + p.n_param);
    const bool user_dt = o.dt0 > 0.0;
    int ph = PH_IDLE, ret = RET_SUCCESS, col = 0, iter = 0, sf = 0, cur = 0, nsteps = 1;
    long long gid = 0;
    bool accept = true, exhausted = false, zero_req = false;
    double t = TF, dt = 0.0, dt0 = 0.0, d1n = 0.0, qold = o.qoldinit, q11 = 1.0, tstop = T0, ssrep = 0.0, ts = 0.0, te = 0.0;
    int nfc = 0, nacc = 0, nrej = 0;
    double zo = 0.0, zn = 0.0;
    constexpr int KST = kst<Tab>(), NPF = KST / 16;
    double* const krec = KSL + slot * KST;
    const double* const ksl = krec + 3 + (lm < NC ? lm : NC - 1);
    double* const kl = KL + (size_t)slot * S * 8 + (lm < NC ? lm : 7);
    double* const f0l = KL + (size_t)slot * S * 8;   // f0 of the initial-dt phase IS the K[0] row (written at INIT0, stage 0 overwrites it only after INIT1)
    double pf[NPF];
    int pf_s = -1, pf_want = -1;
    static_for<0, NPF>([&](auto i) { pf[i] = 0.0; });
    const double* cot = p.cot;
    size_t cot_si = 0, cot_sc = 0;
    double* const fmine = facws + (size_t)blockIdx.x * fac_doubles_per_block<Tab>();

    auto fetch_interval = [&](int s) {
        pf_s = s;
        const double* base = dense_rec<true>(p, s, nfld, gid);
        static_for<0, NPF>([&](auto i) {
            const int f = lm + 16 * (int)decltype(i)::value;
            pf[i] = base[f < nfld ? f : 0];
        });
    };
    auto load_interval = [&](int s) {
        if (pf_s != s) fetch_interval(s);
        sf = s;
        static_for<0, NPF>([&](auto i) { krec[lm + 16 * (int)decltype(i)::value] = pf[i]; });
        ts = krec[0];
        te = krec[1];
        pf_want = s - 1;
    };
    auto bcast = [&](double ownv, double (&out)[NC]) { static_for<0, NC>([&](auto c) { out[c] = rshfl(ownv, decltype(c)::value); }); };
    auto SV = [&](int i) { return p.saveat[i]; };
    auto tstop_from_cur = [&]() { return (cur >= 0 && SV(cur) > T0) ? SV(cur) : T0; };
    auto at_tstop = [&](double tt) {
        bool mod = false;
        while (cur >= 0 && SV(cur) >= tt) {
            if (SV(cur) == tt) {
                if (lm < NC) zo += cot[(size_t)cur * cot_si + (size_t)lm * cot_sc];
                mod = true;
            }
            cur -= 1;
        }
        return mod;
    };
    auto results = [&]() {
        if (lm == 0) {
            if (p.stats) { int64_t* st = p.stats + (size_t)gid * 8; st[4] = nfc; st[5] = nacc; st[6] = nrej; }
            if (ret != RET_SUCCESS) p.retcode[gid] = ret;
        }
        if (p.grad_u0 && lm < NC) p.grad_u0[(size_t)gid * n + lm] = zo;
    };
    // a 64-term hidden product: four 16-term chains (four MFMAs each) added left to right; the A fragment of a chain is read from
    // the block's LDS copy of the weights (transposed: A[i][k] = W[k][unit(i)])
    auto hidden = [&](const double* W, const double* T, bool transposed, double (&out)[4]) {
        v4d acc[4];
        static_for<0, 4>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
            static_for<0, 4>([&](auto q) {
                constexpr int s = 4 * b + decltype(q)::value;
                const int colk = 4 * s + kq;
                const double a = transposed ? W[colk + urow * LDW] : W[urow + colk * LDW];
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(4 * s + kq) * TLD + jc], acc[b], 0, 0, 0);
            });
        });
        static_for<0, 4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            out[r] = ((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r];
        });
    };
    __syncthreads();

    for (;;) {
        // ---- A. an idle slot takes the next trajectory of the ensemble ----
        if (ph == PH_IDLE && !exhausted) {
            for (;;) {
                int g = 0;
                if (lm == 0) g = atomicAdd(queue, 1);
                g = __shfl(g, 0, 16);
                if (g >= p.N) { exhausted = true; break; }
                if (p.retcode[g] != RET_SUCCESS) continue;
                gid = g;
                if (p.cot_in) { cot = p.cot_in + (size_t)gid * p.ns * n; cot_si = n; cot_sc = 1; }
                else { cot = p.cot + gid; cot_si = (size_t)n * p.Npad; cot_sc = p.Npad; }
                nsteps = p.dense_n[gid];
                pf_s = -1; pf_want = -1;
                cur = p.ns - 1;
                zo = 0.0;
                t = TF; qold = o.qoldinit; q11 = 1.0; accept = true; iter = 0; ret = RET_SUCCESS; col = 0;
                nfc = 0; nacc = 0; nrej = 0;
                load_interval(nsteps - 1);
                at_tstop(TF);
                tstop = tstop_from_cur();
                zero_req = true;
                if (user_dt) {
                    dt = tdir * o.dt0;
                    if constexpr (Tab::FSAL) nfc += 1;
                    ph = 0;
                } else ph = PH_INIT0;
                break;
            }
        }

        // ---- B. the evaluation this slot needs now ----
        bool ev = false;
        double tev = t;
        int cs = 0;
        double zsrc = zo;
        if (ph == PH_INIT0) {
            ev = true;
        } else if (ph == PH_INIT1) {
            ev = true;
            const double dt0t = tdir * dt0;
            zsrc = __builtin_fma(dt0t, f0l[lm < NC ? lm : 7], zo);
            tev = t + dt0t;
            cs = 1;
        } else if (ph >= 0) {
            const int s = ph;
            bool go = true;
            if (s == 0) {   // loopheader!
                if (iter > 0 && !accept) {
                    double den = q11 / o.gamma;
                    const double iq = 1.0 / o.qmin;
                    if (iq < den) den = iq;
                    dt = dt / den;
                }
                iter += 1;
                if (fabs(dt) > dtmax) dt = tdir * dtmax;
                {
                    const double rem = fabs(tstop - t);
                    if (fabs(dt) > rem) dt = tdir * rem;
                }
                if (iter > o.maxiters) { ret = RET_MAXITERS; go = false; }
                else if (dt != dt) { ret = RET_UNSTABLE; go = false; }
                else if (fabs(dt) <= REAL_EPS * fabs(t) && fabs(dt) < fabs(tstop - t)) { ret = RET_DTLESSTHANMIN; go = false; }
            }
            if (go) {
                ev = true;
                if (s > 0) {
                    const double* Ar = TB + s * 16;
                    double acc = Ar[0] * kl[0];
                    static_for<1, S - 1>([&](auto j) { acc = __builtin_fma(Ar[decltype(j)::value], kl[8 * decltype(j)::value], acc); });
                    zsrc = __builtin_fma(dt, acc, zo);
                }
                tev = t + TB[288 + s] * dt;
                cs = __builtin_popcount(MASK & ((1u << s) - 1u));
            } else {
                ph = PH_FLUSH;
                results();
            }
        }
        double zs[NC];
        bcast(zsrc, zs);

        // ---- C. the forward state at tev, the network inputs ----
        double y[NC];
        static_for<0, NC>([&](auto c) { y[c] = 1.0; });
        if (ev) {
            while (tev < ts && sf > 0) load_interval(sf - 1);
            while (tev >= te && sf < nsteps - 1) load_interval(sf + 1);
            const double dtf = te - ts;
            const double thv = (tev - ts) / dtf;
            const double* rq = RQL + lm * 8;
            double hq = rq[0];
            static_for<1, 7>([&](auto i) { hq = __builtin_fma(thv, hq, rq[decltype(i)::value]); });
            BQ[slot * 16 + lm] = (lm == 0 ? thv : thv * thv) * hq;
            double acc = 0.0;
            bool first = true;
            static_for<0, NK>([&](auto q) {
                if constexpr (Tab::dense_uses(decltype(q)::value)) {
                    const double bqv = BQ[slot * 16 + decltype(q)::value];
                    const double kq_ = ksl[NC + NC * (int)decltype(q)::value];
                    acc = first ? kq_ * bqv : __builtin_fma(kq_, bqv, acc);
                    first = false;
                }
            });
            if (lm < NC) YS[slot * 8 + lm] = __builtin_fma(dtf, acc, ksl[0]);
            static_for<0, NC>([&](auto c) { y[c] = YS[slot * 8 + decltype(c)::value]; });
            const double xin[NIN] = {y[0] / y[4], y[1], y[2], y[3], y[4], y[5] / y[4], y[6]};
            const double d4v[NOUT] = {zs[0], zs[1], zs[2], zs[3], zs[5], 0.0, 0.0};
            if (lm < NIN) {
                double xo = 0.0, dq = 0.0;
                static_for<0, NIN>([&](auto c) { xo = (lm == (int)decltype(c)::value) ? xin[c] : xo; dq = (lm == (int)decltype(c)::value) ? d4v[c] : dq; });
                XIN[lm * 16 + slot] = xo;
                D4S[lm * 16 + slot] = dq;
                double* xf = XF + (cs * NSLOTS + slot) * XFW;
                xf[lm] = xo;
                xf[NIN + lm] = dq;
            }
        }
        if (lm == 0) { RCS[slot] = cs; REV[slot] = ev ? 1 : 0; }
        if (!__syncthreads_or(ph != PH_IDLE)) break;
        double a3[4], dv1[4];
        {
            v4d z = v4d{0.0, 0.0, 0.0, 0.0};
            static_for<0, 2>([&](auto sc) { z = __builtin_amdgcn_mfma_f64_16x16x4f64(W1A[sc], XIN[(4 * decltype(sc)::value + kq) * 16 + jc], z, 0, 0, 0); });
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                T_A1[(u0r + r) * TLD + jc] = dtanh(z[r]);
            });
            __syncthreads();
            if (pf_want >= 0) { fetch_interval(pf_want); pf_want = -1; }
            double hz[4];
            hidden(WL2, T_A1, false, hz);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                T_A2[(u0r + r) * TLD + jc] = dtanh(hz[r] + B2L[u0r + r]);
            });
            __syncthreads();
            hidden(WL3, T_A2, false, hz);
            v4d s3 = v4d{0.0, 0.0, 0.0, 0.0};
            static_for<0, 2>([&](auto sc) { s3 = __builtin_amdgcn_mfma_f64_16x16x4f64(W4T[sc], D4S[(4 * decltype(sc)::value + kq) * 16 + jc], s3, 0, 0, 0); });
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                a3[r] = dtanh(hz[r] + B3L[u0r + r]);
                T_D3[(u0r + r) * TLD + jc] = s3[r] * __builtin_fma(-a3[r], a3[r], 1.0);
            });
            __syncthreads();
            hidden(WL3, T_D3, true, hz);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const double a2 = T_A2[(u0r + r) * TLD + jc];
                T_D2[(u0r + r) * TLD + jc] = hz[r] * __builtin_fma(-a2, a2, 1.0);
            });
            __syncthreads();
            hidden(WL2, T_D2, true, hz);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const double a1 = T_A1[(u0r + r) * TLD + jc];
                dv1[r] = hz[r] * __builtin_fma(-a1, a1, 1.0);
            });
            static_for<0, NIN>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const double* wl = W1L + m * H + u0r;
                double x = (wl[0] * dv1[0] + wl[1] * dv1[1]) + (wl[2] * dv1[2] + wl[3] * dv1[3]);
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                if (kq == 0) GXP[(m * NSLOTS + jc) * 4 + w] = x;
            });
        }
        // factors of this evaluation to the workspace: a3 and delta1 straight from the registers that hold them (units 4kq .. 4kq+3 of
        // column jc: 32 contiguous bytes per lane), the four tiles transposed by the wavefront that owns the slot (lane i = unit i)
        if (REV[jc]) {
            double* dst = fmine + ((size_t)jc * NSTC + RCS[jc]) * NFAC * H + u0r;
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                dst[2 * H + r] = a3[r];
                dst[3 * H + r] = dv1[r];
            });
        }
        __syncthreads();
        {
            const int evi = ev ? 1 : 0;
            double va1[4], va2[4], vd3[4], vd2[4];
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int sl = 4 * w + q;
                va1[q] = T_A1[l * TLD + sl]; va2[q] = T_A2[l * TLD + sl]; vd3[q] = T_D3[l * TLD + sl]; vd2[q] = T_D2[l * TLD + sl];
            });
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int sl = 4 * w + q;
                if (__builtin_amdgcn_readlane(evi, 16 * q)) {
                    const int cs_ = __builtin_amdgcn_readlane(cs, 16 * q);
                    double* dst = fmine + ((size_t)sl * NSTC + cs_) * NFAC * H + l;
                    dst[0] = va1[q];
                    dst[H] = va2[q];
                    dst[4 * H] = vd2[q];
                    dst[5 * H] = vd3[q];
                }
            });
        }
        // ---- D. the slot's row: state cotangent of this evaluation, and what it asks of the parameter-slot pass ----
        int req = RQ_NONE;
        double kr[NC];
        static_for<0, NC>([&](auto c) { kr[c] = 0.0; });
        if (ev) {
            double gx[NIN];
            static_for<0, NIN>([&](auto mm) {
                const double* g4 = GXP + (decltype(mm)::value * NSLOTS + slot) * 4;
                gx[mm] = (g4[0] + g4[1]) + (g4[2] + g4[3]);
            });
            const double Sv = y[0], Nv = y[4], Dv = y[5];
            kr[0] = -(gx[0] / Nv);
            kr[1] = -__builtin_fma(sgc, zs[6], gx[1]);
            kr[2] = -gx[2];
            kr[3] = -gx[3];
            kr[4] = -(((gx[4] - gx[0] * Sv / (Nv * Nv)) - gx[5] * Dv / (Nv * Nv)) - muc * zs[4]);
            kr[5] = -(gx[5] / Nv);
            kr[6] = -gx[6];
            double ko = 0.0;
            static_for<0, NC>([&](auto c) { ko = (lm == (int)decltype(c)::value) ? kr[c] : ko; });
            if (ph == PH_INIT0) {
                kl[0] = ko;   // (= f0: the K[0] row)
                req = RQ_NORM01;
            } else if (ph == PH_INIT1) {
                req = RQ_NORM2;
            } else {
                const int s = ph;
                kl[8 * s] = ko;
                if (s == S - 1) {
                    if constexpr (Tab::FSAL) zn = zsrc;
                    else {
                        double acc = TB[256] * kl[0];
                        static_for<1, S>([&](auto j) { acc = __builtin_fma(TB[256 + decltype(j)::value], kl[8 * decltype(j)::value], acc); });
                        zn = __builtin_fma(dt, acc, zo);
                    }
                    double acc = TB[272] * kl[0];
                    static_for<1, S>([&](auto j) { acc = __builtin_fma(TB[272 + decltype(j)::value], kl[8 * decltype(j)::value], acc); });
                    const double a0 = fabs(zo), a1 = fabs(zn);
                    double res[NC];
                    bcast((dt * acc) / __builtin_fma((a0 > a1 ? a0 : a1), o.reltol, o.abstol), res);
                    ssrep = 0.0;
                    static_for<0, NC>([&](auto c) { ssrep = __builtin_fma(res[c], res[c], ssrep); });
                    req = RQ_STEP;
                } else {
                    ph += 1;
                }
            }
        } else if (ph == PH_FLUSH) {
            req = RQ_FLUSH;
        }
        if (lm == 0) {
            REQI[slot] = req; REQZ[slot] = zero_req ? 1 : 0; RDT[slot] = dt; RG[slot] = gid; RCOL[slot] = col; ROK[slot] = ret == RET_SUCCESS ? 1 : 0;
        }
        zero_req = false;
        __syncthreads();

        // ---- E. the parameter-slot work the slots asked for: every request is worked on by all four wavefronts, a quarter of the slots each ----
        {
            const int q16 = l & 15;
            const int r_mode = REQI[q16], r_zr = REQZ[q16], r_col = RCOL[q16], r_ok = ROK[q16];
            const long long r_g = RG[q16];
            const double r_dt = RDT[q16];
            unsigned pend = (unsigned)__ballot(l < 16 && (r_mode != RQ_NONE || r_zr != 0));
            auto rl32 = [&](int v, int src) { return __builtin_amdgcn_readlane(v, src); };
            auto rl64 = [&](long long v, int src) {
                return (long long)(((unsigned long long)(unsigned)rl32((int)((unsigned long long)v >> 32), src) << 32) | (unsigned)rl32((int)(unsigned long long)v, src));
            };
#pragma unroll 1
            while (pend != 0u) {
                const int sl = __builtin_ctz(pend);
                pend &= pend - 1u;
                const int mode = rl32(r_mode, sl);
                const int zr = rl32(r_zr, sl);
                const long long g = rl64(r_g, sl);
                const int cl = rl32(r_col, sl);
                const double dt_req = __longlong_as_double(rl64(__double_as_longlong(r_dt), sl));
                double* mbase = p.slot_glob + (size_t)g * (2 * NSLK * H) + l;
                double* mcur = mbase + (size_t)cl * (NSLK * H);
                double* mnew = mbase + (size_t)(1 - cl) * (NSLK * H);
                if (zr) {   // a fresh trajectory: its current mu column starts at zero
#pragma unroll 4
                    for (int k = 0; k < QW; ++k) { mcur[(size_t)(QW * w + k) * H] = 0.0; mcur[(size_t)(H + QW * w + k) * H] = 0.0; }
                    for (int e = w; e < NEX; e += 4) mcur[(size_t)(2 * H + e) * H] = 0.0;
                }
                const double* fb = fmine + (size_t)sl * NSTC * NFAC * H;
                double hh = 0.0, ll = 0.0;
                double* stg = ASTG + w * 2 * NSTC * QW;
                double* sw = SUMW + (sl * 4 + w) * 2;
                if (mode == RQ_STEP) {
                    const double ps = slot_pass<S, MASK, 0>(fb, stg, XF, sl, l, w, TB + 256, TB + 272, dt_req, o.abstol, o.reltol, mcur, mnew, hh, ll);
                    const double tot = group_sum<64>(ps);
                    if (l == 0) sw[0] = tot;
                } else if (mode == RQ_NORM01) {
                    slot_pass<1, 1u, 1>(fb, stg, XF, sl, l, w, TB + 256, TB + 272, 0.0, o.abstol, o.reltol, mcur, mnew, hh, ll);
                    group_dd_sum<64>(hh, ll);
                    if (l == 0) { sw[0] = hh; sw[1] = ll; }
                } else if (mode == RQ_NORM2) {
                    slot_pass<2, 3u, 2>(fb, stg, XF, sl, l, w, TB + 256, TB + 272, 0.0, o.abstol, o.reltol, mcur, mnew, hh, ll);
                    group_dd_sum<64>(hh, ll);
                    if (l == 0) { sw[0] = hh; sw[1] = ll; }
                } else if (mode == RQ_FLUSH) {   // the trajectory's gradient row (zeros if it failed)
                    const bool ok = rl32(r_ok, sl) != 0;
                    double* row = p.grad_part + (size_t)g * p.n_param;
#pragma unroll 4
                    for (int k = QW * w; k < QW * w + QW; ++k) {
                        row[OFF_W2 + l + k * H] = ok ? mcur[(size_t)k * H] : 0.0;
                        row[OFF_W3 + l + k * H] = ok ? mcur[(size_t)(H + k) * H] : 0.0;
                    }
                    for (int e = w; e < NEX; e += 4) {
                        const int idx = extra_index(e, l);
                        if (idx >= 0) row[idx] = ok ? mcur[(size_t)(2 * H + e) * H] : 0.0;
                    }
                }
            }
        }
        __syncthreads();

        // ---- F. what depended on the parameter pass: flush, initial-dt norms, the end of a step ----
        if (req == RQ_FLUSH) {
            ph = PH_IDLE;
        } else if (req == RQ_NORM01) {
            // ode_determine_initdt, first half (the slot sums first -- mu == 0: only the g0 terms --, then the replicated components)
            double lam[NC];
            bcast(zo, lam);
            double h0 = 0.0, l0 = 0.0, h1 = 0.0, l1 = 0.0;
            static_for<0, 4>([&](auto q) { dd_acc(h1, l1, SUMW[(slot * 4 + decltype(q)::value) * 2]); dd_acc(h1, l1, SUMW[(slot * 4 + decltype(q)::value) * 2 + 1]); });
            static_for<0, NC>([&](auto c) {
                const double sk = __builtin_fma(fabs(lam[c]), o.reltol, o.abstol);
                const double q0 = lam[c] / sk, q1 = f0l[decltype(c)::value] / sk;
                dd_acc(h0, l0, q0 * q0);
                dd_acc(h1, l1, q1 * q1);
            });
            const double s0 = h0 + l0, s1 = h1 + l1;
            const double d0 = __builtin_sqrt(s0 / ntot);
            d1n = __builtin_sqrt(s1 / ntot);
            dt0 = (d0 < 1e-5 || d1n < 1e-5) ? 1e-6 : (d0 / d1n) / 100.0;
            if (dt0 > dtmax) dt0 = dtmax;
            if (d1n != d1n) {
                ret = RET_UNSTABLE;
                ph = PH_FLUSH;
                nfc = 2 + (Tab::FSAL ? 1 : 0); nacc = 0; nrej = 0;
                results();
            } else if (dt0 < 10.0 * REAL_EPS) {
                dt = tdir * 1e-6;
                nfc += 2;
                if constexpr (Tab::FSAL) nfc += 1;
                ph = 0;
            } else {
                ph = PH_INIT1;
            }
        } else if (req == RQ_NORM2) {
            double lam[NC];
            bcast(zo, lam);
            double h2 = 0.0, l2 = 0.0;
            static_for<0, 4>([&](auto q) { dd_acc(h2, l2, SUMW[(slot * 4 + decltype(q)::value) * 2]); dd_acc(h2, l2, SUMW[(slot * 4 + decltype(q)::value) * 2 + 1]); });
            static_for<0, NC>([&](auto c) {
                const double sk = __builtin_fma(fabs(lam[c]), o.reltol, o.abstol);
                const double q = (kr[c] - f0l[decltype(c)::value]) / sk;
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
            ph = 0;
        } else if (req == RQ_STEP) {
            nfc += Tab::FSAL ? S - 1 : S;
            double ss = ssrep;
            ss += ((SUMW[slot * 8] + SUMW[slot * 8 + 2]) + SUMW[slot * 8 + 4]) + SUMW[slot * 8 + 6];
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
                double* row = p.trace + ((size_t)p.trace_cap + (iter - 1)) * 5;
                row[0] = t; row[1] = dt; row[2] = EEst; row[3] = q; row[4] = accept ? 1.0 : 0.0;
            }
            bool fin = false;
            if (accept) {
                nacc += 1;
                qold = EEst > o.qoldinit ? EEst : o.qoldinit;
                double dtnew = dt / q;
                const double ttmp = t + dt;
                {
                    const double mxt = t > tstop ? t : tstop;
                    t = fabs(ttmp - tstop) < 100.0 * ulp_of(mxt) ? tstop : ttmp;
                }
                if (fabs(dtnew) > dtmax) dtnew = tdir * dtmax;
                dt = dtnew;
                zo = zn;
                const bool bad = ((__ballot(lm < NC && zn != zn) >> (16 * rr)) & 0xFFFFull) != 0;
                col = 1 - col;   // slot_accept: the candidate column becomes current
                if (bad) { ret = RET_UNSTABLE; fin = true; }
                if (t == tstop) {
                    const bool modified = at_tstop(t);
                    if (tstop == T0) fin = true;   // done
                    else {
                        tstop = tstop_from_cur();
                        if (modified && Tab::FSAL) nfc += 1;
                    }
                }
            } else {
                nrej += 1;
                if (EEst != EEst) { ret = RET_UNSTABLE; fin = true; }
            }
            if (fin) {
                ph = PH_FLUSH;
                results();
            } else {
                ph = 0;
            }
        }
    }
}

}  // namespace nodels2
}  // namespace ude
