// ude_node_lsf.h -- the `fast` interpolating adjoint of the SEIR script's pure neural ODE (dudt_node, SEIR_exposure/seir_exposure.jl:53-83:
// FastChain 7-64-64-64-7 tanh, 9287 parameters) with the parameter cotangent as a BLOCK-LEVEL MATRIX-CORE ACCUMULATION: the scheme of
// ude_seir_lsf.h (read its header first) around the network of ude_node_ls.h.
//
// Per trip, K = the sixteen slot columns, weights w_k = dt_k b_{s_k} on the delta side:
//      dW2 += (-(w delta2)) . a1^T        dW3 += (-(w delta3)) . a2^T          (16 MFMAs each)
//      db2 += (-(w delta2)) . 1           db3 += (-(w delta3)) . 1
//      dW1 | db1 += (-(w delta1)) . [x0 .. x6 1]^T                             dW4^T += a3 . (-(w delta4))^T
// into 12 x 4 doubles per lane of block-resident accumulators; db4 on the slot's row.  W2 and W3 sit in padded LDS copies (row and
// column fragments conflict-free, ude_node_ls.h); the hidden units are PERMUTED over the registers as in ude_seir_lsf.h (a lane
// holds four consecutive units), so the seven input-cotangent sums need two lane exchanges each instead of eight.
// Rejected attempts are replayed with negated weights; the association is the oracle's UDEO_SENSE_FAST_MM: one trajectory is
// bit-identical in all 9287 entries, per trajectory the backward step counts and dL/du0 are bit-identical to the fast mode.
// One block per compute unit (158 KB of LDS).  Float64, shared time grid.
#pragma once
#include "ude_node_ls.h"

namespace ude {
namespace nodelf {

using seirls::v4d;
using seirls::TABL;
using seirls::kst;
using seirls::rshfl;
using nodels::H;
using nodels::NSLOTS;
using nodels::BLOCKT;
using nodels::NC;
using nodels::NIN;
using nodels::NOUT;
using nodels::TLD;
using nodels::LDW;
using nodels::OFF_W1;
using nodels::OFF_B1;
using nodels::OFF_W2;
using nodels::OFF_B2;
using nodels::OFF_W3;
using nodels::OFF_B3;
using nodels::OFF_W4;
using nodels::OFF_B4;

enum { PH_IDLE = -4, PH_FLUSH = -3, PH_INIT0 = -2, PH_INIT1 = -1 };   // >= 0: stage s of a step attempt

template <class Tab>
constexpr int lds_doubles() {
    return 2 * H * LDW + 6 * H * TLD + 8 * 16 + 8 * 16 + 16 + NIN * NSLOTS * 4 + NSLOTS * 16 + NSLOTS * 8 + TABL + 16 * 8 + NSLOTS * 8 +
           NSLOTS * kst<Tab>() + NIN * H + NSLOTS * Tab::S * 8 + 2 * H;
}

template <class Tab>
__global__ void __launch_bounds__(BLOCKT, 1) node_lsf_adj_kernel(const KParams p, double* __restrict__ /*unused*/, int* __restrict__ /*unused*/) {
    constexpr int S = Tab::S, NK = Tab::NK;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* WL2 = sm;                         // [64][65]: W2[i][k] at WL2[i + k * LDW]
    double* WL3 = WL2 + H * LDW;
    double* T_A1 = WL3 + H * LDW;             // [64][17] tiles [unit][slot]
    double* T_A2 = T_A1 + H * TLD;
    double* T_A3 = T_A2 + H * TLD;
    double* T_D3 = T_A3 + H * TLD;
    double* T_D2 = T_D3 + H * TLD;
    double* T_D1 = T_D2 + H * TLD;
    double* XIN = T_D1 + H * TLD;             // [8][16]: x0..x6, 1
    double* D4S = XIN + 8 * 16;               // [8][16]: delta4_0..6, 0
    double* WSL = D4S + 8 * 16;               // [16] weight dt b_s of the slot's evaluation (0: contributes nothing)
    double* GXP = WSL + 16;                   // [7][16][4]: per wavefront partial sums of the input cotangent
    double* BQ = GXP + NIN * NSLOTS * 4;      // [16][16]
    double* YS = BQ + NSLOTS * 16;            // [16][8]
    double* TB = YS + NSLOTS * 8;             // tableau: A[16][16], B, BT, C
    double* RQL = TB + TABL;                  // [16 lanes q][8]: Horner tables of b_q(theta)
    double* F0L = RQL + 16 * 8;               // [16 slots][8]: f0 of the initial-dt phase
    double* KSL = F0L + NSLOTS * 8;           // [16 slots][KST]: the stored record of the slot's current forward interval
    double* W1L = KSL + NSLOTS * kst<Tab>();  // [7][64]
    double* KL = W1L + NIN * H;               // [16 slots][S][8]: stage derivatives of lambda
    double* MB4 = BQ;                         // [16][8] the slots' shares of db4 (END of the kernel only: the interpolation weights' space)
    double* B2L = KL + NSLOTS * S * 8;        // [64] b2, [64] b3
    double* B3L = B2L + H;

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int kq = l >> 4, jc = l & 15;       // matrix view
    const int rr = l >> 4, lm = l & 15;       // scalar view: slot 4w + rr, lane lm of its row
    const int slot = 4 * w + rr;
    const double* __restrict__ th = p.theta;
    const TabDev* __restrict__ tab = p.tab;
    const int n = NC;
    const int nfld = 3 + n + NK * n;

    for (int i = tid; i < H * H; i += BLOCKT) { WL2[(i % H) + (i / H) * LDW] = th[OFF_W2 + i]; WL3[(i % H) + (i / H) * LDW] = th[OFF_W3 + i]; }
    // (ude_seir_lsf.h) tile row i = kq + 4r of a network product is hidden unit 16w + 4 (i & 3) + (i >> 2): register r of lane (kq, .)
    // is unit 16w + 4kq + r
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
    // every column finite from the first trip on
    for (int i = tid; i < 6 * H * TLD; i += BLOCKT) T_A1[i] = 0.0;
    for (int i = tid; i < 8 * 16; i += BLOCKT) { XIN[i] = i >= 7 * 16 ? 1.0 : 0.0; D4S[i] = 0.0; }
    if (tid < 16) WSL[tid] = 0.0;
    for (int i = tid; i < 16 * 16; i += BLOCKT) TB[i] = tab->A[i >> 4][i & 15];
    if (tid < 16) { TB[256 + tid] = tab->B[tid]; TB[272 + tid] = tab->BT[tid]; TB[288 + tid] = tab->C[tid]; }
    for (int i = tid; i < NSLOTS * 8; i += BLOCKT) { YS[i] = 0.0; F0L[i] = 0.0; }
    for (int i = tid; i < NSLOTS * S * 8; i += BLOCKT) KL[i] = 0.0;
    for (int i = tid; i < 16 * 8; i += BLOCKT) RQL[i] = ((i >> 3) < NK && (i & 7) < 7) ? tab->R[i >> 3][i & 7] : 0.0;

    // ---- the block's share of the gradient ----
    v4d gW2[4], gW3[4], gB2, gB3, gW1, gW4;
    static_for<0, 4>([&](auto c) { gW2[c] = v4d{0.0, 0.0, 0.0, 0.0}; gW3[c] = v4d{0.0, 0.0, 0.0, 0.0}; });
    gB2 = v4d{0.0, 0.0, 0.0, 0.0}; gB3 = gB2; gW1 = gB2; gW4 = gB2;
    double mb4 = 0.0;   // (row view, lane o < 7 of the slot's row) db4[o] share of this slot

    const OptsR o(p.o);
    const double T0 = p.t0, TF = p.tf, tdir = -1.0;
    const double dtmax = o.dtmax;
    const double ntot = (double)p.n_state;    // fast mode: lambda alone is under error control
    const bool user_dt = o.dt0 > 0.0;
    const int nblk = gridDim.x;
    int ph = PH_IDLE, ret = RET_SUCCESS, iter = 0, sf = 0, cur = 0, nsteps = 1, jtraj = 0;
    long long gid = 0;
    bool accept = true, exhausted = false, replay = false;
    double t = TF, dt = 0.0, dt0 = 0.0, d1n = 0.0, qold = o.qoldinit, q11 = 1.0, tstop = T0, ts = 0.0, te = 0.0;
    int nfc = 0, nacc = 0, nrej = 0;
    double zo = 0.0;                          // this lane's component of lambda
    constexpr int KST = kst<Tab>(), NPF = KST / 16;
    double* const krec = KSL + slot * KST;
    const double* const ksl = krec + 3 + (lm < NC ? lm : NC - 1);
    double* const kl = KL + (size_t)slot * S * 8 + (lm < NC ? lm : 7);
    double* const f0l = F0L + slot * 8;
    double pf[NPF];
    int pf_s = -1, pf_want = -1;
    static_for<0, NPF>([&](auto i) { pf[i] = 0.0; });
    const double* cot = p.cot;
    size_t cot_si = 0, cot_sc = 0;

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
    // a trajectory whose BACKWARD solve stops early has already added the stages it evaluated to the block's accumulators and cannot be
    // taken out again: the block then reports NaN for its whole gradient row -- ude_last_failures' contract ("such trajectories
    // contribute nothing to the gradient") cannot be kept in this mode, so the gradient is refused loudly instead of returned polluted
    // (advisor, round 5; include/udecore.h documents it next to UDE_SENSE_INTERPOLATING_ADJOINT_FAST)
    int bwd_failed = 0;
    auto results = [&]() {
        if (ret != RET_SUCCESS) bwd_failed = 1;
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
        // ---- A. an idle slot takes its next trajectory: g = block + nblocks (slot + 16 j) ----
        if (ph == PH_IDLE && !exhausted) {
            for (;;) {
                const long long g = (long long)blockIdx.x + (long long)nblk * (slot + 16ll * jtraj);
                jtraj += 1;
                if (g >= p.N) { exhausted = true; break; }
                if (p.retcode[g] != RET_SUCCESS) continue;
                gid = g;
                if (p.cot_in) { cot = p.cot_in + (size_t)gid * p.ns * n; cot_si = n; cot_sc = 1; }
                else { cot = p.cot + gid; cot_si = (size_t)n * p.Npad; cot_sc = p.Npad; }
                nsteps = p.dense_n[gid];
                pf_s = -1; pf_want = -1;
                cur = p.ns - 1;
                zo = 0.0;
                t = TF; qold = o.qoldinit; q11 = 1.0; accept = true; iter = 0; ret = RET_SUCCESS; replay = false;
                nfc = 0; nacc = 0; nrej = 0;
                load_interval(nsteps - 1);
                at_tstop(TF);
                tstop = tstop_from_cur();
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
        double tev = t, wgt = 0.0;
        double zsrc = zo;
        if (ph == PH_INIT0) {
            ev = true;
        } else if (ph == PH_INIT1) {
            ev = true;
            const double dt0t = tdir * dt0;
            zsrc = __builtin_fma(dt0t, f0l[lm < NC ? lm : 7], zo);
            tev = t + dt0t;
        } else if (ph >= 0) {
            const int s = ph;
            bool go = true;
            if (s == 0 && !replay) {   // loopheader!
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
                const double bs = TB[256 + s];
                wgt = bs != 0.0 ? (replay ? -(dt * bs) : dt * bs) : 0.0;
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
            // network input [S/N, E, I, R, N, D/N, C] and the output-layer cotangent delta4 = [lam0 lam1 lam2 lam3 lam5 0 0]
            const double xin[NIN] = {y[0] / y[4], y[1], y[2], y[3], y[4], y[5] / y[4], y[6]};
            const double d4v[NOUT] = {zs[0], zs[1], zs[2], zs[3], zs[5], 0.0, 0.0};
            if (lm < NIN) {
                double xo = 0.0, dq = 0.0;
                static_for<0, NIN>([&](auto c) { xo = (lm == (int)decltype(c)::value) ? xin[c] : xo; dq = (lm == (int)decltype(c)::value) ? d4v[c] : dq; });
                XIN[lm * 16 + slot] = xo;
                D4S[lm * 16 + slot] = dq;
                mb4 = wgt != 0.0 ? mb4 + (-(wgt * dq)) : mb4;   // db4[o]: delta4[o] times 1
            }
            if (lm == 0) WSL[slot] = wgt;
        } else {   // no evaluation: a finite column with zero weight
            if (lm < NIN) { XIN[lm * 16 + slot] = 0.0; D4S[lm * 16 + slot] = 0.0; }
            if (lm == 0) WSL[slot] = 0.0;
        }
        if (!__syncthreads_or(ph != PH_IDLE)) break;
        {
            // first layer: 7 inputs + bias in two k-steps
            v4d z = v4d{0.0, 0.0, 0.0, 0.0};
            static_for<0, 2>([&](auto sc) { z = __builtin_amdgcn_mfma_f64_16x16x4f64(W1A[sc], XIN[(4 * decltype(sc)::value + kq) * 16 + jc], z, 0, 0, 0); });
            double a3[4], dv1[4];
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
            // delta3 = (W4^T delta4) (1 - a3^2): the 7-term chain, its zero eighth term included
            v4d s3 = v4d{0.0, 0.0, 0.0, 0.0};
            static_for<0, 2>([&](auto sc) { s3 = __builtin_amdgcn_mfma_f64_16x16x4f64(W4T[sc], D4S[(4 * decltype(sc)::value + kq) * 16 + jc], s3, 0, 0, 0); });
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                a3[r] = dtanh(hz[r] + B3L[u0r + r]);
                T_A3[(u0r + r) * TLD + jc] = a3[r];
                T_D3[(u0r + r) * TLD + jc] = s3[r] * __builtin_fma(-a3[r], a3[r], 1.0);
            });
            __syncthreads();
            hidden(WL3, T_D3, true, hz);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const double a2 = T_A2[(u0r + r) * TLD + jc];   // (this lane's own value: read back instead of held across two barriers)
                T_D2[(u0r + r) * TLD + jc] = hz[r] * __builtin_fma(-a2, a2, 1.0);
            });
            __syncthreads();
            hidden(WL2, T_D2, true, hz);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const double a1 = T_A1[(u0r + r) * TLD + jc];
                dv1[r] = hz[r] * __builtin_fma(-a1, a1, 1.0);
                T_D1[(u0r + r) * TLD + jc] = dv1[r];
            });
            // input cotangent: rounded products W1[u][m] delta1[u] under the adjacent-pair tree over the 64 units
            static_for<0, NIN>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const double* wl = W1L + m * H + u0r;
                double x = (wl[0] * dv1[0] + wl[1] * dv1[1]) + (wl[2] * dv1[2] + wl[3] * dv1[3]);   // levels 1, 2: this lane's four units
                x += __shfl_xor(x, 16, 64);                                                          // level 3
                x += __shfl_xor(x, 32, 64);                                                          // level 4
                if (kq == 0) GXP[(m * NSLOTS + jc) * 4 + w] = x;
            });
            // ---- the parameter cotangent of this trip: K = the sixteen slot columns, weights on the delta side ----
            // (a rolled loop over the four k-steps: unrolled, the operand loads of all 48 products are hoisted and 40 registers spill)
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * q + kq;
                const double wk = WSL[k];
                const int rown = (16 * w + jc) * TLD + k;     // A operand: row jc of this wavefront's tile = unit 16w + jc (not permuted)
                const double Ad3 = -(wk * T_D3[rown]);
                const double Ad2 = -(wk * T_D2[rown]);
                const double Ad1 = -(wk * T_D1[rown]);
                const double Aa3 = T_A3[rown];
                static_for<0, 4>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    gW2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad2, T_A1[(16 * c + jc) * TLD + k], gW2[c], 0, 0, 0);
                    gW3[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad3, T_A2[(16 * c + jc) * TLD + k], gW3[c], 0, 0, 0);
                });
                const double one0 = jc == 0 ? 1.0 : 0.0;
                gB2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad2, one0, gB2, 0, 0, 0);
                gB3 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad3, one0, gB3, 0, 0, 0);
                gW1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad1, jc < 8 ? XIN[(jc & 7) * 16 + k] : 0.0, gW1, 0, 0, 0);
                gW4 = __builtin_amdgcn_mfma_f64_16x16x4f64(Aa3, jc < NOUT ? -(wk * D4S[(jc & 7) * 16 + k]) : 0.0, gW4, 0, 0, 0);
            }
        }
        __syncthreads();

        // ---- D. the slot's row: state cotangent of this evaluation, then its state machine ----
        if (ph == PH_FLUSH) {
            ph = PH_IDLE;
        } else if (ev) {
            bcast(zsrc, zs);   // (again: seven registers less across the matrix phase)
            double gx[NIN];
            static_for<0, NIN>([&](auto mm) {
                const double* g4 = GXP + (decltype(mm)::value * NSLOTS + slot) * 4;
                gx[mm] = (g4[0] + g4[1]) + (g4[2] + g4[3]);   // levels 5, 6
            });
            const double Sv = y[0], Nv = y[4], Dv = y[5];
            double kr[NC];
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
                // ode_determine_initdt, first half: only the n components of lambda are under error control
                f0l[lm < NC ? lm : 7] = ko;
                kl[0] = ko;
                double lam[NC];
                bcast(zo, lam);
                double h0 = 0.0, l0 = 0.0, h1 = 0.0, l1 = 0.0;
                static_for<0, NC>([&](auto c) {
                    const double sk = __builtin_fma(fabs(lam[c]), o.reltol, o.abstol);
                    const double q0 = lam[c] / sk, q1 = kr[c] / sk;
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
                    ph = PH_IDLE;
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
            } else if (ph == PH_INIT1) {
                double lam[NC];
                bcast(zo, lam);
                double h2 = 0.0, l2 = 0.0;
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
            } else {
                const int s = ph;
                kl[8 * s] = ko;
                if (s < S - 1) {
                    ph += 1;
                } else if (replay) {
                    // the rejected attempt has been taken back out of the accumulators: on with the reduced step (accept is still false)
                    replay = false;
                    ph = 0;
                } else {
                    // perform_step! is complete: new state, error estimate over lambda, controller
                    nfc += Tab::FSAL ? S - 1 : S;
                    double zn;
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
                        if (bad) { ret = RET_UNSTABLE; fin = true; }
                        if (t == tstop) {
                            const bool modified = at_tstop(t);
                            if (tstop == T0) fin = true;   // done
                            else {
                                tstop = tstop_from_cur();
                                if (modified && Tab::FSAL) nfc += 1;   // reset_fsal! after u_modified! (counted as upstream does)
                            }
                        }
                        ph = 0;
                    } else {
                        nrej += 1;
                        if (EEst != EEst) { ret = RET_UNSTABLE; fin = true; }
                        else { replay = true; ph = 0; }   // take the attempt's contributions back before the step is repeated
                    }
                    if (fin) {
                        ph = PH_IDLE;   // (nothing is pending: the slot takes its next trajectory in the next trip)
                        results();
                    }
                }
            }
        }
    }

    // ---- the block's row of the partial-gradient matrix ----
    __syncthreads();
    if (lm < NOUT) MB4[slot * 8 + lm] = mb4;
    __syncthreads();
    double* row = p.grad_part + (size_t)blockIdx.x * p.n_param;
    if (__syncthreads_or(bwd_failed)) {   // (block-uniform)
        for (int i = tid; i < p.n_param; i += BLOCKT) row[i] = __builtin_nan("");
        return;
    }
    static_for<0, 4>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int unit = 16 * w + kq + 4 * r;    // accumulator rows are NOT permuted: tile row i = kq + 4r is unit 16w + i
        static_for<0, 4>([&](auto cc) {
            row[OFF_W2 + unit + (16 * (int)decltype(cc)::value + jc) * H] = gW2[cc][r];
            row[OFF_W3 + unit + (16 * (int)decltype(cc)::value + jc) * H] = gW3[cc][r];
        });
        if (jc == 0) { row[OFF_B2 + unit] = gB2[r]; row[OFF_B3 + unit] = gB3[r]; }
        if (jc < NIN) row[OFF_W1 + unit + jc * H] = gW1[r];
        if (jc == NIN) row[OFF_B1 + unit] = gW1[r];
        if (jc < NOUT) row[OFF_W4 + jc + unit * NOUT] = gW4[r];
    });
    if (tid < NOUT) {
        double s = MB4[tid];
        for (int i = 1; i < NSLOTS; ++i) s += MB4[i * 8 + tid];
        row[OFF_B4 + tid] = s;
    }
}

}  // namespace nodelf
}  // namespace ude
