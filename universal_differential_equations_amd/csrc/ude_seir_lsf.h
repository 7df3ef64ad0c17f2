// ude_seir_lsf.h -- the `fast` interpolating adjoint (UDE_SENSE_INTERPOLATING_ADJOINT_FAST: only lambda under error control, the
// parameter cotangent mu a quadrature) of the SEIR exposure UDE (dudt_, SEIR_exposure/seir_exposure.jl:114-147) on the lock-step
// architecture of ude_seir_ls.h, with the parameter cotangent as a BLOCK-LEVEL MATRIX-CORE ACCUMULATION (SURVEY.md 7 step 6, 8(d) C3:
// "`fast` mode: none [HBM bytes] ... FP64 MFMA is the bound").
//
// In the fast mode mu never feeds back into the solve, so  dL/dW2 = - sum over (trajectory, step, stage) (dt b_s) delta2 (x) a1  is a
// [64 x K] . [K x 64] product whose K dimension may be ANY enumeration of the evaluations.  The sixteen slot columns of a trip ARE
// such a K block: T_D2[unit][slot] and T_A1[unit][slot], the LDS tiles the network already leaves behind, are its A and B operands.
// So every trip ends with
//      dW2 += (-(w_k delta2[:, k])) . a1[:, k]^T      w_k = dt_k b_{s_k} of slot k (0 for a slot that is idle, in its initial-dt
//      db2 += (-(w_k delta2[:, k])) . 1                evaluations or at a stage with b_s = 0)
//      dW1 | db1 += (-(w_k delta1[:, k])) . [x0 x1 x2 1]_k^T
//      dW3 += a2[:, k] . (-(w_k delta3_k))
// on v_mfma_f64_16x16x4 into accumulators that stay in the registers of the block for the whole launch (wavefront w: rows 16w..16w+15;
// 7 x 4 doubles per lane), and ONE row of 4481 doubles per block leaves the chip at the end.  No mu in HBM, no per-stage factor
// workspace, no step-end parameter pass, no per-trajectory gradient row: the kernel reads the forward records and writes dL/du0.
//
// What that needs:
//   * the accumulation happens when the stage is evaluated, i.e. BEFORE the step's accept / reject decision.  A rejected attempt
//     (rare: the controller aims at acceptance) is REPLAYED -- the same stages at the same points, bit-identical factors -- with the
//     weights negated, then the attempt with the reduced step follows.  The oracle's FAST_MM mode restates exactly this sequence;
//   * ARITH-SPEC of this mode (oracle/ude_oracle_impl.h, UDEO_SENSE_FAST_MM): every parameter cotangent is ONE fused chain
//     mu = fma(-((dt b_s) delta), a, mu) over the evaluations in the order the block makes them (what the matrix core executes:
//     d = fma(a_k, b_k, d), k ascending -- tools/probe/mfma_order_probe.hip).  A single trajectory is bit-identical to the oracle
//     (every one of the 4481 entries); several trajectories of a block interleave in one chain, blocks are added in block order
//     (<= 1e-12 relative to the oracle's sum over trajectories, as for every other kernel);
//   * trajectories are dealt to the blocks round-robin (trajectory g -> block g mod nblocks, slot (g / nblocks) mod 16): no queue,
//     so the order of every chain -- and with it every bit of the result -- is the same in every run;
//   * a column whose slot makes no evaluation is kept FINITE (zero inputs, zero weight): fma(-0, finite, acc) == acc.
// Lean enough for TWO blocks per compute unit (launch bounds 256 x 2: <= 256 registers per lane, 73 KB of LDS): the row phases of
// one block (interval lookup, interpolation, controller: dependent scalar chains) overlap the matrix phase of the other.
// Per trajectory the backward step counts and dL/du0 are bit-identical to the oracle's fast mode (the lambda solve is the same
// whichever way mu is accumulated).  Float64, shared time grid.
#pragma once
#include "ude_seir_ls.h"

namespace ude {
namespace seirlf {

using seirls::v4d;
using seirls::H;
using seirls::NSLOTS;
using seirls::BLOCKT;
using seirls::NC;
using seirls::TLD;
using seirls::TABL;
using seirls::kst;
using seirls::rshfl;
using seirls::OFF_W1;
using seirls::OFF_B1;
using seirls::OFF_W2;
using seirls::OFF_B2;
using seirls::OFF_W3;
using seirls::OFF_B3;

// (round 5 measured four variants of this kernel and kept none -- two blocks per CU with W2 re-fetched every trip, the small products deferred
//  into the next trip, the small products on the vector unit, W2 requested in front of barrier 1: profiles/r05_experiments.md, HISTORY.md;
//  their branches are gone from this source since round 6)

enum { PH_IDLE = -4, PH_FLUSH = -3, PH_INIT0 = -2, PH_INIT1 = -1 };   // >= 0: stage s of a step attempt

template <class Tab>
constexpr int lds_doubles() {
    return 4 * H * TLD + 2 * (4 * 16 + 16 + 16) + 3 * NSLOTS * 4 + NSLOTS * 16 + NSLOTS * 8 + TABL + 16 * 8 + NSLOTS * 8 + NSLOTS * kst<Tab>() + 3 * H +
           NSLOTS * Tab::S * 8 + NSLOTS + 2 * H;
}

// GEN = true (round 5): the RUNTIME-SHAPE instance -- any exposure-UDE chain 3 -> H1 -> H2 -> 1 (tanh, tanh, identity), H1, H2 <= 64 (the set
// udecore.hip's seir_gen_ls_shape admits), weights zero-padded to 64 x 64, every product of the network in the association of ITS length (ude_seir_ls2.h); the block's
// accumulators hold the padded 64 x 64 gradient (a padded unit has delta = 0 and a = 0: exact zeros), written through the runtime offsets
template <class Tab, bool GEN = false>
__global__ void __launch_bounds__(BLOCKT, 1) seir_lsf_adj_kernel(const KParams p, double* __restrict__ /*unused: no factor workspace*/,
                                                                              int* __restrict__ /*unused: no queue*/) {
    constexpr int S = Tab::S, NK = Tab::NK;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* T_A1 = sm;                        // [64][17] a1[unit][slot]
    double* T_D2 = T_A1 + H * TLD;            // delta2
    double* T_D1 = T_D2 + H * TLD;            // delta1
    double* T_A2 = T_D1 + H * TLD;            // a2
    // per-slot inputs of a trip, TWO copies (trip parity): the small products of the parameter cotangent of trip T are formed in trip
    // T + 1, under the first layer's tanh (see "the parameter cotangent" below), when the slots' rows have already written trip T + 1's
    double* XIN0 = T_A2 + H * TLD;            // [2][ [4][16]: x0 x1 x2 1 | [16] delta3 | [16] weight dt b_s (0: contributes nothing) ]
    constexpr int XSZ = 4 * 16 + 16 + 16;
    double* GXP = XIN0 + 2 * XSZ;             // [3][16][4]: per wavefront partial sums of the input cotangent
    double* BQ = GXP + 3 * NSLOTS * 4;        // [16][16]
    double* YS = BQ + NSLOTS * 16;            // [16][8]
    double* TB = YS + NSLOTS * 8;             // tableau: A[16][16], B, BT, C
    double* RQL = TB + TABL;                  // [16 lanes q][8]: Horner tables of b_q(theta)
    double* F0L = RQL + 16 * 8;               // [16 slots][8]: f0 of the initial-dt phase
    double* KSL = F0L + NSLOTS * 8;           // [16 slots][KST]: the stored record of the slot's current forward interval
    double* W1L = KSL + NSLOTS * kst<Tab>();  // [3][64]
    double* KL = W1L + 3 * H;                 // [16 slots][S][8]: stage derivatives of lambda, component c of the slot at [.][c]
    double* MB3 = KL + NSLOTS * S * 8;        // [16] the slots' shares of db3 (end of the kernel)
    double* B2L = MB3 + NSLOTS;               // [64] b2, [64] w3 (read where they are used: sixteen registers per lane less)
    double* W3L = B2L + H;

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int kq = l >> 4, jc = l & 15;       // matrix view: k index / column (= slot) of this lane
    const int rr = l >> 4, lm = l & 15;       // scalar view: row of the wavefront (slot 4w + rr), lane inside the row
    const int slot = 4 * w + rr;
    const double* __restrict__ th = p.theta;
    const TabDev* __restrict__ tab = p.tab;
    const int n = NC;
    const int nfld = 3 + n + NK * n;

    // ---- weights.  The output rows of this wavefront's tiles are PERMUTED: register r of lane (kq, .) is hidden unit 16w + 4kq + r
    // (tile row i = kq + 4r  <->  unit 16w + 4 (i & 3) + (i >> 2)), so that a lane holds four CONSECUTIVE units: the first two levels of
    // the adjacent-pair tree of the input cotangent are in-lane, the next two are two lane exchanges, the last two cross the wavefronts
    // through 192 words of LDS -- no [slot][row] product tiles (25 KB) ----
    const int urow = 16 * w + 4 * (jc & 3) + (jc >> 2);   // the unit whose weights this lane supplies as A-operand row jc
    // the two A-operand fragments of W2 and of W2^T (2 x 32 doubles per lane), resident in registers for the whole launch
    // layer widths and the offsets of theta = [W1 (H1 x 3) | b1 | W2 (H2 x H1) | b2 | W3 (1 x H2) | b3] (column-major)
    const int H1 = GEN ? p.mc.dims[1] : H, H2 = GEN ? p.mc.dims[2] : H;
    const int oW1 = GEN ? 0 : OFF_W1, oB1 = GEN ? 3 * H1 : OFF_B1, oW2 = GEN ? oB1 + H1 : OFF_W2, oB2 = GEN ? oW2 + H1 * H2 : OFF_B2,
              oW3 = GEN ? oB2 + H2 : OFF_W3, oB3 = GEN ? oW3 + H2 : OFF_B3;
    const bool blk_fwd = H1 == H, blk_bwd = H2 == H;   // a 64-term product: four 16-term chains; a shorter one: ONE ascending chain
    double W2A[16], W2T[16];
    static_for<0, 16>([&](auto sc) {
        const int col = 4 * decltype(sc)::value + kq;
        W2A[sc] = (!GEN || (urow < H2 && col < H1)) ? th[oW2 + urow + col * H2] : 0.0;      // A[i][k] = W2[unit(i)][4s + k]
        W2T[sc] = (!GEN || (col < H2 && urow < H1)) ? th[oW2 + col + urow * H2] : 0.0;      // A[i][k] = W2[4s + k][unit(i)]
    });
    const double W1A = (!GEN || urow < H1) ? (kq < 3 ? th[oW1 + urow + kq * H1] : th[oB1 + urow]) : 0.0;
    for (int i = tid; i < 3 * H; i += BLOCKT) W1L[i] = (!GEN || (i % H) < H1) ? th[oW1 + (i % H) + (i / H) * H1] : 0.0;
    if (tid < H) { B2L[tid] = (!GEN || tid < H2) ? th[oB2 + tid] : 0.0; W3L[tid] = (!GEN || tid < H2) ? th[oW3 + tid] : 0.0; }
    const int u0r = 16 * w + 4 * kq;          // first of this lane's four units
    double b2r[4], w3r[4];
    static_for<0, 4>([&](auto r) {
        const int un = u0r + decltype(r)::value;
        b2r[r] = (!GEN || un < H2) ? th[oB2 + un] : 0.0;
        w3r[r] = (!GEN || un < H2) ? th[oW3 + un] : 0.0;
    });
#define LSF_B2(r) b2r[r]
#define LSF_W3(r) w3r[r]
    const double Fc = p.mc.consts[0], b0c = p.mc.consts[1], muc = p.mc.consts[4], sgc = p.mc.consts[5], gac = p.mc.consts[6],
                 dc = p.mc.consts[7], lac = p.mc.consts[8];
    // every column finite from the first trip on (a column without an evaluation multiplies its zero weight with what the tiles hold)
    for (int i = tid; i < 4 * H * TLD; i += BLOCKT) T_A1[i] = 0.0;
    if (tid < 2 * XSZ) XIN0[tid] = ((tid % XSZ) >= 48 && (tid % XSZ) < 64) ? 1.0 : 0.0;
    if (tid < 16) MB3[tid] = 0.0;
    double* const XIN = XIN0;
    double* const D3S = XIN + 4 * 16;
    double* const WSL = D3S + 16;
    for (int i = tid; i < 16 * 16; i += BLOCKT) TB[i] = tab->A[i >> 4][i & 15];
    if (tid < 16) { TB[256 + tid] = tab->B[tid]; TB[272 + tid] = tab->BT[tid]; TB[288 + tid] = tab->C[tid]; }
    for (int i = tid; i < NSLOTS * 8; i += BLOCKT) { YS[i] = 0.0; F0L[i] = 0.0; }
    for (int i = tid; i < NSLOTS * S * 8; i += BLOCKT) KL[i] = 0.0;
    for (int i = tid; i < 16 * 8; i += BLOCKT) RQL[i] = ((i >> 3) < NK && (i & 7) < 7) ? tab->R[i >> 3][i & 7] : 0.0;

    // ---- the block's share of the gradient: stays in registers until the end ----
    v4d gW2[4], gB2, gW1, gW3;
    static_for<0, 4>([&](auto c) { gW2[c] = v4d{0.0, 0.0, 0.0, 0.0}; });
    gB2 = v4d{0.0, 0.0, 0.0, 0.0}; gW1 = gB2; gW3 = gB2;
    double mb3 = 0.0;   // (row view, lane 0 of the slot's row) db3 share of this slot: fma(-(w d3), 1, mb3)

    // ---- per-slot state on the slot's row: component c on lane c ----
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
    double* const kl = KL + (size_t)slot * S * 8 + (lm < NC ? lm : 7);   // K[j] of this lane's component at kl[8 j]
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
        pf_want = s - 1;   // fetched inside the matrix phase
    };
    auto bcast = [&](double ownv, double (&out)[NC]) { static_for<0, NC>([&](auto c) { out[c] = rshfl(ownv, decltype(c)::value); }); };
    auto SV = [&](int i) { return p.saveat[i]; };
    auto tstop_from_cur = [&]() { return (cur >= 0 && SV(cur) > T0) ? SV(cur) : T0; };
    // the jump(s) at a save time: lambda += dL/du (this lane's component)
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
    // db2, dW1 | db1, dW3 of one trip from the tiles and that trip's copy of the slots' inputs
    auto small_products = [&](const double* xin) {
        const double* d3s = xin + 4 * 16;
        const double* wsl = d3s + 16;
        static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int k = 4 * q + kq;
            const double wk = wsl[k];
            const int rown = (16 * w + jc) * TLD + k;
            const double Ad2 = -(wk * T_D2[rown]);
            const double Ad1 = -(wk * T_D1[rown]);
            const double Aa2 = T_A2[rown];
            gB2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad2, jc == 0 ? 1.0 : 0.0, gB2, 0, 0, 0);
            gW1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad1, jc < 4 ? xin[(jc & 3) * 16 + k] : 0.0, gW1, 0, 0, 0);
            gW3 = __builtin_amdgcn_mfma_f64_16x16x4f64(Aa2, jc == 0 ? -(wk * d3s[k]) : 0.0, gW3, 0, 0, 0);
        });
    };
    __syncthreads();
#if defined(UDE_LSF_CLOCKS)   // timing experiment: cycles of wavefront 0 of block 0 per section of a trip (tools/lsf_prof.py)
    unsigned long long tk = __builtin_readcyclecounter(), tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ntrip = 0;
#define LSF_TICK(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tsec[i] += now_ - tk; tk = now_; }
#else
#define LSF_TICK(i)
#endif

    for (;;) {
        // ---- A. an idle slot takes its next trajectory: g = block + nblocks (slot + 16 j) ----
        if (ph == PH_IDLE && !exhausted) {
            for (;;) {
                const long long g = (long long)blockIdx.x + (long long)nblk * (slot + 16ll * jtraj);
                jtraj += 1;
                if (g >= p.N) { exhausted = true; break; }
                if (p.retcode[g] != RET_SUCCESS) continue;   // (its forward solve failed: it contributes nothing)
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
                at_tstop(TF);   // init_cb: the jump at t = tf precedes the first step
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
                // the weight of this evaluation in the quadrature: dt b_s; a rejected attempt is taken back by its replay
                const double bs = TB[256 + s];
                wgt = bs != 0.0 ? (replay ? -(dt * bs) : dt * bs) : 0.0;
            } else {
                ph = PH_FLUSH;   // ended with an error: this trip without an evaluation, idle from the next
                results();
            }
        }
        double zs[NC];
        bcast(zsrc, zs);
        LSF_TICK(0)

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
            const double x0 = y[0] / y[4], x1 = y[2], x2 = y[5] / y[4];
            const double d3 = (zs[1] - zs[0]) * 1.0;
            if (lm == 0) {
                XIN[0 * 16 + slot] = x0; XIN[1 * 16 + slot] = x1; XIN[2 * 16 + slot] = x2;
                D3S[slot] = d3;
                WSL[slot] = wgt;
                mb3 = wgt != 0.0 ? mb3 + (-(wgt * d3)) : mb3;   // db3: delta3 times 1
            }
        } else if (lm == 0) {   // no evaluation: a finite column with zero weight
            XIN[0 * 16 + slot] = 0.0; XIN[1 * 16 + slot] = 0.0; XIN[2 * 16 + slot] = 0.0;
            D3S[slot] = 0.0;
            WSL[slot] = 0.0;
        }
        LSF_TICK(1)
        if (!__syncthreads_or(ph != PH_IDLE)) break;   // (all slots idle, no trajectory left: done)
        LSF_TICK(2)
        {
            // layer 1 (3 inputs + bias in one k-step)
            v4d z = __builtin_amdgcn_mfma_f64_16x16x4f64(W1A, XIN[kq * 16 + jc], v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            // the small products of the PREVIOUS trip's parameter cotangent (db2, dW1 | db1, dW3: 12 MFMAs whose operands -- the delta2,
            // delta1, a2 tiles and the other copy of the slots' inputs -- are untouched until the next barrier): the matrix pipe works
            // on them while the vector unit evaluates the four tanh below
            double a1[4], dv1[4];
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                a1[r] = dtanh(z[r]);
                T_A1[(u0r + r) * TLD + jc] = a1[r];
            });
            __syncthreads();
            LSF_TICK(3)
            if (pf_want >= 0) { fetch_interval(pf_want); pf_want = -1; }
            // hidden layer: four 16-term chains (four MFMAs each) added left to right
            {
                v4d acc[4];
                if (!GEN || blk_fwd) {
                    static_for<0, 4>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
                        static_for<0, 4>([&](auto q) {
                            constexpr int s = 4 * b + decltype(q)::value;
                            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2A[s], T_A1[(4 * s + kq) * TLD + jc], acc[b], 0, 0, 0);
                        });
                    });
                } else {   // fewer than 64 inputs: one ascending chain (the trailing zero terms are exact)
                    acc[0] = v4d{0.0, 0.0, 0.0, 0.0};
                    static_for<0, 16>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2A[s], T_A1[(4 * s + kq) * TLD + jc], acc[0], 0, 0, 0);
                    });
                }
                const double d3j = D3S[jc];
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const double z2 = ((!GEN || blk_fwd) ? (((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r]) : acc[0][r]) + LSF_B2(r);
                    const double a2 = dtanh(z2);
                    const double d2 = __builtin_fma(LSF_W3(r), d3j, 0.0) * __builtin_fma(-a2, a2, 1.0);
                    T_D2[(u0r + r) * TLD + jc] = d2;
                    T_A2[(u0r + r) * TLD + jc] = a2;
                });
            }
            __syncthreads();
            LSF_TICK(4)
            // transposed hidden layer on the deltas
            {
                v4d acc[4];
                if (!GEN || blk_bwd) {
                    static_for<0, 4>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
                        static_for<0, 4>([&](auto q) {
                            constexpr int s = 4 * b + decltype(q)::value;
                            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2T[s], T_D2[(4 * s + kq) * TLD + jc], acc[b], 0, 0, 0);
                        });
                    });
                } else {
                    acc[0] = v4d{0.0, 0.0, 0.0, 0.0};
                    static_for<0, 16>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2T[s], T_D2[(4 * s + kq) * TLD + jc], acc[0], 0, 0, 0);
                    });
                }
                double pg[3] = {0.0, 0.0, 0.0};
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const double s1 = (!GEN || blk_bwd) ? (((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r]) : acc[0][r];
                    dv1[r] = s1 * __builtin_fma(-a1[r], a1[r], 1.0);
                    T_D1[(u0r + r) * TLD + jc] = dv1[r];
                });
                // input cotangent: rounded products W1[u][m] delta1[u] under the adjacent-pair tree over the 64 units
                static_for<0, 3>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    const double* wl = W1L + m * H + u0r;
                    double x = (wl[0] * dv1[0] + wl[1] * dv1[1]) + (wl[2] * dv1[2] + wl[3] * dv1[3]);   // levels 1, 2: this lane's four units
                    x += __shfl_xor(x, 16, 64);                                                          // level 3
                    x += __shfl_xor(x, 32, 64);                                                          // level 4
                    pg[m] = x;
                });
                if (kq == 0) static_for<0, 3>([&](auto mc) { GXP[(decltype(mc)::value * NSLOTS + jc) * 4 + w] = pg[decltype(mc)::value]; });
            }
            LSF_TICK(5)
            // ---- the parameter cotangent of this trip: K = the sixteen slot columns, weights on the delta side.  dW2 here (a1 is
            // overwritten by the next trip's first layer); the small products follow in the next trip (small_products) ----
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int k = 4 * q + kq;
                const double wk = WSL[k];
                const double Ad2 = -(wk * T_D2[(16 * w + jc) * TLD + k]);     // A operand: row jc of this wavefront's tile = unit 16w + jc (not permuted)
                static_for<0, 4>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    gW2[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad2, T_A1[(16 * c + jc) * TLD + k], gW2[c], 0, 0, 0);
                });
            });
            small_products(XIN);
        }
        __syncthreads();
        LSF_TICK(6)

        // ---- D. the slot's row: state cotangent of this evaluation, then its state machine ----
        if (ph == PH_FLUSH) {
            ph = PH_IDLE;
        } else if (ev) {
            double gx[3];
            static_for<0, 3>([&](auto mm) {
                const double* g4 = GXP + (decltype(mm)::value * NSLOTS + slot) * 4;
                gx[mm] = (g4[0] + g4[1]) + (g4[2] + g4[3]);   // levels 5, 6
            });
            if (GEN && !blk_fwd) {
                // fewer than 64 units in the first layer: the input cotangent is ONE ascending chain over the units (wide_dot, n < 64;
                // n = 32 is the tree case and not served), formed by lane m < 3 of the slot's row from the delta1 tile
                double acc = 0.0;
                const double* wl = W1L + (lm < 3 ? lm : 0) * H;
                for (int u = 0; u < H1; ++u) acc = __builtin_fma(wl[u], T_D1[u * TLD + slot], acc);
                static_for<0, 3>([&](auto mm) { gx[mm] = rshfl(acc, decltype(mm)::value); });
            }
            const double Sv = y[0], Nv = y[4], Dv = y[5];
            const double cc = b0c * Fc / Nv;
            const double cN = b0c * Sv * Fc / (Nv * Nv);
            double kr[NC];
            kr[0] = -((-cc - muc) * zs[0] + cc * zs[1] + gx[0] / Nv);
            kr[1] = -(-(sgc + muc) * zs[1] + sgc * zs[2] + sgc * zs[6]);
            kr[2] = -(-(gac + muc) * zs[2] + gac * zs[3] + dc * gac * zs[5] + gx[1]);
            kr[3] = -(-muc * zs[3]);
            kr[4] = -(cN * zs[0] - cN * zs[1] - muc * zs[4] - gx[0] * Sv / (Nv * Nv) - gx[2] * Dv / (Nv * Nv));
            kr[5] = -(-lac * zs[5] + gx[2] / Nv);
            kr[6] = -0.0;
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
        LSF_TICK(7)
#if defined(UDE_LSF_CLOCKS)
        ntrip += 1;
#endif
    }

#if defined(UDE_LSF_CLOCKS)
    if (p.trace && blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 8; ++i) p.trace[i] = (double)tsec[i];
        p.trace[8] = (double)ntrip;
    }
#endif
    // ---- the block's row of the partial-gradient matrix (every entry written: a block without trajectories writes zeros) ----
    __syncthreads();
    if (lm == 0) MB3[slot] = mb3;
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
            const int c1 = 16 * (int)decltype(cc)::value + jc;
            if (!GEN || (unit < H2 && c1 < H1)) row[oW2 + unit + c1 * H2] = gW2[cc][r];
        });
        if (jc == 0 && (!GEN || unit < H2)) { row[oB2 + unit] = gB2[r]; row[oW3 + unit] = gW3[r]; }
        if (!GEN || unit < H1) {
            if (jc < 3) row[oW1 + unit + jc * H1] = gW1[r];
            if (jc == 3) row[oB1 + unit] = gW1[r];
        }
    });
    if (tid == 0) {
        double s = MB3[0];
        for (int i = 1; i < NSLOTS; ++i) s += MB3[i];
        row[oB3] = s;
    }
}

}  // namespace seirlf
}  // namespace ude
