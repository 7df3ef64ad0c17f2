// ude_node_ls.h -- the interpolating adjoint of the SEIR script's pure neural ODE (dudt_node, SEIR_exposure/seir_exposure.jl:53-83:
// FastChain 7-64-64-64-7 tanh, 9287 parameters) on the lock-step architecture of ude_seir_ls.h: sixteen trajectory slots of a
// persistent block as the columns of v_mfma_f64_16x16x4, every slot its own little state machine, trajectories from a queue.
//
// What differs from the exposure UDE:
//   * two 64x64 layers.  Their weights do NOT live in registers (the two pairs of A-operand fragments would be 128 registers per
//     lane): W2 and W3 sit once per block in LDS and every wavefront reads the fragment it needs right in front of the matrix
//     product that uses it (16 words per lane and product);
//   * first layer 7 -> 64: two k-steps, the eighth term adds the bias (fma(b, 1, acc) == acc + b); output layer 64 -> 7 only
//     enters transposed (delta3 = W4^T delta4: a 7-term chain, the eighth term a zero): two k-steps as well;
//   * the seven input-cotangent sums (rounded products, adjacent-pair tree over the 64 hidden rows) are reduced where the
//     products are: row i = 16w + 4r + kq lives in register r of lane (kq, slot) of wavefront w, so the first two tree levels
//     are lane exchanges (xor 16, xor 32), the next two are in-lane, the last two cross the wavefronts through 7 x 16 x 4 words of LDS;
//   * 146 parameter slots per hidden row (W2 and W3 rows, W1 row, b1 b2 b3, W4 column, b4): per stage the factors a1 a2 a3 delta1
//     delta2 delta3 go to an HBM workspace, the step-end pass stages the 2 x 16 columns of a1 / a2 a wavefront needs in LDS.
// ARITH-SPEC is SeirNode<64>'s, operation for operation: per trajectory every number is bit-identical to adj_kernel<SeirNode<64>>
// and to the oracle.  Float64, shared time grid, parity mode.
#pragma once
#include "ude_seir_ls.h"

namespace ude {
namespace nodels {

using seirls::v4d;
using seirls::stage_mask;
using seirls::popc;
using seirls::cslot;
using seirls::rshfl;
using seirls::TABL;
constexpr int H = 64, NSLOTS = 16, BLOCKT = 256, NC = 7, NIN = 7, NOUT = 7, TLD = 17;
constexpr int NEX = NIN + 3 + NOUT + 1;   // extra slots of a hidden row: W1[j, 0..6], b1, b2, b3, W4[0..6, j], b4[j] (rows 0..6)
constexpr int NSLK = 2 * H + NEX;         // 146
constexpr int OFF_W1 = 0, OFF_B1 = NIN * H, OFF_W2 = OFF_B1 + H, OFF_B2 = OFF_W2 + H * H, OFF_W3 = OFF_B2 + H, OFF_B3 = OFF_W3 + H * H,
              OFF_W4 = OFF_B3 + H, OFF_B4 = OFF_W4 + NOUT * H;
constexpr int NFAC = 6;                   // factor rows per stage in HBM: a1 a2 a3 delta1 delta2 delta3
constexpr int QW = H / 4;                 // columns of a weight block per wavefront
constexpr int XFW = 14;
constexpr int LDW = 65;                  // leading dimension of the LDS copies of W2 / W3: row AND column fragments conflict-free                   // x0..x6 | delta4_0..6 per (stage, slot)

#ifndef NL_PF_AT
#define NL_PF_AT 3   // where the prefetch of the next-lower forward interval is issued: 0 at the switch itself, 1 / 2 / 3 inside the matrix phase (3: in front of the third layer)
#endif
#define NL_PF_ISSUE if (pf_want >= 0) { fetch_interval(pf_want); pf_want = -1; }
template <class Tab>
constexpr int lds_doubles() {
    constexpr int NSTC = popc(stage_mask<Tab>());
    return 2 * H * LDW + 4 * H * TLD + 8 * 16 + 8 * 16 + NIN * NSLOTS * 4 + NSTC * NSLOTS * XFW + NSLOTS * 16 + NSLOTS * 8 + TABL + 6 * NSLOTS +
           NSLOTS * 4 * 2 + NSLOTS * seirls::kst<Tab>() + NSLOTS * 8 + 16 * 8 + NSLOTS * 16 + 4 * 2 * NSTC * QW + NIN * H;
}
template <class Tab>
constexpr size_t fac_doubles_per_block() { return (size_t)NSLOTS * popc(stage_mask<Tab>()) * NFAC * H; }

enum { PH_IDLE = -4, PH_FLUSH = -3, PH_INIT0 = -2, PH_INIT1 = -1 };
enum { RQ_NONE = -1, RQ_STEP = 0, RQ_NORM01 = 1, RQ_NORM2 = 2, RQ_FLUSH = 4 };

// extra slot e of hidden row `lane`: theta index (or -1)
__device__ __forceinline__ int extra_index(int e, int lane) {
    if (e < NIN) return OFF_W1 + lane + e * H;
    if (e == NIN) return OFF_B1 + lane;
    if (e == NIN + 1) return OFF_B2 + lane;
    if (e == NIN + 2) return OFF_B3 + lane;
    if (e < NIN + 3 + NOUT) return OFF_W4 + (e - NIN - 3) + lane * NOUT;
    return lane < NOUT ? OFF_B4 + lane : -1;
}

// The 146 parameter slots of ONE trajectory, a quarter per wavefront (lane i = hidden row i): wavefront w takes columns 16w .. 16w+15
// of the W2 block and of the W3 block and the extra slots e with e % 4 == w.  The loops of SeirNode<64>::step_slots / init_norm01 /
// init_norm2:  MODE 0 end of a step (candidate mu_new, returns this lane's sum of squared residuals), MODE 1 / 2 initial-dt norms.
//   fbase: this trajectory's factors [cs][a1 a2 a3 delta1 delta2 delta3][64];  stg: this wavefront's LDS staging [2][NSTC][16];
//   xf: LDS [cs][slot][14]
template <int NST, unsigned MASK, int MODE>
__device__ __forceinline__ double slot_pass(const double* __restrict__ fbase, double* stg, const double* xf, int slot, int lane, int w,
                                            const double* Bw, const double* BTw, double dt, double abstol, double reltol,
                                            const double* __restrict__ mu, double* __restrict__ mu_new, double& hh, double& ll) {
    constexpr int CH = 8;
    constexpr int NSTC = popc(MASK);
    // loads first: the per-lane delta rows, the two 16-column strips of a1 / a2 (lanes 0..15 and 16..31), mu of the W2 strip
    double d2[NST], d3[NST];
    static_for<0, NST>([&](auto s) {
        if constexpr ((MASK >> decltype(s)::value) & 1u) {
            constexpr int cs = cslot<MASK>(decltype(s)::value);
            d2[s] = fbase[(size_t)(cs * NFAC + 4) * H + lane];
            d3[s] = fbase[(size_t)(cs * NFAC + 5) * H + lane];
        }
    });
    double av[NST];
    static_for<0, NST>([&](auto s) {
        if constexpr ((MASK >> decltype(s)::value) & 1u) {
            constexpr int cs = cslot<MASK>(decltype(s)::value);
            av[s] = lane < 32 ? fbase[(size_t)(cs * NFAC + (lane >> 4)) * H + QW * w + (lane & 15)] : 0.0;   // a1 strip | a2 strip
        }
    });
    double mcur[CH], mnext[CH];
    static_for<0, CH>([&](auto i) {
        mcur[i] = MODE == 0 ? mu[(size_t)(QW * w + decltype(i)::value) * H] : 0.0;
        mnext[i] = MODE == 0 ? mu[(size_t)(QW * w + CH + decltype(i)::value) * H] : 0.0;
    });
    static_for<0, NST>([&](auto s) {
        if constexpr ((MASK >> decltype(s)::value) & 1u) {
            constexpr int cs = cslot<MASK>(decltype(s)::value);
            if (lane < 32) stg[((lane >> 4) * NSTC + cs) * QW + (lane & 15)] = av[s];
        }
    });
    double bb[NST], bt[NST];
    if constexpr (MODE == 0) static_for<0, NST>([&](auto s) { bb[s] = Bw[decltype(s)::value]; bt[s] = BTw[decltype(s)::value]; });
    double ps = 0.0;
    auto body = [&](int sl, const double* g, double m0) {
        if constexpr (MODE == 0) {
            double ab = bb[0] * g[0], ae = bt[0] * g[0];
            static_for<1, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) {
                    ab = __builtin_fma(bb[s], g[s], ab);
                    ae = __builtin_fma(bt[s], g[s], ae);
                }
            });
            const double m1 = __builtin_fma(dt, ab, m0);
            mu_new[(size_t)sl * H] = m1;
            const double a0 = fabs(m0), a1 = fabs(m1);
            const double res = (dt * ae) / __builtin_fma((a0 > a1 ? a0 : a1), reltol, abstol);
            ps = __builtin_fma(res, res, ps);
        } else {
            const double sk = __builtin_fma(fabs(m0), reltol, abstol);
            const double q = MODE == 1 ? g[0] / sk : (g[NST - 1] - g[0]) / sk;
            dd_acc(hh, ll, q * q);
        }
    };
    // W2 block (delta2 x a1), then W3 block (delta3 x a2): 16 columns each, chunks of 8, the next chunk of mu always in flight
    static_for<0, 2>([&](auto lc) {
        constexpr int LAYER = decltype(lc)::value;
        const double* dl = LAYER == 0 ? d2 : d3;
        if constexpr (LAYER == 1)
            static_for<0, CH>([&](auto i) {
                mcur[i] = MODE == 0 ? mu[(size_t)(H + QW * w + decltype(i)::value) * H] : 0.0;
                mnext[i] = MODE == 0 ? mu[(size_t)(H + QW * w + CH + decltype(i)::value) * H] : 0.0;
            });
#pragma unroll 1
        for (int k0 = 0; k0 < QW; k0 += CH) {
            static_for<0, CH>([&](auto i) {
                const int k = k0 + decltype(i)::value;
                double g[NST];
                static_for<0, NST>([&](auto s) {
                    if constexpr ((MASK >> decltype(s)::value) & 1u) g[s] = -(dl[s] * stg[(LAYER * NSTC + cslot<MASK>(decltype(s)::value)) * QW + k]);
                });
                body(LAYER * H + QW * w + k, g, mcur[i]);
            });
            static_for<0, CH>([&](auto i) { mcur[i] = mnext[i]; });
        }
    });
    // the extra slots of this wavefront: e = w, w + 4, ... (a3 and delta1 of the lane from HBM, x | delta4 of the slot from LDS)
    double a3[NST], d1[NST], mex[5];
    static_for<0, NST>([&](auto s) {
        if constexpr ((MASK >> decltype(s)::value) & 1u) {
            constexpr int cs = cslot<MASK>(decltype(s)::value);
            a3[s] = fbase[(size_t)(cs * NFAC + 2) * H + lane];
            d1[s] = fbase[(size_t)(cs * NFAC + 3) * H + lane];
        }
    });
    static_for<0, 5>([&](auto i) { mex[i] = (MODE == 0 && w + 4 * (int)decltype(i)::value < NEX) ? mu[(size_t)(2 * H + w + 4 * (int)decltype(i)::value) * H] : 0.0; });
    static_for<0, NEX>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if ((e & 3) == w) {
            double g[NST];
            static_for<0, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) {
                    const double* xs = xf + (cslot<MASK>(decltype(s)::value) * NSLOTS + slot) * XFW;   // x0..x6 | delta4_0..6 of this stage
                    double v;
                    if constexpr (e < NIN) v = -(d1[s] * xs[e]);
                    else if constexpr (e == NIN) v = -d1[s];
                    else if constexpr (e == NIN + 1) v = -d2[s];
                    else if constexpr (e == NIN + 2) v = -d3[s];
                    else if constexpr (e < NIN + 3 + NOUT) v = -(xs[NIN + (e - NIN - 3)] * a3[s]);
                    else v = lane < NOUT ? -xs[NIN + (lane < NOUT ? lane : 0)] : -0.0;
                    g[s] = v;
                }
            });
            body(2 * H + e, g, mex[e >> 2]);
        }
    });
    return ps;
}

template <class Tab>
__global__ void __launch_bounds__(BLOCKT, 1) node_ls_adj_kernel(const KParams p, double* __restrict__ facws, int* __restrict__ queue) {
    constexpr int S = Tab::S, NK = Tab::NK;
    constexpr unsigned MASK = stage_mask<Tab>();
    constexpr int NSTC = popc(MASK);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* WL2 = sm;                          // [64][65]: W2[i][k] at WL2[i + k * LDW]
    double* WL3 = WL2 + H * LDW;
    double* T_A1 = WL3 + H * LDW;                // [64][17] tiles: B operands of the next product, and the rows of the factor copy
    double* T_A2 = T_A1 + H * TLD;
    double* T_D3 = T_A2 + H * TLD;
    double* T_D2 = T_D3 + H * TLD;
    double* XIN = T_D2 + H * TLD;              // [8][16]: x0..x6, 1
    double* D4S = XIN + 8 * 16;                // [8][16]: delta4_0..6, 0
    double* PGS = D4S + 8 * 16;                // [7][16 slots][4 wavefronts]: 16-row sums of the input-cotangent products
    double* XF = PGS + NIN * NSLOTS * 4;       // [NSTC][16][14]
    double* BQ = XF + NSTC * NSLOTS * XFW;     // [16][16]
    double* YS = BQ + NSLOTS * 16;             // [16][8]
    double* TB = YS + NSLOTS * 8;              // tableau: A[16][16], B, BT, C
    double* RDT = TB + TABL;                   // [16] step size of a step request
    long long* RG = reinterpret_cast<long long*>(RDT + NSLOTS);
    int* REQI = reinterpret_cast<int*>(RG + NSLOTS);
    int* REQZ = REQI + NSLOTS;
    int* RCOL = REQZ + NSLOTS;
    int* ROK = RCOL + NSLOTS;
    int* RCS = ROK + NSLOTS;
    int* REV = RCS + NSLOTS;
    double* SUMW = RDT + 6 * NSLOTS;           // [16][4][2]
    double* KSL = SUMW + NSLOTS * 4 * 2;       // [16 slots][KST]: interval cache, the stored record of the slot's current forward interval
    double* F0L = KSL + NSLOTS * seirls::kst<Tab>();   // [16 slots][8]: f0 of the initial-dt phase
    double* RQL = F0L + NSLOTS * 8;            // [16][8]
    double* ZK = RQL + 16 * 8;                 // [16][16]
    double* ASTG = ZK + NSLOTS * 16;           // [4 wavefronts][2][NSTC][16]
    double* W1L = ASTG + 4 * 2 * NSTC * QW;    // [7][64]: W1[i][m] at W1L[m * H + i] (read where the input-cotangent products are formed)

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int kq = l >> 4, jc = l & 15;
    const int rr = l >> 4, lm = l & 15;
    const int slot = 4 * w + rr;
    const double* __restrict__ th = p.theta;
    const TabDev* __restrict__ tab = p.tab;
    const int n = NC;
    const int nfld = 3 + n + NK * n;

    for (int i = tid; i < H * H; i += BLOCKT) { WL2[(i % H) + (i / H) * LDW] = th[OFF_W2 + i]; WL3[(i % H) + (i / H) * LDW] = th[OFF_W3 + i]; }
    // narrow layers as A-operand fragments: first layer rows 16w + jc, k = input (k = 7: the bias); W4^T rows 16w + jc, k = output row (k = 7: 0)
    double W1A[2], W4T[2];
    static_for<0, 2>([&](auto sc) {
        const int k = 4 * decltype(sc)::value + kq, row = 16 * w + jc;
        W1A[sc] = k < NIN ? th[OFF_W1 + row + k * H] : th[OFF_B1 + row];
        W4T[sc] = k < NOUT ? th[OFF_W4 + k + row * NOUT] : 0.0;
    });
    for (int i = tid; i < NIN * H; i += BLOCKT) W1L[i] = th[OFF_W1 + i];
    double b2r[4], b3r[4];
    static_for<0, 4>([&](auto r) {
        const int row = 16 * w + kq + 4 * decltype(r)::value;
        b2r[r] = th[OFF_B2 + row];
        b3r[r] = th[OFF_B3 + row];
    });
    const double muc = p.mc.consts[4], sgc = p.mc.consts[5];
    if (tid < 16) { XIN[7 * 16 + tid] = 1.0; D4S[7 * 16 + tid] = 0.0; }
    for (int i = tid; i < 16 * 16; i += BLOCKT) TB[i] = tab->A[i >> 4][i & 15];
    if (tid < 16) { TB[256 + tid] = tab->B[tid]; TB[272 + tid] = tab->BT[tid]; TB[288 + tid] = tab->C[tid]; }
    for (int i = tid; i < NSLOTS * 8; i += BLOCKT) { YS[i] = 0.0; SUMW[i] = 0.0; }
    for (int i = tid; i < 8 * 16; i += BLOCKT) { if (i < 7 * 16) { XIN[i] = 1.0; D4S[i] = 0.0; } }

    // ---- per-slot state (replicated in the 16 lanes of the slot's row; component c on lane c) ----
    const OptsR o(p.o);
    const double T0 = p.t0, TF = p.tf, tdir = -1.0;
    const double dtmax = o.dtmax;
    const double ntot = (double)(p.n_state + p.n_param);
    const bool user_dt = o.dt0 > 0.0;
    int ph = PH_IDLE, ret = RET_SUCCESS, col = 0, iter = 0, sf = 0, cur = 0, nsteps = 1;
    long long gid = 0;
    bool accept = true, exhausted = false, zero_req = false;
    double t = TF, dt = 0.0, dt0 = 0.0, d1n = 0.0, qold = o.qoldinit, q11 = 1.0, tstop = T0, ssrep = 0.0;
    long long nfc = 0, nacc = 0, nrej = 0;
    double lam[NC], K[S], ts = 0.0, te = 0.0;
    static_for<0, NC>([&](auto c) { lam[c] = 0.0; });
    static_for<0, S>([&](auto s) { K[s] = 0.0; });
    for (int i = tid; i < 16 * 8; i += BLOCKT) RQL[i] = ((i >> 3) < NK && (i & 7) < 7) ? tab->R[i >> 3][i & 7] : 0.0;
    constexpr int KST = seirls::kst<Tab>(), NPF = KST / 16;
    double* const krec = KSL + slot * KST;                              // the slot's record: field f at krec[f]
    const double* const ksl = krec + 3 + (lm < NC ? lm : NC - 1);       // this lane's component: u_start at ksl[0], k_q at ksl[NC + NC q]
    double* const f0l = F0L + slot * 8;                                 // f0[c] at f0l[c]
    // (ude_seir_ls.h) the record of the next-lower interval is prefetched into pf: field lm + 16 i on lane lm of the row
    double pf[NPF];
    int pf_s = -1, pf_want = -1;
    static_for<0, NPF>([&](auto i) { pf[i] = 0.0; });
    const double* cot = p.cot;
    size_t cot_si = 0, cot_sc = 0;
    double* const fmine = facws + (size_t)blockIdx.x * fac_doubles_per_block<Tab>();   // this block's factor workspace

    auto fetch_interval = [&](int s) {
        pf_s = s;
        const double* base = dense_rec<true>(p, s, nfld, gid);   // (record-major: ude_kernels.h)
        static_for<0, NPF>([&](auto i) {
            const int f = lm + 16 * (int)decltype(i)::value;
            pf[i] = base[f < nfld ? f : 0];
        });
    };
    auto load_interval = [&](int s) {
        if (pf_s != s) fetch_interval(s);   // (row-uniform; the first interval of a trajectory, a step upwards)
        sf = s;
        static_for<0, NPF>([&](auto i) { krec[lm + 16 * (int)decltype(i)::value] = pf[i]; });
        ts = krec[0];
        te = krec[1];
#if NL_PF_AT == 0
        if (s > 0) fetch_interval(s - 1);
#else
        pf_want = s - 1;   // issued inside the matrix phase (NL_PF_ISSUE), in front of its longest stretch without a memory wait
#endif
    };
    auto own = [&](const double (&v)[NC]) {
        double r = 0.0;
        static_for<0, NC>([&](auto c) { r = (lm == (int)decltype(c)::value) ? v[c] : r; });
        return r;
    };
    auto bcast = [&](double ownv, double (&out)[NC]) { static_for<0, NC>([&](auto c) { out[c] = rshfl(ownv, decltype(c)::value); }); };
    auto SV = [&](int i) { return p.saveat[i]; };
    auto tstop_from_cur = [&]() { return (cur >= 0 && SV(cur) > T0) ? SV(cur) : T0; };
    auto at_tstop = [&](double tt) {
        bool mod = false;
        while (cur >= 0 && SV(cur) >= tt) {
            if (SV(cur) == tt) {
                static_for<0, NC>([&](auto c) { lam[c] += cot[(size_t)cur * cot_si + (size_t)decltype(c)::value * cot_sc]; });
                mod = true;
            }
            cur -= 1;
        }
        return mod;
    };
    __syncthreads();
#if defined(LS_EXP) && LS_EXP == 9
    unsigned long long tk = __builtin_readcyclecounter(), tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ntrip = 0, ecyc[4] = {0, 0, 0, 0}, ecnt[4] = {0, 0, 0, 0};
#define LS_E0 const unsigned long long e0_ = __builtin_readcyclecounter();
#define LS_E1(i) { ecyc[i] += __builtin_readcyclecounter() - e0_; ecnt[i] += 1; }
#define LS_TICK(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tsec[i] += now_ - tk; tk = now_; }
#else
#define LS_TICK(i)
#define LS_E0
#define LS_E1(i)
#endif

    for (;;) {
        // ---- A. an idle slot takes the next trajectory of the ensemble ----
        if (ph == PH_IDLE && !exhausted) {
            for (;;) {
                int g = 0;
                if (lm == 0) g = atomicAdd(queue, 1);
                g = __shfl(g, 0, 16);
                if (g >= p.N) { exhausted = true; break; }
                if (p.retcode[g] != RET_SUCCESS) continue;   // (its forward solve failed: no gradient row, the host-cleared zeros stay)
                gid = g;
                if (p.cot_in) { cot = p.cot_in + (size_t)gid * p.ns * n; cot_si = n; cot_sc = 1; }
                else { cot = p.cot + gid; cot_si = (size_t)n * p.Npad; cot_sc = p.Npad; }
                nsteps = p.dense_n[gid];
                pf_s = -1; pf_want = -1;
                cur = p.ns - 1;
                static_for<0, NC>([&](auto c) { lam[c] = 0.0; });
                t = TF; qold = o.qoldinit; q11 = 1.0; accept = true; iter = 0; ret = RET_SUCCESS; col = 0;
                nfc = 0; nacc = 0; nrej = 0;
                load_interval(nsteps - 1);
                at_tstop(TF);   // init_cb: the jump at t = tf precedes the first step
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
        LS_TICK(0)

        // ---- B. the evaluation this slot needs now ----
        bool ev = false;
        double tev = t, zs[NC], kr[NC], znew[NC];
        int cs = 0;
        static_for<0, NC>([&](auto c) { zs[c] = lam[c]; kr[c] = 0.0; znew[c] = lam[c]; });
        const double zo = own(lam);
        if (ph == PH_INIT0) {
            ev = true;
        } else if (ph == PH_INIT1) {
            ev = true;
            const double dt0t = tdir * dt0;
            static_for<0, NC>([&](auto c) { zs[c] = __builtin_fma(dt0t, f0l[decltype(c)::value], lam[c]); });
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
                    // all S - 1 possible terms: the coefficients of stages >= s are zero in the table, fma(0, K, acc) == acc exactly
                    const double* Ar = TB + s * 16;
                    double acc = Ar[0] * K[0];
                    static_for<1, S - 1>([&](auto j) { acc = __builtin_fma(Ar[decltype(j)::value], K[j], acc); });
                    bcast(__builtin_fma(dt, acc, zo), zs);
                }
                tev = t + TB[288 + s] * dt;
                cs = __builtin_popcount(MASK & ((1u << s) - 1u));
            } else {
                ph = PH_FLUSH;   // ended with an error: results now, the (zero) gradient row in the next trip
                if (lm == 0) {
                    if (p.stats) { int64_t* st = p.stats + (size_t)gid * 8; st[4] = nfc; st[5] = nacc; st[6] = nrej; }
                    p.retcode[gid] = ret;
                }
                if (p.grad_u0 && lm < NC) p.grad_u0[(size_t)gid * n + lm] = zo;
            }
        }

        LS_TICK(1)
        // ---- C. one adjoint evaluation of all 16 slots ----
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
                double* xf = XF + (cs * NSLOTS + slot) * XFW;
                xf[lm] = xo;
                xf[NIN + lm] = dq;
            }
        }
        if (lm == 0) { RCS[slot] = cs; REV[slot] = ev ? 1 : 0; }
        if (!__syncthreads_or(ph != PH_IDLE)) break;   // (the barrier in front of the matrix products; all slots idle and the queue empty: done)
        LS_TICK(2)
        double a3v[4], dv1[4];
        {
            // first layer: 7 inputs + bias in two k-steps
            v4d z = v4d{0.0, 0.0, 0.0, 0.0};
            static_for<0, 2>([&](auto sc) { z = __builtin_amdgcn_mfma_f64_16x16x4f64(W1A[sc], XIN[(4 * decltype(sc)::value + kq) * 16 + jc], z, 0, 0, 0); });
            double a1[4], a2[4];
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                a1[r] = dtanh(z[r]);
                T_A1[(16 * w + kq + 4 * r) * TLD + jc] = a1[r];
            });
            __syncthreads();
            // a 64-term hidden product: four 16-term chains (four MFMAs each) added left to right; the A fragment of a chain is read
            // from the block's LDS copy of the weights right here (TRANSPOSED: A[i][k] = W[k][i])
            auto hidden = [&](const double* W, const double* T, bool transposed, double (&out)[4]) {
                v4d acc[4];
                static_for<0, 4>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
                    static_for<0, 4>([&](auto q) {
                        constexpr int s = 4 * b + decltype(q)::value;
                        const int row = 16 * w + jc, colk = 4 * s + kq;
                        const double a = transposed ? W[colk + row * LDW] : W[row + colk * LDW];
                        acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(4 * s + kq) * TLD + jc], acc[b], 0, 0, 0);
                    });
                });
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    out[r] = ((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r];
                });
            };
#if NL_PF_AT == 2
            NL_PF_ISSUE
#endif
            double hz[4];
            hidden(WL2, T_A1, false, hz);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                a2[r] = dtanh(hz[r] + b2r[r]);
                T_A2[(16 * w + kq + 4 * r) * TLD + jc] = a2[r];
            });
            __syncthreads();
#if NL_PF_AT == 3
            NL_PF_ISSUE
#endif
            hidden(WL3, T_A2, false, hz);
            // delta3 = (W4^T delta4) (1 - a3^2): the 7-term chain, its zero eighth term included
            v4d s3 = v4d{0.0, 0.0, 0.0, 0.0};
            static_for<0, 2>([&](auto sc) { s3 = __builtin_amdgcn_mfma_f64_16x16x4f64(W4T[sc], D4S[(4 * decltype(sc)::value + kq) * 16 + jc], s3, 0, 0, 0); });
#if NL_PF_AT == 1
            NL_PF_ISSUE
#endif
            double dv3[4], dv2[4];
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                a3v[r] = dtanh(hz[r] + b3r[r]);
                dv3[r] = s3[r] * __builtin_fma(-a3v[r], a3v[r], 1.0);
                T_D3[(16 * w + kq + 4 * r) * TLD + jc] = dv3[r];
            });
            __syncthreads();
            hidden(WL3, T_D3, true, hz);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                dv2[r] = hz[r] * __builtin_fma(-a2[r], a2[r], 1.0);
                T_D2[(16 * w + kq + 4 * r) * TLD + jc] = dv2[r];
            });
            __syncthreads();
            hidden(WL2, T_D2, true, hz);
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                dv1[r] = hz[r] * __builtin_fma(-a1[r], a1[r], 1.0);
            });
            // input cotangent: rounded products w1[i][m] delta1[i], adjacent-pair tree over the 64 rows -- levels 1, 2 across the kq lanes,
            // levels 3, 4 over the four registers, levels 5, 6 across the wavefronts (LDS)
            static_for<0, NIN>([&](auto mm) {
                double v[4];
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    double x = W1L[decltype(mm)::value * H + 16 * w + kq + 4 * r] * dv1[r];
                    x += __shfl_xor(x, 16, 64);
                    x += __shfl_xor(x, 32, 64);
                    v[r] = x;
                });
                const double s16 = (v[0] + v[1]) + (v[2] + v[3]);
                if (kq == 0) PGS[(decltype(mm)::value * NSLOTS + jc) * 4 + w] = s16;
            });
        }
        LS_TICK(3)
        // factors of this evaluation to the workspace: a3 and delta1 straight from the registers that hold them (rows kq + 4r of
        // column jc), the four tiles transposed by the wavefront that owns the slot (lane i = hidden row i)
        {
            if (REV[jc]) {
                double* dst = fmine + ((size_t)jc * NSTC + RCS[jc]) * NFAC * H + 16 * w + kq;
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    dst[2 * H + 4 * r] = a3v[r];
                    dst[3 * H + 4 * r] = dv1[r];
                });
            }
        }
        __syncthreads();   // (the four tiles and the 16-row sums are complete)
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
        if (ev) {
            double gx[NIN];
            static_for<0, NIN>([&](auto mm) {
                const double* pq = PGS + (decltype(mm)::value * NSLOTS + slot) * 4;
                gx[mm] = (pq[0] + pq[1]) + (pq[2] + pq[3]);
            });
            const double Sv = y[0], Nv = y[4], Dv = y[5];
            double dl[NC];
            dl[0] = gx[0] / Nv;
            dl[1] = __builtin_fma(sgc, zs[6], gx[1]);
            dl[2] = gx[2];
            dl[3] = gx[3];
            dl[4] = ((gx[4] - gx[0] * Sv / (Nv * Nv)) - gx[5] * Dv / (Nv * Nv)) - muc * zs[4];
            dl[5] = gx[5] / Nv;
            dl[6] = gx[6];
            static_for<0, NC>([&](auto c) { kr[c] = -dl[c]; });
            if (ph == PH_INIT0) {
                if (lm == 0) static_for<0, NC>([&](auto c) { f0l[decltype(c)::value] = kr[c]; });
                K[0] = own(kr);
                req = RQ_NORM01;
            } else if (ph == PH_INIT1) {
                req = RQ_NORM2;
            } else {
                const int s = ph;
                const double ko = own(kr);
                static_for<0, S>([&](auto j) { K[j] = ((int)decltype(j)::value == s) ? ko : K[j]; });
                if (s == S - 1) {
                    if constexpr (Tab::FSAL) static_for<0, NC>([&](auto c) { znew[c] = zs[c]; });
                    else {
                        double acc = TB[256] * K[0];
                        static_for<1, S>([&](auto j) { acc = __builtin_fma(TB[256 + decltype(j)::value], K[j], acc); });
                        bcast(__builtin_fma(dt, acc, zo), znew);
                    }
                    double acc = TB[272] * K[0];
                    static_for<1, S>([&](auto j) { acc = __builtin_fma(TB[272 + decltype(j)::value], K[j], acc); });
                    const double a0 = fabs(zo), a1 = fabs(own(znew));
                    double res[NC];
                    bcast((dt * acc) / __builtin_fma((a0 > a1 ? a0 : a1), o.reltol, o.abstol), res);
                    ssrep = 0.0;
                    static_for<0, NC>([&](auto c) { ssrep = __builtin_fma(res[c], res[c], ssrep); });
                    req = RQ_STEP;
                }
            }
        } else if (ph == PH_FLUSH) {
            req = RQ_FLUSH;
        }
        if (lm == 0) {
            REQI[slot] = req; REQZ[slot] = zero_req ? 1 : 0; RDT[slot] = dt; RG[slot] = gid; RCOL[slot] = col; ROK[slot] = ret == RET_SUCCESS ? 1 : 0;
            static_for<0, NC>([&](auto c) { ZK[slot * 16 + decltype(c)::value] = znew[c]; ZK[slot * 16 + 8 + decltype(c)::value] = kr[c]; });
        }
        zero_req = false;
        __syncthreads();
        LS_TICK(4)

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
                LS_E0
                if (mode == RQ_STEP) {
                    const double ps = slot_pass<S, MASK, 0>(fb, stg, XF, sl, l, w, TB + 256, TB + 272, dt_req, o.abstol, o.reltol, mcur, mnew, hh, ll);
                    const double tot = group_sum<64>(ps);
                    if (l == 0) sw[0] = tot;
                    LS_E1(0)
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
#if defined(LS_EXP) && LS_EXP == 9
        { const unsigned long long now_ = __builtin_readcyclecounter(); ecyc[3] += now_ - tk; }
#endif
        __syncthreads();
        LS_TICK(5)

        // ---- F. the slot's row moves its state machine on ----
        static_for<0, NC>([&](auto c) { znew[c] = ZK[slot * 16 + decltype(c)::value]; kr[c] = ZK[slot * 16 + 8 + decltype(c)::value]; });
        if (ph == PH_FLUSH) {
            if (req == RQ_FLUSH) ph = PH_IDLE;
        } else if (ph == PH_INIT0 && ev) {
            // ode_determine_initdt, first half (the slot sums first -- mu == 0: only the g0 terms --, then the replicated components)
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
                if (lm == 0) {
                    if (p.stats) { int64_t* st = p.stats + (size_t)gid * 8; st[4] = 2 + (Tab::FSAL ? 1 : 0); st[5] = 0; st[6] = 0; }
                    p.retcode[gid] = ret;
                }
                if (p.grad_u0 && lm < NC) p.grad_u0[(size_t)gid * n + lm] = zo;
            } else if (dt0 < 10.0 * REAL_EPS) {
                dt = tdir * 1e-6;
                nfc += 2;
                if constexpr (Tab::FSAL) nfc += 1;
                ph = 0;
            } else {
                ph = PH_INIT1;
            }
        } else if (ph == PH_INIT1 && ev) {
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
        } else if (ph >= 0 && ev) {
            if (ph < S - 1) {
                ph += 1;
            } else {
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
                    bool bad = false;
                    static_for<0, NC>([&](auto c) {
                        lam[c] = znew[c];
                        bad = bad || (znew[c] != znew[c]);
                    });
                    col = 1 - col;   // slot_accept: the candidate column becomes current
                    if (bad) { ret = RET_UNSTABLE; fin = true; }
                    if (t == tstop) {
                        const bool modified = at_tstop(t);
                        if (tstop == T0) fin = true;   // done
                        else {
                            tstop = tstop_from_cur();
                            if (modified && Tab::FSAL) nfc += 1;   // reset_fsal! after u_modified! (counted as upstream does)
                        }
                    }
                } else {
                    nrej += 1;
                    if (EEst != EEst) { ret = RET_UNSTABLE; fin = true; }
                }
                if (fin) {
                    ph = PH_FLUSH;
                    if (lm == 0) {
                        if (p.stats) { int64_t* st = p.stats + (size_t)gid * 8; st[4] = nfc; st[5] = nacc; st[6] = nrej; }
                        if (ret != RET_SUCCESS) p.retcode[gid] = ret;
                    }
                    if (p.grad_u0 && lm < NC) p.grad_u0[(size_t)gid * n + lm] = own(lam);
                } else {
                    ph = 0;
                }
            }
        }
        LS_TICK(6)
#if defined(LS_EXP) && LS_EXP == 9
        ntrip += 1;
#endif
    }
#if defined(LS_EXP) && LS_EXP == 9
    if (p.trace && blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 7; ++i) p.trace[i] = (double)tsec[i];
        p.trace[7] = (double)ntrip;
        for (int i = 0; i < 4; ++i) { p.trace[8 + i] = (double)ecyc[i]; p.trace[12 + i] = (double)ecnt[i]; }
    }
#endif
}

}  // namespace nodels
}  // namespace ude
