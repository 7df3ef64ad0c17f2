// ude_seir_ls.h -- the interpolating adjoint of the SEIR exposure UDE (dudt_, SEIR_exposure/seir_exposure.jl:114-147) with the
// trajectories of a block stepping in LOCK-STEP as columns of FP64 matrix-core products.
//
// adj_kernel<SeirUde<64>> gives every trajectory a wavefront of its own: the 64x64 hidden layer is 64 dependent fma's per lane
// behind 64 LDS broadcast reads, twice per evaluation, with one wavefront per SIMD and nothing to hide the latency behind
// (4.9 % of the FP64 peak).  Here a block of four wavefronts owns SIXTEEN trajectory slots:
//   * the network of an adjoint evaluation runs for all 16 slots at once on v_mfma_f64_16x16x4 (D = C + A(16x4) B(4x16),
//     d = fma(a_k, b_k, d) for k ascending: tools/probe/mfma_order_probe.hip): wavefront w owns hidden rows 16w .. 16w+15, its
//     rows of W2 and of W2^T are A-operand fragments in registers (no LDS weight copy, no broadcast reads), activations and
//     deltas of the 16 slots cross wavefronts as B operands through [row][slot] LDS tiles;
//   * ARITH-SPEC is kept bit for bit: a 64-term hidden dot is four 16-term chains (four MFMAs each, from C = 0) added left to
//     right -- the oracle's wide-dot rule --, the 3-term first layer is one MFMA whose fourth k-step adds the bias
//     (fma(b, 1, acc) == acc + b), the three input-cotangent sums are rounded products reduced by the adjacent-pair tree;
//   * everything per trajectory that is not a matrix product -- interval lookup and dense-output interpolation of the forward
//     state, the stage combinations, the error norm, the PI controller, save-point jumps -- runs on the 16-lane ROW that owns
//     the slot (slot 4w + r on row r of wavefront w), component c on lane c of the row, exactly the Driver's sequence;
//   * the parameter cotangent stays deferred: every stage leaves its factors -- a1 of every (slot, stage) in LDS (64 KB), a2 delta1
//     delta2 per hidden row in an HBM workspace, x and delta3 per slot in LDS -- and at the end of a step the four wavefronts
//     form, a quarter of the 71 parameter slots per lane each, the RK-weighted sums, the candidate mu and the slots' share of the
//     error norm with the very loops of SeirUde<64>::step_slots (mu in HBM, two columns that swap on acceptance);
//   * the blocks are PERSISTENT: every slot runs the Driver's sequence as its own state machine (initial-dt evaluations, the
//     stages of a step attempt, the end of the step), one trip of the block's loop = one adjoint evaluation of every busy slot
//     whatever stage it is at; a slot whose trajectory has ended takes the next one from a global queue.
// Per trajectory every number -- step counts, dL/du0, each of the 4481 gradient entries -- is bit-identical to the oracle and
// to adj_kernel<SeirUde<64>>.  Float64, shared time grid, interpolating adjoint with mu under error control (parity mode).
#pragma once
#include "ude_kernels.h"

namespace ude {
namespace seirls {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int H = 64, NSLOTS = 16, BLOCKT = 256, NC = 7;
constexpr int TLD = 17;   // leading dimension of the [row][slot] tiles (odd: the transposed copy to the factor workspace is conflict-free)
constexpr int PLD = 65;   // slot stride of the [slot][row] product tiles
constexpr int NSLK = H + 7;  // parameter slots per hidden row: W2[i, 0..63], W1[i, 0..2], b1[i], b2[i], W3[i], b3 (row 0)
constexpr int OFF_W1 = 0, OFF_B1 = 3 * H, OFF_W2 = 4 * H, OFF_B2 = 4 * H + H * H, OFF_W3 = OFF_B2 + H, OFF_B3 = OFF_W3 + H;

template <class Tab>
constexpr unsigned stage_mask() {
    unsigned m = 0;
    for (int s = 0; s < Tab::S; ++s)
        if (Tab::B(s) != 0.0 || Tab::BT(s) != 0.0) m |= 1u << s;
    return m;
}
constexpr int popc(unsigned m) { int c = 0; for (; m; m &= m - 1) ++c; return c; }
template <unsigned MASK>
constexpr int cslot(int s) { return popc(MASK & ((1u << s) - 1u)); }   // compacted storage slot of stage s

constexpr int NFAC = 3;     // per-stage factor rows kept in HBM: a2, delta1, delta2 (a1 of every slot and stage stays in LDS)
constexpr int TABL = 16 * 16 + 3 * 16;   // LDS copy of the tableau: A[16][16], B[16], BT[16], C[16]
// interval cache of a slot: the forward record of the interval as it is stored (t, t_end, dt, u[7], k_q[7] for q < NK), rounded up to
// whole 16-lane rows (the slot's sixteen lanes fetch it round-robin)
template <class Tab>
constexpr int kst() { return (3 + NC + Tab::NK * NC + 15) / 16 * 16; }
template <class Tab>
constexpr int lds_doubles() {
    constexpr int NSTC = popc(stage_mask<Tab>());
    return 4 * H * TLD + 4 * 16 + 16 + 3 * NSLOTS * PLD + NSTC * NSLOTS * 4 + NSLOTS * 16 + NSLOTS * 8 + TABL + 6 * NSLOTS + NSLOTS * 4 * 2 +
           NSLOTS * NSTC * H + NSLOTS * kst<Tab>() + NSLOTS * 8 + 16 * 8 + NSLOTS * 16 + 3 * H;
}
// doubles of factor workspace per block
template <class Tab>
constexpr size_t fac_doubles_per_block() { return (size_t)NSLOTS * popc(stage_mask<Tab>()) * NFAC * H; }

__device__ __forceinline__ double rshfl(double x, int c) { return __shfl(x, c, 16); }   // lane c of this 16-lane row
__device__ __forceinline__ double row_tree4(double v0, double v1, double v2, double v3) {
    double x = (v0 + v1) + (v2 + v3);
    x += dpp_mov<DPP_QUAD(1, 0, 3, 2)>(x);
    x += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(x);
    x += dpp_mov<DPP_ROW_HALF_MIRROR>(x);
    x += dpp_mov<DPP_ROW_MIRROR>(x);
    return x;
}

// the 71 parameter slots of ONE trajectory, a quarter per wavefront (lane i = hidden row i; wavefront w: W2 columns 16w .. 16w+15
// and the extra slots 2w, 2w+1): the loops of SeirUde<64>::step_slots / init_norm01 / init_norm2.  Every request of a trip is
// worked on by all four wavefronts, so the trip waits for a quarter pass per request instead of a whole pass of the busiest wavefront.
//   MODE 0: end of a step -- candidate mu_new, returns this lane's sum of squared residuals
//   MODE 1 / 2: the initial-dt norms (h, l) += (g0 / sk)^2  /  ((g1 - g0) / sk)^2 in real-real arithmetic (mu == 0 there)
constexpr int QW = H / 4;   // W2 columns per wavefront
#ifndef LS_PF_AT
#define LS_PF_AT 2   // where the prefetch of the next-lower forward interval is issued: 0 at the switch itself, 1 / 2 inside the matrix phase
#endif
#define LS_PF_ISSUE if (pf_want >= 0) { fetch_interval(pf_want); pf_want = -1; }
#ifndef LS_CUT
#define LS_CUT 0   // timing experiments only (results wrong): 1 no mu loads, 2 no mu stores, 4 no division, 8 no factor loads
#endif
template <int NST, unsigned MASK, int MODE>
__device__ __forceinline__ double slot_pass(const double* __restrict__ fbase /* this trajectory's factors: [cs][a2 | delta1 | delta2][64] */,
                                            const double* a1s /* LDS [cs][64]: a1 of this slot's stages */, const double* xf /* LDS [cs][16][4] */,
                                            int slot, int lane, int w, const double* Bw, const double* BTw, double dt, double abstol, double reltol,
                                            const double (&mq)[2 * 8 + 2] /* MODE 0: this lane's 18 mu words, loaded by the caller (mu_load) */, double* __restrict__ mu_new,
                                            double& hh, double& ll) {
    constexpr int CH = 8;
    // every load of the pass is issued before the first use: delta2 (needed first), mu, then the rows only the extra slots use
    double a2[NST], d1[NST], d2[NST];
    static_for<0, NST>([&](auto s) {
        if constexpr ((MASK >> decltype(s)::value) & 1u) {
            constexpr int cs = cslot<MASK>(decltype(s)::value);
            d2[s] = (LS_CUT & 8) ? dt + 3.0 : fbase[(size_t)(cs * NFAC + 2) * H + lane];
        }
    });
    double mcur[CH], mnext[CH], mex[2];
    static_for<0, CH>([&](auto i) {
        mcur[i] = MODE == 0 ? mq[i] : 0.0;
        mnext[i] = MODE == 0 ? mq[CH + decltype(i)::value] : 0.0;
    });
    static_for<0, 2>([&](auto i) { mex[i] = MODE == 0 ? mq[2 * CH + decltype(i)::value] : 0.0; });
    static_for<0, NST>([&](auto s) {
        if constexpr ((MASK >> decltype(s)::value) & 1u) {
            constexpr int cs = cslot<MASK>(decltype(s)::value);
            a2[s] = (LS_CUT & 8) ? dt + 1.0 : fbase[(size_t)(cs * NFAC) * H + lane];
            d1[s] = (LS_CUT & 8) ? dt + 2.0 : fbase[(size_t)(cs * NFAC + 1) * H + lane];
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
            if (!(LS_CUT & 2)) mu_new[(size_t)sl * H] = m1;
            const double a0 = fabs(m0), a1 = fabs(m1);
            const double res = (LS_CUT & 4) ? (dt * ae) * __builtin_fma((a0 > a1 ? a0 : a1), reltol, abstol) : (dt * ae) / __builtin_fma((a0 > a1 ? a0 : a1), reltol, abstol);
            ps = __builtin_fma(res, res, ps);
        } else {
            const double sk = __builtin_fma(fabs(m0), reltol, abstol);
            const double q = MODE == 1 ? g[0] / sk : (g[NST - 1] - g[0]) / sk;
            dd_acc(hh, ll, q * q);
        }
    };
    // (a rolled loop on purpose: with the two chunks written out the register allocator spills 350 registers)
#pragma unroll 1
    for (int k0 = QW * w; k0 < QW * w + QW; k0 += CH) {
        static_for<0, CH>([&](auto i) {
            const int k = k0 + decltype(i)::value;
            double g[NST];
            static_for<0, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) g[s] = -(d2[s] * a1s[cslot<MASK>(decltype(s)::value) * H + k]);
            });
            body(k, g, mcur[i]);
        });
        static_for<0, CH>([&](auto i) { mcur[i] = mnext[i]; });
    }
    static_for<0, 7>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        if ((e >> 1) == w) {
            double g[NST];
            static_for<0, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) {
                    const double* xs = xf + (cslot<MASK>(decltype(s)::value) * NSLOTS + slot) * 4;   // x0 x1 x2 delta3 of this stage
                    double v;
                    if constexpr (e < 3) v = -(d1[s] * xs[e]);
                    else if constexpr (e == 3) v = -d1[s];
                    else if constexpr (e == 4) v = -d2[s];
                    else if constexpr (e == 5) v = -(xs[3] * a2[s]);
                    else v = lane == 0 ? -xs[3] : -0.0;
                    g[s] = v;
                }
            });
            body(H + e, g, mex[e & 1]);
        }
    });
    return ps;
}

enum { PH_IDLE = -4, PH_FLUSH = -3, PH_INIT0 = -2, PH_INIT1 = -1 };   // >= 0: stage s of a step attempt
enum { RQ_NONE = -1, RQ_STEP = 0, RQ_NORM01 = 1, RQ_NORM2 = 2, RQ_FLUSH = 4 };

// PERSISTENT blocks: every slot runs the Driver's sequence as its own little state machine (initial-dt evaluations, the stages
// of a step attempt, the end of the step); one trip of the block's loop = ONE adjoint evaluation of every busy slot, whatever
// stage each of them is at, followed by the parameter-slot work the slots asked for.  A slot whose trajectory has ended writes
// its gradient row and takes the next trajectory of the ensemble from a global queue: no second, half-empty round of blocks,
// no slot idling until the slowest trajectory of its block is through.
template <class Tab>
__global__ void __launch_bounds__(BLOCKT, 1) seir_ls_adj_kernel(const KParams p, double* __restrict__ facws, int* __restrict__ queue) {
    constexpr int S = Tab::S, NK = Tab::NK;
    constexpr unsigned MASK = stage_mask<Tab>();
    constexpr int NSTC = popc(MASK);
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* T_A1 = sm;
    double* T_D2 = T_A1 + H * TLD;
    double* T_A2 = T_D2 + H * TLD;
    double* T_D1 = T_A2 + H * TLD;
    double* XIN = T_D1 + H * TLD;             // [4][16]: x0 x1 x2 1
    double* D3S = XIN + 4 * 16;               // [16]
    double* PG = D3S + 16;                    // [3][16][PLD]
    double* XF = PG + 3 * NSLOTS * PLD;       // [NSTC][16][4]
    double* BQ = XF + NSTC * NSLOTS * 4;      // [16][16]
    double* YS = BQ + NSLOTS * 16;            // [16][8]
    double* TB = YS + NSLOTS * 8;             // tableau: A[16][16], B, BT, C
    double* RDT = TB + TABL;                  // [16] step size of a step request
    long long* RG = reinterpret_cast<long long*>(RDT + NSLOTS);    // [16] trajectory of the slot
    int* REQI = reinterpret_cast<int*>(RG + NSLOTS);               // [16] request, [16] zero-mu flag, [16] current mu column, [16] success, [16] cs of this trip
    int* REQZ = REQI + NSLOTS;
    int* RCOL = REQZ + NSLOTS;
    int* ROK = RCOL + NSLOTS;
    int* RCS = ROK + NSLOTS;
    int* REV = RCS + NSLOTS;
    double* SUMW = RDT + 6 * NSLOTS;          // [16][4][2] per slot and wavefront: ps | (h, l)
    double* A1P = SUMW + NSLOTS * 4 * 2;      // [16 slots][NSTC][64]: a1 of every stage of the slot's current step (read by broadcast in E)
    double* KSL = A1P + NSLOTS * NSTC * H;         // [16 slots][KST]: interval cache, the stored record of the slot's current forward interval
    double* F0L = KSL + NSLOTS * kst<Tab>();       // [16 slots][8]: f0 of the initial-dt phase
    double* RQL = F0L + NSLOTS * 8;                // [16 lanes q][8]: Horner tables of b_q(theta)
    double* ZK = RQL + 16 * 8;                // [16 slots][16]: znew[7] | kr[7] parked across the parameter-slot work of a trip
    double* W1L = ZK + NSLOTS * 16;           // [3][64]: W1[i][m] at W1L[m * H + i] (read where the input-cotangent products are formed)

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int kq = l >> 4, jc = l & 15;       // matrix view: k index / column (= slot) of this lane
    const int rr = l >> 4, lm = l & 15;       // scalar view: row of the wavefront (slot 4w + rr), lane inside the row
    const int slot = 4 * w + rr;
    const double* __restrict__ th = p.theta;
    const TabDev* __restrict__ tab = p.tab;
    const int n = NC;
    const int nfld = 3 + n + NK * n;

    // ---- weights: A-operand fragments and per-row constants of this wavefront's 16 hidden rows ----
    double W2A[16], W2T[16];
    {
        const int row = 16 * w + jc;
        static_for<0, 16>([&](auto sc) {
            const int col = 4 * decltype(sc)::value + kq;
            W2A[sc] = th[OFF_W2 + row + col * H];      // A[i][k] = W2[16w + i][4s + k]
            W2T[sc] = th[OFF_W2 + col + row * H];      // A[i][k] = W2[4s + k][16w + i]
        });
    }
    const double W1A = kq < 3 ? th[OFF_W1 + (16 * w + jc) + kq * H] : th[OFF_B1 + 16 * w + jc];
    for (int i = tid; i < 3 * H; i += BLOCKT) W1L[i] = th[OFF_W1 + i];
    double b2r[4], w3r[4];
    static_for<0, 4>([&](auto r) {
        const int row = 16 * w + kq + 4 * decltype(r)::value;
        b2r[r] = th[OFF_B2 + row];
        w3r[r] = th[OFF_W3 + row];
    });
    const double Fc = p.mc.consts[0], b0c = p.mc.consts[1], muc = p.mc.consts[4], sgc = p.mc.consts[5], gac = p.mc.consts[6],
                 dc = p.mc.consts[7], lac = p.mc.consts[8];
    if (tid < 16) XIN[3 * 16 + tid] = 1.0;
    for (int i = tid; i < 16 * 16; i += BLOCKT) TB[i] = tab->A[i >> 4][i & 15];
    if (tid < 16) { TB[256 + tid] = tab->B[tid]; TB[272 + tid] = tab->BT[tid]; TB[288 + tid] = tab->C[tid]; }
    for (int i = tid; i < NSLOTS * 8; i += BLOCKT) { YS[i] = 0.0; SUMW[i] = 0.0; }

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
    constexpr int KST = kst<Tab>(), NPF = KST / 16;
    double* const krec = KSL + slot * KST;                              // the slot's record: field f at krec[f]
    const double* const ksl = krec + 3 + (lm < NC ? lm : NC - 1);       // this lane's component: u_start at ksl[0], k_q at ksl[NC + NC q]
    double* const f0l = F0L + slot * 8;                                 // f0[c] at f0l[c]
    // the backward solve walks the stored intervals downwards: while interval s is in use the record of s - 1 is already on its way
    // from HBM into pf (field lm + 16 i on lane lm of the row), so the switch to s - 1 is an LDS write of data that has long arrived
    // instead of a dependent HBM round trip in front of the block's barrier
    double pf[NPF];
    int pf_s = -1, pf_want = -1;
    static_for<0, NPF>([&](auto i) { pf[i] = 0.0; });
    // this lane's 18 words of a trajectory's current mu column (W2 columns 16w .. 16w + 15 of hidden row `l`, extra slots 2w, 2w + 1)
    double mq[2 * 8 + 2];
    static_for<0, 18>([&](auto i) { mq[i] = 0.0; });
    auto mu_load = [&](const double* mc) {
        static_for<0, 16>([&](auto i) { mq[i] = (LS_CUT & 1) ? 0.0 : mc[(size_t)(QW * w + (int)decltype(i)::value) * H]; });
        static_for<0, 2>([&](auto i) { mq[16 + decltype(i)::value] = (!(LS_CUT & 1) && 2 * w + (int)decltype(i)::value < 7) ? mc[(size_t)(H + 2 * w + (int)decltype(i)::value) * H] : 0.0; });
    };
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
#if LS_PF_AT == 0
        if (s > 0) fetch_interval(s - 1);
#else
        pf_want = s - 1;   // issued inside the matrix phase (LS_PF_ISSUE): nothing that follows there waits on a younger memory operation
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
#if !(LS_CUT & 16)   // (16: timing experiment -- never leave the first interval)
            while (tev < ts && sf > 0) load_interval(sf - 1);
            while (tev >= te && sf < nsteps - 1) load_interval(sf + 1);
#endif
            const double dtf = te - ts;
#if LS_CUT & 16
            const double thv = fmin(fmax((tev - ts) / dtf, 0.0), 1.0);
#else
            const double thv = (tev - ts) / dtf;
#endif
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
                double* xf = XF + (cs * NSLOTS + slot) * 4;
                xf[0] = x0; xf[1] = x1; xf[2] = x2; xf[3] = d3;
            }
        }
        if (!__syncthreads_or(ph != PH_IDLE)) break;   // (the barrier in front of the matrix products; all slots idle and the queue empty: done)
        LS_TICK(2)
#if LS_PF_AT == 1
        LS_PF_ISSUE
#endif
        {
            // layer 1 (3 inputs + bias in one k-step)
            v4d z = __builtin_amdgcn_mfma_f64_16x16x4f64(W1A, XIN[kq * 16 + jc], v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            double a1[4], a2[4], dv1[4], dv2[4];
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                a1[r] = dtanh(z[r]);
                T_A1[(16 * w + kq + 4 * r) * TLD + jc] = a1[r];
            });
            __syncthreads();
#if LS_PF_AT == 2
            LS_PF_ISSUE
#endif
            // hidden layer: four 16-term chains (four MFMAs each) added left to right
            {
                v4d acc[4];
                static_for<0, 4>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
                    static_for<0, 4>([&](auto q) {
                        constexpr int s = 4 * b + decltype(q)::value;
                        acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2A[s], T_A1[(4 * s + kq) * TLD + jc], acc[b], 0, 0, 0);
                    });
                });
                const double d3j = D3S[jc];
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const double z2 = (((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r]) + b2r[r];
                    a2[r] = dtanh(z2);
                    dv2[r] = __builtin_fma(w3r[r], d3j, 0.0) * __builtin_fma(-a2[r], a2[r], 1.0);
                    const int row = 16 * w + kq + 4 * r;
                    T_D2[row * TLD + jc] = dv2[r];
                    T_A2[row * TLD + jc] = a2[r];
                });
            }
            __syncthreads();
            // transposed hidden layer on the deltas
            {
                v4d acc[4];
                static_for<0, 4>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    acc[b] = v4d{0.0, 0.0, 0.0, 0.0};
                    static_for<0, 4>([&](auto q) {
                        constexpr int s = 4 * b + decltype(q)::value;
                        acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2T[s], T_D2[(4 * s + kq) * TLD + jc], acc[b], 0, 0, 0);
                    });
                });
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const double s1 = ((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r];
                    dv1[r] = s1 * __builtin_fma(-a1[r], a1[r], 1.0);
                    const int row = 16 * w + kq + 4 * r;
                    T_D1[row * TLD + jc] = dv1[r];
                    static_for<0, 3>([&](auto mm) { PG[(decltype(mm)::value * NSLOTS + jc) * PLD + row] = W1L[decltype(mm)::value * H + row] * dv1[r]; });
                });
            }
        }
        __syncthreads();
        LS_TICK(3)
        // factors of this evaluation to the workspace, slot-major: wavefront w copies its own four slots (lane i = hidden row i)
        // (whether and where: from the registers of the slot's row, all four slots' tile reads issued together)
        {
            const int evi = ev ? 1 : 0;
            double va1[4], va2[4], vd1[4], vd2[4];
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int sl = 4 * w + q;
                va1[q] = T_A1[l * TLD + sl]; va2[q] = T_A2[l * TLD + sl]; vd1[q] = T_D1[l * TLD + sl]; vd2[q] = T_D2[l * TLD + sl];
            });
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int sl = 4 * w + q;
                if (__builtin_amdgcn_readlane(evi, 16 * q)) {
                    const int cs_ = __builtin_amdgcn_readlane(cs, 16 * q);
                    double* dst = fmine + ((size_t)sl * NSTC + cs_) * NFAC * H + l;
                    A1P[(sl * NSTC + cs_) * H + l] = va1[q];
                    dst[0] = va2[q];
                    dst[H] = vd1[q];
                    dst[2 * H] = vd2[q];
                }
            });
        }
        // ---- D. the slot's row: state cotangent of this evaluation, and what it asks of the parameter-slot pass ----
        int req = RQ_NONE;
        if (ev) {
            double gx[3];
            static_for<0, 3>([&](auto mm) {
                const double* pr = PG + (decltype(mm)::value * NSLOTS + slot) * PLD + 4 * lm;
                gx[mm] = row_tree4(pr[0], pr[1], pr[2], pr[3]);
            });
            const double Sv = y[0], Nv = y[4], Dv = y[5];
            const double cc = b0c * Fc / Nv;
            const double cN = b0c * Sv * Fc / (Nv * Nv);
            double dl[NC];
            dl[0] = (-cc - muc) * zs[0] + cc * zs[1] + gx[0] / Nv;
            dl[1] = -(sgc + muc) * zs[1] + sgc * zs[2] + sgc * zs[6];
            dl[2] = -(gac + muc) * zs[2] + gac * zs[3] + dc * gac * zs[5] + gx[1];
            dl[3] = -muc * zs[3];
            dl[4] = cN * zs[0] - cN * zs[1] - muc * zs[4] - gx[0] * Sv / (Nv * Nv) - gx[2] * Dv / (Nv * Nv);
            dl[5] = -lac * zs[5] + gx[2] / Nv;
            dl[6] = 0.0;
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
                    // perform_step! is complete: new state and the replicated part of the error norm
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

        // ---- E. the parameter-slot work the slots asked for: every request is worked on by all four wavefronts, a quarter of the
        // slots each (no cross-wavefront dependence: a wavefront only ever touches its own columns of mu) ----
        {
            // (the 16 request records in one go: lane q holds slot q's, the loop below walks the slots that asked)
            const int q16 = l & 15;
            const int r_mode = REQI[q16], r_zr = REQZ[q16], r_col = RCOL[q16], r_ok = ROK[q16];
            const long long r_g = RG[q16];
            const double r_dt = RDT[q16];
            unsigned pend = (unsigned)__ballot(l < 16 && (r_mode != RQ_NONE || r_zr != 0));
            auto rl32 = [&](int v, int src) { return __builtin_amdgcn_readlane(v, src); };
#pragma unroll 1
            while (pend != 0u) {
                const int sl = __builtin_ctz(pend);
                pend &= pend - 1u;
                const int mode = rl32(r_mode, sl);
                const int zr = rl32(r_zr, sl);
                const long long g = (long long)(((unsigned long long)(unsigned)rl32((int)((unsigned long long)r_g >> 32), sl) << 32) |
                                                (unsigned)rl32((int)(unsigned long long)r_g, sl));
                const int cl = rl32(r_col, sl);
                const double dt_req = __longlong_as_double((long long)(((unsigned long long)(unsigned)rl32((int)((unsigned long long)__double_as_longlong(r_dt) >> 32), sl) << 32) |
                                                                       (unsigned)rl32((int)(unsigned long long)__double_as_longlong(r_dt), sl)));
                double* mbase = p.slot_glob + (size_t)g * (2 * NSLK * H) + l;
                double* mcur = mbase + (size_t)cl * (NSLK * H);
                double* mnew = mbase + (size_t)(1 - cl) * (NSLK * H);
                if (zr) {   // a fresh trajectory: its current mu column starts at zero
#pragma unroll
                    for (int k = 0; k < QW; ++k) mcur[(size_t)(QW * w + k) * H] = 0.0;
                    if (2 * w < 7) mcur[(size_t)(H + 2 * w) * H] = 0.0;
                    if (2 * w + 1 < 7) mcur[(size_t)(H + 2 * w + 1) * H] = 0.0;
                }
                const double* fb = fmine + (size_t)sl * NSTC * NFAC * H;
                double hh = 0.0, ll = 0.0;
                const double* a1s = A1P + sl * NSTC * H;
                double* sw = SUMW + (sl * 4 + w) * 2;
                LS_E0
                if (mode == RQ_STEP) {
                    mu_load(mcur);
                    const double ps = slot_pass<S, MASK, 0>(fb, a1s, XF, sl, l, w, TB + 256, TB + 272, dt_req, o.abstol, o.reltol, mq, mnew, hh, ll);
                    const double tot = group_sum<64>(ps);
                    if (l == 0) sw[0] = tot;
                    LS_E1(0)
                } else if (mode == RQ_NORM01) {
                    slot_pass<1, 1u, 1>(fb, a1s, XF, sl, l, w, TB + 256, TB + 272, 0.0, o.abstol, o.reltol, mq, mnew, hh, ll);
                    group_dd_sum<64>(hh, ll);
                    if (l == 0) { sw[0] = hh; sw[1] = ll; }
                } else if (mode == RQ_NORM2) {
                    slot_pass<2, 3u, 2>(fb, a1s, XF, sl, l, w, TB + 256, TB + 272, 0.0, o.abstol, o.reltol, mq, mnew, hh, ll);
                    group_dd_sum<64>(hh, ll);
                    if (l == 0) { sw[0] = hh; sw[1] = ll; }
                } else if (mode == RQ_FLUSH) {   // the trajectory's gradient row (zeros if it failed)
                    const bool ok = rl32(r_ok, sl) != 0;
                    double* row = p.grad_part + (size_t)g * p.n_param;
#pragma unroll 4
                    for (int k = QW * w; k < QW * w + QW; ++k) row[OFF_W2 + l + k * H] = ok ? mcur[(size_t)k * H] : 0.0;
                    static_for<0, 7>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        if ((e >> 1) == w) {
                            const int idx = e < 3 ? OFF_W1 + l + e * H : e == 3 ? OFF_B1 + l : e == 4 ? OFF_B2 + l : e == 5 ? OFF_W3 + l : (l == 0 ? OFF_B3 : -1);
                            if (idx >= 0) row[idx] = ok ? mcur[(size_t)(H + e) * H] : 0.0;
                        }
                    });
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

}  // namespace seirls
}  // namespace ude
