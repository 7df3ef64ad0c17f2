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

// (the round-3/4 backward kernel that stood here -- seir_ls_adj_kernel, one trip = phases A..F over [slot][row] product tiles -- was superseded by
//  ude_seir_ls2.h in round 5 and is gone from the sources since round 6; its measurements and cycle profiles: HISTORY.md 12, 12a.  What
//  remains in this header is what the second-generation and the fast-mode kernels share: layout constants, the step-end parameter
//  pass, the factor workspace.)


}  // namespace seirls
}  // namespace ude
