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

// (the round-3/4 backward kernel that stood here -- node_ls_adj_kernel, one trip = phases A..F over [slot][row] product tiles -- was superseded by
//  ude_node_ls2.h in round 5 and is gone from the sources since round 6; its measurements and cycle profiles: HISTORY.md 12, 12a.  What
//  remains in this header is what the second-generation and the fast-mode kernels share: layout constants, the step-end parameter
//  pass, the factor workspace.)


}  // namespace nodels
}  // namespace ude
