// ude_seir_ls2.h -- second generation of the lock-step PARITY-mode adjoint of the SEIR exposure UDE (ude_seir_ls.h: read its header
// first; same arithmetic, same results bit for bit, same host interface).  What changed (round 5) is everything around the step-end
// parameter pass, taken over from the fast-mode kernel ude_seir_lsf.h, whose trip costs 15.4 k cycles where phases A-D + F of
// ude_seir_ls.h cost 18.6 k:
//   * the hidden units are PERMUTED over the registers of the matrix view (register r of lane (kq, .) = unit 16w + 4kq + r): the three
//     input-cotangent sums need two in-lane levels, two lane exchanges and 192 words of LDS instead of three [slot][row] product
//     tiles (25 KB, 12 LDS writes + 12 reads + 4 DPP levels per lane);
//   * a slot's row keeps only ITS component of lambda in a register; the stage derivatives K live in LDS (10 KB), the replicated
//     copies (zs, lambda, znew) are broadcast where they are used: ~100 registers per lane less, no scratch;
//   * phases D and F are ONE row phase where no parameter pass lies between them (every evaluation that is not the last stage of
//     an attempt and not an initial-dt evaluation): the state machine moves on in front of the E barrier.
// The step-end pass itself (E: slot_pass of ude_seir_ls.h, mu in HBM, a1 in LDS, a2 / delta1 / delta2 through the factor workspace)
// is unchanged.
#pragma once
#include "ude_seir_ls.h"

namespace ude {
namespace seirls2 {

using namespace seirls;

#ifndef UDE_LS2_MU_PREFETCH
#define UDE_LS2_MU_PREFETCH 0   // 1: the mu words of the trip's first step-end request are fetched in front of the row phase D instead of at the
#endif                          // start of the pass -- measured (round 5): 10.69 against 10.53 ms, 36 B of scratch appear: not kept


template <class Tab>
constexpr int lds_doubles() {
    constexpr int NSTC = popc(stage_mask<Tab>());
    return 4 * H * TLD + 4 * 16 + 16 + 3 * NSLOTS * 4 + NSTC * NSLOTS * 4 + NSLOTS * 16 + NSLOTS * 8 + TABL + 6 * NSLOTS + NSLOTS * 4 * 2 +
           NSLOTS * NSTC * H + NSLOTS * kst<Tab>() + NSLOTS * 8 + 16 * 8 + NSLOTS * 16 + 3 * H + NSLOTS * Tab::S * 8;
}

// GEN = true: the RUNTIME-SHAPE instance -- any exposure-UDE chain 3 -> H1 -> H2 -> 1 (tanh, tanh, identity) with H1, H2 <= 64 that udecore.hip's seir_gen_ls_shape admits (no 32- / 64-term product with < 16 results)
// (`Lux.Chain` / `FastChain` accept any chain upstream; ude_model_generic.h serves every other shape one wavefront per trajectory).  The
// weights are zero-padded to 64 x 64 -- a padded unit has activation tanh(0) = 0 and delta 0, a padded parameter slot gradient and
// residual 0: all exact -- and every product keeps the ORACLE'S association for ITS length (wide_dot): a 64-term product is four
// 16-term chains added left to right, a shorter one is ONE ascending chain (sixteen k-steps into one accumulator; fma(0, x, acc) == acc
// on the padding); the input cotangent is the adjacent-pair tree for H1 == 64 and an ascending chain over the units otherwise.
template <class Tab, bool GEN = false>
__global__ void __launch_bounds__(BLOCKT, 1) seir_ls2_adj_kernel(const KParams p, double* __restrict__ facws, int* __restrict__ queue) {
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
    double* GXP = D3S + 16;                   // [3][16][4]: per wavefront partial sums of the input cotangent
    double* XF = GXP + 3 * NSLOTS * 4;        // [NSTC][16][4]
    double* BQ = XF + NSTC * NSLOTS * 4;      // [16][16]
    double* YS = BQ + NSLOTS * 16;            // [16][8]
    double* TB = YS + NSLOTS * 8;             // tableau: A[16][16], B, BT, C
    double* RDT = TB + TABL;                  // [16] step size of a step request
    long long* RG = reinterpret_cast<long long*>(RDT + NSLOTS);
    int* REQI = reinterpret_cast<int*>(RG + NSLOTS);
    int* REQZ = REQI + NSLOTS;
    int* RCOL = REQZ + NSLOTS;
    int* ROK = RCOL + NSLOTS;
    int* RCS = ROK + NSLOTS;
    int* REV = RCS + NSLOTS;
    double* SUMW = RDT + 6 * NSLOTS;          // [16][4][2] per slot and wavefront: ps | (h, l)
    double* A1P = SUMW + NSLOTS * 4 * 2;      // [16 slots][NSTC][64]: a1 of every stage of the slot's current step
    double* KSL = A1P + NSLOTS * NSTC * H;    // [16 slots][KST]: interval cache
    double* F0L = KSL + NSLOTS * kst<Tab>();  // [16 slots][8]: f0 of the initial-dt phase
    double* RQL = F0L + NSLOTS * 8;           // [16 lanes q][8]: Horner tables of b_q(theta)
    double* ZK = RQL + 16 * 8;                // [16 slots][16]: (unused half | ssrep, new own components are kept in registers)
    double* W1L = ZK + NSLOTS * 16;           // [3][64]
    double* KL = W1L + 3 * H;                 // [16 slots][S][8]: stage derivatives of lambda
    (void)REV; (void)ZK;

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int kq = l >> 4, jc = l & 15;       // matrix view
    const int rr = l >> 4, lm = l & 15;       // scalar view
    const int slot = 4 * w + rr;
    const double* __restrict__ th = p.theta;
    const TabDev* __restrict__ tab = p.tab;
    const int n = NC;
    const int nfld = 3 + n + NK * n;

    // ---- weights: A-operand fragments with PERMUTED rows (ude_seir_lsf.h) ----
    // layer widths and the offsets of theta = [W1 (H1 x 3) | b1 | W2 (H2 x H1) | b2 | W3 (1 x H2) | b3] (column-major, as Lux / FastChain / Flux lay it out)
    const int H1 = GEN ? p.mc.dims[1] : H, H2 = GEN ? p.mc.dims[2] : H;
    const int oW1 = 0, oB1 = 3 * H1, oW2 = oB1 + H1, oB2 = oW2 + H1 * H2, oW3 = oB2 + H2, oB3 = oW3 + H2;
    const bool blk_fwd = H1 == H, blk_bwd = H2 == H;   // a 64-term product: four 16-term chains; a shorter one: ONE ascending chain
    double W2A[16], W2T[16];
    const int urow = 16 * w + 4 * (jc & 3) + (jc >> 2);
    static_for<0, 16>([&](auto sc) {
        const int col = 4 * decltype(sc)::value + kq;
        W2A[sc] = (urow < H2 && col < H1) ? th[oW2 + urow + col * H2] : 0.0;      // A[i][k] = W2[unit2(i)][unit1 = 4s + k]
        W2T[sc] = (col < H2 && urow < H1) ? th[oW2 + col + urow * H2] : 0.0;      // A[i][k] = W2[unit2 = 4s + k][unit1(i)]
    });
    const double W1A = urow < H1 ? (kq < 3 ? th[oW1 + urow + kq * H1] : th[oB1 + urow]) : 0.0;
    for (int i = tid; i < 3 * H; i += BLOCKT) W1L[i] = (i % H) < H1 ? th[oW1 + (i % H) + (i / H) * H1] : 0.0;
    const int u0r = 16 * w + 4 * kq;
    double b2r[4], w3r[4];
    static_for<0, 4>([&](auto r) {
        const int un = u0r + decltype(r)::value;
        b2r[r] = un < H2 ? th[oB2 + un] : 0.0;
        w3r[r] = un < H2 ? th[oW3 + un] : 0.0;
    });
    const double Fc = p.mc.consts[0], b0c = p.mc.consts[1], muc = p.mc.consts[4], sgc = p.mc.consts[5], gac = p.mc.consts[6],
                 dc = p.mc.consts[7], lac = p.mc.consts[8];
    if (tid < 16) XIN[3 * 16 + tid] = 1.0;
    for (int i = tid; i < 16 * 16; i += BLOCKT) TB[i] = tab->A[i >> 4][i & 15];
    if (tid < 16) { TB[256 + tid] = tab->B[tid]; TB[272 + tid] = tab->BT[tid]; TB[288 + tid] = tab->C[tid]; }
    for (int i = tid; i < NSLOTS * 8; i += BLOCKT) { YS[i] = 0.0; SUMW[i] = 0.0; F0L[i] = 0.0; }
    for (int i = tid; i < NSLOTS * S * 8; i += BLOCKT) KL[i] = 0.0;
    for (int i = tid; i < 16 * 8; i += BLOCKT) RQL[i] = ((i >> 3) < NK && (i & 7) < 7) ? tab->R[i >> 3][i & 7] : 0.0;

    // ---- per-slot state on the slot's row: component c on lane c ----
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
    double zo = 0.0, zn = 0.0;                // this lane's component of lambda / of the candidate
    constexpr int KST = kst<Tab>(), NPF = KST / 16;
    double* const krec = KSL + slot * KST;
    const double* const ksl = krec + 3 + (lm < NC ? lm : NC - 1);
    double* const kl = KL + (size_t)slot * S * 8 + (lm < NC ? lm : 7);
    double* const f0l = F0L + slot * 8;
    double pf[NPF];
    int pf_s = -1, pf_want = -1;
    static_for<0, NPF>([&](auto i) { pf[i] = 0.0; });
    double mq[2 * 8 + 2];
    static_for<0, 18>([&](auto i) { mq[i] = 0.0; });
    auto mu_load = [&](const double* mc) {
        static_for<0, 16>([&](auto i) { mq[i] = mc[(size_t)(QW * w + (int)decltype(i)::value) * H]; });
        static_for<0, 2>([&](auto i) { mq[16 + decltype(i)::value] = (2 * w + (int)decltype(i)::value < 7) ? mc[(size_t)(H + 2 * w + (int)decltype(i)::value) * H] : 0.0; });
    };
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
    __syncthreads();
#if defined(UDE_LS2_CLOCKS)   // timing experiment: cycles of wavefront 0 of block 0 per section of a trip (tools/ls2_prof.py)
    unsigned long long tk = __builtin_readcyclecounter(), tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ntrip = 0;
#define LS2_TICK(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tsec[i] += now_ - tk; tk = now_; }
#else
#define LS2_TICK(i)
#endif

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
                ph = PH_FLUSH;   // ended with an error: results now, the (zero) gradient row in the next trip
                if (lm == 0 && ret != RET_SUCCESS) p.retcode[gid] = ret;
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
            const double x0 = y[0] / y[4], x1 = y[2], x2 = y[5] / y[4];
            const double d3 = (zs[1] - zs[0]) * 1.0;
            if (lm == 0) {
                XIN[0 * 16 + slot] = x0; XIN[1 * 16 + slot] = x1; XIN[2 * 16 + slot] = x2;
                D3S[slot] = d3;
                double* xf = XF + (cs * NSLOTS + slot) * 4;
                xf[0] = x0; xf[1] = x1; xf[2] = x2; xf[3] = d3;
            }
        }
        LS2_TICK(0)
#if UDE_LS2_MU_PREFETCH
        // which slots END A STEP with this trip is known now: the first of them has its current mu column fetched behind the matrix
        // phase, in front of the row phase D (2.4 k cycles that the round trip hides behind), instead of at the start of its pass
        if (lm == 0) { RCS[slot] = (ev && ph == S - 1 && !zero_req) ? 1 : 0; RG[slot] = gid; RCOL[slot] = col; }
#endif
        if (!__syncthreads_or(ph != PH_IDLE)) break;
        LS2_TICK(1)
        {
            v4d z = __builtin_amdgcn_mfma_f64_16x16x4f64(W1A, XIN[kq * 16 + jc], v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            double a1[4], dv1[4];
            static_for<0, 4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                a1[r] = dtanh(z[r]);
                T_A1[(u0r + r) * TLD + jc] = a1[r];
            });
            __syncthreads();
            if (pf_want >= 0) { fetch_interval(pf_want); pf_want = -1; }
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
                } else {   // fewer than 64 inputs: one ascending chain (the padded k-steps add fma(0, a, acc) == acc)
                    acc[0] = v4d{0.0, 0.0, 0.0, 0.0};
                    static_for<0, 16>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(W2A[s], T_A1[(4 * s + kq) * TLD + jc], acc[0], 0, 0, 0);
                    });
                }
                const double d3j = D3S[jc];
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const double z2 = ((!GEN || blk_fwd) ? (((acc[0][r] + acc[1][r]) + acc[2][r]) + acc[3][r]) : acc[0][r]) + b2r[r];
                    const double a2 = dtanh(z2);
                    T_D2[(u0r + r) * TLD + jc] = __builtin_fma(w3r[r], d3j, 0.0) * __builtin_fma(-a2, a2, 1.0);
                    T_A2[(u0r + r) * TLD + jc] = a2;
                });
            }
            __syncthreads();
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
                static_for<0, 3>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    const double* wl = W1L + m * H + u0r;
                    double x = (wl[0] * dv1[0] + wl[1] * dv1[1]) + (wl[2] * dv1[2] + wl[3] * dv1[3]);
                    x += __shfl_xor(x, 16, 64);
                    x += __shfl_xor(x, 32, 64);
                    pg[m] = x;
                });
                if (kq == 0) static_for<0, 3>([&](auto mc) { GXP[(decltype(mc)::value * NSLOTS + jc) * 4 + w] = pg[decltype(mc)::value]; });
            }
        }
        __syncthreads();
        LS2_TICK(2)
        int pf_mu = -1;   // slot whose mu words are already in mq
#if UDE_LS2_MU_PREFETCH
        {
            const int q16 = l & 15;
            const unsigned nx = (unsigned)__ballot(l < 16 && RCS[q16] != 0);
            if (nx != 0u) {
                pf_mu = __builtin_ctz(nx);
                const long long g = RG[pf_mu];
                mu_load(p.slot_glob + (size_t)g * (2 * NSLK * H) + l + (size_t)RCOL[pf_mu] * (NSLK * H));
            }
        }
#endif
        // factors of this evaluation to the workspace, slot-major: wavefront w copies its own four slots (lane i = hidden unit i)
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
        double kr[NC];
        static_for<0, NC>([&](auto c) { kr[c] = 0.0; });
        if (ev) {
            double gx[3];
            static_for<0, 3>([&](auto mm) {
                const double* g4 = GXP + (decltype(mm)::value * NSLOTS + slot) * 4;
                gx[mm] = (g4[0] + g4[1]) + (g4[2] + g4[3]);
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
                f0l[lm < NC ? lm : 7] = ko;
                kl[0] = ko;
                req = RQ_NORM01;
            } else if (ph == PH_INIT1) {
                req = RQ_NORM2;
            } else {
                const int s = ph;
                kl[8 * s] = ko;
                if (s == S - 1) {
                    // perform_step! is complete: new state and the replicated part of the error norm
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
                    ph += 1;   // (no parameter pass between this evaluation and the next: the state machine moves on here)
                }
            }
        } else if (ph == PH_FLUSH) {
            req = RQ_FLUSH;
        }
        if (lm == 0) {
            REQI[slot] = req; REQZ[slot] = zero_req ? 1 : 0; RDT[slot] = dt; RG[slot] = gid; RCOL[slot] = col; ROK[slot] = ret == RET_SUCCESS ? 1 : 0;
        }
        zero_req = false;
        LS2_TICK(3)
        __syncthreads();
        LS2_TICK(4)

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
                if (mode == RQ_STEP) {
                    if (sl != pf_mu) mu_load(mcur);   // (the first step-end request of the trip: fetched in front of phase D)
                    const double ps = slot_pass<S, MASK, 0>(fb, a1s, XF, sl, l, w, TB + 256, TB + 272, dt_req, o.abstol, o.reltol, mq, mnew, hh, ll);
                    const double tot = group_sum<64>(ps);
                    if (l == 0) sw[0] = tot;
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
                    for (int k = QW * w; k < QW * w + QW; ++k)
                        if (!GEN || (l < H2 && k < H1)) row[oW2 + l + k * H2] = ok ? mcur[(size_t)k * H] : 0.0;
                    static_for<0, 7>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        if ((e >> 1) == w) {
                            const int idx = e < 3 ? (l < H1 ? oW1 + l + e * H1 : -1) : e == 3 ? (l < H1 ? oB1 + l : -1) : e == 4 ? (l < H2 ? oB2 + l : -1) :
                                            e == 5 ? (l < H2 ? oW3 + l : -1) : (l == 0 ? oB3 : -1);
                            if (idx >= 0) row[idx] = ok ? mcur[(size_t)(H + e) * H] : 0.0;
                        }
                    });
                }
            }
        }
        LS2_TICK(5)
        __syncthreads();
        LS2_TICK(6)

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
        LS2_TICK(7)
#if defined(UDE_LS2_CLOCKS)
        ntrip += 1;
#endif
    }
#if defined(UDE_LS2_CLOCKS)
    if (p.trace && blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 8; ++i) p.trace[i] = (double)tsec[i];
        p.trace[8] = (double)ntrip;
    }
#endif
}

}  // namespace seirls2
}  // namespace ude
