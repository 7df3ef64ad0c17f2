// ude_models.h -- the UDE right-hand-side families (a3/a4/a5) as device policies.
//
// A model policy provides, for a lane group of size G:
//   NS   replicated state components            NSL  per-lane slots of the parameter cotangent
//   rhs(ctx, u, du)                             du = f(u, theta)
//   vjp<WANT_PARAM>(ctx, u, lam, dlam, g)       dlam = (df/du)^T lam ; g[slot] = this lane's (df/dtheta)^T lam
//   slot_index(mc, r, s)                        theta index of lane r's slot s, or -1 (padding)
#pragma once
#include "ude_coop.h"

#ifndef SEIR_FWD_BLOCKS
#define SEIR_FWD_BLOCKS 2  // (3: 3.87 instead of 3.95 ms for 552 B of scratch per lane -- not worth it)
#endif
namespace ude {

// neighbours on a periodic grid of n points, for 0 <= i < n: (i - 1) mod n and (i + 1) mod n without the integer division a runtime
// `% n` costs (~25 instructions each; round 4: two of them per stencil point sat in every Fisher-KPP loop)
__device__ __forceinline__ int wrap_prev(int i, int n) { return i == 0 ? n - 1 : i - 1; }
__device__ __forceinline__ int wrap_next(int i, int n) { return i + 1 == n ? 0 : i + 1; }


struct ModelConsts {
    int32_t n_state, n_param, nn_offset, stencil_offset, d0_offset;
    int32_t kind, n_layers, dims[9], act[8];  // the descriptor's chain (read by the runtime-shape model only, ude_model_generic.h)
    int32_t lin_idx[2];
    double lin_sign[2], lin_const[2];
    double consts[16];
};


// default: theta is copied linearly into LDS
struct LinearTheta {
    static __host__ __device__ constexpr int theta_lds(int np) { return (np + 1) & ~1; }
    static __device__ __forceinline__ void stage_theta(real* th, const real* theta, int np, int tid, int nthreads) {
        for (int i = tid; i < np; i += nthreads) th[i] = theta[i];
    }
    static constexpr int SCRATCH = 0;
    static constexpr bool THETA_GLOBAL = false;  // init() receives the LDS copy of theta
    static constexpr bool FUSED_ACC = false;     // parameter cotangent returned as g[] (not folded into accumulators)
    static constexpr bool SLOTS_GLOBAL = false;  // slot state mu in LDS (per-thread column)
    static constexpr bool CPL = false;           // component-per-lane stage storage (replicated small states, G = 64)
    static constexpr bool DEFERRED = false;      // adjoint keeps per-stage FACTORS of the parameter cotangent (see SeirUde)
    static constexpr bool DADJ_K_FROM_DENSE = false;  // reverse sweep reads the stage derivatives from HBM (no LDS copy)
};

// ---------------------------------------------------------------------------------------------
// lotka!  (LotkaVolterra/scenario_1.jl:30-34): theta = (alpha, beta, gamma, delta).  No network:
// every lane of the group computes the same thing (use G = 1).
// ---------------------------------------------------------------------------------------------
template <int G>
struct LvTrue : LinearTheta {
    static constexpr int NS = 2, NSL = 4, NTHETA_LDS = 4;
    static constexpr bool STATE_DISTRIBUTED = false;
    struct Ctx {
        const real* th;
        int r;
    };
    static __device__ __forceinline__ void init(Ctx& c, real* th_lds, real*, real*, int, const ModelConsts&, int r, const real* = nullptr) {
        c.th = th_lds;
        c.r = r;
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const real* u, real* du) {
        const real a = c.th[0], b = c.th[1], g = c.th[2], d = c.th[3];
        du[0] = a * u[0] - b * u[1] * u[0];
        du[1] = g * u[0] * u[1] - d * u[1];
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const real* u, const real* lam, real* dlam,
                                               real* g) {
        const real a = c.th[0], b = c.th[1], gm = c.th[2], d = c.th[3];
        dlam[0] = (a - b * u[1]) * lam[0] + (gm * u[1]) * lam[1];
        dlam[1] = (-b * u[0]) * lam[0] + (gm * u[0] - d) * lam[1];
        if constexpr (WANT_PARAM) {
            const real on = c.r == 0 ? 1.0 : 0.0;
            g[0] = on * (u[0] * lam[0]);
            g[1] = on * (-u[1] * u[0] * lam[0]);
            g[2] = on * (u[0] * u[1] * lam[1]);
            g[3] = on * (-u[1] * lam[1]);
        }
    }
    static __device__ __forceinline__ int slot_index(const ModelConsts&, int r, int s) { return r == 0 ? s : -1; }
};

// ---------------------------------------------------------------------------------------------
// ude_dynamics!  (scenario_1.jl:69-73; scenario_2.jl:90-95 trainable delta; hudson_bay.jl:85-91
// trainable p1,p2):  du_i = lin_i * u_i + NN_i(u),  lin_i = lin_const_i or lin_sign_i*theta[lin_idx_i]
// ---------------------------------------------------------------------------------------------
// NLIN: slots for trainable diagonal coefficients (2: scenario_2's delta / hudson_bay's p1, p2 may be trained; 0: both
// diagonal coefficients are constants as in scenario_1.jl:69-73 -- no slots, no accumulators, no divisions for them)
// WREG = false: the weights stay in LDS (the register budget of the two-wavefronts-per-SIMD instances)
// (LvUde's register activation cache exists exactly where its weights are register-resident: the member type AdjSys looks for)
template <bool ON, class Mlp> struct LvActReg {};
template <class Mlp> struct LvActReg<true, Mlp> { using ActCache = typename Mlp::Cache; };

template <class Net, int G, int NLIN = 2, bool WREG = true>
struct LvUde : LinearTheta, LvActReg<(WREG && (G >= 5) && (Net::maxdim() <= 8)), CoopMlp<Net, G>> {
    using Mlp = CoopMlp<Net, G>;
    static_assert(Net::dim(0) == 2 && Net::dim(Net::L) == 2, "LV UDE network maps R^2 -> R^2");
    static_assert(NLIN == 0 || NLIN == 2, "none or both diagonal slots");
    static constexpr int NS = 2;
    static constexpr int NSL = Mlp::NSLOT + NLIN;  // + the two (optional) trainable diagonal coefficients
    static constexpr bool STATE_DISTRIBUTED = false;
    // weights in registers when the lane's share is small (narrow layers spread over >= 5 lanes), else read from LDS
    static constexpr bool REGW = WREG && (G >= 5) && (Net::maxdim() <= 8);
    // per-member parameters (UDE_PT_THETA): with the weights in registers theta is only read by init(); wide nets (2-32-2), whose
    // weights are read where they lie at every use, read the member's own column in HBM instead of the block's LDS copy then
    // (every lane of a group reads the same words: broadcast loads, L2-resident -- slower than the shared-theta path, same bits)
    static constexpr bool PER_MEMBER_THETA = true;
    struct Ctx {
        const real* th;   // full theta (LDS)
        const real* nn;   // th + nn_offset
        real lin[2];
        real lead_on[2];  // sign if this lane owns a trainable diagonal coefficient, else 0
        int r;
        typename Mlp::lds_t* gb;  // LDS gather row of this lane group (non-power-of-two groups)
        typename Mlp::WReg w;  // (unused members are never materialised when REGW is false)
    };
    static constexpr int SCRATCH = 64;  // one gather word per lane (LV blocks are one wavefront; every LDS byte counts: 4 blocks per CU)
    static __device__ __forceinline__ void init(Ctx& c, real* th_lds, real* scratch, real*, int, const ModelConsts& mc, int r, const real* = nullptr) {
        c.gb = (typename Mlp::lds_t*)scratch + (threadIdx.x - r);
        c.th = th_lds;
        c.nn = th_lds + mc.nn_offset;
        c.r = r;
        for (int i = 0; i < 2; ++i) {
            c.lin[i] = mc.lin_idx[i] >= 0 ? (real)mc.lin_sign[i] * th_lds[mc.lin_idx[i]] : (real)mc.lin_const[i];
            c.lead_on[i] = (mc.lin_idx[i] >= 0 && r == 0) ? (real)mc.lin_sign[i] : real(0);
        }
        if constexpr (Net::RT && REGW) Mlp::load_weights_rt(c.nn, r, c.w, mc);          // run-time shape, padded register copy
        else if constexpr (Net::RT) Mlp::load_rt_info(c.w.rt, mc);                     // run-time shape, weights read where they lie (widths > 8)
        else if constexpr (REGW) Mlp::load_weights(c.nn, r, c.w);
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const real* u, real* du) {
        typename Mlp::Cache cache;
        cache.gb = c.gb;
        real y[2];
        if constexpr (REGW) Mlp::forward(c.w, c.r, u, cache, y);
        else if constexpr (Net::RT) Mlp::forward(typename Mlp::PtrRt{c.nn, &c.w.rt}, c.r, u, cache, y);
        else Mlp::forward(c.nn, c.r, u, cache, y);
        du[0] = rfma(c.lin[0], u[0], y[0]);
        du[1] = rfma(c.lin[1], u[1], y[1]);
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const real* u, const real* lam, real* dlam,
                                               real* g) {
        typename Mlp::Cache cache;
        cache.gb = c.gb;
        real y[2], gx[2];
        if constexpr (REGW) {
            Mlp::forward(c.w, c.r, u, cache, y);
            Mlp::template vjp<WANT_PARAM>(c.w, c.r, cache, lam, gx, g);
        } else if constexpr (Net::RT) {
            const typename Mlp::PtrRt pr{c.nn, &c.w.rt};
            Mlp::forward(pr, c.r, u, cache, y);
            Mlp::template vjp<WANT_PARAM>(pr, c.r, cache, lam, gx, g);
        } else {
            Mlp::forward(c.nn, c.r, u, cache, y);
            Mlp::template vjp<WANT_PARAM>(c.nn, c.r, cache, lam, gx, g);
        }
        dlam[0] = rfma(c.lin[0], lam[0], gx[0]);
        dlam[1] = rfma(c.lin[1], lam[1], gx[1]);
        if constexpr (WANT_PARAM && NLIN == 2) {
            g[Mlp::NSLOT + 0] = (c.lead_on[0] * u[0]) * lam[0];
            g[Mlp::NSLOT + 1] = (c.lead_on[1] * u[1]) * lam[1];
        }
    }
    // round 6: the activations of the LAST adjoint evaluation stay in registers (AdjSys::ACT_REG, ude_kernels.h): the network input of an adjoint
    // evaluation is the interpolated forward state u(t), a function of t alone, so an evaluation at the same time as the one before it -- the
    // second of the two stages at t + dt that both tableaux end with, the evaluation after a save-time jump -- skips its forward pass (three
    // `exp` / `tanh` per lane, the layer dots and their gathers) and runs the reverse sweep on the activations it finds.  Same numbers, same bits.
    // Only where the weights are register-resident too (narrow nets on >= 5 lanes: ActCache comes from LvActReg below); configs[1]: adj_kernel -4.5 %.
    template <bool WANT_PARAM, class AC>
    static __device__ __forceinline__ void vjp_c(const Ctx& c, const real* u, const real* lam, real* dlam, real* g, AC& cache, bool same) {
        static_assert(REGW, "register-resident weights");
        cache.gb = c.gb;
        real y[2], gx[2];
        if (!same) Mlp::forward(c.w, c.r, u, cache, y);
        Mlp::template vjp<WANT_PARAM>(c.w, c.r, cache, lam, gx, g);
        dlam[0] = rfma(c.lin[0], lam[0], gx[0]);
        dlam[1] = rfma(c.lin[1], lam[1], gx[1]);
        if constexpr (WANT_PARAM && NLIN == 2) {
            g[Mlp::NSLOT + 0] = (c.lead_on[0] * u[0]) * lam[0];
            g[Mlp::NSLOT + 1] = (c.lead_on[1] * u[1]) * lam[1];
        }
    }
    static __device__ __forceinline__ int slot_index(const ModelConsts& mc, int r, int s) {
        if (s >= Mlp::NSLOT) {
            const int i = s - Mlp::NSLOT;
            return (r == 0 && mc.lin_idx[i] >= 0) ? mc.lin_idx[i] : -1;
        }
        int k;
        if constexpr (Net::RT) k = Mlp::slot_index_rt(mc, r, s);
        else k = Mlp::slot_index(r, s);
        return k < 0 ? -1 : mc.nn_offset + k;
    }
};

#ifndef UDE_F32  // Float64-only models (no Float32 problem of the reference uses them): absent from -DUDE_F32 translation units
// ---------------------------------------------------------------------------------------------
// corona!  (SEIR_exposure/seir_exposure.jl:16-30): mechanistic 7-state model, consts = p_[0..8] =
// F, beta0, alpha, kappa, mu, sigma, gamma, d, lambda.  No trainable parameters (data generation).
// ---------------------------------------------------------------------------------------------
template <int G>
struct SeirTrue : LinearTheta {
    static constexpr int NS = 7, NSL = 0;
    static constexpr bool STATE_DISTRIBUTED = false;
    struct Ctx {
        double F, b0, al, ka, mu, sg, ga, d, la;
        int r;
    };
    static __device__ __forceinline__ void init(Ctx& c, double*, double*, double*, int, const ModelConsts& mc, int r, const double* = nullptr) {
        c.F = mc.consts[0]; c.b0 = mc.consts[1]; c.al = mc.consts[2]; c.ka = mc.consts[3]; c.mu = mc.consts[4];
        c.sg = mc.consts[5]; c.ga = mc.consts[6]; c.d = mc.consts[7]; c.la = mc.consts[8];
        c.r = r;
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const double* u, double* du) {
        const double S = u[0], E = u[1], I = u[2], Rr = u[3], N = u[4], D = u[5];
        const double beta = c.b0 * (1.0 - c.al) * dpow(1.0 - D / N, c.ka);
        du[0] = -c.b0 * S * c.F / N - beta * S * I / N - c.mu * S;
        du[1] = c.b0 * S * c.F / N + beta * S * I / N - (c.sg + c.mu) * E;
        du[2] = c.sg * E - (c.ga + c.mu) * I;
        du[3] = c.ga * I - c.mu * Rr;
        du[4] = -c.mu * N;
        du[5] = c.d * c.ga * I - c.la * D;
        du[6] = c.sg * E;
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx&, const double*, const double*, double*, double*) {}
    static __device__ __forceinline__ int slot_index(const ModelConsts&, int, int) { return -1; }
};

// ---------------------------------------------------------------------------------------------
// dudt_(u,p,t) = corona with the exposure term replaced by ann([S/N, I, D/N], p)[1]
//   (SEIR_exposure/seir_exposure.jl:114-131), ann = 3 -> 64 (tanh) -> 64 (tanh) -> 1, theta = FastChain layout
//   [W1 (64x3, column-major); b1; W2 (64x64); b2; W3 (1x64); b3] = 4481 parameters.
// One trajectory per BLOCK of NW = G/64 wavefronts.  Everything the network needs lives in registers:
//   lane (w, j) holds row j of W2 restricted to ITS k-blocks (forward), column j restricted to its i-blocks
//   (transposed product), W1[j,:], b1[j], b2[j], W3[j]; the 64-term dots follow the ARITH-SPEC wide-dot rule
//   (oracle: wide_dot): hidden-layer dots = 4 blocks of 16 terms, the blocks dealt to the wavefronts, block sums
//   exchanged through LDS and added left to right; the two reductions to a replicated scalar (output layer,
//   input cotangent) are wavefront tree sums.  Activations cross lanes with v_readlane (no LDS round trip).
// The parameter cotangent is register-slot state: lane (w, j) owns W2[j, its k's] plus a share of the 7 "extra"
// per-neuron parameter rows, so mu and the RK accumulators never touch LDS inside an evaluation.
// ---------------------------------------------------------------------------------------------
template <int G>
struct SeirUde {
    static_assert(G == 64 || G == 128 || G == 256, "SEIR UDE kernel: 1, 2 or 4 wavefronts per trajectory");
    static constexpr int NW = G / 64, H = 64, NBLK = 4, BPW = NBLK / NW, KB = 16 * BPW;  // k's per lane
    static constexpr int NS = 7;
    static constexpr int NEXTRA = 7, XS = (NEXTRA + NW - 1) / NW;  // W1[:,0..2], b1, b2, W3, b3
    static constexpr int NSL = KB + XS;
    static constexpr bool STATE_DISTRIBUTED = false;
    // one wavefront per trajectory (G = 64): W2 is read from a padded LDS copy shared by the block's trajectories,
    // the 71 parameter-cotangent accumulators per lane stay in registers (fused accumulation: no g[] array) and mu
    // itself (touched once per step) lives in HBM.  G = 128/256: W2 slices in registers, theta read from HBM once.
    static constexpr bool ONE = NW == 1;
    static constexpr bool THETA_GLOBAL = !ONE, FUSED_ACC = ONE, SLOTS_GLOBAL = ONE, CPL = ONE;
    // DEFERRED (ONE): the W2 cotangent of a stage is the outer product -delta2 (x) a1, so the adjoint pass stores the
    // FACTORS of every stage (5 doubles per lane and stage, LDS) and forms the RK-weighted sums slot by slot at the
    // end of the step, fused with the error norm and the mu update: the identical fma chains (ascending stage
    // order), but no 2 x 71 persistent accumulators per lane and no per-evaluation broadcast of a1 for them.
    static constexpr bool DEFERRED = ONE;
    static constexpr bool DADJ_K_FROM_DENSE = false;
    static constexpr bool COMPACT_STAGES = false;
    static constexpr int NSTG = 10, NFAC = 5, WPB = 4;  // stages stored (Vern7), factor fields, wavefronts per block
    static constexpr int LD = 65;  // leading dimension of the LDS copy of W2: row AND column reads conflict-free
    static constexpr int NPARAM = 3 * H + H + H * H + H + H + 1;  // 4481
    static constexpr int OFF_W1 = 0, OFF_B1 = 3 * H, OFF_W2 = 4 * H, OFF_B2 = 4 * H + H * H, OFF_W3 = OFF_B2 + H,
                         OFF_B3 = OFF_W3 + H;
    static constexpr int SCRATCH = ONE ? WPB * (NSTG * NFAC + 2) * H : 3 * NBLK * H;  // stage factors + 2 broadcast rows / block sums
    static constexpr int SCRATCH_FWD = ONE ? WPB * 2 * H : SCRATCH;  // forward / rhs kernels: no stage factors (41 KB per block)
    static constexpr int FWD_BLOCKS = ONE ? SEIR_FWD_BLOCKS : 1;
    typedef __attribute__((address_space(3))) double lds_t;
    struct Ctx {
        double w2row[ONE ? 1 : KB], w2col[ONE ? 1 : KB], w1[3], b1, b2, w3, b3;
        const lds_t* W2p;   // LDS, ld = 65 (ONE)
        lds_t* bc;          // ONE: two wave-private broadcast rows (a1 / delta2): lane j writes, every lane reads all 64
        double *pf, *pb;    // LDS block-sum exchange (multi-wave)
        lds_t* fac;         // LDS stage factors of this wavefront: field f of stage s at fac[(s*NFAC + f)*H + lane]
        double F, b0, mu_c, sg, ga, d, la;
        int j, w, r;
        mutable int flip;
    };
    static __host__ __device__ constexpr int theta_lds(int) { return ONE ? ((H * LD + 1) & ~1) : 0; }
    static __device__ __forceinline__ void stage_theta(double* th, const double* theta, int, int tid, int nthreads) {
        if constexpr (ONE)
            for (int i = tid; i < H * H; i += nthreads) th[(i % H) + (i / H) * LD] = theta[OFF_W2 + i];
    }
    // th: LDS copy of W2 (ONE) / theta in HBM (multi-wave); theta_g: theta in HBM
    static __device__ __forceinline__ void init(Ctx& c, double* th, double* scratch, double*, int,
                                                const ModelConsts& mc, int r, const double* theta_g) {
        const int j = r & 63;
        const int w = __builtin_amdgcn_readfirstlane(r >> 6);
        c.j = j; c.w = w; c.r = r; c.flip = 0;
        c.W2p = (const lds_t*)th;
        c.bc = (lds_t*)scratch + ((threadIdx.x >> 6) % WPB) * 2 * H;  // (the broadcast rows first: all the forward kernels need of the scratch)
        c.pf = scratch; c.pb = scratch + 2 * NBLK * H;
        c.fac = (lds_t*)scratch + WPB * 2 * H + ((threadIdx.x >> 6) % WPB) * (NSTG * NFAC * H);
        if constexpr (!ONE)
            static_for<0, KB>([&](auto i) {
                const int k = w * KB + i;
                c.w2row[i] = theta_g[OFF_W2 + j + k * H];  // W2[j, k]
                c.w2col[i] = theta_g[OFF_W2 + k + j * H];  // W2[k, j]
            });
        static_for<0, 3>([&](auto m) { c.w1[m] = theta_g[OFF_W1 + j + m * H]; });
        c.b1 = theta_g[OFF_B1 + j]; c.b2 = theta_g[OFF_B2 + j]; c.w3 = theta_g[OFF_W3 + j]; c.b3 = theta_g[OFF_B3];
        c.F = mc.consts[0]; c.b0 = mc.consts[1]; c.mu_c = mc.consts[4]; c.sg = mc.consts[5]; c.ga = mc.consts[6];
        c.d = mc.consts[7]; c.la = mc.consts[8];
    }
    // 64-term hidden dot (ARITH-SPEC wide-dot rule: 4 blocks of 16, block sums added left to right)
    //   ONE: the lane runs the four chains itself, W2 from LDS (row j, or column j when TRANSPOSED)
    //   multi-wave: this wavefront's blocks -> LDS, barrier, all four added
    template <bool TRANSPOSED>
    static __device__ __forceinline__ double hidden_dot(const Ctx& c, double v, double* buf, double* vk) {
        if constexpr (ONE) {
            // one block of 16 terms per trip of a RUNTIME loop: bounds the loads / scalars in flight (the fully
            // unrolled 64-term dot made the compiler hoist everything and spill)
            // the 64 inputs cross lanes through a wave-private LDS row (uniform-address reads broadcast; LDS is in
            // order per wavefront, so the row needs no barrier) -- cheaper than 128 v_readlane + hazard nops per dot
            double tot = 0.0;
            const lds_t* wp = TRANSPOSED ? c.W2p + c.j * LD : c.W2p + c.j;
            lds_t* row = c.bc + (TRANSPOSED ? H : 0);
            row[c.j] = v;
#pragma unroll 1
            for (int b = 0; b < NBLK; ++b) {
                double acc = 0.0;
                static_for<0, 16>([&](auto ic) {
                    const int k = b * 16 + decltype(ic)::value;
                    const double x = row[k];
                    const double wv = TRANSPOSED ? wp[k] : wp[k * LD];
                    acc = __builtin_fma(wv, x, acc);
                });
                tot = b == 0 ? acc : tot + acc;
            }
            return tot;
        } else {
            static_for<0, BPW>([&](auto bc) {
                constexpr int b = bc;
                double acc = 0.0;
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = b * 16 + ic;
                    const double x = readlane_real(v, c.w * KB + i);
                    if (vk) vk[i] = x;
                    acc = __builtin_fma(TRANSPOSED ? c.w2col[i] : c.w2row[i], x, acc);
                });
                buf[(c.w * BPW + b) * H + c.j] = acc;
            });
            __syncthreads();
            double tot = buf[c.j];
            static_for<1, NBLK>([&](auto b) { tot += buf[b * H + c.j]; });
            return tot;
        }
    }
    // forward network; a1k (optional, multi-wave): this lane's k-range of the first hidden activation
    static __device__ __forceinline__ double net(const Ctx& c, const double* x, double& a1, double& a2, double* a1k) {
        double z1 = 0.0;
        static_for<0, 3>([&](auto k) { z1 = __builtin_fma(c.w1[k], x[k], z1); });
        z1 += c.b1;
        a1 = dtanh(z1);
        double* pf = c.pf + (c.flip & 1) * NBLK * H;
        c.flip ^= 1;
        const double z2 = hidden_dot<false>(c, a1, pf, a1k) + c.b2;
        a2 = dtanh(z2);
        return wave_tree_sum(c.w3 * a2) + c.b3;
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const double* u, double* du) {
        const double S = u[0], E = u[1], I = u[2], Rr = u[3], N = u[4], D = u[5];
        const double x[3] = {S / N, I, D / N};
        double a1, a2;
        const double z = net(c, x, a1, a2, nullptr);
        du[0] = -c.b0 * S * c.F / N - z - c.mu_c * S;
        du[1] = c.b0 * S * c.F / N + z - (c.sg + c.mu_c) * E;
        du[2] = c.sg * E - (c.ga + c.mu_c) * I;
        du[3] = c.ga * I - c.mu_c * Rr;
        du[4] = -c.mu_c * N;
        du[5] = c.d * c.ga * I - c.la * D;
        du[6] = c.sg * E;
    }
    // extra parameter row e (0..6) of neuron j: theta index
    static __device__ __forceinline__ int extra_index(int e, int j) {
        switch (e) {
            case 0: case 1: case 2: return OFF_W1 + j + e * H;
            case 3: return OFF_B1 + j;
            case 4: return OFF_B2 + j;
            case 5: return OFF_W3 + j;
            default: return (e == 6 && j == 0) ? OFF_B3 : -1;
        }
    }
    // shared part of the reverse sweep: deltas of the three layers and the state cotangent
    struct Bwd {
        double x[3], a1, a2, d1, d2, d3;
    };
    static __device__ __forceinline__ void sweep(const Ctx& c, const double* u, const double* lam, double* dlam, Bwd& q,
                                                 double* a1k) {
        const double S = u[0], N = u[4], D = u[5];
        q.x[0] = S / N; q.x[1] = u[2]; q.x[2] = D / N;
        net(c, q.x, q.a1, q.a2, a1k);
        q.d3 = (lam[1] - lam[0]) * 1.0;  // output layer is linear
        q.d2 = __builtin_fma(c.w3, q.d3, 0.0) * __builtin_fma(-q.a2, q.a2, 1.0);
        const double s1 = hidden_dot<true>(c, q.d2, c.pb, nullptr);
        q.d1 = s1 * __builtin_fma(-q.a1, q.a1, 1.0);
        double gx[3];
        static_for<0, 3>([&](auto m) { gx[m] = wave_tree_sum(c.w1[m] * q.d1); });
        const double cc = c.b0 * c.F / N;
        const double cN = c.b0 * S * c.F / (N * N);
        dlam[0] = (-cc - c.mu_c) * lam[0] + cc * lam[1] + gx[0] / N;
        dlam[1] = -(c.sg + c.mu_c) * lam[1] + c.sg * lam[2] + c.sg * lam[6];
        dlam[2] = -(c.ga + c.mu_c) * lam[2] + c.ga * lam[3] + c.d * c.ga * lam[5] + gx[1];
        dlam[3] = -c.mu_c * lam[3];
        dlam[4] = cN * lam[0] - cN * lam[1] - c.mu_c * lam[4] - gx[0] * S / (N * N) - gx[2] * D / (N * N);
        dlam[5] = -c.la * lam[5] + gx[2] / N;
        dlam[6] = 0.0;
    }
    static __device__ __forceinline__ double extra_value(const Ctx& c, const Bwd& q, int e) {
        switch (e) {
            case 0: return q.d1 * q.x[0];
            case 1: return q.d1 * q.x[1];
            case 2: return q.d1 * q.x[2];
            case 3: return q.d1;
            case 4: return q.d2;
            case 5: return q.d3 * q.a2;
            case 6: return (c.j == 0) ? q.d3 : 0.0;
            default: return 0.0;
        }
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const double* u, const double* lam, double* dlam,
                                               double* g) {
        Bwd q;
        double a1k[KB];
        sweep(c, u, lam, dlam, q, ONE ? nullptr : a1k);
        if constexpr (WANT_PARAM && !ONE) {
            static_for<0, KB>([&](auto i) { g[i] = q.d2 * a1k[i]; });
            static_for<0, XS>([&](auto qq) { g[KB + qq] = extra_value(c, q, qq * NW + c.w); });  // wave-uniform row
        }
    }
    // fused accumulation (ONE): ab[s] = fma(bs, -g_s, ab[s]) (ae likewise with es) for this lane's 71 parameters;
    // g = (df/dtheta)^T lam, so the accumulators hold the NEGATED cotangent the adjoint ODE integrates
    template <bool WANT_E>
    static __device__ __forceinline__ void vjp_acc(const Ctx& c, const double* u, const double* lam, double* dlam,
                                                   double* ab, double* ae, double bs, double es) {
        Bwd q;
        sweep(c, u, lam, dlam, q, nullptr);
        auto upd = [&](auto sc, double gpos) {
            constexpr int s = sc;
            ab[s] = __builtin_fma(bs, -gpos, ab[s]);
            if constexpr (WANT_E) ae[s] = __builtin_fma(es, -gpos, ae[s]);
        };
        static_for<0, H>([&](auto k) { upd(k, q.d2 * readlane_real(q.a1, decltype(k)::value)); });
        static_for<0, NEXTRA>([&](auto e) { upd(std::integral_constant<int, H + decltype(e)::value>{}, extra_value(c, q, decltype(e)::value)); });
    }
    // ---- deferred parameter cotangent (ONE) ----
    // reverse sweep at stage s: state cotangent out, factors (a1, a2, delta1, delta2 | x0 x1 x2 d3 on lanes 0..3) to LDS
    static __device__ __forceinline__ void vjp_store(const Ctx& c, const double* u, const double* lam, double* dlam, int s) {
        Bwd q;
        sweep(c, u, lam, dlam, q, nullptr);
        lds_t* f = c.fac + s * (NFAC * H) + c.j;
        f[0] = q.a1; f[H] = q.a2; f[2 * H] = q.d1; f[3 * H] = q.d2;
        f[4 * H] = c.j == 0 ? q.x[0] : c.j == 1 ? q.x[1] : c.j == 2 ? q.x[2] : q.d3;
    }
    // g_s of slot `slot` of this lane: the NEGATED cotangent -(df/dtheta)^T lam at stage s (what the adjoint integrates)
    //   slot k < 64: W2[j,k]: -(delta2_s[j] * a1_s[k]);  64..66: W1[j,m]: -(delta1*x_m); 67: b1: -delta1; 68: b2: -delta2;
    //   69: W3[j]: -(d3*a2); 70: b3 (lane 0): -d3
    struct Fac {  // this lane's factors of all stages (registers)
        double a2[NSTG], d1[NSTG], d2[NSTG];
    };
    // MASK: bit s set = stage s carries a nonzero B or BT weight (stages with both zero contribute fma(0, g, acc) == acc
    // exactly, so their factors are never loaded and their products never formed)
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ void load_factors(const Ctx& c, Fac& f) {
        static_for<0, NST>([&](auto s) {
            if constexpr ((MASK >> decltype(s)::value) & 1u) {
                const lds_t* p = c.fac + decltype(s)::value * (NFAC * H) + c.j;
                f.a2[s] = p[H]; f.d1[s] = p[2 * H]; f.d2[s] = p[3 * H];
            }
        });
    }
    // g_s (all stored stages) of one slot; slot index wave-uniform
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ void g_w2(const Ctx& c, const Fac& f, int k, double* g) {
        static_for<0, NST>([&](auto s) {
            if constexpr ((MASK >> decltype(s)::value) & 1u) g[s] = -(f.d2[s] * c.fac[decltype(s)::value * (NFAC * H) + k]);
        });
    }
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ void g_extra(const Ctx& c, const Fac& f, int e, double* g) {
        static_for<0, NST>([&](auto s) {
          if constexpr ((MASK >> decltype(s)::value) & 1u) {
            const lds_t* p = c.fac + decltype(s)::value * (NFAC * H) + 4 * H;  // x0 x1 x2 d3
            double v;
            switch (e) {
                case 0: v = -(f.d1[s] * p[0]); break;
                case 1: v = -(f.d1[s] * p[1]); break;
                case 2: v = -(f.d1[s] * p[2]); break;
                case 3: v = -f.d1[s]; break;
                case 4: v = -f.d2[s]; break;
                case 5: v = -(p[3] * f.a2[s]); break;
                default: v = c.j == 0 ? -p[3] : -0.0;
            }
            g[s] = v;
          }
        });
    }
    // slots in order 0..70, mu read in chunks of CH (next chunk in flight while this one is processed):
    // body(slot, g[NST], m) with m = mu[slot]
    template <int NST, unsigned MASK, class Body>
    static __device__ __forceinline__ void for_each_slot(const Ctx& c, const Fac& f, const double* mu, int ms, Body body) {
        constexpr int CH = 8;
        double mcur[CH], mnext[CH];
        static_for<0, CH>([&](auto i) { mcur[i] = mu[(size_t)decltype(i)::value * ms]; });
#pragma unroll 1
        for (int k0 = 0; k0 < H; k0 += CH) {
            // prefetch: next W2 chunk, or (last round) the 7 extras
            static_for<0, CH>([&](auto i) {
                const int sl = k0 + CH + decltype(i)::value;
                mnext[i] = sl < NSL ? mu[(size_t)sl * ms] : 0.0;
            });
            static_for<0, CH>([&](auto i) {
                double g[NST];
                g_w2<NST, MASK>(c, f, k0 + decltype(i)::value, g);
                body(k0 + decltype(i)::value, g, mcur[i]);
            });
            static_for<0, CH>([&](auto i) { mcur[i] = mnext[i]; });
        }
        static_for<0, NEXTRA>([&](auto e) {
            double g[NST];
            g_extra<NST, MASK>(c, f, decltype(e)::value, g);
            body(H + decltype(e)::value, g, mcur[e]);
        });
    }
    // end of a step with NST stages: for every slot, ab = sum_s B_s g_s and ae = sum_s BT_s g_s (fma chains in ascending
    // stage order, started by the product), candidate mu_new = fma(dt, ab, mu), residual^2 accumulated in slot order
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ double step_slots(const Ctx& c, const double* B, const double* BT, double dt,
                                                        double abstol, double reltol, const double* mu, double* mu_new,
                                                        int ms) {
        static_assert(MASK & 1u, "the first stage starts the chains");
        Fac f;
        load_factors<NST, MASK>(c, f);
        double bb[NST], bt[NST];  // tableau weights as scalars (one load per step, not per slot)
        static_for<0, NST>([&](auto s) { bb[s] = uniform_real(B[s]); bt[s] = uniform_real(BT[s]); });
        double ps = 0.0;
        for_each_slot<NST, MASK>(c, f, mu, ms, [&](int slot, const double* g, double m0) {
            double ab = bb[0] * g[0], ae = bt[0] * g[0];
            static_for<1, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) {
                    ab = __builtin_fma(bb[s], g[s], ab);
                    ae = __builtin_fma(bt[s], g[s], ae);
                }
            });
            const double m1 = __builtin_fma(dt, ab, m0);
            mu_new[(size_t)slot * ms] = m1;
            const double a0 = fabs(m0), a1 = fabs(m1);
            const double res = (dt * ae) / __builtin_fma((a0 > a1 ? a0 : a1), reltol, abstol);
            ps = __builtin_fma(res, res, ps);
        });
        return ps;
    }
    // fast adjoint mode (lambda-only error control): mu += dt * sum_s B_s g_s in place, on accepted steps only -- the
    // same chain and the same final fma as the candidate of step_slots, without the error half and its division
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ void commit_slots(const Ctx& c, const double* B, double dt, double* mu, int ms) {
        static_assert(MASK & 1u, "the first stage starts the chains");
        Fac f;
        load_factors<NST, MASK>(c, f);
        double bb[NST];
        static_for<0, NST>([&](auto s) { bb[s] = uniform_real(B[s]); });
        for_each_slot<NST, MASK>(c, f, mu, ms, [&](int slot, const double* g, double m0) {
            double ab = bb[0] * g[0];
            static_for<1, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) ab = __builtin_fma(bb[s], g[s], ab);
            });
            mu[(size_t)slot * ms] = __builtin_fma(dt, ab, m0);
        });
    }
    // initial-dt norms: stage 0 holds g0 = f0's slot part, stage 1 (second call) g1
    static __device__ __forceinline__ void init_norm01(const Ctx& c, double abstol, double reltol, const double* mu, int ms,
                                                       double& h0, double& l0, double& h1, double& l1) {
        Fac f;
        load_factors<1, 1u>(c, f);
        for_each_slot<1, 1u>(c, f, mu, ms, [&](int, const double* g, double m) {
            const double sk = __builtin_fma(fabs(m), reltol, abstol);
            const double q0 = m / sk, q1 = g[0] / sk;
            dd_acc(h0, l0, q0 * q0);
            dd_acc(h1, l1, q1 * q1);
        });
    }
    static __device__ __forceinline__ void init_norm2(const Ctx& c, double abstol, double reltol, const double* mu, int ms,
                                                      double& h2, double& l2) {
        Fac f;
        load_factors<2, 3u>(c, f);
        for_each_slot<2, 3u>(c, f, mu, ms, [&](int, const double* g, double m) {
            const double sk = __builtin_fma(fabs(m), reltol, abstol);
            const double q = (g[1] - g[0]) / sk;
            dd_acc(h2, l2, q * q);
        });
    }
    static __device__ __forceinline__ int slot_index(const ModelConsts&, int r, int s) {
        const int j = r & 63, w = r >> 6;
        if (s < KB) return OFF_W2 + j + (w * KB + s) * H;
        const int e = (s - KB) * NW + w;
        return e < NEXTRA ? extra_index(e, j) : -1;
    }
};

}  // namespace ude
#include "ude_model_node.h"
namespace ude {
#endif  // UDE_F32
}  // namespace ude
#ifdef UDE_INST_GENERIC  // (only the translation units of the runtime-shape instances pull the model in: build.py; Float64 and Float32)
#include "ude_model_generic.h"
#endif
#ifdef UDE_INST_KPPGEN   // the runtime-shape pointwise Fisher-KPP network
#include "ude_model_kpp_generic.h"
#endif
namespace ude {

// ---------------------------------------------------------------------------------------------
// Fisher-KPP (FisherKPP/Fisher-KPP-CNN.jl, LotkaVolterra/scenario_3.jl): 1-D reaction-diffusion on a periodic
// grid of n_state points.  The STATE is distributed: point i = c*G + r lives on lane r (register slot c);
// neighbours are exchanged through an LDS row.  One block per trajectory (BLOCK == G).
// ---------------------------------------------------------------------------------------------
// rc_ode(rho,p,t) = (D*lap)*rho + r*rho*(1-rho)   (Fisher-KPP-CNN.jl:51-63); consts = D/dx^2, -2D/dx^2, r
template <int G, int PPL>
struct KppTrue : LinearTheta {
    static __host__ __device__ constexpr int point(int c, int r) { return c * G + r; }  // grid point of (register slot, lane)
    static constexpr int NS = PPL, NSL = 0;
    static constexpr bool STATE_DISTRIBUTED = true;
    static constexpr int SCRATCH = G * PPL + 2;
    struct Ctx {
        real* row;
        real coff, cdiag, rr;
        int r, n;
    };
    static __device__ __forceinline__ void init(Ctx& c, real*, real* scratch, real*, int, const ModelConsts& mc, int r, const real* = nullptr) {
        c.row = scratch;
        c.coff = mc.consts[0]; c.cdiag = mc.consts[1]; c.rr = mc.consts[2];
        c.r = r; c.n = mc.n_state;
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const real* u, real* du) {
        const int n = c.n;
        __syncthreads();
        static_for<0, PPL>([&](auto cc) { const int i = cc * G + c.r; if (i < n) c.row[i] = u[cc]; });
        __syncthreads();
        static_for<0, PPL>([&](auto cc) {
            const int i = cc * G + c.r;
            if (i < n) {
                const int im = wrap_prev(i, n), ip = wrap_next(i, n);
                // ARITH-SPEC dense gemv model (oracle: UDEO_KIND_KPP_TRUE): the three nonzeros of the dense row in ascending
                // column order, column blocks of 8 -- a fused chain from 0 inside a block, block sums added in block order
                // (the shape that reproduces the Float32 golden 243 / 39 / 1, profiles/r03_f32_golden_search.md)
                int j0, j1, j2;
                real c0, c1, c2;
                if (i == 0) { j0 = i; c0 = c.cdiag; j1 = ip; c1 = c.coff; j2 = im; c2 = c.coff; }
                else if (i == n - 1) { j0 = ip; c0 = c.coff; j1 = im; c1 = c.coff; j2 = i; c2 = c.cdiag; }
                else { j0 = im; c0 = c.coff; j1 = i; c1 = c.cdiag; j2 = ip; c2 = c.coff; }
                const real p0 = rfma(c0, c.row[j0], real(0));  // (a chain STARTS as fma(a, x, +0): the oracle's sign of zero)
                const bool s01 = (j0 >> 3) == (j1 >> 3), s12 = (j1 >> 3) == (j2 >> 3);
                const real x1 = c.row[j1], x2 = c.row[j2];
                // [j0 j1 j2] | [j0 j1][j2] | [j0][j1 j2] | [j0][j1][j2]
                const real a01 = rfma(c1, x1, s01 ? p0 : real(0));           // chain through j1 (continued or restarted)
                const real y01 = s01 ? real(0) : p0;                         // closed block sum so far (only if j0's block ended)
                real acc;
                if (s12) {
                    const real a012 = rfma(c2, x2, a01);
                    acc = s01 ? a012 : y01 + a012;
                } else {
                    const real yb = s01 ? a01 : y01 + a01;
                    acc = yb + rfma(c2, x2, real(0));
                }
                du[cc] = acc + (c.rr * u[cc]) * (real(1) - u[cc]);
            } else {
                du[cc] = 0.0;
            }
        });
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx&, const real*, const real*, real*, real*) {}
    static __device__ __forceinline__ int slot_index(const ModelConsts&, int, int) { return -1; }
};

// nn_ode(u,p,t)  (Fisher-KPP-CNN.jl:111-126; scenario_3.jl:103-114):
//   du_i = NN(u_i) + D0*(w1*u_{i-1} + w2*u_i + w3*u_{i+1}),  theta = [NN; w1 w2 w3 unused; D0]
// Forward: every lane runs the whole pointwise network for its points (weights broadcast from LDS).
// Adjoint: per tile of G points the layer inputs a_l and deltas d_l go to LDS ([row][point]); then the lanes
// switch role and OWN PARAMETERS (theta index p = r + G*m): each accumulates sum_i d_l[j][i]*a_{l-1}[k][i]
// over the points in ascending order -- the oracle's order, so the result is bit-identical.
template <class Net, int G, int PPL>
struct KppUde : LinearTheta {
    static constexpr bool RECOMPUTE_OK = true;   // checkpointed adjoint available (AdjSys::RECOMPUTE)
    // per-member parameters (UDE_PT_THETA; Fisher-KPP-CNN-Small.jl:311-391 repeats its training five times from five initial networks):
    // theta is read through the pointer init() receives, so the member's own column in HBM stands in for the block's LDS copy
    static constexpr bool PER_MEMBER_THETA = true;
    static __host__ __device__ constexpr int point(int c, int r) { return c * G + r; }
    using Mlp = CoopMlp<Net, 1>;
    static_assert(Net::dim(0) == 1 && Net::dim(Net::L) == 1, "pointwise reaction network R -> R");
    static constexpr int NS = PPL;
    static constexpr int NP = Net::nparam + 5;
    static constexpr int NSL = (NP + G - 1) / G;
    static constexpr bool STATE_DISTRIBUTED = true;
    static constexpr int L = Net::L;
    static constexpr int rows_a() { int s = 0; for (int l = 0; l < L; ++l) s += Net::dim(l); return s; }
    static constexpr int rows_d() { int s = 0; for (int l = 0; l < L; ++l) s += Net::dim(l + 1); return s; }
    static constexpr int a_off(int l) { int s = 0; for (int i = 0; i < l; ++i) s += Net::dim(i); return s; }
    static constexpr int d_off(int l) { int s = 0; for (int i = 0; i < l; ++i) s += Net::dim(i + 1); return s; }
    static constexpr int NPT = G * PPL;
    static constexpr int RA = rows_a() | 1, RD = rows_d() | 1;                // odd row strides of the [point][row] tiles
    static constexpr int SCRATCH = 2 * NPT + 4 + (RA + RD) * G;               // u row, lambda row, A tile, D tile
    struct Ctx {
        const real* th;
        const real* nn;
        real *urow, *lrow, *A, *Dt;
        real w1, w2, w3, D0;
        int r, n, so, d0o, nno;
        int a_row[NSL], d_row[NSL];  // per owned parameter: LDS row of its a factor (-1: bias) and of its delta (-1: not NN)
        int kind[NSL];               // 0 NN weight/bias, 1 w1, 2 w2, 3 w3, 4 D0, -1 padding / unused slot
    };
    static __device__ __forceinline__ void init(Ctx& c, real* th_lds, real* scratch, real*, int, const ModelConsts& mc, int r, const real* = nullptr) {
        c.th = th_lds;
        c.nn = th_lds + mc.nn_offset;
        c.urow = scratch; c.lrow = scratch + NPT + 2; c.A = scratch + 2 * NPT + 4; c.Dt = c.A + RA * G;
        c.r = r; c.n = mc.n_state; c.so = mc.stencil_offset; c.d0o = mc.d0_offset; c.nno = mc.nn_offset;
        c.w1 = th_lds[c.so]; c.w2 = th_lds[c.so + 1]; c.w3 = th_lds[c.so + 2]; c.D0 = th_lds[c.d0o];
        for (int m = 0; m < NSL; ++m) {
            const int p = r + G * m;
            c.kind[m] = -1; c.a_row[m] = -1; c.d_row[m] = -1;
            if (p >= mc.n_param) continue;
            if (p == c.so) c.kind[m] = 1;
            else if (p == c.so + 1) c.kind[m] = 2;
            else if (p == c.so + 2) c.kind[m] = 3;
            else if (p == c.d0o) c.kind[m] = 4;
            else if (p >= c.nno && p < c.nno + Net::nparam) {
                const int q = p - c.nno;
                static_for<0, L>([&](auto lc) {
                    constexpr int l = lc;
                    constexpr int in = Net::dim(l), out = Net::dim(l + 1);
                    if (q >= Net::off(l) && q < Net::off(l) + in * out + out) {
                        const int e = q - Net::off(l);
                        c.kind[m] = 0;
                        if (e < in * out) { c.d_row[m] = d_off(l) + e % out; c.a_row[m] = a_off(l) + e / out; }
                        else { c.d_row[m] = d_off(l) + (e - in * out); c.a_row[m] = -1; }
                    }
                });
            }
        }
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const real* u, real* du) {
        const int n = c.n;
        __syncthreads();
        static_for<0, PPL>([&](auto cc) { const int i = cc * G + c.r; if (i < n) c.urow[i] = u[cc]; });
        __syncthreads();
        for (int cc = 0; cc < PPL; ++cc) {
            const int i = cc * G + c.r;
            real out = 0.0;
            if (i < n) {
                const int im = wrap_prev(i, n), ip = wrap_next(i, n);
                typename Mlp::Cache cache;
                real y[1];
                const real ui = c.urow[i];
                Mlp::forward(c.nn, 0, &ui, cache, y);
                const real cnn = c.w1 * c.urow[im] + c.w2 * ui + c.w3 * c.urow[ip];
                out = y[0] + c.D0 * cnn;
            }
            du[cc] = out;
        }
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const real* u, const real* lam, real* dlam, real* g) {
        const int n = c.n;
        __syncthreads();
        static_for<0, PPL>([&](auto cc) {
            const int i = cc * G + c.r;
            if (i < n) { c.urow[i] = u[cc]; c.lrow[i] = lam[cc]; }
        });
        __syncthreads();
        // ARITH-SPEC: fused chains over blocks of 256 consecutive points, block sums added left to right
        static_assert(256 % G == 0, "a 256-point block is a whole number of tiles");
        real acc[NSL], tot[NSL];
        static_for<0, NSL>([&](auto m) { acc[m] = 0.0; tot[m] = 0.0; });
        for (int cc = 0; cc < PPL; ++cc) {  // tile cc = points cc*G .. cc*G + G-1 (ascending)
            const int i = cc * G + c.r;
            real gxi = 0.0;
            if (i < n) {
                typename Mlp::Cache cache;
                real y[1], gx[1];
                const real ui = c.urow[i], li = c.lrow[i];
                Mlp::forward(c.nn, 0, &ui, cache, y);
                static_for<0, L>([&](auto lc) {
                    constexpr int l = lc;
                    static_for<0, Net::dim(l)>([&](auto k) { c.A[c.r * RA + (a_off(l) + k)] = cache.a[l][k]; });
                });
                Mlp::template vjp_sink<false>(c.nn, 0, cache, &li, gx, (real*)nullptr,
                                              [&](int l, int m, real d) { c.Dt[c.r * RD + (d_off_rt(l) + m)] = d; });
                gxi = gx[0];
            }
            // transpose of the periodic stencil (the oracle's expression)
            if (i < n) {
                const int im = wrap_prev(i, n), ip = wrap_next(i, n);
                dlam[cc] = gxi + c.D0 * (c.w1 * c.lrow[ip] + c.w2 * c.lrow[i] + c.w3 * c.lrow[im]);
            } else {
                dlam[cc] = 0.0;
            }
            __syncthreads();
            if constexpr (WANT_PARAM) {
                const int npts = (n - cc * G) < G ? (n - cc * G) : G;  // points of this tile
                static_for<0, NSL>([&](auto mc) {
                    constexpr int m = mc;
                    if (c.kind[m] == 0) {
                        // tiles are [point][row] (row stride odd): the owners of different parameters read different
                        // banks of the same point's row block -- no bank conflicts in this (dominant) loop
                        const real* dr = c.Dt + c.d_row[m];
                        if (c.a_row[m] >= 0) {
                            const real* ar = c.A + c.a_row[m];
                            for (int q = 0; q < npts; ++q) acc[m] = rfma(dr[q * RD], ar[q * RA], acc[m]);  // ARITH-SPEC: fused chain
                        } else {
                            for (int q = 0; q < npts; ++q) acc[m] += dr[q * RD];
                        }
                    }
                });
            }
            __syncthreads();
            if constexpr (WANT_PARAM) {
                if (((cc + 1) * G) % 256 == 0 || cc == PPL - 1) {  // end of a 256-point block (or of the grid)
                    const bool firstb = (cc * G) < 256;
                    static_for<0, NSL>([&](auto m) { tot[m] = firstb ? acc[m] : tot[m] + acc[m]; acc[m] = 0.0; });
                }
            }
        }
        if constexpr (WANT_PARAM) {
            static_for<0, NSL>([&](auto m) { acc[m] = tot[m]; });
            // stencil weights and D0: fused sums over blocks of 256 points (oracle order), by the owning lanes
            static_for<0, NSL>([&](auto mc) {
                constexpr int m = mc;
                const int kd = c.kind[m];
                if (kd >= 1) {
                    real s = 0.0, st = 0.0;
                    for (int i = 0; i < n; ++i) {
                        if (i > 0 && i % 256 == 0) { st = i == 256 ? s : st + s; s = 0.0; }
                        const int im = wrap_prev(i, n), ip = wrap_next(i, n);
                        if (kd == 1) s = rfma(c.lrow[i], c.urow[im], s);
                        else if (kd == 2) s = rfma(c.lrow[i], c.urow[i], s);
                        else if (kd == 3) s = rfma(c.lrow[i], c.urow[ip], s);
                        else s = rfma(c.lrow[i], c.w1 * c.urow[im] + c.w2 * c.urow[i] + c.w3 * c.urow[ip], s);
                    }
                    s = n > 256 ? st + s : s;
                    acc[m] = kd == 4 ? s : c.D0 * s;
                }
            });
            static_for<0, NSL>([&](auto m) { g[m] = acc[m]; });
        }
        __syncthreads();
    }
    static constexpr int d_off_rt(int l) { return d_off(l); }
    static __device__ __forceinline__ int slot_index(const ModelConsts& mc, int r, int s) {
        const int p = r + G * s;
        if (p >= mc.n_param) return -1;
        if (p == mc.stencil_offset + 3) return -1;  // the unused conv bias (Fisher-KPP-CNN.jl:100-109): gradient stays 0
        return p;
    }
};


#ifndef UDE_F32  // (FP64 matrix-core kernel: Float64 only)
// ---------------------------------------------------------------------------------------------
// nn_ode on LARGE grids (BASELINE configs[3]: 1024 points): one trajectory per block of 4 wavefronts, the pointwise
// network AND its parameter contraction on the FP64 matrix cores.
//
// v_mfma_f64_16x16x4 computes D = C + A(16x4) B(4x16) as d = fma(a_k, b_k, d) for k = 0..3 IN ASCENDING ORDER
// (tools/probe/mfma_order_probe.hip: 51200/51200 results bit-identical to that chain) -- exactly ARITH-SPEC's fma chain
// in ascending index order started from 0.  So for a tile of 16 grid points (columns):
//   forward   z_l = W_l a_l          A = W_l (neurons x inputs, registers, zero padded), B = a_l, chain over inputs
//   backward  g   = W_l^T delta_l    A = W_l^T, B = delta_l,                                  chain over outputs
//   contraction dW_l += delta_l [a_l ; 1]^T   A = delta_l, B = [a_l ; 1], chain over the GRID POINTS (fused, ascending)
// and the output layout of one product (lane (k, j), register r <-> row k + 4r, column j) IS the B-operand layout of
// the next one (k-step s <-> register s): activations and deltas never leave the registers between layers.  Only the
// contraction needs points along K, i.e. a transpose, through a wave-private [point][row] LDS tile.  Zero padding is
// exact: fma(0, b, acc) == acc for finite b.
// Wavefront w owns the 256 consecutive points 256w .. 256w+255 (16 column tiles), one ARITH-SPEC block of the
// parameter sums; the four block sums meet in LDS and are added left to right by the parameter's final owner
// (theta index p = r + 256 m: the Driver's register slots).  The state itself is distributed 4 points per lane.
// ---------------------------------------------------------------------------------------------
// NWV_: wavefronts per PDE.  The adjoint (and everything that sums parameter cotangents over blocks of 256 points: ARITH-SPEC) runs
// with four; the FORWARD solve, which has no such sums, runs with eight (FwdModel below): two wavefronts per SIMD instead of one on
// the CU that owns the PDE -- the 256 PDEs of configs[3] are one block per CU whatever the block size is.
template <class Net, int NWV_ = 4>
struct KppUdeW : LinearTheta {
    static constexpr bool RECOMPUTE_OK = true;   // checkpointed adjoint available (AdjSys::RECOMPUTE)
    static constexpr bool PER_MEMBER_THETA = true;   // (theta is only read by init(): operand tables and coefficients live in registers afterwards)
    static constexpr int NWV = NWV_, G = 64 * NWV, PPL = 1024 / G, TP = 64, BLK = TP * PPL, NTILE = BLK / 16;
    static_assert(NWV == 4 || NWV == 8, "1024 points on four or eight wavefronts");
    static constexpr int FWD_BLOCKS = NWV == 8 ? 2 : 1;
    using FwdModel = KppUdeW<Net, 8>;            // make_launch: forward / rhs kernels of the four-wavefront model
    static constexpr bool DADJ_K_FROM_DENSE = true;  // (k and kbar of a 1024-point state do not both fit next to the tiles)
    static __host__ __device__ constexpr int point(int c, int r) { return (r >> 6) * BLK + c * TP + (r & 63); }
    static_assert(Net::dim(0) == 1 && Net::dim(Net::L) == 1, "pointwise reaction network R -> R");
    static constexpr int NS = PPL;
    static constexpr int NP = Net::nparam + 5;
    static constexpr int NSL = (NP + G - 1) / G;
    static constexpr bool STATE_DISTRIBUTED = true;
    static constexpr int L = Net::L;
    static constexpr int NPT = G * PPL;
    // a run-time shape (Net::RT, round 5: NetCfgRt<1, 1, 4, 16> -- any reaction chain 1 -> a -> b -> c -> 1 with tanh hidden layers of width
    // <= 16 on the large grids): the compile-time shape is the padded one, init() builds the operand tables from the true chain (zeros
    // beyond its widths: a padded unit has z = 0, a = tanh 0 = 0, delta = 0), the block sums go to the true chain's theta indices.
    // The hidden activations are tanh by contract (udecore.hip checks the descriptor).
    static constexpr int hact(int l) { return Net::RT ? (l + 1 < Net::L ? ACT_TANH : ACT_IDENTITY) : Net::act(l); }
    static constexpr bool acts_ok() {  // the reverse sweep rebuilds act' from the activation VALUE (tanh only)
        for (int l = 0; l + 1 < L; ++l)
            if (hact(l) != ACT_TANH) return false;
        return hact(L - 1) == ACT_IDENTITY;
    }
    static_assert(acts_ok(), "KppUdeW: tanh hidden layers (tanh(0) == 0 keeps the padding rows at zero), linear output");
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) double lds_t;
    static constexpr int MT(int n) { return (n + 15) / 16; }  // 16-row output tiles
    static constexpr int KS(int n) { return (n + 3) / 4; }    // k-steps of 4
    static constexpr int MAXR = 4 * MT(Net::maxdim());         // registers of one activation / delta vector (D layout)
    // operand tables (per lane): forward A_l[mt][s], backward AT_l[mt][s], biases in D layout
    static constexpr int af_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += MT(Net::dim(i + 1)) * KS(Net::dim(i)); return o; }
    static constexpr int ab_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += MT(Net::dim(i)) * KS(Net::dim(i + 1)); return o; }
    static constexpr int bs_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += 4 * MT(Net::dim(i + 1)); return o; }
    static constexpr int NAF = af_off(L), NAB = ab_off(L), NBS = bs_off(L);
    // contraction: LDS tile [16 points][rows], rows = a_0 | a_1 .. a_{L-1} | delta_0 .. delta_{L-1} | 1
    static constexpr int a_row(int l) { int o = 0; for (int i = 0; i < l; ++i) o += Net::dim(i); return o; }
    static constexpr int ROWS_A = a_row(L);
    static constexpr int d_row(int l) { int o = ROWS_A; for (int i = 0; i < l; ++i) o += Net::dim(i + 1); return o; }
    static constexpr int ONE_COL = d_row(L);
    static constexpr int RS = (ONE_COL + 1) | 1;  // odd row stride
    static constexpr int TILE = 16 * RS;
    static constexpr int cmt(int l) { return MT(Net::dim(l + 1)); }      // contraction output tiles: delta rows ...
    static constexpr int cnt_(int l) { return MT(Net::dim(l) + 1); }     // ... x [a ; 1] columns
    static constexpr int acc_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += cmt(i) * cnt_(i); return o; }
    static constexpr int NACC = acc_off(L);
    static constexpr int NPP = (NP + 1) & ~1;  // block-sum row (aliases the tile once the column tiles are consumed)
    static_assert(NPP <= TILE, "block sums must fit the tile they alias");
    static constexpr int SCRATCH = 3 * (NPT + 2) + NWV * TILE;  // u, lambda, result rows + tiles
    static constexpr int SCRATCH_FWD = 3 * (NPT + 2);           // forward / rhs kernels: the rows only (the tiles belong to the parameter contraction)
    struct Ctx {
        double af[NAF], ab[NAB], bs[NBS];
        lds_t *urow, *lrow, *orow, *tile, *part;
        double w1, w2, w3, D0;
        int r, lane, w, l16, kq, n, so, d0o, nno;
        int rdim[Net::RT ? Net::L + 1 : 1], roff[Net::RT ? Net::L : 1], np;   // run-time shape: true widths, theta offsets of the layers; parameter count
    };
    static __device__ __forceinline__ void init(Ctx& c, double* th_lds, double* scratch, double*, int, const ModelConsts& mc, int r, const double* = nullptr) {
        const double* th = th_lds;   // (a generic pointer: the block's LDS copy, or -- UDE_PT_THETA -- the member's own column in HBM)
        lds_t* sc = (lds_t*)scratch;
        c.urow = sc; c.lrow = sc + NPT + 2; c.orow = sc + 2 * (NPT + 2);
        c.part = sc + 3 * (NPT + 2);  // [NWV][TILE]; wavefront w's tile = its block-sum row afterwards
        c.r = r; c.lane = r & 63; c.w = r >> 6; c.l16 = c.lane & 15; c.kq = c.lane >> 4;
        c.tile = c.part + c.w * TILE;
        c.n = mc.n_state; c.so = mc.stencil_offset; c.d0o = mc.d0_offset; c.nno = mc.nn_offset;
        c.w1 = th[c.so]; c.w2 = th[c.so + 1]; c.w3 = th[c.so + 2]; c.D0 = th[c.d0o];
        const double* nn = th + mc.nn_offset;
        c.np = mc.n_param;
        if constexpr (Net::RT) {
            int o = 0;
            static_for<0, L>([&](auto lc) {
                constexpr int l = lc;
                c.rdim[l] = mc.dims[l]; c.roff[l] = o;
                o += mc.dims[l] * mc.dims[l + 1] + mc.dims[l + 1];
            });
            c.rdim[L] = mc.dims[L];
        }
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            constexpr int inp = Net::dim(l), outp = Net::dim(l + 1);   // (padded) compile-time widths: the tile structure
            int in = inp, out = outp, off = Net::off(l);               // the widths and the offset theta is read through
            if constexpr (Net::RT) { in = c.rdim[l]; out = c.rdim[l + 1]; off = c.roff[l]; }
            static_for<0, MT(outp)>([&](auto m) {
                static_for<0, KS(inp)>([&](auto s) {  // forward: A[i][k] = W_l[i][k]
                    const int i = c.l16 + 16 * decltype(m)::value, k = 4 * decltype(s)::value + c.kq;
                    c.af[af_off(l) + decltype(m)::value * KS(inp) + decltype(s)::value] = (i < out && k < in) ? (double)nn[off + i + k * out] : 0.0;
                });
                static_for<0, 4>([&](auto rr) {
                    const int i = c.kq + 4 * decltype(rr)::value + 16 * decltype(m)::value;
                    c.bs[bs_off(l) + 4 * decltype(m)::value + decltype(rr)::value] = i < out ? (double)nn[off + in * out + i] : 0.0;
                });
            });
            static_for<0, MT(inp)>([&](auto m) {
                static_for<0, KS(outp)>([&](auto s) {  // backward: A[i][k] = W_l[k][i]
                    const int i = c.l16 + 16 * decltype(m)::value, k = 4 * decltype(s)::value + c.kq;
                    c.ab[ab_off(l) + decltype(m)::value * KS(outp) + decltype(s)::value] = (i < in && k < out) ? (double)nn[off + k + i * out] : 0.0;
                });
            });
        });
    }
    static __device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // forward pass of one column tile.  b0: this lane's B operand of layer 0 (u of point l16 on lanes k = 0, else 0).
    // act[l] (l >= 1): output of layer l-1 = input of layer l, D layout (register m*4 + r <-> neuron kq + 4r + 16m).
    // Returns the network output y (row 0: meaningful on lanes with kq == 0).
    static __device__ __forceinline__ double forward_tile(const Ctx& c, double b0, double (&act)[L][MAXR]) {
        double y = 0.0;
        static_for<0, L>([&](auto lc) {
            constexpr int l = lc;
            constexpr int in = Net::dim(l), out = Net::dim(l + 1);
            static_for<0, MT(out)>([&](auto m) {
                v4d d = v4d{0.0, 0.0, 0.0, 0.0};
                static_for<0, KS(in)>([&](auto s) {
                    const double bop = l == 0 ? b0 : act[l][decltype(s)::value];
                    d = mfma(c.af[af_off(l) + decltype(m)::value * KS(in) + decltype(s)::value], bop, d);
                });
                static_for<0, 4>([&](auto rr) {
                    constexpr int r0 = 4 * decltype(rr)::value + 16 * decltype(m)::value;  // first neuron of this register
                    if constexpr (r0 < out) {
                        const double z = d[decltype(rr)::value] + c.bs[bs_off(l) + 4 * decltype(m)::value + decltype(rr)::value];
                        if constexpr (l + 1 < L) act[l + 1][4 * decltype(m)::value + decltype(rr)::value] = act_fwd<hact(l)>(z);
                        else if constexpr (r0 == 0) y = z;
                    } else if constexpr (l + 1 < L) {
                        act[l + 1][4 * decltype(m)::value + decltype(rr)::value] = 0.0;
                    }
                });
            });
        });
        return y;
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const double* u, double* du) {
        const int n = c.n;
        __syncthreads();
        static_for<0, PPL>([&](auto cc) { const int i = point(cc, c.r); if (i < n) c.urow[i] = u[cc]; });
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < NTILE; ++t) {
            const int i = c.w * BLK + 16 * t + c.l16;  // this lane's column
            const bool on = c.kq == 0 && i < n;
            const double ui = on ? c.urow[i] : 0.0;
            double act[L][MAXR];
            const double y = forward_tile(c, ui, act);
            if (on) {
                const int im = wrap_prev(i, n), ip = wrap_next(i, n);
                const double cnn = c.w1 * c.urow[im] + c.w2 * ui + c.w3 * c.urow[ip];
                c.orow[i] = y + c.D0 * cnn;
            }
        }
        __syncthreads();  // (columns were produced by other lanes than the ones that own the points)
        static_for<0, PPL>([&](auto cc) { const int i = point(cc, c.r); du[cc] = i < n ? c.orow[i] : 0.0; });
    }
    static __device__ __forceinline__ void wave_sync() {
        // the tile is private to the wavefront (lock-step lanes): only the LDS queue has to drain
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const double* u, const double* lam, double* dlam, double* g) {
        const int n = c.n;
        __syncthreads();
        static_for<0, PPL>([&](auto cc) {
            const int i = point(cc, c.r);
            if (i < n) { c.urow[i] = u[cc]; c.lrow[i] = lam[cc]; }
        });
        __syncthreads();
        v4d acc[NACC];  // output tiles of the wavefront's block sums (the fused chains run on across its 16 column tiles)
        static_for<0, NACC>([&](auto m) { acc[m] = v4d{0.0, 0.0, 0.0, 0.0}; });
        const int l16 = c.l16, kq = c.kq;
#pragma unroll 1
        for (int t = 0; t < NTILE; ++t) {
            const int i = c.w * BLK + 16 * t + l16;  // this lane's column
            const bool on = kq == 0 && i < n;
            const double ui = on ? c.urow[i] : 0.0, li = on ? c.lrow[i] : 0.0;
            double act[L][MAXR];
            forward_tile(c, ui, act);
            if constexpr (WANT_PARAM) {
                wave_sync();  // the contraction of the previous column tile is done with the LDS tile
                lds_t* row = c.tile + l16 * RS;
                if (kq == 0) { row[0] = ui; row[ONE_COL] = 1.0; }
                static_for<1, L>([&](auto lc) {  // a_l, l >= 1 (D layout -> rows)
                    constexpr int l = lc;
                    constexpr int in = Net::dim(l);
                    static_for<0, 4 * MT(in)>([&](auto q) {
                        constexpr int r0 = 4 * (decltype(q)::value % 4) + 16 * (decltype(q)::value / 4);
                        if constexpr (r0 < in) { if (r0 + kq < in) row[a_row(l) + r0 + kq] = act[l][q]; }
                    });
                });
            }
            // reverse sweep: dcur = delta of layer l (D layout), from the linear output layer down
            double dcur[MAXR];
            dcur[0] = li * 1.0;
            double gxi = 0.0;
            static_for<0, L>([&](auto lr) {
                constexpr int l = L - 1 - lr;
                constexpr int in = Net::dim(l), out = Net::dim(l + 1);
                if constexpr (WANT_PARAM) {
                    lds_t* row = c.tile + l16 * RS;
                    static_for<0, 4 * MT(out)>([&](auto q) {
                        constexpr int r0 = 4 * (decltype(q)::value % 4) + 16 * (decltype(q)::value / 4);
                        if constexpr (r0 < out) { if (r0 + kq < out) row[d_row(l) + r0 + kq] = dcur[q]; }
                    });
                }
                double dprev[MAXR];
                static_for<0, MT(in)>([&](auto m) {
                    v4d gq = v4d{0.0, 0.0, 0.0, 0.0};
                    static_for<0, KS(out)>([&](auto s) {
                        gq = mfma(c.ab[ab_off(l) + decltype(m)::value * KS(out) + decltype(s)::value], dcur[decltype(s)::value], gq);
                    });
                    static_for<0, 4>([&](auto rr) {
                        constexpr int q = 4 * decltype(m)::value + decltype(rr)::value;
                        constexpr int r0 = 4 * decltype(rr)::value + 16 * decltype(m)::value;
                        if constexpr (l > 0) {
                            if constexpr (r0 < in) dprev[q] = gq[decltype(rr)::value] * act_bwd<hact(l - 1)>(0.0, act[l][q]);
                            else dprev[q] = 0.0;
                        } else if constexpr (q == 0) {
                            gxi = gq[0];
                        }
                    });
                });
                if constexpr (l > 0) static_for<0, 4 * MT(in)>([&](auto q) { dcur[q] = dprev[q]; });
            });
            if (on) {  // transpose of the periodic stencil (the oracle's expression)
                const int im = wrap_prev(i, n), ip = wrap_next(i, n);
                c.orow[i] = gxi + c.D0 * (c.w1 * c.lrow[ip] + c.w2 * c.lrow[i] + c.w3 * c.lrow[im]);
            }
            if constexpr (WANT_PARAM) {
                // contraction over the 16 points of the tile: 4 k-steps per output tile.  Lane (i, k) = (l16, kq):
                // A[i][k] = delta_i(point 4s + k), B[k][j] = [a ; 1]_j(point 4s + k); rows / columns beyond the layer's
                // sizes are clamped (their outputs are never written out)
                wave_sync();
                static_for<0, L>([&](auto lc) {
                    constexpr int l = lc;
                    constexpr int in = Net::dim(l), out = Net::dim(l + 1);
                    int aoff[cmt(l)], boff[cnt_(l)];
                    static_for<0, cmt(l)>([&](auto m) {
                        const int ii = l16 + 16 * decltype(m)::value;
                        aoff[m] = d_row(l) + (ii < out ? ii : out - 1);
                    });
                    static_for<0, cnt_(l)>([&](auto nn) {
                        const int j = l16 + 16 * decltype(nn)::value;
                        boff[nn] = j < in ? a_row(l) + j : ONE_COL;
                    });
                    static_for<0, 4>([&](auto s4) {
                        const lds_t* pt = c.tile + (4 * decltype(s4)::value + kq) * RS;
                        double av[cmt(l)], bv[cnt_(l)];
                        static_for<0, cmt(l)>([&](auto m) { av[m] = pt[aoff[m]]; });
                        static_for<0, cnt_(l)>([&](auto nn) { bv[nn] = pt[boff[nn]]; });
                        static_for<0, cmt(l)>([&](auto m) {
                            static_for<0, cnt_(l)>([&](auto nn) {
                                constexpr int ai = acc_off(l) + decltype(m)::value * cnt_(l) + decltype(nn)::value;
                                acc[ai] = mfma(av[m], bv[nn], acc[ai]);
                            });
                        });
                    });
                });
            }
        }
        __syncthreads();  // (columns were produced by other lanes than the ones that own the points)
        static_for<0, PPL>([&](auto cc) { const int i = point(cc, c.r); dlam[cc] = i < n ? c.orow[i] : 0.0; });
        if constexpr (WANT_PARAM) {
            // stencil weights and D0: the block's four fused sums as ONE more matrix product, row 0 of
            //   lambda(1 x 256) . [u_{i-1}, u_i, u_{i+1}, w1 u_{i-1} + w2 u_i + w3 u_{i+1}](256 x 4)
            v4d sacc = v4d{0.0, 0.0, 0.0, 0.0};
            {
                const int b0 = c.w * BLK;
#pragma unroll 4
                for (int s4 = 0; s4 < BLK / 4; ++s4) {
                    // (branch-free: a point beyond the grid reads point 0 and contributes zeros -- as an `if (i < n)` body this was an
                    //  EXEC-masked region per k-step, 64 of them per evaluation and wavefront)
                    const int i = b0 + 4 * s4 + kq;
                    const bool in = i < n;
                    const int ic = in ? i : 0;
                    const int im = wrap_prev(ic, n), ip = wrap_next(ic, n);
                    const double um = c.urow[im], u0 = c.urow[ic], up = c.urow[ip];
                    const double comb = c.w1 * um + c.w2 * u0 + c.w3 * up;
                    const double bsel = l16 == 0 ? um : l16 == 1 ? u0 : l16 == 2 ? up : l16 == 3 ? comb : 0.0;
                    const double av = in ? c.lrow[ic] : 0.0, bv = in ? bsel : 0.0;
                    sacc = mfma(av, bv, sacc);
                }
            }
            // block sums -> this wavefront's row (aliases its tile).  Output element: row = lane / 16 + 4 r, column = lane % 16
            wave_sync();
            lds_t* prow = c.tile;
            static_for<0, L>([&](auto lc) {
                constexpr int l = lc;
                constexpr int in = Net::dim(l), out = Net::dim(l + 1);
                static_for<0, cmt(l)>([&](auto m) {
                    static_for<0, cnt_(l)>([&](auto nn) {
                        constexpr int ai = acc_off(l) + decltype(m)::value * cnt_(l) + decltype(nn)::value;
                        const int j = l16 + 16 * decltype(nn)::value;
                        static_for<0, 4>([&](auto rr) {
                            const int ii = kq + 4 * decltype(rr)::value + 16 * decltype(m)::value;
                            if constexpr (Net::RT) {   // (the bias column of the padded shape is column `in`; the true chain's weights are the columns j < its width)
                                const int inr = c.rdim[l], outr = c.rdim[l + 1];
                                if (ii < outr && (j < inr || j == in)) prow[c.nno + c.roff[l] + (j < inr ? ii + j * outr : inr * outr + ii)] = acc[ai][decltype(rr)::value];
                            } else
                            if (ii < out && j <= in) prow[c.nno + Net::off(l) + (j < in ? ii + j * out : in * out + ii)] = acc[ai][decltype(rr)::value];
                        });
                    });
                });
            });
            if (kq == 0 && l16 < 3) prow[c.so + l16] = sacc[0];   // row 0 lives in register 0 of lanes 0..15
            if (kq == 0 && l16 == 3) { prow[c.d0o] = sacc[0]; prow[c.so + 3] = 0.0; }
            __syncthreads();
            static_for<0, NSL>([&](auto s) {
                const int p = c.r + G * decltype(s)::value;
                double v = 0.0;
                if (p < (Net::RT ? c.np : NP)) {
                    v = c.part[p];
                    static_for<1, NWV>([&](auto w) { v += c.part[decltype(w)::value * TILE + p]; });
                    if (p >= c.so && p < c.so + 3) v = c.D0 * v;
                }
                g[s] = v;
            });
        }
        __syncthreads();
    }
    static __device__ __forceinline__ int slot_index(const ModelConsts& mc, int r, int s) {
        const int p = r + G * s;
        if (p >= mc.n_param) return -1;
        if (p == mc.stencil_offset + 3) return -1;  // the unused conv bias (Fisher-KPP-CNN.jl:100-109): gradient stays 0
        return p;
    }
};

#endif  // UDE_F32

}  // namespace ude
#include "ude_model_kpp_vec.h"   // round 6: the 1024-point network on the vector unit, packed matrix-core contraction
