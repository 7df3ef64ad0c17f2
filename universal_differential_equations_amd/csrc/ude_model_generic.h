// ude_model_generic.h -- the runtime-shape fallback (included by ude_models.h; Float64 and, for the LV kind, Float32 translation units).
//
// The reference's networks are script variables: `U = Lux.Chain(Dense(2,5,rbf), ...)` (LotkaVolterra/scenario_1.jl:62-64),
// `ann = FastChain(FastDense(3,64,tanh), ...)` (SEIR_exposure/seir_exposure.jl:114), `n_weights` of
// FisherKPP/Fisher-KPP-CNN-Small.jl:88 -- any chain is accepted upstream.  The compiled instances cover the shapes the scripts
// ship; THIS model covers every other ude_model_desc of the replicated-state kinds (UDE_KIND_LV_UDE, UDE_KIND_SEIR_UDE,
// UDE_KIND_SEIR_NODE): up to 8 Dense layers of width <= 64, activations identity / tanh / rbf / relu, the layer sizes and
// activations read from the kernel arguments (ModelConsts::dims / act).
//
// One wavefront per trajectory; lane j = neuron j of every layer; weights are read from HBM (L2-resident: every trajectory
// reads the same theta) -- there is no LDS copy whose size would depend on the shape; the activations of a layer cross lanes
// through rows of the wavefront's LDS stage storage.  Arithmetic = oracle/ude_oracle_impl.h: mlp_forward / mlp_vjp_acc / wide_dot
// (ARITH-SPEC): dots of fewer than 64 terms are one fma chain from 0 in ascending order; 64-term dots with >= 16 results are
// four 16-term chains added left to right; 64-term dots with fewer than 16 results are rounded products summed by the
// adjacent-pair tree of the wavefront.  Per trajectory every number is therefore bit-identical to the oracle.
// The parameter cotangent is DEFERRED exactly as in SeirNode: every adjoint stage leaves its factors (a_l and delta_l rows) in
// LDS and the RK-weighted sums of all slots are formed at the end of the step, fused with error norm and candidate mu (mu in
// HBM, two columns that swap on acceptance).  This is a fallback: it is written for generality, not for speed.
#pragma once

namespace ude {

enum { GK_LV_UDE = 1, GK_SEIR_UDE = 3, GK_SEIR_NODE = 6 };  // = UDE_KIND_* of include/udecore.h (ude_model_desc.kind)

// LMAX_: the deepest chain an instance takes.  The stage storage of a wavefront is 8 x (2 LMAX x 64 + 16) doubles of LDS: 67 KB for
// LMAX = 8 (two wavefronts per CU), 34 KB for LMAX = 4 (four: every SIMD busy) -- chains of <= 4 layers, i.e. every network of the
// reference's scripts, run on the LMAX = 4 instance
template <int NSTATE, int LMAX_ = 8>
struct GenericUde {
    static constexpr int H = 64, LMAX = LMAX_;
    static constexpr int NS = NSTATE;
    static constexpr int NSLW = LMAX * (H + 1);   // weight + bias slots of lane j: (in_l + 1) per layer, layer after layer
    static constexpr int NSL = NSLW + 2;           // + the two (optional) trainable diagonal coefficients of the LV kind (lane 0)
    static constexpr bool STATE_DISTRIBUTED = false;
    static constexpr bool THETA_GLOBAL = true, FUSED_ACC = true, SLOTS_GLOBAL = true, CPL = true, DEFERRED = true;
    static constexpr bool DADJ_K_FROM_DENSE = false, COMPACT_STAGES = true;
    // discretise-then-optimise sweep (round 4): the reverse sweep's VJPs leave their factors in the stage storage like the adjoint's
    // evaluations do, and dadj_flush adds them to the accumulators (HBM column) in VJP order -- `acc += delta * a`, the oracle's
    // discrete_sweep sequence -- whenever the NSTC stage slots are full and at the end of the sweep
    static constexpr bool DADJ_DEFERRED = true;
    static constexpr int NSTG = 10, NSTC = 8;      // tableau stages, stored (compacted) stages
    static constexpr int RX = 16;                  // short row per stage: x_0..x_6 (the network input) | u0 u1 lam0 lam1 (LV diagonal slots)
    static constexpr int STG = 2 * LMAX * H + RX;  // doubles of one stored stage: A rows (a_0 .. a_{L-1}), delta rows (delta_1 .. delta_L), short row
    static constexpr int WORK = (LMAX + 1) * H + LMAX * H + 2 * H;  // forward working rows: a_0..a_L, dphi_0..dphi_{L-1}, spare
    static constexpr int SCRATCH = NSTC * STG + WORK;
    static constexpr int SCRATCH_FWD = WORK;
    static constexpr int FWD_BLOCKS = 2;           // forward / rhs kernels compiled for two wavefronts per SIMD (256 registers; their LDS share is WORK only)
    typedef __attribute__((address_space(3))) real lds_t;
    struct Ctx {
        const real* nn;     // theta + nn_offset (HBM)
        lds_t* work;          // a rows [l * H + lane], then dphi rows
        lds_t* fac;           // stage storage of this wavefront
        const ModelConsts* mc;  // dims / act of the chain: kernel arguments, read with wave-uniform indices (scalar loads)
        int L, kind, nnp;       // layers, kind, number of NN parameters
        real lin[2], lin_on[2];
        real mu_c, sg, F, b0, ga, dd, la;   // SEIR constants
        int j, r;
    };
    static __host__ __device__ constexpr int theta_lds(int) { return 0; }
    static __device__ __forceinline__ void stage_theta(real*, const real*, int, int, int) {}
    static __device__ __forceinline__ void init(Ctx& c, real* theta, real* scratch, real*, int, const ModelConsts& mc, int r,
                                                const real* theta_g) {
        (void)theta;
        c.j = r & 63; c.r = r;
        c.nn = theta_g + mc.nn_offset;
        c.work = (lds_t*)scratch;               // (the working rows first: all the forward / rhs kernels need of the scratch)
        c.fac = (lds_t*)scratch + WORK;
        c.mc = &mc;
        c.L = mc.n_layers; c.kind = mc.kind;
        int o = 0;
        for (int l = 0; l < mc.n_layers; ++l) o += mc.dims[l] * mc.dims[l + 1] + mc.dims[l + 1];
        c.nnp = o;
        for (int i = 0; i < 2; ++i) {
            c.lin[i] = mc.lin_idx[i] >= 0 ? (real)mc.lin_sign[i] * theta_g[mc.lin_idx[i]] : (real)mc.lin_const[i];
            c.lin_on[i] = (mc.lin_idx[i] >= 0 && r == 0) ? mc.lin_sign[i] : real(0);
        }
        c.F = (real)mc.consts[0]; c.b0 = (real)mc.consts[1]; c.mu_c = (real)mc.consts[4]; c.sg = (real)mc.consts[5]; c.ga = (real)mc.consts[6];
        c.dd = (real)mc.consts[7]; c.la = (real)mc.consts[8];
    }
    static __device__ __forceinline__ real actf(int a, real z) {
        return a == ACT_TANH ? rtanh(z) : a == ACT_RBF ? rexp(-(z * z)) : a == ACT_RELU ? (z > real(0) ? z : real(0)) : z;
    }
    static __device__ __forceinline__ real dactf(int a, real z, real av) {
        return a == ACT_TANH ? rfma(-av, av, real(1)) : a == ACT_RBF ? (real(-2) * z) * av : a == ACT_RELU ? (z > real(0) ? real(1) : real(0)) : real(1);
    }
    // ---- wide_dot (oracle): result r of `nres` results over n terms; term i = w[i * ws] * x_i ----
    // x lives in an LDS row (lane i wrote x_i); lane j owns result j.  wbase(j): address of the lane's term 0, ws: term stride.
    static __device__ __forceinline__ real dot_lane(const real* w, int ws, const lds_t* x, int n) {
        // The weights come from HBM / L2 (theta is shared by all trajectories, no LDS copy): what bounds a dot is the number of
        // dependent memory round trips, not the fma's.  Loads are therefore issued in batches of GEN_BATCH before the first fma of
        // the batch (the chain itself stays strictly ascending): 4 round trips per 64-term dot with 16, one with 64.
#ifndef GEN_BATCH
#define GEN_BATCH 16
#endif
        if (n < 64) {  // one chain from 0, ascending
            real acc = real(0);
            int i = 0;
#pragma unroll 1
            for (; i + GEN_BATCH <= n; i += GEN_BATCH) {
                real wv[GEN_BATCH];
#pragma unroll
                for (int u = 0; u < GEN_BATCH; ++u) wv[u] = w[(size_t)(i + u) * ws];
#pragma unroll
                for (int u = 0; u < GEN_BATCH; ++u) acc = rfma(wv[u], x[i + u], acc);
            }
#pragma unroll 4
            for (; i < n; ++i) acc = rfma(w[(size_t)i * ws], x[i], acc);
            return acc;
        }
        real tot = real(0);  // 64 terms: four 16-term chains, block sums left to right
#if GEN_BATCH >= 64
        real wv[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) wv[i] = w[(size_t)i * ws];
#pragma unroll
        for (int b = 0; b < 64; b += 16) {
            real acc = real(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = rfma(wv[b + i], x[b + i], acc);
            tot = b == 0 ? acc : tot + acc;
        }
#else
#pragma unroll 1
        for (int b = 0; b < 64; b += 16) {
            real acc = real(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = rfma(w[(size_t)(b + i) * ws], x[b + i], acc);
            tot = b == 0 ? acc : tot + acc;
        }
#endif
        return tot;
    }
    // forward chain: x on lanes 0..dims[0]-1 (others 0).  A rows go to `arow` (row l at arow[l * H]), dphi rows to c.work.
    // Returns this lane's a_L (valid on lanes < dims[L]).
    static __device__ __forceinline__ real forward(const Ctx& c, real xlane, lds_t* arow, bool want_dphi) {
        const int j = c.j;
        lds_t* dphi = c.work + (LMAX + 1) * H;
        real a = xlane;
        arow[j] = a;
        int off = 0;
#pragma unroll 1
        for (int l = 0; l < c.L; ++l) {
            const int in = c.mc->dims[l], out = c.mc->dims[l + 1], actl = c.mc->act[l];
            const real* W = c.nn + off;
            off += in * out + out;
            const int jj = j < out ? j : out - 1;   // (lanes beyond the layer compute a discarded copy of the last neuron: in-bounds reads)
            const lds_t* x = arow + l * H;
            real z;
            if ((in == 64 || in == 32) && out < 16) {
                // 32 or 64 terms reduced to a few replicated scalars (wide-dot rule: fewer than 16 results): rounded products,
                // adjacent-pair tree over the wavefront (lane i = term i; with 32 terms the upper half contributes -0.0, the identity
                // of IEEE addition: x + (-0.0) == x for every x); 16 .. 64 results: the lane-parallel chains of dot_lane
                z = real(0);
                const int ji = j < in ? j : in - 1;
#pragma unroll 1
                for (int rr = 0; rr < out; ++rr) {
                    const real pr = W[rr + (size_t)ji * out] * a;
                    const real s = wave_tree_sum(j < in ? pr : real(-0.0));
                    z = (j == rr) ? s : z;
                }
            } else {
                z = dot_lane(W + jj, out, x, in);
            }
            z += W[(size_t)in * out + jj];
            const real av = actf(actl, z);
            a = j < out ? av : real(0);
            // the next layer's input row: row l + 1 of the same storage, except that a_L (never a factor) goes to the spare row
            lds_t* nxt = (l + 1 < c.L) ? arow + (l + 1) * H : c.work + LMAX * H;
            nxt[j] = a;
            if (want_dphi) dphi[l * H + j] = j < out ? dactf(actl, z, av) : real(0);
        }
        return a;
    }
    // reverse chain: gy on lanes 0..dims[L]-1; delta rows to `drow` (row l = delta of layer l's OUTPUT); returns this lane's input cotangent
    static __device__ __forceinline__ real backward(const Ctx& c, real gylane, lds_t* drow) {
        const int j = c.j;
        const lds_t* dphi = c.work + (LMAX + 1) * H;
        real delta = gylane;
        int off = c.nnp;
#pragma unroll 1
        for (int l = c.L - 1; l >= 0; --l) {
            const int in = c.mc->dims[l], out = c.mc->dims[l + 1];
            off -= in * out + out;
            const real* W = c.nn + off;
            delta = delta * dphi[l * H + j];
            delta = j < out ? delta : real(0);
            drow[l * H + j] = delta;
            real prev;
            if ((out == 64 || out == 32) && in < 16) {
                prev = real(0);
                const int jo = j < out ? j : out - 1;
#pragma unroll 1
                for (int k = 0; k < in; ++k) {
                    const real pr = W[jo + (size_t)k * out] * delta;
                    const real s = wave_tree_sum(j < out ? pr : real(-0.0));
                    prev = (j == k) ? s : prev;
                }
            } else {
                const int kk = j < in ? j : in - 1;
                prev = dot_lane(W + (size_t)kk * out, 1, drow + l * H, out);
            }
            delta = j < in ? prev : real(0);
        }
        return delta;
    }
    // ---- kind wiring (oracle: udeo_rhs / udeo_rhs_vjp) ----
    static __device__ __forceinline__ real input_lane(const Ctx& c, const real* u) {
        const int j = c.j;
        real x = real(0);
        if (c.kind == GK_LV_UDE) {
            x = j == 0 ? u[0] : j == 1 ? u[1 % NS] : real(0);
        } else if (c.kind == GK_SEIR_UDE) {
            if constexpr (NS == 7) x = j == 0 ? u[0] / u[4] : j == 1 ? u[2] : j == 2 ? u[5] / u[4] : real(0);
        } else {
            if constexpr (NS == 7)
                x = j == 0 ? u[0] / u[4] : j == 1 ? u[1] : j == 2 ? u[2] : j == 3 ? u[3] : j == 4 ? u[4] : j == 5 ? u[5] / u[4] : j == 6 ? u[6] : real(0);
        }
        return x;
    }
    static __device__ __forceinline__ void rhs_from(const Ctx& c, const real* u, real aL, real* du) {
        if (c.kind == GK_LV_UDE) {
            du[0] = rfma(c.lin[0], u[0], readlane_real(aL, 0));
            du[1 % NS] = rfma(c.lin[1], u[1 % NS], readlane_real(aL, 1));
        } else if constexpr (NS == 7) {
            const real S = u[0], E = u[1], I = u[2], Rr = u[3], N = u[4], D = u[5];
            if (c.kind == GK_SEIR_UDE) {
                const real z = readlane_real(aL, 0);
                du[0] = -c.b0 * S * c.F / N - z - c.mu_c * S;
                du[1] = c.b0 * S * c.F / N + z - (c.sg + c.mu_c) * E;
                du[2] = c.sg * E - (c.ga + c.mu_c) * I;
                du[3] = c.ga * I - c.mu_c * Rr;
                du[4] = -c.mu_c * N;
                du[5] = c.dd * c.ga * I - c.la * D;
                du[6] = c.sg * E;
            } else {
                du[0] = readlane_real(aL, 0); du[1] = readlane_real(aL, 1); du[2] = readlane_real(aL, 2); du[3] = readlane_real(aL, 3);
                du[4] = -c.mu_c * N;
                du[5] = readlane_real(aL, 4);
                du[6] = c.sg * E;
            }
        }
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const real* u, real* du) {
        const real aL = forward(c, input_lane(c, u), c.work, false);
        rhs_from(c, u, aL, du);
    }
    // reverse sweep of one evaluation: factors into stage slot q (A rows, delta rows, short row), state cotangent out
    static __device__ __forceinline__ void sweep(const Ctx& c, const real* u, const real* lam, real* dlam, int q) {
        const int j = c.j;
        lds_t* st = c.fac + q * STG;
        const real xl = input_lane(c, u);
        forward(c, xl, st, true);
        real gy = real(0);
        if (c.kind == GK_LV_UDE) gy = j == 0 ? lam[0] : j == 1 ? lam[1 % NS] : real(0);
        else if constexpr (NS == 7) {
            if (c.kind == GK_SEIR_UDE) gy = j == 0 ? lam[1] - lam[0] : real(0);
            else gy = j == 0 ? lam[0] : j == 1 ? lam[1] : j == 2 ? lam[2] : j == 3 ? lam[3] : j == 4 ? lam[5] : real(0);
        }
        const real gxl = backward(c, gy, st + LMAX * H);
        // short row: the network input x_0..x_6 is row 0 of the A rows already; the LV diagonal slots need u and lambda
        real sh = real(0);
        if (c.kind == GK_LV_UDE) sh = j == 0 ? u[0] : j == 1 ? u[1 % NS] : j == 2 ? lam[0] : j == 3 ? lam[1 % NS] : real(0);
        st[2 * LMAX * H + (j < RX ? j : RX - 1)] = sh;
        if (c.kind == GK_LV_UDE) {
            dlam[0] = rfma(c.lin[0], lam[0], readlane_real(gxl, 0));
            dlam[1 % NS] = rfma(c.lin[1], lam[1 % NS], readlane_real(gxl, 1));
        } else if constexpr (NS == 7) {
            const real S = u[0], N = u[4], D = u[5];
            if (c.kind == GK_SEIR_UDE) {
                const real g0 = readlane_real(gxl, 0), g1 = readlane_real(gxl, 1), g2 = readlane_real(gxl, 2);
                const real cc = c.b0 * c.F / N;
                const real cN = c.b0 * S * c.F / (N * N);
                dlam[0] = (-cc - c.mu_c) * lam[0] + cc * lam[1] + g0 / N;
                dlam[1] = -(c.sg + c.mu_c) * lam[1] + c.sg * lam[2] + c.sg * lam[6];
                dlam[2] = -(c.ga + c.mu_c) * lam[2] + c.ga * lam[3] + c.dd * c.ga * lam[5] + g1;
                dlam[3] = -c.mu_c * lam[3];
                dlam[4] = cN * lam[0] - cN * lam[1] - c.mu_c * lam[4] - g0 * S / (N * N) - g2 * D / (N * N);
                dlam[5] = -c.la * lam[5] + g2 / N;
                dlam[6] = real(0);
            } else {
                real gx[7];
                static_for<0, 7>([&](auto m) { gx[m] = readlane_real(gxl, decltype(m)::value); });
                dlam[0] = gx[0] / N;
                dlam[1] = rfma(c.sg, lam[6], gx[1]);
                dlam[2] = gx[2];
                dlam[3] = gx[3];
                dlam[4] = ((gx[4] - gx[0] * S / (N * N)) - gx[5] * D / (N * N)) - c.mu_c * lam[4];
                dlam[5] = gx[5] / N;
                dlam[6] = gx[6];
            }
        }
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const real* u, const real* lam, real* dlam, real*) {
        static_assert(!WANT_PARAM, "deferred parameter cotangent only");
        sweep(c, u, lam, dlam, 0);
    }
    template <bool WANT_E>
    static __device__ __forceinline__ void vjp_acc(const Ctx&, const real*, const real*, real*, real*, real*, real, real) {}
    static __device__ __forceinline__ void vjp_store(const Ctx& c, const real* u, const real* lam, real* dlam, int s) {
        sweep(c, u, lam, dlam, s);
    }
    template <unsigned MASK>
    static constexpr int slot_of(int s) {
        int q = 0;
        for (int i = 0; i < s; ++i) q += (MASK >> i) & 1u;
        return q;
    }
    // every slot of this lane in slot order: body(slot, g[NST], m0) with g_s = the NEGATED cotangent of stage s, m0 = mu[slot].
    // Lanes beyond a layer's width own nothing there: their factors are stored as zeros, every one of their sums is an exact 0.
    template <int NST, unsigned MASK, class Body>
    static __device__ __forceinline__ void for_each_slot(const Ctx& c, const real* mu, int ms, Body body) {
        const int j = c.j;
        int sb = 0;
#pragma unroll 1
        for (int l = 0; l < c.L; ++l) {
            const int in = c.mc->dims[l];
            real d[NST];  // this lane's delta of layer l at every stored stage
            static_for<0, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) d[s] = c.fac[slot_of<MASK>(decltype(s)::value) * STG + (LMAX + l) * H + j];
            });
            // mu of the layer's in + 1 slots comes from HBM: a chunk of GEN_CH slots is requested while the previous chunk is worked on
            // (one slot ahead -- the round-3 form -- left a dependent HBM round trip per ~100 cycles of work: what bounded the kernel)
#ifndef GEN_CH
#define GEN_CH 8
#endif
            real mcur[GEN_CH], mnxt[GEN_CH];
            static_for<0, GEN_CH>([&](auto u) { const int kk = (int)decltype(u)::value; mcur[u] = mu[(size_t)(sb + (kk <= in ? kk : in)) * ms]; });
#pragma unroll 1
            for (int k0 = 0; k0 <= in; k0 += GEN_CH) {
                static_for<0, GEN_CH>([&](auto u) {
                    const int kk = k0 + GEN_CH + (int)decltype(u)::value;
                    mnxt[u] = mu[(size_t)(sb + (kk <= in ? kk : in)) * ms];   // (reads past the layer are clamped and discarded)
                });
                static_for<0, GEN_CH>([&](auto u) {
                    const int k = k0 + (int)decltype(u)::value;
                    if (k <= in) {   // k == in: the bias
                        real g[NST];
                        static_for<0, NST>([&](auto s) {
                            if constexpr ((MASK >> decltype(s)::value) & 1u) {
                                const real a = k < in ? c.fac[slot_of<MASK>(decltype(s)::value) * STG + l * H + k] : real(1);
                                g[s] = k < in ? -(d[s] * a) : -d[s];
                            }
                        });
                        body(sb + k, g, mcur[u]);
                    }
                });
                static_for<0, GEN_CH>([&](auto u) { mcur[u] = mnxt[u]; });
            }
            sb += in + 1;
        }
        if (c.kind == GK_LV_UDE) {  // the two trainable diagonal coefficients (lane 0): g = -((sign u_i) lam_i)
            static_for<0, 2>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                real g[NST];
                static_for<0, NST>([&](auto s) {
                    if constexpr ((MASK >> decltype(s)::value) & 1u) {
                        const lds_t* p = c.fac + slot_of<MASK>(decltype(s)::value) * STG + 2 * LMAX * H;
                        g[s] = -((c.lin_on[i] * p[i]) * p[2 + i]);
                    }
                });
                body(NSLW + i, g, mu[(size_t)(NSLW + i) * ms]);
            });
        }
    }
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ acc_t step_slots(const Ctx& c, const real* B, const real* BT, real dt, real abstol,
                                                       real reltol, const real* mu, real* mu_new, int ms) {
        static_assert(MASK & 1u, "the first stage starts the chains");
        real bb[NST], bt[NST];
        static_for<0, NST>([&](auto s) { bb[s] = uniform_real(B[s]); bt[s] = uniform_real(BT[s]); });
        acc_t ps = 0.0;
        for_each_slot<NST, MASK>(c, mu, ms, [&](int slot, const real* g, real m0) {
            real ab = bb[0] * g[0], ae = bt[0] * g[0];
            static_for<1, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) {
                    ab = rfma(bb[s], g[s], ab);
                    ae = rfma(bt[s], g[s], ae);
                }
            });
            const real m1 = rfma(dt, ab, m0);
            mu_new[(size_t)slot * ms] = m1;
            const real a0 = rabs(m0), a1 = rabs(m1);
            const real res = (dt * ae) / rfma((a0 > a1 ? a0 : a1), reltol, abstol);
            ps = afma(res, res, ps);
        });
        return ps;
    }
    template <int NST, unsigned MASK>
    static __device__ __forceinline__ void commit_slots(const Ctx& c, const real* B, real dt, real* mu, int ms) {
        static_assert(MASK & 1u, "the first stage starts the chains");
        real bb[NST];
        static_for<0, NST>([&](auto s) { bb[s] = uniform_real(B[s]); });
        for_each_slot<NST, MASK>(c, mu, ms, [&](int slot, const real* g, real m0) {
            real ab = bb[0] * g[0];
            static_for<1, NST>([&](auto s) {
                if constexpr ((MASK >> decltype(s)::value) & 1u) ab = rfma(bb[s], g[s], ab);
            });
            mu[(size_t)slot * ms] = rfma(dt, ab, m0);
        });
    }
    // discrete sweep: add the `cnt` stored VJPs (stage slots 0 .. cnt-1, in that order) to this lane's accumulators: acc += delta_s * a_s
    static __device__ __forceinline__ void dadj_flush(const Ctx& c, int cnt, real* acc, int ms) {
        const int j = c.j;
        int sb = 0;
#pragma unroll 1
        for (int l = 0; l < c.L; ++l) {
            const int in = c.mc->dims[l];
            real d[NSTC];
            static_for<0, NSTC>([&](auto s) { d[s] = c.fac[decltype(s)::value * STG + (LMAX + l) * H + j]; });
#pragma unroll 1
            for (int k = 0; k <= in; ++k) {   // k == in: the bias
                real m = acc[(size_t)(sb + k) * ms];
                static_for<0, NSTC>([&](auto s) {
                    const real a = k < in ? c.fac[decltype(s)::value * STG + l * H + k] : real(1);
                    const real g = k < in ? d[s] * a : d[s];
                    m = (int)decltype(s)::value < cnt ? m + g : m;
                });
                acc[(size_t)(sb + k) * ms] = m;
            }
            sb += in + 1;
        }
        if (c.kind == GK_LV_UDE) {
            static_for<0, 2>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                real m = acc[(size_t)(NSLW + i) * ms];
                static_for<0, NSTC>([&](auto s) {
                    const lds_t* p = c.fac + decltype(s)::value * STG + 2 * LMAX * H;
                    const real g = (c.lin_on[i] * p[i]) * p[2 + i];
                    m = (int)decltype(s)::value < cnt ? m + g : m;
                });
                acc[(size_t)(NSLW + i) * ms] = m;
            });
        }
    }
    static __device__ __forceinline__ void init_norm01(const Ctx& c, real abstol, real reltol, const real* mu, int ms,
                                                       real& h0, real& l0, real& h1, real& l1) {
        for_each_slot<1, 1u>(c, mu, ms, [&](int, const real* g, real m) {
            const real sk = rfma(rabs(m), reltol, abstol);
            const real q0 = m / sk, q1 = g[0] / sk;
            dd_acc(h0, l0, q0 * q0);
            dd_acc(h1, l1, q1 * q1);
        });
    }
    static __device__ __forceinline__ void init_norm2(const Ctx& c, real abstol, real reltol, const real* mu, int ms,
                                                      real& h2, real& l2) {
        for_each_slot<2, 3u>(c, mu, ms, [&](int, const real* g, real m) {
            const real sk = rfma(rabs(m), reltol, abstol);
            const real q = (g[1] - g[0]) / sk;
            dd_acc(h2, l2, q * q);
        });
    }
    // theta index of lane r's slot s, or -1
    static __device__ __forceinline__ int slot_index(const ModelConsts& mc, int r, int s) {
        const int j = r & 63;
        if (s >= NSLW) {
            const int i = s - NSLW;
            return (j == 0 && i < 2 && mc.kind == GK_LV_UDE && mc.lin_idx[i] >= 0) ? mc.lin_idx[i] : -1;
        }
        int o = 0, sb = 0;
        for (int l = 0; l < mc.n_layers; ++l) {
            const int in = mc.dims[l], out = mc.dims[l + 1];
            if (s < sb + in + 1) {
                if (j >= out) return -1;
                const int k = s - sb;
                return mc.nn_offset + o + (k < in ? j + k * out : in * out + j);
            }
            o += in * out + out;
            sb += in + 1;
        }
        return -1;
    }
};

}  // namespace ude
