// ude_models.h -- the UDE right-hand-side families (a3/a4/a5) as device policies.
//
// A model policy provides, for a lane group of size G:
//   NS   replicated state components            NSL  per-lane slots of the parameter cotangent
//   rhs(ctx, u, du)                             du = f(u, theta)
//   vjp<WANT_PARAM>(ctx, u, lam, dlam, g)       dlam = (df/du)^T lam ; g[slot] = this lane's (df/dtheta)^T lam
//   slot_index(mc, r, s)                        theta index of lane r's slot s, or -1 (padding)
#pragma once
#include "ude_coop.h"

namespace ude {

struct ModelConsts {
    int32_t n_state, n_param, nn_offset, stencil_offset, d0_offset;
    int32_t lin_idx[2];
    double lin_sign[2], lin_const[2];
    double consts[16];
};


// default: theta is copied linearly into LDS
struct LinearTheta {
    static __host__ __device__ constexpr int theta_lds(int np) { return (np + 1) & ~1; }
    static __device__ __forceinline__ void stage_theta(double* th, const double* theta, int np, int tid, int nthreads) {
        for (int i = tid; i < np; i += nthreads) th[i] = theta[i];
    }
    static constexpr int SCRATCH = 0;
};

// ---------------------------------------------------------------------------------------------
// lotka!  (LotkaVolterra/scenario_1.jl:30-34): theta = (alpha, beta, gamma, delta).  No network:
// every lane of the group computes the same thing (use G = 1).
// ---------------------------------------------------------------------------------------------
template <int G>
struct LvTrue : LinearTheta {
    static constexpr int NS = 2, NSL = 4, NTHETA_LDS = 4;
    static constexpr bool SLOTS_IN_LDS = false, STATE_DISTRIBUTED = false;
    struct Ctx {
        const double* th;
        int r;
    };
    static __device__ __forceinline__ void init(Ctx& c, double* th_lds, double*, double*, int, const ModelConsts&, int r) {
        c.th = th_lds;
        c.r = r;
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const double* u, double* du) {
        const double a = c.th[0], b = c.th[1], g = c.th[2], d = c.th[3];
        du[0] = a * u[0] - b * u[1] * u[0];
        du[1] = g * u[0] * u[1] - d * u[1];
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const double* u, const double* lam, double* dlam,
                                               double* g) {
        const double a = c.th[0], b = c.th[1], gm = c.th[2], d = c.th[3];
        dlam[0] = (a - b * u[1]) * lam[0] + (gm * u[1]) * lam[1];
        dlam[1] = (-b * u[0]) * lam[0] + (gm * u[0] - d) * lam[1];
        if constexpr (WANT_PARAM) {
            const double on = c.r == 0 ? 1.0 : 0.0;
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
template <class Net, int G>
struct LvUde : LinearTheta {
    using Mlp = CoopMlp<Net, G>;
    static_assert(Net::dim(0) == 2 && Net::dim(Net::L) == 2, "LV UDE network maps R^2 -> R^2");
    static constexpr int NS = 2;
    static constexpr int NSL = Mlp::NSLOT + 2;  // + the two (optional) trainable diagonal coefficients
    static constexpr bool SLOTS_IN_LDS = false, STATE_DISTRIBUTED = false;
    struct Ctx {
        const double* th;   // full theta (LDS)
        const double* nn;   // th + nn_offset
        double lin[2];
        double lead_on[2];  // sign if this lane owns a trainable diagonal coefficient, else 0
        int r;
    };
    static __device__ __forceinline__ void init(Ctx& c, double* th_lds, double*, double*, int, const ModelConsts& mc, int r) {
        c.th = th_lds;
        c.nn = th_lds + mc.nn_offset;
        c.r = r;
        for (int i = 0; i < 2; ++i) {
            c.lin[i] = mc.lin_idx[i] >= 0 ? mc.lin_sign[i] * th_lds[mc.lin_idx[i]] : mc.lin_const[i];
            c.lead_on[i] = (mc.lin_idx[i] >= 0 && r == 0) ? mc.lin_sign[i] : 0.0;
        }
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const double* u, double* du) {
        typename Mlp::Cache cache;
        double y[2];
        Mlp::forward(c.nn, c.r, u, cache, y);
        du[0] = __builtin_fma(c.lin[0], u[0], y[0]);
        du[1] = __builtin_fma(c.lin[1], u[1], y[1]);
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const double* u, const double* lam, double* dlam,
                                               double* g) {
        typename Mlp::Cache cache;
        double y[2], gx[2];
        Mlp::forward(c.nn, c.r, u, cache, y);
        Mlp::template vjp<WANT_PARAM>(c.nn, c.r, cache, lam, gx, g);
        dlam[0] = __builtin_fma(c.lin[0], lam[0], gx[0]);
        dlam[1] = __builtin_fma(c.lin[1], lam[1], gx[1]);
        if constexpr (WANT_PARAM) {
            g[Mlp::NSLOT + 0] = (c.lead_on[0] * u[0]) * lam[0];
            g[Mlp::NSLOT + 1] = (c.lead_on[1] * u[1]) * lam[1];
        }
    }
    static __device__ __forceinline__ int slot_index(const ModelConsts& mc, int r, int s) {
        if (s >= Mlp::NSLOT) {
            const int i = s - Mlp::NSLOT;
            return (r == 0 && mc.lin_idx[i] >= 0) ? mc.lin_idx[i] : -1;
        }
        const int k = Mlp::slot_index(r, s);
        return k < 0 ? -1 : mc.nn_offset + k;
    }
};

// ---------------------------------------------------------------------------------------------
// corona!  (SEIR_exposure/seir_exposure.jl:16-30): mechanistic 7-state model, consts = p_[0..8] =
// F, beta0, alpha, kappa, mu, sigma, gamma, d, lambda.  No trainable parameters (data generation).
// ---------------------------------------------------------------------------------------------
template <int G>
struct SeirTrue : LinearTheta {
    static constexpr int NS = 7, NSL = 0;
    static constexpr bool SLOTS_IN_LDS = false, STATE_DISTRIBUTED = false;
    struct Ctx {
        double F, b0, al, ka, mu, sg, ga, d, la;
        int r;
    };
    static __device__ __forceinline__ void init(Ctx& c, double*, double*, double*, int, const ModelConsts& mc, int r) {
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
// dudt_  (seir_exposure.jl:114-130): z = ann([S/N, I, D/N]) with ann = 3 -> 64 -> 64 -> 1 tanh
// (4481 parameters); dS = -beta0*S*F/N - z - mu*S, dE = beta0*S*F/N + z - (sigma+mu)*E, ...
//
// One WAVEFRONT per trajectory (G = 64): lane j owns hidden neuron j of both hidden layers; activations and
// deltas of a layer are exchanged through LDS (one 64-double row each, broadcast reads); theta is staged in
// LDS with the 64x64 matrix padded to a leading dimension of 65 so that row reads (forward) and column
// reads (backward) are both bank-conflict free.  The parameter cotangent mu and its two accumulators are
// theta-indexed LDS arrays (3 x 4481 doubles = 105 KiB): "parity mode" of SURVEY.md 7.6 -- mu takes part in
// the error norm exactly as upstream's augmented state does.
// ---------------------------------------------------------------------------------------------
template <int G>
struct SeirUde {
    static_assert(G == 64, "SEIR UDE kernel is wavefront-per-trajectory");
    static constexpr int NS = 7, NSL = 0, H = 64, LD = 65;
    static constexpr bool SLOTS_IN_LDS = true, STATE_DISTRIBUTED = false;
    static constexpr int NPARAM = 3 * H + H + H * H + H + H + 1;           // 4481
    static constexpr int THETA_LDS = 3 * H + H + H * LD + H + H + 2;       // padded copy
    static constexpr int SCRATCH = 4 * H;                                  // act1, act2, delta2, delta1
    struct Ctx {
        const double *W1, *b1, *W2p, *b2, *W3, *b3;
        double *act1, *act2, *dl2, *dl1;
        double *mu, *ab, *ae;
        double F, b0, mu_c, sg, ga, d, la;
        int r;
    };
    static __host__ __device__ constexpr int theta_lds(int) { return THETA_LDS; }
    // theta (global) -> padded LDS copy
    static __device__ __forceinline__ void stage_theta(double* th, const double* theta, int, int tid, int nthreads) {
        for (int i = tid; i < 3 * H + H; i += nthreads) th[i] = theta[i];                       // W1, b1
        for (int i = tid; i < H * H; i += nthreads) th[4 * H + (i % H) + (i / H) * LD] = theta[4 * H + i];  // W2 -> ld 65
        for (int i = tid; i < H + H + 1; i += nthreads) th[4 * H + H * LD + i] = theta[4 * H + H * H + i];  // b2, W3, b3
    }
    static __device__ __forceinline__ void init(Ctx& c, double* th, double* scratch, double* slots, int np_pad,
                                                const ModelConsts& mc, int r) {
        c.W1 = th; c.b1 = th + 3 * H; c.W2p = th + 4 * H; c.b2 = c.W2p + H * LD; c.W3 = c.b2 + H; c.b3 = c.W3 + H;
        c.act1 = scratch; c.act2 = scratch + H; c.dl2 = scratch + 2 * H; c.dl1 = scratch + 3 * H;
        c.mu = slots; c.ab = slots + np_pad; c.ae = slots + 2 * np_pad;
        c.F = mc.consts[0]; c.b0 = mc.consts[1]; c.mu_c = mc.consts[4]; c.sg = mc.consts[5]; c.ga = mc.consts[6];
        c.d = mc.consts[7]; c.la = mc.consts[8];
        c.r = r;
    }
    // forward network: returns z; leaves act1/act2 in LDS and this lane's pre-activation-free cache in registers
    static __device__ __forceinline__ double net(const Ctx& c, const double* x, double& a1, double& a2) {
        const int j = c.r;
        double acc = 0.0;
        static_for<0, 3>([&](auto k) { acc = __builtin_fma(c.W1[j + k * H], x[k], acc); });
        acc += c.b1[j];
        a1 = dtanh(acc);
        c.act1[j] = a1;
        __syncthreads();
        acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < H; ++k) acc = __builtin_fma(c.W2p[j + k * LD], c.act1[k], acc);
        acc += c.b2[j];
        a2 = dtanh(acc);
        c.act2[j] = a2;
        __syncthreads();
        double zz = 0.0;
#pragma unroll 8
        for (int k = 0; k < H; ++k) zz = __builtin_fma(c.W3[k], c.act2[k], zz);
        zz += c.b3[0];
        return zz;
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const double* u, double* du) {
        const double S = u[0], E = u[1], I = u[2], Rr = u[3], N = u[4], D = u[5];
        const double x[3] = {S / N, I, D / N};
        double a1, a2;
        const double z = net(c, x, a1, a2);
        __syncthreads();  // act rows are rewritten by the next evaluation
        du[0] = -c.b0 * S * c.F / N - z - c.mu_c * S;
        du[1] = c.b0 * S * c.F / N + z - (c.sg + c.mu_c) * E;
        du[2] = c.sg * E - (c.ga + c.mu_c) * I;
        du[3] = c.ga * I - c.mu_c * Rr;
        du[4] = -c.mu_c * N;
        du[5] = c.d * c.ga * I - c.la * D;
        du[6] = c.sg * E;
    }
    // adjoint evaluation: dlam = (df/du)^T lam; the NEGATED parameter cotangent g = -(df/dtheta)^T lam is
    // accumulated straight into the LDS accumulators: ab = first ? bs*g : fma(bs, g, ab) (ae likewise with es)
    static __device__ __forceinline__ void vjp_acc(const Ctx& c, const double* u, const double* lam, double* dlam,
                                                   double bs, double es, bool first) {
        const int j = c.r;
        const double S = u[0], N = u[4], D = u[5];
        const double x[3] = {S / N, u[2], D / N};
        double a1, a2;
        net(c, x, a1, a2);
        auto acc = [&](int idx, double gpos) {
            const double g = -gpos;
            c.ab[idx] = first ? bs * g : __builtin_fma(bs, g, c.ab[idx]);
            c.ae[idx] = first ? es * g : __builtin_fma(es, g, c.ae[idx]);
        };
        const double d3 = (lam[1] - lam[0]) * 1.0;                        // output layer is linear
        // layer 3 (W3: 1 x 64, b3): lane k owns W3[k]
        acc(4 * H + H * H + H + j, d3 * c.act2[j]);
        if (j == 0) acc(4 * H + H * H + 2 * H, d3);
        // delta2_j = (W3[j] * d3) * tanh'(a2_j)
        const double d2 = __builtin_fma(c.W3[j], d3, 0.0) * __builtin_fma(-a2, a2, 1.0);
        c.dl2[j] = d2;
        // layer 2 row j: dW2[j,k] = d2 * act1[k], db2[j] = d2
#pragma unroll 8
        for (int k = 0; k < H; ++k) acc(4 * H + j + k * H, d2 * c.act1[k]);
        acc(4 * H + H * H + j, d2);
        __syncthreads();
        // delta1_j = (sum_i W2[i,j] * delta2[i]) * tanh'(a1_j)   (column j of W2)
        double s1 = 0.0;
#pragma unroll 8
        for (int i = 0; i < H; ++i) s1 = __builtin_fma(c.W2p[i + j * LD], c.dl2[i], s1);
        const double d1 = s1 * __builtin_fma(-a1, a1, 1.0);
        c.dl1[j] = d1;
        static_for<0, 3>([&](auto m) { acc(j + m * H, d1 * x[m]); });
        acc(3 * H + j, d1);
        __syncthreads();
        double gx[3];
        static_for<0, 3>([&](auto m) {
            double s = 0.0;
#pragma unroll 8
            for (int i = 0; i < H; ++i) s = __builtin_fma(c.W1[i + m * H], c.dl1[i], s);
            gx[m] = s;
        });
        __syncthreads();
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
};

}  // namespace ude
