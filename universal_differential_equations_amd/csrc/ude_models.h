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

// ---------------------------------------------------------------------------------------------
// lotka!  (LotkaVolterra/scenario_1.jl:30-34): theta = (alpha, beta, gamma, delta).  No network:
// every lane of the group computes the same thing (use G = 1).
// ---------------------------------------------------------------------------------------------
template <int G>
struct LvTrue {
    static constexpr int NS = 2, NSL = 4, NTHETA_LDS = 4;
    struct Ctx {
        const double* th;
        int r;
    };
    static __device__ __forceinline__ void init(Ctx& c, const double* th_lds, const ModelConsts&, int r) {
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
struct LvUde {
    using Mlp = CoopMlp<Net, G>;
    static_assert(Net::dim(0) == 2 && Net::dim(Net::L) == 2, "LV UDE network maps R^2 -> R^2");
    static constexpr int NS = 2;
    static constexpr int NSL = Mlp::NSLOT + 2;  // + the two (optional) trainable diagonal coefficients
    struct Ctx {
        const double* th;   // full theta (LDS)
        const double* nn;   // th + nn_offset
        double lin[2];
        double lead_on[2];  // sign if this lane owns a trainable diagonal coefficient, else 0
        int r;
    };
    static __device__ __forceinline__ void init(Ctx& c, const double* th_lds, const ModelConsts& mc, int r) {
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

}  // namespace ude
