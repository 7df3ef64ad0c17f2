// ude_model_kpp_generic.h -- nn_ode (FisherKPP/Fisher-KPP-CNN.jl:111-126, Fisher-KPP-CNN-Small.jl:88-94, LotkaVolterra/scenario_3.jl:103-114)
// with a RUNTIME-SHAPE pointwise reaction network: du_i = NN(u_i) + D0 (w1 u_{i-1} + w2 u_i + w3 u_{i+1}), theta = [NN; w1 w2 w3 unused; D0],
// where NN is any chain 1 -> ... -> 1 of <= 4 Dense layers of width <= 32 (identity / tanh / rbf / relu) with at most KG_NPMAX
// parameters in all.  `n_weights` and the layer list of the scripts are variables; KppUde<Net, G, PPL> covers the shapes they ship as
// compiled instances, THIS model covers the others on grids of <= 32 points (the reference's 26).
//
// Layout = KppUde's: the STATE is distributed (point i on lane i), every lane runs the whole pointwise network for its point, then
// the lanes switch role and OWN PARAMETERS (theta index p = r + G m) and accumulate sum_i delta_l[j][i] a_{l-1}[k][i] over the points
// in ascending order as a fused chain (the oracle's order: bit-identical).  What differs: layer sizes and activations are kernel
// arguments (ModelConsts::dims / act, wave-uniform scalar loads), the per-point activations a_l and deltas live in the lane's own rows
// of the [point][row] LDS tiles instead of registers, and the weights are read from the block's LDS copy of theta with runtime
// indices (broadcast reads).  Arithmetic = oracle mlp_forward / mlp_vjp_acc (chains in ascending order from 0; all dots have < 64 terms).
#pragma once

namespace ude {

constexpr int KG_LMAX = 4, KG_WMAX = 32;
constexpr int KG_NPMAX = 768;   // parameters in all (network + 5): 24 slots per lane on 32 lanes

template <int G, int PPL>
struct KppGenericUde : LinearTheta {
    static_assert(PPL == 1 && G == 32, "runtime-shape Fisher-KPP: one point per lane, grids of <= 32 points");
    static constexpr bool RECOMPUTE_OK = true;
    static constexpr bool PER_MEMBER_THETA = true;   // UDE_PT_THETA: theta is read through init()'s pointer (the member's column in HBM)
    static __host__ __device__ constexpr int point(int c, int r) { return c * G + r; }
    static constexpr int NS = PPL;
    static constexpr int NSL = KG_NPMAX / G;
    static constexpr bool STATE_DISTRIBUTED = true;
    static constexpr int NPT = G * PPL;
    static constexpr int RA = (1 + (KG_LMAX - 1) * KG_WMAX) | 1, RD = ((KG_LMAX - 1) * KG_WMAX + 1) | 1;   // odd row strides
    static constexpr int SCRATCH = 2 * NPT + 4 + (RA + 2 * RD) * G;   // u row, lambda row, A tile, D tile, dphi tile
    struct Ctx {
        const real* th;
        const real* nn;
        const ModelConsts* mc;
        real *urow, *lrow, *A, *Dt, *Ph;
        real w1, w2, w3, D0;
        int r, n, so, d0o, nno, L, nnp;
        int a_row[NSL], d_row[NSL];  // per owned parameter: LDS row of its a factor (-1: bias) and of its delta
        int kind[NSL];               // 0 NN weight/bias, 1 w1, 2 w2, 3 w3, 4 D0, -1 padding / unused slot
    };
    static __device__ __forceinline__ void init(Ctx& c, real* th_lds, real* scratch, real*, int, const ModelConsts& mc, int r, const real* = nullptr) {
        c.th = th_lds;
        c.nn = th_lds + mc.nn_offset;
        c.mc = &mc;
        c.urow = scratch; c.lrow = scratch + NPT + 2; c.A = scratch + 2 * NPT + 4; c.Dt = c.A + RA * G; c.Ph = c.Dt + RD * G;
        c.r = r; c.n = mc.n_state; c.so = mc.stencil_offset; c.d0o = mc.d0_offset; c.nno = mc.nn_offset; c.L = mc.n_layers;
        int np = 0;
        for (int l = 0; l < mc.n_layers; ++l) np += mc.dims[l] * mc.dims[l + 1] + mc.dims[l + 1];
        c.nnp = np;
        c.w1 = th_lds[c.so]; c.w2 = th_lds[c.so + 1]; c.w3 = th_lds[c.so + 2]; c.D0 = th_lds[c.d0o];
        for (int m = 0; m < NSL; ++m) {
            const int p = r + G * m;
            c.kind[m] = -1; c.a_row[m] = -1; c.d_row[m] = -1;
            if (p >= mc.n_param) continue;
            if (p == c.so) c.kind[m] = 1;
            else if (p == c.so + 1) c.kind[m] = 2;
            else if (p == c.so + 2) c.kind[m] = 3;
            else if (p == c.d0o) c.kind[m] = 4;
            else if (p >= c.nno && p < c.nno + np) {
                int q = p - c.nno, aoff = 0, doff = 0;
                for (int l = 0; l < mc.n_layers; ++l) {
                    const int in = mc.dims[l], out = mc.dims[l + 1], sz = in * out + out;
                    if (q < sz) {
                        c.kind[m] = 0;
                        if (q < in * out) { c.d_row[m] = doff + q % out; c.a_row[m] = aoff + q / out; }
                        else { c.d_row[m] = doff + (q - in * out); c.a_row[m] = -1; }
                        break;
                    }
                    q -= sz; aoff += in; doff += out;
                }
            }
        }
    }
    static __device__ __forceinline__ real actf(int a, real z) {
        return a == ACT_TANH ? rtanh(z) : a == ACT_RBF ? rexp(-(z * z)) : a == ACT_RELU ? (z > real(0) ? z : real(0)) : z;
    }
    // ARITH-SPEC dot of n terms (term i = w[i * ws] * x[i]) of a product with nres results (oracle: wide_dot; n <= 32 here): 32 terms
    // reducing to fewer than 16 results are rounded products under the adjacent-pair tree, everything else one ascending fma chain
    static __device__ __forceinline__ real dot_rule(const real* w, int ws, const real* x, int n, int nres) {
        if (n == 32 && nres < 16) {
            real v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = w[i * ws] * x[i];
#pragma unroll
            for (int m = 32; m > 1; m >>= 1)
#pragma unroll
                for (int i = 0; i < m / 2; ++i) v[i] = v[2 * i] + v[2 * i + 1];
            return v[0];
        }
        real acc = real(0);
#pragma unroll 4
        for (int k = 0; k < n; ++k) acc = rfma(w[k * ws], x[k], acc);
        return acc;
    }
    static __device__ __forceinline__ real dactf(int a, real z, real av) {
        return a == ACT_TANH ? rfma(-av, av, real(1)) : a == ACT_RBF ? (real(-2) * z) * av : a == ACT_RELU ? (z > real(0) ? real(1) : real(0)) : real(1);
    }
    // the pointwise network at this lane's point: activations of every layer into the lane's A row (a_0 = u_i first), act' into its
    // dphi row when the reverse sweep follows; returns NN(u_i)
    static __device__ __forceinline__ real forward(const Ctx& c, real ui, bool want_dphi) {
        real* a = c.A + c.r * RA;
        real* ph = c.Ph + c.r * RD;
        a[0] = ui;
        int off = 0, aoff = 0, doff = 0;
        real y = ui;
#pragma unroll 1
        for (int l = 0; l < c.L; ++l) {
            const int in = c.mc->dims[l], out = c.mc->dims[l + 1], actl = c.mc->act[l];
            const real* W = c.nn + off;
#pragma unroll 1
            for (int j = 0; j < out; ++j) {
                real acc = dot_rule(W + j, out, a + aoff, in, out);
                acc += W[in * out + j];
                const real av = actf(actl, acc);
                if (l + 1 < c.L) a[aoff + in + j] = av; else y = av;
                if (want_dphi) ph[doff + j] = dactf(actl, acc, av);
            }
            off += in * out + out; aoff += in; doff += out;
        }
        return y;
    }
    // reverse sweep at this lane's point: deltas of every layer into the lane's D row; returns dNN/du_i * li
    static __device__ __forceinline__ real backward(const Ctx& c, real li) {
        const real* ph = c.Ph + c.r * RD;
        real* d = c.Dt + c.r * RD;
        int off = c.nnp, doff = 0;
        for (int l = 0; l < c.L; ++l) doff += c.mc->dims[l + 1];
        // delta of the output layer (one neuron)
        real gx = li;
#pragma unroll 1
        for (int l = c.L - 1; l >= 0; --l) {
            const int in = c.mc->dims[l], out = c.mc->dims[l + 1];
            off -= in * out + out; doff -= out;
            const real* W = c.nn + off;
            if (l == c.L - 1) d[doff] = gx * ph[doff];
            else
#pragma unroll 1
                for (int j = 0; j < out; ++j) d[doff + j] = d[doff + j] * ph[doff + j];   // (prev of the layer above was parked here)
            if (l > 0) {
                const int pin = c.mc->dims[l];   // = in: the outputs of layer l - 1
                real* dp = d + doff - pin;
#pragma unroll 1
                for (int k = 0; k < in; ++k) dp[k] = dot_rule(W + k * out, 1, d + doff, out, in);
            } else {
                gx = dot_rule(W, 1, d + doff, out, 1);   // in == 1
            }
        }
        return gx;
    }
    static __device__ __forceinline__ void rhs(const Ctx& c, const real* u, real* du) {
        const int n = c.n;
        __syncthreads();
        if (c.r < n) c.urow[c.r] = u[0];
        __syncthreads();
        real out = real(0);
        const int i = c.r;
        if (i < n) {
            const int im = wrap_prev(i, n), ip = wrap_next(i, n);
            const real ui = c.urow[i];
            const real y = forward(c, ui, false);
            const real cnn = c.w1 * c.urow[im] + c.w2 * ui + c.w3 * c.urow[ip];
            out = y + c.D0 * cnn;
        }
        du[0] = out;
    }
    template <bool WANT_PARAM>
    static __device__ __forceinline__ void vjp(const Ctx& c, const real* u, const real* lam, real* dlam, real* g) {
        const int n = c.n, i = c.r;
        __syncthreads();
        if (i < n) { c.urow[i] = u[0]; c.lrow[i] = lam[0]; }
        __syncthreads();
        real gxi = real(0);
        if (i < n) {
            forward(c, c.urow[i], true);
            gxi = backward(c, c.lrow[i]);
            const int im = wrap_prev(i, n), ip = wrap_next(i, n);
            dlam[0] = gxi + c.D0 * (c.w1 * c.lrow[ip] + c.w2 * c.lrow[i] + c.w3 * c.lrow[im]);   // transpose of the periodic stencil
        } else {
            dlam[0] = real(0);
        }
        __syncthreads();
        if constexpr (WANT_PARAM) {
            // grids of <= 32 points are ONE ARITH-SPEC block: a fused chain over the points in ascending order (bias: plain adds)
#pragma unroll 1
            for (int m = 0; m < NSL; ++m) {
                real acc = real(0);
                const int kd = c.kind[m];
                if (kd == 0) {
                    const real* dr = c.Dt + c.d_row[m];
                    if (c.a_row[m] >= 0) {
                        const real* ar = c.A + c.a_row[m];
                        for (int q = 0; q < n; ++q) acc = rfma(dr[q * RD], ar[q * RA], acc);
                    } else {
                        for (int q = 0; q < n; ++q) acc += dr[q * RD];
                    }
                } else if (kd >= 1) {
                    real s = real(0);
                    for (int q = 0; q < n; ++q) {
                        const int im = wrap_prev(q, n), ip = wrap_next(q, n);
                        if (kd == 1) s = rfma(c.lrow[q], c.urow[im], s);
                        else if (kd == 2) s = rfma(c.lrow[q], c.urow[q], s);
                        else if (kd == 3) s = rfma(c.lrow[q], c.urow[ip], s);
                        else s = rfma(c.lrow[q], c.w1 * c.urow[im] + c.w2 * c.urow[q] + c.w3 * c.urow[ip], s);
                    }
                    acc = kd == 4 ? s : c.D0 * s;
                }
                g[m] = acc;
            }
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

}  // namespace ude
