// ude_sde.h -- SURVEY.md 8(f) N1 / BASELINE configs[4]: the deep-BSDE training step of highdim_pde/lambaem.jl:8-48 as fused
// gfx950 kernels.  An ensemble of adaptive Euler-Maruyama (StochasticDiffEq LambaEM) solves of
//     dX = sigma dW,   du = lambda |z|^2 dt + z . dW,   z = sigma^T grad u net([X; t])          (NNPDENS: F, G)
// steps in lock-step per block of 32 trajectories; the three network evaluations of a step attempt (drift/diffusion at
// (X, t), drift at (X, t + dt), diffusion at the Lamba probe point) are batched as columns of FP32 matrix-core GEMMs
// (v_mfma_f32_32x32x2_f32: bit for bit an fmaf chain over k ascending with the bias as C operand, i.e. exactly the
// oracle's Dense layer).  Each of the block's four wavefronts owns a 32-row slab of every layer's output and keeps
// ITS weights in registers for the whole solve (216 VGPR/AGPR): the weights never touch LDS; activations ping-pong
// between two LDS tiles.  The per-trajectory scalar work (tree sums over the 100 components, error estimate, PI
// controller, rejection sampling with memory, Philox/Box-Muller normals) runs one trajectory per wavefront at a time
// with components 2l, 2l+1 on lane l.  Algorithm and arithmetic order: oracle/sde_oracle_impl.h (the restatement;
// parity with upstream itself is unpinned, see oracle/sde_oracle.h).
#pragma once
#include "ude_math.h"

namespace ude {
namespace hjb {

typedef float v16f __attribute__((ext_vector_type(16)));

enum { RET_SUCCESS = 0, RET_MAXITERS = 1, RET_UNSTABLE = 3, RET_STORE_OVERFLOW = 4, RET_STACK_OVERFLOW = 5 };
constexpr int NT = 32;      // trajectories per block (one N-tile per evaluation kind)
constexpr int LDA = 97;     // leading dimension of the [row][column] activation tiles (odd: row and column walks conflict-free)
constexpr int XLD = 128;    // per-trajectory row of the state / increment arrays
constexpr int STACK = 32;   // RSwM stack depth
constexpr int LDT = 33;     // leading dimension of the backward kernel's [row][column] tiles

template <int D, int H>
struct Cfg {
    static_assert(D % 2 == 0 && D <= 126 && H <= 127, "components 2l, 2l+1 per lane; one 128-row tile per layer");
    static constexpr int DIN = D + 1;
    static constexpr int KS1 = (DIN + 1) / 2, KSH = (H + 1) / 2, KS4T = (D + 1) / 2;  // k-steps (K = 2 per MFMA)
    static constexpr int RX = (D + 2 + 3) & ~3, RA = (H + 1 + 3) & ~3, RE = (D + 3) & ~3;  // record strides (floats)
    // theta_sg offsets (Flux.params order: W (out x in, column-major), b per Dense layer; lambaem.jl:27-30)
    static constexpr int W1 = 0, B1 = H * DIN, W2 = B1 + H, B2 = W2 + H * H, W3 = B2 + H, B3 = W3 + H * H, W4 = B3 + H,
                         B4 = W4 + D * H, NP = B4 + D;
    // theta_u0 offsets (lambaem.jl:23-25)
    static constexpr int U1 = 0, C1 = H * D, U2 = C1 + H, C2 = U2 + H * H, U3 = C2 + H, C3 = U3 + H, NPU = C3 + 1;
};

struct HjbParams {
    int64_t M;
    int32_t cap, maxiters, adaptive, record;
    uint32_t iter;
    uint64_t seed;
    float lam, sig, t0, t1, abstol, reltol, qmin, qmax, gamma, qoldinit, beta1, beta2, dtmax, dt_user;
    const float* x0;
    const float* theta;   // [theta_u0; theta_sg]
    float* prep;          // [0] u0, [1] initial dt, [2 .. 2+H) a1 of the u0 chain, [2+H .. 2+2H) a2
    float *rXin, *rA1, *rA2, *rA3, *rE4;  // accepted-step records, column = traj * cap + step
    int32_t* nacc;        // accepted steps per trajectory
    float* stackW;        // RSwM stack increments [trajectory][depth][XLD]
    float* uT;
    float* XT;
    double* loss_traj;
    float* ubar;
    int64_t* stats;
    int32_t* retcode;
    float* part;          // backward: per-block partial gradients of theta_sg
    float* grad;          // np floats
    double* loss;
    int32_t* nfail;
    int32_t* queue;       // next trajectory of the ensemble (forward kernel's slot queue)
    unsigned long long* prof;  // debug (UDE_HJB_PROF=1): per-phase clock sums of block 0 [phase 0..7], iterations at [8]
};

// ---- Philox4x32-10 + Box-Muller (oracle/sde_oracle.c restates the same sequences) ----------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double sincos2pi(double u, double& cosout) {
    const double v = u * 4.0;
    const int q = (int)v;
    const double x = (v - (double)q) * 1.5707963267948966;
    const double x2 = x * x;
    double ps = -1.0 / 51090942171709440000.0;
    ps = __builtin_fma(ps, x2, 1.0 / 121645100408832000.0);
    ps = __builtin_fma(ps, x2, -1.0 / 355687428096000.0);
    ps = __builtin_fma(ps, x2, 1.0 / 1307674368000.0);
    ps = __builtin_fma(ps, x2, -1.0 / 6227020800.0);
    ps = __builtin_fma(ps, x2, 1.0 / 39916800.0);
    ps = __builtin_fma(ps, x2, -1.0 / 362880.0);
    ps = __builtin_fma(ps, x2, 1.0 / 5040.0);
    ps = __builtin_fma(ps, x2, -1.0 / 120.0);
    ps = __builtin_fma(ps, x2, 1.0 / 6.0);
    const double s = __builtin_fma(-(x * x2), ps, x);
    double pc = 1.0 / 2432902008176640000.0;
    pc = __builtin_fma(pc, x2, -1.0 / 6402373705728000.0);
    pc = __builtin_fma(pc, x2, 1.0 / 20922789888000.0);
    pc = __builtin_fma(pc, x2, -1.0 / 87178291200.0);
    pc = __builtin_fma(pc, x2, 1.0 / 479001600.0);
    pc = __builtin_fma(pc, x2, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, x2, 1.0 / 40320.0);
    pc = __builtin_fma(pc, x2, -1.0 / 720.0);
    pc = __builtin_fma(pc, x2, 1.0 / 24.0);
    pc = __builtin_fma(pc, x2, -0.5);
    const double c = __builtin_fma(pc, x2, 1.0);
    const int qq = q & 3;
    const double so = qq == 0 ? s : qq == 1 ? c : qq == 2 ? -s : -c;
    cosout = qq == 0 ? c : qq == 1 ? -s : qq == 2 ? -c : s;
    return so;
}

// the two standard normals of lane l (components 2l, 2l+1) of draw event `ev`: chunk l/2, Box-Muller pair l%2
__device__ __forceinline__ void normal_pair(uint64_t seed, uint32_t iter, uint32_t traj, uint32_t ev, int l, double& n0, double& n1) {
    uint32_t r[4];
    philox4x32_10((uint32_t)(l >> 1), ev, traj, iter, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int h = l & 1;
    const uint32_t a = h ? r[2] : r[0], b = h ? r[3] : r[1];
    const double u1 = ((double)a + 0.5) * 2.3283064365386963e-10;
    const double u2 = (double)b * 2.3283064365386963e-10;
    const double rad = sqrt(-2.0 * dlog(u1));
    double co;
    const double si = sincos2pi(u2, co);
    n0 = rad * co;
    n1 = rad * si;
}

// ---- wavefront tree sum (binary tree over adjacent index pairs; every lane receives the total) ---------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_tree_sum_f32(float x) {
    x += dpp_f32<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f32<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f32<0x141>(x);  // row_half_mirror
    x += dpp_f32<0x140>(x);  // row_mirror
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
// tsum of a vector whose components 2l, 2l+1 sit on lane l (zeros beyond the vector)
__device__ __forceinline__ float tsum2(float v0, float v1) { return wave_tree_sum_f32(v0 + v1); }

// relu as the oracle states it (oracle/sde_oracle_impl.h: dense): the value of max(0, x), with the SIGN BIT of a non-positive
// pre-activation kept (-0.0 for a negative one): the reverse sweeps need "pre-activation >= 0" (relu'(0) = 1, the derivative
// Tracker / ForwardDiff give NNlib's relu(x) = max(zero(x), x)), and the zero's sign never changes a sum
__device__ __forceinline__ float relu_enc(float a) { return a > 0.0f ? a : __uint_as_float(__float_as_uint(a) & 0x80000000u); }
__device__ __forceinline__ bool relu_on(float a) { return (__float_as_uint(a) >> 31) == 0u; }

// ---- one Dense layer on the matrix cores: out[32w .. 32w+31][columns of tiles nt0..nt1) = act(W in + b) -----------------
// wf: this lane's weight fragments (A operand: row 32w + (l&31), k = 2s + (l>>5)); in/out: LDS tiles [row][LDA]
template <int KS, bool RELU>
__device__ __forceinline__ void layer(const float (&wf)[KS], const float* in, float* out, const float* bias, int nt0, int nt1, int w,
                                      int l, float* rec, int recLD, const int* naccS, const int* doneS, const int* jS, int cap) {
    const int rbase = 32 * w + 4 * (l >> 5);
    for (int nt = nt0; nt < nt1; ++nt) {
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias[rbase + (r & 3) + 8 * (r >> 2)];
        const float* bp = in + (l >> 5) * LDA + nt * 32 + (l & 31);
        // B operands are fetched PF k-steps ahead of the matrix-core instruction that consumes them: the dependent MFMA
        // chain (64 cycles per instruction) then never waits for an LDS round trip
        // (sched_barrier: the machine scheduler otherwise sinks every load next to its use to save registers)
        constexpr int PF = KS < 6 ? KS : 6;
        float bq[PF];
        static_for<0, PF>([&](auto ic) { bq[ic] = bp[2 * decltype(ic)::value * LDA]; });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, KS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const float b = bq[s % PF];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s], b, acc, 0, 0, 0);
            if constexpr (s + PF < KS) bq[s % PF] = bp[2 * (s + PF) * LDA];
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = relu_enc(acc[r]);
        }
        float* op = out + rbase * LDA + nt * 32 + (l & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) op[((r & 3) + 8 * (r >> 2)) * LDA] = acc[r];
        if (rec && nt == 0) {  // activations of the (X, t) evaluation: speculative record of the step under way
            const int tr = l & 31;
            const int slot = naccS[tr];
            if (!doneS[tr] && slot < cap) {
                float* rp = rec + ((size_t)jS[tr] * cap + slot) * recLD;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row0 = rbase + 8 * g;
                    if (row0 < recLD) {
                        float4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                        *reinterpret_cast<float4*>(rp + row0) = v;
                    }
                }
            }
        }
    }
}

// The first layer of the evaluations at (X, t) and (X, t + dt) of one step attempt (column tiles 0 and 1): the two input columns of a
// trajectory differ in the TIME row only, and that row is the LAST term of the layer's chain (k = D, with the zero pad row k = D + 1 the
// operands of the last matrix instruction s = KS - 1).  The chain over k < D is therefore computed ONCE and both tiles finish it with their
// own last instruction: 1 + 1 instead of KS + KS matrix instructions for tile 1 (round 6; the same fmaf chain in the same order, so the
// same bits as layer<KS, true>(.., 0, 2, ..) and as the oracle's Dense layer).  Tile 1 of `in` needs its time and pad rows only.
template <int KS>
__device__ __forceinline__ void layer1_shared_t(const float (&wf)[KS], const float* in, float* out, const float* bias, int w, int l, float* rec,
                                                int recLD, const int* naccS, const int* doneS, const int* jS, int cap) {
    const int rbase = 32 * w + 4 * (l >> 5);
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[rbase + (r & 3) + 8 * (r >> 2)];
    const float* bp = in + (l >> 5) * LDA + (l & 31);
    constexpr int PF = KS < 6 ? KS : 6;
    float bq[PF];
    static_for<0, PF>([&](auto ic) { bq[ic] = bp[2 * decltype(ic)::value * LDA]; });
    const float b1 = bp[2 * (KS - 1) * LDA + 32];   // tile 1's operand of the last instruction: rows D (t + dt) and D + 1 (0)
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, KS - 1>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const float b = bq[s % PF];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s], b, acc, 0, 0, 0);
        if constexpr (s + PF < KS) bq[s % PF] = bp[2 * (s + PF) * LDA];
        __builtin_amdgcn_sched_barrier(0);
    });
    // (tile 1 first, into registers of its own; tile 0 in place: 32 accumulator registers live for two instructions, 16 otherwise)
    v16f acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[KS - 1], b1, acc, 0, 0, 0);
    v16f acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[KS - 1], bq[(KS - 1) % PF], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    float* op = out + rbase * LDA + (l & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) op[((r & 3) + 8 * (r >> 2)) * LDA + 32] = relu_enc(acc1[r]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = relu_enc(acc0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) op[((r & 3) + 8 * (r >> 2)) * LDA] = acc0[r];
    if (rec) {  // activations of the (X, t) evaluation: speculative record of the step under way (as layer())
        const int tr = l & 31;
        const int slot = naccS[tr];
        if (!doneS[tr] && slot < cap) {
            float* rp = rec + ((size_t)jS[tr] * cap + slot) * recLD;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row0 = rbase + 8 * g;
                if (row0 < recLD) {
                    float4 v = {acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
                    *reinterpret_cast<float4*>(rp + row0) = v;
                }
            }
        }
    }
}

template <int D, int H>
__device__ __forceinline__ void load_fwd_weights(const float* th, int w, int l, float (&wf1)[Cfg<D, H>::KS1], float (&wf2)[Cfg<D, H>::KSH],
                                                 float (&wf3)[Cfg<D, H>::KSH], float (&wf4)[Cfg<D, H>::KSH]) {
    using C = Cfg<D, H>;
    const int row = 32 * w + (l & 31), kk = l >> 5;
    static_for<0, C::KS1>([&](auto sc) {
        const int k = 2 * decltype(sc)::value + kk;
        wf1[sc] = (row < H && k < C::DIN) ? th[C::W1 + row + k * H] : 0.0f;
    });
    static_for<0, C::KSH>([&](auto sc) {
        const int k = 2 * decltype(sc)::value + kk;
        const bool in = row < H && k < H;
        wf2[sc] = in ? th[C::W2 + row + k * H] : 0.0f;
        wf3[sc] = in ? th[C::W3 + row + k * H] : 0.0f;
        wf4[sc] = (row < D && k < H) ? th[C::W4 + row + k * D] : 0.0f;
    });
}

template <int D, int H>
__device__ __forceinline__ void load_bias_table(const float* th, float* biasS, int tid, int nthreads) {
    using C = Cfg<D, H>;
    for (int i = tid; i < 4 * 128; i += nthreads) {
        const int L = i >> 7, r = i & 127;
        float v = 0.0f;
        if (L == 0 && r < H) v = th[C::B1 + r];
        if (L == 1 && r < H) v = th[C::B2 + r];
        if (L == 2 && r < H) v = th[C::B3 + r];
        if (L == 3 && r < D) v = th[C::B4 + r];
        biasS[i] = v;
    }
}

// LDS of the forward kernel (floats)
template <int D, int H>
constexpr int fwd_lds_floats() { return 2 * 128 * LDA + 2 * NT * XLD + 4 * 128 + 16 * NT + NT * STACK; }

// sum over the 128-padded component vector when lane m of a 16-lane row holds components 8m .. 8m+7: three in-lane
// levels of the adjacent-pair tree, four DPP levels inside the row (every lane of the row receives the total)
__device__ __forceinline__ float row_tree_sum(float x) {
    x += dpp_f32<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f32<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f32<0x141>(x);  // row_half_mirror
    x += dpp_f32<0x140>(x);  // row_mirror
    return x;
}
__device__ __forceinline__ float tsum8(const float (&v)[8]) {
    return row_tree_sum(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
}
// the 8 standard normals of lane m (components 8m .. 8m+7) of a draw event: Philox chunks 2m, 2m+1
__device__ __forceinline__ void normals8(uint64_t seed, uint32_t iter, uint32_t traj, uint32_t ev, int m, float (&n)[8]) {
#ifdef HJB_EXP_NORNG  // timing experiment only: no Philox / Box-Muller (results are NOT normals)
    for (int i = 0; i < 8; ++i) n[i] = (float)((int)((traj * 2654435761u + ev * 40503u + (uint32_t)(8 * m + i) * 2246822519u) >> 8) - (1 << 23)) * 1.1e-7f;
    return;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t r[4];
        philox4x32_10((uint32_t)(2 * m + h), ev, traj, iter, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const double u1 = ((double)r[2 * g] + 0.5) * 2.3283064365386963e-10;
            const double u2 = (double)r[2 * g + 1] * 2.3283064365386963e-10;
            const double rad = sqrt(-2.0 * dlog(u1));
            double co;
            const double si = sincos2pi(u2, co);
            n[4 * h + 2 * g] = (float)(rad * co);
            n[4 * h + 2 * g + 1] = (float)(rad * si);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: adaptive solves in lock-step, 32 trajectory slots per block; a slot whose trajectory has finished takes the
// next one from a global queue (atomic counter), so the block's columns stay busy until the ensemble is exhausted.
// Scalar phases: a wavefront handles four of its eight slots at a time, one per 16-lane row, lane m of the row holding
// components 8m .. 8m+7 (reductions = 3 in-lane + 4 DPP levels of the adjacent-pair tree; no LDS shuffles).
// ---------------------------------------------------------------------------------------------------------------------
// ADAPTIVE: LambaEM's adaptive stepping (three evaluations per attempt) or the fixed-step Euler-Maruyama mode (one) -- a template parameter
// since round 6: the kernel of each mode carries only its own code (the run-time branch kept both first-layer forms alive)
template <int D, int H, bool ADAPTIVE>
__global__ void __launch_bounds__(256) hjb_fwd_kernel(const HjbParams p) {
    using C = Cfg<D, H>;
    static_assert(D % 4 == 0 && D + 2 <= 128, "component octets per lane");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* bufA = sm;
    float* bufB = bufA + 128 * LDA;
    float* Xs = bufB + 128 * LDA;
    float* dWs = Xs + NT * XLD;
    float* biasS = dWs + NT * XLD;
    float* fT = biasS + 4 * 128;  // per-slot float state: t, dt, u, qold, q11
    float* fDt = fT + NT;
    float* fU = fDt + NT;
    float* fQold = fU + NT;
    float* fQ11 = fQold + NT;
    int* iLast = reinterpret_cast<int*>(fQ11 + NT);
    int* iDone = iLast + NT;
    int* iNacc = iDone + NT;
    int* iNrej = iNacc + NT;
    int* iNstack = iNrej + NT;
    int* iIter = iNstack + NT;
    int* iEv = iIter + NT;
    int* iNdraw = iEv + NT;
    int* iJ = iNdraw + NT;        // trajectory of the slot
    int* iOvf = iJ + NT;          // the slot's trajectory has outgrown the accepted-step store (it keeps stepping, unrecorded, so that its
                                  // true step count reaches the host: ONE re-run with the right capacity instead of a x4 ladder)
    float* sLs = fT + 16 * NT;    // lengths of the RSwM stack pieces [slot][depth] (the increments themselves: HBM)

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const float* thsg = p.theta + C::NPU;
    float wf1[C::KS1], wf2[C::KSH], wf3[C::KSH], wf4[C::KSH];
    load_fwd_weights<D, H>(thsg, w, l, wf1, wf2, wf3, wf4);
    load_bias_table<D, H>(thsg, biasS, tid, 256);
    for (int i = tid; i < 2 * 128 * LDA; i += 256) bufA[i] = 0.0f;  // (bufA and bufB are contiguous)
    const int rr = l >> 4, m = l & 15, cb = 8 * m;
    const bool lane_on = cb < D + 2;  // lane 12 (D = 100) also carries the time row D and the zero pad row D + 1
    const float u0 = p.prep[0];
    const float dt_init = p.prep[1];
    __syncthreads();

    // (re)start slot tr with trajectory j: state, first increment, input columns of the first attempt
    auto start_slot = [&](int tr, int64_t j) {
        float dt = dt_init;
        int last = 0;
        const float rem = p.t1 - p.t0;
        if (dt >= rem) { dt = rem; last = 1; }
        if (lane_on) {
            float nrm[8];
            normals8(p.seed, p.iter, (uint32_t)j, 0u, m, nrm);
            const float s = __builtin_sqrtf(dt);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cb + i;
                if (c < D) {
                    const float x = p.x0[c];
                    Xs[tr * XLD + c] = x;
                    dWs[tr * XLD + c] = s * nrm[i];
                    bufA[c * LDA + tr] = x;
                    // (no X rows for tile 1: the evaluation at t + dt shares tile 0's chain over them -- layer1_shared_t)
                } else if (c == D) {
                    bufA[c * LDA + tr] = p.t0;
                    bufA[c * LDA + 32 + tr] = p.t0 + dt;
                } else if (c == D + 1) {
                    bufA[c * LDA + tr] = 0.0f;
                    bufA[c * LDA + 32 + tr] = 0.0f;
                }
            }
        }
        if (m == 0) {
            fT[tr] = p.t0; fDt[tr] = dt; fU[tr] = u0; fQold[tr] = p.qoldinit; fQ11[tr] = 1.0f;
            iLast[tr] = last; iDone[tr] = 0; iNacc[tr] = 0; iNrej[tr] = 0; iNstack[tr] = 0; iIter[tr] = 0;
            iEv[tr] = 1; iNdraw[tr] = 1; iJ[tr] = (int)j; iOvf[tr] = 0;
        }
    };

    for (int ps = 0; ps < 2; ++ps) {
        const int tr = w * 8 + ps * 4 + rr;
        const int64_t j = (int64_t)blockIdx.x * NT + tr;
        if (j < p.M) start_slot(tr, j);
        else if (m == 0) { iDone[tr] = 1; iJ[tr] = 0; iNacc[tr] = 0; iOvf[tr] = 0; }
    }
    __syncthreads();

    constexpr int nA = ADAPTIVE ? 2 : 1;
    // debug phase clocks (s_memtime; one lane of block 0): [0] evaluations 1+2, [1] S1, [2] evaluation 3, [3] S2, [4] loop-end barrier
    const bool prof = p.prof != nullptr && blockIdx.x == 0 && tid == 0;
    unsigned long long tk = prof ? __builtin_readcyclecounter() : 0ull;
    auto tick = [&](int ph) {
        if (prof) {
            const unsigned long long now = __builtin_readcyclecounter();
            p.prof[ph] += now - tk;
            tk = now;
        }
    };
    for (;;) {
        if (prof) p.prof[8] += 1;
        if constexpr (ADAPTIVE) layer1_shared_t<C::KS1>(wf1, bufA, bufB, biasS, w, l, p.record ? p.rA1 : nullptr, C::RA, iNacc, iDone, iJ, p.cap);
        else layer<C::KS1, true>(wf1, bufA, bufB, biasS, 0, 1, w, l, p.record ? p.rA1 : nullptr, C::RA, iNacc, iDone, iJ, p.cap);
        __syncthreads();
        layer<C::KSH, true>(wf2, bufB, bufA, biasS + 128, 0, nA, w, l, p.record ? p.rA2 : nullptr, C::RA, iNacc, iDone, iJ, p.cap);
        __syncthreads();
        layer<C::KSH, true>(wf3, bufA, bufB, biasS + 256, 0, nA, w, l, p.record ? p.rA3 : nullptr, C::RA, iNacc, iDone, iJ, p.cap);
        __syncthreads();
        layer<C::KSH, false>(wf4, bufB, bufA, biasS + 384, 0, nA, w, l, nullptr, 0, iNacc, iDone, iJ, p.cap);
        __syncthreads();
        tick(0);
        if constexpr (ADAPTIVE) {
            // ---- S1: the Lamba probe point utilde = K + ||G||_F sqrt(dt): input column of the third evaluation ----
            for (int ps = 0; ps < 2; ++ps) {
                const int tr = w * 8 + ps * 4 + rr;
                if (!iDone[tr]) {
                    const float t = fT[tr], dt = fDt[tr];
                    float zz[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float z = (cb + i < D) ? bufA[(cb + i) * LDA + tr] : 0.0f;
                        zz[i] = z * z;
                    }
                    const float Sz = tsum8(zz);
                    const float gs = __builtin_sqrtf(__builtin_fmaf((float)D * p.sig, p.sig, Sz));
                    const float cc = gs * __builtin_sqrtf(dt);
                    if (lane_on) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int c = cb + i;
                            if (c < D) bufA[c * LDA + 64 + tr] = Xs[tr * XLD + c] + cc;
                            else if (c == D) bufA[c * LDA + 64 + tr] = t;
                            else if (c == D + 1) bufA[c * LDA + 64 + tr] = 0.0f;
                        }
                    }
                }
            }
            __syncthreads();
            tick(1);
            layer<C::KS1, true>(wf1, bufA, bufB, biasS, 2, 3, w, l, nullptr, 0, iNacc, iDone, iJ, p.cap);
            __syncthreads();
            layer<C::KSH, true>(wf2, bufB, bufA, biasS + 128, 2, 3, w, l, nullptr, 0, iNacc, iDone, iJ, p.cap);
            __syncthreads();
            layer<C::KSH, true>(wf3, bufA, bufB, biasS + 256, 2, 3, w, l, nullptr, 0, iNacc, iDone, iJ, p.cap);
            __syncthreads();
            layer<C::KSH, false>(wf4, bufB, bufA, biasS + 384, 2, 3, w, l, nullptr, 0, iNacc, iDone, iJ, p.cap);
            __syncthreads();
            tick(2);
        }
        // ---- S2: the step of every live slot (four per wavefront pass); writes the next attempt's input columns ----
        int alldone = 1;
        for (int ps = 0; ps < 2; ++ps) {
            const int tr = w * 8 + ps * 4 + rr;
            if (iDone[tr]) continue;
            const int64_t j = iJ[tr];
            float t = fT[tr], dt = fDt[tr], u = fU[tr], qold = fQold[tr], q11 = fQ11[tr];
            int last = iLast[tr], nacc = iNacc[tr], nrej = iNrej[tr], nstack = iNstack[tr], it = iIter[tr], ndraw = iNdraw[tr];
            uint32_t ev = (uint32_t)iEv[tr];
            int ret = RET_SUCCESS;
            bool fin = false;
            float X[8], dW[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool on = cb + i < D;
                X[i] = on ? Xs[tr * XLD + cb + i] : 0.0f;
                dW[i] = on ? dWs[tr * XLD + cb + i] : 0.0f;
            }
            // the ONE draw a step can consume (a bridged stack piece, OR the fresh remainder, OR the bridge of a rejected step --
            // always with event counter ev): generated once up front, so a wavefront whose four slots take different
            // branches does not run Philox + Box-Muller once per branch
            float nrm[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) nrm[i] = 0.0f;
            if (cb < D) normals8(p.seed, p.iter, (uint32_t)j, ev, m, nrm);
            if (it + 1 > p.maxiters) {
                ret = RET_MAXITERS;
                fin = true;
            } else {
                it += 1;
                const float sq = __builtin_sqrtf(dt);
                float z[8], tmp[8], Xn[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = (cb + i < D) ? bufA[(cb + i) * LDA + tr] : 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) tmp[i] = z[i] * z[i];
                const float Sz = tsum8(tmp);
                const float F = p.lam * Sz;
#pragma unroll
                for (int i = 0; i < 8; ++i) tmp[i] = z[i] * dW[i];
                const float zdW = tsum8(tmp);
#pragma unroll
                for (int i = 0; i < 8; ++i) Xn[i] = __builtin_fmaf(p.sig, dW[i], X[i]);
                const float un = __builtin_fmaf(dt, F, u) + zdW;
                float EE = 0.0f, qq = 1.0f;
                bool accept = true;
                if constexpr (ADAPTIVE) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float y = (cb + i < D) ? bufA[(cb + i) * LDA + 32 + tr] : 0.0f;
                        tmp[i] = y * y;
                    }
                    const float F2 = p.lam * tsum8(tmp);
                    const float Ed = (dt * (F2 - F)) * 0.5f;
                    // non-diagonal noise: the diffusion enters through SCALAR norms (oracle/sde_oracle_impl.h, [UP?]) --
                    // ggprime = (||G(utilde)||_F - ||G(h)||_F) / sqrt(dt), En = ggprime * RMS(dW.^2) / 2, one number added to the
                    // residual of EVERY component; the drift part Ed lives on the u row only
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float y = (cb + i < D) ? bufA[(cb + i) * LDA + 64 + tr] : 0.0f;
                        tmp[i] = y * y;
                    }
                    const float dsig = (float)D * p.sig;
                    const float gs3 = __builtin_sqrtf(__builtin_fmaf(dsig, p.sig, tsum8(tmp)));
                    const float gs = __builtin_sqrtf(__builtin_fmaf(dsig, p.sig, Sz));
                    const float ggp = (gs3 - gs) / sq;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float w2 = dW[i] * dW[i];
                        tmp[i] = w2 * w2;
                    }
                    const float nW2 = __builtin_sqrtf(tsum8(tmp) / (float)D);
                    const float En = (ggp * nW2) * 0.5f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a0 = fabsf(X[i]), a1 = fabsf(Xn[i]);
                        const float r = En / __builtin_fmaf((a0 > a1 ? a0 : a1), p.reltol, p.abstol);
                        tmp[i] = (cb + i < D) ? r * r : 0.0f;
                    }
                    const float sX = tsum8(tmp);
                    const float au = fabsf(u), aun = fabsf(un);
                    const float res = (Ed + En) / __builtin_fmaf((au > aun ? au : aun), p.reltol, p.abstol);
                    EE = __builtin_sqrtf((sX + res * res) / (float)(D + 1));
                    if (EE == 0.0f) {
                        qq = 1.0f / p.qmax;
                        q11 = 1.0f;
                    } else {
                        q11 = (float)fastpow((double)EE, (double)p.beta1);
                        qq = q11 / (float)fastpow((double)qold, (double)p.beta2);
                        qq = qq / p.gamma;
                        const float lo = 1.0f / p.qmax, hi = 1.0f / p.qmin;
                        if (qq > hi) qq = hi;
                        if (qq < lo) qq = lo;
                    }
                    accept = EE <= 1.0f;
                    if (EE != EE) { ret = RET_UNSTABLE; fin = true; }
                }
                tick(5);
                if (!fin && accept) {
                    {
                        if (p.record && nacc >= p.cap && m == 0) iOvf[tr] = 1;  // (read back by the same lanes at the end of the trajectory)
                        if (p.record && lane_on && nacc < p.cap) {
                            const size_t col = (size_t)j * p.cap + nacc;
                            float* rx = p.rXin + col * C::RX + cb;
                            const float coef = (2.0f * p.lam) * dt;
                            if (cb + 8 <= D) {
                                *reinterpret_cast<float4*>(rx) = float4{X[0], X[1], X[2], X[3]};
                                *reinterpret_cast<float4*>(rx + 4) = float4{X[4], X[5], X[6], X[7]};
                                float* re = p.rE4 + col * C::RE + cb;
                                *reinterpret_cast<float4*>(re) = float4{__builtin_fmaf(coef, z[0], dW[0]), __builtin_fmaf(coef, z[1], dW[1]),
                                                                        __builtin_fmaf(coef, z[2], dW[2]), __builtin_fmaf(coef, z[3], dW[3])};
                                *reinterpret_cast<float4*>(re + 4) = float4{__builtin_fmaf(coef, z[4], dW[4]), __builtin_fmaf(coef, z[5], dW[5]),
                                                                            __builtin_fmaf(coef, z[6], dW[6]), __builtin_fmaf(coef, z[7], dW[7])};
                            } else {
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    const int c = cb + i;
                                    if (c < D) { rx[i] = X[i]; p.rE4[col * C::RE + c] = __builtin_fmaf(coef, z[i], dW[i]); }
                                    else if (c == D) rx[i] = t;
                                }
                            }
                        }
                        nacc += 1;
                        t = last ? p.t1 : t + dt;
                        u = un;
                        float badf = (un != un) ? 1.0f : 0.0f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { X[i] = Xn[i]; badf += (Xn[i] != Xn[i]) ? 1.0f : 0.0f; }
                        const bool bad = row_tree_sum(badf) > 0.0f;
                        if (bad) {
                            ret = RET_UNSTABLE;
                            fin = true;
                        } else if (t >= p.t1) {
                            fin = true;
                        } else {
                            float dtn = dt;
                            if constexpr (ADAPTIVE) {
                                qold = EE > p.qoldinit ? EE : p.qoldinit;
                                dtn = dt / qq;
                                if (dtn > p.dtmax) dtn = p.dtmax;
                            }
                            const float rem = p.t1 - t;
                            last = 0;
                            if (dtn >= rem) { dtn = rem; last = 1; }
                            // the increment over [t, t + dtn]: whole stack pieces, the last one bridged, the rest fresh
                            float acch = 0.0f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) dW[i] = 0.0f;
                            float* sW = p.stackW + ((size_t)j * STACK) * XLD + cb;
                            float* sL = sLs + tr * STACK;
                            while (nstack > 0 && acch < dtn) {
                                float* top = sW + (size_t)(nstack - 1) * XLD;
                                const float L = sL[nstack - 1];
                                float tw[8];
#pragma unroll
                                for (int i = 0; i < 8; ++i) tw[i] = 0.0f;
                                if (cb < D) {
                                    const float4 a = *reinterpret_cast<float4*>(top), b = *reinterpret_cast<float4*>(top + 4);
                                    tw[0] = a.x; tw[1] = a.y; tw[2] = a.z; tw[3] = a.w; tw[4] = b.x; tw[5] = b.y; tw[6] = b.z; tw[7] = b.w;
                                }
                                if (acch + L <= dtn) {
                                    acch = acch + L;
#pragma unroll
                                    for (int i = 0; i < 8; ++i) dW[i] = dW[i] + tw[i];
                                    nstack -= 1;
                                } else {
                                    const float rl = dtn - acch, fr = rl / L;
                                    ev += 1; ndraw += 1;
                                    const float sd = __builtin_sqrtf((1.0f - fr) * rl);
                                    if (cb < D) {
                                        float wv[8];
#pragma unroll
                                        for (int i = 0; i < 8; ++i) {
                                            wv[i] = __builtin_fmaf(fr, tw[i], sd * nrm[i]);
                                            dW[i] = dW[i] + wv[i];
                                        }
                                        *reinterpret_cast<float4*>(top) = float4{tw[0] - wv[0], tw[1] - wv[1], tw[2] - wv[2], tw[3] - wv[3]};
                                        *reinterpret_cast<float4*>(top + 4) = float4{tw[4] - wv[4], tw[5] - wv[5], tw[6] - wv[6], tw[7] - wv[7]};
                                    }
                                    if (m == 0) sL[nstack - 1] = L - rl;
                                    acch = dtn;
                                }
                            }
                            if (acch < dtn) {
                                const float sd = __builtin_sqrtf(dtn - acch);
                                if (cb < D) {
#pragma unroll
                                    for (int i = 0; i < 8; ++i) dW[i] = __builtin_fmaf(sd, nrm[i], dW[i]);
                                }
                                ev += 1; ndraw += 1;
                            }
                            dt = dtn;
                        }
                    }
                } else if (!fin) {
                    nrej += 1;
                    float den = q11 / p.gamma;
                    const float iq = 1.0f / p.qmin;
                    if (iq < den) den = iq;
                    const float dtn = dt / den, fr = dtn / dt;
                    if (nstack >= STACK) {
                        ret = RET_STACK_OVERFLOW;
                        fin = true;
                    } else {
                        const float sd = __builtin_sqrtf((1.0f - fr) * dtn);
                        if (cb < D) {
                            float wv[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) wv[i] = __builtin_fmaf(fr, dW[i], sd * nrm[i]);
                            float* top = p.stackW + ((size_t)j * STACK + nstack) * XLD + cb;
                            *reinterpret_cast<float4*>(top) = float4{dW[0] - wv[0], dW[1] - wv[1], dW[2] - wv[2], dW[3] - wv[3]};
                            *reinterpret_cast<float4*>(top + 4) = float4{dW[4] - wv[4], dW[5] - wv[5], dW[6] - wv[6], dW[7] - wv[7]};
#pragma unroll
                            for (int i = 0; i < 8; ++i) dW[i] = wv[i];
                        }
                        ev += 1; ndraw += 1;
                        if (m == 0) sLs[tr * STACK + nstack] = dt - dtn;
                        nstack += 1;
                        dt = dtn;
                        last = 0;
                    }
                }
            }
            tick(6);
            if (fin) {
                // loss term of this trajectory: (g(X_T) - u_T)^2, g(X) = log(0.5 + 0.5 |X|^2)  (lambaem.jl:14)
                float lj = 0.0f, ub = 0.0f;
                if (ret == RET_SUCCESS && iOvf[tr]) ret = RET_STORE_OVERFLOW;  // out of the loss (+Inf) and of the gradient; p.nacc = its true count
                if (ret == RET_SUCCESS) {
                    float tmp[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) tmp[i] = X[i] * X[i];
                    const float S = tsum8(tmp);
                    const float g = (float)dlog((double)__builtin_fmaf(0.5f, S, 0.5f));
                    const float e = g - u;
                    lj = e * e;
                    ub = (-2.0f * e) / (float)p.M;
                }
                if (p.XT && cb < D) {
                    float* xo = p.XT + (size_t)j * D + cb;
#pragma unroll
                    for (int i = 0; i < 8; ++i) if (cb + i < D) xo[i] = X[i];
                }
                if (m == 0) {
                    if (p.uT) p.uT[j] = u;
                    p.loss_traj[j] = (double)lj;
                    p.ubar[j] = ub;
                    p.retcode[j] = ret;
                    p.nacc[j] = nacc;
                    if (p.stats) {
                        int64_t* s = p.stats + (size_t)j * 4;
                        s[0] = (int64_t)it * (ADAPTIVE ? 3 : 1); s[1] = nacc; s[2] = nrej; s[3] = ndraw;
                    }
                }
                // the slot takes the next trajectory of the ensemble, if any
                int jn = 0;
                if (m == 0) jn = atomicAdd(p.queue, 1);
                jn = __shfl(jn, l & 48, 64);
                if (jn < p.M) {
                    start_slot(tr, jn);
                    alldone = 0;
                } else if (m == 0) {
                    iDone[tr] = 1;
                }
            } else {
                alldone = 0;
                if (lane_on) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int c = cb + i;
                        if (c < D) {
                            Xs[tr * XLD + c] = X[i];
                            dWs[tr * XLD + c] = dW[i];
                            bufA[c * LDA + tr] = X[i];
                        } else if (c == D) {
                            bufA[c * LDA + tr] = t;
                            bufA[c * LDA + 32 + tr] = t + dt;
                        } else if (c == D + 1) {
                            bufA[c * LDA + tr] = 0.0f;
                            bufA[c * LDA + 32 + tr] = 0.0f;
                        }
                    }
                }
                if (m == 0) {
                    fT[tr] = t; fDt[tr] = dt; fU[tr] = u; fQold[tr] = qold; fQ11[tr] = q11;
                    iLast[tr] = last; iNacc[tr] = nacc; iNrej[tr] = nrej; iNstack[tr] = nstack; iIter[tr] = it;
                    iEv[tr] = (int)ev; iNdraw[tr] = ndraw;
                }
            }
        }
        tick(3);
        const bool stop = __syncthreads_and(alldone);
        tick(4);
        if (stop) break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// sigma^T grad u chain for n input columns (parity aid: the matrix-core layers against the oracle's fmaf chains)
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int H>
__global__ void __launch_bounds__(256) hjb_net_kernel(const float* thsg, int64_t n, const float* xin, float* z) {
    using C = Cfg<D, H>;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* bufA = sm;
    float* bufB = bufA + 128 * LDA;
    float* biasS = bufB + 128 * LDA;
    __shared__ int zero32[32];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    float wf1[C::KS1], wf2[C::KSH], wf3[C::KSH], wf4[C::KSH];
    load_fwd_weights<D, H>(thsg, w, l, wf1, wf2, wf3, wf4);
    load_bias_table<D, H>(thsg, biasS, tid, 256);
    if (tid < 32) zero32[tid] = 0;
    for (int64_t base = (int64_t)blockIdx.x * 32; base < n; base += (int64_t)gridDim.x * 32) {
        __syncthreads();
        for (int i = tid; i < 128 * 32; i += 256) {
            const int k = i >> 5, c = i & 31;
            bufA[k * LDA + c] = (k < C::DIN && base + c < n) ? xin[(size_t)(base + c) * C::DIN + k] : 0.0f;
        }
        __syncthreads();
        layer<C::KS1, true>(wf1, bufA, bufB, biasS, 0, 1, w, l, nullptr, 0, zero32, zero32, zero32, 0);
        __syncthreads();
        layer<C::KSH, true>(wf2, bufB, bufA, biasS + 128, 0, 1, w, l, nullptr, 0, zero32, zero32, zero32, 0);
        __syncthreads();
        layer<C::KSH, true>(wf3, bufA, bufB, biasS + 256, 0, 1, w, l, nullptr, 0, zero32, zero32, zero32, 0);
        __syncthreads();
        layer<C::KSH, false>(wf4, bufB, bufA, biasS + 384, 0, 1, w, l, nullptr, 0, zero32, zero32, zero32, 0);
        __syncthreads();
        for (int i = tid; i < D * 32; i += 256) {
            const int k = i >> 5, c = i & 31;
            if (base + c < n) z[(size_t)(base + c) * D + k] = bufA[k * LDA + c];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// prep: u0 = u0 net(x0) and the initial dt (sde_determine_initdt), one block of 128 threads, thread j = neuron j
// (VALU fmaf chains: the same operation sequence as the matrix cores and the oracle)
// ---------------------------------------------------------------------------------------------------------------------
template <int IN, int OUT, bool RELU>
__device__ __forceinline__ void dense_valu(const float* W, const float* b, const float* a, float* o, int j) {
    if (j < OUT) {
        float acc = b[j];
        for (int k = 0; k < IN; ++k) acc = __builtin_fmaf(W[j + (size_t)k * OUT], a[k], acc);
        o[j] = RELU ? relu_enc(acc) : acc;
    }
}

template <int D, int H>
__device__ __forceinline__ void sg_valu(const float* th, const float* xin, float* a1, float* a2, float* a3, float* z, int j) {
    using C = Cfg<D, H>;
    dense_valu<C::DIN, H, true>(th + C::W1, th + C::B1, xin, a1, j);
    __syncthreads();
    dense_valu<H, H, true>(th + C::W2, th + C::B2, a1, a2, j);
    __syncthreads();
    dense_valu<H, H, true>(th + C::W3, th + C::B3, a2, a3, j);
    __syncthreads();
    dense_valu<H, D, false>(th + C::W4, th + C::B4, a3, z, j);
    __syncthreads();
}

template <int D, int H>
__global__ void __launch_bounds__(128) hjb_prep_kernel(const HjbParams p) {
    using C = Cfg<D, H>;
    __shared__ float xin[128], a1[128], a2[128], a3[128], z0[128], zB[128], x0s[128], sc[4];
    const int j = threadIdx.x;
    const float* thu = p.theta;
    const float* thsg = p.theta + C::NPU;
    if (j < D) x0s[j] = p.x0[j];
    __syncthreads();
    dense_valu<D, H, true>(thu + C::U1, thu + C::C1, x0s, a1, j);
    __syncthreads();
    dense_valu<H, H, true>(thu + C::U2, thu + C::C2, a1, a2, j);
    __syncthreads();
    if (j == 0) {
        float acc = thu[C::C3];
        for (int k = 0; k < H; ++k) acc = __builtin_fmaf(thu[C::U3 + k], a2[k], acc);
        sc[0] = acc;
        p.prep[0] = acc;
    }
    if (j < H) { p.prep[2 + j] = a1[j]; p.prep[2 + H + j] = a2[j]; }
    __syncthreads();
    if (!(p.adaptive && !(p.dt_user > 0.0f))) {
        if (j == 0) p.prep[1] = p.dt_user;
        return;
    }
    const float u0 = sc[0];
    if (j < D) xin[j] = x0s[j];
    if (j == D) xin[D] = p.t0;
    __syncthreads();
    sg_valu<D, H>(thsg, xin, a1, a2, a3, z0, j);
    // wave 0: lanes hold components 2l, 2l+1
    const int l = j & 63, c0 = 2 * l, c1 = 2 * l + 1;
    const bool v0 = c0 < D, v1 = c1 < D;
    const float sku = __builtin_fmaf(fabsf(u0), p.reltol, p.abstol);
    float dt0 = 0.0f, d1 = 0.0f, F0 = 0.0f;
    if (j < 64) {
        const float xa = v0 ? x0s[c0] : 0.0f, xb = v1 ? x0s[c1] : 0.0f;
        const float ska = __builtin_fmaf(fabsf(xa), p.reltol, p.abstol), skb = __builtin_fmaf(fabsf(xb), p.reltol, p.abstol);
        const float qa = __fdiv_rn(xa, ska), qb = __fdiv_rn(xb, skb);
        const float qu = __fdiv_rn(u0, sku);
        const float d0 = __builtin_sqrtf(__fdiv_rn(tsum2(v0 ? qa * qa : 0.0f, v1 ? qb * qb : 0.0f) + qu * qu, (float)(D + 1)));
        const float za = v0 ? z0[c0] : 0.0f, zb = v1 ? z0[c1] : 0.0f;
        F0 = p.lam * tsum2(za * za, zb * zb);
        const float s3 = 3.0f * p.sig;
        const float ra = __fdiv_rn(s3, ska), rb = __fdiv_rn(s3, skb);
        const float sA = tsum2(v0 ? ra * ra : 0.0f, v1 ? rb * rb : 0.0f);
        const float ga = __fdiv_rn(fabsf(F0) + 3.0f * fabsf(za), sku), gb = __fdiv_rn(fabsf(F0) + 3.0f * fabsf(zb), sku);
        d1 = __builtin_sqrtf(__fdiv_rn(sA + tsum2(v0 ? ga * ga : 0.0f, v1 ? gb * gb : 0.0f), (float)((D + 1) * D)));
        dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : __fdiv_rn(__fdiv_rn(d0, d1), 100.0f);
        if (dt0 > p.dtmax) dt0 = p.dtmax;
        if (j == 0) { sc[1] = dt0; sc[2] = d1; sc[3] = F0; }
    }
    __syncthreads();
    dt0 = sc[1]; d1 = sc[2]; F0 = sc[3];
    if (j == D) xin[D] = p.t0 + dt0;
    __syncthreads();
    sg_valu<D, H>(thsg, xin, a1, a2, a3, zB, j);
    if (j < 64) {
        const float xa = v0 ? x0s[c0] : 0.0f, xb = v1 ? x0s[c1] : 0.0f;
        const float ska = __builtin_fmaf(fabsf(xa), p.reltol, p.abstol), skb = __builtin_fmaf(fabsf(xb), p.reltol, p.abstol);
        const float za = v0 ? z0[c0] : 0.0f, zb = v1 ? z0[c1] : 0.0f;
        const float ya = v0 ? zB[c0] : 0.0f, yb = v1 ? zB[c1] : 0.0f;
        const float F1 = p.lam * tsum2(ya * ya, yb * yb);
        const float s6 = 6.0f * p.sig;
        const float ra = __fdiv_rn(s6, ska), rb = __fdiv_rn(s6, skb);
        const float sA = tsum2(v0 ? ra * ra : 0.0f, v1 ? rb * rb : 0.0f);
        const float dF = fabsf(F1 - F0);
        auto gk = [&](float zz, float yy) {
            const float g0 = 3.0f * zz, g1 = 3.0f * yy;
            const float m1 = fabsf(g0 - g1), m2 = fabsf(g0 + g1);
            return __fdiv_rn(dF + (m1 > m2 ? m1 : m2), sku);
        };
        const float ga = gk(za, ya), gb = gk(zb, yb);
        const float d2 = __fdiv_rn(__builtin_sqrtf(__fdiv_rn(sA + tsum2(v0 ? ga * ga : 0.0f, v1 ? gb * gb : 0.0f), (float)((D + 1) * D))), dt0);
        const float mx = d1 > d2 ? d1 : d2;
        float dt1;
        if (mx <= 1e-15f) {
            dt1 = dt0 * 1e-3f;
            if (dt1 < 1e-6f) dt1 = 1e-6f;
        } else {
            dt1 = (float)dpow10(-(2.0 + dlog10((double)mx)) / 1.0);
        }
        float r = 100.0f * dt0;
        if (dt1 < r) r = dt1;
        if (p.dtmax < r) r = p.dtmax;
        if (j == 0) p.prep[1] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: the columns (trajectory, accepted step) are independent (X never depends on theta): per tile of 32 columns
//   delta4 = ubar (2 lambda dt z + dW), delta_l = relu'(a_l) .* (W_{l+1}^T delta_{l+1})  (matrix cores, transposed weight
//   fragments in registers), then dW_l += delta_l [a_{l-1}; 1]^T with the columns along K (accumulators live in
//   registers across all the block's tiles); per-block partial gradients are summed by hjb_reduce_kernel in block order.
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int H>
constexpr int bwd_lds_floats() { return 8 * 128 * LDT; }

template <int KS>
__device__ __forceinline__ void layer_bwd(const float (&wt)[KS], const float* din, float* dout, const float* mask, int w, int l) {
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float* bp = din + (l >> 5) * LDT + (l & 31);
    constexpr int PF = KS < 6 ? KS : 6;   // operand prefetch distance (see layer())
    float bq[PF];
    static_for<0, PF>([&](auto ic) { bq[ic] = bp[2 * decltype(ic)::value * LDT]; });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, KS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const float b = bq[s % PF];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[s], b, acc, 0, 0, 0);
        if constexpr (s + PF < KS) bq[s % PF] = bp[2 * (s + PF) * LDT];
        __builtin_amdgcn_sched_barrier(0);
    });
    const int rbase = 32 * w + 4 * (l >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int idx = (rbase + (r & 3) + 8 * (r >> 2)) * LDT + (l & 31);
        dout[idx] = relu_on(mask[idx]) ? acc[r] : 0.0f;
    }
}

// G[nt] += delta (rows 32w.., columns as K) x act^T (rows nt*32.., columns as K)
__device__ __forceinline__ void outer_acc(v16f (&G)[4], const float* dl, const float* act, int w, int l) {
    const float* ap = dl + (32 * w + (l & 31)) * LDT + (l >> 5);
    const float* bp = act + (l & 31) * LDT + (l >> 5);
    // all operands of the tile's 16 k-steps are fetched first (4 independent accumulator chains keep the pipe full)
    float av[16], bv[4][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        av[s] = ap[2 * s];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bv[nt][s] = bp[nt * 32 * LDT + 2 * s];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) G[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[nt][s], G[nt], 0, 0, 0);
    }
}

#ifndef HJB_BWD_LU
#define HJB_BWD_LU 2
#endif
template <int D, int H>
__global__ void __launch_bounds__(256) hjb_bwd_kernel(const HjbParams p) {
    using C = Cfg<D, H>;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xinT = sm;
    float* a1T = xinT + 128 * LDT;
    float* a2T = a1T + 128 * LDT;
    float* a3T = a2T + 128 * LDT;
    float* d4T = a3T + 128 * LDT;
    float* d3T = d4T + 128 * LDT;
    float* d2T = d3T + 128 * LDT;
    float* d1T = d2T + 128 * LDT;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const float* th = p.theta + C::NPU;
    float wt4[C::KS4T], wt3[C::KSH], wt2[C::KSH];
    {
        const int i = 32 * w + (l & 31), kk = l >> 5;
        static_for<0, C::KS4T>([&](auto sc) {
            const int k = 2 * decltype(sc)::value + kk;
            wt4[sc] = (i < H && k < D) ? th[C::W4 + k + i * D] : 0.0f;
        });
        static_for<0, C::KSH>([&](auto sc) {
            const int k = 2 * decltype(sc)::value + kk;
            const bool in = i < H && k < H;
            wt3[sc] = in ? th[C::W3 + k + i * H] : 0.0f;
            wt2[sc] = in ? th[C::W2 + k + i * H] : 0.0f;
        });
    }
    v16f G1[4], G2[4], G3[4], G4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { G1[nt][r] = 0.0f; G2[nt][r] = 0.0f; G3[nt][r] = 0.0f; G4[nt][r] = 0.0f; }
    for (int i = tid; i < 8 * 128 * LDT; i += 256) sm[i] = 0.0f;
    __syncthreads();

    // debug phase clocks (UDE_HJB_PROF): [10] tile loads, [11] three transposed layers, [12] four outer products, [13] tiles
    const bool prof = p.prof != nullptr && blockIdx.x == 0 && tid == 0;
    unsigned long long tk = prof ? __builtin_readcyclecounter() : 0ull;
    auto tick = [&](int ph) {
        if (prof) {
            const unsigned long long now = __builtin_readcyclecounter();
            p.prof[ph] += now - tk;
            tk = now;
        }
    };
    for (int64_t j = blockIdx.x; j < p.M; j += gridDim.x) {
        if (p.retcode[j] != RET_SUCCESS) continue;
        const int nacc = p.nacc[j];
        const float ub = p.ubar[j];
        for (int t0 = 0; t0 < nacc; t0 += 32) {
            if (prof) p.prof[13] += 1;
            tick(14);
            const int nv = nacc - t0 < 32 ? nacc - t0 : 32;
            const size_t col0 = (size_t)j * p.cap + t0;
            __syncthreads();  // the previous tile's readers are done
            tick(9);
            // tiles [k][column]: thread walks k (coalesced in HBM), one column per pass.  LU passes are loaded together and
            // stored together: a pass-by-pass loop waited for its five loads before the next pass could issue its own
            // (16 dependent HBM round trips per tile = 41 % of this kernel, measured with UDE_HJB_PROF)
            constexpr int LU = HJB_BWD_LU;
            static_assert((32 * 128 / 256) % LU == 0, "passes per tile");
            for (int i0 = tid; i0 < 32 * 128; i0 += 256 * LU) {
                float vx[LU], v1[LU], v2[LU], v3[LU], v4[LU];
#pragma unroll
                for (int u = 0; u < LU; ++u) {  // UNCONDITIONAL loads from clamped (always valid) addresses: all 5 x LU in flight together
                    const int i = i0 + 256 * u;
                    const int c = i >> 7, k = i & 127;
                    const size_t col = col0 + (c < nv ? c : 0);
                    vx[u] = p.rXin[col * C::RX + (k < C::DIN ? k : 0)];
                    const int kh = k < H ? k : 0;
                    v1[u] = p.rA1[col * C::RA + kh]; v2[u] = p.rA2[col * C::RA + kh]; v3[u] = p.rA3[col * C::RA + kh];
                    v4[u] = p.rE4[col * C::RE + (k < D ? k : 0)];
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const int i = i0 + 256 * u;
                    const int c = i >> 7, k = i & 127;
                    const bool on = c < nv;
                    vx[u] = !on ? 0.0f : k < C::DIN ? vx[u] : k == C::DIN ? 1.0f : 0.0f;  // (k == DIN: bias slot)
                    const float one_h = (on && k == H) ? 1.0f : 0.0f;
                    v1[u] = (on && k < H) ? v1[u] : one_h;
                    v2[u] = (on && k < H) ? v2[u] : one_h;
                    v3[u] = (on && k < H) ? v3[u] : one_h;
                    v4[u] = (on && k < D) ? ub * v4[u] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const int i = i0 + 256 * u;
                    const int c = i >> 7, k = i & 127;
                    xinT[k * LDT + c] = vx[u]; a1T[k * LDT + c] = v1[u]; a2T[k * LDT + c] = v2[u]; a3T[k * LDT + c] = v3[u]; d4T[k * LDT + c] = v4[u];
                }
            }
            tick(15);
            __syncthreads();
            tick(10);
            layer_bwd<C::KS4T>(wt4, d4T, d3T, a3T, w, l);
            __syncthreads();
            layer_bwd<C::KSH>(wt3, d3T, d2T, a2T, w, l);
            __syncthreads();
            layer_bwd<C::KSH>(wt2, d2T, d1T, a1T, w, l);
            __syncthreads();
            tick(11);
            // (rows H.. of the delta tiles: the transposed fragments are zero there, so the masked value is 0)
            outer_acc(G1, d1T, xinT, w, l);
            outer_acc(G2, d2T, a1T, w, l);
            outer_acc(G3, d3T, a2T, w, l);
            outer_acc(G4, d4T, a3T, w, l);
            tick(12);
        }
    }
    // ---- this block's partial gradient of theta_sg ----
    float* row = p.part + (size_t)blockIdx.x * C::NP;
    const int rbase = 32 * w + 4 * (l >> 5);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int k = nt * 32 + (l & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jr = rbase + (r & 3) + 8 * (r >> 2);
            if (jr < H) {
                if (k < C::DIN) row[C::W1 + jr + k * H] = G1[nt][r];
                else if (k == C::DIN) row[C::B1 + jr] = G1[nt][r];
                if (k < H) { row[C::W2 + jr + k * H] = G2[nt][r]; row[C::W3 + jr + k * H] = G3[nt][r]; }
                else if (k == H) { row[C::B2 + jr] = G2[nt][r]; row[C::B3 + jr] = G3[nt][r]; }
            }
            if (jr < D) {
                if (k < H) row[C::W4 + jr + k * D] = G4[nt][r];
                else if (k == H) row[C::B4 + jr] = G4[nt][r];
            }
        }
    }
}

// grad_sg[i] = sum over blocks (fixed order, double accumulation); loss = mean_j loss_traj (Inf if a trajectory failed);
// the u0 chain's gradient from U = sum_j ubar_j (one extra block)
template <int D, int H>
__global__ void __launch_bounds__(256) hjb_reduce_kernel(const HjbParams p, int nblocks) {
    using C = Cfg<D, H>;
    __shared__ double shd[256];
    __shared__ int shi[256];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < (int)gridDim.x - 1) {
        const int i = blockIdx.x * 256 + tid;
        if (p.grad && i < C::NP) {
            double s = 0.0;
            for (int b = 0; b < nblocks; ++b) s += (double)p.part[(size_t)b * C::NP + i];
            p.grad[C::NPU + i] = (float)s;
        }
        return;
    }
    // last block: loss, failure count, U, and the u0 chain's gradient
    double ls = 0.0, us = 0.0;
    int nf = 0;
    for (int64_t j = tid; j < p.M; j += 256) {
        ls += p.loss_traj[j];
        us += (double)p.ubar[j];
        nf += p.retcode[j] != RET_SUCCESS;
    }
    shd[tid] = ls; shi[tid] = nf;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (tid < m) { shd[tid] += shd[tid + m]; shi[tid] += shi[tid + m]; }
        __syncthreads();
    }
    const double lsum = shd[0];
    const int nfail = shi[0];
    __syncthreads();
    shd[tid] = us;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (tid < m) shd[tid] += shd[tid + m];
        __syncthreads();
    }
    const float U = (float)shd[0];
    if (tid == 0) {
        *p.loss = nfail ? __builtin_inf() : lsum / (double)p.M;
        if (p.nfail) *p.nfail = nfail;
    }
    if (!p.grad) return;
    __shared__ float d2[128], d1[128];
    const float* thu = p.theta;
    const float* a1 = p.prep + 2;
    const float* a2 = p.prep + 2 + H;
    if (tid < H) d2[tid] = relu_on(a2[tid]) ? thu[C::U3 + tid] * U : 0.0f;
    __syncthreads();
    if (tid < H) {
        float acc = 0.0f;
        for (int k = 0; k < H; ++k) acc = __builtin_fmaf(thu[C::U2 + k + (size_t)tid * H], d2[k], acc);
        d1[tid] = relu_on(a1[tid]) ? acc : 0.0f;
    }
    __syncthreads();
    float* g = p.grad;
    for (int i = tid; i < H * D; i += 256) g[C::U1 + i] = d1[i % H] * p.x0[i / H];
    for (int i = tid; i < H * H; i += 256) g[C::U2 + i] = d2[i % H] * a1[i / H];
    if (tid < H) { g[C::C1 + tid] = d1[tid]; g[C::C2 + tid] = d2[tid]; g[C::U3 + tid] = U * a2[tid]; }
    if (tid == 0) g[C::C3] = U;
}

}  // namespace hjb
}  // namespace ude
