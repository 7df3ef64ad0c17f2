// udecore.hip -- C ABI (include/udecore.h) over the fused HIP kernels.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/udecore.h"
#include "ude_registry.h"

using namespace ude;

namespace ude {
// A trajectory whose retcode is not Success contributes nothing to the gradient; the reference would see an Inf loss (or an
// error) from such a solve, so the ensemble loss becomes +Inf and the count of failed trajectories is published -- an
// optimiser never silently trains on a partial objective.  Column sums: fixed strided partial sums + fixed LDS tree,
// accumulated in double for both scalar types (deterministic for a given launch shape, independent of timing).
// the three reductions of a gradient call in ONE launch (block i < ncols: column i of the partial-gradient matrix; block
// ncols: the loss sum, then the failure count and the +Inf rule) -- same sums, same trees, two launches less
template <class T>
__global__ void finish_kernel(const T* part, int64_t nrows, int32_t ncols, T* grad_out, const T* loss_traj, int64_t N, T* loss_out,
                              const int32_t* retcode, int32_t* nfail_out) {
    __shared__ double sh[256];
    __shared__ int shi[256];
    const int i = blockIdx.x;
    if (i < ncols) {
        double s = 0.0;
        for (int64_t w0 = threadIdx.x; w0 < nrows; w0 += 256 * 4) {
            double pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t w = w0 + 256 * u;
                pv[u] = (double)part[(size_t)(w < nrows ? w : 0) * ncols + i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (w0 + 256 * u < nrows) s += pv[u];
        }
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int m = 128; m > 0; m >>= 1) {
            if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
            __syncthreads();
        }
        if (threadIdx.x == 0) grad_out[i] = (T)sh[0];
        return;
    }
    double s = 0.0;
    int c = 0;
    // (eight strided elements requested together, added in the same order as a plain loop: a one-at-a-time loop was 40
    //  dependent HBM round trips for 10 000 trajectories -- 16 us of a 1.6 ms step)
    for (int64_t j0 = threadIdx.x; j0 < N; j0 += 256 * 8) {
        double lv[8];
        int rv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t j = j0 + 256 * u;
            const int64_t jj = j < N ? j : 0;
            lv[u] = loss_out ? (double)loss_traj[jj] : 0.0;
            rv[u] = retcode[jj];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (j0 + 256 * u < N) {
                s += lv[u];
                c += rv[u] != 0;
            }
        }
    }
    sh[threadIdx.x] = s;
    shi[threadIdx.x] = c;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if ((int)threadIdx.x < m) { sh[threadIdx.x] += sh[threadIdx.x + m]; shi[threadIdx.x] += shi[threadIdx.x + m]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (nfail_out) *nfail_out = shi[0];
        if (loss_out) *loss_out = shi[0] > 0 ? (T)__builtin_inf() : (T)sh[0];
    }
}

// wide partial-gradient matrices (SEIR: 6252 rows x 4481 columns, neural ODE: x 9287): the column-per-block layout above reads
// 8 bytes per 64-byte line (224 MB of data cost 350 us).  Here a block of 1024 threads owns 32 adjacent columns: 32 lanes read
// 256 contiguous bytes of a row, 32 row-lanes stride over the rows (eight rows requested together), partials meet in LDS and
// are added in ascending row-lane order.  Block gridDim.x - 1 does the loss / failure bookkeeping as in finish_kernel.
template <class T>
__global__ void __launch_bounds__(1024) finish_wide_kernel(const T* part, int64_t nrows, int32_t ncols, T* grad_out, const T* loss_traj, int64_t N,
                                                           T* loss_out, const int32_t* retcode, int32_t* nfail_out) {
    __shared__ double sh[32][33];
    __shared__ int shi[1024];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < (int)gridDim.x - 1) {
        const int cx = tid & 31, ry = tid >> 5;
        const int col = blockIdx.x * 32 + cx;
        const int cc = col < ncols ? col : ncols - 1;
        double s = 0.0;
        for (int64_t w0 = ry; w0 < nrows; w0 += 32 * 8) {
            double pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t w = w0 + 32 * u;
                pv[u] = (double)part[(size_t)(w < nrows ? w : 0) * ncols + cc];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (w0 + 32 * u < nrows) s += pv[u];
        }
        sh[ry][cx] = s;
        __syncthreads();
        if (ry == 0 && col < ncols) {
            double t = sh[0][cx];
            for (int q = 1; q < 32; ++q) t += sh[q][cx];
            grad_out[col] = (T)t;
        }
        return;
    }
    double s = 0.0;
    int c = 0;
    for (int64_t j = tid; j < N; j += 1024) {
        if (loss_out) s += (double)loss_traj[j];
        c += retcode[j] != 0;
    }
    double* shd = &sh[0][0];  // 1056 doubles
    shd[tid] = s;
    shi[tid] = c;
    __syncthreads();
    for (int m = 512; m > 0; m >>= 1) {
        if (tid < m) { shd[tid] += shd[tid + m]; shi[tid] += shi[tid + m]; }
        __syncthreads();
    }
    if (tid == 0) {
        if (nfail_out) *nfail_out = shi[0];
        if (loss_out) *loss_out = shi[0] > 0 ? (T)__builtin_inf() : (T)shd[0];
    }
}

__global__ void fastpow_kernel(const double* x, const double* y, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fastpow(x[i], y[i]);
}

// ARITH-SPEC primitives evaluated on the device, for bitwise comparison with the oracle
__global__ void math_kernel(int op, const double* x, const double* y, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y[i];
    double r;
    switch (op) {
        case 0: r = fastpow(a, b); break;
        case 1: r = dexp(a); break;
        case 2: r = dtanh(a); break;
        case 3: r = dlog10(a); break;
        case 4: r = dpow10(a); break;
        case 5: r = sqrt(a); break;
        case 6: r = a / b; break;
        case 8: r = dlog(a); break;
        case 9: r = dpow(a, b); break;
        default: r = __builtin_fma(a, b, a); break;
    }
    out[i] = r;
}

// op 10: the operation sequence of the FP64 matrix cores.  Per wavefront: lane l supplies A[l%16][l/16] = x, B[l/16][l%16] = y,
// C = 0; out = register 0 of D, i.e. D[l/16][l%16].  tests/ check it against the ascending fused chain ARITH-SPEC assumes.
__global__ void mfma_probe_kernel(const double* x, const double* y, double* out) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], y[i], v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
    out[i] = d[0];
}

}  // namespace ude

#include "ude_ctx.h"
#ifdef UDE_DEBUG_HOOKS
// Debug build only (libudecore_dbg.so, build.py): the shipping library contains none of this and reads no environment
// variable on its launch path.  ude_poison_gen.h is generated into build/ by tools/gen_poison_header.py at build time.
// UDE_EXP_POISON=<kind>[,<what>]: between the forward and the backward kernel every register and every LDS byte of the chip
// is set to a known pattern -- a kernel whose result changes with the pattern reads state it never wrote
#include "ude_poison_gen.h"
__global__ void __launch_bounds__(256) poison_regs_kernel(unsigned pat, unsigned* sink, unsigned mask, unsigned lo, unsigned hi) {
    if (hi > lo) { UDE_POISON_ASM_RANGE(0x9e3779b1u, lo, hi); } else if (mask) { UDE_POISON_ASM_MASK(0x9e3779b1u, mask); } else if (pat == 0x9e3779b1u) { UDE_POISON_ASM_LANES(pat); } else { UDE_POISON_ASM(pat); }
    if (pat == 0x12345u && sink) sink[threadIdx.x] = pat;
}
__global__ void __launch_bounds__(256) poison_lds_kernel(unsigned pat, int words, unsigned* sink) {
    extern __shared__ unsigned pl[];
    for (int i = threadIdx.x; i < words; i += 256) pl[i] = pat;
    __syncthreads();
    if (pat == 0x12345u && sink) sink[threadIdx.x] = pl[threadIdx.x];
}
void ude_poison_chip_dbg(hipStream_t st, bool before_forward) {  // (what & 4: also in front of the forward kernels)
    static const char* e = getenv("UDE_EXP_POISON");  // read once per process
    if (!e) return;
    int kind = 0, what = 3;
    unsigned mask = 0;  // kind 4: bit k < 8: v[32k, 32k+32) get lane-varying garbage, bit 8+k: a[32k, 32k+32); every other register zero
    unsigned lo = 0, hi = 0;  // kind 5: registers [lo, hi) of v0..v255 (0..255), a0..a255 (256..511) get the garbage, the others zero
    if (sscanf(e, "5,%d,%u,%u", &what, &lo, &hi) == 3) kind = 5; else { lo = hi = 0; sscanf(e, "%d,%d,%x", &kind, &what, &mask); }
    const unsigned pat = kind == 0 ? 0u : kind == 1 ? 0x7ff80000u : kind == 2 ? 0x41f00000u : 0x9e3779b1u;  // 3: different garbage in every lane and register
    if (before_forward && !(what & 4)) return;
    if (what & 1) hipLaunchKernelGGL(poison_regs_kernel, dim3(2048), dim3(256), 0, st, pat, (unsigned*)nullptr, kind == 4 ? mask : 0u, lo, hi);
    if (what & 2) {
        (void)hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(256), 160 * 1024, st, pat, 160 * 256, (unsigned*)nullptr);
    }
}
// Positive control of the hook (tests/test_gpu_poison.py): fill the chip with one pattern, then let a single wavefront report
// what two registers it never wrote hold.  A debug library built without the poison kernels -- or a poison launch that does
// not reach the register file -- fails this instead of silently turning every poison test into a plain re-run.
__global__ void __launch_bounds__(64) poison_probe_kernel(unsigned* out) {
    unsigned v, a;
    asm volatile("v_mov_b32 %0, v200\n\tv_accvgpr_read_b32 %1, a100" : "=v"(v), "=v"(a) : : "v200", "a100");
    out[threadIdx.x] = v;
    out[64 + threadIdx.x] = a;
}
extern "C" int ude_dbg_poison_selftest(ude_ctx* c, unsigned pat, unsigned* out_host /* 128 */) {
    if (!c || !out_host) return UDE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned* d = nullptr;
    HIPCHK(c, hipMalloc(&d, 128 * sizeof(unsigned)));
    hipLaunchKernelGGL(poison_regs_kernel, dim3(2048), dim3(256), 0, c->stream, pat, (unsigned*)nullptr, 0u, 0u, 0u);
    hipLaunchKernelGGL(poison_probe_kernel, dim3(1), dim3(64), 0, c->stream, d);
    HIPCHK(c, hipMemcpyAsync(out_host, d, 128 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(d));
    return UDE_OK;
}
#endif

// ---------------------------------------------------------------------------------------------
// compiled model table (instances live in their own translation units, see build.py)
// ---------------------------------------------------------------------------------------------
#include "ude_instances_gen.h"
// lock-step matrix-core adjoint of the SEIR exposure UDE (csrc/ude_seir_ls.hip)
extern "C" void ude_seir_ls_get(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, size_t* fac_doubles_per_block);
extern "C" void ude_seir_ls_get_fwd(int alg, void (**kern)(const KParams, int*), size_t* lds_bytes, int* blocks_per_cu);
extern "C" void ude_seir_ls_get_fwd_gen(int alg, void (**kern)(const KParams, int*), size_t* lds_bytes, int* blocks_per_cu);
extern "C" void ude_seir_ls_get_gen(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, size_t* fac_doubles_per_block);
// the `fast` mode on the same architecture: parameter cotangent as a block-level matrix-core accumulation (csrc/ude_seir_lsf.h)
extern "C" void ude_seir_lsf_get(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, int* blocks_per_cu);
extern "C" void ude_seir_lsf_get_gen(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, int* blocks_per_cu);
extern "C" void ude_node_lsf_get(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, int* blocks_per_cu);
extern "C" void ude_node_ls_get_fwd(int alg, void (**kern)(const KParams, int*), size_t* lds_bytes, int* blocks_per_cu);
extern "C" void ude_node_ls_get(int alg, void (**kern)(const KParams, double*, int*), size_t* lds_bytes, size_t* fac_doubles_per_block);

struct InstanceRow {
    int mid, alg, G, W;
    void (*get)(Launch*);
};
static const InstanceRow kInstances[] = {UDE_INSTANCE_TABLE};

static bool dims_are(const ude_model_desc* m, std::initializer_list<int> d, std::initializer_list<int> a) {
    if ((int)d.size() != m->n_layers + 1) return false;
    int i = 0;
    for (int v : d)
        if (m->dims[i++] != v) return false;
    i = 0;
    for (int v : a)
        if (m->act[i++] != v) return false;
    return true;
}

static int model_id(const ude_model_desc* m) {
    if (m->dtype == 1) {  // Float32 problems: hudson_bay.jl:77-104, scenario_3.jl:26-57 and :83-126
        if (m->kind == UDE_KIND_LV_UDE && m->n_state == 2 &&
            dims_are(m, {2, 5, 5, 5, 2}, {ACT_RBF, ACT_RBF, ACT_TANH, ACT_IDENTITY}))
            return MID_LV_HUDSON_F32;
        if (m->kind == UDE_KIND_KPP_TRUE && m->n_param == 0 && m->n_state >= 3 && m->n_state <= 32) return MID_KPP_TRUE_32_F32;
        if (m->kind == UDE_KIND_KPP_UDE && m->nn_offset == 0 && m->n_state >= 3 && m->n_state <= 32 &&
            dims_are(m, {1, 5, 5, 5, 1}, {ACT_RBF, ACT_RBF, ACT_RBF, ACT_IDENTITY}) && m->n_param == 81 && m->stencil_offset == 76 &&
            m->d0_offset == 80)
            return MID_KPP_S3_32_F32;
        return MID_NONE;
    }
    if (m->dtype != 0) return MID_NONE;
    if (m->kind == UDE_KIND_LV_TRUE && m->n_state == 2 && m->n_param == 4) return MID_LV_TRUE;
    if (m->kind == UDE_KIND_LV_UDE && m->n_state == 2) {
        if (dims_are(m, {2, 5, 5, 5, 2}, {ACT_RBF, ACT_RBF, ACT_RBF, ACT_IDENTITY})) return MID_LV_S1;
        if (dims_are(m, {2, 5, 5, 5, 2}, {ACT_RBF, ACT_RBF, ACT_TANH, ACT_IDENTITY})) return MID_LV_HUDSON;
        if (dims_are(m, {2, 32, 2}, {ACT_TANH, ACT_IDENTITY})) return MID_LV_TANH32;
    }
    if (m->kind == UDE_KIND_KPP_TRUE && m->n_param == 0 && m->n_state >= 3)
        return m->n_state <= 32 ? MID_KPP_TRUE_32 : m->n_state <= 1024 ? MID_KPP_TRUE_1024 : MID_NONE;
    if (m->kind == UDE_KIND_KPP_UDE && m->nn_offset == 0 && m->n_state >= 3) {
        if (dims_are(m, {1, 10, 20, 10, 1}, {ACT_TANH, ACT_TANH, ACT_TANH, ACT_IDENTITY}) && m->n_param == 466 &&
            m->stencil_offset == 461 && m->d0_offset == 465)
            return m->n_state <= 32 ? MID_KPP_UDE_32 : m->n_state <= 1024 ? MID_KPP_UDE_1024 : MID_NONE;
        if (dims_are(m, {1, 3, 1}, {ACT_TANH, ACT_IDENTITY}) && m->n_param == 15 && m->stencil_offset == 10 && m->d0_offset == 14 &&
            m->n_state <= 32)
            return MID_KPP_SMALL_32;  // the only variant the reference publishes timings for (Fisher-KPP-CNN-Small.jl:319-341)
        // n_weights = 1, 2 of Fisher-KPP-CNN-Small.jl:88 (timing log :343-391): theta = [W1 (n); b1 (n); W2 (n); b2; w1 w2 w3 unused; D0]
        for (int nw = 1; nw <= 2; ++nw)
            if (dims_are(m, {1, nw, 1}, {ACT_TANH, ACT_IDENTITY}) && m->n_param == 3 * nw + 6 && m->stencil_offset == 3 * nw + 1 &&
                m->d0_offset == 3 * nw + 5 && m->n_state <= 32)
                return nw == 1 ? MID_KPP_SMALL1_32 : MID_KPP_SMALL2_32;
        if (dims_are(m, {1, 5, 5, 5, 1}, {ACT_RBF, ACT_RBF, ACT_RBF, ACT_IDENTITY}) && m->n_param == 81 &&
            m->stencil_offset == 76 && m->d0_offset == 80 && m->n_state <= 32)
            return MID_KPP_S3_32;
    }
    if (m->kind == UDE_KIND_SEIR_TRUE && m->n_state == 7 && m->n_param == 0) return MID_SEIR_TRUE;
    if (m->kind == UDE_KIND_SEIR_UDE && m->n_state == 7 && m->nn_offset == 0 && m->n_param == 4481 &&
        dims_are(m, {3, 64, 64, 1}, {ACT_TANH, ACT_TANH, ACT_IDENTITY}))
        return MID_SEIR_UDE;
    if (m->kind == UDE_KIND_SEIR_NODE && m->n_state == 7 && m->nn_offset == 0 && m->n_param == 9287 &&
        dims_are(m, {7, 64, 64, 64, 7}, {ACT_TANH, ACT_TANH, ACT_TANH, ACT_IDENTITY}))
        return MID_SEIR_NODE;
    return MID_NONE;
}

// the runtime-shape fallback (csrc/ude_model_generic.h) takes any chain of 1..8 Dense layers of width 1..64 with activations
// identity / tanh / rbf / relu whose ends fit the kind's wiring: LV 2 -> 2, SEIR exposure 3 -> 1, neural ODE 7 -> 7
static int generic_id(const ude_model_desc* m, bool narrow_ok = true) {
    if ((m->dtype != 0 && m->dtype != 1) || m->n_layers < 1 || m->n_layers > UDE_MAX_LAYERS || m->nn_offset < 0) return MID_NONE;
    if (m->dtype == 1 && m->kind != UDE_KIND_LV_UDE) return MID_NONE;   // Float32: the LV kind (the reference's Float32 ODE problems)
    int np = 0;
    for (int l = 0; l <= m->n_layers; ++l)
        if (m->dims[l] < 1 || m->dims[l] > 64) return MID_NONE;
    for (int l = 0; l < m->n_layers; ++l) {
        if (m->act[l] < UDE_ACT_IDENTITY || m->act[l] > UDE_ACT_RELU) return MID_NONE;
        np += m->dims[l] * m->dims[l + 1] + m->dims[l + 1];
    }
    if (m->nn_offset + np > m->n_param) return MID_NONE;
    const int in = m->dims[0], out = m->dims[m->n_layers];
    if (m->kind == UDE_KIND_LV_UDE && m->n_state == 2 && in == 2 && out == 2) {
        for (int i = 0; i < 2; ++i)
            if (m->lin_idx[i] >= m->n_param || (m->lin_idx[i] >= m->nn_offset && m->lin_idx[i] < m->nn_offset + np)) return MID_NONE;
        // round 5: the scripts' own depth with EDITED widths / activations -- two or three hidden layers of width <= 8, linear output
        // layer -- on the lane-group kernels of the compiled instances (NetCfgRt: padded register copy of the weights, eight lanes per
        // trajectory) instead of one wavefront per trajectory
        if (narrow_ok && m->n_layers == 2 && m->act[1] == UDE_ACT_IDENTITY && m->dtype == 0 && m->dims[1] <= 16)   // one hidden layer
            return m->dims[1] <= 8 ? MID_LV_RT2 : MID_LV_RT2_W16;
        if (narrow_ok && (m->n_layers == 3 || m->n_layers == 4) && m->act[m->n_layers - 1] == UDE_ACT_IDENTITY) {
            int wmax = 0;
            for (int l = 1; l < m->n_layers; ++l) wmax = m->dims[l] > wmax ? m->dims[l] : wmax;
            const bool f32 = m->dtype == 1;
            if (wmax <= 5) return m->n_layers == 3 ? (f32 ? MID_LV_RT3_W5_F32 : MID_LV_RT3_W5) : (f32 ? MID_LV_RT4_W5_F32 : MID_LV_RT4_W5);   // five lanes per trajectory: the headline instance's layout
            if (wmax <= 8) return m->n_layers == 3 ? (f32 ? MID_LV_RT3_F32 : MID_LV_RT3) : (f32 ? MID_LV_RT4_F32 : MID_LV_RT4);
            if (wmax <= 16 && !f32) return m->n_layers == 3 ? MID_LV_RT3_W16 : MID_LV_RT4_W16;   // sixteen lanes, weights from the LDS copy of theta
        }
        if (m->dtype == 1) return m->n_layers <= 4 ? MID_GENERIC_2_L4_F32 : MID_GENERIC_2_F32;
        return m->n_layers <= 4 ? MID_GENERIC_2_L4 : MID_GENERIC_2;   // (<= 4 layers: the instance with half the stage storage)
    }
    // nn_ode with a pointwise reaction network that has no compiled instance: <= 4 layers of width <= 32, <= 768 parameters in all,
    // theta = [NN; w1 w2 w3 unused; D0] (Fisher-KPP-CNN.jl:100-109), grids of 3 .. 32 points, Float64
    // ... and on the LARGE grids (33 .. 1024 points) a chain 1 -> a -> b -> c -> 1 with tanh hidden layers of width <= 16: the run-time-shape
    // instance of the matrix-core kernel of the 1024-point instance (KppUdeW over NetCfgRt; Tsit5: the Vern7 stage storage does not fit)
    if (m->kind == UDE_KIND_KPP_UDE && m->dtype == 0 && in == 1 && out == 1 && m->n_layers == 4 && m->nn_offset == 0 && m->n_state > 32 &&
        m->n_state <= 1024 && m->n_param == np + 5 && m->stencil_offset == np && m->d0_offset == np + 4 && m->dims[1] <= 16 && m->dims[2] <= 16 &&
        m->dims[3] <= 16 && m->act[0] == UDE_ACT_TANH && m->act[1] == UDE_ACT_TANH && m->act[2] == UDE_ACT_TANH && m->act[3] == UDE_ACT_IDENTITY)
        return MID_KPP_RT_1024;
    if (m->kind == UDE_KIND_KPP_UDE && m->dtype == 0 && in == 1 && out == 1 && m->n_layers <= 4 && m->nn_offset == 0 && m->n_state >= 3 &&
        m->n_state <= 32 && m->n_param == np + 5 && m->n_param <= 768 && m->stencil_offset == np && m->d0_offset == np + 4) {
        for (int l = 0; l <= m->n_layers; ++l)
            if (m->dims[l] > 32) return MID_NONE;
        return MID_KPP_GENERIC_32;
    }
    if (m->kind == UDE_KIND_SEIR_UDE && m->n_state == 7 && in == 3 && out == 1) return m->n_layers <= 4 ? MID_GENERIC_7_L4 : MID_GENERIC_7;
    if (m->kind == UDE_KIND_SEIR_NODE && m->n_state == 7 && in == 7 && out == 7) return m->n_layers <= 4 ? MID_GENERIC_7_L4 : MID_GENERIC_7;
    return MID_NONE;
}

// exposure-UDE chains WITHOUT a compiled instance that the lock-step matrix-core backward kernel serves all the same (csrc/ude_seir_ls2.h, GEN):
// 3 -> H1 -> H2 -> 1 with tanh, tanh, identity, H1, H2 <= 64, Float64 -- except the shapes with a product the kernels do not have in the
// oracle's association (wide_dot: a product of 32 or 64 terms with fewer than 16 results is the adjacent-pair TREE): H1 = 32 (the
// 32-term input cotangent), H1 = 64 with H2 < 16 (forward hidden layer), H2 = 32 or 64 with H1 < 16 (transposed hidden layer)
static bool seir_gen_ls_shape(const ude_model_desc* m) {
    if (m->kind != UDE_KIND_SEIR_UDE || m->dtype != 0 || m->n_state != 7 || m->n_layers != 3 || m->nn_offset != 0) return false;
    const int h1 = m->dims[1], h2 = m->dims[2];
    if (m->dims[0] != 3 || m->dims[3] != 1 || h1 < 1 || h1 > 64 || h1 == 32 || h2 < 1 || h2 > 64) return false;
    if ((h1 == 64 && h2 < 16) || ((h2 == 32 || h2 == 64) && h1 < 16)) return false;
    if (m->act[0] != UDE_ACT_TANH || m->act[1] != UDE_ACT_TANH || m->act[2] != UDE_ACT_IDENTITY) return false;
    return m->n_param == 3 * h1 + h1 + h1 * h2 + h2 + h2 + 1;
}
// ... and the ones the lock-step FORWARD kernel serves (ude_seir_ls_fwd.h, GEN): additionally H2 != 32 (a 32-term output layer is a tree case)
static bool seir_gen_ls_fwd_shape(const ude_model_desc* m) { return seir_gen_ls_shape(m) && m->dims[2] != 32; }

// blocks of the lock-step SEIR backward kernel: 16 trajectory slots each, at most one block per compute unit (the slots refill
// from a queue)
static int64_t ls_blocks(ude_ctx* c, int64_t N, int per_cu = 1) {
    if (c->ncu <= 0) {
        hipDeviceProp_t prop;
        c->ncu = (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int64_t nb = (N + 15) / 16;
    return nb < (int64_t)per_cu * c->ncu ? nb : (int64_t)per_cu * c->ncu;
}
#ifndef UDE_SEIR_LS_DEFAULT
#define UDE_SEIR_LS_DEFAULT 1   // 1: the lock-step matrix-core backward kernel is the default for the SEIR exposure UDE
#endif
#ifndef UDE_NODE_LS_DEFAULT
#define UDE_NODE_LS_DEFAULT 1   // ... and so does the SEIR neural ODE (csrc/ude_node_ls.h); lanes_per_traj = 64 selects the wavefront-per-trajectory kernel
#endif
#ifndef UDE_NODE_LS_FWD
#define UDE_NODE_LS_FWD 1       // the neural ODE's forward solve on the lock-step architecture (ude_node_ls_fwd.h)
#endif
#ifndef UDE_SEIR_LS_FWD
#define UDE_SEIR_LS_FWD 1       // 1: ... and the forward pass of a gradient call runs on the same architecture (ude_seir_ls_fwd.h)
#endif
static int default_lanes(int mid, bool discrete) {
    switch (mid) {
        case MID_LV_TRUE: return 1;
        case MID_LV_RT3_W5:
        case MID_LV_RT4_W5:
        case MID_LV_RT3_W5_F32:
        case MID_LV_RT4_W5_F32:
        case MID_LV_S1: return 5;  // 12 trajectories per wavefront: every lane of the 5-wide layers busy, C2 fits in one round
        case MID_LV_HUDSON:
        case MID_LV_RT2:
        case MID_LV_RT3:
        case MID_LV_RT4:
        case MID_LV_RT3_F32:
        case MID_LV_RT4_F32:
        case MID_LV_HUDSON_F32: return 8;
        case MID_LV_RT2_W16:
        case MID_LV_RT3_W16:
        case MID_LV_RT4_W16:
        case MID_LV_TANH32: return 16;  // two hidden neurons per lane, four trajectories per wavefront, 253 registers = two wavefronts per SIMD.
                                        // Round 4 (32-term tree sums by the group's butterfly, parameter slots by input): 10k-trajectory gradient
                                        // 1.98 ms against 2.27 ms with 8 lanes and 2.28 ms with 32; 16 lanes stay ahead from 5k to 40k trajectories
        case MID_SEIR_TRUE: return 1;
        case MID_SEIR_UDE: return 64;  // wavefront per trajectory, 4 per block
        case MID_SEIR_NODE: return 64;  // wavefront per trajectory, 3 per block (two 64x64 layers + the compacted stage factors fill the LDS)
        case MID_GENERIC_2:
        case MID_GENERIC_7:
        case MID_GENERIC_2_L4:
        case MID_GENERIC_7_L4:
        case MID_GENERIC_2_F32:
        case MID_GENERIC_2_L4_F32: return 64;  // wavefront per trajectory, one per block
        case MID_KPP_TRUE_32:
        case MID_KPP_UDE_32:
        case MID_KPP_S3_32:
        case MID_KPP_SMALL_32:
        case MID_KPP_SMALL1_32:
        case MID_KPP_SMALL2_32:
        case MID_KPP_TRUE_32_F32:
        case MID_KPP_S3_32_F32:
        case MID_KPP_GENERIC_32: return 32;
        case MID_KPP_TRUE_1024: return 64;
        case MID_KPP_RT_1024:
        case MID_KPP_UDE_1024: return 256;  // 4 wavefronts per PDE
    }
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Cost-ordered adjoint launch (round 6; SURVEY.md 7, "hard parts": sort / bucket trajectories by expected cost).  The trajectories of a
// wavefront make their backward step attempts together: a wavefront runs as long as its slowest trajectory (lane_step_util 0.82 on
// the LV ensembles), and a launch of several rounds of wavefronts additionally ends with whatever round happens to hold the slowest
// ones.  The cost of a backward solve is not known in advance -- but a gradient is asked for again and again with slowly moving parameters, and what
// predicts a member's cost best is what it cost LAST TIME (lock-step utilisation 0.82 -> 0.999 when the wavefronts are filled in that order; the
// member's loss: 0.93, its forward step count -- 26 or 27 for every member -- 0.84).  Counting sort by the previous call's backward attempt counts,
// most expensive first, BEFORE the forward kernel: forward and backward kernel both run in that order, so the internal workspaces are indexed by lane-
// group position and stay coalesced.  The order INSIDE a bucket is whatever the atomics produce -- harmless, because in this mode every member
// writes its own gradient row and the rows are added in member order (KParams::perm).  The first call on an ensemble runs the identity order.
// the N gradient rows of a cost-ordered launch, first level of their sum: block (x, y) adds rows [y * chunk, (y + 1) * chunk) of 32 adjacent
// columns (coalesced 256-byte reads; fixed order: 32 row-lanes striding the chunk, then the 32 partials left to right) into out[y][col];
// the finish kernel adds the chunk rows.  Every association is fixed by (N, chunk) alone -- not by the permutation.
constexpr int ROWSUM_CHUNK = 1024;
__global__ void __launch_bounds__(1024) rows_chunk_sum_kernel(const double* part, int64_t nrows, int32_t ncols, double* out) {
    __shared__ double sh[32][33];
    const int tid = threadIdx.x, cx = tid & 31, ry = tid >> 5;
    const int col = blockIdx.x * 32 + cx;
    const int cc = col < ncols ? col : ncols - 1;
    const int64_t r0 = (int64_t)blockIdx.y * ROWSUM_CHUNK, r1 = r0 + ROWSUM_CHUNK < nrows ? r0 + ROWSUM_CHUNK : nrows;
    double s = 0.0;
    for (int64_t w0 = r0 + ry; w0 < r1; w0 += 32 * 8) {
        double pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t w = w0 + 32 * u;
            pv[u] = part[(size_t)(w < r1 ? w : r0) * ncols + cc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (w0 + 32 * u < r1) s += pv[u];
    }
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < ncols) {
        double t = sh[0][cx];
        for (int q = 1; q < 32; ++q) t += sh[q][cx];
        out[(size_t)blockIdx.y * ncols + col] = t;
    }
}

constexpr int SORT_BUCKETS = 4096;
__device__ __forceinline__ int cost_bucket(double loss) {
    // positive doubles order like their bit patterns: exponent and the top four mantissa bits, 2^-64 .. 2^192 mapped onto 0 .. 4095;
    // NaN / Inf (a failed forward solve: the adjoint kernel skips it) land in the last bucket
    const long long b = (long long)(__double_as_longlong(loss > 0.0 ? loss : 0.0) >> 48) - ((1023LL - 64) << 4);
    return b < 0 ? 0 : b >= SORT_BUCKETS ? SORT_BUCKETS - 1 : (int)b;
}
// key of member i: its backward step attempts in the previous call on the same ensemble where the context has them (a training loop calls
// again and again with slowly moving parameters: lock-step utilisation 0.999 when sorted by them), else the bucket of its loss
__device__ __forceinline__ int cost_key(const double* loss_traj, const int32_t* prev, int64_t i) {
    if (prev) { const int a = prev[i]; return a < 0 ? 0 : a >= SORT_BUCKETS ? SORT_BUCKETS - 1 : a; }
    return cost_bucket(loss_traj[i]);
}
// (histogram and scatter go through a block-private LDS histogram first: the keys of an ensemble fall into a few dozen buckets, and 160 000
//  global atomics on forty addresses serialise -- the first version spent 1.3 ms of a 15 ms step in these two kernels)
__global__ void __launch_bounds__(1024) sort_hist_kernel(const double* loss_traj, const int32_t* prev, int64_t N, int32_t* hist) {
    __shared__ int lh[SORT_BUCKETS];
    for (int b = threadIdx.x; b < SORT_BUCKETS; b += 1024) lh[b] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i < N) atomicAdd(&lh[cost_key(loss_traj, prev, i)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < SORT_BUCKETS; b += 1024)
        if (lh[b]) atomicAdd(&hist[b], lh[b]);
}
// exclusive prefix sums over the buckets in DESCENDING key order (one block of 1024 threads, four buckets each).  With keys that ARE costs
// (the previous call's attempt counts) the block also decides whether sorting can pay at all: an ensemble whose members all cost the same
// (spread of the keys within 1/16 of their mean: the 2-8-8-8-2 tanh ensemble of `lv_shape8` has lane_step_util 0.9995) keeps the identity
// order -- a permuted launch reads its members' workspace columns scattered, which costs the latency-bound LV kernels 7 % (measured).
__global__ void __launch_bounds__(1024) sort_scan_kernel(const int32_t* hist, int32_t* offset, int keys_are_costs) {
    __shared__ int sh[1024];
    __shared__ long long shw[1024];
    __shared__ int shlo[1024], shhi[1024];
    const int t = threadIdx.x;
    int v[4], s = 0, lo = SORT_BUCKETS, hi = -1;
    long long wsum = 0;
    for (int q = 0; q < 4; ++q) {
        const int b = SORT_BUCKETS - 1 - (4 * t + q);
        v[q] = hist[b];
        s += v[q];
        wsum += (long long)v[q] * b;
        if (v[q] > 0) { lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
    }
    sh[t] = s; shw[t] = wsum; shlo[t] = lo; shhi[t] = hi;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int x = t >= d ? sh[t - d] : 0;
        const long long xw = t >= d ? shw[t - d] : 0;
        const int xl = t >= d ? shlo[t - d] : SORT_BUCKETS, xh = t >= d ? shhi[t - d] : -1;
        __syncthreads();
        sh[t] += x; shw[t] += xw; shlo[t] = xl < shlo[t] ? xl : shlo[t]; shhi[t] = xh > shhi[t] ? xh : shhi[t];
        __syncthreads();
    }
    int base = sh[t] - s;
    for (int q = 0; q < 4; ++q) { offset[SORT_BUCKETS - 1 - (4 * t + q)] = base; base += v[q]; }
    if (t == 1023) {   // (inclusive totals)
        const long long n = sh[t], mean16 = n > 0 ? shw[t] / n / 16 : 0;
        offset[SORT_BUCKETS] = (keys_are_costs && n > 0 && (long long)(shhi[t] - shlo[t]) <= mean16) ? 1 : 0;   // 1: keep the identity order
    }
}
__global__ void __launch_bounds__(1024) sort_scatter_kernel(const double* loss_traj, const int32_t* prev, int64_t N, int32_t* offset, int32_t* perm, int identity) {
    // block-private ranks: a member's position inside its bucket = (the block's reservation in the bucket) + (its rank among the block's
    // members of that bucket); one global atomic per (block, non-empty bucket)
    __shared__ int lh[SORT_BUCKETS];
    __shared__ int lbase[SORT_BUCKETS];
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    // (identity: the scan's verdict, or UDE_COST_SORT=2 -- the mode's machinery with the identity order: what the indirection alone costs)
    if (identity || offset[SORT_BUCKETS]) {
        if (i < N) perm[i] = (int32_t)i;
        return;
    }
    for (int b = threadIdx.x; b < SORT_BUCKETS; b += 1024) lh[b] = 0;
    __syncthreads();
    int key = 0, rank = 0;
    if (i < N) { key = cost_key(loss_traj, prev, i); rank = atomicAdd(&lh[key], 1); }
    __syncthreads();
    for (int b = threadIdx.x; b < SORT_BUCKETS; b += 1024)
        if (lh[b]) lbase[b] = atomicAdd(&offset[b], lh[b]);
    __syncthreads();
    if (i < N) perm[lbase[key] + rank] = (int32_t)i;
}
// the three kernels above as ONE block for ensembles of up to SORT_SMALL_MAX members (histogram and offsets in LDS): a launch of a
// millisecond or two (10 000 members on 8 or 16 lanes are 1250 / 2500 wavefronts: more than one round) cannot afford four extra launches
constexpr int SORT_SMALL_MAX = 32768;
__global__ void __launch_bounds__(1024) sort_small_kernel(const double* loss_traj, const int32_t* prev, int64_t N, int32_t* perm, int identity) {
    __shared__ int hist[SORT_BUCKETS];
    __shared__ int off[SORT_BUCKETS];
    __shared__ int sh[1024];
    __shared__ long long shw[1024];
    __shared__ int shlo[1024], shhi[1024];
    __shared__ int keep;
    const int t = threadIdx.x;
    if (identity) {   // (no costs of a previous call yet, or UDE_COST_SORT=2: the mode's machinery with the identity order)
        for (int64_t i = t; i < N; i += 1024) perm[i] = (int32_t)i;
        return;
    }
    for (int b = t; b < SORT_BUCKETS; b += 1024) hist[b] = 0;
    __syncthreads();
    for (int64_t i = t; i < N; i += 1024) atomicAdd(&hist[cost_key(loss_traj, prev, i)], 1);
    __syncthreads();
    int v[4], s = 0, lo = SORT_BUCKETS, hi = -1;
    long long wsum = 0;
    for (int q = 0; q < 4; ++q) {
        const int b = SORT_BUCKETS - 1 - (4 * t + q);
        v[q] = hist[b];
        s += v[q];
        wsum += (long long)v[q] * b;
        if (v[q] > 0) { lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
    }
    sh[t] = s; shw[t] = wsum; shlo[t] = lo; shhi[t] = hi;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int x = t >= d ? sh[t - d] : 0;
        const long long xw = t >= d ? shw[t - d] : 0;
        const int xl = t >= d ? shlo[t - d] : SORT_BUCKETS, xh = t >= d ? shhi[t - d] : -1;
        __syncthreads();
        sh[t] += x; shw[t] += xw; shlo[t] = xl < shlo[t] ? xl : shlo[t]; shhi[t] = xh > shhi[t] ? xh : shhi[t];
        __syncthreads();
    }
    int base = sh[t] - s;
    for (int q = 0; q < 4; ++q) { off[SORT_BUCKETS - 1 - (4 * t + q)] = base; base += v[q]; }
    if (t == 1023) {
        const long long n = sh[t], mean16 = n > 0 ? shw[t] / n / 16 : 0;
        keep = (prev && n > 0 && (long long)(shhi[t] - shlo[t]) <= mean16) ? 1 : 0;   // (same verdict as sort_scan_kernel)
    }
    __syncthreads();
    for (int64_t i = t; i < N; i += 1024) {
        if (keep) perm[i] = (int32_t)i;
        else perm[atomicAdd(&off[cost_key(loss_traj, prev, i)], 1)] = (int32_t)i;
    }
}

// after the backward kernel: what every member's backward solve cost (accepted + rejected attempts), for the next call's sort
__global__ void cost_save_kernel(const int64_t* stats, int64_t N, int32_t* prev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) prev[i] = (int32_t)(stats[i * 8 + 5] + stats[i * 8 + 6]);
}

static int resolve(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, Launch& l, int& G, bool* generic = nullptr) {
    int mid = model_id(m);
    if (generic) *generic = false;
    if (mid == MID_NONE) {  // no compiled instance for this exact shape: the runtime-shape kernel, if the descriptor fits it
        mid = generic_id(m);
        if (generic) *generic = mid != MID_NONE;
    }
    if (mid == MID_NONE)
        return fail(c, UDE_ERR_UNSUPPORTED, "no kernel for model kind=%d dtype=%d n_layers=%d: not a compiled instance (udecore.hip model table) and "
                                            "outside the runtime-shape fallbacks (LV / SEIR kinds: <= 8 layers of width <= 64; Fisher-KPP on <= 32 points: <= 4 layers of width <= 32, on 33 .. 1024 points: three tanh layers of width <= 16; Float64)",
                    m->kind, m->dtype, m->n_layers);
    G = c->lo.lanes_per_traj > 0 ? c->lo.lanes_per_traj : default_lanes(mid, o->sensealg == UDE_SENSE_DISCRETE);
    if (G == 8) {   // (an explicit lanes_per_traj = 8: the width-8 instance takes narrower chains too)
        mid = mid == MID_LV_RT3_W5 ? MID_LV_RT3 : mid == MID_LV_RT4_W5 ? MID_LV_RT4 : mid == MID_LV_RT3_W5_F32 ? MID_LV_RT3_F32 : mid == MID_LV_RT4_W5_F32 ? MID_LV_RT4_F32 : mid;
    }
    // (16 = the lock-step kernels of grad_dev_impl; every other kernel of these models is one wavefront per trajectory.  Round 6, advisor: the
    //  runtime-shape exposure chains -- MID_GENERIC_7[_L4] -- reach their lock-step GEN instances through the same request: an explicit
    //  lanes_per_traj = 16 on 3-64-63-1 used to end in "no kernel instance ... lanes_per_traj 16")
    if ((mid == MID_SEIR_UDE || mid == MID_SEIR_NODE || mid == MID_GENERIC_7 || mid == MID_GENERIC_7_L4) && G == 16) G = 64;
    const int W = c->lo.waves_per_simd > 0 ? c->lo.waves_per_simd : 1;
    bool ok = false;
    // scenario_1's chain with both diagonal coefficients constant has a leaner instance (no slots for them) where compiled
    const int mid_first = (mid == MID_LV_S1 && m->lin_idx[0] < 0 && m->lin_idx[1] < 0) ? MID_LV_S1N : mid;
    for (int pass = 0; pass < 2 && !ok; ++pass) {
        const int want = pass == 0 ? mid_first : mid;
        for (const InstanceRow& row : kInstances)
            if (row.mid == want && row.alg == o->alg && row.G == G && row.W == W) {
                row.get(&l);
                ok = true;
                break;
            }
    }
    // lanes_per_traj = 64 on an LV-kind model that has no 64-lane instance of its own: "one wavefront per trajectory" is the
    // runtime-shape kernel (lane j = neuron j of every layer) -- the layout north_star names; same bits as every other instance
    if (!ok && G == 64 && W == 1 && m->kind == UDE_KIND_LV_UDE) {
        const int gid = generic_id(m, false);
        for (const InstanceRow& row : kInstances)
            if (gid != MID_NONE && row.mid == gid && row.alg == o->alg && row.G == 64 && row.W == 1) {
                row.get(&l);
                ok = true;
                if (generic) *generic = true;
                break;
            }
    }
    if (!ok) return fail(c, UDE_ERR_UNSUPPORTED, "no kernel instance for model %d alg %d lanes_per_traj %d waves_per_simd %d", mid, o->alg, G, W);
    return UDE_OK;
}

// the algorithm's tableau in the problem's scalar type (the kernels of a Float32 instance read it as TabDevT<float>)
static const TabDev* tab_for(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o) {
    const int a = o->alg == UDE_ALG_VERN7 ? 1 : 0;
    const char* base = (const char*)c->tabs.p;
    if (m->dtype == 1) return (const TabDev*)(base + 2 * sizeof(TabDevT<double>) + a * sizeof(TabDevT<float>));
    return (const TabDev*)(base + a * sizeof(TabDevT<double>));
}

static void fill_params(KParams& p, const ude_model_desc* m, const ude_solve_opts* o, double t0, double tf) {
    memset(&p, 0, sizeof p);
    const int order = o->alg == UDE_ALG_VERN7 ? 7 : 5;
    p.o.abstol = o->abstol > 0 ? o->abstol : 1e-6;
    p.o.reltol = o->reltol > 0 ? o->reltol : 1e-3;
    p.o.dtmax = o->dtmax > 0 ? o->dtmax : fabs(tf - t0);
    p.o.dt0 = o->dt0;
    p.o.qmin = o->qmin > 0 ? o->qmin : 0.2;
    p.o.qmax = o->qmax > 0 ? o->qmax : 10.0;
    p.o.gamma = o->gamma > 0 ? o->gamma : 0.9;
    p.o.qoldinit = o->qoldinit > 0 ? o->qoldinit : 1e-4;
    p.o.beta2 = o->beta2 > 0 ? o->beta2 : 2.0 / (5.0 * order);
    p.o.beta1 = o->beta1 > 0 ? o->beta1 : 7.0 / (10.0 * order);
    p.o.maxiters = o->maxiters > 0 ? o->maxiters : 100000;
    p.t0 = t0;
    p.tf = tf;
    p.n_state = m->n_state;
    p.n_param = m->n_param;
    p.mc.n_state = m->n_state;
    p.mc.n_param = m->n_param;
    p.mc.nn_offset = m->nn_offset;
    p.mc.stencil_offset = m->stencil_offset;
    p.mc.d0_offset = m->d0_offset;
    p.mc.kind = m->kind;
    p.mc.n_layers = m->n_layers;
    for (int i = 0; i <= UDE_MAX_LAYERS; ++i) p.mc.dims[i] = (i <= m->n_layers && m->n_layers <= UDE_MAX_LAYERS) ? m->dims[i] : 0;
    for (int i = 0; i < UDE_MAX_LAYERS; ++i) p.mc.act[i] = i < m->n_layers ? m->act[i] : 0;
    for (int i = 0; i < 2; ++i) {
        p.mc.lin_idx[i] = m->lin_idx[i];
        p.mc.lin_sign[i] = m->lin_sign[i];
        p.mc.lin_const[i] = m->lin_const[i];
    }
    for (int i = 0; i < 16; ++i) p.mc.consts[i] = m->consts[i];
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int ude_version(void) { return UDE_VERSION; }
extern "C" void ude_destroy(ude_ctx* c);

extern "C" int ude_create(int32_t device_id, ude_ctx** out) {
    if (!out) return UDE_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return UDE_ERR_HIP;
    if (hipSetDevice(device_id) != hipSuccess) return UDE_ERR_HIP;
    ude_ctx* c = new ude_ctx();
    c->device = device_id;
    bool ok = hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming) == hipSuccess;
    for (auto& e : c->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (ok) {   // tableaux in device memory: [0] Tsit5, [1] Vern7
        struct { TabDevT<double> d[2]; TabDevT<float> f[2]; } host = {{make_tabdev<Tsit5Tab, double>(), make_tabdev<Vern7Tab, double>()},
                                                                      {make_tabdev<Tsit5Tab, float>(), make_tabdev<Vern7Tab, float>()}};
        ok = hipMalloc(&c->tabs.p, sizeof host) == hipSuccess &&
             hipMemcpy(c->tabs.p, &host, sizeof host, hipMemcpyHostToDevice) == hipSuccess;
        c->tabs.cap = sizeof host;
    }
    if (!ok) {  // release whatever was created
        ude_destroy(c);
        return UDE_ERR_HIP;
    }
    *out = c;
    return UDE_OK;
}

extern "C" void ude_destroy(ude_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipDeviceSynchronize();
    if (c->ev_sync) (void)hipEventDestroy(c->ev_sync);
    DevBuf* bufs[] = {&c->dense, &c->dense_n, &c->cot, &c->loss_traj, &c->grad_part, &c->retcode, &c->stats, &c->trace, &c->tabs, &c->slot_glob, &c->nfail, &c->tspan_pt, &c->ls_fac,
                      &c->s_u0, &c->s_theta, &c->s_saveat, &c->s_out, &c->s_data, &c->s_mask, &c->s_gtheta,
                      &c->s_gu0, &c->s_loss, &c->s_lpt, &c->s_stats, &c->s_ret};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (DevBuf& b : c->hj)
        if (b.p) (void)hipFree(b.p);
    for (auto& e : c->hj_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    delete c;
}

extern "C" const char* ude_last_error(ude_ctx* c) { return c ? c->err.c_str() : "null context"; }

// true while `st` is being captured into a hipGraph (torch.cuda.graph / hipStreamBeginCapture): no event records for the
// timing hooks, no cross-stream hops, no allocation may happen inside the capture
static bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st == nullptr) return false;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs == hipStreamCaptureStatusActive;
}

extern "C" int ude_set_stream(ude_ctx* c, void* s) {
    if (!c) return UDE_ERR_INVALID;
    hipStream_t ns = (hipStream_t)s;
    if (ns != c->stream) {
        // the workspaces are shared by every call on this context: work already queued on the old stream must
        // finish before work on the new stream may reuse them
        HIPCHK(c, hipSetDevice(c->device));
        if (!capturing(ns) && !capturing(c->stream)) {  // (a capture starts from an idle context: the caller synchronises first)
            // the old stream may already have been destroyed by the host: then (or on any other failure of the event hop) fall
            // back to a device-wide sync -- the context must never stay bound to a dead stream
            if (hipEventRecord(c->ev_sync, c->stream) != hipSuccess || hipStreamWaitEvent(ns, c->ev_sync, 0) != hipSuccess) {
                (void)hipGetLastError();
                c->stream = ns;
                HIPCHK(c, hipDeviceSynchronize());
            }
        }
        c->stream = ns;
    }
    return UDE_OK;
}

extern "C" int ude_set_launch_opts(ude_ctx* c, const ude_launch_opts* lo) {
    if (!c || !lo) return UDE_ERR_INVALID;
    c->lo = *lo;
    return UDE_OK;
}

extern "C" int ude_model_supported(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int32_t need_adjoint) {
    if (!c || !m || !o) return UDE_ERR_INVALID;
    Launch l;
    int G;
    bool generic = false;
    const int rc = resolve(c, m, o, l, G, &generic);
    if (rc) return rc;
    if (need_adjoint && o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT_CHECKPOINTED && (!l.adj_ckpt || o->per_trajectory))
        return UDE_ERR_UNSUPPORTED;
    if ((o->per_trajectory & UDE_PT_THETA) && !l.per_member)   // (the same answer the solve / gradient entry points give)
        return fail(c, UDE_ERR_UNSUPPORTED, "per-member parameters (UDE_PT_THETA): this model / lanes_per_traj has no per-member kernel (LV-kind compiled instances and the Fisher-KPP kinds have one)");
    if (need_adjoint && o->sensealg == UDE_SENSE_DISCRETE && !l.dadj)
        return fail(c, UDE_ERR_UNSUPPORTED, "the runtime-shape kernel has no discretise-then-optimise sweep: use the interpolating adjoint");
    return generic ? 1 : UDE_OK;   // 0: compiled fast instance, 1: runtime-shape fallback kernel
}

extern "C" int ude_set_trace(ude_ctx* c, int64_t traj, int32_t cap) {
    if (!c) return UDE_ERR_INVALID;
    c->trace_traj = traj;
    c->trace_cap = traj >= 0 ? cap : 0;
    if (c->trace_cap > 0) {
        int rc = ensure(c, c->trace, sizeof(double) * 10 * (size_t)cap);
        if (rc) return rc;
        HIPCHK(c, hipMemset(c->trace.p, 0, sizeof(double) * 10 * (size_t)cap));
    }
    return UDE_OK;
}

extern "C" int ude_get_trace(ude_ctx* c, double* out_host) {
    if (!c || !out_host || c->trace_cap <= 0) return UDE_ERR_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out_host, c->trace.p, sizeof(double) * 10 * (size_t)c->trace_cap, hipMemcpyDeviceToHost));
    return UDE_OK;
}

extern "C" int ude_last_failures(ude_ctx* c, const int32_t* retcode_dev, int64_t N, int32_t* nfail, int32_t* grown) {
    if (!c || !nfail) return UDE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *nfail = 0;
    if (grown) *grown = 0;
    if (!c->nfail.p) return UDE_OK;
    HIPCHK(c, hipMemcpy(nfail, c->nfail.p, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (*nfail > 0 && retcode_dev && N > 0 && c->lo.max_dense_steps <= 0 && c->auto_cap < (1 << 20)) {
        std::vector<int32_t> r(N);
        HIPCHK(c, hipMemcpy(r.data(), retcode_dev, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
        for (int64_t j = 0; j < N; ++j)
            if (r[j] == UDE_RET_DENSE_OVERFLOW) {
                c->auto_cap *= 4;
                if (grown) *grown = 1;
                break;
            }
    }
    return UDE_OK;
}

extern "C" int ude_last_kernel_ms(ude_ctx* c, float* fwd_ms, float* bwd_ms) {
    if (!c) return UDE_ERR_INVALID;
    if (fwd_ms) {
        *fwd_ms = 0.f;
        if (c->ev_fwd) HIPCHK(c, hipEventElapsedTime(fwd_ms, c->ev[0], c->ev[1]));
    }
    if (bwd_ms) {
        *bwd_ms = 0.f;
        if (c->ev_bwd) HIPCHK(c, hipEventElapsedTime(bwd_ms, c->ev[2], c->ev[3]));
    }
    return UDE_OK;
}

// ---------------------------------------------------------------------------------------------
// device-resident entry points
// ---------------------------------------------------------------------------------------------
static int common_checks(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N, const void* u0,
                         const double* tspan_host, const void* theta, const void* saveat, int32_t ns) {
    if (!c) return UDE_ERR_INVALID;
    if (!m || !o || !u0 || !tspan_host || !saveat || N <= 0 || ns <= 0) return fail(c, UDE_ERR_INVALID, "null argument or empty ensemble");
    if (m->n_param > 0 && !theta) return fail(c, UDE_ERR_INVALID, "theta is null");
    const int64_t npairs = (o->per_trajectory & UDE_PT_TSPAN) ? N : 1;
    for (int64_t j = 0; j < npairs; ++j)
        if (!(tspan_host[2 * j + 1] > tspan_host[2 * j])) return fail(c, UDE_ERR_INVALID, "tspan must be increasing");
    return UDE_OK;
}

// per-trajectory time grids (o->per_trajectory): tspan pairs to the device, kernel variants with their own grid
static int setup_time_grids(ude_ctx* c, const ude_solve_opts* o, int64_t N, const double* tspan_host, KParams& p) {
    p.tspan_pt = nullptr;
    p.saveat_pt = (o->per_trajectory & UDE_PT_SAVEAT) ? 1 : 0;
    p.dtmax_auto = o->dtmax > 0 ? 0 : 1;
    if (o->per_trajectory & UDE_PT_TSPAN) {
        int rc = ensure(c, c->tspan_pt, sizeof(double) * 2 * N);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->tspan_pt.p, tspan_host, sizeof(double) * 2 * N, hipMemcpyHostToDevice, c->stream));
        p.tspan_pt = (const double*)c->tspan_pt.p;
    }
    return UDE_OK;
}

// tspan is always a HOST pointer (two doubles), also for the _dev entry points
static int solve_dev_impl(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N, const double* u0,
                          const double* tspan, const double* theta, const double* saveat, int32_t ns,
                          double* u_out, int64_t* stats, int32_t* retcode) {
    int rc = common_checks(c, m, o, N, u0, tspan, theta, saveat, ns);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    Launch l;
    int G;
    if ((rc = resolve(c, m, o, l, G))) return rc;
    KParams p;
    fill_params(p, m, o, tspan[0], tspan[1]);
    p.tab = tab_for(c, m, o);
    if (c->trace_cap > 0) { p.trace = (double*)c->trace.p; p.trace_traj = c->trace_traj; p.trace_cap = c->trace_cap; }
    p.N = N;
    p.Npad = N;
    p.ns = ns;
    p.u0 = u0;
    p.theta = theta;
    p.saveat = saveat;
    p.u_out = u_out;
    p.stats = stats;
    if (!retcode) {
        if ((rc = ensure(c, c->retcode, sizeof(int32_t) * N))) return rc;
        retcode = (int32_t*)c->retcode.p;
    }
    p.retcode = retcode;
    const int BLOCK = l.block_fwd;
    const int64_t gpb = BLOCK / l.G_fwd;  // trajectories (lane groups) per block
    const unsigned grid = (unsigned)((N + gpb - 1) / gpb);
    const size_t shmem = l.lds_bytes(m->n_param, false);
    if ((rc = setup_time_grids(c, o, N, tspan, p))) return rc;
    if (o->per_trajectory & UDE_PT_THETA) {
        if (!l.per_member) return fail(c, UDE_ERR_UNSUPPORTED, "per-member parameters (UDE_PT_THETA): this model / lanes_per_traj has no per-member kernel (LV-kind compiled instances and the Fisher-KPP kinds have one)");
        p.theta_pm = m->n_param;
    }
    void (*kfwd)(const KParams) = o->per_trajectory ? l.fwd_pt : l.fwd;
    if (shmem > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void*)kfwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    // SEIR exposure UDE / its neural ODE, Float64, shared time grid, lanes_per_traj 0 (default) or 16: the lock-step matrix-core forward kernel
    const bool is_node = model_id(m) == MID_SEIR_NODE;
    const bool gen_ls = UDE_SEIR_LS_FWD && model_id(m) == MID_NONE && seir_gen_ls_fwd_shape(m);
    const bool seir_ls = ((UDE_SEIR_LS_FWD && model_id(m) == MID_SEIR_UDE) || (UDE_NODE_LS_FWD && is_node) || gen_ls) && o->per_trajectory == 0 &&
                         (c->lo.lanes_per_traj == 16 || (c->lo.lanes_per_traj == 0 && (is_node ? UDE_NODE_LS_DEFAULT : UDE_SEIR_LS_DEFAULT)));
    if (seir_ls && (rc = ensure(c, c->ls_fac, 64))) return rc;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    ude_poison_chip(c->stream, true);
    if (seir_ls) {
        void (*lf)(const KParams, int*) = nullptr;
        size_t lf_lds = 0;
        int lf_per_cu = 1;
        (is_node ? ude_node_ls_get_fwd : gen_ls ? ude_seir_ls_get_fwd_gen : ude_seir_ls_get_fwd)(o->alg == UDE_ALG_VERN7 ? 1 : 0, &lf, &lf_lds, &lf_per_cu);
        int* queue = (int*)c->ls_fac.p;
        HIPCHK(c, hipMemsetAsync(queue, 0, sizeof(int), c->stream));
        HIPCHK(c, hipFuncSetAttribute((const void*)lf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf_lds));
        hipLaunchKernelGGL(lf, dim3((unsigned)ls_blocks(c, N, lf_per_cu)), dim3(256), lf_lds, c->stream, p, queue);
    } else
    hipLaunchKernelGGL(kfwd, dim3(grid), dim3(BLOCK), shmem, c->stream, p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    c->ev_fwd = true;
    c->ev_bwd = false;
    return UDE_OK;
}

static int grad_dev_impl(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N, const double* u0,
                         const double* tspan, const double* theta, const double* saveat, int32_t ns,
                         const double* cot_in, const double* data, const uint8_t* row_mask, double* loss,
                         double* loss_per_traj, double* u_out, double* grad_theta, double* grad_u0, int64_t* stats,
                         int32_t* retcode) {
    int rc = common_checks(c, m, o, N, u0, tspan, theta, saveat, ns);
    if (rc) return rc;
    if (!grad_theta) return fail(c, UDE_ERR_INVALID, "grad_theta is null");
    if (!cot_in && !data) return fail(c, UDE_ERR_INVALID, "need a cotangent or data");
    if (m->n_param <= 0) return fail(c, UDE_ERR_UNSUPPORTED, "model has no parameters to differentiate");
    HIPCHK(c, hipSetDevice(c->device));
    Launch l;
    int G;
    // SEIR exposure UDE, interpolating adjoint in parity mode, shared time grid: lanes_per_traj = 16 selects the lock-step
    // matrix-core backward kernel (16 trajectories per block as MFMA columns, csrc/ude_seir_ls.h); the forward kernel is the
    // wavefront-per-trajectory one either way
    const int want_lanes = c->lo.lanes_per_traj;
    const bool seir_ls = model_id(m) == MID_SEIR_UDE && o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT && o->per_trajectory == 0 &&
                         (want_lanes == 16 || (want_lanes == 0 && UDE_SEIR_LS_DEFAULT));
    const bool node_ls = model_id(m) == MID_SEIR_NODE && o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT && o->per_trajectory == 0 &&
                         (want_lanes == 16 || (want_lanes == 0 && UDE_NODE_LS_DEFAULT));
    // ... and in the `fast` mode (lambda-only error control) the block-level matrix-core accumulation of the parameter cotangent:
    // no mu in HBM, one gradient row per BLOCK (csrc/ude_seir_lsf.h); lanes_per_traj = 64 keeps the wavefront-per-trajectory kernel
    const bool seir_lsf = model_id(m) == MID_SEIR_UDE && o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT_FAST && o->per_trajectory == 0 &&
                          (want_lanes == 16 || (want_lanes == 0 && UDE_SEIR_LS_DEFAULT));
    const bool node_lsf = model_id(m) == MID_SEIR_NODE && o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT_FAST && o->per_trajectory == 0 &&
                          (want_lanes == 16 || (want_lanes == 0 && UDE_NODE_LS_DEFAULT));
    // ... and a runtime-shape exposure chain (seir_gen_ls_shape) in the `fast` mode: the runtime-shape instance of the same kernel
    const bool seir_gen_lsf = model_id(m) == MID_NONE && seir_gen_ls_shape(m) && o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT_FAST && o->per_trajectory == 0 &&
                              (want_lanes == 16 || (want_lanes == 0 && UDE_SEIR_LS_DEFAULT));
    const bool any_lsf = seir_lsf || node_lsf || seir_gen_lsf;
    // a runtime-shape exposure UDE 3 -> H1 -> H2 -> 1 (no compiled instance): its backward pass on the lock-step kernel, zero-padded to 64 x 64
    // (round 5: 47.6 -> 11 ms for 3-64-63-1 on the configs[2] share), and so does its forward pass (H2 != 32: seir_gen_ls_fwd_shape)
    const bool seir_gen_ls = model_id(m) == MID_NONE && seir_gen_ls_shape(m) && o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT && o->per_trajectory == 0 &&
                             (want_lanes == 16 || (want_lanes == 0 && UDE_SEIR_LS_DEFAULT));
    const bool any_ls = seir_ls || node_ls || seir_gen_ls;
    const int ls_slk = node_ls ? 146 : 71;   // parameter slots per hidden row (mu: two columns of ls_slk x 64 per trajectory)
    if ((rc = resolve(c, m, o, l, G))) return rc;
    KParams p;
    fill_params(p, m, o, tspan[0], tspan[1]);
    p.tab = tab_for(c, m, o);
    if (c->trace_cap > 0) { p.trace = (double*)c->trace.p; p.trace_traj = c->trace_traj; p.trace_cap = c->trace_cap; }
    const int n = m->n_state, np = m->n_param;
    const size_t es = m->dtype == 1 ? sizeof(float) : sizeof(double);  // bytes of the problem's scalar type
    const int cap = c->lo.max_dense_steps > 0 ? c->lo.max_dense_steps : c->auto_cap;
    const int BLOCK = l.block;
    const int64_t gpb = BLOCK / G;  // trajectories (lane groups) per block
    const unsigned grid = (unsigned)((N + gpb - 1) / gpb);
    int64_t nwaves = (int64_t)grid * ((BLOCK >= 64 && G <= 64) ? BLOCK / 64 : 1);  // rows of the partial-gradient matrix
    void (*ls_kern)(const KParams, double*, int*) = nullptr;
    size_t ls_lds = 0, ls_fac = 0;
    if (any_ls) {
        (seir_gen_ls ? ude_seir_ls_get_gen : seir_ls ? ude_seir_ls_get : ude_node_ls_get)(o->alg == UDE_ALG_VERN7 ? 1 : 0, &ls_kern, &ls_lds, &ls_fac);
        if (nwaves < N) nwaves = N;   // one gradient row per trajectory
    }
    int lsf_per_cu = 1;
    int64_t lsf_blocks = 0;
    if (any_lsf) {
        (seir_gen_lsf ? ude_seir_lsf_get_gen : seir_lsf ? ude_seir_lsf_get : ude_node_lsf_get)(o->alg == UDE_ALG_VERN7 ? 1 : 0, &ls_kern, &ls_lds, &lsf_per_cu);
        lsf_blocks = ls_blocks(c, N, lsf_per_cu);
        // more than one block per CU, but not enough full blocks for all of them: every CU gets the same number of (partly filled)
        // blocks -- the trajectories are dealt round-robin -- instead of some CUs two full blocks and the others one
        if (lsf_per_cu > 1 && lsf_blocks > c->ncu && lsf_blocks < (int64_t)lsf_per_cu * c->ncu) lsf_blocks = (int64_t)lsf_per_cu * c->ncu;
        nwaves = lsf_blocks;          // one gradient row per BLOCK
    }
    const bool ckpt = o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT_CHECKPOINTED;
    // cost-ordered adjoint launch (sort kernels above): lane-group kernels with several trajectories per wavefront, Float64, a loss to sort
    // by, and MORE wavefronts than the chip has SIMDs (a one-round launch ends with its slowest wavefront whatever the order is: the
    // 10 000-trajectory headline is 834 wavefronts on 1024 SIMDs and stays as it was).  UDE_COST_SORT=0 / 1 forces it off / on (A/B runs).
    static const char* cs_env = getenv("UDE_COST_SORT");
    (void)ls_blocks(c, 1);   // (c->ncu)
    const bool cost_sort = !any_ls && !any_lsf && l.adj_sorted && es == 8 && data && !cot_in && !(o->per_trajectory) &&
                           o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT && (cs_env ? atoi(cs_env) != 0 : nwaves > 4 * (int64_t)c->ncu);
    if (cost_sort) nwaves = N;   // one gradient row per trajectory (KParams::perm)
    const int nf = ckpt ? 3 + n : 3 + n + l.nf * n;   // dense fields per accepted step (checkpointed: t, t_end, dt, u)
    p.ckpt = ckpt ? 1 : 0;
    const bool pm = (o->per_trajectory & UDE_PT_THETA) != 0;   // per-member parameters: theta np x N in, grad_theta np x N out
    if (pm && (!l.per_member || any_ls || ckpt || o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT_FAST))
        return fail(c, UDE_ERR_UNSUPPORTED, "per-member parameters (UDE_PT_THETA) exist for the compiled LV-kind instances and the Fisher-KPP kinds (Model::PER_MEMBER_THETA: "
                                            "weights in registers, or read from the member's column in HBM at every use), "
                                            "with the interpolating adjoint or the discrete sweep on the dense store");
    p.theta_pm = pm ? np : 0;
    p.N = N;
    p.Npad = (N + 7) / 8 * 8;
    p.ns = ns;
    p.cap = cap;
    p.u0 = u0;
    p.theta = theta;
    p.saveat = saveat;
    p.u_out = u_out;
    p.data = cot_in ? nullptr : data;
    p.row_mask = row_mask;
    p.cot_in = cot_in;
    if ((rc = ensure(c, c->dense, es * (size_t)cap * nf * p.Npad))) return rc;
    if ((rc = ensure(c, c->dense_n, sizeof(int32_t) * N))) return rc;
    if ((rc = ensure(c, c->cot, es * (size_t)ns * n * p.Npad))) return rc;
    if ((rc = ensure(c, c->loss_traj, es * N))) return rc;
    if (!pm && (rc = ensure(c, c->grad_part, es * (size_t)nwaves * np))) return rc;
    p.perm = nullptr;
    if (cost_sort) {
        if ((rc = ensure(c, c->perm, sizeof(int32_t) * N))) return rc;
        if ((rc = ensure(c, c->sort_ws, sizeof(int32_t) * (2 * SORT_BUCKETS + 1)))) return rc;
        void* before = c->prev_cost.p;
        if ((rc = ensure(c, c->prev_cost, sizeof(int32_t) * N))) return rc;
        if (c->prev_cost.p != before) c->prev_cost_n = 0;   // (a new buffer holds nobody's costs)
    }
    p.slot_glob = nullptr;
    if (any_ls) {  // mu of every trajectory: two columns of 71 (146) slots x 64 hidden rows; the stage factors of every block of 16 slots
        if ((rc = ensure(c, c->slot_glob, sizeof(double) * (size_t)N * 2 * ls_slk * 64))) return rc;
        if ((rc = ensure(c, c->ls_fac, sizeof(double) * (size_t)ls_blocks(c, N) * ls_fac + 64))) return rc;
        p.slot_glob = (double*)c->slot_glob.p;
    } else if (l.slot_glob > 0) {  // slot state mu of the adjoint in HBM: [slot][thread]
        if ((rc = ensure(c, c->slot_glob, es * (size_t)l.slot_glob * grid * BLOCK))) return rc;
        p.slot_glob = (double*)c->slot_glob.p;
    }
    if (!retcode) {
        if ((rc = ensure(c, c->retcode, sizeof(int32_t) * N))) return rc;
        retcode = (int32_t*)c->retcode.p;
    }
    if (!stats) {
        if ((rc = ensure(c, c->stats, sizeof(int64_t) * 8 * N))) return rc;
        stats = (int64_t*)c->stats.p;
    }
    p.stats = stats;
    p.retcode = retcode;
    p.dense = (double*)c->dense.p;
    p.dense_n = (int32_t*)c->dense_n.p;
    p.cot = (double*)c->cot.p;
    p.loss_traj = loss_per_traj ? loss_per_traj : (double*)c->loss_traj.p;
    p.grad_part = pm ? grad_theta : (double*)c->grad_part.p;   // (per-member: the trajectories write the caller's np x N array directly)
    p.grad_u0 = grad_u0;
    const bool discrete = o->sensealg == UDE_SENSE_DISCRETE;
    const bool fast = o->sensealg == UDE_SENSE_INTERPOLATING_ADJOINT_FAST;
    if (o->sensealg != UDE_SENSE_INTERPOLATING_ADJOINT && !discrete && !fast && !ckpt) return fail(c, UDE_ERR_INVALID, "unknown sensealg %d", o->sensealg);
    if ((rc = setup_time_grids(c, o, N, tspan, p))) return rc;
    const bool pt = o->per_trajectory != 0;
    if (fast && pt) return fail(c, UDE_ERR_UNSUPPORTED, "UDE_SENSE_INTERPOLATING_ADJOINT_FAST has no per-trajectory time-grid instances");
    if (ckpt && (pt || !l.adj_ckpt))
        return fail(c, UDE_ERR_UNSUPPORTED, "the checkpointed adjoint (store u only, recompute the stages) runs on a shared time grid, and for a "
                                            "distributed state only where its recomputed stage vectors fit the registers (not: 1024-point Fisher-KPP with Vern7)");
    void (*bwd)(const KParams) = discrete ? (pt ? l.dadj_pt : l.dadj) : fast ? l.adj_fast : ckpt ? l.adj_ckpt : (pt ? l.adj_pt : l.adj);
    if (!bwd) return fail(c, UDE_ERR_UNSUPPORTED, "the runtime-shape kernel has no discretise-then-optimise sweep: use the interpolating adjoint");
    void (*kfwd)(const KParams) = pt ? l.fwd_pt : l.fwd;
    const size_t shmem_f = l.lds_bytes(np, false);
    const size_t shmem_a = l.lds_bytes(np, true, discrete);
    if (shmem_a > 160 * 1024 || shmem_f > 160 * 1024)
        return fail(c, UDE_ERR_UNSUPPORTED, "kernel instance needs %zu / %zu bytes of LDS (forward / backward), more than the 160 KiB of a CU: "
                                            "this (model, lanes_per_traj, sensealg) combination is not available", shmem_f, shmem_a);
    if (shmem_a > 64 * 1024)  // more than the default dynamic-LDS limit: opt in (MI355X has 160 KiB per CU)
        HIPCHK(c, hipFuncSetAttribute((const void*)bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_a));
    if (shmem_a > 64 * 1024 && cost_sort)
        HIPCHK(c, hipFuncSetAttribute((const void*)l.adj_sorted, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_a));
    if (shmem_f > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void*)kfwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_f));
    const bool cap_graph = capturing(c->stream);  // inside a hipGraph capture: the per-kernel timing events are left out
#ifdef UDE_DEBUG_HOOKS
    static const char* ws_fill = getenv("UDE_EXP_WS_FILL");  // debug build: every workspace byte the kernels may read set to a known value first
    if (ws_fill) {
        const int v = atoi(ws_fill);
        HIPCHK(c, hipMemsetAsync(c->dense.p, v, c->dense.cap, c->stream));
        HIPCHK(c, hipMemsetAsync(c->dense_n.p, v, c->dense_n.cap, c->stream));
        HIPCHK(c, hipMemsetAsync(c->cot.p, v, c->cot.cap, c->stream));
        if (c->slot_glob.p) HIPCHK(c, hipMemsetAsync(c->slot_glob.p, v, c->slot_glob.cap, c->stream));
    }
#endif
    HIPCHK(c, hipMemsetAsync(p.grad_part, 0, es * (size_t)(pm ? N : nwaves) * np, c->stream));
    if (!cap_graph) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    ude_poison_chip(c->stream, true);
    const bool gen_fwd = (seir_gen_ls || seir_gen_lsf) && UDE_SEIR_LS_FWD && seir_gen_ls_fwd_shape(m);
    if (((seir_ls || seir_lsf || gen_fwd) && UDE_SEIR_LS_FWD) || ((node_ls || node_lsf) && UDE_NODE_LS_FWD)) {
        void (*lf)(const KParams, int*) = nullptr;
        size_t lf_lds = 0;
        int lf_per_cu = 1;
        ((node_ls || node_lsf) ? ude_node_ls_get_fwd : gen_fwd ? ude_seir_ls_get_fwd_gen : ude_seir_ls_get_fwd)(o->alg == UDE_ALG_VERN7 ? 1 : 0, &lf, &lf_lds, &lf_per_cu);
        const int64_t nblk = ls_blocks(c, N);
        if (any_lsf && (rc = ensure(c, c->ls_fac, sizeof(double) * 2))) return rc;   // (only the forward kernel's queue counter lives there)
        int* queue = (int*)((double*)c->ls_fac.p + (any_lsf ? 0 : (size_t)nblk * ls_fac)) + 1;   // (the backward kernel's counter is the int in front of it)
        HIPCHK(c, hipMemsetAsync(queue, 0, sizeof(int), c->stream));
        HIPCHK(c, hipFuncSetAttribute((const void*)lf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf_lds));
        hipLaunchKernelGGL(lf, dim3((unsigned)ls_blocks(c, N, lf_per_cu)), dim3(256), lf_lds, c->stream, p, queue);
    } else
    {   // (the forward kernel may run with fewer threads per block than the adjoint: Launch::block_fwd)
        if (cost_sort) {
            // the members in the order of what their backward solves cost in the PREVIOUS call on this ensemble (same size, model, algorithm,
            // tolerances, grid size), most expensive first; without such a call: the identity.  Forward and backward kernel run in that order,
            // the internal workspaces are indexed by the lane group's position (adjacent groups, adjacent words: a first version permuted
            // the backward launch alone and read its workspace columns scattered, +7 % on these latency-bound kernels)
            int32_t* hist = (int32_t*)c->sort_ws.p;
            const int64_t sig = ((int64_t)m->kind << 48) ^ ((int64_t)np << 32) ^ ((int64_t)ns << 16) ^ ((int64_t)o->alg << 8) ^ (int64_t)o->sensealg ^
                                (int64_t)(o->abstol * 1e15) ^ ((int64_t)(o->reltol * 1e15) << 1);
            const int32_t* prev = (c->prev_cost_n == N && c->prev_cost_sig == sig && c->prev_cost.p) ? (const int32_t*)c->prev_cost.p : (const int32_t*)nullptr;
            const int ident = (!prev || (cs_env && atoi(cs_env) == 2)) ? 1 : 0;
            if (N <= SORT_SMALL_MAX)
                hipLaunchKernelGGL(sort_small_kernel, dim3(1), dim3(1024), 0, c->stream, (const double*)nullptr, prev, N, (int32_t*)c->perm.p, ident);
            else {
                HIPCHK(c, hipMemsetAsync(hist, 0, sizeof(int32_t) * (2 * SORT_BUCKETS + 1), c->stream));
                if (!ident) {
                    hipLaunchKernelGGL(sort_hist_kernel, dim3((unsigned)((N + 1023) / 1024)), dim3(1024), 0, c->stream, (const double*)nullptr, prev, N, hist);
                    hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const int32_t*)hist, hist + SORT_BUCKETS, 1);
                }
                hipLaunchKernelGGL(sort_scatter_kernel, dim3((unsigned)((N + 1023) / 1024)), dim3(1024), 0, c->stream, (const double*)nullptr, prev, N, hist + SORT_BUCKETS,
                                   (int32_t*)c->perm.p, ident);
            }
            p.perm = (const int32_t*)c->perm.p;
            c->prev_cost_sig = sig;
            if (!cap_graph) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));   // (the forward kernel's own interval)
        }
        const int64_t gpb_f = l.block_fwd / l.G_fwd;
        hipLaunchKernelGGL(cost_sort ? l.fwd_sorted : kfwd, dim3((unsigned)((N + gpb_f - 1) / gpb_f)), dim3(l.block_fwd), shmem_f, c->stream, p);
    }
    HIPCHK(c, hipGetLastError());
    if (!cap_graph) {
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    }
    ude_poison_chip(c->stream, false);
    if (any_ls) {
        // persistent blocks (one per CU at most), trajectories handed out through a queue counter that lives behind the factor workspace
        const int64_t nblk = ls_blocks(c, N);
        int* queue = (int*)((double*)c->ls_fac.p + (size_t)nblk * ls_fac);
        HIPCHK(c, hipMemsetAsync(queue, 0, sizeof(int), c->stream));
        HIPCHK(c, hipFuncSetAttribute((const void*)ls_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ls_lds));
        hipLaunchKernelGGL(ls_kern, dim3((unsigned)nblk), dim3(256), ls_lds, c->stream, p, (double*)c->ls_fac.p, queue);
    } else if (any_lsf) {
        // persistent blocks, trajectories dealt round-robin (no queue: every sum is in the same order in every run)
        HIPCHK(c, hipFuncSetAttribute((const void*)ls_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ls_lds));
        hipLaunchKernelGGL(ls_kern, dim3((unsigned)lsf_blocks), dim3(256), ls_lds, c->stream, p, (double*)nullptr, (int*)nullptr);
    } else {
    if (cost_sort && !cap_graph) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    hipLaunchKernelGGL(cost_sort ? l.adj_sorted : bwd, dim3(grid), dim3(BLOCK), shmem_a, c->stream, p);
    if (cost_sort) {
        HIPCHK(c, hipGetLastError());
        if (!cap_graph) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
        hipLaunchKernelGGL(cost_save_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, (const int64_t*)stats, N, (int32_t*)c->prev_cost.p);
        c->prev_cost_n = N;
    }
    }
    HIPCHK(c, hipGetLastError());
    if (!cap_graph) {
        if (!cost_sort) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
        c->ev_fwd = c->ev_bwd = true;
    }
    if ((rc = ensure(c, c->nfail, sizeof(int32_t)))) return rc;
    double* lossp = (loss && !cot_in) ? loss : (double*)nullptr;
    if (pm) {  // nothing to reduce over trajectories: only the loss sum and the failure count (zero gradient columns)
        if (m->dtype == 1)
            hipLaunchKernelGGL(finish_kernel<float>, dim3(1), dim3(256), 0, c->stream, (const float*)nullptr, (int64_t)0, (int32_t)0, (float*)nullptr,
                               (const float*)p.loss_traj, N, (float*)lossp, (const int32_t*)retcode, (int32_t*)c->nfail.p);
        else
            hipLaunchKernelGGL(finish_kernel<double>, dim3(1), dim3(256), 0, c->stream, (const double*)nullptr, (int64_t)0, (int32_t)0, (double*)nullptr,
                               (const double*)p.loss_traj, N, lossp, (const int32_t*)retcode, (int32_t*)c->nfail.p);
        HIPCHK(c, hipGetLastError());
        return UDE_OK;
    }
    if (cost_sort) {   // N rows -> ceil(N / 1024) chunk rows (many blocks: a hundred MB of rows must not be one block's stream), then the usual finish
        const int64_t nchunks = (N + ROWSUM_CHUNK - 1) / ROWSUM_CHUNK;
        if ((rc = ensure(c, c->rowsum, sizeof(double) * (size_t)nchunks * np))) return rc;
        hipLaunchKernelGGL(rows_chunk_sum_kernel, dim3((unsigned)((np + 31) / 32), (unsigned)nchunks), dim3(1024), 0, c->stream, (const double*)p.grad_part, N, (int32_t)np,
                           (double*)c->rowsum.p);
        HIPCHK(c, hipGetLastError());
        p.grad_part = (double*)c->rowsum.p;
        nwaves = nchunks;
    }
    if (m->dtype == 1)  // Float32 problem: every real-valued array behind these pointers is float
        hipLaunchKernelGGL(finish_kernel<float>, dim3(np + 1), dim3(256), 0, c->stream, (const float*)p.grad_part, nwaves, (int32_t)np,
                           (float*)grad_theta, (const float*)p.loss_traj, N, (float*)lossp, (const int32_t*)retcode, (int32_t*)c->nfail.p);
    else if (np >= 512 && nwaves >= 256)  // wide and tall: coalesced 32-column tiles
        hipLaunchKernelGGL(finish_wide_kernel<double>, dim3((np + 31) / 32 + 1), dim3(1024), 0, c->stream, (const double*)p.grad_part, nwaves,
                           (int32_t)np, grad_theta, (const double*)p.loss_traj, N, lossp, (const int32_t*)retcode, (int32_t*)c->nfail.p);
    else
        hipLaunchKernelGGL(finish_kernel<double>, dim3(np + 1), dim3(256), 0, c->stream, (const double*)p.grad_part, nwaves, (int32_t)np,
                           grad_theta, (const double*)p.loss_traj, N, lossp, (const int32_t*)retcode, (int32_t*)c->nfail.p);
    HIPCHK(c, hipGetLastError());
    return UDE_OK;
}

extern "C" int ude_solve_ensemble_dev(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                                      const ude_real* u0_, const double* tspan, const ude_real* theta_,
                                      const ude_real* saveat_, int32_t ns, ude_real* u_out_, int64_t* stats,
                                      int32_t* retcode) {
    const double* u0 = (const double*)u0_;
    const double* theta = (const double*)theta_;
    const double* saveat = (const double*)saveat_;
    double* u_out = (double*)u_out_;
    return solve_dev_impl(c, m, o, N, u0, tspan, theta, saveat, ns, u_out, stats, retcode);
}

// du = f(u, theta) for N states (device buffers; u and du are n x N, one state per column)
extern "C" int ude_rhs_ensemble_dev(ude_ctx* c, const ude_model_desc* m, int64_t N, const ude_real* u_, const ude_real* theta_, ude_real* du_) {
    const double* u = (const double*)u_;
    const double* theta = (const double*)theta_;
    double* du = (double*)du_;
    if (!c) return UDE_ERR_INVALID;
    if (!m || !u || !du || N <= 0) return fail(c, UDE_ERR_INVALID, "null / empty argument");
    HIPCHK(c, hipSetDevice(c->device));
    ude_solve_opts o{};
    o.alg = UDE_ALG_TSIT5;
    Launch l;
    int G = 0, rc;
    if ((rc = resolve(c, m, &o, l, G))) return rc;
    KParams p{};
    const double tspan0 = 0.0;
    fill_params(p, m, &o, tspan0, 1.0);
    p.tab = tab_for(c, m, &o);
    p.N = N;
    p.Npad = (N + 7) / 8 * 8;
    p.u0 = u;
    p.theta = theta;
    p.u_out = du;
    const int BLOCK = l.block_fwd;
    const int64_t gpb = BLOCK / l.G_fwd;
    const unsigned grid = (unsigned)((N + gpb - 1) / gpb);
    const size_t shmem = l.lds_bytes(m->n_param, false);
    if (shmem > 160 * 1024) return fail(c, UDE_ERR_UNSUPPORTED, "kernel instance needs %zu bytes of LDS", shmem);
    if (shmem > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)l.rhs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(l.rhs, dim3(grid), dim3(BLOCK), shmem, c->stream, p);
    HIPCHK(c, hipGetLastError());
    return UDE_OK;
}

extern "C" int ude_vjp_ensemble_dev(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                                    const ude_real* u0_, const double* tspan, const ude_real* theta_, const ude_real* saveat_,
                                    int32_t ns, const ude_real* cotangent_, ude_real* u_out_, ude_real* grad_theta_,
                                    ude_real* grad_u0_, int64_t* stats, int32_t* retcode) {
    const double* u0 = (const double*)u0_;
    const double* theta = (const double*)theta_;
    const double* saveat = (const double*)saveat_;
    const double* cotangent = (const double*)cotangent_;
    double* u_out = (double*)u_out_;
    double* grad_theta = (double*)grad_theta_;
    double* grad_u0 = (double*)grad_u0_;
    if (c && !cotangent) return fail(c, UDE_ERR_INVALID, "cotangent is null");
    return grad_dev_impl(c, m, o, N, u0, tspan, theta, saveat, ns, cotangent, nullptr, nullptr, nullptr, nullptr, u_out,
                         grad_theta, grad_u0, stats, retcode);
}

extern "C" int ude_loss_grad_ensemble_dev(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                                          const ude_real* u0_, const double* tspan, const ude_real* theta_,
                                          const ude_real* saveat_, int32_t ns, const ude_real* data_,
                                          const uint8_t* row_mask, ude_real* loss_, ude_real* loss_per_traj_,
                                          ude_real* grad_theta_, ude_real* grad_u0_, ude_real* u_out_, int64_t* stats,
                                          int32_t* retcode) {
    const double* u0 = (const double*)u0_;
    const double* theta = (const double*)theta_;
    const double* saveat = (const double*)saveat_;
    const double* data = (const double*)data_;
    double* loss = (double*)loss_;
    double* loss_per_traj = (double*)loss_per_traj_;
    double* grad_theta = (double*)grad_theta_;
    double* grad_u0 = (double*)grad_u0_;
    double* u_out = (double*)u_out_;
    if (c && !data) return fail(c, UDE_ERR_INVALID, "data is null");
    return grad_dev_impl(c, m, o, N, u0, tspan, theta, saveat, ns, nullptr, data, row_mask, loss, loss_per_traj, u_out,
                         grad_theta, grad_u0, stats, retcode);
}

// ---------------------------------------------------------------------------------------------
// host-buffer entry points (what a Julia ccall binds): stage H->D, run, D->H, block
// ---------------------------------------------------------------------------------------------
static int up(ude_ctx* c, DevBuf& b, const void* src, size_t bytes, void** dst) {
    *dst = nullptr;
    if (!src || bytes == 0) return UDE_OK;
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    *dst = b.p;
    return UDE_OK;
}
static int dn(ude_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!dst || !src || bytes == 0) return UDE_OK;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return UDE_OK;
}
static int any_failed(ude_ctx* c, const int32_t* rc, int64_t N) {
    for (int64_t j = 0; j < N; ++j)
        if (rc[j] != UDE_RET_SUCCESS)
            return fail(c, UDE_ERR_TRAJECTORY, "trajectory %lld ended with retcode %d (see retcode array)", (long long)j, rc[j]);
    return UDE_OK;
}

extern "C" int ude_solve_ensemble(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                                  const ude_real* u0_, const double* tspan, const ude_real* theta_, const ude_real* saveat_,
                                  int32_t ns, ude_real* u_out_, int64_t* stats, int32_t* retcode) {
    const double* u0 = (const double*)u0_;
    const double* theta = (const double*)theta_;
    const double* saveat = (const double*)saveat_;
    double* u_out = (double*)u_out_;
    int rc = common_checks(c, m, o, N, u0, tspan, theta, saveat, ns);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = m->n_state;
    const size_t es = m->dtype == 1 ? sizeof(float) : sizeof(double);
    void *du0, *dth, *dsv;
    if ((rc = up(c, c->s_u0, u0, es * n * N, &du0))) return rc;
    if ((rc = up(c, c->s_theta, theta, es * m->n_param * ((o->per_trajectory & UDE_PT_THETA) ? (size_t)N : 1), &dth))) return rc;
    if ((rc = up(c, c->s_saveat, saveat, es * ns * ((o->per_trajectory & UDE_PT_SAVEAT) ? N : 1), &dsv))) return rc;
    if ((rc = ensure(c, c->s_out, es * n * ns * N))) return rc;
    if ((rc = ensure(c, c->s_stats, sizeof(int64_t) * 8 * N))) return rc;
    if ((rc = ensure(c, c->s_ret, sizeof(int32_t) * N))) return rc;
    rc = solve_dev_impl(c, m, o, N, (double*)du0, tspan, (double*)dth, (double*)dsv, ns, (double*)c->s_out.p,
                        (int64_t*)c->s_stats.p, (int32_t*)c->s_ret.p);
    if (rc) return rc;
    std::vector<int32_t> rtmp(N);
    if ((rc = dn(c, u_out, c->s_out.p, es * n * ns * N))) return rc;
    if ((rc = dn(c, stats, c->s_stats.p, sizeof(int64_t) * 8 * N))) return rc;
    if ((rc = dn(c, rtmp.data(), c->s_ret.p, sizeof(int32_t) * N))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (retcode) memcpy(retcode, rtmp.data(), sizeof(int32_t) * N);
    return any_failed(c, rtmp.data(), N);
}

static int grad_host(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N, const double* u0,
                     const double* tspan, const double* theta, const double* saveat, int32_t ns, const double* cot,
                     const double* data, const uint8_t* row_mask, double* loss, double* loss_per_traj, double* u_out,
                     double* grad_theta, double* grad_u0, int64_t* stats, int32_t* retcode) {
    int rc = common_checks(c, m, o, N, u0, tspan, theta, saveat, ns);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = m->n_state, np = m->n_param;
    const size_t es = m->dtype == 1 ? sizeof(float) : sizeof(double);
    void *du0, *dth, *dsv, *ddat, *dmask;
    if ((rc = up(c, c->s_u0, u0, es * n * N, &du0))) return rc;
    const size_t nth = (o->per_trajectory & UDE_PT_THETA) ? (size_t)N : 1;   // parameter vectors in / gradients out
    if ((rc = up(c, c->s_theta, theta, es * np * nth, &dth))) return rc;
    if ((rc = up(c, c->s_saveat, saveat, es * ns * ((o->per_trajectory & UDE_PT_SAVEAT) ? N : 1), &dsv))) return rc;
    if ((rc = up(c, c->s_data, cot ? cot : data, es * n * ns * N, &ddat))) return rc;
    if ((rc = up(c, c->s_mask, row_mask, n, &dmask))) return rc;
    if ((rc = ensure(c, c->s_out, es * n * ns * N))) return rc;
    if ((rc = ensure(c, c->s_stats, sizeof(int64_t) * 8 * N))) return rc;
    if ((rc = ensure(c, c->s_ret, sizeof(int32_t) * N))) return rc;
    if ((rc = ensure(c, c->s_gtheta, es * np * nth))) return rc;
    if ((rc = ensure(c, c->s_gu0, es * n * N))) return rc;
    if ((rc = ensure(c, c->s_loss, es))) return rc;
    if ((rc = ensure(c, c->s_lpt, es * N))) return rc;
    std::vector<int32_t> rtmp(N);
    for (;;) {
        HIPCHK(c, hipMemsetAsync(c->s_stats.p, 0, sizeof(int64_t) * 8 * N, c->stream));
        rc = grad_dev_impl(c, m, o, N, (double*)du0, tspan, (double*)dth, (double*)dsv, ns, cot ? (double*)ddat : nullptr,
                           cot ? nullptr : (double*)ddat, (uint8_t*)dmask, (double*)c->s_loss.p, (double*)c->s_lpt.p,
                           (double*)c->s_out.p, (double*)c->s_gtheta.p, (double*)c->s_gu0.p, (int64_t*)c->s_stats.p,
                           (int32_t*)c->s_ret.p);
        if (rc) return rc;
        if ((rc = dn(c, rtmp.data(), c->s_ret.p, sizeof(int32_t) * N))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // a forward pass that outgrew the dense store is re-run with four times the capacity (the oracle's and
        // upstream's solution arrays simply grow) unless the caller pinned max_dense_steps
        bool overflow = false;
        for (int64_t j = 0; j < N; ++j) overflow = overflow || rtmp[j] == UDE_RET_DENSE_OVERFLOW;
        if (!overflow || c->lo.max_dense_steps > 0 || c->auto_cap >= (1 << 20)) break;
        c->auto_cap *= 4;
    }
    if ((rc = dn(c, u_out, c->s_out.p, es * n * ns * N))) return rc;
    if ((rc = dn(c, stats, c->s_stats.p, sizeof(int64_t) * 8 * N))) return rc;
    if ((rc = dn(c, grad_theta, c->s_gtheta.p, es * np * nth))) return rc;
    if ((rc = dn(c, grad_u0, c->s_gu0.p, es * n * N))) return rc;
    if (!cot) {
        if ((rc = dn(c, loss, c->s_loss.p, es))) return rc;
        if ((rc = dn(c, loss_per_traj, c->s_lpt.p, es * N))) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (retcode) memcpy(retcode, rtmp.data(), sizeof(int32_t) * N);
    return any_failed(c, rtmp.data(), N);
}

extern "C" int ude_vjp_ensemble(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                                const ude_real* u0_, const double* tspan, const ude_real* theta_, const ude_real* saveat_,
                                int32_t ns, const ude_real* cotangent_, ude_real* u_out_, ude_real* grad_theta_,
                                ude_real* grad_u0_, int64_t* stats, int32_t* retcode) {
    const double* u0 = (const double*)u0_;
    const double* theta = (const double*)theta_;
    const double* saveat = (const double*)saveat_;
    const double* cotangent = (const double*)cotangent_;
    double* u_out = (double*)u_out_;
    double* grad_theta = (double*)grad_theta_;
    double* grad_u0 = (double*)grad_u0_;
    if (c && !cotangent) return fail(c, UDE_ERR_INVALID, "cotangent is null");
    return grad_host(c, m, o, N, u0, tspan, theta, saveat, ns, cotangent, nullptr, nullptr, nullptr, nullptr, u_out,
                     grad_theta, grad_u0, stats, retcode);
}

extern "C" int ude_loss_grad_ensemble(ude_ctx* c, const ude_model_desc* m, const ude_solve_opts* o, int64_t N,
                                      const ude_real* u0_, const double* tspan, const ude_real* theta_,
                                      const ude_real* saveat_, int32_t ns, const ude_real* data_, const uint8_t* row_mask,
                                      ude_real* loss_, ude_real* loss_per_traj_, ude_real* grad_theta_, ude_real* grad_u0_,
                                      ude_real* u_out_, int64_t* stats, int32_t* retcode) {
    const double* u0 = (const double*)u0_;
    const double* theta = (const double*)theta_;
    const double* saveat = (const double*)saveat_;
    const double* data = (const double*)data_;
    double* loss = (double*)loss_;
    double* loss_per_traj = (double*)loss_per_traj_;
    double* grad_theta = (double*)grad_theta_;
    double* grad_u0 = (double*)grad_u0_;
    double* u_out = (double*)u_out_;
    if (c && !data) return fail(c, UDE_ERR_INVALID, "data is null");
    return grad_host(c, m, o, N, u0, tspan, theta, saveat, ns, nullptr, data, row_mask, loss, loss_per_traj, u_out,
                     grad_theta, grad_u0, stats, retcode);
}

extern "C" int ude_rhs_ensemble(ude_ctx* c, const ude_model_desc* m, int64_t N, const ude_real* u_, const ude_real* theta_, ude_real* du_) {
    const double* u = (const double*)u_;
    const double* theta = (const double*)theta_;
    double* du = (double*)du_;
    if (!c) return UDE_ERR_INVALID;
    if (!m || !u || !du || N <= 0) return fail(c, UDE_ERR_INVALID, "null / empty argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    void *du0, *dth;
    const size_t es = m->dtype == 1 ? sizeof(float) : sizeof(double);
    const size_t bytes = es * (size_t)N * m->n_state;
    if ((rc = up(c, c->s_u0, u, bytes, &du0))) return rc;
    if ((rc = up(c, c->s_theta, theta, es * (size_t)m->n_param, &dth))) return rc;
    if ((rc = ensure(c, c->s_out, bytes))) return rc;
    if ((rc = ude_rhs_ensemble_dev(c, m, N, (const double*)du0, (const double*)dth, (double*)c->s_out.p))) return rc;
    if ((rc = dn(c, du, c->s_out.p, bytes))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return UDE_OK;
}

extern "C" int ude_math_dev(ude_ctx* c, int32_t op, int64_t n, const double* x, const double* y, double* out) {
    if (!c || !x || !y || !out || n <= 0) return UDE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    void *dx, *dy;
    if ((rc = up(c, c->s_u0, x, sizeof(double) * n, &dx))) return rc;
    if ((rc = up(c, c->s_data, y, sizeof(double) * n, &dy))) return rc;
    if ((rc = ensure(c, c->s_out, sizeof(double) * n))) return rc;
    if (op == 10) {
        if (n % 64) return fail(c, UDE_ERR_INVALID, "op 10 (mfma probe): n must be a multiple of 64");
        hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)(n / 64)), dim3(64), 0, c->stream, (const double*)dx,
                           (const double*)dy, (double*)c->s_out.p);
    } else
    hipLaunchKernelGGL(math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (int)op, (const double*)dx,
                       (const double*)dy, (double*)c->s_out.p, n);
    HIPCHK(c, hipGetLastError());
    if ((rc = dn(c, out, c->s_out.p, sizeof(double) * n))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return UDE_OK;
}

extern "C" int ude_fastpow_dev(ude_ctx* c, int64_t n, const double* x, const double* y, double* out) {
    if (!c || !x || !y || !out || n <= 0) return UDE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    void *dx, *dy;
    if ((rc = up(c, c->s_u0, x, sizeof(double) * n, &dx))) return rc;
    if ((rc = up(c, c->s_data, y, sizeof(double) * n, &dy))) return rc;
    if ((rc = ensure(c, c->s_out, sizeof(double) * n))) return rc;
    hipLaunchKernelGGL(fastpow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const double*)dx,
                       (const double*)dy, (double*)c->s_out.p, n);
    HIPCHK(c, hipGetLastError());
    if ((rc = dn(c, out, c->s_out.p, sizeof(double) * n))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return UDE_OK;
}
