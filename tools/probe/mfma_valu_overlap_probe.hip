// Do v_mfma_f64_16x16x4 and FP64 vector instructions overlap on gfx950 -- inside one wavefront, and between two wavefronts
// of one SIMD?  (The part's FP64 matrix peak equals its FP64 vector peak; a design that wants `tanh` to run "under" the
// matrix products has to know whether the two share a datapath.)  Every loop body is one asm volatile block of independent
// instructions, so the order in the text is the order of issue.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_valu_overlap_probe mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef double v4d __attribute__((ext_vector_type(4)));

#define MF(n) "v_mfma_f64_16x16x4_f64 %" #n ", %8, %9, %" #n "\n"
#define F64(n) "v_fma_f64 %" #n ", %" #n ", %10, %11\n"
#define F32(n) "v_fma_f32 %" #n ", %" #n ", %12, %13\n"
#define I32(n) "v_add_u32 %" #n ", %" #n ", %14\n"
#define RCP(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define X4(a) a a a a
#define X8(a) X4(a) X4(a)
#define X16(a) X8(a) X8(a)

// MODE: 0 = 4 MFMAs only; 1 = VALU block only (KIND, 16 x 4 instructions); 2 = each MFMA followed by 16 VALU instructions
// KIND: 0 v_fma_f64, 1 v_fma_f32, 2 v_add_u32, 3 v_rcp_f64
template <int MODE, int KIND>
__device__ __forceinline__ void body(v4d& m0, v4d& m1, v4d& m2, v4d& m3, double& d0, double& d1, double& d2, double& d3, float& f0, float& f1,
                                     float& f2, float& f3, unsigned& i0, unsigned& i1, unsigned& i2, unsigned& i3, double a, double b) {
    const double ca = 1.0000001, cb = 1e-9;
    const float fa = 1.0001f, fb = 1e-6f;
    const unsigned ic = 3u;
#define OPS "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(ca), "v"(cb), "v"(fa), "v"(fb), "v"(ic)
#define VB64 X4(F64(4) F64(5) F64(6) F64(7))
#define VBR X4(RCP(4) RCP(5) RCP(6) RCP(7))
    if constexpr (KIND == 0 || KIND == 3) {
        if constexpr (MODE == 0) asm volatile(MF(0) MF(1) MF(2) MF(3) : OPS);
        else if constexpr (MODE == 1 && KIND == 0) asm volatile(VB64 VB64 VB64 VB64 : OPS);
        else if constexpr (MODE == 2 && KIND == 0) asm volatile(MF(0) VB64 MF(1) VB64 MF(2) VB64 MF(3) VB64 : OPS);
        else if constexpr (MODE == 1 && KIND == 3) asm volatile(VBR VBR VBR VBR : OPS);
        else asm volatile(MF(0) VBR MF(1) VBR MF(2) VBR MF(3) VBR : OPS);
    } else if constexpr (KIND == 1) {
#define OPSF "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b), "v"(ca), "v"(cb), "v"(fa), "v"(fb), "v"(ic)
#define VB32 X4(F32(4) F32(5) F32(6) F32(7))
        if constexpr (MODE == 1) asm volatile(VB32 VB32 VB32 VB32 : OPSF);
        else asm volatile(MF(0) VB32 MF(1) VB32 MF(2) VB32 MF(3) VB32 : OPSF);
    } else {
#define OPSI "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a), "v"(b), "v"(ca), "v"(cb), "v"(fa), "v"(fb), "v"(ic)
#define VBI X4(I32(4) I32(5) I32(6) I32(7))
        if constexpr (MODE == 1) asm volatile(VBI VBI VBI VBI : OPSI);
        else asm volatile(MF(0) VBI MF(1) VBI MF(2) VBI MF(3) VBI : OPSI);
    }
}

// ROLE: 0 = every wavefront runs MODE; 1 = wavefronts 0..3 of the block run MFMAs only and wavefronts 4..7 the VALU block only
// (two wavefronts per SIMD with an 8-wavefront block: does the vector work of one run under the matrix work of the other?)
template <int MODE, int KIND, int ROLE>
__global__ void __launch_bounds__(512) probe(double* out, int iters, long long* cycles) {
    v4d m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    double d0 = 1.0 + threadIdx.x * 1e-9, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    float f0 = 1.0f + threadIdx.x * 1e-6f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    unsigned i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    const long long t0 = wall_clock64();
    if constexpr (ROLE == 0) {
        for (int i = 0; i < iters; ++i) body<MODE, KIND>(m0, m1, m2, m3, d0, d1, d2, d3, f0, f1, f2, f3, i0, i1, i2, i3, a, b);
    } else {
        if (w < 4) for (int i = 0; i < iters; ++i) body<0, 0>(m0, m1, m2, m3, d0, d1, d2, d3, f0, f1, f2, f3, i0, i1, i2, i3, a, b);
        else for (int i = 0; i < iters; ++i) body<1, KIND>(m0, m1, m2, m3, d0, d1, d2, d3, f0, f1, f2, f3, i0, i1, i2, i3, a, b);
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = m0[0] + m1[1] + m2[2] + m3[3] + d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3 + i0 + i1 + i2 + i3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}


// ---- the same question for the FP32 matrix instructions (the deep-BSDE kernels): 16 accumulator registers per 32x32 product
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define MF32(n) "v_mfma_f32_32x32x2_f32 %" #n ", %6, %7, %" #n "\n"
#define MF16(n) "v_mfma_f32_16x16x4_f32 %" #n ", %6, %7, %" #n "\n"
#define G64(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
#define G32(n) "v_fma_f32 %" #n ", %" #n ", %10, %11\n"
template <int MODE, int KIND, int ROLE>
__global__ void __launch_bounds__(512) probe32(double* out, int iters, long long* cycles) {
    v16f m0, m1; v4f n0 = {0, 0, 0, 0}, n1 = n0;
    for (int i = 0; i < 16; ++i) { m0[i] = 0; m1[i] = 0; }
    double d0 = 1.0 + threadIdx.x * 1e-9, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    float f0 = 1.0f + threadIdx.x * 1e-6f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    const double ca = 1.0000001, cb = 1e-9;
    const float fa = 1.0001f, fb = 1e-6f;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    const long long t0 = wall_clock64();
    const bool mat = ROLE == 0 ? (MODE != 1) : (w < 4), vec = ROLE == 0 ? (MODE != 0) : (w >= 4);
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 4) {        // 2 x 32x32x2 (64 cycles each at 256 flop/clk/CU... measured here) + v_fma_f64
#define O32 "+v"(m0), "+v"(m1), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b), "v"(ca), "v"(cb), "v"(fa), "v"(fb)
#define W64 X4(G64(2) G64(3) G64(4) G64(5))
            if (mat && vec) asm volatile(MF32(0) W64 W64 MF32(1) W64 W64 : O32);
            else if (mat) asm volatile(MF32(0) MF32(1) : O32);
            else asm volatile(W64 W64 W64 W64 : O32);
        } else if constexpr (KIND == 6) { // ... + v_fma_f32
#define O32F "+v"(m0), "+v"(m1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b), "v"(ca), "v"(cb), "v"(fa), "v"(fb)
#define W32 X4(G32(2) G32(3) G32(4) G32(5))
            if (mat && vec) asm volatile(MF32(0) W32 W32 MF32(1) W32 W32 : O32F);
            else if (mat) asm volatile(MF32(0) MF32(1) : O32F);
            else asm volatile(W32 W32 W32 W32 : O32F);
        } else {                          // KIND 5: 2 x 16x16x4 f32 + v_fma_f32
#define O16F "+v"(n0), "+v"(n1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b), "v"(ca), "v"(cb), "v"(fa), "v"(fb)
            if (mat && vec) asm volatile(MF16(0) W32 W32 MF16(1) W32 W32 : O16F);
            else if (mat) asm volatile(MF16(0) MF16(1) : O16F);
            else asm volatile(W32 W32 W32 W32 : O16F);
        }
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = m0[0] + m1[1] + n0[0] + n1[1] + d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    double* out; long long* cyc; long long h;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
    int clk_khz = 0, wall_khz = 0;
    (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("shader clock %d kHz, wall clock counter %d kHz; one body = 4 MFMA and / or 64 vector instructions\n", clk_khz, wall_khz);
    const double tick_to_clk = (double)clk_khz / wall_khz;
#define RUN(MODE, KIND, ROLE, THREADS, NAME)                                                       \
    hipLaunchKernelGGL((probe<MODE, KIND, ROLE>), dim3(1), dim3(THREADS), 0, 0, out, iters, cyc);   \
    (void)hipDeviceSynchronize();                                                                  \
    hipLaunchKernelGGL((probe<MODE, KIND, ROLE>), dim3(1), dim3(THREADS), 0, 0, out, iters, cyc);   \
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                            \
    printf("%-86s %9.1f shader clocks per body\n", NAME, (double)h / iters * tick_to_clk);
    puts("-- one wavefront per SIMD (block of 256)");
    RUN(0, 0, 0, 256, "4 x v_mfma_f64_16x16x4 (independent)")
    RUN(1, 0, 0, 256, "64 x v_fma_f64")
    RUN(2, 0, 0, 256, "4 x (mfma f64 + 16 v_fma_f64) interleaved")
    RUN(1, 1, 0, 256, "64 x v_fma_f32")
    RUN(2, 1, 0, 256, "4 x (mfma f64 + 16 v_fma_f32) interleaved")
    RUN(1, 2, 0, 256, "64 x v_add_u32")
    RUN(2, 2, 0, 256, "4 x (mfma f64 + 16 v_add_u32) interleaved")
    RUN(1, 3, 0, 256, "64 x v_rcp_f64")
    RUN(2, 3, 0, 256, "4 x (mfma f64 + 16 v_rcp_f64) interleaved")
    puts("-- two wavefronts per SIMD (block of 512): both run the same body");
    RUN(0, 0, 0, 512, "4 x mfma f64, two wavefronts per SIMD")
    RUN(1, 0, 0, 512, "64 x v_fma_f64, two wavefronts per SIMD")
    RUN(2, 0, 0, 512, "interleaved f64, two wavefronts per SIMD")
    puts("-- two wavefronts per SIMD: wavefronts 0-3 matrix only, wavefronts 4-7 vector only (same iteration count)");
    RUN(0, 0, 1, 512, "mfma f64 || v_fma_f64")
    RUN(0, 1, 1, 512, "mfma f64 || v_fma_f32")
    RUN(0, 2, 1, 512, "mfma f64 || v_add_u32")
    RUN(0, 3, 1, 512, "mfma f64 || v_rcp_f64")
#define RUN32(MODE, KIND, ROLE, THREADS, NAME)                                                     \
    hipLaunchKernelGGL((probe32<MODE, KIND, ROLE>), dim3(1), dim3(THREADS), 0, 0, out, iters, cyc); \
    (void)hipDeviceSynchronize();                                                                  \
    hipLaunchKernelGGL((probe32<MODE, KIND, ROLE>), dim3(1), dim3(THREADS), 0, 0, out, iters, cyc); \
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                            \
    printf("%-86s %9.1f shader clocks per body\n", NAME, (double)h / iters * tick_to_clk);
    puts("-- FP32 matrix instructions, one wavefront per SIMD; one body = 2 MFMA and / or 64 vector instructions");
    RUN32(0, 4, 0, 256, "2 x v_mfma_f32_32x32x2")
    RUN32(1, 4, 0, 256, "64 x v_fma_f64")
    RUN32(2, 4, 0, 256, "2 x (mfma f32 32x32x2 + 32 v_fma_f64) interleaved")
    RUN32(1, 6, 0, 256, "64 x v_fma_f32")
    RUN32(2, 6, 0, 256, "2 x (mfma f32 32x32x2 + 32 v_fma_f32) interleaved")
    RUN32(0, 5, 0, 256, "2 x v_mfma_f32_16x16x4")
    RUN32(2, 5, 0, 256, "2 x (mfma f32 16x16x4 + 32 v_fma_f32) interleaved")
    puts("-- FP32 matrix instructions, two wavefronts per SIMD: wavefronts 0-3 matrix only, 4-7 vector only");
    RUN32(2, 4, 1, 512, "mfma f32 32x32x2 || v_fma_f64")
    RUN32(2, 6, 1, 512, "mfma f32 32x32x2 || v_fma_f32")
    RUN32(2, 5, 1, 512, "mfma f32 16x16x4 || v_fma_f32")
    return 0;
}
