// Issue rate and dependent-chain latency of v_mfma_f64_16x16x4 (and, for comparison, a dependent v_fma_f64 chain) on one
// wavefront per SIMD: what a design that puts ARITH-SPEC chains on the matrix cores has to schedule around.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_rate_probe mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void mfma_loop(double* out, int iters, long long* cycles) {
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    v4d acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    const long long t1 = clock64();
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
__global__ void fma2_loop(double* out, int iters, long long* cycles) {  // two independent chains
    double x = 1.0 + threadIdx.x * 1e-9, y = 2.0 + threadIdx.x * 1e-9;
    const double a = 1.0000001, b = 1e-9;
    const long long t0 = clock64();
    for (int i = 0; i < iters / 16; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) { x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
__global__ void fma_loop(double* out, int iters, long long* cycles) {
    double x = 1.0 + threadIdx.x * 1e-9;
    const double a = 1.0000001, b = 1e-9;
    const long long t0 = clock64();
    for (int i = 0; i < iters / 16; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) x = __builtin_fma(x, a, b);
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    double* out; long long* cyc; long long h;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
    const int iters = 4096;
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    int wall_khz = 0; (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("shader clock %d kHz, clock64 counter %d kHz\n", clk_khz, wall_khz);
#define RUN(K, NAME, PER)                                                                          \
    hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, out, iters, cyc);                              \
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                            \
    printf("%-44s %8.2f counter ticks per instruction\n", NAME, (double)h / iters / (PER));
    RUN(mfma_loop<1>, "mfma f64 16x16x4, 1 dependent chain", 1)
    RUN(mfma_loop<2>, "mfma f64 16x16x4, 2 independent chains", 2)
    RUN(mfma_loop<4>, "mfma f64 16x16x4, 4 independent chains", 4)
    RUN(mfma_loop<8>, "mfma f64 16x16x4, 8 independent chains", 8)
    RUN(fma_loop, "v_fma_f64, 1 dependent chain (x16 unrolled)", 1)
    RUN(fma2_loop, "v_fma_f64, 2 independent chains", 2)
    return 0;
}
