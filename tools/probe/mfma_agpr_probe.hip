#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define MD(n) "v_mfma_f64_16x16x4_f64 %" #n ", %4, %5, %" #n "\n"
template <int MODE>
__global__ void __launch_bounds__(256) probe(double* out, int iters, long long* cycles) {
    v4d q0 = {0,0,0,0}, q1 = q0, q2 = q0, q3 = q0;
    const double da = 1.0 + threadIdx.x * 1e-9, db = 1.0 - threadIdx.x * 1e-9;
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) asm volatile(MD(0) MD(1) MD(2) MD(3) : "+a"(q0), "+a"(q1), "+a"(q2), "+a"(q3) : "v"(da), "v"(db));
        else if constexpr (MODE == 1) asm volatile(MD(0) MD(1) MD(2) MD(3) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(da), "v"(db));
        else if constexpr (MODE == 2) asm volatile(MD(0) MD(0) MD(0) MD(0) : "+a"(q0), "+a"(q1), "+a"(q2), "+a"(q3) : "v"(da), "v"(db));
        else asm volatile(MD(0) MD(0) MD(0) MD(0) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(da), "v"(db));
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = q0[0] + q1[1] + q2[0] + q3[0];
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
#define RUN(M, NAME) hipLaunchKernelGGL(probe<M>, dim3(1), dim3(256), 0, 0, out, iters, cyc); (void)hipDeviceSynchronize(); \
    hipLaunchKernelGGL(probe<M>, dim3(1), dim3(256), 0, 0, out, iters, cyc); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-60s %7.2f clocks per MFMA\n", NAME, (double)h / iters * 24.0 / 4);
    RUN(0, "f64 16x16x4, AGPR accumulators, 4 independent")
    RUN(1, "f64 16x16x4, VGPR accumulators, 4 independent")
    RUN(2, "f64 16x16x4, AGPR accumulator, 1 dependent chain")
    RUN(3, "f64 16x16x4, VGPR accumulator, 1 dependent chain")
    return 0;
}
