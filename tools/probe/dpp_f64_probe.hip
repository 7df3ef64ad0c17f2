// v_fmac_f64_dpp row_newbcast: semantics (lane k of each 16-lane row broadcast as src0) and issue rate vs plain v_fma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#define X4(a) a a a a
__global__ void sem(double* out) {
    double w = 100.0 + threadIdx.x;        // lane value
    const double one = 1.0; double a = 2.0, z0 = 0.5, z1 = 0.5, z2 = 0.5, z3 = 0.25;
    asm volatile("s_nop 4\n"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %1, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %2, %4, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %3, %4, %6 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                 : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "v"(w), "v"(a), "v"(one));
    out[threadIdx.x * 4 + 0] = z0; out[threadIdx.x * 4 + 1] = z1; out[threadIdx.x * 4 + 2] = z2; out[threadIdx.x * 4 + 3] = z3;
}
template <int MODE>
__global__ void __launch_bounds__(256) rate(double* out, int iters, long long* cycles) {
    double d0 = 1.0 + threadIdx.x * 1e-9, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    const double w = 1.0000001 + threadIdx.x * 1e-12, a = 1e-9;
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#define OPS "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(w), "v"(a)
#define D(n, k) "v_fmac_f64_dpp %" #n ", %8, %9 row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n"
#define F(n) "v_fma_f64 %" #n ", %8, %9, %" #n "\n"
#define C(n) "v_fmac_f64 %" #n ", %8, %9\n"
        if constexpr (MODE == 0) asm volatile(X4(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)) X4(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)) : OPS);
        else if constexpr (MODE == 1) asm volatile(X4(D(0, 0) D(1, 3) D(2, 5) D(3, 7) D(4, 9) D(5, 11) D(6, 13) D(7, 15)) X4(D(0, 1) D(1, 2) D(2, 4) D(3, 6) D(4, 8) D(5, 10) D(6, 12) D(7, 14)) : OPS);
        else asm volatile(X4(C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)) X4(C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)) : OPS);
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, out);
    double ho[256]; (void)hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l & ~15;
        const double e0 = 0.5 + (100.0 + row + 0) * 2.0, e1 = 0.5 + (100.0 + row + 5) * 2.0, e2 = 0.5 + (100.0 + row + 15) * 2.0, e3 = 0.25 + (100.0 + row + 7);
        if (ho[l * 4] != e0 || ho[l * 4 + 1] != e1 || ho[l * 4 + 2] != e2 || ho[l * 4 + 3] != e3) { if (bad < 4) printf("lane %d: %g %g %g %g expected %g %g %g %g\n", l, ho[l*4], ho[l*4+1], ho[l*4+2], ho[l*4+3], e0, e1, e2, e3); ++bad; }
    }
    printf("row_newbcast semantics: %s\n", bad ? "UNEXPECTED" : "lane k of each 16-lane row is src0 for all lanes of the row: ok");
    const int iters = 20000;
#define RUN(MODE, NAME) hipLaunchKernelGGL(rate<MODE>, dim3(1), dim3(256), 0, 0, out, iters, cyc); (void)hipDeviceSynchronize(); \
    hipLaunchKernelGGL(rate<MODE>, dim3(1), dim3(256), 0, 0, out, iters, cyc); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-40s %7.2f shader clocks per instruction\n", NAME, (double)h / iters * 24.0 / 64);
    RUN(0, "v_fma_f64 (8 chains)")
    RUN(2, "v_fmac_f64 (8 chains)")
    RUN(1, "v_fmac_f64_dpp row_newbcast (8 chains)")
    return 0;
}
