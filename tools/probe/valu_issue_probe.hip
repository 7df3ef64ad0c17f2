// Issue cost (shader clocks per instruction, one wavefront per SIMD, 8 independent chains unless noted) of the FP64 vector
// instructions the UDE kernels are made of -- the numbers DESIGN.md prices instruction mixes with.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_issue_probe valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define X4(a) a a a a
#define X8(a) X4(a) X4(a)
#define R8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define OPS "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(w), "v"(a), "s"(sc), "v"(x0)
// %8..%11 four ints, %12 w, %13 a, %14 sc (SGPR pair), %15 x0 (one more distinct VGPR source)
#define FMA3(n) "v_fma_f64 %" #n ", %12, %13, %" #n "\n"          /* three VGPR sources, one of them the destination */
#define FMA3D(n) "v_fma_f64 %" #n ", %12, %13, %15\n"            /* three VGPR sources, none the destination */
#define FMAS(n) "v_fma_f64 %" #n ", %" #n ", %13, %14\n"          /* Horner step with a scalar addend */
#define FMAC(n) "v_fmac_f64 %" #n ", %12, %13\n"
#define FMAK(n) "v_fma_f64 %" #n ", %" #n ", %13, 1.0\n"          /* inline-constant addend */
#define MUL(n) "v_mul_f64 %" #n ", %" #n ", %13\n"
#define MUL2(n) "v_mul_f64 %" #n ", %12, %13\n"
#define ADD(n) "v_add_f64 %" #n ", %" #n ", %13\n"
#define CND(n) "v_cndmask_b32 %8, %8, %9, vcc\nv_cndmask_b32 %10, %10, %11, vcc\n"    /* (writes the low half only: same issue cost) */
#define LDEXP(n) "v_ldexp_f64 %" #n ", %" #n ", %8\n"
#define RND(n) "v_rndne_f64 %" #n ", %" #n "\n"
#define RCP(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define CMP(n) "v_cmp_gt_f64 vcc, %" #n ", %13\n"
#define MOV64(n) "v_mov_b64 %" #n ", %12\n"
#define MOV32(n) "v_mov_b32 %8, %9\nv_mov_b32 %10, %11\n"
#define ADDU(n) "v_add_u32 %8, %8, %9\nv_add_u32 %10, %10, %11\n"
#define F32(n) "v_fma_f32 %8, %8, %9, %9\nv_fma_f32 %10, %10, %11, %11\n"
#define NOP(n) "s_nop 0\n"
#define DEP(n) "v_fma_f64 %0, %0, %13, %14\n"                      /* ONE dependent chain */
#define DEPC(n) "v_fmac_f64 %0, %12, %13\n"
#define DEP2(n) "v_fma_f64 %0, %0, %13, %14\nv_fma_f64 %1, %1, %13, %14\n"
template <int M>
__global__ void __launch_bounds__(256) rate(double* out, int iters, long long* cycles) {
    double d0 = 1.0 + threadIdx.x * 1e-9, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    const double w = 1.0000001 + threadIdx.x * 1e-12, a = 1e-9, x0 = 3.0;
    const double sc = 0.5;
    int i0 = 1, i1 = 2, i2 = 3, i3 = 4;
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if constexpr (M == 0) asm volatile(X8(R8(FMA3)) : OPS : "vcc");
        else if constexpr (M == 1) asm volatile(X8(R8(FMAS)) : OPS : "vcc");
        else if constexpr (M == 2) asm volatile(X8(R8(FMAC)) : OPS : "vcc");
        else if constexpr (M == 3) asm volatile(X8(R8(MUL)) : OPS : "vcc");
        else if constexpr (M == 4) asm volatile(X8(R8(ADD)) : OPS : "vcc");
        else if constexpr (M == 5) asm volatile(X4(R8(CND)) : OPS : "vcc");
        else if constexpr (M == 6) asm volatile(X8(R8(LDEXP)) : OPS : "vcc");
        else if constexpr (M == 7) asm volatile(X8(R8(RND)) : OPS : "vcc");
        else if constexpr (M == 8) asm volatile(X8(R8(RCP)) : OPS : "vcc");
        else if constexpr (M == 9) asm volatile(X8(R8(CMP)) : OPS : "vcc");
        else if constexpr (M == 10) asm volatile(X8(R8(MOV64)) : OPS : "vcc");
        else if constexpr (M == 11) asm volatile(X4(R8(ADDU)) : OPS : "vcc");
        else if constexpr (M == 12) asm volatile(X8(R8(DEP)) : OPS : "vcc");
        else if constexpr (M == 13) asm volatile(X8(R8(DEPC)) : OPS : "vcc");
        else if constexpr (M == 14) asm volatile(X8(X4(DEP2(0))) : OPS : "vcc");
        else if constexpr (M == 15) asm volatile(X8(R8(FMAK)) : OPS : "vcc");
        else if constexpr (M == 16) asm volatile(X8(R8(FMA3D)) : OPS : "vcc");
        else if constexpr (M == 17) asm volatile(X8(R8(MUL2)) : OPS : "vcc");
        else if constexpr (M == 18) asm volatile(X8(R8(NOP)) : OPS : "vcc");
        else if constexpr (M == 19) asm volatile(X4(R8(F32)) : OPS : "vcc");
        else if constexpr (M == 20) asm volatile(X4(R8(MOV32)) : OPS : "vcc");
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + i0 + i1 + i2 + i3;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
#define RUN(M, NAME) hipLaunchKernelGGL(rate<M>, dim3(1), dim3(256), 0, 0, out, iters, cyc); (void)hipDeviceSynchronize(); \
    hipLaunchKernelGGL(rate<M>, dim3(1), dim3(256), 0, 0, out, iters, cyc); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-64s %7.2f\n", NAME, (double)h / iters * 24.0 / 64);
    puts("shader clocks per instruction, one wavefront per SIMD");
    RUN(0, "v_fma_f64 d, v, v, d (3 VGPR sources)")
    RUN(16, "v_fma_f64 d, v, v, v (3 VGPR sources, none the destination)")
    RUN(1, "v_fma_f64 d, d, v, s (scalar addend: the Horner step)")
    RUN(15, "v_fma_f64 d, d, v, 1.0 (inline constant addend)")
    RUN(2, "v_fmac_f64 d, v, v")
    RUN(3, "v_mul_f64 d, d, v")
    RUN(17, "v_mul_f64 d, v, v")
    RUN(4, "v_add_f64 d, d, v")
    RUN(5, "v_cndmask_b32")
    RUN(6, "v_ldexp_f64")
    RUN(7, "v_rndne_f64")
    RUN(8, "v_rcp_f64")
    RUN(9, "v_cmp_gt_f64 vcc")
    RUN(10, "v_mov_b64")
    RUN(20, "v_mov_b32")
    RUN(11, "v_add_u32")
    RUN(19, "v_fma_f32")
    RUN(18, "s_nop 0")
    RUN(12, "v_fma_f64 d, d, v, s: ONE dependent chain")
    RUN(14, "v_fma_f64 d, d, v, s: TWO interleaved dependent chains")
    RUN(13, "v_fmac_f64: ONE dependent chain")
    return 0;
}
