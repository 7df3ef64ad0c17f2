// ude_math.h -- device scalar math for the UDE core (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ude_real.h"

namespace ude {

// ---------------------------------------------------------------------------------------------
// DiffEqBase.fastpow (the PI controller's power function; upstream DiffEqBase 6.94.4, exercised by
// every adaptive solve in the reference, e.g. LotkaVolterra/scenario_1.jl:84,206).  Evaluated in
// Float32: Float64(exp2(Float32(y) * fastlog2(Float32(x)))).  Every operation below is a single
// correctly-rounded IEEE op (no contraction), so the device result is bit-identical to the CPU
// oracle's -- the accept/reject sequence of a trajectory must not depend on where it runs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fastlog2(float x) {
#pragma clang fp contract(off)  // every operation rounds separately, exactly like the Julia source
    const float a = 0.338953f, b = 2.198599f, c = 1.523692f;
    const uint32_t ux = __float_as_uint(x);
    const uint32_t ex = (ux & 0x7F800000u) >> 23;
    float fexp, signif;
    if (ux & 0x00400000u) {  // significand > 1.5
        signif = __uint_as_float((ux & 0x007FFFFFu) | 0x3f000000u);
        fexp = (float)ex - 126.0f;
    } else {
        signif = __uint_as_float((ux & 0x007FFFFFu) | 0x3f800000u);
        fexp = (float)ex - 127.0f;
    }
    signif = signif - 1.0f;
    const float t1 = a * signif;
    const float t2 = t1 + b;
    const float t3 = signif * t2;
    const float t4 = signif + c;
    const float t5 = __fdiv_rn(t3, t4);
    return fexp + t5;
}

// the integer-valued double k (|k| < 2^31) as int32 WITHOUT a float -> int cast: the low word of k + 1.5 * 2^52 (exact).  A cast
// of a NaN is undefined in the language (poison in LLVM), and exp / tanh have no NaN test in front of it; this is defined for
// every bit pattern (a NaN gives some integer, and ldexp(NaN, anything) is NaN) and costs the one v_add_f64 the v_cvt_i32_f64 did.
__device__ __forceinline__ int exponent_of(double k) { return __double2loint(k + 6755399441055744.0); }

// correctly-rounded-by-construction exp2 for Float32 arguments (Julia evaluates its Float32 exp2
// kernel in Float64 and rounds once): Taylor-13 of e^z in Float64 with a fixed fma order.
__device__ __forceinline__ float exp2_f32(float x) {
#pragma clang fp contract(off)
    // (branch-free like dexp below: the special cases are selects at the end, the arithmetic runs for every argument)
    const double xd = (double)x;
    const double n = __builtin_rint(xd);
    const double z = (xd - n) * 0.6931471805599453;  // xd - n is exact
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, z, 1.0 / 479001600.0);
    p = __builtin_fma(p, z, 1.0 / 39916800.0);
    p = __builtin_fma(p, z, 1.0 / 3628800.0);
    p = __builtin_fma(p, z, 1.0 / 362880.0);
    p = __builtin_fma(p, z, 1.0 / 40320.0);
    p = __builtin_fma(p, z, 1.0 / 5040.0);
    p = __builtin_fma(p, z, 1.0 / 720.0);
    p = __builtin_fma(p, z, 1.0 / 120.0);
    p = __builtin_fma(p, z, 1.0 / 24.0);
    p = __builtin_fma(p, z, 1.0 / 6.0);
    p = __builtin_fma(p, z, 0.5);
    p = __builtin_fma(p, z, 1.0);
    p = __builtin_fma(p, z, 1.0);
    float res = (float)__builtin_ldexp(p, exponent_of(n));
    res = x > 127.0f ? __builtin_inff() : res;
    res = x < -126.0f ? 0.0f : res;
    return x != x ? x : res;
}

__device__ __forceinline__ double fastpow(double x, double y) {
#pragma clang fp contract(off)
    const float prod = (float)y * fastlog2((float)x);
    return (double)exp2_f32(prod);
}

// ---------------------------------------------------------------------------------------------
// ARITH-SPEC elementary functions.  The accept/reject sequence of an adaptive solve is chaotic in the
// last bit of every intermediate (a sliver step in front of a tstop has a pure-rounding-noise error
// estimate), so the kernels evaluate exp / tanh / log10 / 10^x with a FIXED sequence of IEEE operations
// (explicit fma, no contraction: the library is compiled with -ffp-contract=off).  The CPU oracle restates
// the same sequences independently (oracle/ude_oracle.c); tests/test_gpu_parity.py checks bitwise equality.
// ---------------------------------------------------------------------------------------------
// Horner steps of the exp / tanh Taylor polynomial, p = 1/13!; p = fma(p, r, 1/12!); ... ; p = fma(p, r, 1/3!), as ONE block of
// three-address v_fma_f64.  Left to itself the compiler picks the two-address v_fmac_f64 for a Horner step and first copies the
// coefficient into the destination (one v_mov_b64, or two literal v_mov_b32, per term), and with one wavefront per SIMD every
// instruction is an issue slot: 13 of the ~57 slots of a tanh in the Fisher-KPP network loop were such copies.  Written as separate
// asm statements the hazard recogniser pads each with an s_nop (a slot again); inside one block the dependent v_fma_f64 need none
// (ordinary VALU -> VALU dependences are interlocked).  Same operations, same order, same rounding as the __builtin_fma chain.
// The ten addends of the Horner steps are wave-uniform constants: as SCALAR operands (one SGPR pair per v_fma_f64: the constant
// bus of gfx9-class VOP3 carries one) they are materialised by s_mov_b32 on the scalar unit -- or stay resident in SGPRs --
// instead of by two v_mov_b32 (or v_accvgpr_read) each on the vector unit in front of every call: 20 vector issue slots less per
// exp / tanh where the compiler does not find the registers to hoist them (the LV kernels: every call), round 4.
#ifndef UDE_TAYLOR_C
#define UDE_TAYLOR_C(x) "s"(x)
#endif
__device__ __forceinline__ double taylor_13_to_3(double r) {
#if defined(UDE_EXP_ESTRIN)
    // TIMING EXPERIMENT ONLY (results differ in the last bits: not ARITH-SPEC, never shipped): the degree-10 polynomial c3 + c4 r + ... +
    // c13 r^10 by Estrin's scheme -- dependent depth 4 instead of 10, 13 instead of 10 instructions.  Round 5 measured what the shorter
    // chain is worth on the latency-bound LV kernels before deciding against a re-specification of exp / tanh.
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = __builtin_fma(1.0 / 24.0, r, 1.0 / 6.0), a1 = __builtin_fma(1.0 / 720.0, r, 1.0 / 120.0), a2 = __builtin_fma(1.0 / 40320.0, r, 1.0 / 5040.0),
                 a3 = __builtin_fma(1.0 / 3628800.0, r, 1.0 / 362880.0), a4 = __builtin_fma(1.0 / 479001600.0, r, 1.0 / 39916800.0);
    const double b0 = __builtin_fma(a1, r2, a0), b1 = __builtin_fma(a3, r2, a2), b2 = __builtin_fma(1.0 / 6227020800.0, r2, a4);
    return __builtin_fma(b2, r8, __builtin_fma(b1, r4, b0));
#endif
    double p;
    asm("v_fma_f64 %0, %2, %1, %3\n\t"
        "v_fma_f64 %0, %0, %1, %4\n\t"
        "v_fma_f64 %0, %0, %1, %5\n\t"
        "v_fma_f64 %0, %0, %1, %6\n\t"
        "v_fma_f64 %0, %0, %1, %7\n\t"
        "v_fma_f64 %0, %0, %1, %8\n\t"
        "v_fma_f64 %0, %0, %1, %9\n\t"
        "v_fma_f64 %0, %0, %1, %10\n\t"
        "v_fma_f64 %0, %0, %1, %11\n\t"
        "v_fma_f64 %0, %0, %1, %12"
        : "=&v"(p)
        : "v"(r), "v"(1.0 / 6227020800.0), UDE_TAYLOR_C(1.0 / 479001600.0), UDE_TAYLOR_C(1.0 / 39916800.0), UDE_TAYLOR_C(1.0 / 3628800.0),
          UDE_TAYLOR_C(1.0 / 362880.0), UDE_TAYLOR_C(1.0 / 40320.0), UDE_TAYLOR_C(1.0 / 5040.0), UDE_TAYLOR_C(1.0 / 720.0),
          UDE_TAYLOR_C(1.0 / 120.0), UDE_TAYLOR_C(1.0 / 24.0), UDE_TAYLOR_C(1.0 / 6.0));
    return p;
}

__device__ __forceinline__ double dexp(double x) {
    // ONE branch-free path: the two range tests are selects at the END (round 4: as early returns the compiler made each an
    // EXEC-masked branch -- s_and_saveexec / s_cbranch_execz / s_or around the arithmetic -- and a basic-block boundary in the middle
    // of every network layer; LV adj_kernel 1.27 -> 1.20 ms without them).  An out-of-range argument runs through the arithmetic
    // (finite garbage, Inf or NaN -- nothing traps) and is replaced; a NaN fails both tests and comes out as NaN.  Same bits as before
    // for every in-range argument: the same operations.
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-k, 0.6931471803691238, x);
    r = __builtin_fma(-k, 1.9082149292705877e-10, r);
    double p = taylor_13_to_3(r);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    double res = __builtin_ldexp(p, exponent_of(k));
    res = x > 709.0 ? __builtin_inf() : res;
    res = x < -745.0 ? 0.0 : res;
    return res;
}

// em / d for d = em + 2, 0 <= em < 2^58: the correctly rounded quotient, bit for bit what `em / d` returns, without the operand
// scaling and the special-case fix-up of the general division -- reciprocal seed, two Newton steps, quotient, one correction by
// the exact residual.  Nothing in it can overflow (2 <= d < 2^58, 0 <= q < 1); the residual fma(-d, q, em) is exact, and for
// em < 2^-52 the divisor IS 2, the reciprocal exactly 0.5 and q = em / 2 correctly rounded by the multiplication itself (also
// where that quotient is subnormal: the residual is then a multiple of the smallest subnormal and the correction rounds the
// exact em / 2 once).  Three instructions and two issue stalls less per tanh than v_div_scale x 2 / v_div_fmas / v_div_fixup.
__device__ __forceinline__ double div_em(double em, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = em * r;
    const double res = __builtin_fma(-d, q, em);
    return __builtin_fma(res, r, q);
}

__device__ __forceinline__ double dtanh(double x) {
    // ARITH-SPEC tanh: ONE branch-free path (lanes of a wavefront never diverge over the argument's size):
    //   z = 2|x| = k ln2 + r;  q = expm1(r) (dexp's Taylor polynomial without its final + 1);
    //   em = expm1(z) = 2^k q + (2^k - 1) (one fma, 2^k - 1 exact);  tanh = em / (em + 2)
    // accurate for small |x| (k = 0: em = q) and large alike; |x| >= 20 rounds to 1.
    // No test for a NaN argument (round 4; it was an EXEC-masked branch per call, 5 % of the Fisher-KPP network loop): both
    // selects are written `!(ax >= 20)` so that a NaN takes the arithmetic path and comes out as NaN (the oracle returns its
    // argument: equal under the NaN-aware comparison of the tests); the exponent goes through exponent_of(), not a cast.
    const double ax = fabs(x);
    const bool small = !(ax >= 20.0);
    // (round 6: z = 2|x| for EVERY argument.  The specification clamps z to 40 for |x| >= 20; those arguments' result is replaced by 1 at
    //  the end whatever the arithmetic produced -- finite garbage, Inf or NaN, nothing traps --, so the clamp's select (two v_cndmask_b32 of
    //  the ~44 instructions of a tanh, and every instruction is an issue slot: DESIGN.md 2b) changes no returned bit and is gone)
#ifdef UDE_TANH_CLAMP   // (the round 1-5 form, for A/B builds)
    const double z = small ? ax + ax : 40.0;
#else
    const double z = ax + ax;
#endif
    const double k = __builtin_rint(z * 1.4426950408889634);
    double r = __builtin_fma(-k, 0.6931471803691238, z);
    r = __builtin_fma(-k, 1.9082149292705877e-10, r);
    double p = taylor_13_to_3(r);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    const double q = p * r;
    const double s = __builtin_ldexp(1.0, exponent_of(k));
    const double em = __builtin_fma(s, q, s - 1.0);
    double t = div_em(em, em + 2.0);
    t = small ? t : 1.0;
    return x < 0 ? -t : t;
}

__device__ __forceinline__ double dlog10(double x) {
    int e;
    double m = __builtin_frexp(x, &e);
    if (m < 0.7071067811865476) { m = m + m; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 23.0;
#pragma unroll
    for (int n = 21; n >= 1; n -= 2) p = __builtin_fma(p, s2, 1.0 / (double)n);
    const double lnm = (s + s) * p;
    return __builtin_fma((double)e, 0.6931471805599453, lnm) * 0.4342944819032518;
}

__device__ __forceinline__ double dpow10(double y) { return dexp(y * 2.302585092994046); }

// natural log and x^y = exp(y*log(x)) for x > 0 (corona!'s (1 - D/N)^kappa, seir_exposure.jl:30)
__device__ __forceinline__ double dlog(double x) {
    int e;
    double m = __builtin_frexp(x, &e);
    if (m < 0.7071067811865476) { m = m + m; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 23.0;
#pragma unroll
    for (int n = 21; n >= 1; n -= 2) p = __builtin_fma(p, s2, 1.0 / (double)n);
    return __builtin_fma((double)e, 0.6931471805599453, (s + s) * p);
}
__device__ __forceinline__ double dpow(double x, double y) { return dexp(y * dlog(x)); }

// ---------------------------------------------------------------------------------------------
// activations (a1): rbf(x) = exp(-x^2) LotkaVolterra/scenario_1.jl:59; tanh seir_exposure.jl:114,
// Fisher-KPP-CNN.jl:92-94; relu highdim_pde/lambaem.jl
// ---------------------------------------------------------------------------------------------
enum { ACT_IDENTITY = 0, ACT_TANH = 1, ACT_RBF = 2, ACT_RELU = 3 };

// elementary functions on `real`: the double kernels above rounded once (Float64: the kernels themselves)
__device__ __forceinline__ real rexp(real x) { return (real)dexp((double)x); }
__device__ __forceinline__ real rtanh(real x) { return (real)dtanh((double)x); }
__device__ __forceinline__ real rlog10(real x) { return (real)dlog10((double)x); }
__device__ __forceinline__ real rpow10(real x) { return (real)dpow10((double)x); }
__device__ __forceinline__ real rpow(real x, real y) { return (real)dpow((double)x, (double)y); }

template <int ACT>
__device__ __forceinline__ real act_fwd(real z) {
    if constexpr (ACT == ACT_TANH) return rtanh(z);
    else if constexpr (ACT == ACT_RBF) return rexp(-(z * z));
    else if constexpr (ACT == ACT_RELU) return z > real(0) ? z : real(0);
    else return z;
}
// derivative from the pre-activation z and the activation value a
template <int ACT>
__device__ __forceinline__ real act_bwd(real z, real a) {
    if constexpr (ACT == ACT_TANH) return rfma(-a, a, real(1));
    else if constexpr (ACT == ACT_RBF) return (real(-2) * z) * a;
    else if constexpr (ACT == ACT_RELU) return z > real(0) ? real(1) : real(0);
    else return real(1);
}

// compile-time loop with integral-constant index
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

}  // namespace ude
