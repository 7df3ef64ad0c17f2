// ude_real.h -- the scalar type of a kernel translation unit.
//
// The reference solves in Float64 (scenario_1/2.jl, seir_exposure.jl, Fisher-KPP-CNN.jl) and in Float32
// (LotkaVolterra/scenario_3.jl:26-57,121-126; hudson_bay.jl:77-104).  Every ODE kernel is written on `real`; an
// instance translation unit is compiled with -DUDE_REAL=float for the Float32 problems (build.py).  Kernels carry
// `real` as a defaulted template argument so that the Float32 and Float64 builds of one (model, algorithm, lanes) are
// different symbols.  ARITH-SPEC holds for both types: explicit fma (rfma), IEEE sqrt / division, and the elementary
// functions = the fixed-order double kernels of ude_math.h rounded once to `real` (oracle: R_EXP, R_TANH, ...).
#pragma once
#include <hip/hip_runtime.h>

#ifdef UDE_F32   // build.py: Float32 instance translation units
#define UDE_REAL float
#endif
#ifndef UDE_REAL
#define UDE_REAL double
#endif

namespace ude {

using real = UDE_REAL;
static_assert(sizeof(real) == 8 || sizeof(real) == 4, "real is double or float");
constexpr bool REAL_IS_F64 = sizeof(real) == 8;
// machine epsilon of `real` (the dt <= eps |t| test of the driver, the initial-dt floor)
constexpr real REAL_EPS = REAL_IS_F64 ? (real)2.220446049250313e-16 : (real)1.1920929e-07;

__device__ __forceinline__ double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float rfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double rsqrt_ieee(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float rsqrt_ieee(float x) { return __builtin_sqrtf(x); }  // (-fhip-fp32-correctly-rounded-divide-sqrt)
__device__ __forceinline__ double rabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float rabs(float x) { return __builtin_fabsf(x); }

// ARITH-SPEC accumulator of the scaled error norm: the sum of squared residuals is formed in Float64 for BOTH scalar
// types.  Float64: the ordinary fma chain.  Float32: every res*res is exact in Float64 and a sum of a few hundred terms
// carries ~1e-14 relative error, so the value rounded back to Float32 does not depend on the summation order (lane tree
// here, sequential loop in the oracle) except within ~1e-7 of a rounding boundary -- the same margin the Float32
// fastpow quantisation gives the Float64 problems.  (Upstream sums in Float32 under @simd, i.e. in an order of LLVM's
// choosing: no order is canonical.)
using acc_t = double;
__device__ __forceinline__ acc_t afma(double a, double b, acc_t c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ acc_t afma(float a, float b, acc_t c) { return __builtin_fma((double)a, (double)b, c); }

// 64-/32-bit cross-lane moves of a `real`
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// (row-masked form: lanes of rows outside RM receive 0)
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_mov_rows(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, RM, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, RM, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double bpermute_real(int byte_addr, double x) {
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bpermute_real(int byte_addr, float x) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(x)));
}
__device__ __forceinline__ double uniform_real(double x) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float uniform_real(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ double readlane_real(double x, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_real(float x, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k)); }

// distance to the next representable number above |x| (the tstop snapping of the driver: 100 ulp)
__device__ __forceinline__ double ulp_of(double x) {
    x = __builtin_fabs(x);
    return __longlong_as_double(__double_as_longlong(x) + 1) - x;
}
__device__ __forceinline__ float ulp_of(float x) {
    x = __builtin_fabsf(x);
    return __int_as_float(__float_as_int(x) + 1) - x;
}

}  // namespace ude
