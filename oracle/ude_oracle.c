/*
 * ude_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See ude_oracle.h.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include "ude_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "ude_tableaux_gen.h"
#include <stdio.h>
int udeo_debug = 0; /* set to 1 (ctypes) to trace every step on stderr */

/* ------------------------------------------------------------------------------------------
 * DiffEqBase.fastpow: Float64(exp2(Float32(y) * fastlog2(Float32(x)))) with
 * fastlog2 = Goldberg's rational approximation including the "significand > 1.5" branch that the
 * golden DEStats require (147/19 instead of 148/20 on long_solution).
 * exp2 on Float32: Julia evaluates its Float32 exp2 kernel in Float64 and rounds once, i.e. the
 * result is (almost always) the correctly rounded Float32.  udeo_exp2f does the same with a
 * fixed-order fma() Horner polynomial so it is reproducible bit-for-bit on any IEEE machine.
 * ------------------------------------------------------------------------------------------ */
float udeo_fastlog2(float x) {
    const float a = 0.338953f, b = 2.198599f, c = 1.523692f;
    uint32_t ux;
    memcpy(&ux, &x, 4);
    const uint32_t ex = (ux & 0x7F800000u) >> 23;
    const uint32_t greater = ux & 0x00400000u;
    uint32_t um;
    float fexp, signif;
    if (greater) {
        um = (ux & 0x007FFFFFu) | 0x3f000000u;
        fexp = (float)ex - 126.0f;
    } else {
        um = (ux & 0x007FFFFFu) | 0x3f800000u;
        fexp = (float)ex - 127.0f;
    }
    memcpy(&signif, &um, 4);
    signif = signif - 1.0f;
    volatile float t1 = a * signif;
    volatile float t2 = t1 + b;
    volatile float t3 = signif * t2;
    volatile float t4 = signif + c;
    volatile float t5 = t3 / t4;
    return fexp + t5;
}

float udeo_exp2f(float x) {
    if (x != x) return x;
    if (x > 127.0f) return INFINITY;
    if (x < -126.0f) return 0.0f;
    const double xd = (double)x;
    const double n = rint(xd);
    const double z = (xd - n) * 0.6931471805599453; /* |z| <= 0.3466 */
    /* e^z, Taylor degree 13, Horner with explicit fma (deterministic, rel. error < 1e-16) */
    double p = 1.0 / 6227020800.0;
    p = fma(p, z, 1.0 / 479001600.0);
    p = fma(p, z, 1.0 / 39916800.0);
    p = fma(p, z, 1.0 / 3628800.0);
    p = fma(p, z, 1.0 / 362880.0);
    p = fma(p, z, 1.0 / 40320.0);
    p = fma(p, z, 1.0 / 5040.0);
    p = fma(p, z, 1.0 / 720.0);
    p = fma(p, z, 1.0 / 120.0);
    p = fma(p, z, 1.0 / 24.0);
    p = fma(p, z, 1.0 / 6.0);
    p = fma(p, z, 0.5);
    p = fma(p, z, 1.0);
    p = fma(p, z, 1.0);
    return (float)ldexp(p, (int)n);
}

double udeo_fastpow(double x, double y) {
    volatile float prod = (float)y * udeo_fastlog2((float)x);
    return (double)udeo_exp2f(prod);
}

int udeo_num_params(const udeo_model_desc* m) {
    int c = 0;
    for (int l = 0; l < m->n_layers; ++l) c += m->dims[l] * m->dims[l + 1] + m->dims[l + 1];
    return c;
}

static double ulp_f64(double x) { x = fabs(x); return nextafter(x, INFINITY) - x; }
static float ulp_f32(float x) { x = fabsf(x); return nextafterf(x, INFINITY) - x; }

/* ---- f64 instantiation ---- */
#define REAL double
#define FN(name) name##_f64
#define R_EXP exp
#define R_TANH tanh
#define R_SQRT sqrt
#define R_FABS fabs
#define R_LOG10 log10
#define R_POW pow
#define R_EPS 2.220446049250313e-16
#include "ude_oracle_impl.h"
#include "ude_oracle_adj.h"
#undef REAL
#undef FN
#undef R_EXP
#undef R_TANH
#undef R_SQRT
#undef R_FABS
#undef R_LOG10
#undef R_POW
#undef R_EPS

/* ---- f32 instantiation ---- */
#define REAL float
#define FN(name) name##_f32
#define R_EXP expf
#define R_TANH tanhf
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_LOG10 log10f
#define R_POW powf
#define R_EPS 1.1920929e-07f
#include "ude_oracle_impl.h"
#undef REAL
#undef FN
