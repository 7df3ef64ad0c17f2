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

/* ------------------------------------------------------------------------------------------
 * ARITH-SPEC elementary functions.  The step sequence of an adaptive solve is chaotic in the last
 * bit of every intermediate (sliver steps before a tstop have a pure-rounding-noise error estimate;
 * see DESIGN.md "Arithmetic specification"), so "bit-exact step counts" between this oracle and the
 * HIP kernels is only meaningful if both evaluate exp/tanh/log10/pow10 with the SAME sequence of
 * IEEE operations.  These are fixed-order fma() kernels (about 1 ulp), restated independently in
 * universal_differential_equations_amd/csrc/ude_math.h.
 * ------------------------------------------------------------------------------------------ */
double udeo_exp(double x) {
    if (x != x) return x;
    if (x > 709.0) return INFINITY;
    if (x < -745.0) return 0.0;
    const double k = rint(x * 1.4426950408889634);
    double r = fma(-k, 0.6931471803691238, x);       /* ln2 hi (32 bits) */
    r = fma(-k, 1.9082149292705877e-10, r);           /* ln2 lo */
    double p = 1.0 / 6227020800.0;                    /* Taylor 13, |r| <= 0.3466 */
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

double udeo_tanh(double x) {
    /* ARITH-SPEC tanh, one path: z = 2|x| = k ln2 + r; q = expm1(r) (udeo_exp's polynomial without its final + 1);
     * em = expm1(z) = 2^k q + (2^k - 1); tanh = em / (em + 2); |x| >= 20 rounds to 1 */
    if (x != x) return x;
    const double ax = fabs(x);
    const double z = ax < 20.0 ? ax + ax : 40.0;
    const double k = rint(z * 1.4426950408889634);
    double r = fma(-k, 0.6931471803691238, z);
    r = fma(-k, 1.9082149292705877e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    const double q = p * r;
    const double s = ldexp(1.0, (int)k);
    const double em = fma(s, q, s - 1.0);
    double t = em / (em + 2.0);
    t = ax < 20.0 ? t : 1.0;
    return x < 0 ? -t : t;
}

double udeo_log10(double x) { /* x > 0, finite (initial-dt heuristic only) */
    int e;
    double m = frexp(x, &e); /* m in [0.5, 1) */
    if (m < 0.7071067811865476) { m = m + m; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 23.0;
    for (int n = 21; n >= 1; n -= 2) p = fma(p, s2, 1.0 / (double)n);
    const double lnm = (s + s) * p;
    return fma((double)e, 0.6931471805599453, lnm) * 0.4342944819032518;
}

double udeo_pow10(double y) { return udeo_exp(y * 2.302585092994046); }

double udeo_log(double x) { /* x > 0, finite */
    int e;
    double m = frexp(x, &e);
    if (m < 0.7071067811865476) { m = m + m; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 23.0;
    for (int n = 21; n >= 1; n -= 2) p = fma(p, s2, 1.0 / (double)n);
    return fma((double)e, 0.6931471805599453, (s + s) * p);
}

double udeo_pow(double x, double y) { return udeo_exp(y * udeo_log(x)); } /* x > 0 */

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
#define R_EXP udeo_exp
#define R_TANH udeo_tanh
#define R_SQRT sqrt
#define R_FABS fabs
#define R_LOG10 udeo_log10
#define R_POW udeo_pow
#define R_POW10(x) udeo_pow10(x)
#define R_FMA fma
#define R_EPS 2.220446049250313e-16
#include "ude_oracle_impl.h"
#define UDEO_ADJ_F64_ONLY 1
#include "ude_oracle_adj.h"
#undef UDEO_ADJ_F64_ONLY
#undef REAL
#undef FN
#undef R_EXP
#undef R_TANH
#undef R_SQRT
#undef R_FABS
#undef R_LOG10
#undef R_POW
#undef R_POW10
#undef R_FMA
#undef R_EPS

/* ---- f32 instantiation ---- */
#define REAL float
#define FN(name) name##_f32
/* Float32 elementary functions = the ARITH-SPEC double kernels rounded once to Float32 (deterministic on every machine,
 * and what the Float32 HIP instances evaluate); within 1 ulp of Julia's Float32 exp / tanh / log10 / ^ */
#define R_EXP(x) ((float)udeo_exp((double)(x)))
#define R_TANH(x) ((float)udeo_tanh((double)(x)))
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_LOG10(x) ((float)udeo_log10((double)(x)))
#define R_POW(x, y) ((float)udeo_pow((double)(x), (double)(y)))
#define R_POW10(x) ((float)udeo_pow10((double)(x)))
#define R_FMA fmaf
#define R_EPS 1.1920929e-07f
#include "ude_oracle_impl.h"
#include "ude_oracle_adj.h"
#undef REAL
#undef FN
