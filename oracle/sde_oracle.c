/*
 * sde_oracle.c -- CPU ORACLE for SURVEY.md 8(f) N1 / BASELINE configs[4] (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See sde_oracle.h.  Part of libude_oracle.so (make -C oracle).
 */
#include "sde_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "ude_oracle.h" /* udeo_log, udeo_log10, udeo_pow10, udeo_fastpow: the ARITH-SPEC scalar kernels */

int udeo_hjb_num_params(int32_t d, int32_t H, int32_t* np_u0, int32_t* np_sg) {
    if (np_u0) *np_u0 = H * d + H + H * H + H + H + 1;                       /* d->H->H->1   (lambaem.jl:23-25) */
    if (np_sg) *np_sg = H * (d + 1) + H + H * H + H + H * H + H + d * H + d; /* d+1->H->H->H->d (lambaem.jl:27-30) */
    return 0;
}

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) */
void udeo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* sin(2 pi u), cos(2 pi u) for u in [0,1): quadrant q = floor(4u), x = (4u - q) pi/2 in [0, pi/2), Taylor polynomials of
 * sin and cos in x^2 evaluated as fixed-order fma chains (abs. error < 3e-16), then the quadrant rotation */
double udeo_sincos2pi(double u, double* cosout) {
    const double v = u * 4.0;
    const int q = (int)v;
    const double x = (v - (double)q) * 1.5707963267948966;
    const double x2 = x * x;
    double ps = -1.0 / 51090942171709440000.0; /* -1/21! */
    ps = fma(ps, x2, 1.0 / 121645100408832000.0);   /* 1/19! */
    ps = fma(ps, x2, -1.0 / 355687428096000.0);     /* -1/17! */
    ps = fma(ps, x2, 1.0 / 1307674368000.0);        /* 1/15! */
    ps = fma(ps, x2, -1.0 / 6227020800.0);          /* -1/13! */
    ps = fma(ps, x2, 1.0 / 39916800.0);             /* 1/11! */
    ps = fma(ps, x2, -1.0 / 362880.0);              /* -1/9! */
    ps = fma(ps, x2, 1.0 / 5040.0);                 /* 1/7! */
    ps = fma(ps, x2, -1.0 / 120.0);                 /* -1/5! */
    ps = fma(ps, x2, 1.0 / 6.0);                    /* 1/3! */
    const double s = fma(-(x * x2), ps, x);         /* x - x^3 (1/3! - x^2/5! + ...) */
    double pc = 1.0 / 2432902008176640000.0;        /* 1/20! */
    pc = fma(pc, x2, -1.0 / 6402373705728000.0);    /* -1/18! */
    pc = fma(pc, x2, 1.0 / 20922789888000.0);       /* 1/16! */
    pc = fma(pc, x2, -1.0 / 87178291200.0);         /* -1/14! */
    pc = fma(pc, x2, 1.0 / 479001600.0);            /* 1/12! */
    pc = fma(pc, x2, -1.0 / 3628800.0);             /* -1/10! */
    pc = fma(pc, x2, 1.0 / 40320.0);                /* 1/8! */
    pc = fma(pc, x2, -1.0 / 720.0);                 /* -1/6! */
    pc = fma(pc, x2, 1.0 / 24.0);                   /* 1/4! */
    pc = fma(pc, x2, -0.5);                         /* -1/2! */
    const double c = fma(pc, x2, 1.0);
    double so, co;
    switch (q & 3) {
        case 0: so = s; co = c; break;
        case 1: so = c; co = -s; break;
        case 2: so = -s; co = -c; break;
        default: so = -c; co = s; break;
    }
    *cosout = co;
    return so;
}

/* the d standard normals of draw event `event` of trajectory `traj` in training iteration `iter`:
 * counter = (chunk, event, traj, iter), chunk c -> components 4c..4c+3 by two Box-Muller pairs */
void udeo_hjb_normals(uint64_t seed, uint32_t iter, uint32_t traj, uint32_t event, int32_t d, double* out) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int32_t ch = 0; 4 * ch < d; ++ch) {
        const uint32_t ctr[4] = {(uint32_t)ch, event, traj, iter};
        uint32_t r[4];
        udeo_philox4x32_10(ctr, key, r);
        for (int h = 0; h < 2; ++h) {
            const double u1 = ((double)r[2 * h] + 0.5) * 2.3283064365386963e-10; /* (0,1) */
            const double u2 = (double)r[2 * h + 1] * 2.3283064365386963e-10;     /* [0,1) */
            const double rad = sqrt(-2.0 * udeo_log(u1));
            double co;
            const double si = udeo_sincos2pi(u2, &co);
            const int c0 = 4 * ch + 2 * h;
            if (c0 < d) out[c0] = rad * co;
            if (c0 + 1 < d) out[c0 + 1] = rad * si;
        }
    }
}

#define REAL float
#define NAME(x) x##_f32
#define FMA fmaf
#define SQRT sqrtf
#define FABS fabsf
#include "sde_oracle_impl.h"
#undef REAL
#undef NAME
#undef FMA
#undef SQRT
#undef FABS

#define REAL double
#define NAME(x) x##_f64
#define FMA fma
#define SQRT sqrt
#define FABS fabs
#include "sde_oracle_impl.h"
#undef REAL
#undef NAME
#undef FMA
#undef SQRT
#undef FABS

int udeo_hjb_loss_grad_f32(const udeo_hjb_desc* D, int64_t M, const float* x0, const float* theta, uint32_t iter,
                           double* loss, float* grad, float* u0_out, float* uT, float* XT, double* loss_traj,
                           int64_t* stats, int32_t* retcode, int32_t nthreads) {
    return loss_grad_f32(D, M, x0, theta, iter, loss, grad, u0_out, uT, XT, loss_traj, stats, retcode, nthreads);
}
int udeo_hjb_loss_grad_f64(const udeo_hjb_desc* D, int64_t M, const double* x0, const double* theta, uint32_t iter,
                           double* loss, double* grad, double* u0_out, double* uT, double* XT, double* loss_traj,
                           int64_t* stats, int32_t* retcode, int32_t nthreads) {
    return loss_grad_f64(D, M, x0, theta, iter, loss, grad, u0_out, uT, XT, loss_traj, stats, retcode, nthreads);
}

void udeo_hjb_net_f32(int32_t d, int32_t H, const float* theta_sg, const float* x_in, float* z) {
    Nets_f32 n;
    /* nets_init expects the full theta: rebuild the pointers of the second chain only */
    memset(&n, 0, sizeof n);
    n.d = d; n.H = H;
    const float* p = theta_sg;
    n.W1 = p; p += (size_t)H * (d + 1); n.b1 = p; p += H;
    n.W2 = p; p += (size_t)H * H; n.b2 = p; p += H;
    n.W3 = p; p += (size_t)H * H; n.b3 = p; p += H;
    n.W4 = p; p += (size_t)d * H; n.b4 = p;
    float a1[128], a2[128], a3[128];
    sg_fwd_f32(&n, x_in, a1, a2, a3, z);
}

int udeo_hjb_path_f32(const udeo_hjb_desc* D, const float* x0, const float* theta, uint32_t iter, uint32_t traj, int32_t cap,
                      float* t_out, float* dt_out, float* X_out, float* dW_out, float* EEst_out) {
    Nets_f32 n;
    nets_init_f32(&n, D->d, D->hls, theta);
    Par_f32 p;
    par_init_f32(&p, D);
    p.cap = cap;
    const float u0 = u0_net_f32(&n, x0, 0.0f, NULL);
    float dt_init = (float)D->dt;
    if (D->adaptive && !(D->dt > 0)) dt_init = initdt_f32(&n, &p, x0, u0);
    TrajStat_f32 st;
    float Xf[128], uf;
    traj_f32(D, &n, &p, x0, u0, dt_init, iter, traj, Xf, &uf, t_out, dt_out, X_out, dW_out, EEst_out, &st);
    return st.ret == UDEO_HJB_RET_SUCCESS ? st.nacc : -st.ret;
}
