// ude_hjb.hip -- C ABI of the stochastic (deep-BSDE / LambaEM) path: include/udecore.h `ude_hjb_*`
// (BASELINE configs[4], highdim_pde/lambaem.jl:8-48).  gfx950 only.
#include <cstdlib>
#include <cstdio>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "ude_ctx.h"
#include "ude_sde.h"

using namespace ude::hjb;

namespace {
enum { B_PREP = 0, B_XIN, B_A1, B_A2, B_A3, B_E4, B_NACC, B_STACKW, B_LT, B_UBAR, B_RET, B_PART, B_LOSS, B_QUEUE,
       S_X0, S_TH, S_GRAD, S_U0, S_UT, S_XT, S_LT, S_STATS, S_RET, S_NIN, S_NOUT };
constexpr int KD = 100, KH = 110;  // the compiled instance (lambaem.jl:8,20)

__global__ void normals_kernel(uint64_t seed, uint32_t iter, uint32_t traj, uint32_t ev, int d, double* out) {
    const int l = threadIdx.x;
    double n0, n1;
    normal_pair(seed, iter, traj, ev, l, n0, n1);
    if (2 * l < d) out[2 * l] = n0;
    if (2 * l + 1 < d) out[2 * l + 1] = n1;
}

int up(ude_ctx* c, DevBuf& b, const void* src, size_t bytes, void** dst) {
    *dst = nullptr;
    if (!src || bytes == 0) return UDE_OK;
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    *dst = b.p;
    return UDE_OK;
}
int dn(ude_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!dst || !src || bytes == 0) return UDE_OK;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return UDE_OK;
}
}  // namespace

// a trajectory that outgrows the accepted-step store keeps stepping unrecorded and reports its true number of accepted steps:
// the automatic capacity becomes the largest count of the call that just ended (+ 1/8), so ONE re-run suffices
static int grow_store(ude_ctx* c, int64_t M) {
    std::vector<int32_t> n(M);
    HIPCHK(c, hipMemcpy(n.data(), c->hj[B_NACC].p, sizeof(int32_t) * M, hipMemcpyDeviceToHost));
    int32_t mx = 0;
    for (int64_t j = 0; j < M; ++j) mx = n[j] > mx ? n[j] : mx;
    const int64_t want = (int64_t)mx + mx / 8 + 32;
    c->hj_auto_cap = (int)(want > 2 * (int64_t)c->hj_auto_cap ? want : 2 * (int64_t)c->hj_auto_cap);
    return UDE_OK;
}

extern "C" int ude_hjb_num_params(int32_t d, int32_t H, int32_t* np_u0, int32_t* np_sg) {
    if (np_u0) *np_u0 = H * d + H + H * H + H + H + 1;
    if (np_sg) *np_sg = H * (d + 1) + H + H * H + H + H * H + H + d * H + d;
    return UDE_OK;
}

extern "C" int ude_hjb_loss_grad_dev(ude_ctx* c, const ude_hjb_desc* D, int64_t M, const float* x0, const float* theta, uint32_t iter,
                                     double* loss, float* grad, float* u0_out, float* uT, float* XT, double* loss_traj,
                                     int64_t* stats, int32_t* retcode) {
    if (!c) return UDE_ERR_INVALID;
    if (!D || !x0 || !theta || !loss || M <= 0) return fail(c, UDE_ERR_INVALID, "null argument or empty ensemble");
    if (D->d != KD || D->hls != KH)
        return fail(c, UDE_ERR_UNSUPPORTED, "no compiled kernel for d = %d, hls = %d (instance: d = %d, hls = %d)", D->d, D->hls, KD, KH);
    if (!(D->t1 > D->t0)) return fail(c, UDE_ERR_INVALID, "tspan must be increasing");
    if (!D->adaptive && !(D->dt > 0)) return fail(c, UDE_ERR_INVALID, "fixed-step Euler-Maruyama needs dt > 0");
    if (D->adaptive && !(D->abstol > 0 && D->reltol >= 0)) return fail(c, UDE_ERR_INVALID, "adaptive stepping needs abstol > 0");
    HIPCHK(c, hipSetDevice(c->device));
    using C = Cfg<KD, KH>;
    HjbParams p;
    memset(&p, 0, sizeof p);
    p.M = M;
    // capacity of the accepted-step store: the caller's, or the context's automatic one (grown x4 by the host-buffer entry point /
    // ude_hjb_last_failures when a trajectory outgrows it); a loss-only call records nothing and has no limit
    p.cap = !grad ? 0x7fffffff : D->max_steps > 0 ? D->max_steps : c->hj_auto_cap;
    c->hj_cap_was_auto = grad && D->max_steps <= 0;
    p.maxiters = D->maxiters > 0 ? D->maxiters : 1000000;
    p.adaptive = D->adaptive ? 1 : 0;
    p.record = grad ? 1 : 0;
    p.iter = iter;
    p.seed = D->seed;
    p.lam = (float)D->lambda; p.sig = (float)D->sigma; p.t0 = (float)D->t0; p.t1 = (float)D->t1;
    p.abstol = (float)D->abstol; p.reltol = (float)D->reltol;
    p.qmin = (float)(D->qmin > 0 ? D->qmin : 0.2);
    p.qmax = (float)(D->qmax > 0 ? D->qmax : 1.125);
    p.gamma = (float)(D->gamma > 0 ? D->gamma : 0.9);
    p.qoldinit = (float)(D->qoldinit > 0 ? D->qoldinit : 1e-4);
    p.beta1 = (float)(D->beta1 > 0 ? D->beta1 : 0.7);
    p.beta2 = (float)(D->beta2 > 0 ? D->beta2 : 0.4);
    p.dtmax = (float)(D->dtmax > 0 ? D->dtmax : D->t1 - D->t0);
    p.dt_user = (float)D->dt;
    p.x0 = x0;
    p.theta = theta;
    int rc;
    const size_t ncol = grad ? (size_t)M * p.cap : 0;
    if (grad) {  // the records must fit the device: refuse instead of driving the box out of memory
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        const size_t need = ncol * sizeof(float) * (size_t)(C::RX + 3 * C::RA + C::RE);
        const size_t have = c->hj[B_XIN].cap + c->hj[B_A1].cap + c->hj[B_A2].cap + c->hj[B_A3].cap + c->hj[B_E4].cap;
        if (need > have && need - have > fr / 10 * 9)
            return fail(c, UDE_ERR_NOMEM, "the accepted-step store for %lld trajectories x %d steps needs %zu MB (free: %zu MB)", (long long)M, p.cap,
                        need >> 20, fr >> 20);
    }
    if ((rc = ensure(c, c->hj[B_PREP], sizeof(float) * (2 + 2 * KH)))) return rc;
    if (grad) {
        if ((rc = ensure(c, c->hj[B_XIN], sizeof(float) * ncol * C::RX))) return rc;
        if ((rc = ensure(c, c->hj[B_A1], sizeof(float) * ncol * C::RA))) return rc;
        if ((rc = ensure(c, c->hj[B_A2], sizeof(float) * ncol * C::RA))) return rc;
        if ((rc = ensure(c, c->hj[B_A3], sizeof(float) * ncol * C::RA))) return rc;
        if ((rc = ensure(c, c->hj[B_E4], sizeof(float) * ncol * C::RE))) return rc;
    }
    if ((rc = ensure(c, c->hj[B_NACC], sizeof(int32_t) * M))) return rc;
    if ((rc = ensure(c, c->hj[B_STACKW], sizeof(float) * (size_t)M * STACK * XLD))) return rc;
    if ((rc = ensure(c, c->hj[B_LT], sizeof(double) * M))) return rc;
    if ((rc = ensure(c, c->hj[B_UBAR], sizeof(float) * M))) return rc;
    if ((rc = ensure(c, c->hj[B_RET], sizeof(int32_t) * M))) return rc;
    if ((rc = ensure(c, c->nfail, sizeof(int32_t)))) return rc;
    if ((rc = ensure(c, c->hj[B_QUEUE], sizeof(int32_t)))) return rc;
    const int nblk = (int)(M < 256 ? M : 256);
    if (grad && (rc = ensure(c, c->hj[B_PART], sizeof(float) * (size_t)nblk * C::NP))) return rc;
    p.prep = (float*)c->hj[B_PREP].p;
    p.rXin = (float*)c->hj[B_XIN].p; p.rA1 = (float*)c->hj[B_A1].p; p.rA2 = (float*)c->hj[B_A2].p;
    p.rA3 = (float*)c->hj[B_A3].p; p.rE4 = (float*)c->hj[B_E4].p;
    p.nacc = (int32_t*)c->hj[B_NACC].p;
    p.stackW = (float*)c->hj[B_STACKW].p;
    p.uT = uT; p.XT = XT;
    p.loss_traj = loss_traj ? loss_traj : (double*)c->hj[B_LT].p;
    p.ubar = (float*)c->hj[B_UBAR].p;
    p.stats = stats;
    p.retcode = retcode ? retcode : (int32_t*)c->hj[B_RET].p;
    p.part = (float*)c->hj[B_PART].p;
    p.grad = grad;
    p.loss = loss;
    p.nfail = (int32_t*)c->nfail.p;
    for (auto& e : c->hj_ev)
        if (!e) HIPCHK(c, hipEventCreate(&e));

    const size_t sh_f = sizeof(float) * fwd_lds_floats<KD, KH>() + 16;
    const size_t sh_b = sizeof(float) * bwd_lds_floats<KD, KH>() + 16;
    HIPCHK(c, hipFuncSetAttribute((const void*)hjb_fwd_kernel<KD, KH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_f));
    HIPCHK(c, hipFuncSetAttribute((const void*)hjb_fwd_kernel<KD, KH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_f));
    hipLaunchKernelGGL((hjb_prep_kernel<KD, KH>), dim3(1), dim3(128), 0, c->stream, p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->hj_ev[0], c->stream));
    // one block of 32 trajectory slots per CU at most; the remaining trajectories are handed out through the queue
    int ncu = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    }
    int64_t nblk_f = (M + NT - 1) / NT;
    if (nblk_f > ncu) nblk_f = ncu;
    const int32_t qstart = (int32_t)(nblk_f * NT);
    p.queue = (int32_t*)c->hj[B_QUEUE].p;
    HIPCHK(c, hipMemcpyAsync(p.queue, &qstart, sizeof qstart, hipMemcpyHostToDevice, c->stream));
#ifdef UDE_DEBUG_HOOKS
    static const bool prof = getenv("UDE_HJB_PROF") != nullptr;  // debug build: phase clocks of block 0 on stderr (blocks the stream)
#else
    constexpr bool prof = false;
#endif
    if (prof) {
        if ((rc = ensure(c, c->hj[30], sizeof(unsigned long long) * 16))) return rc;
        p.prof = (unsigned long long*)c->hj[30].p;
        HIPCHK(c, hipMemsetAsync(p.prof, 0, sizeof(unsigned long long) * 16, c->stream));
    }
    ude_poison_chip(c->stream, true);
    if (p.adaptive) hipLaunchKernelGGL((hjb_fwd_kernel<KD, KH, true>), dim3((unsigned)nblk_f), dim3(256), sh_f, c->stream, p);
    else hipLaunchKernelGGL((hjb_fwd_kernel<KD, KH, false>), dim3((unsigned)nblk_f), dim3(256), sh_f, c->stream, p);
    HIPCHK(c, hipGetLastError());
    if (prof) {
        unsigned long long h[16];
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(h, p.prof, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "[hjb prof] block 0: %llu iterations; clock ticks: eval1+2 %llu | S1 %llu | eval3 %llu | S2: estimate+controller %llu, accept/reject %llu, finish/write-back %llu | end barrier %llu\n", h[8], h[0], h[1],
                h[2], h[5], h[6], h[3], h[4]);
    }
    HIPCHK(c, hipEventRecord(c->hj_ev[1], c->stream));
    HIPCHK(c, hipEventRecord(c->hj_ev[2], c->stream));
    if (grad) {
        HIPCHK(c, hipFuncSetAttribute((const void*)hjb_bwd_kernel<KD, KH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_b));
        ude_poison_chip(c->stream, false);
        hipLaunchKernelGGL((hjb_bwd_kernel<KD, KH>), dim3(nblk), dim3(256), sh_b, c->stream, p);
        HIPCHK(c, hipGetLastError());
        if (prof) {
            unsigned long long h[16];
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipMemcpy(h, p.prof, sizeof h, hipMemcpyDeviceToHost));
            fprintf(stderr, "[hjb prof] backward, block 0: %llu tiles of 32 columns; clock ticks: entry barrier %llu | loads+LDS stores %llu | exit barrier %llu | transposed layers %llu | outer products %llu\n",
                    h[13], h[9], h[15], h[10], h[11], h[12]);
        }
    }
    HIPCHK(c, hipEventRecord(c->hj_ev[3], c->stream));
    hipLaunchKernelGGL((hjb_reduce_kernel<KD, KH>), dim3((C::NP + 255) / 256 + 1), dim3(256), 0, c->stream, p, nblk);
    HIPCHK(c, hipGetLastError());
    if (u0_out) HIPCHK(c, hipMemcpyAsync(u0_out, p.prep, sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return UDE_OK;
}

extern "C" int ude_hjb_last_kernel_ms(ude_ctx* c, float* fwd_ms, float* bwd_ms) {
    if (!c) return UDE_ERR_INVALID;
    if (fwd_ms) *fwd_ms = 0.f;
    if (bwd_ms) *bwd_ms = 0.f;
    if (!c->hj_ev[0]) return UDE_OK;
    if (fwd_ms) HIPCHK(c, hipEventElapsedTime(fwd_ms, c->hj_ev[0], c->hj_ev[1]));
    if (bwd_ms) HIPCHK(c, hipEventElapsedTime(bwd_ms, c->hj_ev[2], c->hj_ev[3]));
    return UDE_OK;
}

extern "C" int ude_hjb_loss_grad(ude_ctx* c, const ude_hjb_desc* D, int64_t M, const float* x0, const float* theta, uint32_t iter,
                                 double* loss, float* grad, float* u0_out, float* uT, float* XT, double* loss_traj,
                                 int64_t* stats, int32_t* retcode) {
    if (!c) return UDE_ERR_INVALID;
    if (!D || !x0 || !theta || !loss || M <= 0) return fail(c, UDE_ERR_INVALID, "null argument or empty ensemble");
    HIPCHK(c, hipSetDevice(c->device));
    int32_t np0, np1;
    ude_hjb_num_params(D->d, D->hls, &np0, &np1);
    const size_t np = (size_t)np0 + np1, d = (size_t)D->d;
    int rc;
    void *dx0, *dth;
    if ((rc = up(c, c->hj[S_X0], x0, sizeof(float) * d, &dx0))) return rc;
    if ((rc = up(c, c->hj[S_TH], theta, sizeof(float) * np, &dth))) return rc;
    if ((rc = ensure(c, c->hj[B_LOSS], sizeof(double)))) return rc;
    if (grad && (rc = ensure(c, c->hj[S_GRAD], sizeof(float) * np))) return rc;
    if ((rc = ensure(c, c->hj[S_U0], sizeof(float)))) return rc;
    if ((rc = ensure(c, c->hj[S_UT], sizeof(float) * M))) return rc;
    if ((rc = ensure(c, c->hj[S_XT], sizeof(float) * M * d))) return rc;
    if ((rc = ensure(c, c->hj[S_LT], sizeof(double) * M))) return rc;
    if ((rc = ensure(c, c->hj[S_STATS], sizeof(int64_t) * 4 * M))) return rc;
    if ((rc = ensure(c, c->hj[S_RET], sizeof(int32_t) * M))) return rc;
    std::vector<int32_t> rtmp(M);
    for (;;) {
        rc = ude_hjb_loss_grad_dev(c, D, M, (const float*)dx0, (const float*)dth, iter, (double*)c->hj[B_LOSS].p,
                                   grad ? (float*)c->hj[S_GRAD].p : nullptr, (float*)c->hj[S_U0].p, (float*)c->hj[S_UT].p,
                                   (float*)c->hj[S_XT].p, (double*)c->hj[S_LT].p, (int64_t*)c->hj[S_STATS].p, (int32_t*)c->hj[S_RET].p);
        if (rc) return rc;
        if ((rc = dn(c, rtmp.data(), c->hj[S_RET].p, sizeof(int32_t) * M))) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // a trajectory that outgrew the accepted-step store: re-run with four times the capacity (upstream's solution
        // arrays simply grow) unless the caller pinned max_steps
        bool overflow = false;
        for (int64_t j = 0; j < M; ++j) overflow = overflow || rtmp[j] == RET_STORE_OVERFLOW;
        if (!overflow || !grad || D->max_steps > 0 || c->hj_auto_cap >= (1 << 24)) break;
        if ((rc = grow_store(c, M))) return rc;
    }
    if ((rc = dn(c, loss, c->hj[B_LOSS].p, sizeof(double)))) return rc;
    if ((rc = dn(c, grad, c->hj[S_GRAD].p, sizeof(float) * np))) return rc;
    if ((rc = dn(c, u0_out, c->hj[S_U0].p, sizeof(float)))) return rc;
    if ((rc = dn(c, uT, c->hj[S_UT].p, sizeof(float) * M))) return rc;
    if ((rc = dn(c, XT, c->hj[S_XT].p, sizeof(float) * M * d))) return rc;
    if ((rc = dn(c, loss_traj, c->hj[S_LT].p, sizeof(double) * M))) return rc;
    if ((rc = dn(c, stats, c->hj[S_STATS].p, sizeof(int64_t) * 4 * M))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (retcode) memcpy(retcode, rtmp.data(), sizeof(int32_t) * M);
    for (int64_t j = 0; j < M; ++j)
        if (rtmp[j] != 0)
            return fail(c, UDE_ERR_TRAJECTORY, "trajectory %lld ended with retcode %d (see retcode array)", (long long)j, rtmp[j]);
    return UDE_OK;
}

extern "C" int ude_hjb_normals(ude_ctx* c, uint64_t seed, uint32_t iter, uint32_t traj, uint32_t event, int32_t d, double* out_host) {
    if (!c || !out_host || d <= 0 || d > 128) return UDE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->hj[S_NOUT], sizeof(double) * 128))) return rc;
    hipLaunchKernelGGL(normals_kernel, dim3(1), dim3(64), 0, c->stream, seed, iter, traj, event, (int)d, (double*)c->hj[S_NOUT].p);
    HIPCHK(c, hipGetLastError());
    if ((rc = dn(c, out_host, c->hj[S_NOUT].p, sizeof(double) * d))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return UDE_OK;
}

extern "C" int ude_hjb_net(ude_ctx* c, int32_t d, int32_t hls, const float* theta_sg_host, int64_t n, const float* x_in_host, float* z_host) {
    if (!c) return UDE_ERR_INVALID;
    if (!theta_sg_host || !x_in_host || !z_host || n <= 0) return fail(c, UDE_ERR_INVALID, "null / empty argument");
    if (d != KD || hls != KH) return fail(c, UDE_ERR_UNSUPPORTED, "no compiled kernel for d = %d, hls = %d", d, hls);
    HIPCHK(c, hipSetDevice(c->device));
    using C = Cfg<KD, KH>;
    int rc;
    void *dth, *dx;
    if ((rc = up(c, c->hj[S_TH], theta_sg_host, sizeof(float) * C::NP, &dth))) return rc;
    if ((rc = up(c, c->hj[S_NIN], x_in_host, sizeof(float) * (size_t)n * C::DIN, &dx))) return rc;
    if ((rc = ensure(c, c->hj[S_NOUT], sizeof(float) * (size_t)n * KD))) return rc;
    const size_t sh = sizeof(float) * (2 * 128 * LDA + 4 * 128) + 16;
    HIPCHK(c, hipFuncSetAttribute((const void*)hjb_net_kernel<KD, KH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    const unsigned grid = (unsigned)((n + 31) / 32 < 1024 ? (n + 31) / 32 : 1024);
    hipLaunchKernelGGL((hjb_net_kernel<KD, KH>), dim3(grid), dim3(256), sh, c->stream, (const float*)dth, n, (const float*)dx,
                       (float*)c->hj[S_NOUT].p);
    HIPCHK(c, hipGetLastError());
    if ((rc = dn(c, z_host, c->hj[S_NOUT].p, sizeof(float) * (size_t)n * KD))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return UDE_OK;
}

// debugging aid for parity work: raw copy of a workspace of the most recent call (which: 0 prep, 1 Xin records, 5 E4 records)
extern "C" int ude_hjb_debug_read(ude_ctx* c, int32_t which, int64_t offset_floats, int64_t n_floats, float* out_host) {
    if (!c || !out_host || which < 0 || which > B_E4 || !c->hj[which].p) return UDE_ERR_INVALID;
    if (offset_floats < 0 || n_floats < 0 || (size_t)(offset_floats + n_floats) * sizeof(float) > c->hj[which].cap)
        return fail(c, UDE_ERR_INVALID, "debug read [%lld, %lld) floats is outside workspace %d (%zu bytes)", (long long)offset_floats,
                    (long long)(offset_floats + n_floats), which, c->hj[which].cap);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out_host, (const float*)c->hj[which].p + offset_floats, sizeof(float) * n_floats, hipMemcpyDeviceToHost));
    return UDE_OK;
}

// Failure accounting of the most recent ude_hjb_loss_grad_dev call (blocks on the context's stream): number of trajectories whose
// retcode is not Success; if one of them outgrew the automatic accepted-step store, the capacity is multiplied by 4 and *grown = 1
// (the caller repeats the call) -- the asynchronous counterpart of what ude_hjb_loss_grad does by itself.
extern "C" int ude_hjb_last_failures(ude_ctx* c, const int32_t* retcode_dev, int64_t M, int32_t* nfail, int32_t* grown) {
    if (!c || !nfail) return UDE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *nfail = 0;
    if (grown) *grown = 0;
    const int32_t* src = retcode_dev ? retcode_dev : (const int32_t*)c->hj[B_RET].p;
    if (!src || M <= 0) return UDE_OK;
    std::vector<int32_t> r(M);
    HIPCHK(c, hipMemcpy(r.data(), src, sizeof(int32_t) * M, hipMemcpyDeviceToHost));
    bool overflow = false;
    for (int64_t j = 0; j < M; ++j) {
        *nfail += r[j] != 0;
        overflow = overflow || r[j] == RET_STORE_OVERFLOW;
    }
    if (overflow && c->hj_cap_was_auto && c->hj_auto_cap < (1 << 24)) {
        int rc = grow_store(c, M);
        if (rc) return rc;
        if (grown) *grown = 1;
    }
    return UDE_OK;
}
