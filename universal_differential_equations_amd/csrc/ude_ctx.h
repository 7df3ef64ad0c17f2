// ude_ctx.h -- the context object behind the C ABI (include/udecore.h), shared by the translation units that implement it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/udecore.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct ude_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    ude_launch_opts lo{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // fwd start/end, bwd start/end
    bool ev_fwd = false, ev_bwd = false;
    // workspaces (grow on demand, reused across calls)
    DevBuf dense, dense_n, cot, loss_traj, grad_part, retcode, stats, trace, tabs, slot_glob, nfail, tspan_pt, ls_fac, perm, sort_ws, rowsum, prev_cost;
    int64_t prev_cost_n = 0;     // cost-ordered launch: prev_cost holds the backward step attempts of the last such call (N members, signature below)
    int64_t prev_cost_sig = 0;
    hipEvent_t ev_sync = nullptr;  // orders work across a change of the bound stream
    int ncu = 0;         // compute units of the device (queried once)
    int auto_cap = 256;  // dense-store capacity used when lo.max_dense_steps == 0; grows x4 on DenseOverflow (host-buffer path)
    int64_t trace_traj = -1;
    int32_t trace_cap = 0;
    // staging for the host-buffer entry points
    DevBuf s_u0, s_theta, s_saveat, s_out, s_data, s_mask, s_gtheta, s_gu0, s_loss, s_lpt, s_stats, s_ret;
    // workspaces of the stochastic (deep-BSDE / LambaEM) path, see ude_hjb.hip
    DevBuf hj[32];
    float hj_fwd_ms = 0.f, hj_bwd_ms = 0.f;
    bool hj_cap_was_auto = false;  // the most recent ude_hjb_loss_grad_dev call recorded steps with the automatic capacity
    int hj_auto_cap = 512;  // accepted-step store per trajectory when ude_hjb_desc.max_steps == 0; grows x4 on StoreOverflow
    hipEvent_t hj_ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

static inline int fail(ude_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return fail(c, UDE_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static inline int ensure(ude_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return UDE_OK;
    if (b.p) {
        HIPCHK(c, hipDeviceSynchronize());  // (work queued on a previously bound stream may still use it)
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return fail(c, UDE_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    b.cap = want;
    return UDE_OK;
}

// debugging hook of the DEBUG build only (-DUDE_DEBUG_HOOKS, libudecore_dbg.so; UDE_EXP_POISON, udecore.hip): garbage into
// every register / LDS byte of the chip in front of a kernel.  In the shipping library this is an empty inline function.
#ifdef UDE_DEBUG_HOOKS
void ude_poison_chip_dbg(hipStream_t st, bool before_forward);
static inline void ude_poison_chip(hipStream_t st, bool before_forward) { ude_poison_chip_dbg(st, before_forward); }
#else
static inline void ude_poison_chip(hipStream_t, bool) {}
#endif
