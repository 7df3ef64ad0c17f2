// ude_comm.hip -- the one exchange step of the data-parallel gradient behind the C ABI:
// all-reduce(sum) of double[np + 4] = gradient (+) loss (+) (nf, naccept, nreject) across the GPUs of a node.
//   * RCCL over xGMI: ncclAllReduce on the context's stream (one process per GPU: ude_comm_create with a shared unique
//     id; or all devices in one process: ude_comm_create_local = ncclCommInitAll).  RCCL is bound with dlopen so that
//     libudecore.so loads on a box without it and shares whichever librccl the process already mapped (PyTorch's).
//   * one-shot P2P reducer (single process, peer access over xGMI): every device reads the buffers of all ranks and adds
//     them in RANK ORDER -- a deterministic fp64 sum, identical bits on every device, two kernel launches of latency
//     instead of a 2(N-1)-step ring for a 704 B .. 36 KB payload.
// The reference is single-process CPU Julia: nothing is replaced, this is the multi-GPU row of the scope table.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ude_ctx.h"

namespace {
// the subset of rccl.h this file needs (ABI-stable since NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclFloat64 = 8, ncclSum = 0 };
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (r.h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.h, "ncclCommInitAll");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.h, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.h, "ncclGroupEnd");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.AllReduce || !r.CommDestroy) r.h = nullptr;
        }
    }
    return r.h ? &r : nullptr;
}

// out[i] = in[0][i] + in[1][i] + ... in rank order (left to right): identical bits on every device
__global__ void p2p_sum_kernel(const double* const* in, int nranks, int64_t n, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = in[0][i];
    for (int r = 1; r < nranks; ++r) s += in[r][i];
    out[i] = s;
}
}  // namespace

struct ude_comm {
    ude_ctx* ctx = nullptr;
    ncclComm_t nccl = nullptr;
    int nranks = 1, rank = 0;
};

#define NCCLCHK(c, call)                                                                                            \
    do {                                                                                                            \
        ncclResult_t r_ = (call);                                                                                   \
        if (r_ != ncclSuccess)                                                                                      \
            return fail(c, UDE_ERR_HIP, "%s failed: %s", #call, rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "rccl error"); \
    } while (0)

extern "C" int ude_comm_unique_id(char id[128]) {
    Rccl* R = rccl();
    if (!R || !id) return UDE_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return UDE_ERR_HIP;
    memcpy(id, u.internal, 128);
    return UDE_OK;
}

extern "C" int ude_comm_create(ude_ctx* c, int32_t nranks, int32_t rank, const char id[128], ude_comm** out) {
    if (!c || !out || !id || nranks < 1 || rank < 0 || rank >= nranks) return UDE_ERR_INVALID;
    *out = nullptr;
    Rccl* R = rccl();
    if (!R) return fail(c, UDE_ERR_UNSUPPORTED, "librccl could not be loaded");
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    ude_comm* m = new ude_comm();
    m->ctx = c; m->nranks = nranks; m->rank = rank;
    ncclResult_t r = R->CommInitRank(&m->nccl, nranks, u, rank);
    if (r != ncclSuccess) {
        delete m;
        return fail(c, UDE_ERR_HIP, "ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(r) : "rccl error");
    }
    *out = m;
    return UDE_OK;
}

extern "C" int ude_comm_create_local(int32_t ndev, ude_ctx* const* ctxs, ude_comm** out) {
    if (!ctxs || !out || ndev < 1) return UDE_ERR_INVALID;
    Rccl* R = rccl();
    if (!R) return fail(ctxs[0], UDE_ERR_UNSUPPORTED, "librccl could not be loaded");
    std::vector<int> devs(ndev);
    std::vector<ncclComm_t> comms(ndev);
    for (int i = 0; i < ndev; ++i) devs[i] = ctxs[i]->device;
    NCCLCHK(ctxs[0], R->CommInitAll(comms.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; ++i) {
        ude_comm* m = new ude_comm();
        m->ctx = ctxs[i]; m->nccl = comms[i]; m->nranks = ndev; m->rank = i;
        out[i] = m;
        // peer access for the P2P reducer (best effort: already-enabled is fine)
        (void)hipSetDevice(ctxs[i]->device);
        for (int j = 0; j < ndev; ++j)
            if (j != i) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, ctxs[i]->device, ctxs[j]->device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(ctxs[j]->device, 0);
            }
        (void)hipGetLastError();
    }
    return UDE_OK;
}

extern "C" void ude_comm_destroy(ude_comm* m) {
    if (!m) return;
    if (m->nccl && rccl()) (void)rccl()->CommDestroy(m->nccl);
    delete m;
}

extern "C" int ude_allreduce_grad(ude_comm* m, double* buf_dev, int64_t n) {
    if (!m || !buf_dev || n <= 0) return UDE_ERR_INVALID;
    ude_ctx* c = m->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    NCCLCHK(c, rccl()->AllReduce(buf_dev, buf_dev, (size_t)n, ncclFloat64, ncclSum, m->nccl, c->stream));
    return UDE_OK;
}

// all ranks of one process at once (a single-threaded host must group the calls: RCCL would deadlock otherwise)
extern "C" int ude_allreduce_grad_local(int32_t ndev, ude_comm* const* comms, double* const* bufs_dev, int64_t n) {
    if (!comms || !bufs_dev || ndev < 1 || n <= 0) return UDE_ERR_INVALID;
    Rccl* R = rccl();
    ude_ctx* c0 = comms[0]->ctx;
    if (R->GroupStart) NCCLCHK(c0, R->GroupStart());
    for (int i = 0; i < ndev; ++i) {
        ude_ctx* c = comms[i]->ctx;
        HIPCHK(c, hipSetDevice(c->device));
        NCCLCHK(c, R->AllReduce(bufs_dev[i], bufs_dev[i], (size_t)n, ncclFloat64, ncclSum, comms[i]->nccl, c->stream));
    }
    if (R->GroupEnd) NCCLCHK(c0, R->GroupEnd());
    return UDE_OK;
}

// one-shot P2P reducer: out_d = sum_r in_r in rank order on every device d, then in_d <- out_d.  Ordering across the
// devices' streams by events: (1) every stream's producer work is done before any device reads, (2) every device has
// finished reading before any buffer is overwritten.
extern "C" int ude_allreduce_grad_p2p(int32_t ndev, ude_comm* const* comms, double* const* bufs_dev, int64_t n) {
    if (!comms || !bufs_dev || ndev < 1 || n <= 0) return UDE_ERR_INVALID;
    std::vector<hipEvent_t> ready(ndev), readdone(ndev);
    ude_ctx* c0 = comms[0]->ctx;
    for (int i = 0; i < ndev; ++i) {
        ude_ctx* c = comms[i]->ctx;
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipEventCreateWithFlags(&ready[i], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&readdone[i], hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(ready[i], c->stream));
    }
    int rc = UDE_OK;
    for (int i = 0; i < ndev && rc == UDE_OK; ++i) {
        ude_ctx* c = comms[i]->ctx;
        HIPCHK(c, hipSetDevice(c->device));
        for (int j = 0; j < ndev; ++j)
            if (j != i) HIPCHK(c, hipStreamWaitEvent(c->stream, ready[j], 0));
        // workspace: [ndev pointers | n doubles]
        const size_t ptr_bytes = (sizeof(double*) * ndev + 15) / 16 * 16;
        if ((rc = ensure(c, c->hj[31], ptr_bytes + sizeof(double) * n))) break;
        HIPCHK(c, hipMemcpyAsync(c->hj[31].p, bufs_dev, sizeof(double*) * ndev, hipMemcpyHostToDevice, c->stream));
        double* out = (double*)((char*)c->hj[31].p + ptr_bytes);
        hipLaunchKernelGGL(p2p_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           (const double* const*)c->hj[31].p, (int)ndev, n, out);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(readdone[i], c->stream));
    }
    for (int i = 0; i < ndev && rc == UDE_OK; ++i) {
        ude_ctx* c = comms[i]->ctx;
        HIPCHK(c, hipSetDevice(c->device));
        for (int j = 0; j < ndev; ++j)
            if (j != i) HIPCHK(c, hipStreamWaitEvent(c->stream, readdone[j], 0));
        const size_t ptr_bytes = (sizeof(double*) * ndev + 15) / 16 * 16;
        HIPCHK(c, hipMemcpyAsync(bufs_dev[i], (char*)c->hj[31].p + ptr_bytes, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    }
    for (int i = 0; i < ndev; ++i) {
        (void)hipSetDevice(comms[i]->ctx->device);
        // (events are released once the work queued behind them has been submitted; destruction is deferred by the runtime)
        (void)hipEventDestroy(ready[i]);
        (void)hipEventDestroy(readdone[i]);
    }
    (void)c0;
    return rc;
}
