// ude_comm.hip -- the one exchange step of the data-parallel gradient behind the C ABI:
// all-reduce(sum) of double[np + 4] = gradient (+) loss (+) (nf, naccept, nreject) across the GPUs of a node.
//   * RCCL over xGMI: ncclAllReduce on the context's stream (one process per GPU: ude_comm_create with a shared unique
//     id; or all devices in one process: ude_comm_create_local = ncclCommInitAll).  RCCL is bound with dlopen so that
//     libudecore.so loads on a box without it and shares whichever librccl the process already mapped (PyTorch's).
//   * one-shot P2P reducer (single process, peer access over xGMI): every device reads the buffers of all ranks and adds
//     them in RANK ORDER -- a deterministic fp64 sum, identical bits on every device; per call and device ONE kernel and
//     ONE device-to-device copy between pre-created events (the pointer table is device resident and re-uploaded only when
//     the caller's buffers change), instead of a 2(N-1)-step ring for a 704 B .. 36 KB payload.  Works without librccl.
// The reference is single-process CPU Julia: nothing is replaced, this is the multi-GPU row of the scope table.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
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
    static std::once_flag once;
    std::call_once(once, [] {
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
    });
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

// state of the one-shot P2P reducer shared by the communicators of one ude_comm_create_local call: everything a call needs
// exists before the first call (events) or is cached across calls (device-resident pointer table, output buffer)
struct P2PGroup {
    int ndev = 0;
    bool peers_ok = false;  // every device can read every other device's memory (checked, not assumed)
    std::vector<int> device;
    std::vector<hipEvent_t> ready, readdone;          // one pair per device, created once
    std::vector<void*> table;                          // per device: [ndev pointers | n doubles] in HBM
    std::vector<size_t> table_cap;
    std::vector<double*> cached;                       // the buffer pointers the tables currently hold
    size_t ptr_bytes() const { return (sizeof(double*) * ndev + 15) / 16 * 16; }
    ~P2PGroup() {
        for (int i = 0; i < ndev; ++i) {
            (void)hipSetDevice(device[i]);
            (void)hipDeviceSynchronize();
            if (ready[i]) (void)hipEventDestroy(ready[i]);
            if (readdone[i]) (void)hipEventDestroy(readdone[i]);
            if (table[i]) (void)hipFree(table[i]);
        }
    }
};

struct P2PMp {
    int nranks = 0, rank = 0;
    size_t nmax = 0;
    char* win = nullptr;                  // this rank's window: [2][nmax] doubles | 64-byte control block (2 x uint64 call flags, uint32 timeout
                                          // counter at +16) | nranks x uint64 "closed" words, word r written by rank r when it disconnects
    std::vector<char*> peer;              // every rank's window as mapped into this process (peer[rank] == win)
    char** peer_dev = nullptr;            // the same table in HBM
    uint64_t seq = 0;
    bool broken = false;                  // a call of this rank timed out (seen by ude_comm_p2p_status / disconnect): every later call is refused
    bool disconnected = false;            // the peers' windows are closed (ude_comm_p2p_disconnect): only ude_comm_destroy is left
    static size_t bytes(size_t nmax, int nranks) { return 2 * nmax * sizeof(double) + 64 + sizeof(uint64_t) * (size_t)nranks; }
};

struct ude_comm {
    ude_ctx* ctx = nullptr;
    ncclComm_t nccl = nullptr;
    int nranks = 1, rank = 0;
    std::shared_ptr<P2PGroup> p2p;   // devices of one process (ude_comm_create_local)
    std::shared_ptr<P2PMp> mp;       // one process per GPU (ude_comm_create_p2p)
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
    for (int i = 0; i < ndev; ++i)
        if (!ctxs[i]) return UDE_ERR_INVALID;
    Rccl* R = rccl();  // optional: without librccl the communicators still serve the P2P reducer
    std::vector<int> devs(ndev);
    std::vector<ncclComm_t> comms(ndev, nullptr);
    for (int i = 0; i < ndev; ++i) devs[i] = ctxs[i]->device;
    if (R) NCCLCHK(ctxs[0], R->CommInitAll(comms.data(), ndev, devs.data()));
    auto grp = std::make_shared<P2PGroup>();
    grp->ndev = ndev;
    grp->device = devs;
    grp->ready.assign(ndev, nullptr);
    grp->readdone.assign(ndev, nullptr);
    grp->table.assign(ndev, nullptr);
    grp->table_cap.assign(ndev, 0);
    grp->cached.assign(ndev, nullptr);
    bool ok = true, peers = true;
    for (int i = 0; i < ndev && ok; ++i) {
        ok = hipSetDevice(devs[i]) == hipSuccess &&
             hipEventCreateWithFlags(&grp->ready[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&grp->readdone[i], hipEventDisableTiming) == hipSuccess;
        for (int j = 0; j < ndev && ok; ++j) {
            if (devs[j] == devs[i]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs[i], devs[j]) != hipSuccess || !can) { peers = false; continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(devs[j], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peers = false;
        }
        (void)hipGetLastError();
    }
    if (!ok) {  // (the group's destructor releases what was created)
        for (int i = 0; i < ndev; ++i)
            if (comms[i] && R) (void)R->CommDestroy(comms[i]);
        return fail(ctxs[0], UDE_ERR_HIP, "could not create the events of the P2P reducer");
    }
    grp->peers_ok = peers;
    for (int i = 0; i < ndev; ++i) {
        ude_comm* m = new ude_comm();
        m->ctx = ctxs[i]; m->nccl = comms[i]; m->nranks = ndev; m->rank = i; m->p2p = grp;
        out[i] = m;
    }
    return UDE_OK;
}

static int p2p_disconnect_impl(ude_comm* m);

// Teardown of a cross-process communicator is a handshake, not a local free: a peer may still be inside its last call, reading this
// rank's window.  ude_comm_p2p_disconnect (called here if the host did not): every rank, stream-ordered behind its own last call,
// stores a "closed" word INTO every peer's window and then waits -- on its OWN window only -- until every peer's word has arrived;
// from then on no peer touches this window again, and this rank never touches a peer's window again, so the mappings are closed
// and the window is freed.  A dead peer ends the wait after the reducer's timeout.  (A host with an out-of-band barrier calls
// ude_comm_p2p_disconnect on every rank, then its barrier, then ude_comm_destroy: every importer has unmapped before any exporter frees.)
extern "C" void ude_comm_destroy(ude_comm* m) {
    if (!m) return;
    if (m->nccl && rccl()) (void)rccl()->CommDestroy(m->nccl);
    if (m->mp) {
        (void)hipSetDevice(m->ctx->device);
        (void)p2p_disconnect_impl(m);
        (void)hipDeviceSynchronize();
        if (m->mp->peer_dev) (void)hipFree(m->mp->peer_dev);
        if (m->mp->win) (void)hipFree(m->mp->win);
    }
    delete m;
}

// payload[np + 1 .. np + 3] = (sum nf, sum naccept, sum nreject) of this device's trajectories, forward + backward, as doubles:
// the tail of the double[np + 4] exchange payload for hosts without a device array library (integer sums: exact, order-free)
namespace {
__global__ void __launch_bounds__(1024) pack_counters_kernel(const long long* __restrict__ stats, long long N, double* __restrict__ out) {
    __shared__ long long part[3][16];
    long long a[3] = {0, 0, 0};
    for (long long j = threadIdx.x; j < N; j += blockDim.x) {
        const long long* s = stats + 8 * j;
        a[0] += s[0] + s[4];
        a[1] += s[1] + s[5];
        a[2] += s[2] + s[6];
    }
    for (int k = 0; k < 3; ++k) {
        long long v = a[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        long long v = 0;
        for (int w = 0; w < 16; ++w) v += part[threadIdx.x][w];
        out[threadIdx.x] = (double)v;
    }
}
}  // namespace

extern "C" int ude_pack_counters_dev(ude_ctx* c, int64_t N, const int64_t* stats_dev, double* payload_dev, int32_t n_param) {
    if (!c || !stats_dev || !payload_dev || N <= 0 || n_param < 0) return UDE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(pack_counters_kernel, dim3(1), dim3(1024), 0, c->stream, (const long long*)stats_dev, (long long)N, payload_dev + n_param + 1);
    HIPCHK(c, hipGetLastError());
    return UDE_OK;
}

extern "C" int ude_allreduce_grad(ude_comm* m, double* buf_dev, int64_t n) {
    if (!m || !buf_dev || n <= 0) return UDE_ERR_INVALID;
    ude_ctx* c = m->ctx;
    if (!m->nccl || !rccl()) return fail(c, UDE_ERR_UNSUPPORTED, "this communicator has no RCCL handle (librccl could not be loaded)");
    HIPCHK(c, hipSetDevice(c->device));
    NCCLCHK(c, rccl()->AllReduce(buf_dev, buf_dev, (size_t)n, ncclFloat64, ncclSum, m->nccl, c->stream));
    return UDE_OK;
}

// all ranks of one process at once (a single-threaded host must group the calls: RCCL would deadlock otherwise)
extern "C" int ude_allreduce_grad_local(int32_t ndev, ude_comm* const* comms, double* const* bufs_dev, int64_t n) {
    if (!comms || !bufs_dev || ndev < 1 || n <= 0) return UDE_ERR_INVALID;
    for (int i = 0; i < ndev; ++i)
        if (!comms[i] || !bufs_dev[i]) return UDE_ERR_INVALID;
    Rccl* R = rccl();
    ude_ctx* c0 = comms[0]->ctx;
    if (!R || !comms[0]->nccl) return fail(c0, UDE_ERR_UNSUPPORTED, "librccl could not be loaded (ude_allreduce_grad_p2p needs no RCCL)");
    if (R->GroupStart) NCCLCHK(c0, R->GroupStart());
    int rc = UDE_OK;
    for (int i = 0; i < ndev && rc == UDE_OK; ++i) {
        ude_ctx* c = comms[i]->ctx;
        if (hipSetDevice(c->device) != hipSuccess) { rc = fail(c, UDE_ERR_HIP, "hipSetDevice(%d) failed", c->device); break; }
        const ncclResult_t r = R->AllReduce(bufs_dev[i], bufs_dev[i], (size_t)n, ncclFloat64, ncclSum, comms[i]->nccl, c->stream);
        if (r != ncclSuccess) rc = fail(c, UDE_ERR_HIP, "ncclAllReduce failed: %s", R->GetErrorString ? R->GetErrorString(r) : "rccl error");
    }
    // the group is ALWAYS closed: an error between GroupStart and GroupEnd must not leave RCCL in group mode
    if (R->GroupEnd) {
        const ncclResult_t r = R->GroupEnd();
        if (r != ncclSuccess && rc == UDE_OK) rc = fail(c0, UDE_ERR_HIP, "ncclGroupEnd failed: %s", R->GetErrorString ? R->GetErrorString(r) : "rccl error");
    }
    return rc;
}

// one-shot P2P reducer: out_d = sum_r in_r in rank order on every device d, then in_d <- out_d.  Ordering across the
// devices' streams by the group's pre-created events: (1) every stream's producer work is done before any device reads,
// (2) every device has finished reading before any buffer is overwritten.  Per call and device: one event record, ndev - 1
// stream waits, ONE kernel, one record, ndev - 1 waits, ONE device-to-device copy -- no allocation, no event creation and no
// host-to-device copy once the tables hold the caller's buffers.
extern "C" int ude_allreduce_grad_p2p(int32_t ndev, ude_comm* const* comms, double* const* bufs_dev, int64_t n) {
    if (!comms || !bufs_dev || ndev < 1 || n <= 0) return UDE_ERR_INVALID;
    for (int i = 0; i < ndev; ++i)
        if (!comms[i] || !bufs_dev[i]) return UDE_ERR_INVALID;
    ude_ctx* c0 = comms[0]->ctx;
    P2PGroup* g = comms[0]->p2p.get();
    if (!g || g->ndev != ndev) return fail(c0, UDE_ERR_INVALID, "ude_allreduce_grad_p2p needs the communicators of one ude_comm_create_local call");
    for (int i = 0; i < ndev; ++i)
        if (comms[i]->p2p.get() != g || comms[i]->rank != i) return fail(c0, UDE_ERR_INVALID, "communicators out of order / from different groups");
    if (!g->peers_ok) return fail(c0, UDE_ERR_UNSUPPORTED, "peer access between the devices of this group is not available: use ude_allreduce_grad_local");
    const size_t pb = g->ptr_bytes(), need = pb + sizeof(double) * (size_t)n;
    // (re)build the device-resident tables only when the payload grew or the caller's buffers moved
    bool same = true;
    for (int i = 0; i < ndev; ++i) same = same && g->cached[i] == bufs_dev[i] && g->table_cap[i] >= need;
    if (!same) {
        for (int i = 0; i < ndev; ++i) {
            ude_ctx* c = comms[i]->ctx;
            HIPCHK(c, hipSetDevice(c->device));
            if (g->table_cap[i] < need) {
                HIPCHK(c, hipDeviceSynchronize());
                if (g->table[i]) HIPCHK(c, hipFree(g->table[i]));
                g->table[i] = nullptr;
                g->table_cap[i] = 0;
                if (hipMalloc(&g->table[i], need + need / 4) != hipSuccess) return fail(c, UDE_ERR_NOMEM, "hipMalloc(%zu) failed", need + need / 4);
                g->table_cap[i] = need + need / 4;
            }
            // stream-ordered after whatever still reads the old table; the source is copied before the call returns
            HIPCHK(c, hipMemcpyAsync(g->table[i], bufs_dev, sizeof(double*) * ndev, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        for (int i = 0; i < ndev; ++i) g->cached[i] = bufs_dev[i];
    }
    for (int i = 0; i < ndev; ++i) {
        ude_ctx* c = comms[i]->ctx;
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipEventRecord(g->ready[i], c->stream));
    }
    for (int i = 0; i < ndev; ++i) {
        ude_ctx* c = comms[i]->ctx;
        HIPCHK(c, hipSetDevice(c->device));
        for (int j = 0; j < ndev; ++j)
            if (j != i) HIPCHK(c, hipStreamWaitEvent(c->stream, g->ready[j], 0));
        double* out = (double*)((char*)g->table[i] + pb);
        hipLaunchKernelGGL(p2p_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           (const double* const*)g->table[i], (int)ndev, n, out);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(g->readdone[i], c->stream));
    }
    for (int i = 0; i < ndev; ++i) {
        ude_ctx* c = comms[i]->ctx;
        HIPCHK(c, hipSetDevice(c->device));
        for (int j = 0; j < ndev; ++j)
            if (j != i) HIPCHK(c, hipStreamWaitEvent(c->stream, g->readdone[j], 0));
        HIPCHK(c, hipMemcpyAsync(bufs_dev[i], (char*)g->table[i] + pb, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    }
    return UDE_OK;
}

// ---------------------------------------------------------------------------------------------
// One process per GPU: the one-shot P2P reducer ACROSS PROCESSES (round 4; SURVEY.md 8(e) asks to compare it with RCCL from the
// layout `bench.py --gpus N` uses).  Every rank owns an exchange WINDOW in its own HBM -- two payload slots (parity of the call
// number) and two flag words -- exported with hipIpcGetMemHandle and opened by every peer (xGMI peer reads).  A call is ONE
// kernel per rank:
//   1. copy the rank's payload into its slot[k & 1], fence, publish flag[k & 1] = k (system-scope release);
//   2. for r = 0 .. nranks-1 IN RANK ORDER: wait until rank r's flag[k & 1] >= k (acquire), add its slot (system-scope loads):
//      a deterministic fp64 sum, identical bits on every rank;
//   3. write the sum back into the caller's buffer.
// Slot reuse needs no second handshake: slot[k & 1] is overwritten by call k + 2, and a rank that has SEEN every peer's flag of call
// k + 1 (step 2 of its own call k + 1) knows every peer has finished call k (calls are stream-ordered per rank).
// A peer that never arrives does not hang the GPU: the wait gives up after `UDE_P2P_TIMEOUT_MS` (default 5000), the result is NaN and a
// sticky counter is raised (ude_comm_p2p_status).  No host synchronisation, no RCCL.
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ uint64_t* p2p_flags(char* w, size_t nmax) { return (uint64_t*)(w + 2 * nmax * sizeof(double)); }

__device__ __forceinline__ unsigned* p2p_timeouts(char* w, size_t nmax) { return (unsigned*)(p2p_flags(w, nmax) + 2); }
__device__ __forceinline__ uint64_t* p2p_closed(char* w, size_t nmax) { return p2p_flags(w, nmax) + 8; }

// elements per thread of the one-block reducer: 1024 x 16 = 16384 doubles (round 6: the neural ODE's payload, np + 4 = 9291, did not fit
// the 8192 of round 4 -- found by the first eight-rank test that used the workloads' own payload sizes)
constexpr int P2P_EPT = 16;

__global__ void __launch_bounds__(1024) p2p_mp_kernel(char* const* peers, int nranks, int rank, size_t nmax, uint64_t seq, double* buf, int64_t n,
                                                      unsigned long long timeout_ticks) {
    const int q = (int)(seq & 1);
    char* mine = peers[rank];
    // A timeout is STICKY: once a call of this rank has given up, the slot-reuse rule (seeing every peer's flag of call k + 1 proves
    // every peer finished call k) no longer holds, so this rank never publishes again -- its slots keep the payloads of its last two
    // good calls, a late peer can still finish THOSE calls correctly and then times out itself -- and every later call is NaN.
    if (__hip_atomic_load(p2p_timeouts(mine, nmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) buf[i] = __builtin_nan("");
        return;
    }
    double* myslot = (double*)mine + (size_t)q * nmax;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
        __hip_atomic_store(myslot + i, buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(p2p_flags(mine, nmax) + q, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ int lost;
    if (threadIdx.x == 0) lost = 0;
    __syncthreads();
    // (the sum of one element is kept in a register across the ranks: n <= blockDim.x * P2P_EPT)
    double acc[P2P_EPT];
    for (int r = 0; r < nranks; ++r) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            uint64_t f;
            while ((f = __hip_atomic_load(p2p_flags(peers[r], nmax) + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < seq) {
                // a peer that has given up will never publish: fail with it at once instead of after the full timeout
                if (__hip_atomic_load(p2p_timeouts(peers[r], nmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { lost = 1; break; }
                if (wall_clock64() - t0 > timeout_ticks) { lost = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (f > seq) lost = 1;   // a slot that already holds a LATER call's payload: the ranks are out of step, never sum it
        }
        __syncthreads();
        if (lost) break;
        const double* slot = (const double*)peers[r] + (size_t)q * nmax;
#pragma unroll
        for (int u = 0; u < P2P_EPT; ++u) {
            const int64_t i = (int64_t)threadIdx.x + (int64_t)u * blockDim.x;
            if (i < n) {
                const double v = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                acc[u] = r == 0 ? v : acc[u] + v;
            }
        }
    }
    __syncthreads();
    const bool bad = lost != 0;
#pragma unroll
    for (int u = 0; u < P2P_EPT; ++u) {
        const int64_t i = (int64_t)threadIdx.x + (int64_t)u * blockDim.x;
        if (i < n) buf[i] = bad ? __builtin_nan("") : acc[u];
    }
    if (bad && threadIdx.x == 0) {
        __hip_atomic_fetch_add(p2p_timeouts(mine, nmax), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the teardown handshake (see ude_comm_destroy): push "closed" into every peer's window, wait for every peer's word in MY window
__global__ void p2p_close_kernel(char* const* peers, int nranks, int rank, size_t nmax, unsigned long long timeout_ticks, int* incomplete) {
    if (threadIdx.x != 0) return;
    __threadfence_system();
    // (a peer that already ran its own disconnect into the timeout may have FREED its window; this rank's mapping of it -- opened with
    //  hipIpcOpenMemHandle, closed only after this kernel -- keeps the pages alive, so the store below lands in mapped memory either way)
    for (int r = 0; r < nranks; ++r)
        __hip_atomic_store(p2p_closed(peers[r], nmax) + rank, (uint64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    char* mine = peers[rank];
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < nranks; ++r)
        while (__hip_atomic_load(p2p_closed(mine, nmax) + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
            if (wall_clock64() - t0 > timeout_ticks) { *incomplete = 1; return; }
            __builtin_amdgcn_s_sleep(8);
        }
}

unsigned long long p2p_timeout_ticks() {
    static const unsigned long long ticks = [] {
        const char* e = getenv("UDE_P2P_TIMEOUT_MS");
        const unsigned long long ms = e ? strtoull(e, nullptr, 10) : 5000ull;
        return ms * 100000ull;   // wall_clock64: 100 MHz
    }();
    return ticks;
}
}  // namespace

static int p2p_disconnect_impl(ude_comm* m) {
    ude_ctx* c = m->ctx;
    P2PMp& mp = *m->mp;
    if (mp.disconnected) return UDE_OK;
    // (advisor, round 5) every exit of this function closes the peers' IPC mappings and frees the flag word; `disconnected` is set once
    // that has happened -- an early return used to leave the peer mappings open behind a communicator that said it had disconnected
    int rc = UDE_OK;
    int* flag = nullptr;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == UDE_OK) rc = fail(c, UDE_ERR_HIP, "ude_comm_p2p_disconnect: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (hip_ok(hipSetDevice(c->device), "hipSetDevice") && mp.peer_dev && (int)mp.peer.size() == mp.nranks) {
        int inc = 0;
        if (hip_ok(hipMalloc((void**)&flag, sizeof(int)), "hipMalloc") && hip_ok(hipMemsetAsync(flag, 0, sizeof(int), c->stream), "hipMemsetAsync")) {
            hipLaunchKernelGGL(p2p_close_kernel, dim3(1), dim3(64), 0, c->stream, (char* const*)mp.peer_dev, mp.nranks, mp.rank, mp.nmax, p2p_timeout_ticks(), flag);
            hip_ok(hipGetLastError(), "launch of the close handshake");
            hip_ok(hipMemcpyAsync(&inc, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
            hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        }
        if (inc && rc == UDE_OK) rc = fail(c, UDE_ERR_TIMEOUT, "ude_comm_p2p_disconnect: a peer did not reach its own disconnect within the timeout (it may still map this rank's window)");
    }
    if (flag) (void)hipFree(flag);
    for (int r = 0; r < (int)mp.peer.size(); ++r)
        if (r != mp.rank && mp.peer[r]) { (void)hipIpcCloseMemHandle(mp.peer[r]); mp.peer[r] = nullptr; }
    mp.disconnected = true;
    return rc;
}

extern "C" int ude_comm_p2p_disconnect(ude_comm* m) {
    if (!m || !m->mp) return UDE_ERR_INVALID;
    return p2p_disconnect_impl(m);
}

extern "C" int ude_comm_create_p2p(ude_ctx* c, int32_t nranks, int32_t rank, int64_t n_max, char handle_out[64], ude_comm** out) {
    if (!c || !out || !handle_out || nranks < 1 || rank < 0 || rank >= nranks || n_max <= 0) return UDE_ERR_INVALID;
    if (n_max > 1024 * P2P_EPT) return fail(c, UDE_ERR_UNSUPPORTED, "the cross-process P2P reducer is a one-block kernel: payloads up to 16384 doubles (np + 4 of every model of the reference fits: the largest is the neural ODE's 9291)");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    auto mp = std::make_shared<P2PMp>();
    mp->nranks = nranks; mp->rank = rank; mp->nmax = (size_t)n_max;
    void* w = nullptr;
    // fine-grained device memory: peer loads and stores at system scope are coherent while the kernels run.  Coarse-grained HBM is
    // only guaranteed visible to other agents at kernel boundaries -- a spin-wait on it may read stale flags or payload -- so there
    // is NO fallback to plain hipMalloc: without a fine-grained pool this transport is unsupported (use RCCL)
    const size_t wbytes = P2PMp::bytes(mp->nmax, nranks);
    if (hipExtMallocWithFlags(&w, wbytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, UDE_ERR_UNSUPPORTED, "no fine-grained device memory for the P2P exchange window (%zu bytes): the cross-process reducer needs it", wbytes);
    }
    mp->win = (char*)w;
    HIPCHK(c, hipMemset(w, 0, wbytes));
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, w);
    if (e != hipSuccess) {
        (void)hipFree(w);
        return fail(c, UDE_ERR_HIP, "hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set where the driver only supports dmabuf IPC)", hipGetErrorString(e));
    }
    memcpy(handle_out, &h, 64);
    ude_comm* m = new ude_comm();
    m->ctx = c; m->nranks = nranks; m->rank = rank; m->mp = mp;
    *out = m;
    return UDE_OK;
}

extern "C" int ude_comm_p2p_connect(ude_comm* m, const char* handles /* nranks x 64, rank order */) {
    if (!m || !m->mp || !handles) return UDE_ERR_INVALID;
    ude_ctx* c = m->ctx;
    P2PMp& mp = *m->mp;
    HIPCHK(c, hipSetDevice(c->device));
    mp.peer.assign(mp.nranks, nullptr);
    for (int r = 0; r < mp.nranks; ++r) {
        if (r == mp.rank) { mp.peer[r] = mp.win; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * 64, 64);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(c, UDE_ERR_HIP, "hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(e));
        mp.peer[r] = (char*)p;
    }
    if (!mp.peer_dev) HIPCHK(c, hipMalloc((void**)&mp.peer_dev, sizeof(char*) * mp.nranks));
    HIPCHK(c, hipMemcpy(mp.peer_dev, mp.peer.data(), sizeof(char*) * mp.nranks, hipMemcpyHostToDevice));
    return UDE_OK;
}

extern "C" int ude_allreduce_grad_p2p_mp(ude_comm* m, double* buf_dev, int64_t n) {
    if (!m || !m->mp || !buf_dev || n <= 0) return UDE_ERR_INVALID;
    ude_ctx* c = m->ctx;
    P2PMp& mp = *m->mp;
    if (!mp.peer_dev) return fail(c, UDE_ERR_INVALID, "ude_comm_p2p_connect has not been called");
    if ((size_t)n > mp.nmax) return fail(c, UDE_ERR_INVALID, "payload of %lld doubles exceeds the window (%zu)", (long long)n, mp.nmax);
    if (mp.disconnected) return fail(c, UDE_ERR_INVALID, "this communicator has been disconnected");
    if (mp.broken) return fail(c, UDE_ERR_TIMEOUT, "an earlier all-reduce of this communicator timed out: the ranks are out of step, create a new communicator");
    HIPCHK(c, hipSetDevice(c->device));
    const unsigned long long ticks = p2p_timeout_ticks();
    mp.seq += 1;
    hipLaunchKernelGGL(p2p_mp_kernel, dim3(1), dim3(1024), 0, c->stream, (char* const*)mp.peer_dev, mp.nranks, mp.rank, mp.nmax, (uint64_t)mp.seq,
                       buf_dev, n, ticks);
    HIPCHK(c, hipGetLastError());
    return UDE_OK;
}

// number of calls of this rank that gave up waiting for a peer (their result is NaN); blocks on the context's stream
extern "C" int ude_comm_p2p_status(ude_comm* m, int32_t* timeouts) {
    if (!m || !m->mp || !timeouts) return UDE_ERR_INVALID;
    ude_ctx* c = m->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned v = 0;
    HIPCHK(c, hipMemcpy(&v, m->mp->win + 2 * m->mp->nmax * sizeof(double) + 16, sizeof(unsigned), hipMemcpyDeviceToHost));
    *timeouts = (int32_t)v;
    if (v) m->mp->broken = true;   // (the kernels refuse by themselves -- NaN, nothing published; from here on the host call fails too)
    return UDE_OK;
}
