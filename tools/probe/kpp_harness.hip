// micro-harness: KppUdeW (matrix-core network) vs KppUdeV (vector network + packed contraction) on the same inputs --
// bit comparison of rhs / vjp outputs and cycles per evaluation.   hipcc -I../../universal_differential_equations_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
__device__ long long g_clk[8];
#define UDE_KPPV_CLK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long t_ = __builtin_readcyclecounter(); g_clk[i] += t_ - g_last; g_last = t_; } } while (0)
__device__ long long g_last;
#include "ude_registry.h"
using namespace ude;

template <class Model, bool VJP>
__global__ void __launch_bounds__(Model::G, Model::FWD_BLOCKS) k(const double* theta, ModelConsts mc, const double* uin, const double* lin, double* out_d, double* out_g, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* th = reinterpret_cast<double*>(smem_raw);
    double* scratch = th + Model::theta_lds(mc.n_param);
    Model::stage_theta(th, theta, mc.n_param, threadIdx.x, Model::G);
    __syncthreads();
    typename Model::Ctx c;
    const int r = threadIdx.x;
    Model::init(c, th, scratch, nullptr, 0, mc, r, theta);
    double u[Model::NS], lam[Model::NS], dl[Model::NS], g[Model::NSL];
    static_for<0, Model::NS>([&](auto cc) { const int i = Model::point(cc, r); u[cc] = i < mc.n_state ? uin[i] : 0.0; lam[cc] = i < mc.n_state ? lin[i] : 0.0; dl[cc] = 0; });
    static_for<0, Model::NSL>([&](auto s) { g[s] = 0; });
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (VJP) Model::template vjp<true>(c, u, lam, dl, g);
        else Model::rhs(c, u, dl);
        if (it + 1 < iters) static_for<0, Model::NS>([&](auto cc) { u[cc] = u[cc] + 1e-300 * dl[cc]; });   // (keeps the loop honest; does not change u)
    }
    const long long t1 = wall_clock64();
    static_for<0, Model::NS>([&](auto cc) { const int i = Model::point(cc, r); if (i < mc.n_state) out_d[(size_t)blockIdx.x * 1024 + i] = dl[cc]; });
    if constexpr (VJP) static_for<0, Model::NSL>([&](auto s) { const int p = Model::slot_index(mc, r, s); if (p >= 0) out_g[(size_t)blockIdx.x * 512 + p] = g[s]; });
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <class Model, bool VJP>
double run(const char* name, const std::vector<double>& th, ModelConsts mc, const std::vector<double>& u, const std::vector<double>& lam, std::vector<double>& od, std::vector<double>& og, int iters, int blocks) {
    double *dth, *du, *dl, *dod, *dog; long long* dc;
    (void)hipMalloc(&dth, th.size() * 8); (void)hipMalloc(&du, 1024 * 8); (void)hipMalloc(&dl, 1024 * 8);
    (void)hipMalloc(&dod, (size_t)blocks * 1024 * 8); (void)hipMalloc(&dog, (size_t)blocks * 512 * 8); (void)hipMalloc(&dc, 8);
    (void)hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(du, u.data(), 1024 * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dl, lam.data(), 1024 * 8, hipMemcpyHostToDevice);
    (void)hipMemset(dod, 0, (size_t)blocks * 1024 * 8); (void)hipMemset(dog, 0, (size_t)blocks * 512 * 8);
    const size_t lds = (size_t)(Model::theta_lds(mc.n_param) + (VJP ? Model::SCRATCH : scratch_fwd<Model>::v) + 7 * 1024) * 8;   // (+ the stage storage the real kernel holds)
    (void)hipFuncSetAttribute((const void*)k<Model, VJP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<Model, VJP>), dim3(blocks), dim3(Model::G), lds, 0, dth, mc, du, dl, dod, dog, iters, dc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return -1; }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<Model, VJP>), dim3(blocks), dim3(Model::G), lds, 0, dth, mc, du, dl, dod, dog, iters, dc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h; (void)hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    od.resize(1024); og.resize(512);
    (void)hipMemcpy(od.data(), dod, 1024 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(og.data(), dog, 512 * 8, hipMemcpyDeviceToHost);
    { long long hc[8]; (void)hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)); long long z8[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z8, sizeof(z8));
      if (hc[1]) printf("      phases (clocks per pass, both launches): other %.0f forward %.0f backward %.0f contraction %.0f\n", hc[0] / (8.0 * iters), hc[1] / (8.0 * iters), hc[2] / (8.0 * iters), hc[3] / (8.0 * iters)); }
    printf("%-34s %s n=%4d blocks=%3d: %9.0f shader clocks per evaluation (block 0), %8.3f us per evaluation (launch)\n", name, VJP ? "vjp" : "rhs", mc.n_state, blocks, (double)h / iters * 24.0, ms * 1e3 / iters);
    (void)hipFree(dth); (void)hipFree(du); (void)hipFree(dl); (void)hipFree(dod); (void)hipFree(dog); (void)hipFree(dc);
    return ms;
}
static unsigned long long hash64(const std::vector<double>& a, int n) { unsigned long long h = 1469598103934665603ull; for (int i = 0; i < n; ++i) { unsigned long long b; memcpy(&b, &a[i], 8); h = (h ^ b) * 1099511628211ull; } return h; }
static int cmp(const char* what, const std::vector<double>& a, const std::vector<double>& b, int n) {
    int bad = 0;
    for (int i = 0; i < n; ++i) if (memcmp(&a[i], &b[i], 8) != 0) { if (bad < 5) printf("   %s[%d]: %.17g vs %.17g\n", what, i, a[i], b[i]); ++bad; }
    printf("   %s: %d of %d differ\n", what, bad, n);
    return bad;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int NP = 461 + 5;
    ModelConsts mc; memset(&mc, 0, sizeof(mc));
    mc.n_param = NP; mc.nn_offset = 0; mc.stencil_offset = 461; mc.d0_offset = 465;
    std::vector<double> th(NP);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0 - 0.5; };
    for (auto& x : th) x = rnd();
    th[461] = 1.0 + 0.1 * rnd(); th[462] = -2.0 + 0.1 * rnd(); th[463] = 1.0 + 0.1 * rnd(); th[464] = 0.0; th[465] = 0.3;
    std::vector<double> u(1024), lam(1024);
    for (int i = 0; i < 1024; ++i) { u[i] = 0.5 + 0.5 * std::sin(0.02 * i) + 0.01 * rnd(); lam[i] = rnd(); }
    int bad = 0;
    for (int n : {1024, 1000, 300, 33}) {
        mc.n_state = n;
        std::vector<double> d0, g0, d1, g1;
        run<KppUdeW<NetKpp, 4>, true>("KppUdeW<4> (matrix network)", th, mc, u, lam, d0, g0, iters, 1);
        run<KppUdeV<NetKpp, 4>, true>("KppUdeV<4> (vector network)", th, mc, u, lam, d1, g1, iters, 1);
        printf("   hashes W dlam %016llx grad %016llx | V dlam %016llx grad %016llx\n", hash64(d0, n), hash64(g0, NP), hash64(d1, n), hash64(g1, NP));
        bad += cmp("dlam", d0, d1, n); bad += cmp("grad", g0, g1, NP);
        run<KppUdeV<NetKpp, 4, 32>, true>("KppUdeV<4,32> (half tiles)", th, mc, u, lam, d1, g1, iters, 1);
        bad += cmp("dlam32", d0, d1, n); bad += cmp("grad32", g0, g1, NP);
        run<KppUdeW<NetKpp, 8>, false>("KppUdeW<8> (matrix network)", th, mc, u, lam, d0, g0, iters, 1);
        run<KppUdeV<NetKpp, 8>, false>("KppUdeV<8> (vector network)", th, mc, u, lam, d1, g1, iters, 1);
        bad += cmp("du", d0, d1, n);
        run<KppUdeW<NetKpp, 4>, false>("KppUdeW<4> (matrix network)", th, mc, u, lam, d0, g0, iters, 1);
        run<KppUdeV<NetKpp, 4>, false>("KppUdeV<4> (vector network)", th, mc, u, lam, d1, g1, iters, 1);
        bad += cmp("du4", d0, d1, n);
    }
    mc.n_state = 1024;
    std::vector<double> d0, g0;
    puts("-- all 256 CUs busy");
    run<KppUdeW<NetKpp, 4>, true>("KppUdeW<4>", th, mc, u, lam, d0, g0, iters, 256);
    run<KppUdeV<NetKpp, 4>, true>("KppUdeV<4>", th, mc, u, lam, d0, g0, iters, 256);
    run<KppUdeW<NetKpp, 8>, false>("KppUdeW<8>", th, mc, u, lam, d0, g0, iters, 256);
    run<KppUdeV<NetKpp, 8>, false>("KppUdeV<8>", th, mc, u, lam, d0, g0, iters, 256);
    printf("%s\n", bad ? "MISMATCH" : "all bits equal");
    return bad != 0;
}
