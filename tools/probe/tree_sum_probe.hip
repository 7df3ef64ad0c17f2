// probe: wavefront tree sum, xor-butterfly (ds_bpermute for the two cross-row levels) vs DPP row_bcast15/31 + readlane(63)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
template <int CTRL> __device__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int RM> __device__ double dpp_rows(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, RM, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, RM, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ double rows16(double x) {
    x += dpp_mov<0xB1>(x); x += dpp_mov<0x4E>(x); x += dpp_mov<0x141>(x); x += dpp_mov<0x140>(x);
    return x;
}
__global__ void k(const double* in, double* a, double* b, double* dbg) {
    double x = in[blockIdx.x * 64 + threadIdx.x];
    double r = rows16(x);
    double y = r; y += __shfl_xor(y, 16, 64); y += __shfl_xor(y, 32, 64);
    double z = r;
    z += dpp_rows<0x142, 0xA>(z);
    dbg[blockIdx.x * 128 + threadIdx.x] = z;
    z += dpp_rows<0x143, 0xC>(z);
    dbg[blockIdx.x * 128 + 64 + threadIdx.x] = z;
    const int lo = __builtin_amdgcn_readlane(__double2loint(z), 63), hi = __builtin_amdgcn_readlane(__double2hiint(z), 63);
    a[blockIdx.x * 64 + threadIdx.x] = y;
    b[blockIdx.x * 64 + threadIdx.x] = __hiloint2double(hi, lo);
}
int main() {
    const int NB = 4096;
    double *h = (double*)malloc(NB * 64 * 8), *ha = (double*)malloc(NB * 64 * 8), *hb = (double*)malloc(NB * 64 * 8), *hd = (double*)malloc(NB * 128 * 8);
    srand(1);
    for (int i = 0; i < NB * 64; ++i) {
        int m = rand() % 10;
        double v = (rand() / (double)RAND_MAX - 0.5) * pow(10.0, rand() % 20 - 10);
        h[i] = m == 0 ? 0.0 : m == 1 ? -0.0 : v;
    }
    for (int i = 0; i < 64 * 8; ++i) h[i] = (i & 1) ? -0.0 : 0.0;      // blocks of signed zeros
    double *d, *da, *db, *dd;
    hipMalloc(&d, NB * 64 * 8); hipMalloc(&da, NB * 64 * 8); hipMalloc(&db, NB * 64 * 8); hipMalloc(&dd, NB * 128 * 8);
    hipMemcpy(d, h, NB * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, d, da, db, dd);
    hipMemcpy(ha, da, NB * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, db, NB * 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hd, dd, NB * 128 * 8, hipMemcpyDeviceToHost);
    long bad = 0, badlanes = 0;
    for (int blk = 0; blk < NB; ++blk) {
        for (int l = 1; l < 64; ++l) if (memcmp(&ha[blk * 64], &ha[blk * 64 + l], 8)) badlanes++;
        if (memcmp(&ha[blk * 64], &hb[blk * 64], 8)) {
            if (bad < 5) printf("blk %d: butterfly %.17g (%016llx) dpp %.17g (%016llx)  rows after A: %.17g %.17g %.17g %.17g\n", blk, ha[blk * 64],
                                *(unsigned long long*)&ha[blk * 64], hb[blk * 64], *(unsigned long long*)&hb[blk * 64], hd[blk * 128 + 0], hd[blk * 128 + 16],
                                hd[blk * 128 + 32], hd[blk * 128 + 48]);
            bad++;
        }
    }
    printf("blocks %d: mismatching %ld ; butterfly lanes disagreeing %ld\n", NB, bad, badlanes);
    return 0;
}
