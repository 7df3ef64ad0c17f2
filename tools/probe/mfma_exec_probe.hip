// What does v_mfma_f64_16x16x4 do under a PARTIAL EXEC mask?  (Needed before ensembles with per-trajectory control flow
// can put trajectories in MFMA columns.)  Lanes whose column j = lane % 16 is >= 8 skip the instruction.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o mfma_exec_probe mfma_exec_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, double* D, int mode) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];
    const double b = B[(l / 16) * 16 + l % 16];
    v4d d = v4d{-7.0, -7.0, -7.0, -7.0};  // sentinel
    const bool active = mode == 0 ? true : mode == 1 ? (l % 16 < 8) : (l / 16 < 2);
    if (active) d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(l / 16 + 4 * r) * 16 + l % 16] = d[r];
}

int main() {
    double hA[64], hB[64], hD[256], *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    srand(7);
    for (int i = 0; i < 64; ++i) { hA[i] = (double)rand() / RAND_MAX - 0.5; hB[i] = (double)rand() / RAND_MAX - 0.5; }
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode);
        (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        long ok_active = 0, n_active = 0, sentinel_inactive = 0, n_inactive = 0, correct_inactive = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double d = 0.0;
                for (int k = 0; k < 4; ++k) d = fma(hA[i * 4 + k], hB[k * 16 + j], d);
                // element (i, j) lives in lane l = (i % 4) * 16 + j, register i / 4
                const int l = (i % 4) * 16 + j;
                const bool active = mode == 0 ? true : mode == 1 ? (l % 16 < 8) : (l / 16 < 2);
                const double got = hD[i * 16 + j];
                if (active) { n_active++; ok_active += memcmp(&d, &got, 8) == 0; }
                else { n_inactive++; sentinel_inactive += got == -7.0; correct_inactive += memcmp(&d, &got, 8) == 0; }
            }
        printf("mode %d (%s): active outputs correct %ld/%ld; inactive lanes kept the sentinel %ld/%ld (hold the product anyway: %ld)\n", mode,
               mode == 0 ? "all lanes" : mode == 1 ? "columns j < 8 active" : "k-quarters 0,1 active", ok_active, n_active, sentinel_inactive,
               n_inactive, correct_inactive);
    }
    return 0;
}
