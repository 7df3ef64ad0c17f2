// Which IEEE operation sequence does v_mfma_f64_16x16x4_f64 execute?  D = C + A(16x4) * B(4x16).
// Compares the device result bit for bit with candidate host evaluations (fma chains in both k orders, separately
// rounded products, pairwise trees).  Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o mfma_order_probe mfma_order_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifndef ROW
#define ROW(l, r) (4 * ((l) / 16) + (r))
#endif
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];   // A[i][k], i = l%16, k = l/16
    const double b = B[(l / 16) * 16 + l % 16];  // B[k][j], k = l/16, j = l%16
    v4d c;
    for (int r = 0; r < 4; ++r) c[r] = C[(ROW(l, r)) * 16 + l % 16];
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(ROW(l, r)) * 16 + l % 16] = d[r];
}

static double rnd() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

int main() {
    double hA[64], hB[64], hC[256], hD[256];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    const char* names[] = {"fma chain k ascending", "fma chain k descending", "rounded products, ascending adds",
                           "rounded products, descending adds", "tree (p0+p1)+(p2+p3) + c", "fma pairs: fma(a0,b0,fma(a1,b1,.)) style 0,1 | 2,3",
                           "c + ((p0+p1)+(p2+p3)) with fma inner"};
    long hits[7] = {0}, total = 0;
    srand(1234);
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < 64; ++i) { hA[i] = rnd() * exp2(rand() % 8 - 4); hB[i] = rnd() * exp2(rand() % 8 - 4); }
        for (int i = 0; i < 256; ++i) hC[i] = rnd();
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double a[4], b[4], p[4];
                for (int k = 0; k < 4; ++k) { a[k] = hA[i * 4 + k]; b[k] = hB[k * 16 + j]; p[k] = a[k] * b[k]; }
                const double c = hC[i * 16 + j], got = hD[i * 16 + j];
                double cand[7];
                cand[0] = fma(a[3], b[3], fma(a[2], b[2], fma(a[1], b[1], fma(a[0], b[0], c))));
                cand[1] = fma(a[0], b[0], fma(a[1], b[1], fma(a[2], b[2], fma(a[3], b[3], c))));
                cand[2] = (((c + p[0]) + p[1]) + p[2]) + p[3];
                cand[3] = (((c + p[3]) + p[2]) + p[1]) + p[0];
                cand[4] = ((p[0] + p[1]) + (p[2] + p[3])) + c;
                cand[5] = fma(a[3], b[3], fma(a[2], b[2], 0.0)) + fma(a[1], b[1], fma(a[0], b[0], c));
                cand[6] = c + (fma(a[1], b[1], p[0]) + fma(a[3], b[3], p[2]));
                for (int q = 0; q < 7; ++q) hits[q] += memcmp(&cand[q], &got, 8) == 0;
                total += 1;
            }
    }
    for (int q = 0; q < 7; ++q) printf("%-60s %ld / %ld\n", names[q], hits[q], total);
    return 0;
}
