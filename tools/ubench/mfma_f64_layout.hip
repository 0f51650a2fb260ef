// Checks the operand / result lane maps of v_mfma_f64_16x16x4_f64 used by gn_build_data_kernel's Gram stage:
// A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15], D[row = (lane>>4) + 4*reg][col = lane&15].
// Exact integer data, K = 8 (two steps).  build: hipcc --offload-arch=gfx950 -O3 mfma_f64_layout.hip -o mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* C) {
    const int lane = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    for (int t = 0; t < 2; ++t) {
        const double a = A[(lane & 15) * 8 + 4 * t + (lane >> 4)];
        const double b = B[(4 * t + (lane >> 4)) * 16 + (lane & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}
int main() {
    double hA[16 * 8], hB[8 * 16], hC[256], ref[256];
    for (int i = 0; i < 128; ++i) { hA[i] = (i * 7 + 3) % 11 - 5; hB[i] = (i * 5 + 1) % 13 - 6; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 8; ++kk) s += hA[i * 8 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hC[i] != ref[i];
    printf("mfma_f64_16x16x4 layout check: %d mismatches of 256\n", bad);
    return bad != 0;
}
