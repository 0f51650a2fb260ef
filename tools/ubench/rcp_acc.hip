// accuracy of raw v_rcp_f64 and after 1 / 2 Newton steps (relative error vs IEEE 1/x)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    double e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r); r1[i] = r;
    e = __builtin_fma(-d, r, 1.0); r = __builtin_fma(r, e, r); r2[i] = r;
}
int main() {
    const int n = 1 << 22;
    double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0);
        hx[i] = (i & 1 ? -1 : 1) * std::ldexp(1.0 + u, (int)(s % 41) - 20); }
    double *x, *r0, *r1, *r2;
    hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
    hipMemcpy(x, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, r0, r1, r2, n);
    hipMemcpy(h0, r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, r1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, r2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { long double t = 1.0L / hx[i];
        m0 = fmax(m0, (double)fabsl((h0[i] - t) / t)); m1 = fmax(m1, (double)fabsl((h1[i] - t) / t)); m2 = fmax(m2, (double)fabsl((h2[i] - t) / t)); }
    printf("max rel err: raw %.3g (2^%.1f)  1NR %.3g (2^%.1f)  2NR %.3g (2^%.1f)\n", m0, log2(m0), m1, log2(m1), m2, log2(m2));
    return 0;
}
