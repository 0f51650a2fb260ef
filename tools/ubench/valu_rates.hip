// Micro-benchmark: VALU issue cost (cycles per wave64 instruction per SIMD) of the
// instructions the voxel kernels lean on, on gfx950.  8 waves/SIMD, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;   // independent chains

#define KERNEL(NAME, DECL, BODY, SINK)                                          \
__global__ __launch_bounds__(256) void NAME(double *out, double seed) {         \
    DECL;                                                                        \
    for (int it = 0; it < ITERS; ++it) {                                         \
        _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) { BODY; }             \
    }                                                                            \
    double acc = 0; _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) acc += SINK; \
    if (acc == 12345.678) out[threadIdx.x] = acc;                                \
}

KERNEL(k_fma_f64, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_add_f64, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_add_f64 %0, %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_mul_f64, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_mul_f64 %0, %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_rcp_f64, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_ldexp_f64, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a[k])), a[k])
KERNEL(k_rndne_f64, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_rndne_f64 %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_min_f64, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_min_f64 %0, %0, %0" : "+v"(a[k])), a[k])
KERNEL(k_cvt_f64_i32, double a[UNROLL]; int b[UNROLL]; for (int k=0;k<UNROLL;++k) {a[k]=seed; b[k]=k+threadIdx.x;}, asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[k]) : "v"(b[k])), a[k])
KERNEL(k_cvt_i32_f64, double a[UNROLL]; int b[UNROLL]; for (int k=0;k<UNROLL;++k) {a[k]=seed+k; b[k]=0;}, asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b[k]) : "v"(a[k])), (double)b[k])
KERNEL(k_cvt_f64_f32, double a[UNROLL]; float b[UNROLL]; for (int k=0;k<UNROLL;++k) {a[k]=seed; b[k]=k+threadIdx.x;}, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(b[k])), a[k])
KERNEL(k_cvt_f32_f64, double a[UNROLL]; float b[UNROLL]; for (int k=0;k<UNROLL;++k) {a[k]=seed+k; b[k]=0;}, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(b[k]) : "v"(a[k])), (double)b[k])
KERNEL(k_cmp_f64, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_cmp_lt_f64 vcc, %0, %0" :: "v"(a[k]) : "vcc"), a[k])
KERNEL(k_fma_f32, float a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=(float)seed+k+threadIdx.x, asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[k])), (double)a[k])
KERNEL(k_add_u32, int a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=(int)seed+k+threadIdx.x, asm volatile("v_add_u32 %0, %0, %0" : "+v"(a[k])), (double)a[k])
KERNEL(k_mul_lo_u32, int a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=(int)seed+k+threadIdx.x, asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(a[k])), (double)a[k])
KERNEL(k_rcp_f32, float a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=(float)seed+k+threadIdx.x, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k])), (double)a[k])
KERNEL(k_cvt_i32_f32, float a[UNROLL]; int b[UNROLL]; for (int k=0;k<UNROLL;++k) {a[k]=(float)seed+k; b[k]=0;}, asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(b[k]) : "v"(a[k])), (double)b[k])
KERNEL(k_pk_fma_f32, double a[UNROLL]; for (int k=0;k<UNROLL;++k) a[k]=seed+k+threadIdx.x, asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[k])), a[k])

typedef void (*kern_t)(double *, double);

int main() {
    double *out; CHECK(hipMalloc(&out, 4096));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_rcp_f64", k_rcp_f64},
        {"v_ldexp_f64", k_ldexp_f64}, {"v_rndne_f64", k_rndne_f64}, {"v_min_f64", k_min_f64},
        {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32},
        {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cmp_lt_f64", k_cmp_f64}, {"v_fma_f32", k_fma_f32},
        {"v_add_u32", k_add_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_rcp_f32", k_rcp_f32},
        {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // 8 waves per SIMD: 32 waves per CU = 8 blocks of 256
    dim3 grid(cus * 8), block(256);
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    for (auto &kk : ks) {
        hipLaunchKernelGGL(kk.k, grid, block, 0, 0, out, 1.5);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kk.k, grid, block, 0, 0, out, 1.5);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        // per SIMD: 8 waves, each ITERS*UNROLL instrs
        double instr_per_simd = 8.0 * ITERS * UNROLL;
        double ns_per_instr = ms * 1e6 / instr_per_simd;
        printf("%-16s %8.3f ms  %6.2f ns/wave-instr/SIMD  = %5.1f cycles @2.4GHz\n", kk.name, ms, ns_per_instr, ns_per_instr * 2.4);
    }
    return 0;
}
