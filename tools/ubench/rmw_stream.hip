// Micro-benchmark: the ceiling of K1's memory pattern on gfx950 -- an in-place read-modify-write of two float32 volumes
// (T and w, 16 B per 4-voxel pack and array) -- against a plain float4 copy, so that bench.py's `copy_ceiling_GBps` is a
// hand-written figure for THIS access pattern and not a torch copy_.
//
//   map      rows   : a wave owns a 1-KiB run of one z row (lane = consecutive 16-B packs)               -- integrate_depth_kernel
//            brick  : a wave owns a 4 x 4 x 16 brick (lane = zpack + 4 y + 16 x: 64-B row segments),      -- integrate_depth_brick_kernel
//                     four waves of a workgroup behind one another in z, workgroups y-fastest
//            brick2 : 4 x 2 x 32 bricks (128-B segments)
//   policy   plain | nt loads | nt stores | nt both   (__builtin_nontemporal_load / _store = global_load/store ... nt)
//   ballast  N dependent v_fma_f32 per lane in 4 chains between "addresses known" and "values used" (stands in for the
//            projection): issued AFTER the loads (early = loads in flight under the arithmetic) or BEFORE them (late)
//   persist  every wave walks `chunk` consecutive workgroup slots with the next slot's loads in flight while it works on the
//            current one (software pipelining across bricks)
//
// Prints one line per variant: microseconds per sweep and GB/s moved (2 arrays x (read + write)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ f4 ld(const f4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int NT> __device__ __forceinline__ void st(f4 *p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__device__ __forceinline__ float ballast(int n, float seed) {
    float a = seed, b = seed + 1.0f, c = seed + 2.0f, d = seed + 3.0f;
    for (int i = 0; i < n; i += 4) {
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(b));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(c));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d));
    }
    return (a + b) + (c + d);
}

// pack offset (in f4 units) of this lane for workgroup slot `wg` (linear), wave `wv`
template <int MAP>
__device__ __forceinline__ size_t pack_offset(int R, long wg, int wv, int lane) {
    const int zp_per_row = R / 4;
    if (MAP == 0) {                                      // rows: slot = 4 consecutive 1-KiB runs
        return (size_t)wg * 256 + wv * 64 + lane;
    } else {
        constexpr int BY = MAP == 1 ? 4 : 2, BZP = MAP == 1 ? 4 : 8;       // brick y extent, z-packs per brick row
        const int nzg = R / (16 * BZP), nyb = R / BY;                        // workgroups along z (4 bricks each), bricks along y
        const int by = (int)(wg % nyb), bzg = (int)((wg / nyb) % nzg), bx = (int)(wg / ((long)nyb * nzg));   // y fastest
        const int x = 4 * bx + lane / (BY * BZP), y = BY * by + (lane / BZP) % BY, zp = (4 * bzg + wv) * BZP + lane % BZP;
        return ((size_t)x * R + y) * zp_per_row + zp;
    }
}

template <int MAP, int LDNT, int STNT, int EARLY>
__global__ __launch_bounds__(256) void rmw_kernel(f4 *T, f4 *W, int R, int nfma, float seed) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    size_t off = pack_offset<MAP>(R, (long)blockIdx.y * gridDim.x + blockIdx.x, wv, lane);
    float m;
    f4 t, w;
    if (EARLY) {
        t = ld<LDNT>(T + off); w = ld<LDNT>(W + off);
        m = ballast(nfma, seed);
    } else {
        m = ballast(nfma, seed);
        asm volatile("" : "+v"(off) : "v"(m));          // the loads may not move above the arithmetic
        t = ld<LDNT>(T + off); w = ld<LDNT>(W + off);
    }
    const f4 d = w + 1.0f;
    t = (t * w + m) / d;
    w = __builtin_elementwise_min(d, (f4)(100.0f));
    st<STNT>(T + off, t); st<STNT>(W + off, w);
}

// persistent: gridDim.x workgroups, each walks slots blockIdx.x * chunk ... + chunk - 1, next slot's loads in flight
template <int MAP, int LDNT, int STNT>
__global__ __launch_bounds__(256) void rmw_persist_kernel(f4 *T, f4 *W, int R, int nfma, float seed, int chunk) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long s0 = (long)blockIdx.x * chunk;
    size_t off = pack_offset<MAP>(R, s0, wv, lane);
    f4 t = ld<LDNT>(T + off), w = ld<LDNT>(W + off);
    for (int i = 0; i < chunk; ++i) {
        size_t off_n = off;
        f4 tn = t, wn = w;
        if (i + 1 < chunk) {
            off_n = pack_offset<MAP>(R, s0 + i + 1, wv, lane);
            tn = ld<LDNT>(T + off_n); wn = ld<LDNT>(W + off_n);
        }
        const float m = ballast(nfma, seed);
        const f4 d = w + 1.0f;
        t = (t * w + m) / d;
        w = __builtin_elementwise_min(d, (f4)(100.0f));
        st<STNT>(T + off, t); st<STNT>(W + off, w);
        off = off_n; t = tn; w = wn;
    }
}

template <int LDNT, int STNT>
__global__ __launch_bounds__(256) void copy_kernel(const f4 *a, f4 *b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) st<STNT>(b + i, ld<LDNT>(a + i));
}

template <int LDNT>
__global__ __launch_bounds__(256) void read_kernel(const f4 *a, f4 *b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 v = ld<LDNT>(a + i);
    if (v.x == 12345.678f && v.y == 1.0f) b[0] = v;
}

template <int STNT>
__global__ __launch_bounds__(256) void write_kernel(f4 *b, size_t n, float s) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) st<STNT>(b + i, (f4)(s));
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { hipEventCreate(&e0); hipEventCreate(&e1); }
    template <typename F> float us(F &&f, int reps = 12, int warm = 3) {
        for (int i = 0; i < warm; ++i) f();
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) f();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        return ms * 1000.0f / reps;
    }
};

int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 512;
    const bool quick = argc > 2 && !strcmp(argv[2], "quick");
    const bool ceiling = argc > 2 && !strcmp(argv[2], "ceiling");     // one JSON line for bench.py: the ceilings of K1's patterns
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const size_t nvox = (size_t)R * R * R, npack = nvox / 4;
    f4 *T, *W;
    CHECK(hipMalloc(&T, nvox * 4)); CHECK(hipMalloc(&W, nvox * 4));
    CHECK(hipMemset(T, 0, nvox * 4)); CHECK(hipMemset(W, 0, nvox * 4));
    Timer tm;
    if (ceiling) {
        const unsigned nbc = (unsigned)(npack / 256);
        const double mv = 16.0 * nvox;
        auto gbs = [&](float us, double bytes) { return bytes / us / 1e3; };
        const float c_plain = tm.us([&] { hipLaunchKernelGGL((copy_kernel<0, 0>), dim3(nbc), dim3(256), 0, 0, T, W, npack); });
        const float c_nt = tm.us([&] { hipLaunchKernelGGL((copy_kernel<1, 1>), dim3(nbc), dim3(256), 0, 0, T, W, npack); });
        CHECK(hipMemset(T, 0, nvox * 4)); CHECK(hipMemset(W, 0, nvox * 4));
        const float r_plain = tm.us([&] { hipLaunchKernelGGL((rmw_kernel<0, 0, 0, 1>), dim3(nbc), dim3(256), 0, 0, T, W, R, 0, 0.5f); });
        const float r_nt = tm.us([&] { hipLaunchKernelGGL((rmw_kernel<0, 1, 1, 1>), dim3(nbc), dim3(256), 0, 0, T, W, R, 0, 0.5f); });
        const float b_plain = tm.us([&] { hipLaunchKernelGGL((rmw_kernel<2, 0, 0, 1>), dim3(nbc), dim3(256), 0, 0, T, W, R, 0, 0.5f); });
        const float b_nt = tm.us([&] { hipLaunchKernelGGL((rmw_kernel<2, 1, 1, 1>), dim3(nbc), dim3(256), 0, 0, T, W, R, 0, 0.5f); });
        printf("{\"grid\": %d, \"bytes_per_sweep\": %.0f, \"copy_float4_GBps\": %.0f, \"copy_float4_nt_GBps\": %.0f, "
               "\"rmw_rows_GBps\": %.0f, \"rmw_rows_nt_GBps\": %.0f, \"rmw_bricks_4x2x32_GBps\": %.0f, \"rmw_bricks_4x2x32_nt_GBps\": %.0f}\n",
               R, mv, gbs(c_plain, 8.0 * nvox), gbs(c_nt, 8.0 * nvox), gbs(r_plain, mv), gbs(r_nt, mv), gbs(b_plain, mv), gbs(b_nt, mv));
        return 0;
    }
    const double moved = 16.0 * nvox;                    // bytes per RMW sweep = bytes of a copy of one array into the other x 2
    printf("# device %s, %d CUs; grid %d^3: T + w = %.0f MB, a sweep moves %.0f MB\n", prop.name, prop.multiProcessorCount, R, 8.0 * nvox / 1e6, moved / 1e6);
    auto line = [&](const char *name, float us, double bytes) { printf("%-58s %8.1f us  %7.0f GB/s\n", name, us, bytes / us / 1e3); fflush(stdout); };
    const unsigned nb = (unsigned)(npack / 256);
    // -- plain streams (one array = half a sweep's bytes each way)
    line("copy T -> w, float4, plain", tm.us([&] { hipLaunchKernelGGL((copy_kernel<0, 0>), dim3(nb), dim3(256), 0, 0, T, W, npack); }), 8.0 * nvox);
    line("copy T -> w, float4, nt loads", tm.us([&] { hipLaunchKernelGGL((copy_kernel<1, 0>), dim3(nb), dim3(256), 0, 0, T, W, npack); }), 8.0 * nvox);
    line("copy T -> w, float4, nt stores", tm.us([&] { hipLaunchKernelGGL((copy_kernel<0, 1>), dim3(nb), dim3(256), 0, 0, T, W, npack); }), 8.0 * nvox);
    line("copy T -> w, float4, nt both", tm.us([&] { hipLaunchKernelGGL((copy_kernel<1, 1>), dim3(nb), dim3(256), 0, 0, T, W, npack); }), 8.0 * nvox);
    line("read T, float4, plain", tm.us([&] { hipLaunchKernelGGL((read_kernel<0>), dim3(nb), dim3(256), 0, 0, T, W, npack); }), 4.0 * nvox);
    line("read T, float4, nt", tm.us([&] { hipLaunchKernelGGL((read_kernel<1>), dim3(nb), dim3(256), 0, 0, T, W, npack); }), 4.0 * nvox);
    line("write T, float4, plain", tm.us([&] { hipLaunchKernelGGL((write_kernel<0>), dim3(nb), dim3(256), 0, 0, T, npack, 1.0f); }), 4.0 * nvox);
    line("write T, float4, nt", tm.us([&] { hipLaunchKernelGGL((write_kernel<1>), dim3(nb), dim3(256), 0, 0, T, npack, 1.0f); }), 4.0 * nvox);
    CHECK(hipMemset(T, 0, nvox * 4)); CHECK(hipMemset(W, 0, nvox * 4));

    const dim3 grid(nb), block(256);
    char name[160];
    const int fmas_full[] = {0, 64, 128, 256, 384, 512};
    const int fmas_quick[] = {0, 256};
    const int *fmas = quick ? fmas_quick : fmas_full;
    const int nf = quick ? 2 : 6;
#define RUN(MAP, MN, L, S, E) do { for (int fi = 0; fi < nf; ++fi) { const int n = fmas[fi]; \
        snprintf(name, sizeof name, "rmw %-6s ld %-5s st %-5s %-5s fma/lane %3d", MN, L ? "nt" : "plain", S ? "nt" : "plain", E ? "early" : "late", n); \
        line(name, tm.us([&] { hipLaunchKernelGGL((rmw_kernel<MAP, L, S, E>), grid, block, 0, 0, T, W, R, n, 0.5f); }), moved); } } while (0)
    RUN(0, "rows", 0, 0, 1); RUN(0, "rows", 0, 0, 0);
    RUN(1, "brick", 0, 0, 1); RUN(1, "brick", 0, 0, 0);
    RUN(0, "rows", 1, 1, 1); RUN(1, "brick", 1, 1, 1);
    RUN(0, "rows", 1, 0, 1); RUN(1, "brick", 1, 0, 1);
    RUN(0, "rows", 0, 1, 1); RUN(1, "brick", 0, 1, 1);
    RUN(2, "brick2", 0, 0, 1); RUN(2, "brick2", 1, 1, 1);
#undef RUN
    const int chunks[] = {2, 4, 8, 16};
#define RUNP(MAP, MN, L, S) do { for (int ci = 0; ci < 4; ++ci) for (int fi = 0; fi < nf; ++fi) { const int n = fmas[fi], ch = chunks[ci]; \
        if (nb % ch) continue; \
        snprintf(name, sizeof name, "rmw %-6s ld %-5s st %-5s persist chunk %2d fma/lane %3d", MN, L ? "nt" : "plain", S ? "nt" : "plain", ch, n); \
        line(name, tm.us([&] { hipLaunchKernelGGL((rmw_persist_kernel<MAP, L, S>), dim3(nb / ch), block, 0, 0, T, W, R, n, 0.5f, ch); }), moved); } } while (0)
    RUNP(0, "rows", 0, 0); RUNP(1, "brick", 0, 0); RUNP(1, "brick", 1, 1);
#undef RUNP
    return 0;
}
