// K1  depth map -> TSDF integration: FusionDM.fuseDepths CPU semantics
// (reference core/fusion_dm.py:180-217), one sweep over the axis-0 planes [x0,x1).
//
// Mapping: volumes are [x][y][z] with z fastest, so consecutive lanes take consecutive
// z-packs (VEC voxels = one 16-byte fp32 access) of one (x,y) row: every wave reads and
// writes whole 1-KiB lines of T and w, and only rows that are actually updated are touched.
//
// Arithmetic.  The reference decides everything in float64: visible iff 0<=u<W-1 etc.,
// pixel = round-half-even(u), update iff sd > -tdist.  Two per-voxel evaluators:
//   exact_voxel() -- IEEE fp64, the reference's operation order (library is built with
//                    -ffp-contract=off): bit-identical to the fp64 CPU path.
//   the FAST path -- for fp32 volumes: index -> (p0,p1,p2,l2) through one host-folded affine
//                    map (one FMA per component and voxel), one v_rcp_f64 + one Newton step
//                    (measured 2^-48.7 relative, profiles/ubench_r1.txt) per PACK of four voxels
//                    -- the reciprocal of the product of their four depths, times the products
//                    of the others: three more roundings -- instead of two IEEE divisions per
//                    voxel, the pixel / frustum decisions in 2^-20-pixel fixed point on
//                    the int32 ALU, and the sd > -tdist decision on float32 differences.
//                    Error budget: < 2^-19 px for u,v; < 2.4e-7*max(|l2|,|z|) m for the
//                    margin.  Whenever a voxel is closer than a guard band (8*2^-20 px;
//                    1e-5 + 2e-6*|l2| m) to ANY decision boundary it is re-evaluated with
//                    exact_voxel(), so the masks are still bit-identical to the fp64 path
//                    (tests assert this on every voxel).  The running average is evaluated
//                    as (T*w + m/scale)/(1+w) in float32 (FMA + corrected reciprocal,
//                    <= 2 ulp), the same quotient as the reference's
//                    (scale*T*w + m)/(scale*(1+w)).
// fp64 volumes always use exact_voxel() and IEEE fp64 division for the running average.
#include "dfh_common.h"

#include <cstdlib>

namespace dfh {

template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
    T v[N];
};

struct IntegrateParams {
    Mat3 K, Kinv;
    Mat34 lw;
    double scale, cx, cy, cz, half, tdist, wmax;
    // FAST path: component r of (2^20*p0, 2^20*p1, p2, l2) = Ax[r]*x + Ay[r]*y + Az[r]*z + Ac[r]
    double Ax[4], Ay[4], Az[4], Ac[4];
    double inv_scale;
    float tdist_f, ts_f, wmax_f;   // tdist, tdist/scale, wmax rounded to float32
    int zp_shift;      // log2(zpacks) if zpacks is a power of two, else -1
    int planes_per_block;   // each block sweeps this many consecutive x planes (tuning knob)
    int X, Y, Z;       // global grid dims
    int x0, nx;        // slab: planes [x0, x0+nx)
    int H, W;
    int zpacks;        // ceil(Z / VEC)
    // brick kernels: max-depth pyramid of this view (levels 1..kPyrLevels, level l = max of z = -depth over 2^l x 2^l
    // pixel cells, invalid pixels count as 0), NULL = no occlusion culling
    const float *pyr;
    int pyr_off[6], pyr_w[6], pyr_h[6];     // per level (index 0 unused)
    int cull;          // 1: brick culling enabled
};

constexpr int kFixShift = 20;                   // pixel coordinates in 2^-20 px fixed point
constexpr int kFixOne = 1 << kFixShift;
constexpr int kFixHalf = 1 << (kFixShift - 1);
constexpr int kFixBand = 8;                     // guard band around multiples of 0.5 px
constexpr int kFastMaxDim = 2048;               // (dim-1) << 20 must fit in int32
// brick = kBrX x kBrY x kBrZ voxels = one wave (64 lanes x 4 voxels); kBrZ / 4 z-packs per row, kBrX * kBrY rows
#ifndef DFH_BRICK_Y
#define DFH_BRICK_Y 4
#endif
constexpr int kBrX = 4, kBrY = DFH_BRICK_Y, kBrZ = 256 / (kBrX * kBrY);      // 4 x 4 x 16 (shipped) or 4 x 2 x 32
constexpr int kBrZP = kBrZ / 4;                  // z-packs per row of a brick
static_assert(kBrX * kBrY * kBrZP == 64, "a brick is one wave");
constexpr int kPyrLevels = 5;                   // pyramid levels 1..5 (cells of 2..32 pixels)
constexpr int kCullMargin = 1 << 10;            // 2^-10 px: corner projections are good to 2^-19 px, the reference's to 1e-12 px
constexpr long kEarlyRowsMaxVoxels = 1L << 23;  // slabs up to half of 256^3 take the early-load row sweep (integrate_depth_rows_early_kernel);
                                                // at 256^3 it is no faster and moves 213 MB instead of 176 MB (PMC, profiles/r2c_summary.json)
constexpr long kTargetBlocks = 1L << 40;        // measured (profiles/kbench_r1.txt): one plane per block is
                                                // fastest at 256^3 and 512^3; the plane loop stays as a knob

// The reference's per-voxel chain, fusion_dm.py:191-203, in its own operation order.
template <typename DepthT, bool PINHOLE>
__device__ __forceinline__ bool exact_voxel(const IntegrateParams &p, const DepthT *__restrict__ depth,
                                            int x, int y, int z, double &sd_out) {
    const double *lw = p.lw.m;
    const double px = p.scale * ((double)x - p.half) + p.cx;       // :191
    const double py = p.scale * ((double)y - p.half) + p.cy;
    const double pz = p.scale * ((double)z - p.half) + p.cz;
    const double l0 = ((lw[0] * px + lw[1] * py) + lw[2] * pz) + lw[3];     // :193
    const double l1 = ((lw[4] * px + lw[5] * py) + lw[6] * pz) + lw[7];
    const double l2 = ((lw[8] * px + lw[9] * py) + lw[10] * pz) + lw[11];
    double p0, p1, p2;
    if (PINHOLE) {          // K = [[fx,0,cx],[0,fy,cy],[0,0,1]]: the dropped terms are exact zeros
        p0 = p.K.m[0] * l0 + p.K.m[2] * l2;
        p1 = p.K.m[4] * l1 + p.K.m[5] * l2;
        p2 = l2;
    } else {
        p0 = (p.K.m[0] * l0 + p.K.m[1] * l1) + p.K.m[2] * l2;
        p1 = (p.K.m[3] * l0 + p.K.m[4] * l1) + p.K.m[5] * l2;
        p2 = (p.K.m[6] * l0 + p.K.m[7] * l1) + p.K.m[8] * l2;
    }
    sd_out = 0.0;
    if (!(p2 != 0.0)) return false;                                 // util.py:318
    const double u = p0 / p2;
    const double v = p1 / p2;
    if (!((u >= 0.0) && (u < (double)(p.W - 1)) && (v >= 0.0) && (v < (double)(p.H - 1)))) return false;  // :195
    const int ui = (int)rint(u);                                    // Python round(): half to even (:196)
    const int vi = (int)rint(v);
    const double zd = -1.0 * (double)depth[(size_t)vi * p.W + ui];
    if (!(zd > 0.0)) return false;                                  // :197
    double cz;
    if (PINHOLE) {
        cz = zd;                                                    // Kinv row 2 == [0,0,1]
    } else {
        cz = (p.Kinv.m[6] * (zd * u) + p.Kinv.m[7] * (zd * v)) + p.Kinv.m[8] * (zd * 1.0);
    }
    const double sd = cz - l2;                                      // :201
    sd_out = sd;
    return sd > -1.0 * p.tdist;                                     // :203
}

// Rare-path wrapper: the coordinates are laundered through empty asm so the compiler cannot
// hoist the exact chain's loop-invariant arithmetic out of the (almost never taken) branch -- and the parameters come
// through a laundered POINTER, so that their loads stay inside the branch too: read through the by-value kernel argument, the
// exact chain's ~30 doubles (K, Kinv, lw, ...) were fetched into scalar registers at the top of the kernel, beside the fast
// path's own 24-32, and the overflow was parked in vector lanes (25 v_writelane + the read-backs per wave, ~10 % of what a
// wave issues).
// (LAZY = false, the multi-view kernels: their parameters are read from the workspace through scalar loads either way and fit
// the scalar registers; with a second, laundered pointer to the same structures the compiler turned the FAST path's scalar
// loads into vector loads -- 512^3 x 8 views 743 -> 1 169 us.)
template <typename DepthT, bool PINHOLE, bool LAZY>
__device__ __forceinline__ bool exact_voxel_rare(const IntegrateParams *pp, const DepthT *__restrict__ depth,
                                                 int x, int y, int z, double &sd_out) {
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
    if (LAZY) asm volatile("" : "+s"(pp));
    return exact_voxel<DepthT, PINHOLE>(*pp, depth, x, y, z, sd_out);
}

// The IntegrateParams a kernel received BY VALUE, as a pointer into its kernel-argument segment (byte_offset = what precedes it:
// three pointers in every single-view kernel): for exact_voxel_rare.
__device__ __forceinline__ const IntegrateParams *kernarg_params(int byte_offset) {
    typedef const char __attribute__((address_space(4))) *KernArgBytes;
    KernArgBytes ka = (KernArgBytes)__builtin_amdgcn_kernarg_segment_ptr();
    return (const IntegrateParams *)(ka + byte_offset);
}
constexpr int kParamsAfterThreePointers = 24;

__device__ __forceinline__ double rcp_nr1(double d) {      // relative error <= 2^-48.7 (measured)
    const double r = __builtin_amdgcn_rcp(d);
    return __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
}

// double -> int32 with the HARDWARE's conversion semantics (v_cvt_i32_f64: truncate, saturate at INT32_MIN/MAX, NaN -> 0).
// A C++ cast of an out-of-range or NaN double is undefined behaviour, and the guard-band logic below relies on
// saturation: voxels far outside the frustum or next to the camera plane produce such values.
__device__ __forceinline__ int cvt_i32_sat(double x) {
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ void pack_coords(const IntegrateParams &p, int &y, int &zp) {
    const int lin = blockIdx.x * 256 + threadIdx.x;      // (y, zpack) inside one x plane
    if (p.zp_shift >= 0) {
        y = lin >> p.zp_shift;
        zp = lin & ((1 << p.zp_shift) - 1);
    } else {
        y = lin / p.zpacks;
        zp = lin - y * p.zpacks;
    }
}

// fp32 volumes: filtered fast path with exact fallback (see file header).
// STRIDED: the 4 voxels of a lane are z, z+64, z+128, z+192 of its wave's 256-voxel run instead of
// 4 consecutive ones, so that in every depth gather and every T/w access consecutive lanes touch
// consecutive voxels (a gather instruction then spans ~6-12 cache lines instead of ~50).
//
// view_pack: one depth view's contribution to the VEC voxels (x, y, z0 + j*ZS): ms[j] = min(tdist, sd) / scale and
// upd[j] = the reference's update condition; returns whether any voxel of the pack is updated.
struct NoHook {
    __device__ __forceinline__ void operator()(bool) const {}
};

// after_gathers(any_inside): called right after the depth gathers have been issued and before their values are used --
// the place to issue further independent loads (the brick sweep starts its T / w loads there, one memory round trip
// instead of two); any_inside = some voxel of the pack projects into the image.
template <typename DepthT, int VEC, bool PINHOLE, bool STRIDED, bool LAZY = true, typename Hook = NoHook>
__device__ __forceinline__ bool view_pack(const IntegrateParams &p, const IntegrateParams *p_rare, const DepthT *__restrict__ depth, int x,
                                          int y, int z0, float (&ms)[VEC], bool (&upd)[VEC], Hook &&after_gathers = NoHook()) {
    constexpr int ZS = STRIDED ? 64 : 1;                 // z step between a lane's voxels
    constexpr int NC = PINHOLE ? 3 : 4;
    const unsigned ulim = (unsigned)(p.W - 1) << kFixShift;
    const unsigned vlim = (unsigned)(p.H - 1) << kFixShift;
    double base[NC];
    {
        const double xf = (double)x, yf = (double)y, zf = (double)z0;
#pragma unroll
        for (int r = 0; r < NC; ++r)
            base[r] = __builtin_fma(p.Az[r], zf, __builtin_fma(p.Ax[r], xf, __builtin_fma(p.Ay[r], yf, p.Ac[r])));
    }
    // p2 is affine in z: if both ends of the pack are well away from the camera plane and on
    // the same side, so is everything between; otherwise the whole pack goes the exact way.
    const double p2_last = __builtin_fma(p.Az[2], (double)((VEC - 1) * ZS), base[2]);
    const bool pack_singular = !((fabs(base[2]) > 1e-6) & (fabs(p2_last) > 1e-6) & ((base[2] > 0.0) == (p2_last > 0.0)));

    // phase 1: project the z-pack; frustum membership and pixel in 2^-20 px fixed point
    double l2v[VEC];
    float l2f[VEC];
    unsigned pix[VEC];
    bool inside[VEC], amb[VEC];
    double rinv[VEC];
    if (VEC == 4) {
        // the four reciprocals from ONE v_rcp_f64 (a quarter-rate instruction) of the product of the four depths: 1 / p2_j =
        // (product of the others) / (product of all).  ~2^-48 relative like rcp_nr1 (three more roundings of 2^-53); the pack's
        // depths are of one sign and away from zero, or the whole pack is re-evaluated exactly (pack_singular)
        double d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __builtin_fma(p.Az[2], (double)(j * ZS), base[2]);
        const double d01 = d[0] * d[1], d23 = d[2] * d[3];
        const double rall = rcp_nr1(d01 * d23);
        const double r01 = rall * d23, r23 = rall * d01;               // 1 / (d0 d1), 1 / (d2 d3)
        rinv[0] = r01 * d[1]; rinv[1] = r01 * d[0]; rinv[2 % VEC] = r23 * d[3 % VEC]; rinv[3 % VEC] = r23 * d[2 % VEC];
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) rinv[j] = rcp_nr1(__builtin_fma(p.Az[2], (double)(j * ZS), base[2]));
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const double jf = (double)(j * ZS);
        const double p0 = __builtin_fma(p.Az[0], jf, base[0]);      // 2^20 * p0
        const double p1 = __builtin_fma(p.Az[1], jf, base[1]);      // 2^20 * p1
        const double p2 = __builtin_fma(p.Az[2], jf, base[2]);
        l2v[j] = PINHOLE ? p2 : __builtin_fma(p.Az[NC - 1], jf, base[NC - 1]);
        l2f[j] = (float)l2v[j];
        const double r = rinv[j];
        const int qu = cvt_i32_sat(p0 * r);              // trunc, saturating; NaN -> 0 (inside the band)
        const int qv = cvt_i32_sat(p1 * r);
        // distance to the nearest multiple of 0.5 px: pixel ties AND the integer frustum edges
        const unsigned du = (unsigned)((qu + kFixBand) & (kFixHalf - 1));
        const unsigned dv = (unsigned)((qv + kFixBand) & (kFixHalf - 1));
        amb[j] = pack_singular | ((du < dv ? du : dv) <= 2u * kFixBand);
        inside[j] = ((unsigned)qu < ulim) & ((unsigned)qv < vlim);
        const int ui = (qu + kFixHalf) >> kFixShift;
        const int vi = (qv + kFixHalf) >> kFixShift;
        pix[j] = inside[j] ? (unsigned)(__mul24(vi, p.W) + ui) : 0u;
    }
    // phase 2: every depth gather of the pack in flight together (pixel 0 when outside)
    DepthT dval[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) dval[j] = depth[pix[j]];
    {
        bool any_inside = false;
#pragma unroll
        for (int j = 0; j < VEC; ++j) any_inside = any_inside | inside[j] | amb[j];
        after_gathers(any_inside);
    }
    // phase 3: sd > -tdist on float32 differences; guard-banded voxels re-run exactly
    bool any = false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float zdf = -(float)dval[j];
        const bool hit = inside[j] & (zdf > 0.0f);
        bool ok, redo, freespace;
        if (PINHOLE) {
            const float d32 = zdf - (l2f[j] - p.tdist_f);            // ~ sd + tdist
            const float band = fmaf(fabsf(l2f[j]), 2e-6f, 1e-5f);
            ok = hit & (d32 > 0.0f);
            redo = amb[j] | (hit & (fabsf(d32) <= band));
            freespace = d32 > 2.0f * p.tdist_f + band;               // sd > tdist for certain
        } else {
            ok = hit;
            redo = amb[j];
            freespace = false;
        }
        float m = p.ts_f;
        if (__builtin_expect(redo, 0)) {
            double sd;
            ok = exact_voxel_rare<DepthT, PINHOLE, LAZY>(p_rare, depth, x, y, z0 + j * ZS, sd);
            m = (float)((sd < p.tdist ? sd : p.tdist) * p.inv_scale);
        } else if (ok & !freespace) {
            double cz = -(double)dval[j];
            if (!PINHOLE) {
                // u, v to ~1e-12 px from the folded map; only the value depends on them here
                const double jf = (double)(j * ZS);
                const double r = rcp_nr1(__builtin_fma(p.Az[2], jf, base[2])) * (1.0 / (double)kFixOne);
                const double u = __builtin_fma(p.Az[0], jf, base[0]) * r;
                const double v = __builtin_fma(p.Az[1], jf, base[1]) * r;
                cz = __builtin_fma(p.Kinv.m[6] * cz, u, __builtin_fma(p.Kinv.m[7] * cz, v, p.Kinv.m[8] * cz));
            }
            const double sd = cz - l2v[j];
            if (!PINHOLE) {
                const double margin = sd + p.tdist;
                ok = margin > 0.0;
                if (fabs(margin) < 1e-7) ok = exact_voxel_rare<DepthT, PINHOLE, LAZY>(p_rare, depth, x, y, z0 + j * ZS, cz);
            }
            m = (float)((sd < p.tdist ? sd : p.tdist) * p.inv_scale);
        }
        ok = ok & (z0 + j * ZS < p.Z);
        ms[j] = m;
        upd[j] = ok;
        any = any | ok;
    }
    return any;
}

// T <- (T*w + m/scale)/(1+w);  w <- min(1+w, wmax)          (fusion_dm.py:209-210)
template <int VEC>
__device__ __forceinline__ void apply_pack(Pack<float, VEC> &t, Pack<float, VEC> &w, const float (&ms)[VEC], const bool (&upd)[VEC],
                                           float wmax_f) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float wt = w.v[j];
        const float d = wt + 1.0f;
        const float n = fmaf(t.v[j], wt, ms[j]);
        const float r = __builtin_amdgcn_rcpf(d);
        float q = n * r;
        q = fmaf(fmaf(-d, q, n), r, q);
        t.v[j] = upd[j] ? q : t.v[j];
        w.v[j] = upd[j] ? fminf(d, wmax_f) : wt;
    }
}

template <typename DepthT, int VEC, bool PINHOLE, bool STRIDED>
__global__ __launch_bounds__(256) void integrate_depth_kernel(float *__restrict__ tsdf,
                                                               float *__restrict__ tsdf_w,
                                                               const DepthT *__restrict__ depth,
                                                               const IntegrateParams p) {
    int y, zp;
    pack_coords(p, y, zp);
    if (y >= p.Y) return;
    const int lane = threadIdx.x & 63;
    const int z0 = STRIDED ? (zp - lane) * VEC + lane : zp * VEC;
    constexpr int ZS = STRIDED ? 64 : 1;
    const int xl_end = min(p.nx, (int)(blockIdx.y + 1) * p.planes_per_block);
    for (int xl = blockIdx.y * p.planes_per_block; xl < xl_end; ++xl) {
        float ms[VEC];              // min(tdist, sd) / scale
        bool upd[VEC];
        if (!view_pack<DepthT, VEC, PINHOLE, STRIDED>(p, kernarg_params(kParamsAfterThreePointers), depth, p.x0 + xl, y, z0, ms, upd)) continue;
        const size_t off = ((size_t)xl * p.Y + y) * p.Z + z0;
        using P = Pack<float, VEC>;
        P t, w;
        if (STRIDED) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { t.v[j] = tsdf[off + j * ZS]; w.v[j] = tsdf_w[off + j * ZS]; }
        } else {
            t = *reinterpret_cast<const P *>(tsdf + off);
            w = *reinterpret_cast<const P *>(tsdf_w + off);
        }
        apply_pack<VEC>(t, w, ms, upd, p.wmax_f);
        if (STRIDED) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (upd[j]) { tsdf[off + j * ZS] = t.v[j]; tsdf_w[off + j * ZS] = w.v[j]; }
            }
        } else {
            *reinterpret_cast<P *>(tsdf + off) = t;
            *reinterpret_cast<P *>(tsdf_w + off) = w;
        }
    }
}

// Row sweep for SMALL slabs (one rank's share of a strongly scaled 256^3 grid: a 5-20 us kernel).  Such a launch is bound by its
// waves' chain of dependent memory round trips (gathers -> T/w -> store), not by bytes or VALU, so T/w of every pack that projects
// into the image are requested together with the depth gathers: two round trips instead of three.  (On large grids the extra
// fetches of packs that turn out not to be updated cost bandwidth: measured equal at 256^3, slower at 512^3 -- the plain kernel
// stays there.)  Same arithmetic, same bits.
template <typename DepthT, bool PINHOLE>
__global__ __launch_bounds__(256) void integrate_depth_rows_early_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                          const DepthT *__restrict__ depth, const IntegrateParams p) {
    int y, zp;
    pack_coords(p, y, zp);
    if (y >= p.Y) return;
    const int z0 = zp * 4;
    const int xl = blockIdx.y;
    float ms[4];
    bool upd[4];
    const size_t off = ((size_t)xl * p.Y + y) * p.Z + z0;
    using P = Pack<float, 4>;
    P t, w;
    bool loaded = false;
    const bool any = view_pack<DepthT, 4, PINHOLE, false>(p, kernarg_params(kParamsAfterThreePointers), depth, p.x0 + xl, y, z0, ms, upd, [&](bool any_inside) {
        if (any_inside) {
            t = *reinterpret_cast<const P *>(tsdf + off);
            w = *reinterpret_cast<const P *>(tsdf_w + off);
            loaded = true;
        }
    });
    if (!any) return;
    if (!loaded) {                                        // (cannot happen: any implies any_inside)
        t = *reinterpret_cast<const P *>(tsdf + off);
        w = *reinterpret_cast<const P *>(tsdf_w + off);
    }
    apply_pack<4>(t, w, ms, upd, p.wmax_f);
    *reinterpret_cast<P *>(tsdf + off) = t;
    *reinterpret_cast<P *>(tsdf_w + off) = w;
}

// Several depth views in ONE sweep of the volume (FusionDM.compute_live_tsdf / the initial fusion loop call fuseDepths
// once per view, core/fusion_dm.py:152-154,166-170): a voxel's T and w are read once, take the views' updates in view
// order in registers -- the same float32 operations as consecutive single-view sweeps, hence the same bits -- and are
// written once.  HBM traffic per view drops from 16 B/voxel to 16/V; the projection work per view stays.
constexpr int kMaxViews = 16;
struct ViewPtrs {
    const void *depth[kMaxViews];
    int n;
};

template <typename DepthT, int VEC, bool PINHOLE>
__global__ __launch_bounds__(256) void integrate_depth_multi_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                     const IntegrateParams *__restrict__ views, const ViewPtrs vp) {
    int y, zp;
    pack_coords(views[0], y, zp);                        // grid geometry is the same for every view
    if (y >= views[0].Y) return;
    const int z0 = zp * VEC;
    const int nx = views[0].nx, ppb = views[0].planes_per_block;
    const int xl_end = min(nx, (int)(blockIdx.y + 1) * ppb);
    for (int xl = blockIdx.y * ppb; xl < xl_end; ++xl) {
        using P = Pack<float, VEC>;
        P t, w;
        bool loaded = false;
        const size_t off = ((size_t)xl * views[0].Y + y) * views[0].Z + z0;
        for (int v = 0; v < vp.n; ++v) {
            const IntegrateParams &p = views[v];         // uniform address: scalar loads
            float ms[VEC];
            bool upd[VEC];
            if (!view_pack<DepthT, VEC, PINHOLE, false, false>(p, &p, static_cast<const DepthT *>(vp.depth[v]), p.x0 + xl, y, z0, ms, upd)) continue;
            if (!loaded) {
                t = *reinterpret_cast<const P *>(tsdf + off);
                w = *reinterpret_cast<const P *>(tsdf_w + off);
                loaded = true;
            }
            apply_pack<VEC>(t, w, ms, upd, p.wmax_f);
        }
        if (loaded) {
            *reinterpret_cast<P *>(tsdf + off) = t;
            *reinterpret_cast<P *>(tsdf_w + off) = w;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Brick mapping with conservative culling.
// The row mapping above gives every wave a 1 x 1 x 256 run of voxels: such a run crosses the frustum planes and the
// occlusion boundary almost always, so the ~50 % of the voxels a view cannot update still pay the whole projection and
// gather.  Here a wave owns a compact 4 x 4 x 16 brick (lane = z-pack + 4 y + 16 x; a quarter-wave touches four 64-byte
// row segments, the four waves of a workgroup sit behind one another in z, so every 256-byte stretch of a row is
// consumed by one workgroup).  Voxel centres of a brick span a box; an affine map followed by the perspective divide
// keeps the image of a box in front of the camera inside the bounding rectangle of its eight projected corners.  So:
//   * all corners in front of the camera and the rectangle entirely outside [0, W-1) x [0, H-1) (with a 2^-10 px
//     margin, the corners being good to 2^-19 px): no voxel is visible -> the wave skips the view;
//   * (pinhole) z_max = the largest valid depth over the pixels the rectangle can round to, from a max-pyramid of the
//     depth map, and z_max + tdist (+ margin) <= the smallest corner depth: every voxel has sd <= -tdist or no depth at
//     all -> nothing is updated -> skip.
// Skipping is only ever done when NO voxel of the brick would be updated, so the result is bit-identical to the row
// kernels (tests compare every voxel).  Bricks that survive run view_pack exactly as before.
struct BrickGeom {
    int nzg;           // workgroups along z (64 voxels each)
    int nyb;           // bricks along y
    int nxb;           // bricks along x (slab)
    int order;         // launch order of the workgroups (tuning): 0 = z fastest, 1 = y fastest, 2 = one contiguous x range per XCD
};

// true when `p`'s view provably updates no voxel of the brick whose first voxel is (x0, y0, z0) (global indices)
template <bool PINHOLE>
__device__ __forceinline__ bool brick_culled(const IntegrateParams &p, int x0, int y0, int z0) {
    constexpr int NC = PINHOLE ? 3 : 4;
    double base[NC];
    {
        const double xf = (double)x0, yf = (double)y0, zf = (double)z0;
#pragma unroll
        for (int r = 0; r < NC; ++r) base[r] = __builtin_fma(p.Az[r], zf, __builtin_fma(p.Ax[r], xf, __builtin_fma(p.Ay[r], yf, p.Ac[r])));
    }
    int umin = 0x7fffffff, umax = (int)0x80000000, vmin = 0x7fffffff, vmax = (int)0x80000000;
    double l2min = __builtin_huge_val();
    bool front = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {                        // the affine map is linear in the corner offsets
        double q[NC];
#pragma unroll
        for (int r = 0; r < NC; ++r)
            q[r] = base[r] + (((c & 1) ? (double)(kBrX - 1) * p.Ax[r] : 0.0) + ((c & 2) ? (double)(kBrY - 1) * p.Ay[r] : 0.0) +
                              ((c & 4) ? (double)(kBrZ - 1) * p.Az[r] : 0.0));
        front = front && q[2] > 1e-6;
        const double rr = rcp_nr1(q[2]);
        const int qu = cvt_i32_sat(q[0] * rr), qv = cvt_i32_sat(q[1] * rr);
        umin = min(umin, qu); umax = max(umax, qu); vmin = min(vmin, qv); vmax = max(vmax, qv);
        l2min = fmin(l2min, q[NC - 1]);
    }
    if (!front) return false;                            // a corner at or behind the camera plane: no claim
    const int ulim = (p.W - 1) << kFixShift, vlim = (p.H - 1) << kFixShift;
    // (the corner sums above differ from view_pack's FMA chain by ~1e-13 relative: far inside the 2^-10 px margin)
    if (umax < -kCullMargin || umin > ulim + kCullMargin || vmax < -kCullMargin || vmin > vlim + kCullMargin) return true;
    if (!PINHOLE || p.pyr == nullptr) return false;
    // pixels the voxels can round to: [floor(umin), ceil(umax)] x [floor(vmin), ceil(vmax)], clipped to the image
    // (corner coordinates may be saturated at INT32_MIN / MAX: clamp to just outside the image BEFORE any arithmetic)
    umin = max(umin, -2 * kFixOne); vmin = max(vmin, -2 * kFixOne);
    umax = min(umax, ulim + 2 * kFixOne); vmax = min(vmax, vlim + 2 * kFixOne);
    int px0 = (umin - kCullMargin) >> kFixShift, px1 = (umax + kCullMargin + kFixOne - 1) >> kFixShift;
    int py0 = (vmin - kCullMargin) >> kFixShift, py1 = (vmax + kCullMargin + kFixOne - 1) >> kFixShift;
    px0 = max(px0, 0); py0 = max(py0, 0); px1 = min(px1, p.W - 1); py1 = min(py1, p.H - 1);
    if (px1 < px0 || py1 < py0) return false;            // (cannot happen after the rectangle test: no claim)
    int L = 1;
    while (L <= kPyrLevels && (((px1 >> L) - (px0 >> L)) > 1 || ((py1 >> L) - (py0 >> L)) > 1)) ++L;
    if (L > kPyrLevels) return false;                    // footprint wider than two 32-pixel cells: no claim
    const float *lv = p.pyr + p.pyr_off[L];
    const int wl = p.pyr_w[L];
    const int cx0 = px0 >> L, cx1 = px1 >> L, cy0 = py0 >> L, cy1 = py1 >> L;
    const float zmax = fmaxf(fmaxf(lv[cy0 * wl + cx0], lv[cy0 * wl + cx1]), fmaxf(lv[cy1 * wl + cx0], lv[cy1 * wl + cx1]));
    // sd = z - l2 <= zmax - l2min for every voxel with a valid pixel; the margin covers the float32 rounding of l2min
    const float l2f = (float)l2min;
    return zmax + p.tdist_f + fmaf(fabsf(l2f), 4e-7f, 1e-5f) <= l2f;
}

// Classification pass: one THREAD per brick, bit v of mask[brick] = view v may update a voxel of the brick.  Bricks are
// numbered ((bx * nyb + by) * nzg + bzg) * 4 + wave, i.e. the four bricks of one sweep workgroup are consecutive.
template <bool PINHOLE, bool BYVAL>
__global__ __launch_bounds__(256) void brick_classify_kernel(const IntegrateParams *__restrict__ views, const IntegrateParams p1,
                                                              int n_views, const BrickGeom g, int n_bricks,
                                                              unsigned short *__restrict__ mask) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_bricks) return;
    const int wv = b & 3;
    int t = b >> 2;
    const int bzg = t % g.nzg; t /= g.nzg;
    const int by = t % g.nyb;
    const int bx = t / g.nyb;
    const int z0 = 4 * kBrZ * bzg + kBrZ * wv;
    unsigned m = 0;
    if (z0 < p1.Z) {
        for (int v = 0; v < n_views; ++v) {
            const IntegrateParams &p = BYVAL ? p1 : views[v];
            if (!(p.cull && brick_culled<PINHOLE>(p, p.x0 + kBrX * bx, kBrY * by, z0))) m |= 1u << v;
        }
    }
    mask[b] = (unsigned short)m;
}

__device__ __forceinline__ void brick_coords(const IntegrateParams &p, const BrickGeom g, int &xl, int &y, int &z0, int &brick) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bzg, by, bx;
    const long lin = (long)blockIdx.y * gridDim.x + blockIdx.x;
    if (g.order == 0) {                                   // z fastest, then y, then x
        bzg = (int)(lin % g.nzg); by = (int)((lin / g.nzg) % g.nyb); bx = (int)(lin / ((long)g.nzg * g.nyb));
    } else if (g.order == 1) {                            // y fastest, then z, then x
        by = (int)(lin % g.nyb); bzg = (int)((lin / g.nyb) % g.nzg); bx = (int)(lin / ((long)g.nzg * g.nyb));
    } else if (g.order == 3) {                            // x fastest, then y, then z
        bx = (int)(lin % g.nxb); by = (int)((lin / g.nxb) % g.nyb); bzg = (int)(lin / ((long)g.nxb * g.nyb));
    } else if (g.order == 4) {                            // y fastest with a diagonal z
        by = (int)(lin % g.nyb); bzg = (int)((lin / g.nyb + by) % g.nzg); bx = (int)(lin / ((long)g.nzg * g.nyb));
    } else if (g.order == 5) {                            // y fastest, then x, then z
        by = (int)(lin % g.nyb); bx = (int)((lin / g.nyb) % g.nxb); bzg = (int)(lin / ((long)g.nxb * g.nyb));
    } else {
        // workgroups are dealt round-robin to the 8 XCDs: give XCD k the k-th eighth of the (x, y, z) ordered bricks
        const long total = (long)gridDim.x * gridDim.y;
        const long per = (total + 7) / 8;
        const long logical = (lin & 7) * per + (lin >> 3);      // (total % 8 == 0, checked on the host)
        bzg = (int)(logical % g.nzg); by = (int)((logical / g.nzg) % g.nyb); bx = (int)(logical / ((long)g.nzg * g.nyb));
    }
    brick = (((bx * g.nyb + by) * g.nzg) + bzg) * 4 + wv;
    xl = kBrX * bx + lane / (kBrY * kBrZP);               // slab-local plane
    y = kBrY * by + (lane / kBrZP) % kBrY;
    z0 = 4 * kBrZ * bzg + kBrZ * wv + 4 * (lane % kBrZP);
}

// One view.  mask == NULL: every brick is swept (no classification pass ran).
template <typename DepthT, bool PINHOLE, bool EARLY>
__global__ __launch_bounds__(256) void integrate_depth_brick_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                     const DepthT *__restrict__ depth, const IntegrateParams p,
                                                                     const BrickGeom g, const unsigned short *__restrict__ mask) {
    int xl, y, z0, brick;
    brick_coords(p, g, xl, y, z0, brick);
    if (mask && __builtin_amdgcn_readfirstlane((int)mask[__builtin_amdgcn_readfirstlane(brick)]) == 0) return;   // (wave-uniform)
    const bool in_grid = xl < p.nx && y < p.Y && z0 < p.Z;
    float ms[4];
    bool upd[4];
    const size_t off = ((size_t)xl * p.Y + y) * p.Z + z0;
    using P = Pack<float, 4>;
    P t, w;
    bool loaded = false;
    // T / w of a pack that projects into the image are requested together with its depth gathers: in the bricks that
    // survive the classification nearly every such pack is updated, so little is fetched in vain and the sweep pays one
    // memory round trip instead of two
    const bool any = view_pack<DepthT, 4, PINHOLE, false>(p, kernarg_params(kParamsAfterThreePointers), depth, p.x0 + xl, y, z0, ms, upd, [&](bool any_inside) {
        if (EARLY && any_inside && in_grid) {
            t = *reinterpret_cast<const P *>(tsdf + off);
            w = *reinterpret_cast<const P *>(tsdf_w + off);
            loaded = true;
        }
    });
    if (!(any && in_grid)) return;
    if (!loaded) {                                        // (late variant; early: cannot happen, any implies any_inside)
        t = *reinterpret_cast<const P *>(tsdf + off);
        w = *reinterpret_cast<const P *>(tsdf_w + off);
    }
    apply_pack<4>(t, w, ms, upd, p.wmax_f);
    *reinterpret_cast<P *>(tsdf + off) = t;
    *reinterpret_cast<P *>(tsdf_w + off) = w;
}

// FRESH: the volume is taken to be (fresh_t, 0) everywhere -- a live volume that starts from np.zeros + tdist / np.zeros
// (core/fusion_dm.py:152-153) -- so nothing is loaded and EVERY pack is written: the fill and the sweep in one pass over the
// volume (one write of it instead of a write, a read of the updated part and another write).
template <typename DepthT, bool PINHOLE, bool FRESH>
__global__ __launch_bounds__(256) void integrate_depth_multi_brick_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                           const IntegrateParams *__restrict__ views, const ViewPtrs vp,
                                                                           const BrickGeom g, const unsigned short *__restrict__ mask,
                                                                           const float fresh_t) {
    int xl, y, z0, brick;
    brick_coords(views[0], g, xl, y, z0, brick);
    unsigned m = mask ? (unsigned)__builtin_amdgcn_readfirstlane((int)mask[__builtin_amdgcn_readfirstlane(brick)]) : (1u << vp.n) - 1u;
    const bool in_grid = xl < views[0].nx && y < views[0].Y && z0 < views[0].Z;
    using P = Pack<float, 4>;
    const size_t off = ((size_t)xl * views[0].Y + y) * views[0].Z + z0;
    P t, w;
    if (FRESH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { t.v[j] = fresh_t; w.v[j] = 0.0f; }
    }
    if (m == 0) {
        if (FRESH && in_grid) {
            *reinterpret_cast<P *>(tsdf + off) = t;
            *reinterpret_cast<P *>(tsdf_w + off) = w;
        }
        return;
    }
    bool loaded = FRESH;
    while (m) {                                           // views in ascending order: the order of consecutive sweeps
        const int v = __builtin_ctz(m);
        m &= m - 1;
        const IntegrateParams &p = views[v];             // uniform address: scalar loads
        float ms[4];
        bool upd[4];
        const bool any = view_pack<DepthT, 4, PINHOLE, false, false>(p, &p, static_cast<const DepthT *>(vp.depth[v]), p.x0 + xl, y, z0, ms, upd);
        if (!(any && in_grid)) continue;
        if (!loaded) {
            t = *reinterpret_cast<const P *>(tsdf + off);
            w = *reinterpret_cast<const P *>(tsdf_w + off);
            loaded = true;
        }
        apply_pack<4>(t, w, ms, upd, p.wmax_f);
    }
    if (loaded && in_grid) {
        *reinterpret_cast<P *>(tsdf + off) = t;
        *reinterpret_cast<P *>(tsdf_w + off) = w;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fill_pair_kernel(T *__restrict__ a, T va, T *__restrict__ b, T vb, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { a[i] = va; b[i] = vb; }
}

// Parameters of up to kParamChunk views from the kernel-argument segment into device memory (see dfh_integrate_depth_multi)
constexpr int kParamChunk = 6;
struct ParamChunk {
    IntegrateParams v[kParamChunk];
};
static_assert(sizeof(IntegrateParams) % 8 == 0 && sizeof(ParamChunk) + 24 <= 4096, "the chunk must fit the kernel-argument segment");
__global__ __launch_bounds__(256) void upload_params_kernel(IntegrateParams *dst, const ParamChunk c, int n) {
    // (read through the segment pointer: indexing the by-value struct dynamically would copy it to scratch memory first)
    typedef const unsigned long long __attribute__((address_space(4))) *KernArgWords;
    KernArgWords ka = (KernArgWords)__builtin_amdgcn_kernarg_segment_ptr() + 1;                                          // behind `dst`
    unsigned long long *out = reinterpret_cast<unsigned long long *>(dst);
    const int words = n * (int)(sizeof(IntegrateParams) / 8);
    for (int i = threadIdx.x; i < words; i += 256) out[i] = ka[i];
    (void)c;
}

// Max-depth pyramid of up to kMaxViews depth maps: blockIdx.z = view, one workgroup per 32 x 32 pixel tile, levels 1..5.
struct PyrViews {
    const void *depth[kMaxViews];
    float *pyr[kMaxViews];
};

template <typename DepthT>
__global__ __launch_bounds__(256) void depth_pyramid_kernel(const PyrViews pv, int H, int W, const IntegrateParams p1) {
    // level geometry is the same for every view: taken from p1 (kernel argument)
    const DepthT *__restrict__ depth = static_cast<const DepthT *>(pv.depth[blockIdx.z]);
    float *__restrict__ pyr = pv.pyr[blockIdx.z];
    __shared__ float s[256];
    const int t = threadIdx.x;
    const int qx = t & 15, qy = t >> 4;
    const int px = 32 * (int)blockIdx.x + 2 * qx, py = 32 * (int)blockIdx.y + 2 * qy;
    auto zval = [&](int x, int y) {
        if (x >= W || y >= H) return 0.0f;
        const float z = -(float)depth[(size_t)y * W + x];
        return z > 0.0f ? z : 0.0f;                       // invalid (0, NaN) pixels never update a voxel
    };
    float m = fmaxf(fmaxf(zval(px, py), zval(px + 1, py)), fmaxf(zval(px, py + 1), zval(px + 1, py + 1)));
    {
        const int cx = 16 * (int)blockIdx.x + qx, cy = 16 * (int)blockIdx.y + qy;
        if (cx < p1.pyr_w[1] && cy < p1.pyr_h[1]) pyr[p1.pyr_off[1] + cy * p1.pyr_w[1] + cx] = m;
    }
    s[t] = m;
    __syncthreads();
    int n = 16;                                           // side of the level held in s[]
#pragma unroll
    for (int L = 2; L <= kPyrLevels; ++L) {
        const int h = n >> 1;
        float v = 0.0f;
        const int x = t % h, y = t / h;
        if (t < h * h) v = fmaxf(fmaxf(s[(2 * y) * n + 2 * x], s[(2 * y) * n + 2 * x + 1]), fmaxf(s[(2 * y + 1) * n + 2 * x], s[(2 * y + 1) * n + 2 * x + 1]));
        __syncthreads();
        if (t < h * h) {
            s[y * h + x] = v;
            const int cx = h * (int)blockIdx.x + x, cy = h * (int)blockIdx.y + y;
            if (cx < p1.pyr_w[L] && cy < p1.pyr_h[L]) pyr[p1.pyr_off[L] + cy * p1.pyr_w[L] + cx] = v;
        }
        __syncthreads();
        n = h;
    }
}

static size_t pyramid_floats(int H, int W, int *off, int *pw, int *ph) {
    size_t tot = 0;
    for (int L = 1; L <= kPyrLevels; ++L) {
        const int wl = (W + (1 << L) - 1) >> L, hl = (H + (1 << L) - 1) >> L;
        if (off) { off[L] = (int)tot; pw[L] = wl; ph[L] = hl; }
        tot += (size_t)wl * hl;
    }
    return (tot + 3) / 4 * 4;                             // whole 16-byte units
}

// A2 (optional): the arithmetic of the reference's OpenCL kernel `fuse_depth` (core/fusion_dm.py:630-674), which is NOT the CPU
// path's: one float32 3x4 map index -> pixel (proj = K lw IND, :695), bilinear depth (:605-622), pixels with no or near depth carve
// free space (dz = -TDIST, :652-653), dz = voxel depth - measured depth (sign opposite to fuseDepths, :655-658), update iff
// dz < TDIST with w <- min(1 + w, WMAX), T <- ((w' - 1) T + max(-TDIST, dz)) / w' (:667-672).  Every operation in float32 in the
// kernel text's order (no contraction).  One voxel per lane, lanes along z.
struct OclParams {
    float proj[12], kinv2[3], tdist, wmax;
    int X, Y, Z, x0, nx, H, W;
};

__global__ __launch_bounds__(256) void integrate_depth_ocl_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                   const float *__restrict__ depth, const OclParams p) {
    const long lin = (long)blockIdx.x * 256 + threadIdx.x;
    if (lin >= (long)p.nx * p.Y * p.Z) return;
    const int z = (int)(lin % p.Z), y = (int)((lin / p.Z) % p.Y), xl = (int)(lin / ((long)p.Z * p.Y));
    const float xf = (float)(p.x0 + xl), yf = (float)y, zf = (float)z;
    const float u = ((p.proj[0] * xf + p.proj[1] * yf) + p.proj[2] * zf) + p.proj[3];       // :640-642
    const float v = ((p.proj[4] * xf + p.proj[5] * yf) + p.proj[6] * zf) + p.proj[7];
    const float w = ((p.proj[8] * xf + p.proj[9] * yf) + p.proj[10] * zf) + p.proj[11];
    float px = u / w, py = v / w;                                                           // :645-646
    // :647 `if (px < 0 || py < 0 || px >= DM_X - 1 || py >= DM_Y - 1) return;` -- a NaN (w == 0) passes that test in the reference
    // and then indexes with an undefined int; here it is skipped
    if (!(px >= 0.0f && py >= 0.0f && px < (float)(p.W - 1) && py < (float)(p.H - 1))) return;
    const int ix = (int)floorf(px), iy = (int)floorf(py);                                   // :607-608
    const float wx = px - (float)ix, wy = py - (float)iy;
    const int lu = iy * p.W + ix, lb = (iy + 1) * p.W + ix;
    const float up = depth[lu] * (1.0f - wx) + depth[lu + 1] * wx;                          // :617-619
    const float bot = depth[lb] * (1.0f - wx) + depth[lb + 1] * wx;
    const float pz = -(up * (1.0f - wy) + bot * wy);                                        // :649
    float dz;
    if (pz <= p.tdist) {
        dz = -p.tdist;                                                                      // :652-653
    } else {
        px *= pz; py *= pz;
        dz = (p.kinv2[0] * (px - u) + p.kinv2[1] * (py - v)) + p.kinv2[2] * (pz - w);       // :657
        dz = -dz;
    }
    if (dz < p.tdist) {                                                                     // :667
        const long off = lin;
        const float old = tsdf[off];
        const float nw = fminf(1.0f + tsdf_w[off], p.wmax);
        tsdf[off] = ((nw - 1.0f) * old + 1.0f * fmaxf(-p.tdist, dz)) / nw;                  // :671
        tsdf_w[off] = nw;
    }
}

// Any volume dtype: the reference's chain evaluated exactly for every voxel.
template <typename VolT, typename DepthT, int VEC, bool PINHOLE>
__global__ __launch_bounds__(256) void integrate_depth_exact_kernel(VolT *__restrict__ tsdf,
                                                                     VolT *__restrict__ tsdf_w,
                                                                     const DepthT *__restrict__ depth,
                                                                     const IntegrateParams p) {
    int y, zp;
    pack_coords(p, y, zp);
    if (y >= p.Y) return;
    const int z0 = zp * VEC;
    const int xl_end = min(p.nx, (int)(blockIdx.y + 1) * p.planes_per_block);
  for (int xl = blockIdx.y * p.planes_per_block; xl < xl_end; ++xl) {
    const int x = p.x0 + xl;
    double sdv[VEC];
    bool upd[VEC];
    bool any = false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int z = z0 + j;
        double sd = 0.0;
        const bool ok = (z < p.Z) && exact_voxel<DepthT, PINHOLE>(p, depth, x, y, z, sd);
        sdv[j] = sd;
        upd[j] = ok;
        any = any | ok;
    }
    if (!any) continue;
    const size_t off = ((size_t)xl * p.Y + y) * p.Z + z0;
    using P = Pack<VolT, VEC>;
    P t = *reinterpret_cast<const P *>(tsdf + off);
    P w = *reinterpret_cast<const P *>(tsdf_w + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (upd[j]) {
            const double wt = (double)w.v[j];
            const double tv = (double)t.v[j];
            const double m = sdv[j] < p.tdist ? sdv[j] : p.tdist;                      // min(tdist, sd)
            t.v[j] = (VolT)((p.scale * tv * wt + m) / (p.scale * (1.0 + wt)));       // :209
            const double nw = 1.0 + wt;
            w.v[j] = (VolT)(nw < p.wmax ? nw : p.wmax);                                // :210
        }
    }
    *reinterpret_cast<P *>(tsdf + off) = t;
    *reinterpret_cast<P *>(tsdf_w + off) = w;
  }
}

template <typename VolT, typename DepthT, int VEC, bool FAST>
static int launch_integrate(void *tsdf, void *tsdf_w, const void *depth, IntegrateParams &p,
                            bool pinhole, hipStream_t stream) {
    const long per_plane = (long)p.Y * p.zpacks;
    const unsigned gx = (unsigned)((per_plane + 255) / 256);
    const unsigned gy = (unsigned)((p.nx + p.planes_per_block - 1) / p.planes_per_block);
    dim3 grid(gx, gy);
    dim3 block(256);
    if constexpr (FAST) {
        // strided lanes need whole 256-voxel runs per wave: 64 packs of one row
        const bool strided = VEC == 4 && p.zpacks % 64 == 0 && getenv("DFH_STRIDED");   // opt-in: measured no faster
        if (strided) {
            if constexpr (VEC == 4) {
                if (pinhole) {
                    hipLaunchKernelGGL((integrate_depth_kernel<DepthT, 4, true, true>), grid, block, 0, stream,
                                       (float *)tsdf, (float *)tsdf_w, (const DepthT *)depth, p);
                } else {
                    hipLaunchKernelGGL((integrate_depth_kernel<DepthT, 4, false, true>), grid, block, 0, stream,
                                       (float *)tsdf, (float *)tsdf_w, (const DepthT *)depth, p);
                }
            }
        } else if (VEC == 4 && p.planes_per_block == 1 && (long)p.nx * p.Y * p.Z <= kEarlyRowsMaxVoxels && !getenv("DFH_K1_LATE_LOADS")) {
            if (pinhole)
                hipLaunchKernelGGL((integrate_depth_rows_early_kernel<DepthT, true>), grid, block, 0, stream, (float *)tsdf, (float *)tsdf_w,
                                   (const DepthT *)depth, p);
            else
                hipLaunchKernelGGL((integrate_depth_rows_early_kernel<DepthT, false>), grid, block, 0, stream, (float *)tsdf, (float *)tsdf_w,
                                   (const DepthT *)depth, p);
        } else if (pinhole) {
            hipLaunchKernelGGL((integrate_depth_kernel<DepthT, VEC, true, false>), grid, block, 0, stream,
                               (float *)tsdf, (float *)tsdf_w, (const DepthT *)depth, p);
        } else {
            hipLaunchKernelGGL((integrate_depth_kernel<DepthT, VEC, false, false>), grid, block, 0, stream,
                               (float *)tsdf, (float *)tsdf_w, (const DepthT *)depth, p);
        }
    } else {
        if (pinhole) {
            hipLaunchKernelGGL((integrate_depth_exact_kernel<VolT, DepthT, VEC, true>), grid, block, 0, stream,
                               (VolT *)tsdf, (VolT *)tsdf_w, (const DepthT *)depth, p);
        } else {
            hipLaunchKernelGGL((integrate_depth_exact_kernel<VolT, DepthT, VEC, false>), grid, block, 0, stream,
                               (VolT *)tsdf, (VolT *)tsdf_w, (const DepthT *)depth, p);
        }
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

// Fold index -> pos -> lpos -> K*lpos into one affine map per output component (host, fp64).
static void fold_affine(IntegrateParams &p) {
    const double *lw = p.lw.m;
    const double *K = p.K.m;
    const double off[3] = {p.cx - p.scale * p.half, p.cy - p.scale * p.half, p.cz - p.scale * p.half};
    double L[3][4];     // lpos_r = L[r][0..2] . idx + L[r][3]
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) L[r][c] = lw[4 * r + c] * p.scale;
        L[r][3] = lw[4 * r + 0] * off[0] + lw[4 * r + 1] * off[1] + lw[4 * r + 2] * off[2] + lw[4 * r + 3];
    }
    for (int r = 0; r < 3; ++r) {      // p_r = K[r] . lpos
        double row[4];
        for (int c = 0; c < 4; ++c) row[c] = K[3 * r + 0] * L[0][c] + K[3 * r + 1] * L[1][c] + K[3 * r + 2] * L[2][c];
        p.Ax[r] = row[0]; p.Ay[r] = row[1]; p.Az[r] = row[2]; p.Ac[r] = row[3];
    }
    p.Ax[3] = L[2][0]; p.Ay[3] = L[2][1]; p.Az[3] = L[2][2]; p.Ac[3] = L[2][3];
    for (int r = 0; r < 2; ++r) {      // exact power-of-two scaling: u, v come out in 2^-20 px units
        p.Ax[r] *= (double)kFixOne; p.Ay[r] *= (double)kFixOne; p.Az[r] *= (double)kFixOne; p.Ac[r] *= (double)kFixOne;
    }
    p.inv_scale = 1.0 / p.scale;
    p.tdist_f = (float)p.tdist;
    p.ts_f = (float)(p.tdist * p.inv_scale);
    p.wmax_f = (float)p.wmax;
}

// One view's parameters (shared by the single- and the multi-view entry points); returns whether K is a pinhole matrix.
static bool fill_params(IntegrateParams &p, const int res[3], int tsdf_res, int x0, int x1, int H, int W, const double K[9],
                        const double Kinv[9], const double lw[12], double scale, const double center[3], double tdist,
                        double wmax, bool vec4) {
    for (int i = 0; i < 9; ++i) { p.K.m[i] = K[i]; p.Kinv.m[i] = Kinv[i]; }
    for (int i = 0; i < 12; ++i) p.lw.m[i] = lw[i];
    p.scale = scale; p.cx = center[0]; p.cy = center[1]; p.cz = center[2];
    p.half = (double)tsdf_res / 2.0;                    // np.zeros(3) + tsdf_res/2 (:183)
    p.tdist = tdist; p.wmax = wmax;
    p.X = res[0]; p.Y = res[1]; p.Z = res[2];
    p.x0 = x0; p.nx = x1 - x0; p.H = H; p.W = W;
    p.zpacks = vec4 ? res[2] / 4 : res[2];
    p.zp_shift = -1;
    for (int b = 0; b < 31; ++b) if (p.zpacks == (1 << b)) p.zp_shift = b;
    p.pyr = nullptr; p.cull = 0;
    for (int l = 0; l < 6; ++l) { p.pyr_off[l] = 0; p.pyr_w[l] = 0; p.pyr_h[l] = 0; }
    fold_affine(p);
    {   // ~kTargetBlocks blocks in total, each looping over consecutive x planes
        const long per_plane_blocks = ((long)p.Y * p.zpacks + 255) / 256;
        long chunks = (kTargetBlocks + per_plane_blocks - 1) / per_plane_blocks;
        if (chunks < 1) chunks = 1;
        if (chunks > p.nx) chunks = p.nx;
        p.planes_per_block = (int)((p.nx + chunks - 1) / chunks);
        const char *env = getenv("DFH_PLANES_PER_BLOCK");       // tuning knob for kbench sweeps
        if (env && atoi(env) > 0) p.planes_per_block = atoi(env);
    }
    return K[1] == 0.0 && K[3] == 0.0 && K[6] == 0.0 && K[7] == 0.0 && K[8] == 1.0 && Kinv[6] == 0.0 && Kinv[7] == 0.0 &&
           Kinv[8] == 1.0;
}

}  // namespace dfh

namespace dfh {

static size_t params_bytes(int n_views) { return ((size_t)n_views * sizeof(IntegrateParams) + 15) / 16 * 16; }

// pyramid geometry into p, pointer = workspace + params + view * pyramid
static void attach_pyramid(IntegrateParams &p, void *workspace, int n_views, int view, int H, int W) {
    const size_t pf = pyramid_floats(H, W, p.pyr_off, p.pyr_w, p.pyr_h);
    p.pyr = reinterpret_cast<const float *>(static_cast<char *>(workspace) + params_bytes(n_views)) + (size_t)view * pf;
}

template <typename DepthT>
static int launch_pyramids(const void *const *depth, int n_views, int H, int W, const IntegrateParams *hp, hipStream_t s) {
    PyrViews pv;
    for (int v = 0; v < n_views; ++v) { pv.depth[v] = depth[v]; pv.pyr[v] = const_cast<float *>(hp[v].pyr); }
    dim3 grid((unsigned)((W + 31) / 32), (unsigned)((H + 31) / 32), (unsigned)n_views);
    hipLaunchKernelGGL(depth_pyramid_kernel<DepthT>, grid, dim3(256), 0, s, pv, H, W, hp[0]);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

static BrickGeom brick_geom(int Y, int Z, int nx, dim3 &grid, int &n_bricks) {
    BrickGeom g;
    g.nzg = (Z + 4 * kBrZ - 1) / (4 * kBrZ);
    g.nyb = (Y + kBrY - 1) / kBrY;
    const int nxb = (nx + kBrX - 1) / kBrX;
    g.nxb = nxb;
    // launch order of the workgroups: measured at 512^3 (profiles/r2_k1_experiments.txt) y-fastest beats z-fastest by 3-10 %
    // (which HBM channels the workgroups in flight hit together), an XCD-contiguous order loses 20 %
    g.order = getenv("DFH_K1_ORDER") ? atoi(getenv("DFH_K1_ORDER")) : 1;
    if (g.order == 2 && ((long)g.nzg * g.nyb * nxb) % 8 != 0) g.order = 0;     // the XCD split needs whole eighths
    grid = dim3((unsigned)(g.nzg * g.nyb), (unsigned)nxb);
    n_bricks = g.nzg * g.nyb * nxb * 4;
    return g;
}

static size_t pyr_bytes(int n_views, int H, int W) { return (size_t)n_views * pyramid_floats(H, W, nullptr, nullptr, nullptr) * sizeof(float); }

}  // namespace dfh

extern "C" size_t dfh_integrate_workspace_bytes(int n_views, int H, int W, const int res[3], int x0, int x1) {
    if (n_views <= 0 || H < 2 || W < 2 || !res || res[1] <= 0 || res[2] <= 0 || x1 < x0) return 0;
    dim3 grid;
    int n_bricks = 0;
    dfh::brick_geom(res[1], res[2], x1 - x0, grid, n_bricks);
    return dfh::params_bytes(n_views) + dfh::pyr_bytes(n_views, H, W) + ((size_t)n_bricks * sizeof(unsigned short) + 15) / 16 * 16;
}

extern "C" int dfh_integrate_depth(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3],
                                   int tsdf_res, int x0, int x1, const void *depth, int depth_dtype,
                                   int H, int W, const double K[9], const double Kinv[9],
                                   const double lw[12], double scale, const double center[3],
                                   double tdist, double wmax, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(tsdf && tsdf_w && depth && res && K && Kinv && lw && center, "dfh_integrate_depth: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_integrate_depth: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(depth_dtype == DFH_F32 || depth_dtype == DFH_F64, "dfh_integrate_depth: bad depth_dtype %d", depth_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0, "dfh_integrate_depth: bad grid %dx%dx%d", res[0], res[1], res[2]);
    DFH_REQUIRE(0 <= x0 && x0 <= x1 && x1 <= res[0], "dfh_integrate_depth: slab [%d,%d) outside [0,%d)", x0, x1, res[0]);
    DFH_REQUIRE(H >= 2 && W >= 2, "dfh_integrate_depth: depth map %dx%d too small", H, W);
    DFH_REQUIRE((long)H * W < (1L << 31), "dfh_integrate_depth: depth map too large");
    DFH_REQUIRE(x1 - x0 <= 65535, "dfh_integrate_depth: slab has more than 65535 planes");
    if (x1 == x0) return DFH_OK;

    IntegrateParams p;
    const size_t esz = vol_dtype == DFH_F32 ? 4 : 8;
    const bool vec4 = (res[2] % 4 == 0) && ((uintptr_t)tsdf % (4 * esz) == 0) && ((uintptr_t)tsdf_w % (4 * esz) == 0) &&
                      !getenv("DFH_FORCE_SCALAR");
    const bool pinhole = fill_params(p, res, tsdf_res, x0, x1, H, W, K, Kinv, lw, scale, center, tdist, wmax, vec4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // fixed-point pixel coordinates need (dim-1) << 20 to fit in int32
    const bool fast_ok = H <= kFastMaxDim && W <= kFastMaxDim && scale > 0.0 && tdist > 0.0;

    // brick sweep + culling: float32 volumes on the fast path, with a workspace for the depth pyramid and the brick masks
    // For ONE view the brick sweep costs two small launches more (pyramid, classification) and pays off on large slabs only:
    // 512^3 337-350 us against 360-394 us for the row sweep (a view that updates nothing: 73 against 233 us), 256^3 48-51
    // against 43.5 us (profiles/r2_k1_experiments.txt).  Smaller slabs keep the row sweep; DFH_K1_BRICKS_MIN overrides.
    const bool have_ws = workspace && workspace_bytes >= dfh_integrate_workspace_bytes(1, H, W, res, x0, x1);
    const long bricks_min = getenv("DFH_K1_BRICKS_MIN") ? atol(getenv("DFH_K1_BRICKS_MIN")) : 131072;
    const long slab_bricks = (long)((res[1] + kBrY - 1) / kBrY) * ((res[2] + 4 * kBrZ - 1) / (4 * kBrZ)) * ((x1 - x0 + kBrX - 1) / kBrX) * 4;
    const bool bricks = vol_dtype == DFH_F32 && fast_ok && vec4 && !getenv("DFH_K1_NO_BRICKS") && slab_bricks >= bricks_min &&
                        (have_ws || getenv("DFH_K1_BRICKS_NOCULL"));
    if (bricks) {
        const bool cull = have_ws && !getenv("DFH_K1_BRICKS_NOCULL");
        dim3 grid;
        int n_bricks = 0;
        const BrickGeom g = brick_geom(p.Y, p.Z, p.nx, grid, n_bricks);
        unsigned short *mask = nullptr;
        if (cull) {
            p.cull = 1;
            attach_pyramid(p, workspace, 1, 0, H, W);
            mask = reinterpret_cast<unsigned short *>(static_cast<char *>(workspace) + params_bytes(1) + pyr_bytes(1, H, W));
            const void *dptr[1] = {depth};
            const int rc = depth_dtype == DFH_F32 ? launch_pyramids<float>(dptr, 1, H, W, &p, s) : launch_pyramids<double>(dptr, 1, H, W, &p, s);
            if (rc != DFH_OK) return rc;
            const dim3 cgrid((unsigned)((n_bricks + 255) / 256));
            if (pinhole) hipLaunchKernelGGL((brick_classify_kernel<true, true>), cgrid, dim3(256), 0, s, nullptr, p, 1, g, n_bricks, mask);
            else hipLaunchKernelGGL((brick_classify_kernel<false, true>), cgrid, dim3(256), 0, s, nullptr, p, 1, g, n_bricks, mask);
        }
        const bool early = !getenv("DFH_K1_LATE_LOADS");
#define DFH_BRICK(DT, PH) do { if (early) hipLaunchKernelGGL((integrate_depth_brick_kernel<DT, PH, true>), grid, dim3(256), 0, s, (float *)tsdf, (float *)tsdf_w, (const DT *)depth, p, g, mask); \
                               else hipLaunchKernelGGL((integrate_depth_brick_kernel<DT, PH, false>), grid, dim3(256), 0, s, (float *)tsdf, (float *)tsdf_w, (const DT *)depth, p, g, mask); } while (0)
        if (depth_dtype == DFH_F32) { if (pinhole) DFH_BRICK(float, true); else DFH_BRICK(float, false); }
        else { if (pinhole) DFH_BRICK(double, true); else DFH_BRICK(double, false); }
#undef DFH_BRICK
        DFH_HIP_CHECK(hipGetLastError());
        return DFH_OK;
    }

#define DFH_DISPATCH(VT, DT, FAST)                                                            \
    return vec4 ? launch_integrate<VT, DT, 4, FAST>(tsdf, tsdf_w, depth, p, pinhole, s)       \
                : launch_integrate<VT, DT, 1, FAST>(tsdf, tsdf_w, depth, p, pinhole, s)
    if (vol_dtype == DFH_F32) {
        if (fast_ok) {
            if (depth_dtype == DFH_F32) { DFH_DISPATCH(float, float, true); }
            DFH_DISPATCH(float, double, true);
        }
        if (depth_dtype == DFH_F32) { DFH_DISPATCH(float, float, false); }
        DFH_DISPATCH(float, double, false);
    }
    if (depth_dtype == DFH_F32) { DFH_DISPATCH(double, float, false); }
    DFH_DISPATCH(double, double, false);
#undef DFH_DISPATCH
}

extern "C" size_t dfh_integrate_multi_workspace_bytes(int n_views) {
    return n_views > 0 ? dfh::params_bytes(n_views) : 0;
}

// fresh: the volumes start as (fresh_value, 0): filled here, or -- brick sweep -- by the sweep itself
static int integrate_multi_fill(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int x0, int x1, double fresh_value, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(tsdf && tsdf_w && res, "dfh_integrate_depth_multi_fresh: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_integrate_depth_multi_fresh: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && 0 <= x0 && x0 <= x1 && x1 <= res[0], "dfh_integrate_depth_multi_fresh: bad grid or slab");
    const size_t n = (size_t)(x1 - x0) * res[1] * res[2];
    if (n == 0) return DFH_OK;
    const unsigned nb = (unsigned)((n + 255) / 256 < 65536 * 16 ? (n + 255) / 256 : 65536 * 16);
    if (vol_dtype == DFH_F32)
        hipLaunchKernelGGL(fill_pair_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (float *)tsdf, (float)fresh_value, (float *)tsdf_w, 0.0f, n);
    else
        hipLaunchKernelGGL(fill_pair_kernel<double>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (double *)tsdf, fresh_value, (double *)tsdf_w, 0.0, n);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

static int integrate_multi_impl(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int tsdf_res, int x0, int x1,
                                int n_views, const void *const *depth, int depth_dtype, int H, int W,
                                const double K[9], const double Kinv[9], const double *lw, double scale,
                                const double center[3], double tdist, double wmax, void *workspace,
                                size_t workspace_bytes, void *stream, bool fresh, double fresh_value) {
    using namespace dfh;
    DFH_REQUIRE(n_views >= 0 && n_views <= kMaxViews, "dfh_integrate_depth_multi: %d views (at most %d per call)", n_views, kMaxViews);
    if (n_views == 0) return fresh ? integrate_multi_fill(tsdf, tsdf_w, vol_dtype, res, x0, x1, fresh_value, stream) : DFH_OK;
    DFH_REQUIRE(tsdf && tsdf_w && depth && res && K && Kinv && lw && center, "dfh_integrate_depth_multi: null pointer");
    for (int v = 0; v < n_views; ++v) DFH_REQUIRE(depth[v], "dfh_integrate_depth_multi: depth map %d is null", v);
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_integrate_depth_multi: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(depth_dtype == DFH_F32 || depth_dtype == DFH_F64, "dfh_integrate_depth_multi: bad depth_dtype %d", depth_dtype);
    const size_t esz = vol_dtype == DFH_F32 ? 4 : 8;
    const bool vec4 = res[0] > 0 && res[1] > 0 && res[2] > 0 && (res[2] % 4 == 0) && ((uintptr_t)tsdf % (4 * esz) == 0) &&
                      ((uintptr_t)tsdf_w % (4 * esz) == 0) && !getenv("DFH_FORCE_SCALAR");
    const bool fast_ok = vol_dtype == DFH_F32 && H <= kFastMaxDim && W <= kFastMaxDim && scale > 0.0 && tdist > 0.0 &&
                         workspace && workspace_bytes >= dfh_integrate_multi_workspace_bytes(n_views) && !getenv("DFH_K1_NO_MULTI");
    if (!fast_ok || n_views == 1 || x1 <= x0) {
        // one sweep per view: the same results (every argument is checked there)
        if (fresh) {
            const int rc = integrate_multi_fill(tsdf, tsdf_w, vol_dtype, res, x0, x1, fresh_value, stream);
            if (rc != DFH_OK) return rc;
        }
        for (int v = 0; v < n_views; ++v) {
            const int rc = dfh_integrate_depth(tsdf, tsdf_w, vol_dtype, res, tsdf_res, x0, x1, depth[v], depth_dtype, H, W, K, Kinv,
                                               lw + 12 * v, scale, center, tdist, wmax, workspace, workspace_bytes, stream);
            if (rc != DFH_OK) return rc;
        }
        return DFH_OK;
    }
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0, "dfh_integrate_depth_multi: bad grid %dx%dx%d", res[0], res[1], res[2]);
    DFH_REQUIRE(0 <= x0 && x0 <= x1 && x1 <= res[0], "dfh_integrate_depth_multi: slab [%d,%d) outside [0,%d)", x0, x1, res[0]);
    DFH_REQUIRE(H >= 2 && W >= 2 && (long)H * W < (1L << 31), "dfh_integrate_depth_multi: bad depth map size %dx%d", H, W);
    DFH_REQUIRE(x1 - x0 <= 65535, "dfh_integrate_depth_multi: slab has more than 65535 planes");
    IntegrateParams hp[kMaxViews];
    ViewPtrs vp;
    bool pinhole = true;
    const bool have_pyr = workspace_bytes >= dfh_integrate_workspace_bytes(n_views, H, W, res, x0, x1);
    const bool bricks = vec4 && !getenv("DFH_K1_NO_BRICKS") && (have_pyr || getenv("DFH_K1_BRICKS_NOCULL"));
    const bool cull = bricks && have_pyr && !getenv("DFH_K1_BRICKS_NOCULL");
    for (int v = 0; v < n_views; ++v) {
        pinhole = fill_params(hp[v], res, tsdf_res, x0, x1, H, W, K, Kinv, lw + 12 * v, scale, center, tdist, wmax, vec4) && pinhole;
        vp.depth[v] = depth[v];
        hp[v].cull = cull ? 1 : 0;
        if (cull) attach_pyramid(hp[v], workspace, n_views, v, H, W);
    }
    vp.n = n_views;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the views' parameters travel as kernel arguments of a tiny launch that writes them into the workspace: a hipMemcpyAsync
    // from this (pageable) stack array is staged by the runtime and left the device idle for ~40 us per call
    for (int v0 = 0; v0 < n_views; v0 += kParamChunk) {
        ParamChunk c;
        const int n = n_views - v0 < kParamChunk ? n_views - v0 : kParamChunk;
        for (int v = 0; v < n; ++v) c.v[v] = hp[v0 + v];
        hipLaunchKernelGGL(upload_params_kernel, dim3(1), dim3(256), 0, s, static_cast<IntegrateParams *>(workspace) + v0, c, n);
    }
    const IntegrateParams &p = hp[0];
    const IntegrateParams *dv = static_cast<const IntegrateParams *>(workspace);
    if (bricks) {
        dim3 bgrid;
        int n_bricks = 0;
        const BrickGeom g = brick_geom(p.Y, p.Z, p.nx, bgrid, n_bricks);
        unsigned short *mask = nullptr;
        if (cull) {
            mask = reinterpret_cast<unsigned short *>(static_cast<char *>(workspace) + params_bytes(n_views) + pyr_bytes(n_views, H, W));
            const int rc = depth_dtype == DFH_F32 ? launch_pyramids<float>(depth, n_views, H, W, hp, s) : launch_pyramids<double>(depth, n_views, H, W, hp, s);
            if (rc != DFH_OK) return rc;
            const dim3 cgrid((unsigned)((n_bricks + 255) / 256));
            if (pinhole) hipLaunchKernelGGL((brick_classify_kernel<true, false>), cgrid, dim3(256), 0, s, dv, p, n_views, g, n_bricks, mask);
            else hipLaunchKernelGGL((brick_classify_kernel<false, false>), cgrid, dim3(256), 0, s, dv, p, n_views, g, n_bricks, mask);
        }
#define DFH_MBRICK(DT, PH, FR) hipLaunchKernelGGL((integrate_depth_multi_brick_kernel<DT, PH, FR>), bgrid, dim3(256), 0, s, (float *)tsdf, (float *)tsdf_w, dv, vp, g, mask, (float)fresh_value)
#define DFH_MBRICK2(DT, PH) do { if (fresh) DFH_MBRICK(DT, PH, true); else DFH_MBRICK(DT, PH, false); } while (0)
        if (depth_dtype == DFH_F32) { if (pinhole) DFH_MBRICK2(float, true); else DFH_MBRICK2(float, false); }
        else { if (pinhole) DFH_MBRICK2(double, true); else DFH_MBRICK2(double, false); }
#undef DFH_MBRICK2
#undef DFH_MBRICK
        DFH_HIP_CHECK(hipGetLastError());
        return DFH_OK;
    }
    if (fresh) {                                                    // (the plain multi-view sweep loads what it updates)
        const int rc = integrate_multi_fill(tsdf, tsdf_w, vol_dtype, res, x0, x1, fresh_value, stream);
        if (rc != DFH_OK) return rc;
    }
    dim3 grid((unsigned)(((long)p.Y * p.zpacks + 255) / 256), (unsigned)((p.nx + p.planes_per_block - 1) / p.planes_per_block)), block(256);
#define DFH_MULTI(DT, VEC, PH) hipLaunchKernelGGL((integrate_depth_multi_kernel<DT, VEC, PH>), grid, block, 0, s, (float *)tsdf, (float *)tsdf_w, dv, vp)
    if (depth_dtype == DFH_F32) {
        if (vec4) { if (pinhole) DFH_MULTI(float, 4, true); else DFH_MULTI(float, 4, false); }
        else { if (pinhole) DFH_MULTI(float, 1, true); else DFH_MULTI(float, 1, false); }
    } else {
        if (vec4) { if (pinhole) DFH_MULTI(double, 4, true); else DFH_MULTI(double, 4, false); }
        else { if (pinhole) DFH_MULTI(double, 1, true); else DFH_MULTI(double, 1, false); }
    }
#undef DFH_MULTI
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

extern "C" int dfh_integrate_depth_multi(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int tsdf_res, int x0, int x1,
                                         int n_views, const void *const *depth, int depth_dtype, int H, int W,
                                         const double K[9], const double Kinv[9], const double *lw, double scale,
                                         const double center[3], double tdist, double wmax, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    return integrate_multi_impl(tsdf, tsdf_w, vol_dtype, res, tsdf_res, x0, x1, n_views, depth, depth_dtype, H, W, K, Kinv, lw, scale, center,
                                tdist, wmax, workspace, workspace_bytes, stream, false, 0.0);
}

extern "C" int dfh_integrate_depth_multi_fresh(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int tsdf_res, int x0, int x1,
                                               double fresh_value, int n_views, const void *const *depth, int depth_dtype, int H, int W,
                                               const double K[9], const double Kinv[9], const double *lw, double scale,
                                               const double center[3], double tdist, double wmax, void *workspace,
                                               size_t workspace_bytes, void *stream) {
    return integrate_multi_impl(tsdf, tsdf_w, vol_dtype, res, tsdf_res, x0, x1, n_views, depth, depth_dtype, H, W, K, Kinv, lw, scale, center,
                                tdist, wmax, workspace, workspace_bytes, stream, true, fresh_value);
}

extern "C" int dfh_integrate_depth_ocl(float *tsdf, float *tsdf_w, const int res[3], int x0, int x1, const float *depth, int H, int W,
                                       const float proj[12], const float kinv_row2[3], float tdist, float wmax, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(tsdf && tsdf_w && res && depth && proj && kinv_row2, "dfh_integrate_depth_ocl: null pointer");
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0, "dfh_integrate_depth_ocl: bad grid %dx%dx%d", res[0], res[1], res[2]);
    DFH_REQUIRE(0 <= x0 && x0 <= x1 && x1 <= res[0], "dfh_integrate_depth_ocl: slab [%d,%d) outside [0,%d)", x0, x1, res[0]);
    DFH_REQUIRE(H >= 2 && W >= 2 && (long)H * W < (1L << 31), "dfh_integrate_depth_ocl: bad depth map size %dx%d", H, W);
    if (x1 == x0) return DFH_OK;
    OclParams p;
    for (int i = 0; i < 12; ++i) p.proj[i] = proj[i];
    for (int i = 0; i < 3; ++i) p.kinv2[i] = kinv_row2[i];
    p.tdist = tdist; p.wmax = wmax;
    p.X = res[0]; p.Y = res[1]; p.Z = res[2]; p.x0 = x0; p.nx = x1 - x0; p.H = H; p.W = W;
    const long n = (long)p.nx * p.Y * p.Z;
    DFH_REQUIRE((n + 255) / 256 < (1L << 31), "dfh_integrate_depth_ocl: slab too large");
    hipLaunchKernelGGL(integrate_depth_ocl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tsdf, tsdf_w, depth, p);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}
