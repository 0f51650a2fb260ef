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
// brick = kBrX x kBrY x kBrZ voxels = 4 x 2 x 32 = one wave instruction's worth (64 lanes x 4 voxels along z); eight z-packs =
// 128 bytes per row: every cache line of T and w is touched by ONE instruction (4 x 4 x 16, round 2's shape, culls a little
// finer on oblique views but its 64-byte segments cost 15-40 % in the sweep: profiles/r3_k1_experiments.txt)
constexpr int kBrX = 4, kBrY = 2, kBrZ = 256 / (kBrX * kBrY);
constexpr int kMaxColumnBricks = 16;            // bricks per wave of the column walk (their masks sit in the first lanes)
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
// gather.  Here a wave works on compact bricks of 4 x BY x (64 / BY) voxels (BY = 4: 4 x 4 x 16, lane = z-pack + 4 y + 16 x,
// a quarter-wave touches four 64-byte row segments; BY = 2: 4 x 2 x 32, 128-byte segments).  Voxel centres of a brick span a
// box; an affine map followed by the perspective divide keeps the image of a box in front of the camera inside the bounding
// rectangle of its eight projected corners.  So:
//   * all corners in front of the camera and the rectangle entirely outside [0, W-1) x [0, H-1) (with a 2^-10 px
//     margin, the corners being good to 2^-19 px): no voxel is visible -> the brick is skipped for that view;
//   * (pinhole) z_max = the largest valid depth over the pixels the rectangle can round to, from a max-pyramid of the
//     depth map, and z_max + tdist (+ margin) <= the smallest corner depth: every voxel has sd <= -tdist or no depth at
//     all -> nothing is updated -> skip.
// Skipping is only ever done when NO voxel of the brick would be updated, so the result is bit-identical to the row
// kernels (tests compare every voxel).  Bricks that survive run view_pack exactly as before.
//
// Round 3: a wave no longer owns ONE brick but walks a COLUMN of them along z (tools/ubench/rmw_stream.hip is the
// measurement behind this).  A wave's slot is held from launch until its last store has been acknowledged, and nothing
// but its own instructions can fill that time: with one brick per wave the life of a wave was
//   launch + index arithmetic + mask load | projection | gathers + T/w in flight | decisions + average | store drain
// = three exposed memory latencies around ~0.5 us of arithmetic, 8 waves per SIMD could not cover them (VALU 43 % busy,
// waves waiting 75 % of their life) and the sweep ran at 0.57 of HBM peak where a bare read-modify-write of the same
// bytes reaches 0.72 (0.80 with non-temporal accesses on whole 128-byte lines).  Walking a column, the wave has the NEXT
// brick's T / w loads in flight while it projects the current one, and the stores of the previous brick drain meanwhile:
// the launch, the index arithmetic and the mask load are paid once per column, the grid is three-dimensional (no
// divisions: the first version spent ~600 scalar instructions per wave on 64-bit divisions of a linear workgroup index).
struct BrickGeom {
    int nbz;           // bricks along z
    int nyb;           // bricks along y
    int nxb;           // bricks along x (slab)
    int nzi;           // bricks per wave: the four waves of workgroup (by, c, bx) take the bricks bz = 4 (c nzi + i) + wave, i < nzi
    int nzc;           // workgroups along z = ceil(nbz / (4 nzi))
};

// The projected bounding rectangle of a brick (2^-20 px fixed point, possibly saturated), its smallest camera depth and
// whether every corner lies in front of the camera.
struct BrickBounds {
    int umin, umax, vmin, vmax;
    double l2min;
    bool front;
};

// corner `c` (bit 0: +x, bit 1: +y, bit 2: +z end) of the brick whose first voxel is (x0, y0, z0)
template <bool PINHOLE, int BY>
__device__ __forceinline__ void brick_corner(const IntegrateParams &p, int x0, int y0, int z0, int c, int &qu, int &qv, double &l2, bool &front) {
    constexpr int NC = PINHOLE ? 3 : 4;
    constexpr int BZ = 256 / (kBrX * BY);
    const double xf = (double)(x0 + ((c & 1) ? kBrX - 1 : 0)), yf = (double)(y0 + ((c & 2) ? BY - 1 : 0)), zf = (double)(z0 + ((c & 4) ? BZ - 1 : 0));
    double q[NC];
#pragma unroll
    for (int r = 0; r < NC; ++r) q[r] = __builtin_fma(p.Az[r], zf, __builtin_fma(p.Ax[r], xf, __builtin_fma(p.Ay[r], yf, p.Ac[r])));
    front = q[2] > 1e-6;
    const double rr = rcp_nr1(q[2]);
    qu = cvt_i32_sat(q[0] * rr);
    qv = cvt_i32_sat(q[1] * rr);
    l2 = q[NC - 1];
}

// true when a view with these bounds provably updates no voxel of the brick
template <bool PINHOLE>
__device__ __forceinline__ bool bounds_culled(const IntegrateParams &p, BrickBounds b) {
    if (!b.front) return false;                          // a corner at or behind the camera plane: no claim
    int umin = b.umin, umax = b.umax, vmin = b.vmin, vmax = b.vmax;
    const int ulim = (p.W - 1) << kFixShift, vlim = (p.H - 1) << kFixShift;
    // (the corners are view_pack's own FMA chain at the corner voxels: ~2^-19 px, far inside the 2^-10 px margin)
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
    if (L > kPyrLevels) {                                // wider than two 32-pixel cells (long bricks seen side-on): up to 4 x 4 of them
        L = kPyrLevels;
        if (((px1 >> L) - (px0 >> L)) > 3 || ((py1 >> L) - (py0 >> L)) > 3) return false;      // no claim
    }
    const float *lv = p.pyr + p.pyr_off[L];
    const int wl = p.pyr_w[L];
    const int cx0 = px0 >> L, cx1 = px1 >> L, cy0 = py0 >> L, cy1 = py1 >> L;
    float zmax = 0.0f;                                   // (pyramid values are >= 0)
    for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) zmax = fmaxf(zmax, lv[cy * wl + cx]);
    // sd = z - l2 <= zmax - l2min for every voxel with a valid pixel; the margin covers the float32 rounding of l2min
    const float l2f = (float)b.l2min;
    return zmax + p.tdist_f + fmaf(fabsf(l2f), 4e-7f, 1e-5f) <= l2f;
}

// one thread, one brick: the eight corners in turn
template <bool PINHOLE, int BY>
__device__ __forceinline__ bool brick_culled(const IntegrateParams &p, int x0, int y0, int z0) {
    BrickBounds b{0x7fffffff, (int)0x80000000, 0x7fffffff, (int)0x80000000, __builtin_huge_val(), true};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int qu, qv;
        double l2;
        bool fr;
        brick_corner<PINHOLE, BY>(p, x0, y0, z0, c, qu, qv, l2, fr);
        b.front = b.front && fr;
        b.umin = min(b.umin, qu); b.umax = max(b.umax, qu); b.vmin = min(b.vmin, qv); b.vmax = max(b.vmax, qv);
        b.l2min = fmin(b.l2min, l2);
    }
    return bounds_culled<PINHOLE>(p, b);
}

// Classification pass: one THREAD per brick, bit v of mask[brick] = view v may update a voxel of the brick.  Bricks are
// numbered (bx * nyb + by) * nbz + bz: the bricks of one column are consecutive.
template <bool PINHOLE, bool BYVAL, int BY>
__global__ __launch_bounds__(256) void brick_classify_kernel(const IntegrateParams *__restrict__ views, const IntegrateParams p1,
                                                              int n_views, const BrickGeom g, int n_bricks,
                                                              unsigned short *__restrict__ mask) {
    constexpr int BZ = 256 / (kBrX * BY);
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_bricks) return;
    const int bz = b % g.nbz;
    const int t = b / g.nbz;
    const int by = t % g.nyb;
    const int bx = t / g.nyb;
    unsigned m = 0;
    for (int v = 0; v < n_views; ++v) {
        const IntegrateParams &p = BYVAL ? p1 : views[v];
        if (!(p.cull && brick_culled<PINHOLE, BY>(p, p.x0 + kBrX * bx, BY * by, BZ * bz))) m |= 1u << v;
    }
    mask[b] = (unsigned short)m;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
static_assert(sizeof(f32x4) == sizeof(Pack<float, 4>), "a pack is one 16-byte access");

// NT: non-temporal accesses (global_load / global_store ... nt).  Measured (tools/ubench/rmw_stream.hip, 512^3): an in-place
// read-modify-write of T and w runs at 5.7 TB/s with the default policy and at 6.4 TB/s non-temporal -- IF every 128-byte
// line is touched by one instruction (rows, 4 x 2 x 32 bricks); with 64-byte segments (4 x 4 x 16) the second toucher of a
// line finds it gone and non-temporal is slower (5.0 TB/s).
template <bool NT>
__device__ __forceinline__ Pack<float, 4> ld_pack(const float *a) {
    const f32x4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a)) : *reinterpret_cast<const f32x4 *>(a);
    Pack<float, 4> r;
    r.v[0] = v.x; r.v[1] = v.y; r.v[2] = v.z; r.v[3] = v.w;
    return r;
}
template <bool NT>
__device__ __forceinline__ void st_pack(float *a, const Pack<float, 4> &r) {
    f32x4 v;
    v.x = r.v[0]; v.y = r.v[1]; v.z = r.v[2]; v.w = r.v[3];
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(a));
    else *reinterpret_cast<f32x4 *>(a) = v;
}

// The column walk shared by the single- and the multi-view sweep: this wave's bricks as a bit set (bit i = brick
// bz0 + 4 i takes part), the lane's voxel coordinates, its row offset.
template <int BY>
struct ColumnLane {
    static constexpr int BZ = 256 / (kBrX * BY), BZP = BZ / 4;
    int xl, y, zl, bz0;
    bool in_xy;
    size_t row;
    unsigned mv;       // lane i: the 16-bit view mask of brick bz0 + 4 i (0 beyond the column)

    // returns the set of bricks to visit: those with a non-zero mask, or (`all`) every brick of the wave inside the grid
    __device__ __forceinline__ unsigned long long init(const BrickGeom &g, int nx, int Y, int Z, const unsigned short *__restrict__ mask,
                                                       unsigned full_mask, bool all) {
        const int lane = threadIdx.x & 63;
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int by = blockIdx.x, c = blockIdx.y, bx = blockIdx.z;
        xl = kBrX * bx + lane / (BY * BZP);
        y = BY * by + (lane / BZP) % BY;
        zl = 4 * (lane % BZP);
        bz0 = 4 * c * g.nzi + wv;
        in_xy = xl < nx && y < Y;
        row = ((size_t)xl * Y + y) * Z;
        const int bz = bz0 + 4 * lane;
        const bool mine = lane < g.nzi && bz < g.nbz;
        mv = 0;
        if (mine) mv = mask ? (unsigned)mask[(size_t)(bx * g.nyb + by) * g.nbz + bz] : full_mask;
        return __ballot(all ? mine : mv != 0);
    }

    __device__ __forceinline__ int z_of(int i) const { return BZ * (bz0 + 4 * i) + zl; }
};

// One view.  mask == NULL: every brick is swept (no classification pass ran).
// PREFETCH: T / w of the wave's NEXT brick are requested right after the depth gathers of the current one (they are in
// flight through its decisions, its stores and the next projection); otherwise a brick's T / w are requested when the walk
// reaches it, ahead of its projection.
#ifdef DFH_K1_WAVES                    // experiment builds: force the register budget of DFH_K1_WAVES waves per SIMD
#define DFH_K1_OCC __attribute__((amdgpu_waves_per_eu(DFH_K1_WAVES, DFH_K1_WAVES)))
#else
#define DFH_K1_OCC
#endif
template <typename DepthT, bool PINHOLE, int BY, bool PREFETCH, bool NT>
__global__ __launch_bounds__(256) DFH_K1_OCC void integrate_depth_column_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                      const DepthT *__restrict__ depth, const IntegrateParams p,
                                                                      const BrickGeom g, const unsigned short *__restrict__ mask) {
    ColumnLane<BY> c;
    unsigned long long alive = c.init(g, p.nx, p.Y, p.Z, mask, 1u, false);
    if (alive == 0) return;
    using P = Pack<float, 4>;
    const IntegrateParams *p_rare = kernarg_params(kParamsAfterThreePointers);
    int i = __builtin_ctzll(alive);
    alive &= alive - 1;
    int z0 = c.z_of(i);
    bool in_grid = c.in_xy && z0 < p.Z;
    P t, w;
    if (PREFETCH && in_grid) { t = ld_pack<NT>(tsdf + c.row + z0); w = ld_pack<NT>(tsdf_w + c.row + z0); }
    for (;;) {
        if (!PREFETCH && in_grid) { t = ld_pack<NT>(tsdf + c.row + z0); w = ld_pack<NT>(tsdf_w + c.row + z0); }
        const bool more = alive != 0;
        int zn = 0;
        bool in_grid_n = false;
        P tn, wn;
        if (more) {
            zn = c.z_of(__builtin_ctzll(alive));
            in_grid_n = c.in_xy && zn < p.Z;
        }
        float ms[4];
        bool upd[4];
        const bool any = view_pack<DepthT, 4, PINHOLE, false>(p, p_rare, depth, p.x0 + c.xl, c.y, z0, ms, upd, [&](bool) {
            if (PREFETCH && in_grid_n) { tn = ld_pack<NT>(tsdf + c.row + zn); wn = ld_pack<NT>(tsdf_w + c.row + zn); }
        });
        if (any && in_grid) {
            apply_pack<4>(t, w, ms, upd, p.wmax_f);
            st_pack<NT>(tsdf + c.row + z0, t);
            st_pack<NT>(tsdf_w + c.row + z0, w);
        }
        if (!more) break;
        alive &= alive - 1;
        z0 = zn;
        in_grid = in_grid_n;
        if (PREFETCH) { t = tn; w = wn; }
    }
}

// Several views in one sweep, column walk.  FRESH: the volume is taken to be (fresh_t, 0) everywhere -- a live volume that
// starts from np.zeros + tdist / np.zeros (core/fusion_dm.py:152-153) -- so nothing is loaded and EVERY pack is written:
// the fill and the sweep in one pass over the volume (one write of it instead of a write, a read of the updated part and
// another write).  Otherwise T / w of a surviving brick are requested before its first view's projection and written if a
// view updated the pack.
template <typename DepthT, bool PINHOLE, int BY, bool FRESH, bool NT>
__global__ __launch_bounds__(256) void integrate_depth_multi_column_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                            const IntegrateParams *__restrict__ views, const ViewPtrs vp,
                                                                            const BrickGeom g, const unsigned short *__restrict__ mask,
                                                                            const float fresh_t) {
    ColumnLane<BY> c;
    const int nx = views[0].nx, Y = views[0].Y, Z = views[0].Z;
    unsigned long long alive = c.init(g, nx, Y, Z, mask, (1u << vp.n) - 1u, FRESH);
    using P = Pack<float, 4>;
    while (alive) {
        const int i = __builtin_ctzll(alive);
        alive &= alive - 1;
        unsigned m = (unsigned)__builtin_amdgcn_readlane((int)c.mv, i);
        const int z0 = c.z_of(i);
        const bool in_grid = c.in_xy && z0 < Z;
        P t, w;
        if (FRESH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { t.v[j] = fresh_t; w.v[j] = 0.0f; }
        } else if (in_grid) {
            t = ld_pack<NT>(tsdf + c.row + z0);
            w = ld_pack<NT>(tsdf_w + c.row + z0);
        }
        bool touched = FRESH;
        while (m) {                                       // views in ascending order: the order of consecutive sweeps
            const int v = __builtin_ctz(m);
            m &= m - 1;
            const IntegrateParams &p = views[v];         // uniform address: scalar loads
            float ms[4];
            bool upd[4];
            const bool any = view_pack<DepthT, 4, PINHOLE, false, false>(p, &p, static_cast<const DepthT *>(vp.depth[v]), p.x0 + c.xl, c.y, z0, ms, upd);
            if (!(any && in_grid)) continue;
            apply_pack<4>(t, w, ms, upd, p.wmax_f);
            touched = true;
        }
        if (touched && in_grid) {
            st_pack<NT>(tsdf + c.row + z0, t);
            st_pack<NT>(tsdf_w + c.row + z0, w);
        }
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
        const bool strided = VEC == 4 && p.zpacks % 64 == 0 && on(opt().k1_strided);   // opt-in: measured no faster
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
        } else if (VEC == 4 && p.planes_per_block == 1 && (long)p.nx * p.Y * p.Z <= kEarlyRowsMaxVoxels && !on(opt().k1_late_loads)) {
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
        if (opt().k1_planes_per_block > 0) p.planes_per_block = (int)opt().k1_planes_per_block;       // tuning knob for kbench sweeps
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

// Column walk geometry for a slab of nx planes.  nzi (bricks per wave): as long a walk as leaves >= ~4 workgroups per
// workgroup slot of the chip (256 CUs x 8), at most kMaxColumnBricks; option k1_nzi overrides.
static BrickGeom brick_geom(int by_shape, int Y, int Z, int nx, dim3 &grid, int &n_bricks) {
    BrickGeom g;
    const int bz_len = 256 / (kBrX * by_shape);
    g.nbz = (Z + bz_len - 1) / bz_len;
    g.nyb = (Y + by_shape - 1) / by_shape;
    g.nxb = (nx + kBrX - 1) / kBrX;
    const int per_col = (g.nbz + 3) / 4;                   // bricks per wave if one workgroup took the whole column
    int nzi = per_col < kMaxColumnBricks ? per_col : kMaxColumnBricks;
    while (nzi > 1 && (long)g.nyb * g.nxb * ((per_col + nzi - 1) / nzi) < 8192) nzi = (nzi + 1) / 2;
    if (opt().k1_nzi > 0) nzi = (int)(opt().k1_nzi < kMaxColumnBricks ? opt().k1_nzi : kMaxColumnBricks);
    g.nzi = nzi;
    g.nzc = (per_col + nzi - 1) / nzi;
    grid = dim3((unsigned)g.nyb, (unsigned)g.nzc, (unsigned)g.nxb);
    n_bricks = g.nbz * g.nyb * g.nxb;
    return g;
}

// Non-temporal T / w accesses (the bricks' 128-byte segments make every line one instruction's): only when the slab (T and w, 8 B per voxel) does not fit
// the 256 MiB Infinity Cache -- at 256^3 (134 MB, cache resident from sweep to sweep) they cost 15 %, at 512^3 they gain 6-9 %
// (profiles/r3_k1_experiments.txt).  Option k1_nt = 0 / 1 forces them off / on.
static bool use_nt(size_t slab_voxels) {
    if (opt().k1_nt >= 0) return opt().k1_nt != 0;
    return slab_voxels * 8 > ((size_t)256 << 20);
}

static size_t pyr_bytes(int n_views, int H, int W) { return (size_t)n_views * pyramid_floats(H, W, nullptr, nullptr, nullptr) * sizeof(float); }

}  // namespace dfh

namespace dfh {

// Which sweep a single-view call takes (dfh_integrate_depth_path reports it; bench.py counts the bytes of that sweep).
//   rows: slabs up to half of 256^3 (kEarlyRowsMaxVoxels; with early loads) and everything the fast path cannot take;
//   columns: the 4 x 2 x 32 brick column walk over every brick -- slabs that fit the 256 MiB Infinity Cache (256^3: 134 MB) are
//     swept whole in less time than the two culling launches (pyramid ~5 us, classification ~8 us) save: 36-40 us unculled, 40
//     culled, 42-46 rows;
//   columns, culled: larger slabs (512^3: 240-250 us against 330 unculled, 410-425 rows).
// Options k1_no_bricks, k1_bricks_min (in 256-voxel bricks), k1_cull = 0 / 1, k1_bricks_nocull override.  (Classification
// inside the sweep's waves -- eight lanes per brick -- was built and measured slower than the pass at both sizes: its pyramid
// look-up is a memory round trip at the start of every wave, profiles/r3_k1_experiments.txt.)
static int single_view_path(int vol_dtype, const int res[3], int x0, int x1, bool fast_ok, bool vec4, bool have_ws) {
    if (vol_dtype != DFH_F32 || !fast_ok) return DFH_K1_PATH_EXACT;
    const long slab_voxels = (long)(x1 - x0) * res[1] * res[2];
    const long bricks_min = opt().k1_bricks_min >= 0 ? opt().k1_bricks_min : kEarlyRowsMaxVoxels / 256 + 1;
    const long slab_bricks = (long)((res[1] + 3) / 4) * ((res[2] + 15) / 16) * ((x1 - x0 + kBrX - 1) / kBrX);
    if (vec4 && !on(opt().k1_no_bricks) && slab_bricks >= bricks_min) {
        const bool want_cull = !on(opt().k1_bricks_nocull) && (opt().k1_cull >= 0 ? opt().k1_cull != 0 : (size_t)slab_voxels * 8 > ((size_t)256 << 20));
        if (!want_cull) return DFH_K1_PATH_COLUMNS;
        if (have_ws) return DFH_K1_PATH_COLUMNS_CULLED;
    }
    return DFH_K1_PATH_ROWS;
}

}  // namespace dfh

extern "C" size_t dfh_integrate_workspace_bytes(int n_views, int H, int W, const int res[3], int x0, int x1) {
    if (n_views <= 0 || H < 2 || W < 2 || !res || res[1] <= 0 || res[2] <= 0 || x1 < x0) return 0;
    dim3 grid;
    int n_bricks = 0;
    dfh::brick_geom(dfh::kBrY, res[1], res[2], x1 - x0, grid, n_bricks);
    return dfh::params_bytes(n_views) + dfh::pyr_bytes(n_views, H, W) + ((size_t)n_bricks * sizeof(unsigned short) + 15) / 16 * 16;
}

extern "C" int dfh_integrate_depth_path(int vol_dtype, const int res[3], int x0, int x1, int H, int W, int have_workspace) {
    using namespace dfh;
    DFH_REQUIRE(res && res[0] > 0 && res[1] > 0 && res[2] > 0 && 0 <= x0 && x0 <= x1 && x1 <= res[0], "dfh_integrate_depth_path: bad grid or slab");
    const bool fast_ok = H <= kFastMaxDim && W <= kFastMaxDim;
    const bool vec4 = res[2] % 4 == 0 && !on(opt().k1_force_scalar);
    return single_view_path(vol_dtype, res, x0, x1, fast_ok, vec4, have_workspace != 0);
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
                      !on(opt().k1_force_scalar);
    const bool pinhole = fill_params(p, res, tsdf_res, x0, x1, H, W, K, Kinv, lw, scale, center, tdist, wmax, vec4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // fixed-point pixel coordinates need (dim-1) << 20 to fit in int32
    const bool fast_ok = H <= kFastMaxDim && W <= kFastMaxDim && scale > 0.0 && tdist > 0.0;

    const bool have_ws = workspace && workspace_bytes >= dfh_integrate_workspace_bytes(1, H, W, res, x0, x1);
    const int path = single_view_path(vol_dtype, res, x0, x1, fast_ok, vec4, have_ws);
    if (path == DFH_K1_PATH_COLUMNS || path == DFH_K1_PATH_COLUMNS_CULLED) {
        const bool cull = path == DFH_K1_PATH_COLUMNS_CULLED;
        dim3 grid;
        int n_bricks = 0;
        const BrickGeom g = brick_geom(kBrY, p.Y, p.Z, p.nx, grid, n_bricks);
        unsigned short *mask = nullptr;
        if (cull) {
            p.cull = 1;
            attach_pyramid(p, workspace, 1, 0, H, W);
            const void *dptr[1] = {depth};
            const int rc = depth_dtype == DFH_F32 ? launch_pyramids<float>(dptr, 1, H, W, &p, s) : launch_pyramids<double>(dptr, 1, H, W, &p, s);
            if (rc != DFH_OK) return rc;
            mask = reinterpret_cast<unsigned short *>(static_cast<char *>(workspace) + params_bytes(1) + pyr_bytes(1, H, W));
            const dim3 cgrid((unsigned)((n_bricks + 255) / 256));
#define DFH_CLASSIFY(PH, BY) hipLaunchKernelGGL((brick_classify_kernel<PH, true, BY>), cgrid, dim3(256), 0, s, nullptr, p, 1, g, n_bricks, mask)
            if (pinhole) DFH_CLASSIFY(true, kBrY); else DFH_CLASSIFY(false, kBrY);
#undef DFH_CLASSIFY
        }
        // T / w of the wave's next brick requested one brick ahead: culled sweeps of large slabs gain 3 % on views that update
        // everything (373 against 384 us at 512^3), cache-resident slabs lose 5-10 % (80 against 70 VGPRs); k1_prefetch forces
        const bool prefetch = opt().k1_prefetch >= 0 ? opt().k1_prefetch != 0 : cull;
        const bool nt = use_nt((size_t)p.nx * p.Y * p.Z);
#define DFH_COL(DT, PH, BY, PF, NT) hipLaunchKernelGGL((integrate_depth_column_kernel<DT, PH, BY, PF, NT>), grid, dim3(256), 0, s, (float *)tsdf, (float *)tsdf_w, (const DT *)depth, p, g, mask)
#define DFH_COL_PF(DT, PH, BY, NT) do { if (prefetch) DFH_COL(DT, PH, BY, true, NT); else DFH_COL(DT, PH, BY, false, NT); } while (0)
#define DFH_COL_SHAPE(DT, PH) do { if (nt) DFH_COL_PF(DT, PH, kBrY, true); else DFH_COL_PF(DT, PH, kBrY, false); } while (0)
        if (depth_dtype == DFH_F32) { if (pinhole) DFH_COL_SHAPE(float, true); else DFH_COL_SHAPE(float, false); }
        else { if (pinhole) DFH_COL_SHAPE(double, true); else DFH_COL_SHAPE(double, false); }
#undef DFH_COL_SHAPE
#undef DFH_COL_PF
#undef DFH_COL
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
                      ((uintptr_t)tsdf_w % (4 * esz) == 0) && !on(opt().k1_force_scalar);
    const bool fast_ok = vol_dtype == DFH_F32 && H <= kFastMaxDim && W <= kFastMaxDim && scale > 0.0 && tdist > 0.0 &&
                         workspace && workspace_bytes >= dfh_integrate_multi_workspace_bytes(n_views) && !on(opt().k1_no_multi);
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
    const bool nocull = on(opt().k1_bricks_nocull);
    const bool bricks = vec4 && !on(opt().k1_no_bricks) && (have_pyr || nocull);
    const bool cull = bricks && have_pyr && !nocull;
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
        const BrickGeom g = brick_geom(kBrY, p.Y, p.Z, p.nx, bgrid, n_bricks);
        unsigned short *mask = nullptr;
        if (cull) {
            mask = reinterpret_cast<unsigned short *>(static_cast<char *>(workspace) + params_bytes(n_views) + pyr_bytes(n_views, H, W));
            const int rc = depth_dtype == DFH_F32 ? launch_pyramids<float>(depth, n_views, H, W, hp, s) : launch_pyramids<double>(depth, n_views, H, W, hp, s);
            if (rc != DFH_OK) return rc;
            const dim3 cgrid((unsigned)((n_bricks + 255) / 256));
#define DFH_CLASSIFY(PH, BY) hipLaunchKernelGGL((brick_classify_kernel<PH, false, BY>), cgrid, dim3(256), 0, s, dv, p, n_views, g, n_bricks, mask)
            if (pinhole) DFH_CLASSIFY(true, kBrY); else DFH_CLASSIFY(false, kBrY);
#undef DFH_CLASSIFY
        }
        const bool nt = use_nt((size_t)p.nx * p.Y * p.Z);
#define DFH_MCOL(DT, PH, BY, FR, NT) hipLaunchKernelGGL((integrate_depth_multi_column_kernel<DT, PH, BY, FR, NT>), bgrid, dim3(256), 0, s, (float *)tsdf, (float *)tsdf_w, dv, vp, g, mask, (float)fresh_value)
#define DFH_MCOL_FR(DT, PH, BY, NT) do { if (fresh) DFH_MCOL(DT, PH, BY, true, NT); else DFH_MCOL(DT, PH, BY, false, NT); } while (0)
#define DFH_MCOL_SHAPE(DT, PH) do { if (nt) DFH_MCOL_FR(DT, PH, kBrY, true); else DFH_MCOL_FR(DT, PH, kBrY, false); } while (0)
        if (depth_dtype == DFH_F32) { if (pinhole) DFH_MCOL_SHAPE(float, true); else DFH_MCOL_SHAPE(float, false); }
        else { if (pinhole) DFH_MCOL_SHAPE(double, true); else DFH_MCOL_SHAPE(double, false); }
#undef DFH_MCOL_SHAPE
#undef DFH_MCOL_FR
#undef DFH_MCOL
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
