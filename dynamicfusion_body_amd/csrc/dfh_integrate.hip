// K1  depth map -> TSDF integration: FusionDM.fuseDepths CPU semantics
// (reference core/fusion_dm.py:180-217), one sweep over the axis-0 planes [x0,x1).
//
// Mapping: volumes are [x][y][z] with z fastest, so consecutive lanes take consecutive
// z-packs (VEC voxels = one 16-byte fp32 access) of one (x,y) row: every wave reads and
// writes whole 1-KiB lines of T and w.  The geometric chain that decides the masks
// (projection, bounds, round-half-even pixel, z>0, sd>-tdist) runs in IEEE fp64 with the
// reference's operation order (library is built with -ffp-contract=off), so the masks are
// bit-identical to the fp64 CPU path; only the final T is rounded to the volume dtype.
#include "dfh_common.h"

namespace dfh {

template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
    T v[N];
};

struct IntegrateParams {
    Mat3 K, Kinv;
    Mat34 lw;
    double scale, cx, cy, cz, half, tdist, wmax;
    int X, Y, Z;       // global grid dims
    int x0, nx;        // slab: planes [x0, x0+nx)
    int H, W;
    int zpacks;        // ceil(Z / VEC)
};

template <typename VolT, typename DepthT, int VEC, bool PINHOLE>
__global__ __launch_bounds__(256) void integrate_depth_kernel(VolT *__restrict__ tsdf,
                                                               VolT *__restrict__ tsdf_w,
                                                               const DepthT *__restrict__ depth,
                                                               const IntegrateParams p) {
    const int lin = blockIdx.x * 256 + threadIdx.x;      // (y, zpack) inside one x plane
    const int y = lin / p.zpacks;
    const int zp = lin - y * p.zpacks;
    if (y >= p.Y) return;
    const int xl = blockIdx.y;                           // local plane
    const int z0 = zp * VEC;

    // pos = scale*(i - res/2) + center   (fusion_dm.py:191)
    const double px = p.scale * ((double)(p.x0 + xl) - p.half) + p.cx;
    const double py = p.scale * ((double)y - p.half) + p.cy;
    const double *lw = p.lw.m;
    // ((lw0*px + lw1*py) + lw2*pz) + lw3: the x/y partial sums are shared by the z-pack
    const double a0 = lw[0] * px + lw[1] * py;
    const double a1 = lw[4] * px + lw[5] * py;
    const double a2 = lw[8] * px + lw[9] * py;

    double sdv[VEC];
    bool upd[VEC];
    bool any = false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int z = z0 + j;
        const double pz = p.scale * ((double)z - p.half) + p.cz;
        const double l0 = (a0 + lw[2] * pz) + lw[3];
        const double l1 = (a1 + lw[6] * pz) + lw[7];
        const double l2 = (a2 + lw[10] * pz) + lw[11];
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
        bool ok = (z < p.Z) && (p2 != 0.0);              // util.py:318
        const double u = p0 / p2;
        const double v = p1 / p2;
        ok = ok && (u >= 0.0) && (u < (double)(p.W - 1)) && (v >= 0.0) && (v < (double)(p.H - 1));  // :195
        double sd = 0.0;
        if (ok) {
            const int ui = (int)rint(u);                 // Python round(): half to even (:196)
            const int vi = (int)rint(v);
            const double zd = -1.0 * (double)depth[(size_t)vi * p.W + ui];
            ok = zd > 0.0;                               // :197
            double cz;
            if (PINHOLE) {
                cz = zd;                                 // Kinv row 2 == [0,0,1]
            } else {
                cz = (p.Kinv.m[6] * (zd * u) + p.Kinv.m[7] * (zd * v)) + p.Kinv.m[8] * (zd * 1.0);
            }
            sd = cz - l2;                                // :201
            ok = ok && (sd > -1.0 * p.tdist);            // :203
        }
        sdv[j] = sd;
        upd[j] = ok;
        any = any || ok;
    }
    if (!any) return;

    const size_t off = ((size_t)xl * p.Y + y) * p.Z + z0;
    using P = Pack<VolT, VEC>;
    P t = *reinterpret_cast<const P *>(tsdf + off);
    P w = *reinterpret_cast<const P *>(tsdf_w + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (upd[j]) {
            const double wt = (double)w.v[j];
            const double tv = (double)t.v[j];
            const double m = sdv[j] < p.tdist ? sdv[j] : p.tdist;          // min(tdist, sd)
            t.v[j] = (VolT)((p.scale * tv * wt + m) / (p.scale * (1.0 + wt)));   // :209
            const double nw = 1.0 + wt;
            w.v[j] = (VolT)(nw < p.wmax ? nw : p.wmax);                    // :210
        }
    }
    *reinterpret_cast<P *>(tsdf + off) = t;
    *reinterpret_cast<P *>(tsdf_w + off) = w;
}

template <typename VolT, typename DepthT, int VEC>
static int launch_integrate(void *tsdf, void *tsdf_w, const void *depth, const IntegrateParams &p,
                            bool pinhole, hipStream_t stream) {
    const long per_plane = (long)p.Y * p.zpacks;
    dim3 grid((unsigned)((per_plane + 255) / 256), (unsigned)p.nx);
    dim3 block(256);
    if (pinhole) {
        hipLaunchKernelGGL((integrate_depth_kernel<VolT, DepthT, VEC, true>), grid, block, 0, stream,
                           (VolT *)tsdf, (VolT *)tsdf_w, (const DepthT *)depth, p);
    } else {
        hipLaunchKernelGGL((integrate_depth_kernel<VolT, DepthT, VEC, false>), grid, block, 0, stream,
                           (VolT *)tsdf, (VolT *)tsdf_w, (const DepthT *)depth, p);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

}  // namespace dfh

extern "C" int dfh_integrate_depth(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3],
                                   int tsdf_res, int x0, int x1, const void *depth, int depth_dtype,
                                   int H, int W, const double K[9], const double Kinv[9],
                                   const double lw[12], double scale, const double center[3],
                                   double tdist, double wmax, void *stream) {
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
    for (int i = 0; i < 9; ++i) { p.K.m[i] = K[i]; p.Kinv.m[i] = Kinv[i]; }
    for (int i = 0; i < 12; ++i) p.lw.m[i] = lw[i];
    p.scale = scale; p.cx = center[0]; p.cy = center[1]; p.cz = center[2];
    p.half = (double)tsdf_res / 2.0;                    // np.zeros(3) + tsdf_res/2 (:183)
    p.tdist = tdist; p.wmax = wmax;
    p.X = res[0]; p.Y = res[1]; p.Z = res[2];
    p.x0 = x0; p.nx = x1 - x0; p.H = H; p.W = W;

    const bool pinhole = K[1] == 0.0 && K[3] == 0.0 && K[6] == 0.0 && K[7] == 0.0 && K[8] == 1.0 &&
                         Kinv[6] == 0.0 && Kinv[7] == 0.0 && Kinv[8] == 1.0;
    const size_t esz = vol_dtype == DFH_F32 ? 4 : 8;
    const bool vec4 = (res[2] % 4 == 0) && ((uintptr_t)tsdf % (4 * esz) == 0) && ((uintptr_t)tsdf_w % (4 * esz) == 0);
    p.zpacks = vec4 ? res[2] / 4 : res[2];
    hipStream_t s = static_cast<hipStream_t>(stream);

#define DFH_DISPATCH(VT, DT)                                                                 \
    return vec4 ? launch_integrate<VT, DT, 4>(tsdf, tsdf_w, depth, p, pinhole, s)            \
                : launch_integrate<VT, DT, 1>(tsdf, tsdf_w, depth, p, pinhole, s)
    if (vol_dtype == DFH_F32) {
        if (depth_dtype == DFH_F32) { DFH_DISPATCH(float, float); }
        DFH_DISPATCH(float, double);
    }
    if (depth_dtype == DFH_F32) { DFH_DISPATCH(double, float); }
    DFH_DISPATCH(double, double);
#undef DFH_DISPATCH
}
