// Device-side restatement of the reference's dual-quaternion warp and TSDF sampler in IEEE
// fp64 with the reference's operation order (library is built with -ffp-contract=off), so
// results agree bit for bit with the fp64 CPU path wherever only +,-,*,/ are involved.
#pragma once
#include "dfh_common.h"

namespace dfh {

struct D3 { double x, y, z; };

// dqb_warp(dq, pos), reference core/util.py:68-72 with quaternion_multiply :255-269 expanded.
// dq = (r | d) w-first.  vq = [1,0,0,0,0,pos]; dqv = dq (x) vq = (r, A + d) where the products
// with vq's exact 0/1 entries vanish; out = (dqv (x) conj(dq))[5:8] = F + S, conj(dq) =
// (w,-x,-y,-z,-d0,d1,d2,d3) (:299-304).  Negations are exact, so signs are folded.
// `pos` must already be float32-representable (the reference rounds it, :69).
__device__ __forceinline__ D3 dqb_warp_exact(const double *q, double px, double py, double pz) {
    const double w1 = q[0], x1 = q[1], y1 = q[2], z1 = q[3];
    const double d0 = q[4], d1 = q[5], d2 = q[6], d3 = q[7];
    // A = quaternion_multiply(r, (0,p)); e = A + d
    const double e0 = ((-x1 * px - y1 * py) - z1 * pz) + d0;
    const double e1 = ((y1 * pz - z1 * py) + w1 * px) + d1;
    const double e2 = ((-x1 * pz) + z1 * px + w1 * py) + d2;
    const double e3 = ((x1 * py - y1 * px) + w1 * pz) + d3;
    // F = quaternion_multiply(r, dc)[1:4], dc = (-d0, d1, d2, d3)
    const double F1 = ((x1 * (-d0) + y1 * d3) - z1 * d2) + w1 * d1;
    const double F2 = ((-x1 * d3 + y1 * (-d0)) + z1 * d1) + w1 * d2;
    const double F3 = ((x1 * d2 - y1 * d1) + z1 * (-d0)) + w1 * d3;
    // S = quaternion_multiply(e, rc)[1:4], rc = (w1, -x1, -y1, -z1)
    const double S1 = ((e1 * w1 - e2 * z1) + e3 * y1) - e0 * x1;
    const double S2 = ((e1 * z1 + e2 * w1) - e3 * x1) - e0 * y1;
    const double S3 = ((-(e1 * y1) + e2 * x1) + e3 * w1) - e0 * z1;
    D3 o;
    o.x = F1 + S1;
    o.y = F2 + S2;
    o.z = F3 + S3;
    return o;
}

// dqb_warp_normal(dq, n), core/util.py:74-76: dual part zeroed, not renormalised.
__device__ __forceinline__ D3 dqb_warp_normal_exact(const double *q, double nx, double ny, double nz) {
    const double w1 = q[0], x1 = q[1], y1 = q[2], z1 = q[3];
    // e = A + 0;  F = products with zeros = 0
    const double e0 = ((-x1 * nx - y1 * ny) - z1 * nz) + 0.0;
    const double e1 = ((y1 * nz - z1 * ny) + w1 * nx) + 0.0;
    const double e2 = ((-x1 * nz) + z1 * nx + w1 * ny) + 0.0;
    const double e3 = ((x1 * ny - y1 * nx) + w1 * nz) + 0.0;
    D3 o;
    o.x = 0.0 + (((e1 * w1 - e2 * z1) + e3 * y1) - e0 * x1);
    o.y = 0.0 + (((e1 * z1 + e2 * w1) - e3 * x1) - e0 * y1);
    o.z = 0.0 + (((-(e1 * y1) + e2 * x1) + e3 * w1) - e0 * z1);
    return o;
}

__device__ __forceinline__ double round_f32(double v) { return (double)(float)v; }

// interpolate_tsdf(pos, tsdf), core/util.py:102-137.  Returns false where the reference
// returns None (:107-108).  x1/y1/z1 = ceil (:113-115); the y-fraction blends the z1 samples
// and the z-fraction the y1 samples (:121-137) -- reproduced, not fixed.
template <typename LiveT>
__device__ __forceinline__ bool interpolate_exact(const LiveT *__restrict__ vol, int RX, int RY, int RZ,
                                                  double px, double py, double pz, double &out) {
    const double mn = fmin(fmin(px, py), pz);
    if (!(mn >= 0.0) || !(px <= (double)(RX - 1)) || !(py <= (double)(RY - 1)) || !(pz <= (double)(RZ - 1))) return false;
    const double fx = floor(px), fy = floor(py), fz = floor(pz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const int x1 = (int)ceil(px), y1 = (int)ceil(py), z1 = (int)ceil(pz);
    const double xd = px - fx, yd = py - fy, zd = pz - fz;
    const size_t sx = (size_t)RY * RZ, sy = (size_t)RZ;
    const double c000 = (double)vol[x0 * sx + y0 * sy + z0];
    const double c100 = (double)vol[x1 * sx + y0 * sy + z0];
    const double c001 = (double)vol[x0 * sx + y1 * sy + z0];
    const double c101 = (double)vol[x1 * sx + y1 * sy + z0];
    const double c010 = (double)vol[x0 * sx + y0 * sy + z1];
    const double c110 = (double)vol[x1 * sx + y0 * sy + z1];
    const double c011 = (double)vol[x0 * sx + y1 * sy + z1];
    const double c111 = (double)vol[x1 * sx + y1 * sy + z1];
    const double c00 = c000 * (1.0 - xd) + c100 * xd;
    const double c01 = c001 * (1.0 - xd) + c101 * xd;
    const double c10 = c010 * (1.0 - xd) + c110 * xd;
    const double c11 = c011 * (1.0 - xd) + c111 * xd;
    const double c0 = c00 * (1.0 - yd) + c10 * yd;
    const double c1 = c01 * (1.0 - yd) + c11 * yd;
    out = c0 * (1.0 - zd) + c1 * zd;
    return true;
}

}  // namespace dfh
