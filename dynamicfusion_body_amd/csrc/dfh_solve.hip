// Warp-field solve.
//
//  * Residual evaluators with the reference's definitions, for parity:
//      dfh_residual_rigid  = FusionDM.computef_lw      (reference core/fusion_dm.py:285-297)
//      dfh_residual_data   = data rows of Fusion.computef / computef_lw (core/fusion.py:444-473)
//      dfh_residual_reg    = regularisation rows of Fusion.computef     (core/fusion.py:475-484)
//  * The Gauss-Newton machinery the reference does not have (it calls scipy's trust-region
//    solver with finite-difference Jacobians, core/fusion.py:382-392): per-sample node search and
//    static blend weights, projective data association with fuseDepths' projection primitives
//    (core/fusion_dm.py:191-200), analytic 6-DoF left-twist Jacobians (derivation in
//    oracle/gn_np.py, pinned against finite differences of the reference's residual), normal
//    equations in 6x6 block-sparse rows, block-Jacobi PCG, and the update dq <- exp(xi) (x) dq.
//
// Everything is fp64: per GN iteration the work is ~1 kflop/sample over ~1e5..1e6 samples, far
// from any roofline that would justify fp32, and fp64 keeps the residual bit-comparable with the
// CPU path.  J^T J accumulation: samples arrive sorted by their k-node tuple; a 256-sample tile
// stages its Jacobian rows in LDS, threads own matrix entries and walk the tile in order, and
// one fp64 atomic per entry and run of equal tuples reaches HBM (no per-sample atomics).
// f32-input MFMA runs at the vector rate on gfx950 and bf16 would break the 1e-4 residual bar,
// so the contraction stays on the VALU (BASELINE north_star: "MFMA only if ...").
#include "dfh_dq.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace dfh {

constexpr int kKMaxS = 8;

struct Q4 { double w, x, y, z; };

__device__ __forceinline__ Q4 qmul(const Q4 &a, const Q4 &b) {
    Q4 o;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return o;
}
__device__ __forceinline__ Q4 qconj(const Q4 &a) { return Q4{a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ Q4 qpure(double x, double y, double z) { return Q4{0.0, x, y, z}; }
__device__ __forceinline__ Q4 qadd(const Q4 &a, const Q4 &b) { return Q4{a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Q4 qscale(const Q4 &a, double s) { return Q4{a.w * s, a.x * s, a.y * s, a.z * s}; }

// Blend the k node DQs of one point (explicit indices, weights from positions exactly like
// Fusion.dq_blend, core/fusion.py:527-551), then warp point (and normal) through the blend and
// m_lw like Fusion.warp (:502-520).  bh receives the normalised blend, wts the raw weights.
__device__ __forceinline__ void blend_from_indices(const double *__restrict__ node_dq, const double *__restrict__ node_pos,
                                                   const double *__restrict__ node_w, const int *idx, int k,
                                                   double px, double py, double pz, double *bh, double *nb_out,
                                                   double *wts) {
    double b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) {
        if (j < k) {
            const int gi = idx[j];
            const double dx = px - node_pos[3 * gi], dy = py - node_pos[3 * gi + 1], dz = pz - node_pos[3 * gi + 2];
            const double dist = sqrt((dx * dx + dy * dy) + dz * dz);
            const double t = dist / (2.0 * node_w[gi]);
            const double wgt = exp(-1.0 * (t * t));
            if (wts) wts[j] = wgt;
#pragma unroll
            for (int c = 0; c < 8; ++c) b[c] = b[c] + wgt * node_dq[8 * gi + c];
        }
    }
    const double n2 = ((b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3])) +
                      ((b[4] * b[4] + b[5] * b[5]) + (b[6] * b[6] + b[7] * b[7]));
    const double n = sqrt(n2);
    if (n == 0.0) {
        bh[0] = 1.0;
#pragma unroll
        for (int c = 1; c < 8; ++c) bh[c] = 0.0;
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) bh[c] = b[c] / n;
    }
    if (nb_out) *nb_out = n;
}

// ------------------------------------------------------------------------------- residuals
__global__ __launch_bounds__(256) void residual_rigid_kernel(const double *__restrict__ verts, const double *__restrict__ norms,
                                                              const double *__restrict__ corr, int n, DQ x,
                                                              double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const D3 wn = dqb_warp_normal_exact(x.q, round_f32(norms[3 * i]), round_f32(norms[3 * i + 1]), round_f32(norms[3 * i + 2]));
    const D3 vp = dqb_warp_exact(x.q, round_f32(verts[3 * i]), round_f32(verts[3 * i + 1]), round_f32(verts[3 * i + 2]));
    const double d0 = vp.x - corr[3 * i], d1 = vp.y - corr[3 * i + 1], d2 = vp.z - corr[3 * i + 2];
    out[i] = (wn.x * d0 + wn.y * d1) + wn.z * d2;                    // fusion_dm.py:293
}

__global__ __launch_bounds__(256) void residual_data_kernel(const double *__restrict__ verts, const double *__restrict__ norms,
                                                             const double *__restrict__ corr, const int *__restrict__ nbr,
                                                             int V, int k, const double *__restrict__ node_dq,
                                                             const double *__restrict__ node_pos,
                                                             const double *__restrict__ node_w, DQ lw,
                                                             double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    int idx[kKMaxS];
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) idx[j] = j < k ? nbr[(size_t)i * k + j] : 0;
    const double px = verts[3 * i], py = verts[3 * i + 1], pz = verts[3 * i + 2];
    double bh[8];
    blend_from_indices(node_dq, node_pos, node_w, idx, k, px, py, pz, bh, nullptr, nullptr);     // fusion.py:508
    const D3 x1 = dqb_warp_exact(bh, round_f32(px), round_f32(py), round_f32(pz));                // :510
    const D3 xp = dqb_warp_exact(lw.q, round_f32(x1.x), round_f32(x1.y), round_f32(x1.z));        // :512
    const D3 n1 = dqb_warp_normal_exact(bh, round_f32(norms[3 * i]), round_f32(norms[3 * i + 1]), round_f32(norms[3 * i + 2]));  // :515
    const D3 np_ = dqb_warp_normal_exact(lw.q, round_f32(n1.x), round_f32(n1.y), round_f32(n1.z));                             // :517
    const double d0 = xp.x - corr[3 * i], d1 = xp.y - corr[3 * i + 1], d2 = xp.z - corr[3 * i + 2];
    out[i] = (np_.x * d0 + np_.y * d1) + np_.z * d2;                 // fusion.py:470
}

__global__ __launch_bounds__(256) void residual_reg_kernel(const int *__restrict__ node_nbr, int N, int k,
                                                            const double *__restrict__ node_dq,
                                                            const double *__restrict__ node_pos,
                                                            const double *__restrict__ node_w, double rw,
                                                            double *__restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * k) return;
    const int i = t / k;
    const int j = node_nbr[t];
    const double vx = round_f32(node_pos[3 * j]), vy = round_f32(node_pos[3 * j + 1]), vz = round_f32(node_pos[3 * j + 2]);
    const D3 yi = dqb_warp_exact(node_dq + 8 * i, vx, vy, vz);
    const D3 yj = dqb_warp_exact(node_dq + 8 * j, vx, vy, vz);
    const double wi = node_w[i], wj = node_w[j];
    const double c = rw * (wi > wj ? wi : wj);                       // rw * max(w_i, w_j), fusion.py:482
    out[3 * t + 0] = c * (yi.x - yj.x);
    out[3 * t + 1] = c * (yi.y - yj.y);
    out[3 * t + 2] = c * (yi.z - yj.z);
}

// Rigid 6-DoF normal equations for the global `_lw`: r_i as above, J_i = [ c_i x m_i | s m_i ]
// (m = warped normal, s = |r_x|^2; derivation in oracle/gn_np.py).  out: 36 (J^T J) + 6 (J^T r)
// + 1 (0.5|r|^2) + 1 (count) doubles.  partial != NULL: every workgroup stores its 29 sums (row blockIdx.x of `partial`) and
// gn_rigid_finish_kernel adds the rows in a fixed order -- same bits every run; partial == NULL (no scratch to be had): one
// atomic per workgroup and entry into `out`.
__global__ __launch_bounds__(256) void gn_build_rigid_kernel(const double *__restrict__ verts, const double *__restrict__ norms,
                                                              const double *__restrict__ corr,
                                                              const unsigned char *__restrict__ valid, int n, DQ x,
                                                              double *__restrict__ out, double *__restrict__ partial) {
    __shared__ double red[256];
    double acc[29];
    for (int e = 0; e < 29; ++e) acc[e] = 0.0;
    const double s = (x.q[0] * x.q[0] + x.q[1] * x.q[1]) + (x.q[2] * x.q[2] + x.q[3] * x.q[3]);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (valid && !valid[i]) continue;
        const D3 m = dqb_warp_normal_exact(x.q, round_f32(norms[3 * i]), round_f32(norms[3 * i + 1]), round_f32(norms[3 * i + 2]));
        const D3 y = dqb_warp_exact(x.q, round_f32(verts[3 * i]), round_f32(verts[3 * i + 1]), round_f32(verts[3 * i + 2]));
        const double c0 = corr[3 * i], c1 = corr[3 * i + 1], c2 = corr[3 * i + 2];
        const double r = (m.x * (y.x - c0) + m.y * (y.y - c1)) + m.z * (y.z - c2);
        const double J[6] = {c1 * m.z - c2 * m.y, c2 * m.x - c0 * m.z, c0 * m.y - c1 * m.x, s * m.x, s * m.y, s * m.z};
        int e = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) acc[e++] += J[a] * J[b];
        for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
        acc[27] += 0.5 * r * r;
        acc[28] += 1.0;
    }
#pragma unroll
    for (int e = 0; e < 29; ++e) {                 // (unrolled: acc[e] with a run-time e would move the 29 sums to scratch memory)
        red[threadIdx.x] = acc[e];
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0 && partial) {
            partial[29 * (size_t)blockIdx.x + e] = red[0];
        } else if (threadIdx.x == 0 && red[0] != 0.0) {
            if (e < 21) {
                int a = 0, rem = e;
                while (rem >= 6 - a) { rem -= 6 - a; ++a; }
                const int b = a + rem;
                atomicAdd(out + 6 * a + b, red[0]);
                if (a != b) atomicAdd(out + 6 * b + a, red[0]);
            } else {
                atomicAdd(out + 36 + (e - 21), red[0]);
            }
        }
        __syncthreads();
    }
}

// out (44 doubles, see above) = the workgroups' 29 sums added in workgroup order (thread t: rows t, t + 256, ...; then a fixed tree)
__global__ __launch_bounds__(256) void gn_rigid_finish_kernel(const double *__restrict__ partial, int rows, double *__restrict__ out) {
    __shared__ double red[256];
    for (int e = 0; e < 29; ++e) {
        double acc = 0.0;
        for (int b = threadIdx.x; b < rows; b += 256) acc += partial[29 * (size_t)b + e];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (e < 21) {
                int a = 0, rem = e;
                while (rem >= 6 - a) { rem -= 6 - a; ++a; }
                const int b = a + rem;
                out[6 * a + b] = red[0];
                out[6 * b + a] = red[0];
            } else {
                out[36 + (e - 21)] = red[0];
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void top8_insert_s(double (&bd)[kKMaxS], int (&bi)[kKMaxS], double d2, int idx) {
    bool ins = false;                       // once inserted, everything below shifts down: equal distances keep their
#pragma unroll
    for (int i = 0; i < kKMaxS; ++i) {      // arrival order (a stable sort: ties go to the lower node index)
        const bool lt = ins || d2 < bd[i];
        ins = lt;
        const double td = bd[i];
        const int ti = bi[i];
        bd[i] = lt ? d2 : td;
        bi[i] = lt ? idx : ti;
        d2 = lt ? td : d2;
        idx = lt ? ti : idx;
    }
}

// ------------------------------------------------------------------------------- batch warp + correspondences
// Fusion.warp for a batch (core/fusion.py:502-520): nbr == NULL -> only the global m_lw is applied
// (the FusionDM case, dqb_warp(_lw, v) / dqb_warp_normal(_lw, n), fusion_dm.py:230-231).
__global__ __launch_bounds__(256) void warp_points_kernel(const double *__restrict__ verts, const double *__restrict__ norms,
                                                           const int *__restrict__ nbr, int V, int k,
                                                           const double *__restrict__ node_dq,
                                                           const double *__restrict__ node_pos,
                                                           const double *__restrict__ node_w, DQ lw,
                                                           double *__restrict__ out_pos, double *__restrict__ out_nrm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    double px = verts[3 * (size_t)i], py = verts[3 * (size_t)i + 1], pz = verts[3 * (size_t)i + 2];
    double nx = norms ? norms[3 * (size_t)i] : 0.0, ny = norms ? norms[3 * (size_t)i + 1] : 0.0, nz = norms ? norms[3 * (size_t)i + 2] : 0.0;
    if (nbr) {
        int idx[kKMaxS];
#pragma unroll
        for (int j = 0; j < kKMaxS; ++j) idx[j] = j < k ? nbr[(size_t)i * k + j] : 0;
        double bh[8];
        blend_from_indices(node_dq, node_pos, node_w, idx, k, px, py, pz, bh, nullptr, nullptr);
        const D3 x1 = dqb_warp_exact(bh, round_f32(px), round_f32(py), round_f32(pz));
        const D3 n1 = dqb_warp_normal_exact(bh, round_f32(nx), round_f32(ny), round_f32(nz));
        px = x1.x; py = x1.y; pz = x1.z; nx = n1.x; ny = n1.y; nz = n1.z;
    }
    const D3 xp = dqb_warp_exact(lw.q, round_f32(px), round_f32(py), round_f32(pz));
    out_pos[3 * (size_t)i] = xp.x; out_pos[3 * (size_t)i + 1] = xp.y; out_pos[3 * (size_t)i + 2] = xp.z;
    if (out_nrm) {
        const D3 np_ = dqb_warp_normal_exact(lw.q, round_f32(nx), round_f32(ny), round_f32(nz));
        out_nrm[3 * (size_t)i] = np_.x; out_nrm[3 * (size_t)i + 1] = np_.y; out_nrm[3 * (size_t)i + 2] = np_.z;
    }
}

// The selection loop of setupCorrespondences (core/fusion_dm.py:229-244, core/fusion.py:258-276):
// k nearest live vertices of every warped vertex (brute force through LDS tiles, nearest first as
// KDTree.query returns them), best = first neighbour with the smallest cost |wn.(vp - p)| below the
// initial best_cost = 1, kept iff best_cost <= tolerance.
__global__ __launch_bounds__(256) void closest_corr_kernel(const double *__restrict__ wpos, const double *__restrict__ wnrm, int V,
                                                            const double *__restrict__ live, int L, int k, double tolerance,
                                                            double *__restrict__ corr, double *__restrict__ cost_out,
                                                            unsigned char *__restrict__ keep) {
    __shared__ double sp[256 * 3];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < V;
    const double px = act ? wpos[3 * (size_t)i] : 0.0, py = act ? wpos[3 * (size_t)i + 1] : 0.0, pz = act ? wpos[3 * (size_t)i + 2] : 0.0;
    double bd[kKMaxS];
    int bi[kKMaxS];
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) { bd[j] = __builtin_huge_val(); bi[j] = -1; }
    for (int base = 0; base < L; base += 256) {
        const int n = min(256, L - base);
        if ((int)threadIdx.x < n) {
            sp[3 * threadIdx.x] = live[3 * (size_t)(base + threadIdx.x)];
            sp[3 * threadIdx.x + 1] = live[3 * (size_t)(base + threadIdx.x) + 1];
            sp[3 * threadIdx.x + 2] = live[3 * (size_t)(base + threadIdx.x) + 2];
        }
        __syncthreads();
        if (act) {
            for (int j = 0; j < n; ++j) {
                const double dx = px - sp[3 * j], dy = py - sp[3 * j + 1], dz = pz - sp[3 * j + 2];
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < bd[kKMaxS - 1]) top8_insert_s(bd, bi, d2, base + j);
            }
        }
        __syncthreads();
    }
    if (!act) return;
    const double nx = wnrm[3 * (size_t)i], ny = wnrm[3 * (size_t)i + 1], nz = wnrm[3 * (size_t)i + 2];
    double best_cost = 1.0;                                     // fusion_dm.py:234
    int best = bi[0];                                           // lverts[nidxs[0]], :233
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) {
        if (j < k) {
            const int q = bi[j];
            const double dx = px - live[3 * (size_t)q], dy = py - live[3 * (size_t)q + 1], dz = pz - live[3 * (size_t)q + 2];
            const double c = fabs((nx * dx + ny * dy) + nz * dz);   // :238
            if (c < best_cost) { best_cost = c; best = q; }
        }
    }
    corr[3 * (size_t)i] = live[3 * (size_t)best];
    corr[3 * (size_t)i + 1] = live[3 * (size_t)best + 1];
    corr[3 * (size_t)i + 2] = live[3 * (size_t)best + 2];
    if (cost_out) cost_out[i] = best_cost;
    keep[i] = best_cost <= tolerance ? 1 : 0;                    // :242
}

// ------------------------------------------------------------------------------- deformation-graph maintenance
// Device side of update_graph / construct_graph (reference core/fusion.py:101-123, 201-239).

// Nearest cloud point of every query (KDTree(cloud).query(q), :209-212: a node's anchor vertex): one workgroup per
// query, threads stride over the cloud, lexicographic (d2, index) minimum -- ties go to the lower index.
__global__ __launch_bounds__(256) void nearest_point_kernel(const double *__restrict__ query, int nq, const double *__restrict__ cloud,
                                                             int nc, int *__restrict__ idx_out, double *__restrict__ d2_out) {
    __shared__ double sd[256];
    __shared__ int si[256];
    const int qi = blockIdx.x;
    const double qx = query[3 * (size_t)qi], qy = query[3 * (size_t)qi + 1], qz = query[3 * (size_t)qi + 2];
    double best = __builtin_huge_val();
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < nc; j += 256) {
        const double dx = qx - cloud[3 * (size_t)j], dy = qy - cloud[3 * (size_t)j + 1], dz = qz - cloud[3 * (size_t)j + 2];
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < best) { best = d2; bi = j; }            // (ascending j per thread: the first minimum is kept)
    }
    sd[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            const double o = sd[threadIdx.x + st];
            const int oi = si[threadIdx.x + st];
            if (o < sd[threadIdx.x] || (o == sd[threadIdx.x] && oi < si[threadIdx.x])) { sd[threadIdx.x] = o; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        idx_out[qi] = si[0];
        if (d2_out) d2_out[qi] = sd[0];
    }
}

// "unsupported surface point" test of update_graph (:215-219): min over the vertex's knn nodes of |node - v| / w >= 1
__global__ __launch_bounds__(256) void graph_unsupported_kernel(const double *__restrict__ verts, int V, const int *__restrict__ nbr, int k,
                                                                 const double *__restrict__ node_pos, const double *__restrict__ node_w,
                                                                 unsigned char *__restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    const double px = verts[3 * (size_t)i], py = verts[3 * (size_t)i + 1], pz = verts[3 * (size_t)i + 2];
    double m = __builtin_huge_val();
    for (int j = 0; j < k; ++j) {
        const int gi = nbr[(size_t)i * k + j];
        const double dx = node_pos[3 * gi] - px, dy = node_pos[3 * gi + 1] - py, dz = node_pos[3 * gi + 2] - pz;
        const double r = sqrt((dx * dx + dy * dy) + dz * dz) / node_w[gi];
        m = r < m ? r : m;
    }
    flag[i] = m >= 1.0 ? 1 : 0;
}

// Fusion.dq_blend for a batch (:527-551): the normalised blend of the given nodes' DQs at every point (identity when
// the blend vanishes) -- the DQ a newly inserted node starts from (:222).
__global__ __launch_bounds__(256) void dq_blend_points_kernel(const double *__restrict__ pts, int P, const int *__restrict__ nbr, int k,
                                                               const double *__restrict__ node_dq, const double *__restrict__ node_pos,
                                                               const double *__restrict__ node_w, double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    int idx[kKMaxS];
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) idx[j] = j < k ? nbr[(size_t)i * k + j] : 0;
    double bh[8];
    blend_from_indices(node_dq, node_pos, node_w, idx, k, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], bh, nullptr, nullptr);
#pragma unroll
    for (int c = 0; c < 8; ++c) out[8 * (size_t)i + c] = bh[c];
}

// ------------------------------------------------------------------------------- sample setup

// k nearest nodes + Gaussian blend weights of arbitrary sample points.  The 256 samples of a workgroup are
// consecutive band voxels, i.e. spatially coherent: with their bounding box B, any sample's k-th nearest node is no
// farther than the k-th smallest over nodes of maxdist(node, B), so only nodes with mindist(node, B) within that bound
// can be among anyone's k nearest.  Those candidates (kept in node order, so ties resolve as in a full scan) are
// scanned; everything else is skipped.  Same result as brute force, ~10x fewer distance evaluations.
constexpr int kKnnCand = 512;              // candidate capacity in LDS; more -> plain scan of all nodes

__global__ __launch_bounds__(256) void sample_knn_kernel(const double *__restrict__ spos, int S, const double *__restrict__ node_pos,
                                                          const double *__restrict__ node_w, int N, int k,
                                                          int *__restrict__ nbr, double *__restrict__ wts) {
    __shared__ double sp[kKnnCand * 3];
    __shared__ int sid[kKnnCand];
    __shared__ double sred[6][4];
    __shared__ double sbox[6];
    __shared__ int scount[5];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = blockIdx.x * 256 + tid;
    const bool act = i < S;
    const double px = act ? spos[3 * (size_t)i] : 0.0, py = act ? spos[3 * (size_t)i + 1] : 0.0, pz = act ? spos[3 * (size_t)i + 2] : 0.0;
    // ---- bounding box of the workgroup's samples
    {
        const double big = __builtin_huge_val();
        double v[6] = {act ? px : big, act ? py : big, act ? pz : big, act ? -px : big, act ? -py : big, act ? -pz : big};   // min of (x, -x)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v[c] = fmin(v[c], __shfl_xor(v[c], o, 64));
            if (lane == 0) sred[c][wv] = v[c];
        }
        __syncthreads();
        if (tid < 6) sbox[tid] = fmin(fmin(sred[tid][0], sred[tid][1]), fmin(sred[tid][2], sred[tid][3]));
        __syncthreads();
    }
    const double lox = sbox[0], loy = sbox[1], loz = sbox[2], hix = -sbox[3], hiy = -sbox[4], hiz = -sbox[5];
    // ---- bound: k-th smallest maxdist^2(node, box); k rounds of "smallest value above the previous one" (ties make the
    //      bound only larger, which is safe)
    auto maxd2 = [&](int n) {
        const double x = node_pos[3 * n], y = node_pos[3 * n + 1], z = node_pos[3 * n + 2];
        const double dx = fmax(fabs(x - lox), fabs(x - hix)), dy = fmax(fabs(y - loy), fabs(y - hiy)), dz = fmax(fabs(z - loz), fabs(z - hiz));
        return (dx * dx + dy * dy) + dz * dz;
    };
    double prev = -1.0;
    for (int r = 0; r < k; ++r) {
        double m = __builtin_huge_val();
        for (int n = tid; n < N; n += 256) {
            const double d = maxd2(n);
            if (d > prev && d < m) m = d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
        if (lane == 0) sred[0][wv] = m;
        __syncthreads();
        prev = fmin(fmin(sred[0][0], sred[0][1]), fmin(sred[0][2], sred[0][3]));
        __syncthreads();
    }
    // (k distinct values were found when N >= k distinct distances exist; with fewer, prev = +inf: every node qualifies)
    const double bound = prev * (1.0 + 1e-12) + 1e-300;
    // ---- candidates: mindist^2(node, box) <= bound, compacted in node order
    int total = 0;
    bool fits = true;
    for (int base = 0; base < N && fits; base += 256) {
        const int n = base + tid;
        bool keep = false;
        if (n < N) {
            const double x = node_pos[3 * n], y = node_pos[3 * n + 1], z = node_pos[3 * n + 2];
            const double dx = fmax(fmax(lox - x, x - hix), 0.0), dy = fmax(fmax(loy - y, y - hiy), 0.0), dz = fmax(fmax(loz - z, z - hiz), 0.0);
            keep = (dx * dx + dy * dy) + dz * dz <= bound;
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) scount[wv] = __popcll(bal);
        __syncthreads();
        int pos = total + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; ++w) pos += scount[w];
        const int add = scount[0] + scount[1] + scount[2] + scount[3];
        if (total + add > kKnnCand) fits = false;                    // (block-uniform)
        else if (keep) {
            sid[pos] = n;
            sp[3 * pos] = node_pos[3 * n]; sp[3 * pos + 1] = node_pos[3 * n + 1]; sp[3 * pos + 2] = node_pos[3 * n + 2];
        }
        total += add;
        __syncthreads();
    }
    double bd[kKMaxS];
    int bi[kKMaxS];
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) { bd[j] = __builtin_huge_val(); bi[j] = -1; }
    if (fits) {
        if (act) {
            for (int j = 0; j < total; ++j) {
                const double dx = px - sp[3 * j], dy = py - sp[3 * j + 1], dz = pz - sp[3 * j + 2];
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < bd[kKMaxS - 1]) top8_insert_s(bd, bi, d2, sid[j]);
            }
        }
    } else {
        for (int base = 0; base < N; base += 256) {                 // too many candidates for LDS: scan all nodes
            const int n = min(256, N - base);
            __syncthreads();
            if (tid < n) {
                sp[3 * tid] = node_pos[3 * (base + tid)];
                sp[3 * tid + 1] = node_pos[3 * (base + tid) + 1];
                sp[3 * tid + 2] = node_pos[3 * (base + tid) + 2];
            }
            __syncthreads();
            if (act) {
                for (int j = 0; j < n; ++j) {
                    const double dx = px - sp[3 * j], dy = py - sp[3 * j + 1], dz = pz - sp[3 * j + 2];
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < bd[kKMaxS - 1]) top8_insert_s(bd, bi, d2, base + j);
                }
            }
        }
    }
    if (!act) return;
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) {
        if (j < k) {
            const int gi = bi[j];
            nbr[(size_t)i * k + j] = gi;
            const double t = sqrt(bd[j]) / (2.0 * node_w[gi]);
            wts[(size_t)i * k + j] = exp(-1.0 * (t * t));
        }
    }
}

// Samples into the order of `order` (the sort by node tuple): positions, normals, node ids and blend weights in one pass.
__global__ __launch_bounds__(256) void permute_samples_kernel(const long *__restrict__ order, int S, int k, const double *__restrict__ pos,
                                                              const double *__restrict__ nrm, const int *__restrict__ nbr,
                                                              const double *__restrict__ wts, double *__restrict__ pos_o,
                                                              double *__restrict__ nrm_o, int *__restrict__ nbr_o, double *__restrict__ wts_o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const size_t src = (size_t)order[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pos_o[3 * (size_t)i + c] = pos[3 * src + c];
        nrm_o[3 * (size_t)i + c] = nrm[3 * src + c];
    }
    for (int j = 0; j < k; ++j) {
        nbr_o[(size_t)i * k + j] = nbr[src * k + j];
        wts_o[(size_t)i * k + j] = wts[src * k + j];
    }
}

// ------------------------------------------------------------------------------- association
struct AssocParams {
    Mat3 K, Kinv;
    Mat34 lw_cam;
    Mat3 Rinv;            // inverse of lw_cam's 3x3 part
    DQ lw;
    double scale, inv_scale, cx, cy, cz, half, max_dist;
    int H, W, k;
};

// normalised blend of a sample's k node DQs with its static weights (identity when the blend vanishes); returns |b|_8
// (1 in the degenerate case)
__device__ __forceinline__ double blend_static(const double *__restrict__ node_dq, const int *idx, const double *w, int k, double *bh) {
    double b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) {
        if (j < k) {
#pragma unroll
            for (int c = 0; c < 8; ++c) b[c] = b[c] + w[j] * node_dq[8 * idx[j] + c];
        }
    }
    double nb = sqrt(((b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3])) +
                     ((b[4] * b[4] + b[5] * b[5]) + (b[6] * b[6] + b[7] * b[7])));
    if (nb == 0.0) { bh[0] = 1.0; for (int c = 1; c < 8; ++c) bh[c] = 0.0; nb = 1.0; }
    else {
        const double inv = 1.0 / nb;                      // one division, eight products (each within 1 ulp of b / nb): the GN path has
        for (int c = 0; c < 8; ++c) bh[c] = b[c] * inv;   // no reference rounding to meet (the residual evaluators use blend_from_indices)
    }
    return nb;
}

// One live view of a frame in device memory (dfh_gn_pack_views): extrinsic, the inverse of its 3x3 part, the depth map.
struct AssocView {
    double lw_cam[12];
    double Rinv[9];
    const void *depth;
    const float *cells;      // per 16 x 16-pixel cell {smallest, largest valid z = -depth} (dfh_gn_pack_views_cells); null: none
    double cull_ok;          // 1: this view's extrinsic is a rigid motion (the depth-interval test below is exact for it)
};                           // 192 bytes
static_assert(sizeof(AssocView) == 192, "AssocView is a 192-byte record");
constexpr int kCellPx = 16;

// Projective association of one warped sample xp (index space) against ONE view: project with the reference's primitives,
// take the nearest depth pixel, back-project.  Returns validity (before the distance gate); c = correspondence in index
// space, d2 = its squared distance from xp.
// The two halves of it: up to the pixel (no memory access), and from the pixel's depth on.  associate_views runs the first
// half for several views, asks for their depth values together and only then goes on: one memory round trip per group of
// views instead of one per view.
__device__ __forceinline__ bool associate_project(const AssocParams &p, const double *lw, const D3 &xp, double &u, double &v) {
    // index -> world -> camera -> pixel (fusion_dm.py:191-195)
    const double wx = p.scale * (xp.x - p.half) + p.cx, wy = p.scale * (xp.y - p.half) + p.cy, wz = p.scale * (xp.z - p.half) + p.cz;
    const double l0 = ((lw[0] * wx + lw[1] * wy) + lw[2] * wz) + lw[3];
    const double l1 = ((lw[4] * wx + lw[5] * wy) + lw[6] * wz) + lw[7];
    const double l2 = ((lw[8] * wx + lw[9] * wy) + lw[10] * wz) + lw[11];
    const double p0 = (p.K.m[0] * l0 + p.K.m[1] * l1) + p.K.m[2] * l2;
    const double p1 = (p.K.m[3] * l0 + p.K.m[4] * l1) + p.K.m[5] * l2;
    const double p2 = (p.K.m[6] * l0 + p.K.m[7] * l1) + p.K.m[8] * l2;
    bool ok = p2 != 0.0;
    // one corrected reciprocal instead of two IEEE divisions (u, v within 1.5 ulp: the association has no reference counterpart
    // whose rounding would have to be met; the oracle comparison is to 1e-9)
    double rp = __builtin_amdgcn_rcp(p2);
    rp = __builtin_fma(rp, __builtin_fma(-p2, rp, 1.0), rp);
    rp = __builtin_fma(rp, __builtin_fma(-p2, rp, 1.0), rp);
    u = p0 * rp; v = p1 * rp;
    return ok && (u >= 0.0) && (u < (double)(p.W - 1)) && (v >= 0.0) && (v < (double)(p.H - 1));
}

// z = -depth[rint(v)][rint(u)] (:196) -> validity, correspondence c in index space, its squared distance d2 from xp
__device__ __forceinline__ bool associate_backproject(const AssocParams &p, const double *lw, const double *Rinv, double z, double u, double v,
                                                      const D3 &xp, double &c0, double &c1, double &c2, double &d2) {
    // back-projection K^-1 (z [u,v,1]) (:198-200), camera -> world -> index
    const double a0 = z * u, a1 = z * v, a2 = z * 1.0;
    const double q0 = (p.Kinv.m[0] * a0 + p.Kinv.m[1] * a1) + p.Kinv.m[2] * a2 - lw[3];
    const double q1 = (p.Kinv.m[3] * a0 + p.Kinv.m[4] * a1) + p.Kinv.m[5] * a2 - lw[7];
    const double q2 = (p.Kinv.m[6] * a0 + p.Kinv.m[7] * a1) + p.Kinv.m[8] * a2 - lw[11];
    const double X = (Rinv[0] * q0 + Rinv[1] * q1) + Rinv[2] * q2;
    const double Y = (Rinv[3] * q0 + Rinv[4] * q1) + Rinv[5] * q2;
    const double Z = (Rinv[6] * q0 + Rinv[7] * q1) + Rinv[8] * q2;
    c0 = (X - p.cx) * p.inv_scale + p.half;
    c1 = (Y - p.cy) * p.inv_scale + p.half;
    c2 = (Z - p.cz) * p.inv_scale + p.half;
    const double dx = c0 - xp.x, dy = c1 - xp.y, dz = c2 - xp.z;
    d2 = dx * dx + dy * dy + dz * dz;
    return z > 0.0;
}

template <typename DepthT>
__device__ __forceinline__ bool associate_view(const AssocParams &p, const double *lw, const double *Rinv, const DepthT *__restrict__ depth,
                                               const D3 &xp, double &c0, double &c1, double &c2, double &d2) {
    double u, v;
    bool ok = associate_project(p, lw, xp, u, v);
    c0 = 0.0; c1 = 0.0; c2 = 0.0; d2 = 0.0;
    if (ok) {
        const int ui = (int)rint(u), vi = (int)rint(v);
        const double z = -1.0 * (double)depth[(size_t)vi * p.W + ui];                  // :196
        ok = associate_backproject(p, lw, Rinv, z, u, v, xp, c0, c1, c2, d2);
    }
    return ok;
}

// One view (the parameters inside p): validity includes the distance gate; c = 0 when invalid.
template <typename DepthT>
__device__ __forceinline__ bool associate_point(const AssocParams &p, const DepthT *__restrict__ depth, const D3 &xp, double (&c)[3]) {
    double c0, c1, c2, d2;
    bool ok = associate_view<DepthT>(p, p.lw_cam.m, p.Rinv.m, depth, xp, c0, c1, c2, d2);
    if (ok && p.max_dist > 0.0) ok = d2 <= p.max_dist * p.max_dist;
    c[0] = ok ? c0 : 0.0; c[1] = ok ? c1 : 0.0; c[2] = ok ? c2 : 0.0;
    return ok;
}

// Several views (BASELINE config 5: the live frame is eight depth maps): every view is tried in turn, the sample keeps the
// correspondence of the view in which it lies CLOSEST to the observed surface (smallest |c - x'|; the gate is applied per
// view; ties go to the lower view index) -- one data row per sample, as with one view, so the block pattern and the plan do
// not depend on the number of views.  The reference has no counterpart: its correspondences are mesh-to-mesh
// (core/fusion.py:255-276); restated in oracle/gn_np.py:associate_depth_views.
// view_mask (wave-uniform): the views to try, bit v = view v (all of them: ~0u).  A tile of the fused build passes the views
// its samples can possibly be valid in (tile_view_mask below): the others are not even projected.
template <typename DepthT>
__device__ __forceinline__ bool associate_views(const AssocParams &p, const AssocView *__restrict__ views, int n_views, const D3 &xp,
                                                double (&c)[3], unsigned view_mask = ~0u) {
    bool any = false;
    double best = __builtin_huge_val();
    c[0] = 0.0; c[1] = 0.0; c[2] = 0.0;
    constexpr int G = 4;                                   // views per group: their depth gathers are in flight together
    unsigned todo = view_mask & (n_views >= 32 ? ~0u : ((1u << n_views) - 1u));
    while (todo) {                                         // (uniform: the views' parameters come through scalar loads)
        double u[G], vv[G], z[G];
        bool ok[G];
        int vi_[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            ok[j] = false; z[j] = 0.0; u[j] = 0.0; vv[j] = 0.0;
            vi_[j] = -1;
            if (todo) {
                const int v = __builtin_ctz(todo);         // (ascending: the surviving views in view order)
                todo &= todo - 1u;
                vi_[j] = v;
                ok[j] = associate_project(p, views[v].lw_cam, xp, u[j], vv[j]);
                if (ok[j]) {
                    const int ui = (int)rint(u[j]), vi = (int)rint(vv[j]);
                    z[j] = -1.0 * (double)static_cast<const DepthT *>(views[v].depth)[(size_t)vi * p.W + ui];     // :196
                }
            }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {                      // (in view order: ties go to the lower index)
            if (ok[j]) {
                double c0, c1, c2, d2;
                bool good = associate_backproject(p, views[vi_[j]].lw_cam, views[vi_[j]].Rinv, z[j], u[j], vv[j], xp, c0, c1, c2, d2);
                if (good && p.max_dist > 0.0) good = d2 <= p.max_dist * p.max_dist;
                if (good && d2 < best) { best = d2; c[0] = c0; c[1] = c1; c[2] = c2; any = true; }
            }
        }
    }
    return any;
}

// Which views can hold a valid correspondence for ANY sample of a tile (round 4; exact: a dropped view yields none).
// A tile's samples share a node tuple, so their warped positions fill a small box B.  For a view with a rigid extrinsic and a
// pinhole K (K^-1's last row = (0, 0, 1)): a correspondence c is the back-projection of a pixel at camera depth z, the sample
// x' has camera depth l2(x'), and |c - x'| (index units) = |c_cam - l| / scale >= |z - l2| / scale.  With B in front of the
// camera its image lies inside the bounding rectangle of its eight projected corners and l2 over B inside the corners' range
// [l2min, l2max] (affine).  The view is dropped when the rectangle misses [0, W-1) x [0, H-1), or the pixels it can round to
// hold no valid depth, or their valid depths [zmin, zmax] (a table of 16 x 16-pixel cells, dfh_gn_pack_views_cells) stay
// further than max_dist from [l2min, l2max].  Thread t of the tile takes corner t & 7 of view t >> 3 (n_views <= 16).
// All kTile threads call this; box = {xmin, xmax, ymin, ymax, zmin, zmax} of the tile's warped samples (an empty tile: min > max).
__device__ __forceinline__ unsigned tile_view_mask(const AssocParams &p, const AssocView *__restrict__ views, int n_views, const double (&box)[6],
                                                   unsigned *s_mask) {
    const int t = threadIdx.x;
    if (t == 0) *s_mask = 0u;
    __syncthreads();
    const int v = t >> 3, corner = t & 7;
    if (v < n_views) {                                                         // (whole groups of eight lanes)
        const AssocView &vw = views[v];
        const double *lw = vw.lw_cam;
        const D3 xp{(corner & 1) ? box[1] : box[0], (corner & 2) ? box[3] : box[2], (corner & 4) ? box[5] : box[4]};
        const double wx = p.scale * (xp.x - p.half) + p.cx, wy = p.scale * (xp.y - p.half) + p.cy, wz = p.scale * (xp.z - p.half) + p.cz;
        const double l0 = ((lw[0] * wx + lw[1] * wy) + lw[2] * wz) + lw[3];
        const double l1 = ((lw[4] * wx + lw[5] * wy) + lw[6] * wz) + lw[7];
        const double l2 = ((lw[8] * wx + lw[9] * wy) + lw[10] * wz) + lw[11];
        const double p0 = (p.K.m[0] * l0 + p.K.m[1] * l1) + p.K.m[2] * l2;
        const double p1 = (p.K.m[3] * l0 + p.K.m[4] * l1) + p.K.m[5] * l2;
        const double p2 = (p.K.m[6] * l0 + p.K.m[7] * l1) + p.K.m[8] * l2;
        bool front = p2 > 1e-9 && l2 > 1e-9;
        const double u = front ? p0 / p2 : 0.0, vv = front ? p1 / p2 : 0.0;
        double umin = u, umax = u, vmin = vv, vmax = vv, lmin = l2, lmax = l2;
#pragma unroll
        for (int o = 1; o <= 4; o <<= 1) {
            umin = fmin(umin, __shfl_xor(umin, o, 8)); umax = fmax(umax, __shfl_xor(umax, o, 8));
            vmin = fmin(vmin, __shfl_xor(vmin, o, 8)); vmax = fmax(vmax, __shfl_xor(vmax, o, 8));
            lmin = fmin(lmin, __shfl_xor(lmin, o, 8)); lmax = fmax(lmax, __shfl_xor(lmax, o, 8));
            front = front & (__shfl_xor(front ? 1 : 0, o, 8) != 0);
        }
        bool keep = true;
        const bool can = front && vw.cells != nullptr && vw.cull_ok == 1.0 && box[0] <= box[1] && p.max_dist > 0.0 &&
                         p.Kinv.m[6] == 0.0 && p.Kinv.m[7] == 0.0 && p.Kinv.m[8] == 1.0;
        if (can) {
            const double eps = 1e-6;                                           // pixels: the corners' own rounding is ~1e-12
            // samples are valid only for 0 <= u < W - 1, 0 <= v < H - 1 (associate_project)
            const double ua = fmax(umin - eps, 0.0), ub = fmin(umax + eps, (double)(p.W - 1));
            const double va = fmax(vmin - eps, 0.0), vb = fmin(vmax + eps, (double)(p.H - 1));
            if (ua > ub || va > vb) {
                keep = false;                                                  // the whole box projects outside the image
            } else {
                // pixels the samples can round to: [floor(ua), ceil(ub)] x [floor(va), ceil(vb)], inside the image
                const int x0 = (int)floor(ua), x1 = min((int)ceil(ub), p.W - 1), y0 = (int)floor(va), y1 = min((int)ceil(vb), p.H - 1);
                const int cx0 = x0 / kCellPx, cx1 = x1 / kCellPx, cy0 = y0 / kCellPx, cy1 = y1 / kCellPx;
                const int nx = cx1 - cx0 + 1, ncell = nx * (cy1 - cy0 + 1), ncx = (p.W + kCellPx - 1) / kCellPx;
                if (ncell <= 64) {                                             // (a larger footprint: keep the view)
                    float zlo = __builtin_huge_valf(), zhi = 0.0f;
                    for (int i = corner; i < ncell; i += 8) {
                        const int cy = cy0 + i / nx, cx = cx0 + i % nx;
                        const float2 mm = *reinterpret_cast<const float2 *>(vw.cells + 2 * ((size_t)cy * ncx + cx));
                        zlo = fminf(zlo, mm.x); zhi = fmaxf(zhi, mm.y);
                    }
#pragma unroll
                    for (int o = 1; o <= 4; o <<= 1) { zlo = fminf(zlo, __shfl_xor(zlo, o, 8)); zhi = fmaxf(zhi, __shfl_xor(zhi, o, 8)); }
                    const double md = p.max_dist * fabs(p.scale) * (1.0 + 1e-6) + 1e-9 * (1.0 + lmax);
                    if (!(zhi > 0.0f) || zlo > zhi) keep = false;              // no valid pixel under the box
                    else if (lmin - (double)zhi > md || (double)zlo - lmax > md) keep = false;
                }
            }
        }
        if (corner == 0 && keep) atomicOr(s_mask, 1u << v);
    }
    __syncthreads();
    return *s_mask;
}

// Warp every sample with the current field, project it into the live depth frame with the
// reference's primitives and back-project the nearest depth pixel: corr (index space), valid.
template <typename DepthT>
__global__ __launch_bounds__(256) void associate_kernel(const double *__restrict__ spos, const int *__restrict__ nbr,
                                                         const double *__restrict__ wts, int S,
                                                         const double *__restrict__ node_dq, const DepthT *__restrict__ depth,
                                                         const AssocParams p, double *__restrict__ corr,
                                                         unsigned char *__restrict__ valid, const AssocView *__restrict__ views, int n_views) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    int idx[kKMaxS];
    double w[kKMaxS];
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) {
        idx[j] = j < p.k ? nbr[(size_t)i * p.k + j] : 0;
        w[j] = j < p.k ? wts[(size_t)i * p.k + j] : 0.0;
    }
    double b[8];
    blend_static(node_dq, idx, w, p.k, b);
    const double px = spos[3 * (size_t)i], py = spos[3 * (size_t)i + 1], pz = spos[3 * (size_t)i + 2];
    const D3 x1 = dqb_warp_exact(b, round_f32(px), round_f32(py), round_f32(pz));
    const D3 xp = dqb_warp_exact(p.lw.q, round_f32(x1.x), round_f32(x1.y), round_f32(x1.z));
    double c[3];
    const bool ok = views ? associate_views<DepthT>(p, views, n_views, xp, c) : associate_point<DepthT>(p, depth, xp, c);
    corr[3 * (size_t)i] = c[0];
    corr[3 * (size_t)i + 1] = c[1];
    corr[3 * (size_t)i + 2] = c[2];
    valid[i] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------- normal equations
// Block-sparse rows: node a owns blocks vals[row_ptr[a] .. row_ptr[a+1]) with sorted column
// nodes col[]; a block is 36 doubles, row-major 6x6.
__device__ __forceinline__ int find_block(const int *__restrict__ row_ptr, const int *__restrict__ col, int a, int b) {
    int lo = row_ptr[a], hi = row_ptr[a + 1] - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int c = col[mid];
        if (c == b) return mid;
        if (c < b) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

struct BuildParams {
    DQ lw;
    int S, k, N;
    double huber;             // > 0: IRLS weight min(1, huber / |r|) on the data rows (the reference's solver runs
};                            //      least_squares(loss='huber'), f_scale 1: core/fusion.py:389); 0: plain least squares

constexpr int kTile = kGnTile;                  // samples (= threads) per tile, dfh_common.h
constexpr int kTileWaves = kTile / 64;

// scratch row of the planned build: {Gram matrix of the row's 6K Jacobian columns as 6x6 sub-blocks (slot sa <= slot sb), each
// stored WHOLE and row-major (36 contiguous doubles; the diagonal ones with both triangles) | J^T r | cost | count | live flag},
// padded to whole 64-byte lines.  The gather reads one sub-block per list entry: 288 contiguous bytes instead of 36 values strewn
// over a packed 24 x 24 triangle (6-12 cache lines) -- its traffic was 9x the live rows' size.
__host__ __device__ constexpr int gn_nsub(int K) { return K * (K + 1) / 2; }
__host__ __device__ constexpr int gn_sub(int K, int sa, int sb) { return sa * K - (sa * (sa - 1)) / 2 + (sb - sa); }   // sa <= sb
__host__ __device__ constexpr int gn_row_gram(int K) { return 36 * gn_nsub(K); }
__host__ __device__ constexpr int gn_row_entries(int K) { return gn_row_gram(K) + 6 * K + 2; }
__host__ __device__ constexpr int gn_row_stride(int K) { return (gn_row_entries(K) + 1 + 7) / 8 * 8; }
// element (ia, ib) of the Gram sub-block for tuple slots (sa, sb), any order, inside a scratch row
__host__ __device__ constexpr int gn_gram_index(int K, int sa, int sb, int ia, int ib) {
    return sa <= sb ? 36 * gn_sub(K, sa, sb) + 6 * ia + ib : 36 * gn_sub(K, sb, sa) + 6 * ib + ia;
}

// Residual and 6-DoF Jacobian rows of one data sample (formulas: oracle/gn_np.py
// data_residual_jacobian).  J is written as k x 6 into Jrow (row-major), returns r.
// second half of data_row: from the normalised blend bh (|b|_8 = nb), the float32-rounded point pf and the warped point xp
__device__ __forceinline__ double data_row_from(const double *__restrict__ node_dq, const int *idx, const double *w, int k,
                                                const double *lwq, const double *bh, double nb, double pfx, double pfy, double pfz,
                                                const D3 &xp, double nx, double ny, double nz, double c0, double c1, double c2,
                                                double *Jrow);

__device__ __forceinline__ double data_row(const double *__restrict__ node_dq, const int *idx, const double *w, int k,
                                           const double *lwq, double px, double py, double pz, double nx, double ny,
                                           double nz, double c0, double c1, double c2, double *Jrow) {
    double bh[8];
    const double nb = blend_static(node_dq, idx, w, k, bh);
    const double pfx = round_f32(px), pfy = round_f32(py), pfz = round_f32(pz);
    const D3 x1 = dqb_warp_exact(bh, pfx, pfy, pfz);
    const D3 xp = dqb_warp_exact(lwq, round_f32(x1.x), round_f32(x1.y), round_f32(x1.z));
    return data_row_from(node_dq, idx, w, k, lwq, bh, nb, pfx, pfy, pfz, xp, nx, ny, nz, c0, c1, c2, Jrow);
}

__device__ __forceinline__ double data_row_from(const double *__restrict__ node_dq, const int *idx, const double *w, int k,
                                                const double *lwq, const double *bh, double nb, double pfx, double pfy, double pfz,
                                                const D3 &xp, double nx, double ny, double nz, double c0, double c1, double c2,
                                                double *Jrow) {
    const double nfx = round_f32(nx), nfy = round_f32(ny), nfz = round_f32(nz);
    const D3 n1 = dqb_warp_normal_exact(bh, nfx, nfy, nfz);
    const D3 np_ = dqb_warp_normal_exact(lwq, round_f32(n1.x), round_f32(n1.y), round_f32(n1.z));
    const double d0 = xp.x - c0, d1 = xp.y - c1, d2 = xp.z - c2;
    const double r = (np_.x * d0 + np_.y * d1) + np_.z * d2;
    // u = A^T n', h = A^T (x' - c),  A^T y = vec(rl* Y rl)
    const Q4 rl{lwq[0], lwq[1], lwq[2], lwq[3]};
    const Q4 rlc = qconj(rl);
    const Q4 U = qmul(qmul(rlc, qpure(np_.x, np_.y, np_.z)), rl);
    const Q4 Hq = qmul(qmul(rlc, qpure(d0, d1, d2)), rl);
    const Q4 Up = qpure(U.x, U.y, U.z), Hp = qpure(Hq.x, Hq.y, Hq.z);
    const Q4 rr{bh[0], bh[1], bh[2], bh[3]}, dd{bh[4], bh[5], bh[6], bh[7]};
    const Q4 Ur = qmul(Up, rr);
    Q4 g_r = qadd(qadd(qmul(Ur, qpure(pfx, pfy, pfz)), qmul(Up, dd)), qmul(qmul(Hp, rr), qpure(nfx, nfy, nfz)));
    g_r = qscale(g_r, -2.0);
    Q4 g_d = qscale(Ur, 2.0);
    const double gb = (g_r.w * bh[0] + g_r.x * bh[1] + g_r.y * bh[2] + g_r.z * bh[3]) +
                      (g_d.w * bh[4] + g_d.x * bh[5] + g_d.y * bh[6] + g_d.z * bh[7]);
    const double inv = 1.0 / nb;
    g_r = Q4{(g_r.w - gb * bh[0]) * inv, (g_r.x - gb * bh[1]) * inv, (g_r.y - gb * bh[2]) * inv, (g_r.z - gb * bh[3]) * inv};
    g_d = Q4{(g_d.w - gb * bh[4]) * inv, (g_d.x - gb * bh[5]) * inv, (g_d.y - gb * bh[6]) * inv, (g_d.z - gb * bh[7]) * inv};
#pragma unroll
    for (int j = 0; j < kKMaxS; ++j) {
        if (j < k) {
            const double *q = node_dq + 8 * idx[j];
            const Q4 rac{q[0], -q[1], -q[2], -q[3]}, dac{q[4], -q[5], -q[6], -q[7]};
            const Q4 a = qadd(qmul(g_r, rac), qmul(g_d, dac));
            const Q4 t = qmul(g_d, rac);
            const double hw = 0.5 * w[j];
            Jrow[6 * j + 0] = hw * a.x; Jrow[6 * j + 1] = hw * a.y; Jrow[6 * j + 2] = hw * a.z;
            Jrow[6 * j + 3] = hw * t.x; Jrow[6 * j + 4] = hw * t.y; Jrow[6 * j + 5] = hw * t.z;
        }
    }
    return r;
}

// One kTile-sample tile per block.  Samples must be sorted by their k-tuple of nodes (any order
// is CORRECT; sorted order just means few runs per tile and therefore few atomics).
// PLANNED: the (tile, tuple) runs are static per frame, so each run owns a row of `partial`
// ({upper triangle of its (6K)^2 Gram matrix | J^T r | 0.5 r^2 | count}, row = run_id[first sample]) and the
// sums are STORED there; gn_gather_kernel then adds the rows into the blocks through a precomputed incidence
// list: no floating-point atomics, same bits every run, about half the memory operations.
// The regulariser's pair rows (gn_reg_pairs, below) are independent of the data rows: in the planned build they are
// computed by extra workgroups appended to the data-row launch (blockIdx.x >= n_tiles) instead of a launch of their own.
struct RegTail {
    const int *node_nbr;        // NULL: no regulariser workgroups
    const double *node_pos, *node_w;
    double *partial_reg;
    double rw;
    int N, k, n_tiles;
    // dfh_gn_iteration: doubles the launch's LAST workgroups set to zero (the solve's workspace: its clearing rides along here
    // instead of being a launch of its own between gather and solve); first_zero_wg = index of the first such workgroup
    double *zero_ptr;
    unsigned long long zero_count;
    int first_zero_wg;
};
constexpr int kZeroPerWg = 4 * kTile;            // doubles one workgroup clears (kTile threads x 4)
__device__ void gn_reg_pairs(int block, const int *__restrict__ node_nbr, int N, int k, const double *__restrict__ node_dq,
                             const double *__restrict__ node_pos, const double *__restrict__ node_w, double rw,
                             const int *__restrict__ row_ptr, const int *__restrict__ col, double *__restrict__ vals,
                             double *__restrict__ rhs, double *__restrict__ cost_count, double *__restrict__ partial_reg);

// ASSOC: the projective association (associate_kernel's arithmetic, same bits) runs inside this kernel -- every sample of the
// tile is warped once, associated against `depth`, its correspondence and validity are written to corr / valid (for the
// callers that read them) and the valid ones go straight on to their Jacobian rows: one launch and one blend + warp per
// sample less per GN iteration.
struct AssocArgs {
    AssocParams ap;
    const float *depth;
    const AssocView *views;     // non-null: n_views float32 views from a dfh_gn_pack_views table (depth / ap.lw_cam unused)
    int n_views;
    int cull;                   // 1: drop, per tile, the views none of its samples can be valid in (tile_view_mask)
};

#ifdef DFH_BUILD_TRACE   // experiment builds only: wall-clock stamps of every tile's phases
__device__ unsigned long long g_build_trace[8192][8];
#define BT_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_build_trace[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define BT_STAMP(k) do {} while (0)
#endif

template <int K, bool PLANNED, bool ASSOC>
__global__ __launch_bounds__(kTile) void gn_build_data_kernel(const double *__restrict__ spos, const double *__restrict__ snrm,
                                                             const int *__restrict__ nbr, const double *__restrict__ wts,
                                                             double *__restrict__ corr,
                                                             unsigned char *__restrict__ valid,
                                                             const double *__restrict__ node_dq, const BuildParams p,
                                                             const int *__restrict__ row_ptr, const int *__restrict__ col,
                                                             double *__restrict__ vals, double *__restrict__ rhs,
                                                             double *__restrict__ cost_count, const int *__restrict__ run_id,
                                                             double *__restrict__ partial, double *__restrict__ tile_cost,
                                                             const RegTail rt, const AssocArgs aa) {
    if (PLANNED && rt.zero_ptr && (int)blockIdx.x >= rt.first_zero_wg) {     // (workgroup-uniform) clearing that rides along
        const unsigned long long i0 = (unsigned long long)((int)blockIdx.x - rt.first_zero_wg) * kZeroPerWg + 4ull * threadIdx.x;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < rt.zero_count) rt.zero_ptr[i0 + j] = 0.0;
        return;
    }
    if (PLANNED && (int)blockIdx.x >= rt.n_tiles) {      // (workgroup-uniform) the regulariser's share of this launch
        gn_reg_pairs((int)blockIdx.x - rt.n_tiles, rt.node_nbr, rt.N, rt.k, node_dq, rt.node_pos, rt.node_w, rt.rw, row_ptr, col, vals,
                     rhs, cost_count, rt.partial_reg);
        return;
    }
    BT_STAMP(0);
    constexpr int NJ = 6 * K;                   // Jacobian entries per sample
    constexpr int LD = NJ + 1;                  // + residual
    __shared__ double sJ[kTile * LD];
    // (the planned build tells runs apart by their scratch-row ids and needs no node tuples in LDS; without them and with 16-bit
    // row ids the workgroup's LDS drops from 58.4 to 52.8 KB: three workgroups per CU instead of two)
    __shared__ int sIdx[PLANNED ? 1 : kTile * K];
    const int tid = threadIdx.x;
    const int s = blockIdx.x * kTile + tid;
    const int tile_n = min(kTile, p.S - blockIdx.x * kTile);
    // ---- each valid sample computes its residual and Jacobian row and appends it to the tile's
    // compacted list in LDS (order preserved): the reduction below only walks valid rows
    int idx[kKMaxS];
    double w[kKMaxS];
    double a_bh[8], a_nb = 1.0, a_pf[3] = {0, 0, 0}, a_c[3] = {0, 0, 0};
    D3 a_xp{0, 0, 0};
    bool a_ok = false;
    // the sample's normal is only needed for its Jacobian row, but asking for it here takes a memory round trip out of that phase
    double a_n[3] = {0.0, 0.0, 0.0};
    if (ASSOC && tid < tile_n) {
        a_n[0] = snrm[3 * (size_t)s]; a_n[1] = snrm[3 * (size_t)s + 1]; a_n[2] = snrm[3 * (size_t)s + 2];
    }
    if (ASSOC && tid < tile_n) {
#pragma unroll
        for (int j = 0; j < kKMaxS; ++j) {
            idx[j] = j < K ? nbr[(size_t)s * K + j] : 0;
            w[j] = j < K ? wts[(size_t)s * K + j] : 0.0;
        }
        a_nb = blend_static(node_dq, idx, w, K, a_bh);
        a_pf[0] = round_f32(spos[3 * (size_t)s]); a_pf[1] = round_f32(spos[3 * (size_t)s + 1]); a_pf[2] = round_f32(spos[3 * (size_t)s + 2]);
        const D3 x1 = dqb_warp_exact(a_bh, a_pf[0], a_pf[1], a_pf[2]);
        a_xp = dqb_warp_exact(p.lw.q, round_f32(x1.x), round_f32(x1.y), round_f32(x1.z));
    }
    unsigned view_mask = ~0u;
    if (ASSOC && aa.views && aa.cull) {                                        // (workgroup-uniform)
        // the tile's warped samples' box -> the views any of them can be valid in
        __shared__ double sBox[kTileWaves][6];
        __shared__ unsigned sMask;
        const bool in = tid < tile_n;
        double bx[6] = {in ? a_xp.x : __builtin_huge_val(), in ? -a_xp.x : __builtin_huge_val(), in ? a_xp.y : __builtin_huge_val(),
                        in ? -a_xp.y : __builtin_huge_val(), in ? a_xp.z : __builtin_huge_val(), in ? -a_xp.z : __builtin_huge_val()};
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) bx[c6] = fmin(bx[c6], __shfl_xor(bx[c6], o, 64));
        if ((tid & 63) == 0)
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) sBox[tid >> 6][c6] = bx[c6];
        __syncthreads();
        double box[6];
#pragma unroll
        for (int c6 = 0; c6 < 6; ++c6) {
            double m = sBox[0][c6];
#pragma unroll
            for (int w_ = 1; w_ < kTileWaves; ++w_) m = fmin(m, sBox[w_][c6]);
            box[c6] = (c6 & 1) ? -m : m;                                       // (maxima were carried negated)
        }
        view_mask = tile_view_mask(aa.ap, aa.views, aa.n_views, box, &sMask);
    }
    if (ASSOC && tid < tile_n)
        a_ok = aa.views ? associate_views<float>(aa.ap, aa.views, aa.n_views, a_xp, a_c, view_mask) : associate_point<float>(aa.ap, aa.depth, a_xp, a_c);
    // corr / valid are outputs only: stored after the last global load of the kernel (stored here, every later s_waitcnt for a
    // load also waited for these stores' acknowledgements)
    auto store_assoc = [&]() {
        if (ASSOC && tid < tile_n) {
            corr[3 * (size_t)s] = a_c[0]; corr[3 * (size_t)s + 1] = a_c[1]; corr[3 * (size_t)s + 2] = a_c[2];
            valid[s] = a_ok ? 1 : 0;
        }
    };
    const bool act = ASSOC ? a_ok : (tid < tile_n && valid[s] != 0);
    BT_STAMP(1);
    __shared__ int sWaveCnt[kTileWaves];
    const unsigned long long bal = __ballot(act);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) sWaveCnt[wv] = __popcll(bal);
    __syncthreads();
    int pos = __popcll(bal & ((1ull << lane) - 1ull));
    for (int w_ = 0; w_ < wv; ++w_) pos += sWaveCnt[w_];
    int n_valid = 0;
#pragma unroll
    for (int w_ = 0; w_ < kTileWaves; ++w_) n_valid += sWaveCnt[w_];
    constexpr int ST_ = gn_row_stride(K);
    double *live = PLANNED ? tile_cost + 2 * (size_t)rt.n_tiles : nullptr;      // one flag per row, dense: 0 = row not written this iteration
    const int row_first = PLANNED ? run_id[blockIdx.x * kTile] : 0;
    const int rows_tile = PLANNED ? run_id[blockIdx.x * kTile + tile_n - 1] - row_first + 1 : 0;
    if (n_valid == 0) {                                              // tiles without a valid sample contribute nothing:
        store_assoc();
        if (PLANNED && tid < rows_tile) live[row_first + tid] = 0.0;                                 // their rows are dead
        if (PLANNED && tid == 0) { tile_cost[2 * blockIdx.x] = 0.0; tile_cost[2 * blockIdx.x + 1] = 0.0; }
        return;
    }
    if (!PLANNED && tid == 0) atomicAdd(cost_count + 1, (double)n_valid);        // valid-sample count
    __shared__ unsigned short sRow[PLANNED ? kTile : 1];             // partial row of every compacted sample, relative to the tile's first
    __shared__ double sObj[kTileWaves];
    double obj = 0.0;
    if (act) {
        if (PLANNED) sRow[pos] = (unsigned short)(run_id[s] - row_first);
        double Jrow[NJ];
        double r;
        if (ASSOC) {
            r = data_row_from(node_dq, idx, w, K, p.lw.q, a_bh, a_nb, a_pf[0], a_pf[1], a_pf[2], a_xp, a_n[0], a_n[1], a_n[2], a_c[0], a_c[1],
                              a_c[2], Jrow);
        } else {
#pragma unroll
            for (int j = 0; j < kKMaxS; ++j) {
                idx[j] = j < K ? nbr[(size_t)s * K + j] : 0;
                w[j] = j < K ? wts[(size_t)s * K + j] : 0.0;
            }
            r = data_row(node_dq, idx, w, K, p.lw.q, spos[3 * (size_t)s], spos[3 * (size_t)s + 1], spos[3 * (size_t)s + 2],
                         snrm[3 * (size_t)s], snrm[3 * (size_t)s + 1], snrm[3 * (size_t)s + 2], corr[3 * (size_t)s],
                         corr[3 * (size_t)s + 1], corr[3 * (size_t)s + 2], Jrow);
        }
        obj = 0.5 * r * r;                                     // this sample's term of the objective
        if (p.huber > 0.0 && fabs(r) > p.huber) {              // Huber: rho = delta (|r| - delta / 2) beyond delta, and the
            obj = p.huber * (fabs(r) - 0.5 * p.huber);         // row and its residual get sqrt of the IRLS weight
            const double sc = sqrt(p.huber / fabs(r));
            r *= sc;
#pragma unroll
            for (int j = 0; j < NJ; ++j) Jrow[j] *= sc;
        }
        if (!PLANNED) {
#pragma unroll
            for (int j = 0; j < K; ++j) sIdx[pos * K + j] = idx[j];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) sJ[pos * LD + j] = Jrow[j];
        sJ[pos * LD + NJ] = r;
    }
    store_assoc();
    BT_STAMP(2);
    if (PLANNED) {                                                   // the tile's objective, added in a fixed order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) obj += __shfl_xor(obj, o, 64);
        if (lane == 0) sObj[wv] = obj;
    }
    __syncthreads();
    if (PLANNED && tid == 0) {
        double o = sObj[0];                                          // (fixed order)
#pragma unroll
        for (int w_ = 1; w_ < kTileWaves; ++w_) o += sObj[w_];
        tile_cost[2 * blockIdx.x] = o;
        tile_cost[2 * blockIdx.x + 1] = (double)n_valid;
    }
    // run boundaries: sRun[0..n_runs] are the offsets in the compacted list where the node tuple changes
    __shared__ int sRun[kTile + 1];
    __shared__ int sNRuns;
    {
        bool head = false;
        if (tid < n_valid) {
            head = tid == 0;
            if (!head) {
                if (PLANNED) {
                    head = sRow[tid] != sRow[tid - 1];               // (a scratch row = a run of equal tuples inside the tile)
                } else {
#pragma unroll
                    for (int j = 0; j < K; ++j) head = head || (sIdx[tid * K + j] != sIdx[(tid - 1) * K + j]);
                }
            }
        }
        // positions of the heads by ballot + prefix (same scheme as the compaction above)
        const unsigned long long hb = __ballot(head);
        if (lane == 0) sWaveCnt[wv] = __popcll(hb);       // (every thread passed the barrier after the first use)
        __syncthreads();
        int hp = __popcll(hb & ((1ull << lane) - 1ull));
        for (int w_ = 0; w_ < wv; ++w_) hp += sWaveCnt[w_];
        if (head) sRun[hp] = tid;
        if (tid == 0) {
            int n = 0;
            for (int w_ = 0; w_ < kTileWaves; ++w_) n += sWaveCnt[w_];
            sRun[n] = n_valid;
            sNRuns = n;
        }
        __syncthreads();
    }
    const int n_runs = sNRuns;
    BT_STAMP(3);
#ifdef DFH_BUILD_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_build_trace[blockIdx.x][6] = (unsigned long long)n_valid | ((unsigned long long)n_runs << 16);
#endif
    // block index of every (slot a, slot b) node pair of every run, searched once, in parallel
    constexpr bool kBlkTable = K <= 4 && !PLANNED;         // 256 runs x K^2 ints must fit next to sJ
    __shared__ int sBlk[kBlkTable ? kTile * K * K : 1];
    if (kBlkTable) {
        for (int q = tid; q < n_runs * K * K; q += kTile) {
            const int rn = q / (K * K), pr = q - rn * (K * K);
            const int t0 = sRun[rn];
            sBlk[q] = find_block(row_ptr, col, sIdx[t0 * K + pr / K], sIdx[t0 * K + pr % K]);
        }
        __syncthreads();
    }
    // entries: PLANNED: the 6x6 sub-blocks (sa <= sb) of the Gram matrix in scratch-row order; else its upper triangle; then NJ
    // entries of J^T r, then the cost
    constexpr int NUP = PLANNED ? gn_row_gram(K) : NJ * (NJ + 1) / 2;
    if (PLANNED) {
        if (tid < n_runs) partial[(size_t)(row_first + sRow[sRun[tid]]) * ST_ + NUP + NJ + 1] = (double)(sRun[tid + 1] - sRun[tid]);
        // live flags of this tile's rows: 1 where a run has valid samples this iteration, 0 elsewhere (dead rows are
        // neither cleared here nor read by the gather)
        __shared__ int sTouched[kTile];
        if (tid < rows_tile) sTouched[tid] = 0;
        __syncthreads();
        if (tid < n_runs) sTouched[sRow[sRun[tid]]] = 1;
        __syncthreads();
        if (tid < rows_tile) live[row_first + tid] = sTouched[tid] ? 1.0 : 0.0;
    }
    BT_STAMP(4);
    if constexpr (PLANNED) {
        // Gram matrix of every run on the matrix cores: G = X^T X with X = the run's rows of [J | r] (n x (6K + 1)), as 16 x 16
        // tiles of v_mfma_f64_16x16x4_f64 (four samples per step, A and B fragments straight from the compacted rows in
        // LDS: two reads per lane and step where the scalar loop read two values per sample and ENTRY).  One wave per
        // run, runs dealt round-robin; the accumulation order (sample order, fused multiply-add) is fixed, so the bits are
        // the same every launch.  Each lane then stores its accumulator elements straight to their places in the scratch
        // row (a 16-lane group writes three 48-byte runs; the row is written whole by this wave).
        constexpr int NC = NJ + 1;                                   // columns: Jacobian + residual
        constexpr int NT = (NC + 15) / 16;                           // 16-column tiles per side
        typedef double d4 __attribute__((ext_vector_type(4)));
        const int li = lane & 15, lk = lane >> 4;
        for (int rn = wv; rn < n_runs; rn += kTile / 64) {
            const int t0 = sRun[rn], t1 = sRun[rn + 1];
            d4 acc[NT * (NT + 1) / 2];
#pragma unroll
            for (int q = 0; q < NT * (NT + 1) / 2; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
            for (int t = t0; t < t1; t += 4) {
                double x[NT];                                        // A[i = li][k = lk] = B[k = lk][j = li] = X[t + lk][16 b + li]
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int c = 16 * b + li;
                    const double *src = sJ + (t + lk) * LD + c;
                    x[b] = (t + lk < t1 && c < NC) ? *src : 0.0;
                }
                int q = 0;
#pragma unroll
                for (int bi = 0; bi < NT; ++bi)
#pragma unroll
                    for (int bj = bi; bj < NT; ++bj, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[bi], x[bj], acc[q], 0, 0, 0);
            }
            double *dst = partial + (size_t)(row_first + sRow[t0]) * ST_;
            int q = 0;
#pragma unroll
            for (int bi = 0; bi < NT; ++bi)
#pragma unroll
                for (int bj = bi; bj < NT; ++bj, ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                    // C/D: column = lane & 15, row = (lane >> 4) + 4 r
                        const int pa = 16 * bi + lk + 4 * r, pb = 16 * bj + li;
                        const double v = acc[q][r];
                        if (pa < NJ && pb < NJ) {
                            const int sa = pa / 6, sb = pb / 6;
                            if (sa <= sb) {
                                dst[36 * gn_sub(K, sa, sb) + 6 * (pa - 6 * sa) + (pb - 6 * sb)] = v;
                                // a diagonal sub-block that straddles two tiles: its lower entries lie in the tile that is
                                // not computed; the product is symmetric bit for bit
                                if (bi != bj && sa == sb) dst[36 * gn_sub(K, sa, sa) + 6 * (pb - 6 * sb) + (pa - 6 * sa)] = v;
                            }
                        } else if (pa < NJ && pb == NJ) {
                            dst[NUP + pa] = v;                       // J^T r
                        } else if (pa == NJ && pb == NJ) {
                            dst[NUP + NJ] = 0.5 * v;                 // cost
                        }
                    }
        }
        BT_STAMP(5);
        return;
    }
    for (int e = tid; e < NUP + NJ + 1; e += kTile) {
        int pa, pb;                              // Jacobian columns of this entry (pb == NJ: residual)
        if (e < NUP && PLANNED) {
            int sub = e / 36, sa = 0;
            const int r36 = e - 36 * sub;
            while (sub >= K - sa) { sub -= K - sa; ++sa; }          // sub-block (sa, sa + sub)
            pa = 6 * sa + r36 / 6; pb = 6 * (sa + sub) + r36 % 6;   // (a diagonal sub-block's lower entries: same products, same order)
        } else if (e < NUP) {
            int row = 0, rem = e;
            while (rem >= NJ - row) { rem -= NJ - row; ++row; }
            pa = row; pb = row + rem;
        } else if (e < NUP + NJ) {
            pa = e - NUP; pb = NJ;
        } else {
            pa = NJ; pb = NJ;
        }
        for (int rn = 0; rn < n_runs; ++rn) {
            const int t0 = sRun[rn], t1 = sRun[rn + 1];
            // four independent chains: the loop is bound by LDS latency, not bandwidth (fixed association order)
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
            int t = t0;
            for (; t + 3 < t1; t += 4) {
                const double a0 = sJ[t * LD + pa], b0 = sJ[t * LD + pb];
                const double a1 = sJ[(t + 1) * LD + pa], b1 = sJ[(t + 1) * LD + pb];
                const double a2 = sJ[(t + 2) * LD + pa], b2 = sJ[(t + 2) * LD + pb];
                const double a3 = sJ[(t + 3) * LD + pa], b3 = sJ[(t + 3) * LD + pb];
                acc0 += a0 * b0; acc1 += a1 * b1; acc2 += a2 * b2; acc3 += a3 * b3;
            }
            for (; t < t1; ++t) acc0 += sJ[t * LD + pa] * sJ[t * LD + pb];
            const double acc = (acc0 + acc1) + (acc2 + acc3);
            if (PLANNED) {
                partial[(size_t)(row_first + sRow[t0]) * ST_ + e] = pa == NJ ? 0.5 * acc : acc;
                continue;
            }
            if (acc == 0.0) continue;
            if (pb == NJ && pa == NJ) {
                atomicAdd(cost_count, 0.5 * acc);
            } else if (pb == NJ) {
                atomicAdd(rhs + 6 * sIdx[t0 * K + pa / 6] + pa % 6, acc);
            } else {
                const int sa = pa / 6, sb = pb / 6;
                const int na = sIdx[t0 * K + sa], nbn = sIdx[t0 * K + sb];
                const int ia = pa % 6, ib = pb % 6;
                const int blk = kBlkTable ? sBlk[rn * K * K + sa * K + sb] : find_block(row_ptr, col, na, nbn);
                if (blk >= 0) atomicAdd(vals + 36 * (size_t)blk + 6 * ia + ib, acc);
                if (!(na == nbn && ia == ib)) {
                    const int blk2 = na == nbn ? blk : (kBlkTable ? sBlk[rn * K * K + sb * K + sa] : find_block(row_ptr, col, nbn, na));
                    if (blk2 >= 0) atomicAdd(vals + 36 * (size_t)blk2 + 6 * ib + ia, acc);
                }
            }
        }
    }
    BT_STAMP(5);
}

// Second half of the planned build.  Workgroups [0, ceil(n_blocks/4)): one WAVE per 6x6 block (a,b), lanes 0..35 own
// entry (ia, ib) of the block and add the block's list (blk_ent = row * K^2 + sa * K + sb) in list order.  Next
// workgroups: J^T r, one wave per node, lane (j, i) takes every tenth entry of the node's list (node_ent = row * K +
// slot) for unknown i.  Last workgroup: cost and valid count, a fixed-order tree over the {cost, count} pairs.
// The regulariser's pair rows (K = 2 layout, own lists) are a second set of lists walked by the same wave right after
// the data rows': value = data sum + regulariser sum, the two roundings of "store, then add in a second launch".

// one wave: sum over block b's list of entry (ia, ib) of the rows' Gram matrices (lanes 0..35; others return 0)
template <int K>
__device__ __forceinline__ double gather_block_list(const double *__restrict__ partial, const double *__restrict__ live,
                                                    const int *__restrict__ blk_ptr, const int *__restrict__ blk_ent, int b, int lane,
                                                    int wv) {
    constexpr int NJ = 6 * K, NE = gn_row_stride(K), kLive = gn_row_entries(K);
    const int beg = blk_ptr[b], end = blk_ptr[b + 1];
    const int ia = (lane % 36) / 6, ib = lane % 6;
    auto value = [&](int ent) {
        const int row = ent / (K * K), pr = ent - row * (K * K);
        return partial[(size_t)row * NE + gn_gram_index(K, pr / K, pr % K, ia, ib)];
    };
    // The walk is a chain of dependent loads (entry -> live flag -> values), so it is organised by hops, not by
    // entries: up to 256 entries and then their flags are fetched together (two hops), the live ones are compacted
    // in list order into LDS, and lanes 0..35 (one per block entry) add them with 16 value loads in flight.
    __shared__ int sLiveB[4][256];
    double acc = 0.0;
    for (int base = beg; base < end; base += 256) {
        int ent[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ent[u] = base + 64 * u + lane < end ? blk_ent[base + 64 * u + lane] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double flag = ent[u] >= 0 ? (live ? live[ent[u] / (K * K)] : partial[(size_t)(ent[u] / (K * K)) * NE + kLive]) : 0.0;
            on[u] = flag != 0.0;
        }
        int nl = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long m = __ballot(on[u]);
            if (on[u]) sLiveB[wv][nl + __popcll(m & ((1ull << lane) - 1ull))] = ent[u];
            nl += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 36) {
            int q = 0;
            for (; q + 15 < nl; q += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = value(sLiveB[wv][q + u]);
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u];
            }
            for (; q + 3 < nl; q += 4) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = value(sLiveB[wv][q + u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += v[u];
            }
            for (; q < nl; ++q) acc += value(sLiveB[wv][q]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    return acc;
}

#ifdef DFH_GATHER_TRACE  // experiment builds only: wall-clock stamps of every block wave's hops
__device__ unsigned long long g_gather_trace[8192][8];
#define GT_STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); if (lane == 0 && b < 8192) g_gather_trace[b][k] = wall_clock64(); } while (0)
#else
#define GT_STAMP(k) do {} while (0)
#endif

// The data rows' list and the regulariser's list of one block walked TOGETHER: the walk is a chain of dependent hops (list
// bounds -> entries -> live flags -> values) and doing the two lists one after the other doubles the chain.  Here every
// hop is issued for both lists at once (4 hops).  Values are added in rounds of 16 loads in flight per lane (absent
// entries add 0.0).  A list with more than kCoopList live entries (the diagonal blocks: every row that touches the node,
// ~170) is split into four contiguous quarters, one per wave of the workgroup, and the quarters' sums are added in order
// -- walked by one wave alone it was eleven dependent rounds of ~2 us and set the whole launch's duration.
// Regulariser lists longer than 64 entries or data lists longer than 256 take the sequential walk.
#ifndef DFH_GATHER_DEPTH
#define DFH_GATHER_DEPTH 4
#endif
constexpr int kGatherDepth = DFH_GATHER_DEPTH;   // value loads in flight per lane and round
constexpr int kCoopList = 3 * kGatherDepth;

struct GatherLds {
    int liveD[4][256];
    int liveR[4][64];
    int coop[4];
    double part[4][4][36];
    double comb[4][3][2][36];
};

// lanes 0..35: sum of entry (ia, ib) = lane / 6, lane % 6 over list[q0, q1).  A list entry's sub-block is 36 contiguous
// doubles, so 18 lanes take it with one 16-byte load each and a load instruction covers THREE entries (lane group g takes
// entries q0 + g, q0 + g + 3, ...): kGatherDepth instructions in flight = 48 entries a round (a round costs ~3 us of
// latency whatever it carries).  Entries whose sub-block is stored transposed (tuple slots sa > sb) are added up in
// stored orientation on their own and transposed once at the end; the three groups' sums are added in group order.
// `comb` = this wave's scratch (3 x 2 x 36 doubles).  Order of the additions: fixed, not list order.
template <int K>
__device__ __forceinline__ double gather_rounds(const double *__restrict__ partial, const int *list, int q0, int q1, int lane,
                                                double (*comb)[2][36]) {
    constexpr int NE = gn_row_stride(K);
    const int g = lane / 18, h = lane - 18 * g;                     // lanes 54..63: g == 3, idle
    double2 sd{0.0, 0.0}, st{0.0, 0.0};
    for (int q = q0; q < q1; q += 3 * kGatherDepth) {
        int e[kGatherDepth];
#pragma unroll
        for (int u = 0; u < kGatherDepth; ++u) e[u] = list[min(q + 3 * u + g, q1 - 1)];
        double2 v[kGatherDepth];
        bool tr[kGatherDepth];
#pragma unroll
        for (int u = 0; u < kGatherDepth; ++u) {
            const int row = e[u] / (K * K), pr = e[u] - row * (K * K);
            const int sa = pr / K, sb = pr - sa * K;
            tr[u] = sa > sb;
            const double2 *src = reinterpret_cast<const double2 *>(partial + (size_t)row * NE + 36 * (tr[u] ? gn_sub(K, sb, sa) : gn_sub(K, sa, sb))) + h;
            v[u] = (g < 3 && q + 3 * u + g < q1) ? *src : double2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < kGatherDepth; ++u) {
            sd.x += tr[u] ? 0.0 : v[u].x; sd.y += tr[u] ? 0.0 : v[u].y;
            st.x += tr[u] ? v[u].x : 0.0; st.y += tr[u] ? v[u].y : 0.0;
        }
    }
    if (g < 3) {
        comb[g][0][2 * h] = sd.x; comb[g][0][2 * h + 1] = sd.y;
        comb[g][1][2 * h] = st.x; comb[g][1][2 * h + 1] = st.y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    double tot = 0.0;
    if (lane < 36) {
        const int m = lane, mt = 6 * (lane % 6) + lane / 6;
        tot = ((comb[0][0][m] + comb[1][0][m]) + comb[2][0][m]) + ((comb[0][1][mt] + comb[1][1][mt]) + comb[2][1][mt]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();                               // (the scratch is reused by the wave's next call)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    return tot;
}

// one workgroup = four blocks, one wave each; every wave of the workgroup must call this (it synchronises)
template <int K>
__device__ __forceinline__ double gather_block_both(const double *__restrict__ partial, const double *__restrict__ live,
                                                    const int *__restrict__ blk_ptr,
                                                    const int *__restrict__ blk_ent, const double *__restrict__ rpartial,
                                                    const int *__restrict__ rblk_ptr, const int *__restrict__ rblk_ent, int b, bool real,
                                                    int lane, int wv, GatherLds &L) {
    constexpr int NE = gn_row_stride(K), kLive = gn_row_entries(K);
    constexpr int NE2 = gn_row_stride(2), kLive2 = gn_row_entries(2);
    const int ia = (lane % 36) / 6, ib = lane % 6;
    double acc = 0.0, racc = 0.0;
    int nl = 0;
    bool coop = false;
    if (real) {
        // hop 1
        GT_STAMP(0);
        const int beg = blk_ptr[b], end = blk_ptr[b + 1];
        const int rbeg = rpartial ? rblk_ptr[b] : 0, rend = rpartial ? rblk_ptr[b + 1] : 0;
        GT_STAMP(1);
        if (end - beg > 256 || rend - rbeg > 64) {
            acc = gather_block_list<K>(partial, live, blk_ptr, blk_ent, b, lane, wv);
            if (rpartial) racc = gather_block_list<2>(rpartial, nullptr, rblk_ptr, rblk_ent, b, lane, wv);
        } else {
            // hop 2: entries
            int ent[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ent[u] = beg + 64 * u + lane < end ? blk_ent[beg + 64 * u + lane] : -1;
            const int rent = rbeg + lane < rend ? rblk_ent[rbeg + lane] : -1;
            GT_STAMP(2);
            // hop 3: live flags
            // (flags first, tests after: `ent >= 0 && flag != 0` in one expression makes every load wait for the one before)
            double flag[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) flag[u] = ent[u] >= 0 ? (live ? live[ent[u] / (K * K)] : partial[(size_t)(ent[u] / (K * K)) * NE + kLive]) : 0.0;
            const double rflag = rent >= 0 ? rpartial[(size_t)(rent / 4) * NE2 + kLive2] : 0.0;
            bool on[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) on[u] = flag[u] != 0.0;
            const bool ron = rflag != 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long m = __ballot(on[u]);
                if (on[u]) L.liveD[wv][nl + __popcll(m & ((1ull << lane) - 1ull))] = ent[u];
                nl += __popcll(m);
            }
            const unsigned long long rm = __ballot(ron);
            if (ron) L.liveR[wv][__popcll(rm & ((1ull << lane) - 1ull))] = rent;
            const int rnl = __popcll(rm);
            __builtin_amdgcn_wave_barrier();
            GT_STAMP(3);
            coop = nl > kCoopList;
            // hop 4: values of both lists in flight together (regulariser: usually 1-2 entries)
            if (rnl > 0) racc = gather_rounds<2>(rpartial, L.liveR[wv], 0, rnl, lane, L.comb[wv]);
            if (!coop && nl > 0) acc = gather_rounds<K>(partial, L.liveD[wv], 0, nl, lane, L.comb[wv]);
#ifdef DFH_GATHER_TRACE
            if (lane == 0 && b < 8192)
                g_gather_trace[b][5] = (unsigned long long)(end - beg) | ((unsigned long long)nl << 16) | ((unsigned long long)(rend - rbeg) << 32) | ((unsigned long long)rnl << 48);
#endif
        }
    }
    if (lane == 0) L.coop[wv] = coop ? nl : 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int n = L.coop[w];
        if (n > 0) {                                              // (workgroup-uniform)
            const int seg = (n + 3) / 4, q0 = wv * seg, q1 = min(n, q0 + seg);
            const double sum = q0 < q1 ? gather_rounds<K>(partial, L.liveD[w], q0, q1, lane, L.comb[wv]) : 0.0;
            if (lane < 36) L.part[w][wv][lane] = sum;
        }
    }
    __syncthreads();
    if (coop && lane < 36) acc = ((L.part[wv][0][lane] + L.part[wv][1][lane]) + L.part[wv][2][lane]) + L.part[wv][3][lane];
    GT_STAMP(4);
    return acc + racc;
}

// one wave: J^T r of node a from its list; the total for unknown i ends up in lanes 0..5
template <int K>
__device__ __forceinline__ double gather_node_list(const double *__restrict__ partial, const double *__restrict__ live,
                                                   const int *__restrict__ node_ptr,
                                                   const int *__restrict__ node_ent, int a, int lane, int wv) {
    constexpr int NUP = gn_row_gram(K), NE = gn_row_stride(K), kLive = gn_row_entries(K);
    double acc = 0.0;
    const int j = lane / 6, i = lane - 6 * j;                  // lanes 60..63 idle
    __shared__ int sLive[4][64];
    const int beg = node_ptr[a], end = node_ptr[a + 1];
    for (int base = beg; base < end; base += 64) {
        // 64 entries and their rows' live flags at once; the live ones, compacted in list order, are then taken
        // ten at a time (lane group j takes the j-th of each ten) -- one value hop per ten entries
        const int n = min(64, end - base);
        int mine = lane < n ? node_ent[base + lane] : -1;
        const double flag = mine >= 0 ? (live ? live[mine / K] : partial[(size_t)(mine / K) * NE + kLive]) : 0.0;
        if (flag == 0.0) mine = -1;
        const unsigned long long live = __ballot(mine >= 0);
        if (mine >= 0) sLive[wv][__popcll(live & ((1ull << lane) - 1ull))] = mine;
        __builtin_amdgcn_wave_barrier();
        const int nl = __popcll(live);
        if (j < 10) {
            for (int m0 = j; m0 < nl; m0 += 80) {               // eight loads in flight per lane (absent entries add 0.0)
                int e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = sLive[wv][min(m0 + 10 * u, nl - 1)];
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = e[u] / K, slot = e[u] - row * K;
                    const double *src = partial + (size_t)row * NE + NUP + slot * 6 + i;
                    v[u] = m0 + 10 * u < nl ? *src : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    double tot = acc;
#pragma unroll
    for (int k = 1; k < 10; ++k) {
        const double o = __shfl(acc, lane + 6 * k, 64);
        tot += (lane + 6 * k < 60) ? o : 0.0;
    }
    return tot;
}

// one workgroup: fixed-order sum of n_cc {cost, count} pairs cc_stride doubles apart; valid in thread 0
__device__ __forceinline__ void gather_cost(const double *__restrict__ cc, int n_cc, int cc_stride, double (*red)[64], double &c0,
                                            double &c1) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double c = 0.0, n = 0.0;
    for (int r = (int)threadIdx.x; r < n_cc; r += 256) {
        c += cc[(size_t)r * cc_stride];
        n += cc[(size_t)r * cc_stride + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); n += __shfl_xor(n, o, 64); }
    __syncthreads();                                   // `red` may still be read from a previous call
    if (lane == 0) { red[0][wv] = c; red[1][wv] = n; }
    __syncthreads();
    c0 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    c1 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
}

// the regulariser's lists for the same launch (partial == NULL: none)
struct RegLists {
    const double *partial;
    const int *blk_ptr, *blk_ent, *node_ptr, *node_ent;
    int n_rows;
};

template <int K>
__global__ __launch_bounds__(256) void gn_gather_kernel(const double *__restrict__ partial, const double *__restrict__ live, int n_rows,
                                                        const int *__restrict__ blk_ptr,
                                                        const int *__restrict__ blk_ent, int n_blocks,
                                                        const int *__restrict__ node_ptr, const int *__restrict__ node_ent,
                                                        int n_nodes, double *__restrict__ vals, double *__restrict__ rhs,
                                                        double *__restrict__ cost_count, const double *__restrict__ cc, int n_cc,
                                                        int cc_stride, bool accumulate, int only_part, const RegLists rl,
                                                        const int2 *__restrict__ upper = nullptr, int n_upper = 0) {
    // upper (optional): J^T J is symmetric and so is the way its blocks are summed -- block (b, a)'s list holds the rows of block
    // (a, b)'s with the slots swapped, walked in the same order -- so only the n_upper blocks with column >= row are walked
    // (upper[u] = {block, its mirror block or -1 on the diagonal}) and each wave stores its sums twice, the second time
    // transposed: half the block waves, half the gather's reads.
    // cc: n_cc {cost, count} pairs, cc_stride doubles apart (per tile for the data term, per pair row for the regulariser)
    // only_part (debug timing): 0 = all, 1 = blocks, 2 = J^T r, 3 = cost
    // accumulate: add to what is there (a gather of its own for the regulariser rows) instead of storing
    const int nrw = (n_nodes + 3) / 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ double red[4][64];
    const int n_walk = upper ? n_upper : n_blocks;
    const int nbw = (n_walk + 3) / 4;
    if (only_part) {
        const int part_of = (int)blockIdx.x < nbw ? 1 : ((int)blockIdx.x < nbw + nrw ? 2 : 3);
        if (part_of != only_part) return;
    }
    if ((int)blockIdx.x < nbw) {
        __shared__ GatherLds L;
        // a workgroup's four blocks are a quarter of the block range apart: neighbouring blocks are the same node's row
        // and have long lists together, which one workgroup would walk alone
        const int u = wv * nbw + (int)blockIdx.x;
        const bool real = u < n_walk;
        int b = u, mir = -1;
        if (upper && real) { const int2 um = upper[u]; b = um.x; mir = um.y; }
        const double acc = gather_block_both<K>(partial, live, blk_ptr, blk_ent, rl.partial, rl.blk_ptr, rl.blk_ent, b, real, lane, wv, L);
        if (real && lane < 36) {
            double *dst = vals + 36 * (size_t)b + lane;
            *dst = accumulate ? *dst + acc : acc;
            if (mir >= 0) {
                double *dm = vals + 36 * (size_t)mir + 6 * (lane % 6) + lane / 6;
                *dm = accumulate ? *dm + acc : acc;
            }
        }
    } else if ((int)blockIdx.x < nbw + nrw) {
        const int a = ((int)blockIdx.x - nbw) * 4 + wv;
        if (a >= n_nodes) return;
        double tot = gather_node_list<K>(partial, live, node_ptr, node_ent, a, lane, wv);
        if (rl.partial) tot = tot + gather_node_list<2>(rl.partial, nullptr, rl.node_ptr, rl.node_ent, a, lane, wv);
        if (lane < 6) rhs[6 * a + lane] = accumulate ? rhs[6 * a + lane] + tot : tot;
    } else {
        double c0, c1;
        gather_cost(cc, n_cc, cc_stride, red, c0, c1);
        if (rl.partial) {
            double r0, r1;
            gather_cost(rl.partial + gn_row_gram(2) + 12, rl.n_rows, gn_row_stride(2), red, r0, r1);
            c0 = c0 + r0; c1 = c1 + r1;
        }
        if (threadIdx.x == 0) {
            cost_count[0] = accumulate ? cost_count[0] + c0 : c0;
            cost_count[1] = accumulate ? cost_count[1] + c1 : c1;
        }
    }
}

// Regularisation rows rho_ij = c_ij (W(q_i,v_j) - W(q_j,v_j)): one WAVE per (i, slot); lanes 0..35
// own one entry (a,b) of the four 6x6 blocks (ii, jj, ij, ji), lanes 0..5 also the gradient, so the
// ~150 fp64 atomics of a pair are issued side by side instead of one after the other.
__device__ void gn_reg_pairs(int block, const int *__restrict__ node_nbr, int N, int k, const double *__restrict__ node_dq,
                             const double *__restrict__ node_pos, const double *__restrict__ node_w, double rw,
                             const int *__restrict__ row_ptr, const int *__restrict__ col, double *__restrict__ vals,
                             double *__restrict__ rhs, double *__restrict__ cost_count, double *__restrict__ partial_reg) {
    // partial_reg != NULL (planned build): the pair's {upper triangle of the 12x12 Gram matrix of [J_i | J_j] |
    // J^T rho | 0.5 rho^2 | 0} is STORED in row t (92 doubles) and gathered like a 2-node data row: no atomics.
    // rows of the K = 2 layout: sub-blocks (i,i) (i,j) (j,j) | J^T rho (12) | cost | count | live flag
    constexpr int NE2 = gn_row_stride(2), kLive2 = gn_row_entries(2), kJtr2 = gn_row_gram(2), kCost2 = gn_row_gram(2) + 12;
    const int t = block * (int)(blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per node pair
    const int lane = threadIdx.x & 63;
    if (t >= N * k) return;
    const int i = t / k;
    const int j = node_nbr[t];
    if (i == j) {                                                  // zero rows, zero Jacobian
        if (partial_reg && lane == 0) {                           // dead row (its cost / count are read unconditionally)
            partial_reg[(size_t)t * NE2 + kLive2] = 0.0;
            partial_reg[(size_t)t * NE2 + kCost2] = 0.0;
            partial_reg[(size_t)t * NE2 + kCost2 + 1] = 0.0;
        }
        return;
    }
    const double vx = round_f32(node_pos[3 * j]), vy = round_f32(node_pos[3 * j + 1]), vz = round_f32(node_pos[3 * j + 2]);
    const double *qi = node_dq + 8 * i, *qj = node_dq + 8 * j;
    const D3 yi = dqb_warp_exact(qi, vx, vy, vz);
    const D3 yj = dqb_warp_exact(qj, vx, vy, vz);
    const double wi = node_w[i], wj = node_w[j];
    const double c = rw * (wi > wj ? wi : wj);
    const double rho[3] = {c * (yi.x - yj.x), c * (yi.y - yj.y), c * (yi.z - yj.z)};
    const double si = (qi[0] * qi[0] + qi[1] * qi[1]) + (qi[2] * qi[2] + qi[3] * qi[3]);
    const double sj = (qj[0] * qj[0] + qj[1] * qj[1]) + (qj[2] * qj[2] + qj[3] * qj[3]);
    // J_i = c [ -[y_i]x | s_i I ],  J_j = -c [ -[y_j]x | s_j I ]   (3 x 6 each), column `col6` on demand
    auto Jcol = [&](const D3 &y, double sgn, double sc, int col6, double (&out)[3]) {
        // -[y]x = [[0, y2, -y1], [-y2, 0, y0], [y1, -y0, 0]]
        const double m[3][3] = {{0.0, y.z, -y.y}, {-y.z, 0.0, y.x}, {y.y, -y.x, 0.0}};
#pragma unroll
        for (int r = 0; r < 3; ++r) out[r] = col6 < 3 ? sgn * c * m[r][col6 % 3] : (r == col6 - 3 ? sgn * c * sc : 0.0);
    };
    if (lane < 36) {
        const int a = lane / 6, b = lane - 6 * a;
        double ia[3], ib[3], ja[3], jb[3];
        Jcol(yi, 1.0, si, a, ia); Jcol(yi, 1.0, si, b, ib);
        Jcol(yj, -1.0, sj, a, ja); Jcol(yj, -1.0, sj, b, jb);
        const double vii = (ia[0] * ib[0] + ia[1] * ib[1]) + ia[2] * ib[2];
        const double vjj = (ja[0] * jb[0] + ja[1] * jb[1]) + ja[2] * jb[2];
        const double vij = (ia[0] * jb[0] + ia[1] * jb[1]) + ia[2] * jb[2];
        const double vji = (ja[0] * ib[0] + ja[1] * ib[1]) + ja[2] * ib[2];
        if (partial_reg) {
            double *P = partial_reg + (size_t)t * NE2;
            P[gn_gram_index(2, 0, 0, a, b)] = vii;                 // (whole 6x6 sub-blocks: lane (b,a) computes the same products)
            P[gn_gram_index(2, 1, 1, a, b)] = vjj;
            P[gn_gram_index(2, 0, 1, a, b)] = vij;
            if (lane < 6) {
                double gi[3], gj[3];
                Jcol(yi, 1.0, si, lane, gi); Jcol(yj, -1.0, sj, lane, gj);
                P[kJtr2 + lane] = (gi[0] * rho[0] + gi[1] * rho[1]) + gi[2] * rho[2];
                P[kJtr2 + 6 + lane] = (gj[0] * rho[0] + gj[1] * rho[1]) + gj[2] * rho[2];
            }
            if (lane == 0) { P[kCost2] = 0.5 * ((rho[0] * rho[0] + rho[1] * rho[1]) + rho[2] * rho[2]); P[kCost2 + 1] = 0.0; P[kLive2] = 1.0; }
            return;
        }
        const int bii = find_block(row_ptr, col, i, i), bjj = find_block(row_ptr, col, j, j);
        const int bij = find_block(row_ptr, col, i, j), bji = find_block(row_ptr, col, j, i);
        if (bii >= 0 && vii != 0.0) atomicAdd(vals + 36 * (size_t)bii + lane, vii);
        if (bjj >= 0 && vjj != 0.0) atomicAdd(vals + 36 * (size_t)bjj + lane, vjj);
        if (bij >= 0 && vij != 0.0) atomicAdd(vals + 36 * (size_t)bij + lane, vij);
        if (bji >= 0 && vji != 0.0) atomicAdd(vals + 36 * (size_t)bji + lane, vji);
        if (lane < 6) {
            double gi[3], gj[3];
            Jcol(yi, 1.0, si, lane, gi); Jcol(yj, -1.0, sj, lane, gj);
            atomicAdd(rhs + 6 * i + lane, (gi[0] * rho[0] + gi[1] * rho[1]) + gi[2] * rho[2]);
            atomicAdd(rhs + 6 * j + lane, (gj[0] * rho[0] + gj[1] * rho[1]) + gj[2] * rho[2]);
        }
        if (lane == 0) atomicAdd(cost_count, 0.5 * ((rho[0] * rho[0] + rho[1] * rho[1]) + rho[2] * rho[2]));
    }
}

__global__ __launch_bounds__(256) void gn_build_reg_kernel(const int *__restrict__ node_nbr, int N, int k,
                                                            const double *__restrict__ node_dq,
                                                            const double *__restrict__ node_pos,
                                                            const double *__restrict__ node_w, double rw,
                                                            const int *__restrict__ row_ptr, const int *__restrict__ col,
                                                            double *__restrict__ vals, double *__restrict__ rhs,
                                                            double *__restrict__ cost_count, double *__restrict__ partial_reg) {
    gn_reg_pairs((int)blockIdx.x, node_nbr, N, k, node_dq, node_pos, node_w, rw, row_ptr, col, vals, rhs, cost_count, partial_reg);
}

// ------------------------------------------------------------------------------- PCG
// Solves (A + lm_abs I + lm_rel diag(A)) x = -rhs with block-Jacobi preconditioning; one thread
// per node row; scalars live in `scal` (3 doubles per iteration: rz, pAp, rz_next).
// Row `r` (r = 0..5, may differ between lanes) of the same inverse, the same bits as inv6's row r, without the 36 outputs: the
// persistent PCG wants one row per lane and was spilling registers around the full inverse in its 1 024-thread form.
// (A^-1)[r][j] = sum_k Li[k][r] Li[k][j] over k >= max(r, j); Li[k][r] is picked from the k-th row with compares (no dynamic
// index), and is exactly 0 for k < r, so the sum may start at k = j: the extra terms add +0.0 to a +0.0.
__device__ __forceinline__ void inv6_row(const double *A, int r, double *row) {
    double L[6][6], Li[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) { L[i][j] = 0.0; Li[i][j] = 0.0; }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= L[j][k] * L[j][k];
        d = d > 0.0 ? sqrt(d) : 1.0;
        L[j][j] = d;
        const double id = 1.0 / d;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i > j) {
                double v = A[6 * i + j];
#pragma unroll
                for (int k = 0; k < 6; ++k) if (k < j) v -= L[i][k] * L[j][k];
                L[i][j] = v * id;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i >= c) {
                double v = i == c ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) if (k >= c && k < i) v -= L[i][k] * Li[k][c];
                Li[i][c] = v / L[i][i];
            }
        }
    }
    double lr[6];                                               // Li[k][r]
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double v = Li[k][0];
#pragma unroll
        for (int c = 1; c < 6; ++c) v = r == c ? Li[k][c] : v;
        lr[k] = v;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k >= j) v += lr[k] * Li[k][j];
        row[j] = v;
    }
}

__device__ __forceinline__ void inv6(const double *A, double *Ainv) {
    // A = L L^T (SPD after damping), A^-1 = L^-T L^-1; every loop has compile-time bounds so the
    // 6x6 arrays live in registers.  A non-positive pivot (rank-deficient block) is replaced by 1:
    // the preconditioner only has to be SPD, not exact.
    double L[6][6], Li[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) { L[i][j] = 0.0; Li[i][j] = 0.0; }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= L[j][k] * L[j][k];
        d = d > 0.0 ? sqrt(d) : 1.0;
        L[j][j] = d;
        const double id = 1.0 / d;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i > j) {
                double v = A[6 * i + j];
#pragma unroll
                for (int k = 0; k < 6; ++k) if (k < j) v -= L[i][k] * L[j][k];
                L[i][j] = v * id;
            }
        }
    }
    // Li = L^-1 (lower triangular), column by column
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i >= c) {
                double v = i == c ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) if (k >= c && k < i) v -= L[i][k] * Li[k][c];
                Li[i][c] = v / L[i][i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k >= i && k >= j) v += Li[k][i] * Li[k][j];
            Ainv[6 * i + j] = v;
        }
}

struct PcgParams {
    int N;
    double lm_abs, lm_rel;
};

// The multi-launch path's dot products without atomics: every workgroup of the producing launch stores ONE partial (its waves'
// values added in a fixed order), every workgroup of the consuming launch adds all partials in the same fixed order -- the same
// bits in every workgroup, every run and on every rank (with atomicAdd the order, hence the last bits, changed from run to run).
__device__ __forceinline__ void wg_store_partial(double wave_value, double *slot) {     // all 256 threads; wave_value on lane 0
    __shared__ double s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = wave_value;
    __syncthreads();
    if (threadIdx.x == 0) slot[blockIdx.x] = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
}
__device__ __forceinline__ double wg_sum_partials(const double *__restrict__ part, int n) {   // all 256 threads
    __shared__ double s_sum[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    const double v = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(256) void pcg_init_kernel(const int *__restrict__ row_ptr, const int *__restrict__ col,
                                                        double *__restrict__ vals, const double *__restrict__ rhs,
                                                        const PcgParams p, double *__restrict__ Minv, double *__restrict__ x,
                                                        double *__restrict__ r, double *__restrict__ pv, double *__restrict__ rz_part) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    double rz = 0.0;
    if (a < p.N) {
    const int blk = find_block(row_ptr, col, a, a);
    double D[36];
    for (int i = 0; i < 36; ++i) D[i] = blk >= 0 ? vals[36 * (size_t)blk + i] : 0.0;
    for (int i = 0; i < 6; ++i) D[7 * i] = D[7 * i] + p.lm_abs + p.lm_rel * D[7 * i];
    if (blk >= 0) for (int i = 0; i < 6; ++i) vals[36 * (size_t)blk + 7 * i] = D[7 * i];     // damping lives in the matrix
    double Di[36];
    inv6(D, Di);
    for (int i = 0; i < 36; ++i) Minv[36 * (size_t)a + i] = Di[i];
    double rl[6], zl[6];
    for (int i = 0; i < 6; ++i) { rl[i] = -rhs[6 * a + i]; x[6 * a + i] = 0.0; r[6 * a + i] = rl[i]; }
    for (int i = 0; i < 6; ++i) {
        double z = 0.0;
        for (int j = 0; j < 6; ++j) z += Di[6 * i + j] * rl[j];
        zl[i] = z;
        pv[6 * a + i] = z;                 // z0; the first SpMV takes p = z (beta = 0)
        rz += rl[i] * z;
    }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rz += __shfl_xor(rz, o, 64);
    wg_store_partial(rz, rz_part);
}

// One 64-lane wave per node row: lane = (block slot b in 0..9) x (output component i in 0..5);
// each lane multiplies row i of its block with the 6 entries of p at the block's column node, the
// ten slots are folded with shuffles, lanes 0..5 hold y and lane 0 adds p.Ap once.
// The direction update p = z + beta p is folded in: every row forms its neighbours' new p on the
// fly from (z, p_prev, beta) and publishes its own new p in p_cur (ping-pong), so CG needs two
// launches per iteration.  scal_prev = {rz, pAp, rz_next} of the previous iteration (NULL: beta = 0).
__global__ __launch_bounds__(256) void pcg_spmv_kernel(const int *__restrict__ row_ptr, const int *__restrict__ col,
                                                        const double *__restrict__ vals, int N, const double *__restrict__ z,
                                                        const double *__restrict__ p_prev, double *__restrict__ p_cur,
                                                        double *__restrict__ Ap, const double *__restrict__ scal_prev,
                                                        double *__restrict__ scal, const double *__restrict__ rz_part, int n_rz_part,
                                                        double *__restrict__ pap_part) {
    const int lane = threadIdx.x & 63;
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    // r.z of this iteration = the partials of the launch that produced z (init or the previous update), added here
    const double rz_now = wg_sum_partials(rz_part, n_rz_part);
    const double rz = scal_prev[0];                                     // the previous iteration's r.z (0 in iteration 0: beta = 0)
    const double beta = rz != 0.0 ? rz_now / rz : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[0] = rz_now;         // rz of this iteration for update_xr and the next SpMV
    const bool row = a < N;
    const int slot = lane / 6, i = lane - 6 * slot;            // lanes 60..63: slot 10 (idle)
    double acc = 0.0;
    const int beg = row ? row_ptr[a] : 0, end = row ? row_ptr[a + 1] : 0;
    if (slot < 10) {
        for (int b = beg + slot; b < end; b += 10) {
            const double *B = vals + 36 * (size_t)b + 6 * i;
            const double *zj = z + 6 * col[b];
            const double *pj = p_prev + 6 * col[b];
            const double q0 = zj[0] + beta * pj[0], q1 = zj[1] + beta * pj[1], q2 = zj[2] + beta * pj[2];
            const double q3 = zj[3] + beta * pj[3], q4 = zj[4] + beta * pj[4], q5 = zj[5] + beta * pj[5];
            acc += ((B[0] * q0 + B[1] * q1) + (B[2] * q2 + B[3] * q3)) + (B[4] * q4 + B[5] * q5);
        }
    }
    double y = acc;
#pragma unroll
    for (int k = 1; k < 10; ++k) {
        const double o = __shfl(acc, lane + 6 * k, 64);
        y += (lane + 6 * k < 60) ? o : 0.0;
    }
    double contrib = 0.0;
    if (row && lane < 6) {
        const double pn = z[6 * a + lane] + beta * p_prev[6 * a + lane];
        p_cur[6 * a + lane] = pn;
        Ap[6 * a + lane] = y;
        contrib = pn * y;
    }
    contrib += __shfl_down(contrib, 4, 64);
    contrib += __shfl_down(contrib, 2, 64);
    contrib += __shfl_down(contrib, 1, 64);
    wg_store_partial(contrib, pap_part);                                 // p.Ap of this workgroup's four rows
}

// x += alpha p, r -= alpha Ap, z = Minv r, rz_next += r.z : one thread per unknown (6 per node; the
// node's six new residual entries are exchanged with shuffles inside the 6-lane group).
__global__ __launch_bounds__(256) void pcg_update_xr_kernel(int N, const double *__restrict__ Minv, double *__restrict__ x,
                                                             double *__restrict__ r, const double *__restrict__ pv,
                                                             const double *__restrict__ Ap, double *__restrict__ z,
                                                             const double *__restrict__ scal, const double *__restrict__ pap_part,
                                                             int n_pap_part, double *__restrict__ rz_part) {
    // 60 of the 64 lanes of a wave are used: 10 nodes per wave, 40 per block
    const int lane = threadIdx.x & 63;
    const int grp = lane / 6, i = lane - 6 * grp;
    const int a = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 10 + grp;
    const bool act = grp < 10 && a < N;
    const double rz = scal[0], pAp = wg_sum_partials(pap_part, n_pap_part);
    const double alpha = pAp != 0.0 ? rz / pAp : 0.0;
    double rn = 0.0;
    if (act) {
        const int u = 6 * a + i;
        x[u] += alpha * pv[u];
        rn = r[u] - alpha * Ap[u];
        r[u] = rn;
    }
    double zz = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double rj = __shfl(rn, 6 * grp + j, 64);
        if (act) zz += Minv[36 * (size_t)a + 6 * i + j] * rj;
    }
    double contrib = 0.0;
    if (act) { z[6 * a + i] = zz; contrib = rn * zz; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) contrib += __shfl_down(contrib, o, 64);
    wg_store_partial(contrib, rz_part);                                  // r.z of this workgroup's 40 rows (the next iteration's)
}

// ---- persistent PCG: the whole iteration loop in one launch --------------------------------------
// Every wave owns ONE node row for the whole solve: its 6x6 blocks (up to kRowCache per lane slot), its rows of the
// block-Jacobi inverse and its six entries of the CG vectors stay in registers; per iteration only the neighbours'
// published vectors are read (agent-scope loads) and ONE grid-wide reduction replaces the kernel boundaries.  The
// grid is sized so that all workgroups are co-resident (<= one per CU on at most half the CUs); every wait is bounded
// and an abort flag makes every wave leave if one ever times out (x is then NaN, never a hang).  Reductions: every
// workgroup adds its waves' values in LDS (fixed order) and publishes the partial; wave 0 reads all workgroups'
// partials and adds them in a fixed order: same bits every run and on every rank, no floating-point atomics, no
// counters, no cache-wide fences.  Measured (512 rows, tools/kbench_pcg.py): 3.0 us per iteration, of which ~1.8 us
// is the hand-off (stores becoming visible across the XCDs + one agent-scope load round trip of ~0.9 us).
constexpr int kRowCache = 3;               // blocks per lane slot held in registers (rows <= 30 blocks)
constexpr unsigned kSpinLimit = 1u << 22;  // default bound of a barrier's spin (~seconds); DFH_PCG_SPIN_LIMIT overrides (tests)
#ifndef DFH_PCG_POLL_GAP
#define DFH_PCG_POLL_GAP 1
#endif
#ifndef DFH_PCG_POLL_DELAY
#define DFH_PCG_POLL_DELAY 16
#endif
constexpr int kPollDelay = DFH_PCG_POLL_DELAY;   // s_sleep units (64 clocks) between a publish and the first look: a look costs a
                                                 // full round trip, one issued at once finds nothing (0 / 8 / 16 / 24 / 32: 4.25 / 3.73 / 3.45 / 3.63 / 3.83 us per iteration)
constexpr int kPollGap = DFH_PCG_POLL_GAP;       // s_sleep units (64 clocks) between two polls
constexpr int kMaxPcgBlocks = 512;         // persistent path only for grids up to this many workgroups

__device__ __forceinline__ double ld_agent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void apply_twist_one(double *__restrict__ d, double ox, double oy, double oz, double vx, double vy, double vz);

// ---- single-reduction PCG (Chronopoulos & Gear) -----------------------------------------------------------
// A grid-wide hand-off costs ~1.5-3 us across the eight XCDs (MI355X_MICROARCH.md, hand-off price list) and the
// textbook recurrence needs two reductions per iteration (p.Ap, then r.z).  This variant has ONE: with u = M^-1 r, w = A u,
//   gamma = r.u, delta = w.u  (both in the same reduction),  beta = gamma / gamma_old,
//   alpha = gamma / (delta - beta gamma / alpha_old),  p = u + beta p,  s = w + beta s  (= A p),
//   x += alpha p,  r -= alpha s,  u = M^-1 r,  w = A u.
// The exchange of the new u between rows would be a second synchronisation; it is avoided by linearity:
// u_new = u - alpha t with t = M^-1 s = v + beta t_old, v = M^-1 w, so every row publishes (u, v, t_old) BEFORE
// the reduction and its neighbours form its u_new themselves once alpha and beta are known -- with the same
// expression the owner uses, hence the same bits.  Same iterates as the textbook PCG in exact arithmetic.
//
// Hand-offs carry their own arrival flag.  Every published double (a workgroup's partial sums, a row's u, v, t) goes
// into a slot whose bits are all zero until the one 8-byte agent-scope store that fills it (a zero value is stored as
// -0.0), so a reader needs no barrier and the writer no store drain: it loads the slot and retries while the bits are
// zero.  The neighbours' (u, v, t) are requested right after a wave's own stores, i.e. while the reduction is still in
// flight, so an iteration's critical path is one store becoming visible plus one load (it was: drain the stores,
// publish the partial, poll the partials, then load the neighbours -- four trips).
//   * partial sums: a fresh pair of slots per workgroup and reduction (zeroed by the launch's memset);
//   * vectors: a ring of four phase regions {u, v, t} x 6N (zeroed by the memset); iteration `it` reads region it % 4
//     and publishes into (it + 1) % 4.  A row's wave clears its own entries of region (it - 1) % 4 after reduction `it`:
//     every reader was finished with them before it contributed to that reduction.  The wave's wait for its neighbours'
//     values in iteration it + 1 (loads issued after the clearing stores; vmcnt counts in issue order) proves the
//     clears complete; only then does the wave contribute to reduction it + 2 and later store the region's next
//     values (phase it + 3).  A reader asks for those only after it has seen reduction it + 2 complete, so it finds
//     zero bits or the new value, never the value of four phases ago.
// A wave whose wait runs out (spin_limit) or that sees the abort flag poisons its row with NaN: the NaN reaches every
// row through the next reduction, so x is NaN everywhere and nothing hangs.
struct BarrierLds2 {
    double wave_part[2][16];
    double total[2];
};
struct alignas(16) WaveLds {             // one wave's scratch for trading values between its lanes
    double q[kRowCache][64];
    double part[6][10];
    double w[6];
};
// orders a wave's LDS writes before its following LDS reads (the hardware executes one wave's LDS operations in
// order; this only keeps the compiler from moving them)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

__device__ __forceinline__ double nz_bits(double v) { return __double_as_longlong(v) == 0 ? -0.0 : v; }
__device__ __forceinline__ bool arrived(double v) { return __double_as_longlong(v) != 0; }

// workgroup barrier for LDS traffic only: vector-memory operations stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Wave-wide sums without the LDS crossbar: DPP row shifts (lanes without a source add 0), then the two cross-row
// broadcasts; the total is read from lane 63 into scalar registers, i.e. the result is wave-uniform.  Fixed order.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp0_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp0_f64<0x111>(v);                 // row_shr:1
    v += dpp0_f64<0x112>(v);                 // row_shr:2
    v += dpp0_f64<0x114>(v);                 // row_shr:4
    v += dpp0_f64<0x118>(v);                 // row_shr:8   -> lane 15 of every row: the row's sum
    v += dpp0_f64<0x142, 0xa>(v);            // row_bcast:15 into rows 1 and 3
    v += dpp0_f64<0x143, 0xc>(v);            // row_bcast:31 into rows 2 and 3
    return readlane_f64(v, 63);
}
__device__ __forceinline__ double sum6_f64(double v) {         // lanes 0..5 -> wave-uniform
    v += dpp0_f64<0x111>(v);
    v += dpp0_f64<0x112>(v);
    v += dpp0_f64<0x114>(v);
    return readlane_f64(v, 5);
}

struct PcgAbort {
    unsigned *flag;                      // this solve's abort flag (zero before the launch); flag[1]: "already counted"
    unsigned long long *count;           // the library's sticky per-device counter of timed-out solves
    unsigned *host_flag;                 // word in pinned host memory, set when the counter is bumped (dfh_pcg_status_peek)
    unsigned spin_limit;
    __device__ __forceinline__ bool raised() const { return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
    __device__ __forceinline__ void raise() const {                                        // one count per timed-out solve
        __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (atomicExch(flag + 1, 1u) == 0u) {
            atomicAdd(count, 1ull);
            if (host_flag) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
};

#ifdef DFH_PCG_TRACE     // experiment builds only (tools/build_variant.sh): wall-clock stamps of one wave's iteration phases
__device__ unsigned long long g_pcg_trace[64][16][12];
#define PCG_STAMP(k) do { if (lane == 0 && tw >= 0 && it < 16) { g_pcg_trace[tw][it][k] = wall_clock64(); if (k == 0) { g_pcg_trace[tw][it][9] = clock64(); g_pcg_trace[tw][it][10] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); } } } while (0)
#define GS_STAMP(k) do { if ((threadIdx.x & 63) == 0 && tr) tr[k] = wall_clock64(); } while (0)
#define PRO_STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == (DFH_PCG_TRACE) && blockIdx.x < 64) g_pcg_trace[blockIdx.x][15][k] = wall_clock64(); } while (0)
#else
#define PCG_STAMP(k) do {} while (0)
#define GS_STAMP(k) do {} while (0)
#define PRO_STAMP(k) do {} while (0)
#endif

// Two grid-wide sums in one pass; slots = 2 * gridDim.x doubles (workgroup b: 2b, 2b+1), zero bits before the launch.
// in_flight() runs in every wave between the publish and the wait: loads issued there travel beside the reduction.
// Returns NaN totals when the wait was given up.
template <class R, class H>
__device__ __forceinline__ void grid_sum2(double *slots, const PcgAbort &ab, BarrierLds2 *lds, double v0, double v1 /* wave-uniform */,
                                          double *s0, double *s1, bool fetch, R &&request, H &&here, int blk, int nblk,
                                          unsigned long long *tr = nullptr) {
    // fetch: the wave also wants its neighbours' published values: request() issues the loads, here() says whether the
    // last request found them all (wave-uniform).  Every wave keeps asking while the reduction is in flight, so the
    // values and the totals are usually both there one load latency after the slowest workgroup's stores land.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    if (lane == 0) { lds->wave_part[0][wave] = v0; lds->wave_part[1][wave] = v1; }
    lds_barrier();
    GS_STAMP(5);
#ifdef DFH_PCG_TRACE
    if (tr && lane == 0) tr[8] = 0ull;
#endif
    if (wave == 0 && lane < 2) {
        double v = 0.0;
        for (int w = 0; w < waves; ++w) v += lds->wave_part[lane][w];
        st_agent(slots + 2 * blk + lane, nz_bits(v));
    }
    bool have = !fetch;
    GS_STAMP(6);
    unsigned spins = 0;
    if (kPollDelay > 0) __builtin_amdgcn_s_sleep(kPollDelay);   // nothing can have arrived yet
    if (wave == 0) {
        const int nb = nblk;                            // lane l adds workgroups l, l + 64, ...
        double t0 = 0.0, t1 = 0.0;
        for (;;) {
            bool all = true;
            t0 = 0.0;
            t1 = 0.0;
            for (int b = lane; b < nb; b += 64) {
                const double p0 = ld_agent(slots + 2 * b), p1 = ld_agent(slots + 2 * b + 1);
                all = all && arrived(p0) && arrived(p1);
                t0 += p0;
                t1 += p1;
            }
            if (!have) {
                request();
                have = here();
            }
            if (__all(all)) break;
#ifdef DFH_PCG_TRACE
            if (tr && lane == 0) tr[8] += 1ull;                 // failed polls
#endif
            if (++spins > ab.spin_limit || ab.raised()) {
                if (lane == 0) ab.raise();
                t0 = t1 = __builtin_nan("");
                break;
            }
            __builtin_amdgcn_s_sleep(kPollGap);
        }
        t0 = wave_sum_f64(t0);
        t1 = wave_sum_f64(t1);
        GS_STAMP(7);
        if (lane == 0) { lds->total[0] = t0; lds->total[1] = t1; }
    } else {
        while (!have) {
            request();
            have = here();
            if (have) break;
            if (++spins > ab.spin_limit || ab.raised()) {      // (the caller's wait sees the flag and poisons the row)
                if (lane == 0) ab.raise();
                break;
            }
            __builtin_amdgcn_s_sleep(kPollGap);
        }
    }
    lds_barrier();
    *s0 = lds->total[0];
    *s1 = lds->total[1];
}

// MAXT = largest workgroup it is launched with: 512 leaves 256 VGPRs per lane (no spills in the prologue's 6x6 inverse)
template <int MAXT>
__global__ __launch_bounds__(MAXT) void pcg_cg1_kernel(const int *__restrict__ row_ptr, const int *__restrict__ col, double *vals,
                                                        const double *__restrict__ rhs, const PcgParams prm, int iters,
                                                        double *__restrict__ x, double *ring /* 4 x {u, v, t} x 6N */, double *part,
                                                        unsigned *abort_flag, unsigned spin_limit, unsigned long long *abort_count,
                                                        unsigned *abort_host, double *__restrict__ update_dq, double update_step,
                                                        int die_stride) {
    // die_stride > 1 (experiment, option pcg_one_xcd): the grid is die_stride times too large and only the workgroups whose index
    // is a multiple of it work -- with round-robin dispatch over the eight XCDs (stride 8) they all sit on ONE die; the others leave
    if (die_stride > 1 && (blockIdx.x % die_stride) != 0) return;
    const int blk = die_stride > 1 ? (int)blockIdx.x / die_stride : (int)blockIdx.x;
    const int nblk = die_stride > 1 ? (int)gridDim.x / die_stride : (int)gridDim.x;
    if (die_stride > 1 && threadIdx.x == 0)                                     // which dies really took part (bit = XCC_ID): flag[3]
        atomicOr(abort_flag + 3, 1u << (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15));
    // update_dq != NULL: the row's wave also applies its twist, update_dq[a] <- exp(update_step * x_a) (x) update_dq[a]
    // abort_count is read by dfh_pcg_status() at the caller's next synchronisation point
    __shared__ BarrierLds2 lds;
    __shared__ WaveLds wlds[MAXT / 64];
    const PcgAbort ab{abort_flag, abort_count, abort_host, spin_limit};
    const int N = prm.N;
    const size_t N6 = 6 * (size_t)N;
    const int lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    const int a = blk * waves + (threadIdx.x >> 6);
    const bool row = a < N;
    const int slot = lane / 6, i = lane - 6 * slot;            // lanes 60..63 idle in the SpMV
    const int beg = row ? row_ptr[a] : 0, end = row ? row_ptr[a + 1] : 0;
    const bool lead = row && lane < 6;
    const double rhs_i = lead ? rhs[6 * a + lane] : 0.0;      // (asked for now: needed after the 6x6 inverse)
    PRO_STAMP(0);
    // register cache of this row's blocks: lane (slot, i) holds row i of blocks beg+slot+10c; the diagonal block gets
    // its damping here (the damping lives in the matrix: it is also written back below)
    double Bc[kRowCache][6];
    int cj[kRowCache];
#pragma unroll
    for (int c = 0; c < kRowCache; ++c) {
        const int b = beg + slot + 10 * c;
        const bool have = slot < 10 && b < end;
        cj[c] = have ? col[b] : -1;
#pragma unroll
        for (int j = 0; j < 6; ++j) Bc[c][j] = have ? vals[36 * (size_t)b + 6 * i + j] : 0.0;
        if (have && cj[c] == a) {
#pragma unroll
            for (int j = 0; j < 6; ++j)                         // (static indices: a lane-dependent one would move Bc to scratch memory)
                if (j == i) {
                    Bc[c][j] = Bc[c][j] + prm.lm_abs + prm.lm_rel * Bc[c][j];
                    vals[36 * (size_t)b + 7 * i] = Bc[c][j];   // (only this lane reads that element, and it has)
                }
        }
    }
    PRO_STAMP(1);
    // block-Jacobi preconditioner: the damped diagonal block, from the register cache when it is there (six lanes hold
    // its rows: 36 shuffles instead of a binary search and a reload), else found and loaded the slow way
    double D[36];
    bool cached = false;
#pragma unroll
    for (int c = 0; c < kRowCache; ++c) {
        const unsigned long long m = __ballot(row && slot < 10 && cj[c] == a);
        if (m != 0ull && !cached) {                            // (wave-uniform)
            const int base = __ffsll((long long)m) - 1;        // lane (slot_d, 0)
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j) D[6 * r + j] = __shfl(Bc[c][j], base + r, 64);
            cached = true;
        }
    }
    if (!cached) {
        const int dblk = lead ? find_block(row_ptr, col, a, a) : -1;
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int t = 0; t < 36; ++t) D[t] = dblk >= 0 ? vals[36 * (size_t)dblk + t] : 0.0;
#pragma unroll
        for (int t = 0; t < 6; ++t) D[7 * t] = D[7 * t] + prm.lm_abs + prm.lm_rel * D[7 * t];
        __builtin_amdgcn_s_waitcnt(0);                          // every read of the undamped diagonal has returned
        if (dblk >= 0) {
            double dv = D[0];
#pragma unroll
            for (int rr = 1; rr < 6; ++rr) dv = lane == rr ? D[7 * rr] : dv;
            vals[36 * (size_t)dblk + 7 * lane] = dv;
        }
    }
    PRO_STAMP(2);
    double Mi[6];
    {
        double mr[6];
        inv6_row(D, lane < 6 ? lane : 0, mr);                   // every lane runs the factorisation: same cost as one lane
#pragma unroll
        for (int j = 0; j < 6; ++j) Mi[j] = lead ? mr[j] : 0.0;
    }
    PRO_STAMP(3);
    // The neighbours' published values of this lane's cached blocks (element i of node cj[c]); rows wider than the
    // cache read the rest of their neighbours one at a time (wait_for).
    double nu[kRowCache], nv[kRowCache], nt[kRowCache];
    const bool wide = __any(slot < 10 && beg + slot + 10 * kRowCache < end);
    auto request = [&](const double *P, bool with_vt) {
#pragma unroll
        for (int c = 0; c < kRowCache; ++c) {
            const bool have = cj[c] >= 0;
            const size_t j6 = 6 * (size_t)(have ? cj[c] : 0) + i;
            nu[c] = have ? ld_agent(P + j6) : 1.0;
            nv[c] = have && with_vt ? ld_agent(P + N6 + j6) : 1.0;
            nt[c] = have && with_vt ? ld_agent(P + 2 * N6 + j6) : 1.0;
        }
    };
    auto all_here = [&]() {
        bool all = true;
#pragma unroll
        for (int c = 0; c < kRowCache; ++c) all = all && arrived(nu[c]) && arrived(nv[c]) && arrived(nt[c]);
        return __all(all) != 0;
    };
    bool mine = true;                                           // false once one of this wave's waits was given up
    auto await = [&](const double *P, bool with_vt) {          // checks the last request first
        unsigned spins = 0;
        while (!all_here()) {
            if (++spins > spin_limit || ab.raised()) {
                if (lane == 0) ab.raise();
                mine = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            request(P, with_vt);
        }
    };
    auto wait_for = [&](const double *p) {                      // one published value (wide rows' tail)
        double v = ld_agent(p);
        unsigned spins = 0;
        while (!arrived(v)) {
            if (++spins > spin_limit || ab.raised()) {
                ab.raise();
                mine = false;
                return __builtin_nan("");
            }
            __builtin_amdgcn_s_sleep(1);
            v = ld_agent(p);
        }
        return v;
    };
    // y = A q for this row: qc[c] = element i of neighbour cj[c]'s vector, tail(j6) = element j6 of the vector for the
    // blocks beyond the cache.  Lanes trade values through the wave's own LDS scratch (a write, then wide reads: a
    // quarter of the instructions the cross-lane shuffles took): the six lanes of a slot read that neighbour's six
    // elements, lane i < 6 then reads and adds the ten slots' row-i partial sums (in slot order).  Result in lanes 0..5.
    WaveLds &wl = wlds[threadIdx.x >> 6];
    auto spmv = [&](const double (&qc)[kRowCache], auto &&tail) {
#pragma unroll
        for (int c = 0; c < kRowCache; ++c) wl.q[c][lane] = qc[c];
        wave_lds_sync();
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < kRowCache; ++c) {
            const double2 *qs = reinterpret_cast<const double2 *>(&wl.q[c][6 * (slot < 10 ? slot : 0)]);
            const double2 q01 = qs[0], q23 = qs[1], q45 = qs[2];
            if (cj[c] >= 0)
                acc += ((Bc[c][0] * q01.x + Bc[c][1] * q01.y) + (Bc[c][2] * q23.x + Bc[c][3] * q23.y)) + (Bc[c][4] * q45.x + Bc[c][5] * q45.y);
        }
        if (wide && slot < 10) {
            for (int b = beg + slot + 10 * kRowCache; b < end; b += 10) {
                const double *B = vals + 36 * (size_t)b + 6 * i;
                const size_t j6 = 6 * (size_t)col[b];
                const double q0 = tail(j6 + 0), q1 = tail(j6 + 1), q2 = tail(j6 + 2), q3 = tail(j6 + 3), q4 = tail(j6 + 4), q5 = tail(j6 + 5);
                acc += ((B[0] * q0 + B[1] * q1) + (B[2] * q2 + B[3] * q3)) + (B[4] * q4 + B[5] * q5);
            }
        }
        if (slot < 10) wl.part[i][slot] = acc;
        wave_lds_sync();
        const double2 *ps = reinterpret_cast<const double2 *>(&wl.part[lane < 6 ? lane : 0][0]);
        const double2 p01 = ps[0], p23 = ps[1], p45 = ps[2], p67 = ps[3], p89 = ps[4];
        return ((((((((p01.x + p01.y) + p23.x) + p23.y) + p45.x) + p45.y) + p67.x) + p67.y) + p89.x) + p89.y;
    };
    auto minv = [&](double v) {                                 // (M^-1 v)_lane from the six entries in lanes 0..5
        if (lane < 6) wl.w[lane] = v;
        wave_lds_sync();
        const double2 *ws = reinterpret_cast<const double2 *>(&wl.w[0]);
        const double2 w01 = ws[0], w23 = ws[1], w45 = ws[2];
        return ((((Mi[0] * w01.x + Mi[1] * w01.y) + Mi[2] * w23.x) + Mi[3] * w23.y) + Mi[4] * w45.x) + Mi[5] * w45.y;
    };
    // phase regions: u at +0, v at +N6, t at +2 N6 (pointer arithmetic, not a table of pointers: the accesses stay
    // global_load/global_store; a generic pointer's flat accesses would also count on lgkmcnt and stall the LDS barriers)
    auto region = [&](int k) { return ring + (size_t)(k & 3) * 3 * N6; };
    double xi = 0.0, ri = lead ? -rhs_i : 0.0, pi = 0.0, si = 0.0, ti = 0.0;
    double ui = minv(ri);
    if (lead) st_agent(region(0) + 6 * a + lane, nz_bits(ui));
    PRO_STAMP(4);
    request(region(0), false);
    await(region(0), false);
    PRO_STAMP(5);
    double wi = spmv(nu, [&](size_t j6) { return wait_for(region(0) + j6); });
    double vi = minv(wi);
    if (lead) {
        st_agent(region(0) + N6 + 6 * a + lane, nz_bits(vi));
        st_agent(region(0) + 2 * N6 + 6 * a + lane, -0.0);
    }
    if (!mine) ui = __builtin_nan("");
    double gamma = 0.0, delta = 0.0;
    double g = sum6_f64(lead ? ri * ui : 0.0), d = sum6_f64(lead ? wi * ui : 0.0);
    PRO_STAMP(6);
    double gamma_prev = 0.0, alpha_prev = 0.0;
#ifdef DFH_PCG_TRACE
    const int tw = (threadIdx.x >> 6) == (DFH_PCG_TRACE) && blockIdx.x < 64 ? (int)blockIdx.x : -1;
#endif
    for (int it = 0; it < iters; ++it) {
        const double *cur = region(it);
        const bool last = it == iters - 1;
        PCG_STAMP(0);
        grid_sum2(part + (size_t)it * 2 * nblk, ab, &lds, g, d, &gamma, &delta, !last, [&]() { request(cur, true); }, all_here, blk, nblk
#ifdef DFH_PCG_TRACE
                  , tw >= 0 && it < 16 ? &g_pcg_trace[tw][it][0] : nullptr
#endif
        );
        PCG_STAMP(1);
        const double beta = gamma_prev != 0.0 ? gamma / gamma_prev : 0.0;
        const double denom = alpha_prev != 0.0 ? delta - (beta * gamma) / alpha_prev : delta;
        const double alpha = denom != 0.0 ? gamma / denom : 0.0;
        if (lead) {
            pi = ui + beta * pi;
            si = wi + beta * si;
            ti = vi + beta * ti;
            xi += alpha * pi;
            ri = ri - alpha * si;
            ui = ui - alpha * ti;
        }
        if (last) break;
        PCG_STAMP(2);
        await(cur, true);
        PCG_STAMP(3);
        double qc[kRowCache];
#pragma unroll
        for (int c = 0; c < kRowCache; ++c) qc[c] = nu[c] - alpha * (nv[c] + beta * nt[c]);
        wi = spmv(qc, [&](size_t j6) { return wait_for(cur + j6) - alpha * (wait_for(cur + N6 + j6) + beta * wait_for(cur + 2 * N6 + j6)); });
        vi = minv(wi);
        double *nxt = region(it + 1);
        if (lead) {
            st_agent(nxt + 6 * a + lane, nz_bits(ui));
            st_agent(nxt + N6 + 6 * a + lane, nz_bits(vi));
            st_agent(nxt + 2 * N6 + 6 * a + lane, nz_bits(ti));
        }
        // region (it - 1) % 4: every reader was done with it before reduction `it`.  Cleared here, behind the publishing stores
        // (issued before the check of the neighbours' values, the clears' acknowledgements were waited for with the loads).
        if (it >= 1 && lead) {
            double *old = region(it - 1);
            st_agent(old + 6 * a + lane, 0.0);
            st_agent(old + N6 + 6 * a + lane, 0.0);
            st_agent(old + 2 * N6 + 6 * a + lane, 0.0);
        }
        if (!mine) ui = __builtin_nan("");
        g = sum6_f64(lead ? ri * ui : 0.0);
        d = sum6_f64(lead ? wi * ui : 0.0);
        PCG_STAMP(4);
        gamma_prev = gamma;
        alpha_prev = alpha;
    }
    if (!update_dq) {
        if (lead) x[6 * a + lane] = xi;
        return;
    }
    // The twist update is ALL OR NOTHING (round 4).  A time-out that falls into the last reduction leaves some workgroups with
    // finished rows and others with NaN; a wave applying its own row's step (round 3) then left node_dq half updated.  Now every
    // workgroup publishes its rows' x and takes a ticket (flag[2], zeroed with the scalars); the workgroup that draws the last
    // ticket knows that every other one is done, looks at the abort flag and at every row's x, and applies all N twists or none:
    // after a timed-out solve node_dq is what it was before the solve.
    if (lead) st_agent(x + 6 * a + lane, xi);
    __shared__ unsigned s_ticket;
    __syncthreads();                                                // (this workgroup's x stores are issued)
    if (threadIdx.x == 0)
        s_ticket = __hip_atomic_fetch_add(abort_flag + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);   // release: after the x stores
    __syncthreads();
    if (s_ticket != (unsigned)nblk - 1u) return;
    bool bad = ab.raised();
    for (int r = (int)threadIdx.x; r < 6 * N; r += (int)blockDim.x) {
        const double v = ld_agent(x + r);
        bad = bad || !(fabs(v) < __builtin_huge_val());
    }
    if (__syncthreads_or(bad ? 1 : 0)) return;
    for (int r = (int)threadIdx.x; r < N; r += (int)blockDim.x) {
        const double *xr = x + 6 * (size_t)r;
        apply_twist_one(update_dq + 8 * (size_t)r, update_step * ld_agent(xr), update_step * ld_agent(xr + 1), update_step * ld_agent(xr + 2),
                        update_step * ld_agent(xr + 3), update_step * ld_agent(xr + 4), update_step * ld_agent(xr + 5));
    }
}

// dq_a <- exp(xi_a) (x) dq_a  (exp: rotation exp(omega), translation v; oracle/gn_np.py)
// dq <- exp(step * xi) (x) dq for one node (exp: rotation exp(omega), translation v; oracle/gn_np.py)
__device__ __forceinline__ void apply_twist_one(double *__restrict__ d, double ox, double oy, double oz, double vx, double vy, double vz) {
    const double th = sqrt(ox * ox + oy * oy + oz * oz);
    const double half = 0.5 * th;
    const double s = th < 1e-8 ? 0.5 - th * th / 48.0 : sin(half) / th;
    const Q4 q{cos(half), s * ox, s * oy, s * oz};
    const Q4 qe = qscale(qmul(qpure(vx, vy, vz), q), 0.5);
    const Q4 r{d[0], d[1], d[2], d[3]}, dd{d[4], d[5], d[6], d[7]};
    const Q4 nr = qmul(q, r);
    const Q4 nd = qadd(qmul(q, dd), qmul(qe, r));
    d[0] = nr.w; d[1] = nr.x; d[2] = nr.y; d[3] = nr.z;
    d[4] = nd.w; d[5] = nd.x; d[6] = nd.y; d[7] = nd.z;
}

// dq <- exp(factor * log(dq)): the node's rigid motion scaled towards the identity (factor in [0, 1]) along its own screw.
// log of a dual quaternion (q | qe) with q = |q| (cos(t/2), sin(t/2) n): omega = t n, v = 2 vec(qe q*) / |q|^2 -- the inverse of
// apply_twist_one's exp for a unit q; a non-unit q (the solve never renormalises) comes back unit.
__global__ __launch_bounds__(256) void relax_twist_kernel(double *__restrict__ node_dq, int N, double factor) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= N) return;
    double *d = node_dq + 8 * (size_t)a;
    const double w = d[0], x = d[1], y = d[2], z = d[3];
    const double n2 = (w * w + x * x) + (y * y + z * z);
    if (!(n2 > 1e-300) || !(n2 < 1e300)) return;                               // (zero / non-finite: left alone)
    const double vn = sqrt(x * x + y * y + z * z);
    const double th = 2.0 * atan2(vn, w);                                      // rotation angle, (-2 pi, 2 pi]
    const double k = vn > 1e-12 ? th / vn : 2.0 / sqrt(n2);                    // omega = k (x, y, z)
    // (0, v) = 2 qe q* / |q|^2
    const double e0 = d[4], e1 = d[5], e2 = d[6], e3 = d[7];
    const double inv = 2.0 / n2;
    const double vx = inv * (-e0 * x + e1 * w - e2 * z + e3 * y);
    const double vy = inv * (-e0 * y + e2 * w - e3 * x + e1 * z);
    const double vz = inv * (-e0 * z + e3 * w - e1 * y + e2 * x);
    d[0] = 1.0; d[1] = d[2] = d[3] = d[4] = d[5] = d[6] = d[7] = 0.0;
    apply_twist_one(d, factor * k * x, factor * k * y, factor * k * z, factor * vx, factor * vy, factor * vz);
}

// ---- the rigid mode of the normal equations (round 4) ------------------------------------------------------------
// Ten block-Jacobi PCG iterations barely move the smoothest mode of the system -- all nodes moving together -- which the
// regulariser does not penalise and the preconditioner does not see: of a pure 0.6-voxel translation the shipped ten GN
// iterations recover 28 % along the normals, the exactly solved loop 70 % (tests/golden/solve_recovery.json).  The coarse
// correction: restrict the system to ONE twist shared by all nodes, x_a = xi for every a -- A_g = sum of all 6x6 blocks,
// g_g = sum of all J^T r -- solve (A_g + lm diag A_g) xi = -g_g and apply xi to every node.  kGlobalWgs workgroups add their
// share of the blocks (wave w of the grid: blocks w, w + n_waves, ...; lane e < 36 one matrix entry, lanes 36..41 the J^T r
// entries of nodes w, w + n_waves, ...), publish 42 partial sums, and the workgroup that draws the last ticket adds the
// partials in index order (same bits every run), solves by Cholesky and applies the twist.
constexpr int kGlobalWgs = 64;
__global__ __launch_bounds__(256) void gn_global_step_kernel(const double *__restrict__ vals, int n_blocks, const double *__restrict__ rhs, int N,
                                                              double lm_rel, double *__restrict__ node_dq, double *__restrict__ xi_out,
                                                              double *__restrict__ scratch /* kGlobalWgs x 42 partials | ticket */) {
    __shared__ double part[4][42];
    __shared__ double sA[36], sg[6], sxi[6];
    __shared__ unsigned s_ticket;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
    double acc = 0.0;
    if (lane < 36) {
        for (int b = wave; b < n_blocks; b += n_waves) acc += vals[36 * (size_t)b + lane];
    } else if (lane < 42) {
        for (int a = wave; a < N; a += n_waves) acc += rhs[6 * (size_t)a + (lane - 36)];
    }
    if (lane < 42) part[wv][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 42)
        __hip_atomic_store(scratch + 42 * (size_t)blockIdx.x + threadIdx.x,
                           ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    unsigned *ticket = reinterpret_cast<unsigned *>(scratch + 42 * (size_t)gridDim.x);
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    if (threadIdx.x < 42) {
        double v = 0.0;
        for (unsigned w = 0; w < gridDim.x; ++w) v += __hip_atomic_load(scratch + 42 * (size_t)w + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 36) sA[threadIdx.x] = v; else sg[threadIdx.x - 36] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double D[36];
#pragma unroll
        for (int e = 0; e < 36; ++e) D[e] = 0.5 * (sA[e] + sA[6 * (e % 6) + e / 6]);        // (symmetric up to summation order: symmetrised)
#pragma unroll
        for (int d = 0; d < 6; ++d) D[7 * d] = D[7 * d] + lm_rel * D[7 * d];
        double row[6];
        inv6_row(D, (int)threadIdx.x, row);
        double x = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) x -= row[j] * sg[j];
        const bool ok = fabs(x) < 1e6;                                                       // (NaN / a singular system: no step)
        sxi[threadIdx.x] = ok ? x : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < 6 && xi_out) xi_out[threadIdx.x] = sxi[threadIdx.x];
    if (threadIdx.x == 0) *ticket = 0u;                                                      // (ready for the next call)
    bool all_ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) all_ok = all_ok && sxi[j] == sxi[j];
    if (!all_ok) return;
    for (int a = threadIdx.x; a < N; a += 256) apply_twist_one(node_dq + 8 * (size_t)a, sxi[0], sxi[1], sxi[2], sxi[3], sxi[4], sxi[5]);
}

// The rigid mode from a SUBSAMPLE of the data rows, without building the block system (round 4): for a twist shared by all
// nodes a sample's Jacobian is the sum of its k node blocks, J_g = sum_a J_a (6 entries), so A_g = sum_s J_g^T J_g and
// g_g = sum_s J_g^T r need neither runs nor a gather.  Every `stride`-th 128-sample tile (the samples are sorted by node tuple:
// a uniform thinning of the surface) is associated and differentiated exactly as in gn_build_data_kernel (same Huber weights);
// the regulariser is left out (a common left twist rotates every regulariser residual rigidly: it only damps this mode).  Per
// tile 21 + 6 sums (+ objective, count) in a fixed order; gn_global_finish_kernel adds the workgroups' partials in index order,
// damps, solves and applies the twist to every node (sharded samples: it stops at the 29 sums, an all-reduce goes in between and
// gn_global_apply_kernel does the rest).  Restated in oracle/gn_np.global_step_sampled.
constexpr int kGlobalVals = 29;                     // 21 upper entries of A_g | 6 of g_g | objective | valid count
constexpr int kGlobalGrid = 1536;                   // workgroups of the rows kernel = partial sets (fixed: the summation order must not follow the device)
// The sums on the matrix cores, like the data rows' Gram matrices: a wave writes {J_g (6) | r | 0} of its 64 samples to LDS and
// accumulates X^T X with v_mfma_f64_16x16x4 (16 steps of four samples per tile; A_g and g_g are its entries (i <= j < 6) and
// (i, 6)); the accumulator is four doubles per lane where 27 running sums per thread made the kernel a 256-VGPR one: one wave
// per SIMD, a tile's whole chain of dependent loads exposed -- 65 us for config 3's 762 tiles, 131 us for config 5's 5.2 k.
template <int K>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(3, 8))) void gn_global_rows_kernel(const double *__restrict__ spos, const double *__restrict__ snrm,
                                                              const int *__restrict__ nbr, const double *__restrict__ wts,
                                                              const double *__restrict__ node_dq, const BuildParams p, int stride, long n_sub,
                                                              double *__restrict__ tile_part, const AssocArgs aa) {
    __shared__ double sPart[kTileWaves][kGlobalVals];
    __shared__ double sX[kTile * 8];
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    double obj_acc = 0.0, cnt_acc = 0.0;
    // a workgroup walks the tiles blockIdx.x, + gridDim.x, ... of the thinned list and keeps its sums: at most kGlobalGrid
    // partial sets for the finish kernel (one set per TILE made that kernel's serial adds the whole step: 1.3 ms)
    for (long sub = blockIdx.x; sub < n_sub; sub += gridDim.x) {
        const long s = sub * stride * kTile + tid;
        double jg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, rr = 0.0;
        if (s < p.S) {
            int idx[kKMaxS];
            double w[kKMaxS];
#pragma unroll
            for (int j = 0; j < kKMaxS; ++j) {
                idx[j] = j < K ? nbr[(size_t)s * K + j] : 0;
                w[j] = j < K ? wts[(size_t)s * K + j] : 0.0;
            }
            double bh[8];
            const double nb = blend_static(node_dq, idx, w, K, bh);
            const double pfx = round_f32(spos[3 * (size_t)s]), pfy = round_f32(spos[3 * (size_t)s + 1]), pfz = round_f32(spos[3 * (size_t)s + 2]);
            const D3 x1 = dqb_warp_exact(bh, pfx, pfy, pfz);
            const D3 xp = dqb_warp_exact(p.lw.q, round_f32(x1.x), round_f32(x1.y), round_f32(x1.z));
            double c[3];
            const bool ok = aa.views ? associate_views<float>(aa.ap, aa.views, aa.n_views, xp, c) : associate_point<float>(aa.ap, aa.depth, xp, c);
            if (ok) {
                double Jrow[6 * K];
                double r = data_row_from(node_dq, idx, w, K, p.lw.q, bh, nb, pfx, pfy, pfz, xp, snrm[3 * (size_t)s], snrm[3 * (size_t)s + 1],
                                         snrm[3 * (size_t)s + 2], c[0], c[1], c[2], Jrow);
                double obj = 0.5 * r * r, sc = 1.0;
                if (p.huber > 0.0 && fabs(r) > p.huber) {
                    obj = p.huber * (fabs(r) - 0.5 * p.huber);
                    sc = sqrt(p.huber / fabs(r));
                }
                rr = r * sc;
#pragma unroll
                for (int c6 = 0; c6 < 6; ++c6) {
                    double v = 0.0;
#pragma unroll
                    for (int a = 0; a < K; ++a) v += Jrow[6 * a + c6];
                    jg[c6] = v * sc;
                }
                obj_acc += obj;
                cnt_acc += 1.0;
            }
        }
        double *row = sX + 8 * tid;
#pragma unroll
        for (int c6 = 0; c6 < 6; ++c6) row[c6] = jg[c6];
        row[6] = rr; row[7] = 0.0;
        // (a wave reads only the rows its own lanes wrote; its LDS operations execute in order: no workgroup barrier)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {                                     // A[i = li][k = lk] = B[k = lk][j = li] = X[64 wv + 4 st + lk][li]
            const double x = li < 8 ? sX[8 * (64 * wv + 4 * st + lk) + li] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {                                          // C/D: column = lane & 15, row = (lane >> 4) + 4 r
        const int pa = lk + 4 * r4, pb = li;
        if (pa < 6 && pb >= pa && pb < 6) sPart[wv][pa * 6 - (pa * (pa - 1)) / 2 + (pb - pa)] = acc[r4];
        else if (pa < 6 && pb == 6) sPart[wv][21 + pa] = acc[r4];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { obj_acc += __shfl_xor(obj_acc, o, 64); cnt_acc += __shfl_xor(cnt_acc, o, 64); }
    if (lane == 0) { sPart[wv][27] = obj_acc; sPart[wv][28] = cnt_acc; }
    __syncthreads();
    if (tid < kGlobalVals) {
        double v = sPart[0][tid];
#pragma unroll
        for (int w_ = 1; w_ < kTileWaves; ++w_) v += sPart[w_][tid];
        tile_part[(size_t)blockIdx.x * kGlobalVals + tid] = v;
    }
}

// the damped 6 x 6 solve and the twist for every node, from the 29 sums in LDS (sv); sxi: scratch
__device__ __forceinline__ void global_solve_apply(const double *sv, double *sxi, double lm_rel, int N, double *__restrict__ node_dq,
                                                   double *__restrict__ xi_out) {
    if (threadIdx.x < 6) {
        double D[36];
        int q = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) { D[6 * i + j] = sv[q]; D[6 * j + i] = sv[q]; ++q; }
#pragma unroll
        for (int d = 0; d < 6; ++d) D[7 * d] = D[7 * d] + lm_rel * D[7 * d];
        double row[6];
        inv6_row(D, (int)threadIdx.x, row);
        double x = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) x -= row[j] * sv[21 + j];
        sxi[threadIdx.x] = (fabs(x) < 1e6 && sv[28] >= 6.0) ? x : 0.0;             // (NaN, a singular system, hardly any data: no step)
    }
    __syncthreads();
    if (xi_out) {
        if (threadIdx.x < 6) xi_out[threadIdx.x] = sxi[threadIdx.x];
        if (threadIdx.x == 6) xi_out[6] = sv[27];
        if (threadIdx.x == 7) xi_out[7] = sv[28];
    }
    for (int a = threadIdx.x; a < N; a += blockDim.x) apply_twist_one(node_dq + 8 * (size_t)a, sxi[0], sxi[1], sxi[2], sxi[3], sxi[4], sxi[5]);
}

// the workgroups' partials added in index order (32 chunks of consecutive sets, sixteen loads in flight per thread, then the
// chunks): 29 sums; APPLY: the solve and the twists in the same launch (one rank: nothing to all-reduce in between).  The old
// pair -- 8 chunks of 128 dependent load-and-add steps, then a launch for the solve -- took 29 + 8 us.
template <bool APPLY>
__global__ __launch_bounds__(1024) void gn_global_finish_kernel(const double *__restrict__ tile_part, int n_sets, double *__restrict__ sums,
                                                                 double lm_rel, int N, double *__restrict__ node_dq, double *__restrict__ xi_out) {
    __shared__ double part[32][32];
    __shared__ double sv[32], sxi[6];
    const int e = threadIdx.x & 31, chunk = threadIdx.x >> 5;
    const int per = (n_sets + 31) / 32;
    const int t0 = chunk * per, t1 = min(n_sets, t0 + per);
    double v = 0.0;
    for (int t = t0; t < t1; t += 16) {
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = (t + j < t1 && e < kGlobalVals) ? tile_part[(size_t)(t + j) * kGlobalVals + e] : 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += x[j];
    }
    part[chunk][e] = v;
    __syncthreads();
    if (threadIdx.x < kGlobalVals) {
        double a = part[0][threadIdx.x];
#pragma unroll
        for (int c = 1; c < 32; ++c) a += part[c][threadIdx.x];
        sv[threadIdx.x] = a;
        sums[threadIdx.x] = a;
    }
    __syncthreads();
    if (APPLY) global_solve_apply(sv, sxi, lm_rel, N, node_dq, xi_out);
}

// (A_g + lm diag A_g) xi = -g_g from the 29 sums (after an all-reduce over ranks, where the samples are sharded), xi to every node
__global__ __launch_bounds__(256) void gn_global_apply_kernel(const double *__restrict__ sums, double lm_rel, int N, double *__restrict__ node_dq,
                                                               double *__restrict__ xi_out /* 6 | objective, count */) {
    __shared__ double sv[kGlobalVals], sxi[6];
    if (threadIdx.x < kGlobalVals) sv[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
    global_solve_apply(sv, sxi, lm_rel, N, node_dq, xi_out);
}

__global__ __launch_bounds__(256) void apply_twist_kernel(double *__restrict__ node_dq, const double *__restrict__ xi, int N,
                                                           double step) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= N) return;
    apply_twist_one(node_dq + 8 * a, step * xi[6 * a], step * xi[6 * a + 1], step * xi[6 * a + 2], step * xi[6 * a + 3],
                    step * xi[6 * a + 4], step * xi[6 * a + 5]);
}

}  // namespace dfh

// =================================================================================== C ABI
static int fill_assoc_params(dfh::AssocParams &p, const double lw_dq[8], int H, int W, const double K[9], const double Kinv[9],
                             const double lw_cam[12], double scale, const double center[3], double half, double max_dist, int knn);

extern "C" {

int dfh_residual_rigid(const double *verts, const double *normals, const double *corr, int n, const double x[8],
                       double *out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n >= 0, "dfh_residual_rigid: negative count");
    if (n == 0) return DFH_OK;
    DFH_REQUIRE(verts && normals && corr && x && out, "dfh_residual_rigid: null pointer");
    DQ q;
    for (int i = 0; i < 8; ++i) q.q[i] = x[i];
    hipLaunchKernelGGL(residual_rigid_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, verts, normals, corr, n, q, out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_gn_build_rigid(const double *verts, const double *normals, const double *corr, const unsigned char *valid, int n,
                       const double x[8], double *out44, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n >= 0 && x && out44, "dfh_gn_build_rigid: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DFH_HIP_CHECK(hipMemsetAsync(out44, 0, sizeof(double) * 44, s));
    if (n == 0) return DFH_OK;
    DFH_REQUIRE(verts && normals && corr, "dfh_gn_build_rigid: null pointer");
    DQ q;
    for (int i = 0; i < 8; ++i) q.q[i] = x[i];
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    // per-workgroup sums in stream-ordered scratch, added in a fixed order by a second launch: the same bits every run.
    // (No scratch -- allocation refused, e.g. inside a stream capture without pool support: atomics, last bits may vary.)
    double *partial = nullptr;
    if (on(opt().rigid_atomic) || hipMallocAsync(reinterpret_cast<void **>(&partial), sizeof(double) * 29 * (size_t)blocks, s) != hipSuccess) {
        (void)hipGetLastError();
        partial = nullptr;
    }
    hipLaunchKernelGGL(gn_build_rigid_kernel, dim3(blocks), dim3(256), 0, s, verts, normals, corr, valid, n, q, out44, partial);
    if (partial) {
        hipLaunchKernelGGL(gn_rigid_finish_kernel, dim3(1), dim3(256), 0, s, partial, blocks, out44);
        DFH_HIP_CHECK(hipFreeAsync(partial, s));
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_residual_data(const double *verts, const double *normals, const double *corr, const int *nbr, int n_verts,
                      int knn, const double *node_dq, const double *node_pos, const double *node_w, int n_nodes,
                      const double lw_dq[8], double *out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_verts >= 0 && n_nodes >= 1, "dfh_residual_data: bad sizes");
    DFH_REQUIRE(knn >= 1 && knn <= kKMaxS, "dfh_residual_data: knn=%d outside [1,%d]", knn, kKMaxS);
    if (n_verts == 0) return DFH_OK;
    DFH_REQUIRE(verts && normals && corr && nbr && node_dq && node_pos && node_w && lw_dq && out, "dfh_residual_data: null pointer");
    DQ q;
    for (int i = 0; i < 8; ++i) q.q[i] = lw_dq[i];
    hipLaunchKernelGGL(residual_data_kernel, dim3((n_verts + 255) / 256), dim3(256), 0, (hipStream_t)stream, verts, normals,
                       corr, nbr, n_verts, knn, node_dq, node_pos, node_w, q, out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_residual_reg(const int *node_nbr, int n_nodes, int knn, const double *node_dq, const double *node_pos,
                     const double *node_w, double rw, double *out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_nodes >= 0 && knn >= 1 && knn <= kKMaxS, "dfh_residual_reg: bad sizes");
    if (n_nodes == 0) return DFH_OK;
    DFH_REQUIRE(node_nbr && node_dq && node_pos && node_w && out, "dfh_residual_reg: null pointer");
    const int n = n_nodes * knn;
    hipLaunchKernelGGL(residual_reg_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, node_nbr, n_nodes, knn,
                       node_dq, node_pos, node_w, rw, out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_warp_points(const double *verts, const double *normals, const int *nbr, int n_verts, int knn, const double *node_dq,
                    const double *node_pos, const double *node_w, int n_nodes, const double lw_dq[8], double *out_pos,
                    double *out_nrm, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_verts >= 0, "dfh_warp_points: negative count");
    if (n_verts == 0) return DFH_OK;
    DFH_REQUIRE(verts && lw_dq && out_pos, "dfh_warp_points: null pointer");
    DFH_REQUIRE((normals == nullptr) == (out_nrm == nullptr) || normals, "dfh_warp_points: out_nrm needs normals");
    if (nbr) {
        DFH_REQUIRE(knn >= 1 && knn <= kKMaxS && n_nodes >= 1 && node_dq && node_pos && node_w, "dfh_warp_points: bad graph arguments");
    }
    DQ q;
    for (int i = 0; i < 8; ++i) q.q[i] = lw_dq[i];
    hipLaunchKernelGGL(warp_points_kernel, dim3((n_verts + 255) / 256), dim3(256), 0, (hipStream_t)stream, verts, normals, nbr,
                       n_verts, knn, node_dq, node_pos, node_w, q, out_pos, normals ? out_nrm : nullptr);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_closest_correspondences(const double *warped_pos, const double *warped_nrm, int n_verts, const double *live_verts,
                                int n_live, int knn, double tolerance, double *corr_out, double *cost_out,
                                unsigned char *keep_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_verts >= 0 && knn >= 1 && knn <= kKMaxS, "dfh_closest_correspondences: bad sizes");
    DFH_REQUIRE(n_live >= knn, "dfh_closest_correspondences: %d live vertices < knn=%d", n_live, knn);
    if (n_verts == 0) return DFH_OK;
    DFH_REQUIRE(warped_pos && warped_nrm && live_verts && corr_out && keep_out, "dfh_closest_correspondences: null pointer");
    hipLaunchKernelGGL(closest_corr_kernel, dim3((n_verts + 255) / 256), dim3(256), 0, (hipStream_t)stream, warped_pos, warped_nrm,
                       n_verts, live_verts, n_live, knn, tolerance, corr_out, cost_out, keep_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_nearest_points(const double *query, int n_query, const double *cloud, int n_cloud, int *idx_out, double *d2_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_query >= 0 && n_cloud >= 1, "dfh_nearest_points: bad sizes");
    if (n_query == 0) return DFH_OK;
    DFH_REQUIRE(query && cloud && idx_out, "dfh_nearest_points: null pointer");
    hipLaunchKernelGGL(nearest_point_kernel, dim3(n_query), dim3(256), 0, (hipStream_t)stream, query, n_query, cloud, n_cloud, idx_out, d2_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_graph_unsupported(const double *verts, int n_verts, const int *nbr, int knn, const double *node_pos, const double *node_w,
                          int n_nodes, unsigned char *flag_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_verts >= 0 && knn >= 1 && knn <= kKMaxS && n_nodes >= 1, "dfh_graph_unsupported: bad sizes");
    if (n_verts == 0) return DFH_OK;
    DFH_REQUIRE(verts && nbr && node_pos && node_w && flag_out, "dfh_graph_unsupported: null pointer");
    hipLaunchKernelGGL(graph_unsupported_kernel, dim3((n_verts + 255) / 256), dim3(256), 0, (hipStream_t)stream, verts, n_verts, nbr, knn,
                       node_pos, node_w, flag_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_dq_blend_points(const double *points, int n_points, const int *nbr, int knn, const double *node_dq, const double *node_pos,
                        const double *node_w, int n_nodes, double *dq_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_points >= 0 && knn >= 1 && knn <= kKMaxS && n_nodes >= 1, "dfh_dq_blend_points: bad sizes");
    if (n_points == 0) return DFH_OK;
    DFH_REQUIRE(points && nbr && node_dq && node_pos && node_w && dq_out, "dfh_dq_blend_points: null pointer");
    hipLaunchKernelGGL(dq_blend_points_kernel, dim3((n_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, points, n_points, nbr, knn,
                       node_dq, node_pos, node_w, dq_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_sample_knn(const double *sample_pos, int n_samples, const double *node_pos, const double *node_w, int n_nodes,
                   int knn, int *nbr_out, double *weights_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_samples >= 0 && knn >= 1 && knn <= kKMaxS && n_nodes >= knn, "dfh_sample_knn: bad sizes");
    if (n_samples == 0) return DFH_OK;
    DFH_REQUIRE(sample_pos && node_pos && node_w && nbr_out && weights_out, "dfh_sample_knn: null pointer");
    hipLaunchKernelGGL(sample_knn_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, (hipStream_t)stream, sample_pos,
                       n_samples, node_pos, node_w, n_nodes, knn, nbr_out, weights_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_permute_samples(const long *order, int n_samples, int knn, const double *pos, const double *nrm, const int *nbr,
                        const double *weights, double *pos_out, double *nrm_out, int *nbr_out, double *weights_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_samples >= 0 && knn >= 1 && knn <= kKMaxS, "dfh_permute_samples: bad sizes");
    if (n_samples == 0) return DFH_OK;
    DFH_REQUIRE(order && pos && nrm && nbr && weights && pos_out && nrm_out && nbr_out && weights_out, "dfh_permute_samples: null pointer");
    hipLaunchKernelGGL(permute_samples_kernel, dim3((n_samples + 255) / 256), dim3(256), 0, (hipStream_t)stream, order, n_samples, knn, pos,
                       nrm, nbr, weights, pos_out, nrm_out, nbr_out, weights_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

// views: a dfh_gn_pack_views table (then depth / lw_cam are unused), or nullptr for the one view (depth, lw_cam)
static int gn_associate_impl(const char *what, const double *sample_pos, const int *nbr, const double *weights, int n_samples, int knn,
                             const double *node_dq, const double lw_dq[8], const void *depth, int depth_dtype, const void *views, int n_views,
                             int H, int W, const double K[9], const double Kinv[9], const double lw_cam[12], double scale,
                             const double center[3], double half, double max_dist, double *corr_out, unsigned char *valid_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_samples >= 0 && knn >= 1 && knn <= kKMaxS, "%s: bad sizes", what);
    DFH_REQUIRE(depth_dtype == DFH_F32 || depth_dtype == DFH_F64, "%s: bad depth_dtype", what);
    DFH_REQUIRE(H >= 2 && W >= 2 && scale != 0.0, "%s: bad depth map / scale", what);
    if (n_samples == 0) return DFH_OK;
    DFH_REQUIRE(sample_pos && nbr && weights && node_dq && lw_dq && (depth || views) && K && Kinv && lw_cam && center && corr_out && valid_out,
                "%s: null pointer", what);
    AssocParams p;
    {
        const int rc = fill_assoc_params(p, lw_dq, H, W, K, Kinv, lw_cam, scale, center, half, max_dist, knn);
        if (rc != DFH_OK) return rc;
    }
    dim3 grid((n_samples + 255) / 256), block(256);
    const AssocView *vw = static_cast<const AssocView *>(views);
    if (depth_dtype == DFH_F32) {
        hipLaunchKernelGGL(associate_kernel<float>, grid, block, 0, (hipStream_t)stream, sample_pos, nbr, weights, n_samples,
                           node_dq, (const float *)depth, p, corr_out, valid_out, vw, n_views);
    } else {
        hipLaunchKernelGGL(associate_kernel<double>, grid, block, 0, (hipStream_t)stream, sample_pos, nbr, weights, n_samples,
                           node_dq, (const double *)depth, p, corr_out, valid_out, vw, n_views);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_gn_associate(const double *sample_pos, const int *nbr, const double *weights, int n_samples, int knn,
                     const double *node_dq, const double lw_dq[8], const void *depth, int depth_dtype, int H, int W,
                     const double K[9], const double Kinv[9], const double lw_cam[12], double scale,
                     const double center[3], double half, double max_dist, double *corr_out,
                     unsigned char *valid_out, void *stream) {
    return gn_associate_impl("dfh_gn_associate", sample_pos, nbr, weights, n_samples, knn, node_dq, lw_dq, depth, depth_dtype, nullptr, 0, H, W,
                             K, Kinv, lw_cam, scale, center, half, max_dist, corr_out, valid_out, stream);
}

static const double kIdentity34[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};

int dfh_gn_associate_views(const double *sample_pos, const int *nbr, const double *weights, int n_samples, int knn,
                           const double *node_dq, const double lw_dq[8], const void *views, int n_views, int depth_dtype, int H, int W,
                           const double K[9], const double Kinv[9], double scale, const double center[3], double half,
                           double max_dist, double *corr_out, unsigned char *valid_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(views && n_views >= 1 && n_views <= DFH_GN_MAX_VIEWS, "dfh_gn_associate_views: needs 1..%d packed views", DFH_GN_MAX_VIEWS);
    return gn_associate_impl("dfh_gn_associate_views", sample_pos, nbr, weights, n_samples, knn, node_dq, lw_dq, nullptr, depth_dtype, views,
                             n_views, H, W, K, Kinv, kIdentity34, scale, center, half, max_dist, corr_out, valid_out, stream);
}

size_t dfh_gn_views_bytes(int n_views) { return n_views > 0 ? (size_t)n_views * sizeof(dfh::AssocView) : 0; }

// The table travels as kernel arguments of a one-workgroup launch that writes it to device memory (a hipMemcpyAsync from
// pageable host memory is staged by the runtime and stalls the stream for tens of microseconds)
namespace dfh {
constexpr int kViewChunk = 16;
struct ViewChunk { AssocView v[kViewChunk]; };
static_assert(sizeof(ViewChunk) + 16 <= 4096, "the chunk must fit the kernel-argument segment");
__global__ __launch_bounds__(256) void upload_views_kernel(AssocView *dst, const ViewChunk c, int n) {
    typedef const unsigned long long __attribute__((address_space(4))) *KernArgWords;
    KernArgWords ka = (KernArgWords)__builtin_amdgcn_kernarg_segment_ptr() + 1;                                          // behind `dst`
    unsigned long long *out = reinterpret_cast<unsigned long long *>(dst);
    const int words = n * (int)(sizeof(AssocView) / 8);
    for (int i = threadIdx.x; i < words; i += 256) out[i] = ka[i];
    (void)c;
}
}  // namespace dfh

namespace dfh {
// {smallest, largest} valid z = -depth of every 16 x 16-pixel cell of every view (no valid pixel: {inf, 0}); block = cell
__global__ __launch_bounds__(256) void view_cells_kernel(const AssocView *__restrict__ views, int H, int W) {
    __shared__ float smin[4], smax[4];
    const AssocView &vw = views[blockIdx.y];
    const int ncx = (W + kCellPx - 1) / kCellPx;
    const int cy = blockIdx.x / ncx, cx = blockIdx.x - cy * ncx;
    const int x = cx * kCellPx + (threadIdx.x & 15), y = cy * kCellPx + (threadIdx.x >> 4);
    float z = 0.0f;
    if (x < W && y < H) z = -static_cast<const float *>(vw.depth)[(size_t)y * W + x];
    float lo = z > 0.0f ? z : __builtin_huge_valf(), hi = z > 0.0f ? z : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *out = const_cast<float *>(vw.cells) + 2 * (size_t)blockIdx.x;
        out[0] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        out[1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}
}  // namespace dfh

static size_t views_cells_offset(int n_views) { return ((size_t)n_views * sizeof(dfh::AssocView) + 255) & ~(size_t)255; }
static size_t view_cells_floats(int H, int W) {
    return 2 * (size_t)((W + dfh::kCellPx - 1) / dfh::kCellPx) * (size_t)((H + dfh::kCellPx - 1) / dfh::kCellPx);
}

size_t dfh_gn_views_bytes_cells(int n_views, int H, int W) {
    if (n_views <= 0 || H <= 0 || W <= 0) return 0;
    return views_cells_offset(n_views) + (size_t)n_views * view_cells_floats(H, W) * sizeof(float);
}

static int pack_views_impl(void *views_out, int n_views, const void *const *depth, const double *lw_cam, int cells, int H, int W, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(views_out && depth && lw_cam, "dfh_gn_pack_views: null pointer");
    DFH_REQUIRE(n_views >= 1 && n_views <= DFH_GN_MAX_VIEWS, "dfh_gn_pack_views: %d views (1..%d)", n_views, DFH_GN_MAX_VIEWS);
    static_assert(DFH_GN_MAX_VIEWS <= kViewChunk, "one upload launch");
    ViewChunk c;
    std::memset(&c, 0, sizeof c);
    for (int v = 0; v < n_views; ++v) {
        DFH_REQUIRE(depth[v], "dfh_gn_pack_views: depth map %d is null", v);
        AssocParams tmp;                                  // (for the inverse of the extrinsic's 3x3 part)
        const double zero3[3] = {0, 0, 0}, eye9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ident8[8] = {1, 0, 0, 0, 0, 0, 0, 0};
        const int rc = fill_assoc_params(tmp, ident8, 2, 2, eye9, eye9, lw_cam + 12 * v, 1.0, zero3, 0.0, 0.0, 1);
        if (rc != DFH_OK) return rc;
        for (int i = 0; i < 12; ++i) c.v[v].lw_cam[i] = lw_cam[12 * v + i];
        for (int i = 0; i < 9; ++i) c.v[v].Rinv[i] = tmp.Rinv.m[i];
        c.v[v].depth = depth[v];
        if (cells) {
            c.v[v].cells = reinterpret_cast<const float *>(static_cast<char *>(views_out) + views_cells_offset(n_views)) + (size_t)v * view_cells_floats(H, W);
            // the depth-interval test of tile_view_mask needs |R x| = |x|: R^T R = I to 1e-9 (what a camera pose is)
            const double *m = lw_cam + 12 * v;
            double worst = 0.0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    const double g = m[a] * m[b] + m[4 + a] * m[4 + b] + m[8 + a] * m[8 + b];
                    worst = std::fmax(worst, std::fabs(g - (a == b ? 1.0 : 0.0)));
                }
            c.v[v].cull_ok = worst <= 1e-9 ? 1.0 : 0.0;
        }
    }
    hipLaunchKernelGGL(upload_views_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<AssocView *>(views_out), c, n_views);
    if (cells) {
        const unsigned ncell = (unsigned)(view_cells_floats(H, W) / 2);
        hipLaunchKernelGGL(view_cells_kernel, dim3(ncell, (unsigned)n_views), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const AssocView *>(views_out), H, W);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_gn_pack_views(void *views_out, int n_views, const void *const *depth, const double *lw_cam, void *stream) {
    return pack_views_impl(views_out, n_views, depth, lw_cam, 0, 0, 0, stream);
}

int dfh_gn_pack_views_cells(void *views_out, int n_views, const void *const *depth, int H, int W, const double *lw_cam, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(H >= 2 && W >= 2, "dfh_gn_pack_views_cells: bad depth map size");
    return pack_views_impl(views_out, n_views, depth, lw_cam, 1, H, W, stream);
}

static int gn_build_impl(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                         double *corr, unsigned char *valid, int n_samples, int knn, const double *node_dq,
                         const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                         const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                         double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                         const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                         const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                         void *stream, const dfh::AssocArgs *assoc = nullptr, double *zero_ptr = nullptr, size_t zero_count = 0,
                         bool *zeroed = nullptr, const int *blk_upper = nullptr, int n_upper = 0) {
    using namespace dfh;
    if (zeroed) *zeroed = false;
    const bool planned = blk_ptr != nullptr;
    DFH_REQUIRE(!assoc || planned, "dfh_gn_build: association inside the build needs a plan");
    DFH_REQUIRE(huber_delta >= 0.0, "dfh_gn_build: negative huber_delta");
    const bool planned_reg = planned && partial_reg != nullptr;
    if (planned_reg) DFH_REQUIRE(rblk_ptr && rblk_ent && rnode_ptr && rnode_ent, "dfh_gn_build_planned: null regulariser plan array");
    DFH_REQUIRE(n_samples >= 0 && n_nodes >= 1 && n_blocks >= 1, "dfh_gn_build: bad sizes");
    DFH_REQUIRE(knn >= 1 && knn <= kKMaxS, "dfh_gn_build: knn=%d outside [1,%d]", knn, kKMaxS);
    DFH_REQUIRE(node_dq && node_pos && node_w && lw_dq && row_ptr && col && vals && rhs && cost_count, "dfh_gn_build: null pointer");
    if (planned) {
        // (a rank whose slab holds no surface has no samples, no rows and EMPTY entry lists: null pointers are fine then)
        DFH_REQUIRE(n_rows >= 0 && node_ptr && (n_rows == 0 || (blk_ent && node_ent)), "dfh_gn_build_planned: null plan array");
        DFH_REQUIRE(n_samples == 0 || (run_id && partial && n_rows > 0), "dfh_gn_build_planned: samples without rows");
    }
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = (n_samples + kTile - 1) / kTile;
    // planned build: the regulariser's pair rows ride along in the data-row launch (they only write partial_reg)
    const bool reg_in_data_launch = planned_reg && node_nbr && rw != 0.0 && n_samples > 0 && !on(opt().gn_reg_own_launch);
    double *tile_cost = planned && partial ? partial + (size_t)n_rows * gn_row_stride(knn) : nullptr;   // 2 doubles per tile, behind the rows; then one live flag per row
    if (planned) {
        // every block / rhs entry / cost is written by the gather, and every row of `partial` by the tile pass (rows
        // without a valid sample this iteration are zeroed there): nothing to clear
    } else if (rhs == vals + 36 * (size_t)n_blocks && cost_count == rhs + 6 * (size_t)n_nodes) {
        // the flat {blocks | rhs | cost,count} layout of the host solver: one memset
        DFH_HIP_CHECK(hipMemsetAsync(vals, 0, sizeof(double) * (36 * (size_t)n_blocks + 6 * (size_t)n_nodes + 2), s));
    } else {
        DFH_HIP_CHECK(hipMemsetAsync(vals, 0, sizeof(double) * 36 * (size_t)n_blocks, s));
        DFH_HIP_CHECK(hipMemsetAsync(rhs, 0, sizeof(double) * 6 * (size_t)n_nodes, s));
        DFH_HIP_CHECK(hipMemsetAsync(cost_count, 0, sizeof(double) * 2, s));
    }
    if (n_samples > 0) {
        DFH_REQUIRE(sample_pos && sample_nrm && nbr && weights && corr && valid, "dfh_gn_build: null sample pointer");
        BuildParams p;
        for (int i = 0; i < 8; ++i) p.lw.q[i] = lw_dq[i];
        p.S = n_samples; p.k = knn; p.N = n_nodes; p.huber = huber_delta;
        RegTail rt = {};
        rt.n_tiles = n_tiles;
        if (reg_in_data_launch) {
            rt.node_nbr = node_nbr; rt.node_pos = node_pos; rt.node_w = node_w; rt.partial_reg = partial_reg;
            rt.rw = rw; rt.N = n_nodes; rt.k = knn;
        }
        unsigned n_wg = (unsigned)(n_tiles + (reg_in_data_launch ? (n_nodes * knn + kTileWaves - 1) / kTileWaves : 0));
        if (planned && zero_ptr && zero_count > 0 && (zero_count + kZeroPerWg - 1) / kZeroPerWg < (1u << 20)) {
            rt.zero_ptr = zero_ptr; rt.zero_count = zero_count; rt.first_zero_wg = (int)n_wg;
            n_wg += (unsigned)((zero_count + kZeroPerWg - 1) / kZeroPerWg);
            if (zeroed) *zeroed = true;
        }
        dim3 grid(n_wg), block(kTile);
        const AssocArgs aa = assoc ? *assoc : AssocArgs{};
#define DFH_BUILD(KK)                                                                                               \
    case KK:                                                                                                        \
        if (assoc)                                                                                                  \
            hipLaunchKernelGGL((gn_build_data_kernel<KK, true, true>), grid, block, 0, s, sample_pos, sample_nrm, nbr, weights, corr, \
                               valid, node_dq, p, row_ptr, col, vals, rhs, cost_count, run_id, partial, tile_cost, rt, aa); \
        else if (planned)                                                                                           \
            hipLaunchKernelGGL((gn_build_data_kernel<KK, true, false>), grid, block, 0, s, sample_pos, sample_nrm, nbr, weights, corr, \
                               valid, node_dq, p, row_ptr, col, vals, rhs, cost_count, run_id, partial, tile_cost, rt, aa); \
        else                                                                                                        \
            hipLaunchKernelGGL((gn_build_data_kernel<KK, false, false>), grid, block, 0, s, sample_pos, sample_nrm, nbr, weights, corr, \
                               valid, node_dq, p, row_ptr, col, vals, rhs, cost_count, run_id, partial, tile_cost, rt, aa); \
        break
        switch (knn) {
            DFH_BUILD(1); DFH_BUILD(2); DFH_BUILD(3); DFH_BUILD(4); DFH_BUILD(5); DFH_BUILD(6); DFH_BUILD(7); DFH_BUILD(8);
        }
#undef DFH_BUILD
        DFH_HIP_CHECK(hipGetLastError());
    }
    const int dbg_part = opt().dbg_gather_part > 0 ? (int)opt().dbg_gather_part : 0;
    // the regulariser's lists ride along in the data rows' gather when its rows were built in the data-row launch
    RegLists rl = {};
    const bool reg_in_gather = reg_in_data_launch && !on(opt().gn_reg_own_gather);
    if (reg_in_gather) {
        rl.partial = partial_reg; rl.blk_ptr = rblk_ptr; rl.blk_ent = rblk_ent; rl.node_ptr = rnode_ptr; rl.node_ent = rnode_ent;
        rl.n_rows = n_nodes * knn;
    }
    if (planned) {
        const int2 *upper = (blk_upper && n_upper > 0 && !on(opt().gn_gather_full)) ? reinterpret_cast<const int2 *>(blk_upper) : nullptr;
        const int n_walk = upper ? n_upper : n_blocks;
        dim3 grid((unsigned)((n_walk + 3) / 4 + (n_nodes + 3) / 4 + 1)), block(256);
#define DFH_GATHER(KK)                                                                                              \
    case KK:                                                                                                        \
        hipLaunchKernelGGL(gn_gather_kernel<KK>, grid, block, 0, s, partial, tile_cost + 2 * (size_t)n_tiles, n_rows, blk_ptr, blk_ent, n_blocks, node_ptr,  \
                           node_ent, n_nodes, vals, rhs, cost_count, tile_cost, n_tiles, 2, false, dbg_part, rl, upper, n_upper);   \
        break
        switch (knn) {
            DFH_GATHER(1); DFH_GATHER(2); DFH_GATHER(3); DFH_GATHER(4); DFH_GATHER(5); DFH_GATHER(6); DFH_GATHER(7); DFH_GATHER(8);
        }
#undef DFH_GATHER
        DFH_HIP_CHECK(hipGetLastError());
    }
    if (node_nbr && rw != 0.0) {
        const int n = n_nodes * knn;
        if (!reg_in_data_launch)
            hipLaunchKernelGGL(gn_build_reg_kernel, dim3((n + 3) / 4), dim3(256), 0, s, node_nbr, n_nodes, knn, node_dq, node_pos,
                               node_w, rw, row_ptr, col, vals, rhs, cost_count, planned_reg ? partial_reg : nullptr);
        if (planned_reg && !reg_in_gather) {
            dim3 grid((unsigned)((n_blocks + 3) / 4 + (n_nodes + 3) / 4 + 1)), block(256);
            hipLaunchKernelGGL(gn_gather_kernel<2>, grid, block, 0, s, partial_reg, (const double *)nullptr, n, rblk_ptr, rblk_ent, n_blocks, rnode_ptr,
                               rnode_ent, n_nodes, vals, rhs, cost_count, partial_reg + gn_row_gram(2) + 12, n, gn_row_stride(2), true, dbg_part,
                               RegLists{});
        }
        DFH_HIP_CHECK(hipGetLastError());
    }
    return DFH_OK;
}

int dfh_gn_build(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                 const double *corr, const unsigned char *valid, int n_samples, int knn, const double *node_dq,
                 const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                 const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                 double *rhs, double *cost_count, void *stream) {
    return gn_build_impl(sample_pos, sample_nrm, nbr, weights, const_cast<double *>(corr), const_cast<unsigned char *>(valid), n_samples,
                         knn, node_dq, node_pos, node_w, node_nbr,
                         n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count, nullptr, 0, nullptr, nullptr, nullptr,
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, stream);
}

size_t dfh_gn_partial_doubles(int knn) {
    if (knn < 1 || knn > dfh::kKMaxS) return 0;
    return (size_t)dfh::gn_row_stride(knn);
}

int dfh_gn_build_planned(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                         const double *corr, const unsigned char *valid, int n_samples, int knn, const double *node_dq,
                         const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                         const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                         double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                         const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                         const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                         void *stream) {
    using namespace dfh;
    DFH_REQUIRE(blk_ptr, "dfh_gn_build_planned: null blk_ptr");
    return gn_build_impl(sample_pos, sample_nrm, nbr, weights, const_cast<double *>(corr), const_cast<unsigned char *>(valid), n_samples,
                         knn, node_dq, node_pos, node_w, node_nbr,
                         n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count, run_id, n_rows, partial, blk_ptr,
                         blk_ent, node_ptr, node_ent, partial_reg, rblk_ptr, rblk_ent, rnode_ptr, rnode_ent, huber_delta, stream);
}

static int fill_assoc_params(dfh::AssocParams &p, const double lw_dq[8], int H, int W, const double K[9], const double Kinv[9],
                             const double lw_cam[12], double scale, const double center[3], double half, double max_dist, int knn) {
    using namespace dfh;
    for (int i = 0; i < 9; ++i) { p.K.m[i] = K[i]; p.Kinv.m[i] = Kinv[i]; }
    for (int i = 0; i < 12; ++i) p.lw_cam.m[i] = lw_cam[i];
    for (int i = 0; i < 8; ++i) p.lw.q[i] = lw_dq[i];
    {   // inverse of the 3x3 part (adjugate)
        const double *m = lw_cam;
        const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        DFH_REQUIRE(det != 0.0, "dfh_gn_associate: singular extrinsic");
        const double id = 1.0 / det;
        p.Rinv.m[0] = (e * i - f * h) * id; p.Rinv.m[1] = (c * h - b * i) * id; p.Rinv.m[2] = (b * f - c * e) * id;
        p.Rinv.m[3] = (f * g - d * i) * id; p.Rinv.m[4] = (a * i - c * g) * id; p.Rinv.m[5] = (c * d - a * f) * id;
        p.Rinv.m[6] = (d * h - e * g) * id; p.Rinv.m[7] = (b * g - a * h) * id; p.Rinv.m[8] = (a * e - b * d) * id;
    }
    p.scale = scale; p.inv_scale = 1.0 / scale; p.cx = center[0]; p.cy = center[1]; p.cz = center[2]; p.half = half; p.max_dist = max_dist;
    p.H = H; p.W = W; p.k = knn;
    return DFH_OK;
}

static int gn_build_planned_assoc_impl(const char *what, const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                               double *corr_out, unsigned char *valid_out, int n_samples, int knn, const double *node_dq,
                               const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                               const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                               double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                               const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                               const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                               const float *depth, const void *views, int n_views, int H, int W, const double K[9], const double Kinv[9],
                               const double lw_cam[12], double scale, const double center[3], double half, double max_dist, void *stream,
                               const int *blk_upper = nullptr, int n_upper = 0) {
    using namespace dfh;
    DFH_REQUIRE(blk_ptr, "%s: null blk_ptr", what);
    DFH_REQUIRE((depth || views) && K && Kinv && lw_cam && center && lw_dq && corr_out && valid_out, "%s: null pointer", what);
    DFH_REQUIRE(H >= 2 && W >= 2 && scale != 0.0, "%s: bad depth map / scale", what);
    AssocArgs aa;
    const int rc = fill_assoc_params(aa.ap, lw_dq, H, W, K, Kinv, lw_cam, scale, center, half, max_dist, knn);
    if (rc != DFH_OK) return rc;
    aa.depth = depth;
    aa.views = static_cast<const AssocView *>(views);
    aa.n_views = n_views;
    // per-tile view culling costs a tile one barrier and one memory round trip (+5 % on the 3-view frame, where the views all
    // face the object and nothing is dropped): taken from four views up (the 8-view orbit: -9 % of the solve stage)
    aa.cull = views && n_views >= 4 && !on(opt().gn_no_view_cull) ? 1 : 0;
    return gn_build_impl(sample_pos, sample_nrm, nbr, weights, corr_out, valid_out, n_samples, knn, node_dq, node_pos, node_w, node_nbr,
                         n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count, run_id, n_rows, partial, blk_ptr,
                         blk_ent, node_ptr, node_ent, partial_reg, rblk_ptr, rblk_ent, rnode_ptr, rnode_ent, huber_delta, stream, &aa,
                         nullptr, 0, nullptr, blk_upper, n_upper);
}

int dfh_gn_build_planned_assoc(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                               double *corr_out, unsigned char *valid_out, int n_samples, int knn, const double *node_dq,
                               const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                               const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                               double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                               const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                               const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                               const float *depth, int H, int W, const double K[9], const double Kinv[9], const double lw_cam[12],
                               double scale, const double center[3], double half, double max_dist, void *stream) {
    return gn_build_planned_assoc_impl("dfh_gn_build_planned_assoc", sample_pos, sample_nrm, nbr, weights, corr_out, valid_out, n_samples, knn,
                                       node_dq, node_pos, node_w, node_nbr, n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count,
                                       run_id, n_rows, partial, blk_ptr, blk_ent, node_ptr, node_ent, partial_reg, rblk_ptr, rblk_ent,
                                       rnode_ptr, rnode_ent, huber_delta, depth, nullptr, 0, H, W, K, Kinv, lw_cam, scale, center, half,
                                       max_dist, stream);
}

int dfh_gn_build_planned_assoc_views(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                               double *corr_out, unsigned char *valid_out, int n_samples, int knn, const double *node_dq,
                               const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                               const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                               double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                               const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                               const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                               const void *views, int n_views, int H, int W, const double K[9], const double Kinv[9],
                               double scale, const double center[3], double half, double max_dist, const int *blk_upper, int n_upper,
                               void *stream) {
    using namespace dfh;
    DFH_REQUIRE(views && n_views >= 1 && n_views <= DFH_GN_MAX_VIEWS, "dfh_gn_build_planned_assoc_views: needs 1..%d packed views", DFH_GN_MAX_VIEWS);
    DFH_REQUIRE(n_upper >= 0 && n_upper <= n_blocks && (n_upper == 0 || blk_upper), "dfh_gn_build_planned_assoc_views: bad upper-block list");
    return gn_build_planned_assoc_impl("dfh_gn_build_planned_assoc_views", sample_pos, sample_nrm, nbr, weights, corr_out, valid_out, n_samples,
                                       knn, node_dq, node_pos, node_w, node_nbr, n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs,
                                       cost_count, run_id, n_rows, partial, blk_ptr, blk_ent, node_ptr, node_ent, partial_reg, rblk_ptr,
                                       rblk_ent, rnode_ptr, rnode_ent, huber_delta, nullptr, views, n_views, H, W, K, Kinv, kIdentity34, scale,
                                       center, half, max_dist, stream, blk_upper, n_upper);
}

size_t dfh_pcg_workspace_bytes(int n_nodes, int iters) {
    if (n_nodes <= 0 || iters < 0) return 0;
    // Minv (36N) + r,pA,Ap,z,pB (5*6N) + scalars (3 per iteration + 6) + per-workgroup partial sums + the persistent
    // kernel's ring of published vectors (4 x 3 x 6N)
    return sizeof(double) * ((size_t)36 * n_nodes + (size_t)30 * n_nodes + 3 * ((size_t)iters + 2) +
                             2 * ((size_t)iters + 1) * (((size_t)n_nodes + 3) / 4) + (size_t)72 * n_nodes);
}

// ---- persistent-PCG bookkeeping ---------------------------------------------------------------------------------
// g_pcg_mode: 0 = auto (persistent kernel when co-residency holds, see pcg_solve_impl), 2 = always the two-launches-per-
// iteration path.  g_abort_count[dev]: device counter the persistent kernel bumps when a barrier times out.
// g_abort_host[dev]: a word of pinned host memory the kernel sets with the counter, so that the host can ask "anything
// timed out?" without a device call (dfh_pcg_status_peek).
namespace dfh { int g_pcg_mode = 0; unsigned long long *g_abort_count[64] = {nullptr}; unsigned *g_abort_host[64] = {nullptr}; }
using dfh::g_abort_count;
using dfh::g_abort_host;

static int pcg_abort_counter(unsigned long long **out) {
    int dev = 0;
    DFH_HIP_CHECK(hipGetDevice(&dev));
    DFH_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    if (!g_abort_count[dev]) {
        unsigned long long *p = nullptr;
        DFH_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p), sizeof(unsigned long long)));
        DFH_HIP_CHECK(hipMemset(p, 0, sizeof(unsigned long long)));
        unsigned *h = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&h), sizeof(unsigned), hipHostMallocMapped) == hipSuccess && h) {
            *h = 0u;
            g_abort_host[dev] = h;                          // (without it the peek always takes the synchronising path)
        } else {
            (void)hipGetLastError();
        }
        g_abort_count[dev] = p;
    }
    *out = g_abort_count[dev];
    return DFH_OK;
}

int dfh_pcg_set_mode(int mode) {
    DFH_REQUIRE(mode == 0 || mode == 2, "dfh_pcg_set_mode: mode %d (0 = auto, 2 = multi-launch)", mode);
    dfh::g_pcg_mode = mode;
    return DFH_OK;
}

int dfh_pcg_status(void *stream, long *aborted_solves_out) {
    using namespace dfh;
    DFH_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    int dev = 0;
    DFH_HIP_CHECK(hipGetDevice(&dev));
    unsigned long long n = 0;
    if (dev >= 0 && dev < 64 && g_abort_count[dev]) {
        DFH_HIP_CHECK(hipMemcpy(&n, g_abort_count[dev], sizeof(n), hipMemcpyDeviceToHost));
        if (n) DFH_HIP_CHECK(hipMemset(g_abort_count[dev], 0, sizeof(n)));
        if (g_abort_host[dev]) *g_abort_host[dev] = 0u;
    }
    if (aborted_solves_out) *aborted_solves_out = (long)n;
    if (n)
        return fail(DFH_E_TIMEOUT, "persistent PCG: %llu solve(s) timed out in a grid barrier (workgroups not co-resident?); x = NaN; "
                                   "a timed-out solve leaves node_dq as it was before that solve (the twist update is all or nothing)", n);
    return DFH_OK;
}

int dfh_pcg_status_peek(void *stream, long *aborted_solves_out) {
    using namespace dfh;
    int dev = 0;
    DFH_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && (!g_abort_count[dev] || (g_abort_host[dev] && *static_cast<volatile unsigned *>(g_abort_host[dev]) == 0u))) {
        if (aborted_solves_out) *aborted_solves_out = 0;    // no persistent solve yet, or none that has completed timed out
        return DFH_OK;
    }
    return dfh_pcg_status(stream, aborted_solves_out);
}

// the part of the workspace a solve expects all-zero at its start: the multi-launch path's first direction and its scalars, the
// persistent kernel's scalars, reduction slots and hand-off ring (zero bits = "not yet published")
static void pcg_zero_range(void *workspace, int n_nodes, int iters, double **begin, size_t *count) {
    const size_t N6 = 6 * (size_t)n_nodes;
    const size_t n_scal = 3 * ((size_t)iters + 2) + 2 * ((size_t)iters + 1) * (((size_t)n_nodes + 3) / 4);
    *begin = static_cast<double *>(workspace) + 36 * (size_t)n_nodes + 4 * N6;         // = pA
    *count = N6 + n_scal + 12 * N6;
}

// Launch shape of a solve with n_nodes rows on the current device, and whether it takes the persistent kernel.
static int pcg_shape(int n_nodes, int *dev_out, int *wpb_out, int *nblk_out, bool *persistent_out) {
    using namespace dfh;
    int dev = 0;
    DFH_HIP_CHECK(hipGetDevice(&dev));
    DeviceInfo &di = device_info(dev);              // per DEVICE: a process may drive several
    if (di.n_cu == 0) DFH_HIP_CHECK(hipDeviceGetAttribute(&di.n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int n_cu = di.n_cu;
    // Workgroups of 8 waves (one row each) as long as they fit one per CU, of 16 beyond.  Measured at 2 048 rows (tools/kbench_pcg.py):
    // 256 workgroups x 8 waves 55 us per 10-iteration solve (slope 4.3 us, prologue 14.5), 128 x 16 waves 90 us (6.4 / 27.3).
    int wpb = (n_nodes + 7) / 8 <= n_cu ? 8 : 16;
    { const long v = opt().pcg_wpb; if (v == 4 || v == 8 || v == 16) wpb = (int)v; }
    const int nblk = (n_nodes + wpb - 1) / wpb;
    // Persistent path only when its grid barrier cannot starve: (1) the occupancy query says a workgroup of this size
    // fits on a CU, (2) the grid has at most one workgroup per CU (other kernels of this process may hold CUs for a while: they
    // end, the waiting workgroups then start; what must NOT run beside it is a second persistent solve that also wants most of
    // the chip -- two of them could wait for each other until the spin bound makes both leave and report DFH_E_TIMEOUT),
    // (3) the caller has not declared co-residency unsafe (dfh_pcg_set_mode(2): several processes time-sharing one GPU),
    // (4) the abort counter exists (it cannot be allocated while the stream is being captured: pcg_solve_impl).  Otherwise:
    // two launches per iteration, no spinning.
    bool persistent = nblk <= n_cu && nblk <= kMaxPcgBlocks && dfh::g_pcg_mode != 2 && !on(opt().pcg_multilaunch);
    if (persistent) {
        int &occ = wpb <= 8 ? di.pcg_occ512 : di.pcg_occ1024;
        if (occ < 0) {
            int nb = 0;
            const hipError_t e = wpb <= 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pcg_cg1_kernel<512>, 64 * 8, 0)
                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pcg_cg1_kernel<1024>, 64 * 16, 0);
            occ = e == hipSuccess ? nb : 0;
        }
        persistent = occ >= 1;
    }
    *dev_out = dev; *wpb_out = wpb; *nblk_out = nblk; *persistent_out = persistent;
    return DFH_OK;
}

int dfh_pcg_path(int n_nodes) {
    DFH_REQUIRE(n_nodes >= 1, "dfh_pcg_path: bad node count");
    int dev = 0, wpb = 0, nblk = 0;
    bool persistent = false;
    const int rc = pcg_shape(n_nodes, &dev, &wpb, &nblk, &persistent);
    if (rc != DFH_OK) return rc;
    return persistent ? 1 : 2;
}

static int pcg_solve_impl(const int *row_ptr, const int *col, double *vals, const double *rhs, int n_nodes, int iters,
                          double lm_abs, double lm_rel, double *x_out, void *workspace, size_t workspace_bytes, double *update_dq,
                          double update_step, void *stream, bool precleared = false) {
    using namespace dfh;
    DFH_REQUIRE(n_nodes >= 1 && iters >= 1, "dfh_pcg_solve: bad sizes");
    DFH_REQUIRE(row_ptr && col && vals && rhs && x_out && workspace, "dfh_pcg_solve: null pointer");
    DFH_REQUIRE(workspace_bytes >= dfh_pcg_workspace_bytes(n_nodes, iters), "dfh_pcg_solve: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double *ws = static_cast<double *>(workspace);
    const size_t N6 = 6 * (size_t)n_nodes;
    double *Minv = ws; ws += 36 * (size_t)n_nodes;
    double *r = ws; ws += N6;
    double *Ap = ws; ws += N6;
    double *z = ws; ws += N6;
    double *pB = ws; ws += N6;
    double *pA = ws; ws += N6;                    // pA and the scalars are adjacent: one memset zeroes both
    double *scal = ws;                            // (beta = 0 in iteration 0 must not meet NaN garbage in pA)
    const size_t n_scal = 3 * ((size_t)iters + 2) + 2 * ((size_t)iters + 1) * (((size_t)n_nodes + 3) / 4);
    double *ring = scal + n_scal;                 // persistent kernel only; zero bits = "not yet published"
    PcgParams p{n_nodes, lm_abs, lm_rel};
    dim3 grid((n_nodes + 255) / 256), block(256);
    // One persistent launch when every row can have its own co-resident wave (pcg_shape: the decision, also behind dfh_pcg_path)
    int dev = 0, wpb = 8, nblk = 1;
    bool persistent = false;
    { const int rc = pcg_shape(n_nodes, &dev, &wpb, &nblk, &persistent); if (rc != DFH_OK) return rc; }
    // experiment (option pcg_one_xcd = the stride, 8 on MI355X): every working workgroup on one die, 16-wave workgroups so that
    // up to 1 024 rows fit its 32 CUs two per CU.  Measured (profiles/r4_pcg_one_xcd.txt) -- not the default.
    int die_stride = 1;
    if (persistent && opt().pcg_one_xcd > 1 && n_nodes <= 1024) {
        die_stride = (int)opt().pcg_one_xcd;
        wpb = 16;
        nblk = (n_nodes + wpb - 1) / wpb;
    }
    unsigned long long *abort_count = nullptr;
    if (persistent) {
        if (dev >= 0 && dev < 64 && g_abort_count[dev]) {
            abort_count = g_abort_count[dev];
        } else {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
            if (cs == hipStreamCaptureStatusNone) {
                const int rc = pcg_abort_counter(&abort_count);
                if (rc != DFH_OK) return rc;
            } else {
                persistent = false;
            }
        }
    }
    unsigned *abort_host = nullptr;
    if (persistent) {
        if (dev >= 0 && dev < 64 && g_abort_host[dev] &&
            hipHostGetDevicePointer(reinterpret_cast<void **>(&abort_host), g_abort_host[dev], 0) != hipSuccess) {
            (void)hipGetLastError();
            abort_host = nullptr;
        }
        if (!precleared) DFH_HIP_CHECK(hipMemsetAsync(scal, 0, sizeof(double) * (n_scal + 12 * N6), s));
        unsigned spin_limit = kSpinLimit;
        if (opt().pcg_spin_limit >= 0) spin_limit = (unsigned)opt().pcg_spin_limit;
        unsigned *flag = reinterpret_cast<unsigned *>(scal + 3 * ((size_t)iters + 1));    // spare scalar: abort flag
        double *part = scal + 3 * ((size_t)iters + 2);                                    // (2 per iteration + 1) reductions x nblk slots
        if (wpb <= 8)
            hipLaunchKernelGGL(pcg_cg1_kernel<512>, dim3(nblk * die_stride), dim3(64 * wpb), 0, s, row_ptr, col, vals, rhs, p, iters, x_out, ring, part,
                               flag, spin_limit, abort_count, abort_host, update_dq, update_step, die_stride);
        else
            hipLaunchKernelGGL(pcg_cg1_kernel<1024>, dim3(nblk * die_stride), dim3(64 * wpb), 0, s, row_ptr, col, vals, rhs, p, iters, x_out, ring, part,
                               flag, spin_limit, abort_count, abort_host, update_dq, update_step, die_stride);
        DFH_HIP_CHECK(hipGetLastError());
        return DFH_OK;
    }
    if (!precleared) DFH_HIP_CHECK(hipMemsetAsync(pA, 0, sizeof(double) * (N6 + n_scal), s));
    // dot products: per-workgroup partials in the slots the persistent kernel uses for its reductions (2 (iters + 1) slots of
    // ceil(N / 4) doubles): slot 2 it = p.Ap of iteration it, 2 it + 1 = r.z after it, slot 2 iters = r.z of the init
    const size_t nq = ((size_t)n_nodes + 3) / 4;
    double *part = scal + 3 * ((size_t)iters + 2);
    const int n_init = (int)grid.x, n_spmv = (n_nodes + 3) / 4, n_upd = (n_nodes + 39) / 40;
    hipLaunchKernelGGL(pcg_init_kernel, grid, block, 0, s, row_ptr, col, vals, rhs, p, Minv, x_out, r, z, part + 2 * (size_t)iters * nq);
    double *p_prev = pA, *p_cur = pB;
    for (int it = 0; it < iters; ++it) {
        double *sc = scal + 3 * ((size_t)it + 1);
        // iteration 0: the scalars in front of sc are zero (cleared above): rz_prev = 0 gives beta = 0
        const double *rz_part = it == 0 ? part + 2 * (size_t)iters * nq : part + (2 * (size_t)it - 1) * nq;
        hipLaunchKernelGGL(pcg_spmv_kernel, dim3(n_spmv), block, 0, s, row_ptr, col, vals, n_nodes, z, p_prev, p_cur,
                           Ap, sc - 3, sc, rz_part, it == 0 ? n_init : n_upd, part + 2 * (size_t)it * nq);
        hipLaunchKernelGGL(pcg_update_xr_kernel, dim3(n_upd), block, 0, s, n_nodes, Minv, x_out, r, p_cur, Ap, z, sc,
                           part + 2 * (size_t)it * nq, n_spmv, part + (2 * (size_t)it + 1) * nq);
        double *t = p_prev; p_prev = p_cur; p_cur = t;
    }
    if (update_dq)
        hipLaunchKernelGGL(apply_twist_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, update_dq, x_out, n_nodes, update_step);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

#ifdef DFH_BUILD_TRACE
int dfh_debug_build_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dfh::g_build_trace), sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? 0 : -1;
}
#endif

#ifdef DFH_GATHER_TRACE
int dfh_debug_gather_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dfh::g_gather_trace), sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? 0 : -1;
}
#endif

#ifdef DFH_PCG_TRACE
int dfh_debug_pcg_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(dfh::g_pcg_trace), sizeof(unsigned long long) * 64 * 16 * 12) == hipSuccess ? 0 : -1;
}
#endif

int dfh_pcg_solve(const int *row_ptr, const int *col, double *vals, const double *rhs, int n_nodes, int iters,
                  double lm_abs, double lm_rel, double *x_out, void *workspace, size_t workspace_bytes, void *stream) {
    return pcg_solve_impl(row_ptr, col, vals, rhs, n_nodes, iters, lm_abs, lm_rel, x_out, workspace, workspace_bytes, nullptr, 0.0, stream);
}

int dfh_pcg_solve_update(const int *row_ptr, const int *col, double *vals, const double *rhs, int n_nodes, int iters,
                         double lm_abs, double lm_rel, double *x_out, void *workspace, size_t workspace_bytes, double *node_dq,
                         double step, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(node_dq, "dfh_pcg_solve_update: null node_dq");
    return pcg_solve_impl(row_ptr, col, vals, rhs, n_nodes, iters, lm_abs, lm_rel, x_out, workspace, workspace_bytes, node_dq, step, stream);
}

static int gn_iteration_impl(const char *what, const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                     double *corr_out, unsigned char *valid_out, int n_samples, int knn, double *node_dq,
                     const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                     const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                     double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                     const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                     const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                     const float *depth, const void *views, int n_views, int H, int W, const double K[9], const double Kinv[9],
                     const double lw_cam[12], double scale, const double center[3], double half, double max_dist,
                     int pcg_iters, double lm_abs, double lm_rel, double *x_out, void *pcg_workspace, size_t pcg_workspace_bytes,
                     double step, void *stream, const int *blk_upper = nullptr, int n_upper = 0) {
    using namespace dfh;
    DFH_REQUIRE(blk_ptr, "%s: null blk_ptr", what);
    DFH_REQUIRE((depth || views) && K && Kinv && lw_cam && center && lw_dq && corr_out && valid_out && node_dq, "%s: null pointer", what);
    DFH_REQUIRE(H >= 2 && W >= 2 && scale != 0.0, "%s: bad depth map / scale", what);
    DFH_REQUIRE(n_nodes >= 1 && pcg_iters >= 1 && x_out && pcg_workspace, "%s: bad solve arguments", what);
    DFH_REQUIRE(pcg_workspace_bytes >= dfh_pcg_workspace_bytes(n_nodes, pcg_iters), "%s: solve workspace too small", what);
    AssocArgs aa;
    int rc = fill_assoc_params(aa.ap, lw_dq, H, W, K, Kinv, lw_cam, scale, center, half, max_dist, knn);
    if (rc != DFH_OK) return rc;
    aa.depth = depth;
    aa.views = static_cast<const AssocView *>(views);
    aa.n_views = n_views;
    // per-tile view culling costs a tile one barrier and one memory round trip (+5 % on the 3-view frame, where the views all
    // face the object and nothing is dropped): taken from four views up (the 8-view orbit: -9 % of the solve stage)
    aa.cull = views && n_views >= 4 && !on(opt().gn_no_view_cull) ? 1 : 0;
    double *zbegin = nullptr;
    size_t zcount = 0;
    pcg_zero_range(pcg_workspace, n_nodes, pcg_iters, &zbegin, &zcount);
    bool zeroed = false;
    rc = gn_build_impl(sample_pos, sample_nrm, nbr, weights, corr_out, valid_out, n_samples, knn, node_dq, node_pos, node_w, node_nbr,
                       n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count, run_id, n_rows, partial, blk_ptr,
                       blk_ent, node_ptr, node_ent, partial_reg, rblk_ptr, rblk_ent, rnode_ptr, rnode_ent, huber_delta, stream, &aa,
                       on(opt().gn_iter_own_clear) ? nullptr : zbegin, zcount, &zeroed, blk_upper, n_upper);
    if (rc != DFH_OK) return rc;
    return pcg_solve_impl(row_ptr, col, vals, rhs, n_nodes, pcg_iters, lm_abs, lm_rel, x_out, pcg_workspace, pcg_workspace_bytes, node_dq,
                          step, stream, zeroed);
}

int dfh_gn_iteration(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                     double *corr_out, unsigned char *valid_out, int n_samples, int knn, double *node_dq,
                     const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                     const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                     double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                     const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                     const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                     const float *depth, int H, int W, const double K[9], const double Kinv[9], const double lw_cam[12],
                     double scale, const double center[3], double half, double max_dist,
                     int pcg_iters, double lm_abs, double lm_rel, double *x_out, void *pcg_workspace, size_t pcg_workspace_bytes,
                     double step, void *stream) {
    return gn_iteration_impl("dfh_gn_iteration", sample_pos, sample_nrm, nbr, weights, corr_out, valid_out, n_samples, knn, node_dq, node_pos,
                             node_w, node_nbr, n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count, run_id, n_rows, partial, blk_ptr,
                             blk_ent, node_ptr, node_ent, partial_reg, rblk_ptr, rblk_ent, rnode_ptr, rnode_ent, huber_delta, depth, nullptr, 0,
                             H, W, K, Kinv, lw_cam, scale, center, half, max_dist, pcg_iters, lm_abs, lm_rel, x_out, pcg_workspace,
                             pcg_workspace_bytes, step, stream);
}

int dfh_gn_iteration_views(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                     double *corr_out, unsigned char *valid_out, int n_samples, int knn, double *node_dq,
                     const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                     const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                     double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                     const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                     const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                     const void *views, int n_views, int H, int W, const double K[9], const double Kinv[9],
                     double scale, const double center[3], double half, double max_dist,
                     int pcg_iters, double lm_abs, double lm_rel, double *x_out, void *pcg_workspace, size_t pcg_workspace_bytes,
                     double step, int n_iters, const int *blk_upper, int n_upper, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(views && n_views >= 1 && n_views <= DFH_GN_MAX_VIEWS, "dfh_gn_iteration_views: needs 1..%d packed views", DFH_GN_MAX_VIEWS);
    DFH_REQUIRE(n_upper >= 0 && n_upper <= n_blocks && (n_upper == 0 || blk_upper), "dfh_gn_iteration_views: bad upper-block list");
    DFH_REQUIRE(n_iters >= 0 && n_iters <= 1000, "dfh_gn_iteration_views: %d iterations", n_iters);
    // the frame's iterations are queued back to back from here: nothing between them depends on the host
    for (int it = 0; it < n_iters; ++it) {
        const int rc = gn_iteration_impl("dfh_gn_iteration_views", sample_pos, sample_nrm, nbr, weights, corr_out, valid_out, n_samples, knn, node_dq,
                                         node_pos, node_w, node_nbr, n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count, run_id, n_rows,
                                         partial, blk_ptr, blk_ent, node_ptr, node_ent, partial_reg, rblk_ptr, rblk_ent, rnode_ptr, rnode_ent,
                                         huber_delta, nullptr, views, n_views, H, W, K, Kinv, kIdentity34, scale, center, half, max_dist,
                                         pcg_iters, lm_abs, lm_rel, x_out, pcg_workspace, pcg_workspace_bytes, step, stream, blk_upper, n_upper);
        if (rc != DFH_OK) return rc;
    }
    return DFH_OK;
}

// A frame's whole solve behind one call (round 4): n_global rigid-mode steps (build + dfh_gn_global_step each), then n_iters
// node iterations (dfh_gn_iteration_views).  The same launches in the same order as the separate calls: the same bits; what it
// saves is the host's way through the binding, twice per rigid-mode step.
int dfh_gn_frame_solve_views(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                     double *corr_out, unsigned char *valid_out, int n_samples, int knn, double *node_dq,
                     const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                     const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                     double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                     const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                     const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                     const void *views, int n_views, int H, int W, const double K[9], const double Kinv[9],
                     double scale, const double center[3], double half, double max_dist,
                     int pcg_iters, double lm_abs, double lm_rel, double *x_out, void *pcg_workspace, size_t pcg_workspace_bytes,
                     double step, int n_iters, const int *blk_upper, int n_upper,
                     int n_global, double global_lm, double *global_xi_out, void *global_scratch, size_t global_scratch_bytes, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(views && n_views >= 1 && n_views <= DFH_GN_MAX_VIEWS, "dfh_gn_frame_solve_views: needs 1..%d packed views", DFH_GN_MAX_VIEWS);
    DFH_REQUIRE(n_global >= 0 && n_global <= 100, "dfh_gn_frame_solve_views: %d rigid-mode steps", n_global);
    if (n_global > 0) {
        DFH_REQUIRE(blk_ptr && K && Kinv && center && lw_dq && corr_out && valid_out && node_dq, "dfh_gn_frame_solve_views: null pointer");
        DFH_REQUIRE(H >= 2 && W >= 2 && scale != 0.0, "dfh_gn_frame_solve_views: bad depth map / scale");
        AssocArgs aa;
        int rc = fill_assoc_params(aa.ap, lw_dq, H, W, K, Kinv, kIdentity34, scale, center, half, max_dist, knn);
        if (rc != DFH_OK) return rc;
        aa.depth = nullptr;
        aa.views = static_cast<const AssocView *>(views);
        aa.n_views = n_views;
        aa.cull = n_views >= 4 && !on(opt().gn_no_view_cull) ? 1 : 0;
        for (int g = 0; g < n_global; ++g) {
            rc = gn_build_impl(sample_pos, sample_nrm, nbr, weights, corr_out, valid_out, n_samples, knn, node_dq, node_pos, node_w, node_nbr,
                               n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count, run_id, n_rows, partial, blk_ptr,
                               blk_ent, node_ptr, node_ent, partial_reg, rblk_ptr, rblk_ent, rnode_ptr, rnode_ent, huber_delta, stream, &aa,
                               nullptr, 0, nullptr, blk_upper, n_upper);
            if (rc != DFH_OK) return rc;
            rc = dfh_gn_global_step(vals, n_blocks, rhs, n_nodes, global_lm, node_dq, global_xi_out, global_scratch, global_scratch_bytes, stream);
            if (rc != DFH_OK) return rc;
        }
    }
    return dfh_gn_iteration_views(sample_pos, sample_nrm, nbr, weights, corr_out, valid_out, n_samples, knn, node_dq, node_pos, node_w, node_nbr,
                                  n_nodes, lw_dq, rw, row_ptr, col, n_blocks, vals, rhs, cost_count, run_id, n_rows, partial, blk_ptr, blk_ent,
                                  node_ptr, node_ent, partial_reg, rblk_ptr, rblk_ent, rnode_ptr, rnode_ent, huber_delta, views, n_views, H, W,
                                  K, Kinv, scale, center, half, max_dist, pcg_iters, lm_abs, lm_rel, x_out, pcg_workspace, pcg_workspace_bytes,
                                  step, n_iters, blk_upper, n_upper, stream);
}

// J^T J is symmetric: block (b, a) is the transpose of block (a, b).  Between ranks only the blocks with col >= row travel
// (about half of `vals`), followed by J^T r and {cost, count}; `src[b]` = index among the travelling blocks of the one that
// holds block b's data (its own, or its mirror's for col < row).  One launch each way, a thread per double.
namespace dfh {
__global__ __launch_bounds__(256) void gn_pack_upper_kernel(const double *__restrict__ system, const int *__restrict__ row_of, const int *__restrict__ col,
                                                             const int *__restrict__ src, int n_blocks, int n_tail, double *__restrict__ packed,
                                                             int n_upper) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long nv = (long)n_blocks * 36;
    if (i < nv) {
        const int b = (int)(i / 36);
        if (col[b] >= row_of[b]) packed[(long)src[b] * 36 + (i - (long)b * 36)] = system[i];
    } else if (i < nv + n_tail) {
        packed[(long)n_upper * 36 + (i - nv)] = system[i];
    }
}
__global__ __launch_bounds__(256) void gn_unpack_upper_kernel(double *__restrict__ system, const int *__restrict__ row_of, const int *__restrict__ col,
                                                               const int *__restrict__ src, int n_blocks, int n_tail, const double *__restrict__ packed,
                                                               int n_upper) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long nv = (long)n_blocks * 36;
    if (i < nv) {
        const int b = (int)(i / 36), e = (int)(i - (long)b * 36);
        const bool upper = col[b] >= row_of[b];
        const int es = upper ? e : (e % 6) * 6 + e / 6;               // the mirror's entry (ib, ia)
        system[i] = packed[(long)src[b] * 36 + es];
    } else if (i < nv + n_tail) {
        system[i] = packed[(long)n_upper * 36 + (i - nv)];
    }
}
}  // namespace dfh

int dfh_gn_pack_upper(const double *system, const int *row_of, const int *col, const int *src, int n_blocks, int n_nodes, int n_upper,
                      double *packed, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(system && row_of && col && src && packed && n_blocks >= 0 && n_nodes >= 0 && n_upper >= 0, "dfh_gn_pack_upper: bad arguments");
    const long n = (long)n_blocks * 36 + 6L * n_nodes + 2;
    hipLaunchKernelGGL(gn_pack_upper_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, system, row_of, col, src,
                       n_blocks, 6 * n_nodes + 2, packed, n_upper);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_gn_unpack_upper(double *system, const int *row_of, const int *col, const int *src, int n_blocks, int n_nodes, int n_upper,
                        const double *packed, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(system && row_of && col && src && packed && n_blocks >= 0 && n_nodes >= 0 && n_upper >= 0, "dfh_gn_unpack_upper: bad arguments");
    const long n = (long)n_blocks * 36 + 6L * n_nodes + 2;
    hipLaunchKernelGGL(gn_unpack_upper_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, system, row_of, col, src,
                       n_blocks, 6 * n_nodes + 2, packed, n_upper);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

size_t dfh_gn_global_step_bytes(void) { return sizeof(double) * (42 * (size_t)dfh::kGlobalWgs + 2); }

int dfh_gn_global_step(const double *vals, int n_blocks, const double *rhs, int n_nodes, double lm_rel, double *node_dq, double *xi_out,
                       void *scratch, size_t scratch_bytes, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(vals && rhs && node_dq && scratch, "dfh_gn_global_step: null pointer");
    DFH_REQUIRE(n_blocks >= 1 && n_nodes >= 1 && lm_rel >= 0.0, "dfh_gn_global_step: bad sizes / damping");
    DFH_REQUIRE(scratch_bytes >= dfh_gn_global_step_bytes(), "dfh_gn_global_step: scratch too small (need %zu bytes, zeroed once)", dfh_gn_global_step_bytes());
    hipLaunchKernelGGL(gn_global_step_kernel, dim3(kGlobalWgs), dim3(256), 0, (hipStream_t)stream, vals, n_blocks, rhs, n_nodes, lm_rel, node_dq,
                       xi_out, static_cast<double *>(scratch));
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

size_t dfh_gn_global_sampled_bytes(int n_samples, int stride) {
    if (n_samples < 0 || stride < 1) return 0;
    const long n_tiles = (n_samples + dfh::kTile - 1) / dfh::kTile;
    (void)n_tiles;
    return sizeof(double) * ((size_t)dfh::kGlobalVals * dfh::kGlobalGrid + 32);                                // workgroup partials | the 29 sums
}

int dfh_gn_global_apply(const double *sums29, double lm_rel, int n_nodes, double *node_dq, double *xi_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(sums29 && node_dq && n_nodes >= 1 && lm_rel >= 0.0, "dfh_gn_global_apply: bad arguments");
    hipLaunchKernelGGL(gn_global_apply_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums29, lm_rel, n_nodes, node_dq, xi_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_gn_global_sampled_views(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights, int n_samples, int knn,
                                double *node_dq, int n_nodes, const double lw_dq[8], double huber_delta, const void *views, int n_views, int H, int W,
                                const double K[9], const double Kinv[9], double scale, const double center[3], double half, double max_dist,
                                int stride, double lm_rel, int n_steps, double *xi_out, double *sums_out, void *scratch, size_t scratch_bytes,
                                void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_steps >= 0 && n_steps <= 100 && stride >= 1, "dfh_gn_global_sampled_views: %d steps, stride %d", n_steps, stride);
    if (n_steps == 0) return DFH_OK;
    DFH_REQUIRE(n_samples >= 0 && node_dq && lw_dq && views && K && Kinv && center && scratch, "dfh_gn_global_sampled_views: null pointer");
    DFH_REQUIRE(n_samples == 0 || (sample_pos && sample_nrm && nbr && weights), "dfh_gn_global_sampled_views: null sample array");
    DFH_REQUIRE(knn == 4, "dfh_gn_global_sampled_views: knn = 4 only (the frame loop's)");
    DFH_REQUIRE(n_views >= 1 && n_views <= DFH_GN_MAX_VIEWS && n_nodes >= 1 && lm_rel >= 0.0 && huber_delta >= 0.0, "dfh_gn_global_sampled_views: bad arguments");
    DFH_REQUIRE(scratch_bytes >= dfh_gn_global_sampled_bytes(n_samples, stride), "dfh_gn_global_sampled_views: scratch too small");
    DFH_REQUIRE(!sums_out || n_steps == 1, "dfh_gn_global_sampled_views: sums_out (the caller reduces over ranks and applies) takes one step per call");
    AssocArgs aa;
    const int rc = fill_assoc_params(aa.ap, lw_dq, H, W, K, Kinv, kIdentity34, scale, center, half, max_dist, knn);
    if (rc != DFH_OK) return rc;
    aa.depth = nullptr;
    aa.views = static_cast<const AssocView *>(views);
    aa.n_views = n_views;
    aa.cull = 0;
    BuildParams bp;
    for (int i = 0; i < 8; ++i) bp.lw.q[i] = lw_dq[i];
    bp.S = n_samples; bp.k = knn; bp.N = n_nodes; bp.huber = huber_delta;
    const long n_tiles = (n_samples + kTile - 1) / kTile;
    const long n_sub = (n_tiles + stride - 1) / stride;
    const int n_wg = (int)std::min<long>(n_sub, kGlobalGrid);
    double *tile_part = static_cast<double *>(scratch);
    double *sums = sums_out ? sums_out : tile_part + (size_t)kGlobalVals * kGlobalGrid;
    hipStream_t st = (hipStream_t)stream;
    for (int g = 0; g < n_steps; ++g) {
        if (n_wg > 0)
            hipLaunchKernelGGL(gn_global_rows_kernel<4>, dim3((unsigned)n_wg), dim3(kTile), 0, st, sample_pos, sample_nrm, nbr, weights,
                               (const double *)node_dq, bp, stride, n_sub, tile_part, aa);
        if (sums_out) hipLaunchKernelGGL(gn_global_finish_kernel<false>, dim3(1), dim3(1024), 0, st, (const double *)tile_part, n_wg, sums, lm_rel, n_nodes, node_dq, xi_out);
        else hipLaunchKernelGGL(gn_global_finish_kernel<true>, dim3(1), dim3(1024), 0, st, (const double *)tile_part, n_wg, sums, lm_rel, n_nodes, node_dq, xi_out);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_relax_twists(double *node_dq, int n_nodes, double factor, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_nodes >= 0 && factor >= 0.0 && factor <= 1.0, "dfh_relax_twists: %d nodes, factor %g (0..1)", n_nodes, factor);
    if (n_nodes == 0 || factor == 1.0) return DFH_OK;
    DFH_REQUIRE(node_dq, "dfh_relax_twists: null pointer");
    hipLaunchKernelGGL(relax_twist_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, (hipStream_t)stream, node_dq, n_nodes, factor);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_apply_twist(double *node_dq, const double *xi, int n_nodes, double step, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_nodes >= 0, "dfh_apply_twist: negative count");
    if (n_nodes == 0) return DFH_OK;
    DFH_REQUIRE(node_dq && xi, "dfh_apply_twist: null pointer");
    hipLaunchKernelGGL(apply_twist_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, (hipStream_t)stream, node_dq, xi, n_nodes, step);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

}  // extern "C"
