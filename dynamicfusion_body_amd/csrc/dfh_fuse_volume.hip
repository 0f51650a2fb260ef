// K2  rigid TSDF -> TSDF fusion:    FusionDM.updateTSDF   (reference core/fusion_dm.py:300-316)
// K3  DQB-warped TSDF -> TSDF fusion: Fusion.updateTSDF   (reference core/fusion.py:153-198)
//
// Both sweep the canonical volume [x][y][z] with z fastest (one 16-byte pack of 4 voxels
// per lane, lanes along z), warp every voxel index into the live volume, sample it with the
// reference's trilinear scheme and blend.  The warp / sampler arithmetic is the exact fp64
// restatement in dfh_dq.h; the blend is evaluated in fp64 with IEEE division and rounded
// once to the volume dtype.
#include "dfh_dq.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace dfh {

template <typename T, int N>
struct alignas(sizeof(T) * N) VPack {
    T v[N];
};

struct RigidParams {
    DQ lw;
    double tdist, wmax;
    int X, Y, Z;            // canonical grid
    int LX, LY, LZ;         // live grid
    int x0, nx;
    int zpacks, zp_shift;
};

__device__ __forceinline__ void pack_coords2(int zpacks, int zp_shift, int &y, int &zp) {
    const int lin = blockIdx.x * 256 + threadIdx.x;
    if (zp_shift >= 0) {
        y = lin >> zp_shift;
        zp = lin & ((1 << zp_shift) - 1);
    } else {
        y = lin / zpacks;
        zp = lin - y * zpacks;
    }
}

template <typename VolT, typename LiveT, int VEC>
__global__ __launch_bounds__(256) void fuse_volume_rigid_kernel(VolT *__restrict__ tsdf, VolT *__restrict__ tsdf_w,
                                                                 const LiveT *__restrict__ live, const RigidParams p) {
    int y, zp;
    pack_coords2(p.zpacks, p.zp_shift, y, zp);
    if (y >= p.Y) return;
    const int xl = blockIdx.y;
    const int x = p.x0 + xl;
    const int z0 = zp * VEC;
    double sv[VEC];
    bool upd[VEC];
    bool any = false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int z = z0 + j;
        const D3 q = dqb_warp_exact(p.lw.q, (double)x, (double)y, (double)z);        // fusion_dm.py:305-306
        double s = 0.0;
        bool ok = (z < p.Z) && interpolate_exact(live, p.LX, p.LY, p.LZ, q.x, q.y, q.z, s);   // :307
        ok = ok && (s > -1.0 * p.tdist);                                               // :308
        sv[j] = s;
        upd[j] = ok;
        any = any | ok;
    }
    if (!any) return;
    const size_t off = ((size_t)xl * p.Y + y) * p.Z + z0;
    using P = VPack<VolT, VEC>;
    P t = *reinterpret_cast<const P *>(tsdf + off);
    P w = *reinterpret_cast<const P *>(tsdf_w + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (upd[j]) {
            const double wt = (double)w.v[j];
            const double m = sv[j] < p.tdist ? sv[j] : p.tdist;
            t.v[j] = (VolT)(((double)t.v[j] * wt + m * 1.0) / (1.0 + wt));           // :311
            const double nw = 1.0 + wt;
            w.v[j] = (VolT)(nw < p.wmax ? nw : p.wmax);                                // :312
        }
    }
    *reinterpret_cast<P *>(tsdf + off) = t;
    *reinterpret_cast<P *>(tsdf_w + off) = w;
}

// K2 fast path for fp32 volumes.  A rigid dual quaternion acts on a point as an affine map
// q = M i + t (M = |r|^2 R(r), t = 2 vec(d r*), folded on the host in fp64), so the per-voxel DQ
// chain collapses to three FMAs per voxel; the trilinear blend uses FMA lerps.  The two decisions
// (inside the live volume, s > -tdist) are guarded: within 1e-9 of a boundary the voxel is
// re-evaluated with the exact chain, so the masks are those of the fp64 path.
struct RigidFastParams {
    double M[9], t[3];
};

template <typename LiveT>
__device__ __forceinline__ double sample_fast(const LiveT *__restrict__ vol, int RY, int RZ, double px, double py, double pz) {
    const double fx = floor(px), fy = floor(py), fz = floor(pz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const int x1 = (int)ceil(px), y1 = (int)ceil(py), z1 = (int)ceil(pz);
    const double xd = px - fx, yd = py - fy, zd = pz - fz;
    const size_t sx = (size_t)RY * RZ, sy = (size_t)RZ;
    const double c000 = (double)vol[x0 * sx + y0 * sy + z0], c100 = (double)vol[x1 * sx + y0 * sy + z0];
    const double c001 = (double)vol[x0 * sx + y1 * sy + z0], c101 = (double)vol[x1 * sx + y1 * sy + z0];
    const double c010 = (double)vol[x0 * sx + y0 * sy + z1], c110 = (double)vol[x1 * sx + y0 * sy + z1];
    const double c011 = (double)vol[x0 * sx + y1 * sy + z1], c111 = (double)vol[x1 * sx + y1 * sy + z1];
    const double c00 = __builtin_fma(xd, c100 - c000, c000), c01 = __builtin_fma(xd, c101 - c001, c001);
    const double c10 = __builtin_fma(xd, c110 - c010, c010), c11 = __builtin_fma(xd, c111 - c011, c011);
    const double c0 = __builtin_fma(yd, c10 - c00, c00);          // y fraction blends the z1 samples (util.py:135)
    const double c1 = __builtin_fma(yd, c11 - c01, c01);
    return __builtin_fma(zd, c1 - c0, c0);                        // z fraction blends the y1 samples (util.py:137)
}

// STRIDED: a lane's 4 voxels are z, z+64, z+128, z+192 of its wave's 256-voxel run, so that in every
// one of the 32 corner gathers (and the T/w accesses) consecutive lanes touch consecutive voxels.
// NT: non-temporal T / w (the slab's pair of volumes exceeds the 256 MiB Infinity Cache: nothing of it is re-read before it is
// evicted, and every 128-byte line is touched by one instruction of one wave -- K1's finding, profiles/r3_rmw_stream_512.txt).
template <typename LiveT, bool STRIDED, bool NT = false>
__global__ __launch_bounds__(256) void fuse_volume_rigid_fast_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                      const LiveT *__restrict__ live, const RigidParams p,
                                                                      const RigidFastParams f) {
    constexpr int VEC = 4;
    int y, zp;
    pack_coords2(p.zpacks, p.zp_shift, y, zp);
    if (y >= p.Y) return;
    const int xl = blockIdx.y;
    const int x = p.x0 + xl;
    const int lane = threadIdx.x & 63;
    const int z0 = STRIDED ? (zp - lane) * VEC + lane : zp * VEC;
    constexpr int ZS = STRIDED ? 64 : 1;
    const double xf = (double)x, yf = (double)y, zf = (double)z0;
    double q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        q[r] = __builtin_fma(f.M[3 * r + 2], zf, __builtin_fma(f.M[3 * r + 1], yf, __builtin_fma(f.M[3 * r], xf, f.t[r])));
    const double hx = (double)(p.LX - 1), hy = (double)(p.LY - 1), hz = (double)(p.LZ - 1);
    const float tdf = (float)p.tdist;
    float mv[VEC];
    bool upd[VEC];
    bool any = false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const double jf = (double)(j * ZS);
        const double qx = __builtin_fma(f.M[2], jf, q[0]), qy = __builtin_fma(f.M[5], jf, q[1]), qz = __builtin_fma(f.M[8], jf, q[2]);
        const double lo = fmin(fmin(qx, qy), qz);
        const double hi = fmin(fmin(hx - qx, hy - qy), hz - qz);
        const double edge = fmin(fabs(lo), fabs(hi));              // distance to the nearest face of [0,R-1]^3 (or worse)
        bool ok = (lo >= 0.0) & (hi >= 0.0);
        bool redo = (int)!(fmin(fmin(fabs(qx), fabs(qy)), fabs(qz)) > 1e-9) | (int)!(fmin(fmin(fabs(hx - qx), fabs(hy - qy)), fabs(hz - qz)) > 1e-9);
        (void)edge;
        double sv = 0.0;
        if (ok) {
            sv = sample_fast(live, p.LY, p.LZ, qx, qy, qz);
            const double margin = sv + p.tdist;
            ok = margin > 0.0;
            redo = redo | !(fabs(margin) > 1e-9 * (1.0 + fabs(sv)));
        }
        if (__builtin_expect(redo, 0)) {
            int xx = x, yy = y, zz = z0 + j * ZS;
            asm volatile("" : "+v"(xx), "+v"(yy), "+v"(zz));       // keep the exact chain out of the hot path
            const D3 e = dqb_warp_exact(p.lw.q, (double)xx, (double)yy, (double)zz);
            ok = interpolate_exact(live, p.LX, p.LY, p.LZ, e.x, e.y, e.z, sv) && (sv > -1.0 * p.tdist);
        }
        ok = ok & (z0 + j * ZS < p.Z);
        mv[j] = (float)(sv < p.tdist ? sv : p.tdist);
        upd[j] = ok;
        any = any | ok;
    }
    (void)tdf;
    if (!any) return;
    const size_t off = ((size_t)xl * p.Y + y) * p.Z + z0;
    using P = VPack<float, VEC>;
    P t, w;
    if (STRIDED) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (NT) { t.v[j] = __builtin_nontemporal_load(tsdf + off + j * ZS); w.v[j] = __builtin_nontemporal_load(tsdf_w + off + j * ZS); }
            else { t.v[j] = tsdf[off + j * ZS]; w.v[j] = tsdf_w[off + j * ZS]; }
        }
    } else {
        t = *reinterpret_cast<const P *>(tsdf + off);
        w = *reinterpret_cast<const P *>(tsdf_w + off);
    }
    const float wmaxf = (float)p.wmax;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float wt = w.v[j];
        const float d = wt + 1.0f;
        const float n = fmaf(t.v[j], wt, mv[j]);                   // (T*w + min(tdist,s)) / (1 + w)
        const float r = __builtin_amdgcn_rcpf(d);
        float qn = n * r;
        qn = fmaf(fmaf(-d, qn, n), r, qn);
        t.v[j] = upd[j] ? qn : t.v[j];
        w.v[j] = upd[j] ? fminf(d, wmaxf) : wt;
    }
    if (STRIDED) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (upd[j]) {
                if (NT) { __builtin_nontemporal_store(t.v[j], tsdf + off + j * ZS); __builtin_nontemporal_store(w.v[j], tsdf_w + off + j * ZS); }
                else { tsdf[off + j * ZS] = t.v[j]; tsdf_w[off + j * ZS] = w.v[j]; }
            }
        }
    } else {
        *reinterpret_cast<P *>(tsdf + off) = t;
        *reinterpret_cast<P *>(tsdf_w + off) = w;
    }
}

// |r|^2 R(r) and 2 vec(d r*) of a (possibly non-unit) dual quaternion, host fp64
static void fold_rigid(const double *q, RigidFastParams &f) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    // r P r* for pure P: rows of the (unnormalised) rotation matrix
    f.M[0] = w * w + x * x - y * y - z * z; f.M[1] = 2 * (x * y - w * z);           f.M[2] = 2 * (x * z + w * y);
    f.M[3] = 2 * (x * y + w * z);           f.M[4] = w * w - x * x + y * y - z * z; f.M[5] = 2 * (y * z - w * x);
    f.M[6] = 2 * (x * z - w * y);           f.M[7] = 2 * (y * z + w * x);           f.M[8] = w * w - x * x - y * y + z * z;
    const double d0 = q[4], d1 = q[5], d2 = q[6], d3 = q[7];
    // d (x) r* , vector part, times 2
    f.t[0] = 2 * (-d0 * x + d1 * w - d2 * z + d3 * y);
    f.t[1] = 2 * (-d0 * y + d1 * z + d2 * w - d3 * x);
    f.t[2] = 2 * (-d0 * z - d1 * y + d2 * x + d3 * w);
}

template <typename VolT, typename LiveT>
static int launch_rigid(void *tsdf, void *tsdf_w, const void *live, RigidParams &p, bool vec4, hipStream_t s) {
    p.zpacks = vec4 ? p.Z / 4 : p.Z;
    p.zp_shift = -1;
    for (int b = 0; b < 31; ++b) if (p.zpacks == (1 << b)) p.zp_shift = b;
    const long per_plane = (long)p.Y * p.zpacks;
    dim3 grid((unsigned)((per_plane + 255) / 256), (unsigned)p.nx), block(256);
    if (vec4) {
        hipLaunchKernelGGL((fuse_volume_rigid_kernel<VolT, LiveT, 4>), grid, block, 0, s, (VolT *)tsdf, (VolT *)tsdf_w,
                           (const LiveT *)live, p);
    } else {
        hipLaunchKernelGGL((fuse_volume_rigid_kernel<VolT, LiveT, 1>), grid, block, 0, s, (VolT *)tsdf, (VolT *)tsdf_w,
                           (const LiveT *)live, p);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

// ------------------------------------------------------------------------------------------
// K3: node search + DQ blending
// ------------------------------------------------------------------------------------------
constexpr int kBX = 4, kBY = 4, kBZ = 16;      // brick = one 256-thread block, z fastest (64-B rows)
constexpr int kCap = 256;                       // candidate nodes kept per brick (<= 256: one per thread when staged)
constexpr int kKMax = 8;                        // knn <= 8
// half diagonal of the voxel-centre span of a brick
#define DFH_BRICK_RADIUS 7.7942286340599480     /* sqrt(1.5^2 + 1.5^2 + 7.5^2) */
// head-room for off-lattice points (dfh_sample_knn_bricks): a point p belongs to the brick of its nearest voxel centre
// v, |p - v| <= sqrt(3)/2, so D_k(p) <= D_k(v) + sqrt(3)/2 and a node among p's k nearest is within
// D_k(p) + sqrt(3)/2 of v, hence of the brick's box
#define DFH_SAMPLE_MARGIN 1.7320508075688774

struct DqbParams {
    DQ lw;
    double tdist, wmax;
    int X, Y, Z;
    int LX, LY, LZ;
    int x0, nx;
    int N, k;
    int nbx, nby, nbz;      // bricks per axis (over the slab)
};

// sorted (ascending, stable) insertion into a KS-slot list held in registers
template <int KS>
__device__ __forceinline__ void topk_insert(double (&bd)[KS], int (&bi)[KS], double d2, int idx) {
    bool ins = false;                       // once inserted, everything below shifts down: equal distances keep their
#pragma unroll
    for (int i = 0; i < KS; ++i) {          // arrival order (a stable sort, KD-tree-like: ties go to the lower node index)
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

template <int KS>
__device__ __forceinline__ double select_k(const double (&bd)[KS], int k) {
    double r = bd[0];
#pragma unroll
    for (int i = 1; i < KS; ++i) r = (k - 1 == i) ? bd[i] : r;
    return r;
}

// Per brick: the nodes that can be among the k nearest of ANY voxel centre p of the brick.
// With c the brick centre and r its radius, D_k(p) <= d_k(c) + r for every p in the brick, so a
// node q can only matter if its distance to the brick's box of voxel centres is <= d_k(c) + r
// (+ DFH_SAMPLE_MARGIN, so that the lists also serve the off-lattice surface samples).
// cand[brick*(kCap+1)] = count (or -1: too many -> scan all nodes).
__global__ __launch_bounds__(256) void dqb_candidates_kernel(const double *__restrict__ node_pos, int *__restrict__ cand,
                                                              const DqbParams p) {
    const long brick = (long)blockIdx.x * 256 + threadIdx.x;
    const long nbricks = (long)p.nbx * p.nby * p.nbz;
    if (brick >= nbricks) return;
    const int bz = (int)(brick % p.nbz);
    const int by = (int)((brick / p.nbz) % p.nby);
    const int bx = (int)(brick / ((long)p.nbz * p.nby));
    const double lx = (double)(p.x0 + bx * kBX), ly = (double)(by * kBY), lz = (double)(bz * kBZ);   // box of voxel centres
    const double hx = lx + (kBX - 1), hy = ly + (kBY - 1), hz = lz + (kBZ - 1);
    const double cx = 0.5 * (lx + hx), cy = 0.5 * (ly + hy), cz = 0.5 * (lz + hz);
    double bd[kKMax];
    int bi[kKMax];
#pragma unroll
    for (int i = 0; i < kKMax; ++i) { bd[i] = __builtin_huge_val(); bi[i] = -1; }
    for (int n = 0; n < p.N; ++n) {
        const double dx = cx - node_pos[3 * n], dy = cy - node_pos[3 * n + 1], dz = cz - node_pos[3 * n + 2];
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < bd[kKMax - 1]) topk_insert<kKMax>(bd, bi, d2, n);
    }
    const double R = sqrt(select_k<kKMax>(bd, p.k)) + DFH_BRICK_RADIUS + DFH_SAMPLE_MARGIN + 1e-6;
    const double R2 = R * R;
    int *c = cand + brick * (kCap + 1);
    int cnt = 0;
    for (int n = 0; n < p.N; ++n) {
        const double qx = node_pos[3 * n], qy = node_pos[3 * n + 1], qz = node_pos[3 * n + 2];
        const double dx = fmax(fmax(lx - qx, qx - hx), 0.0), dy = fmax(fmax(ly - qy, qy - hy), 0.0), dz = fmax(fmax(lz - qz, qz - hz), 0.0);
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 <= R2) {
            if (cnt < kCap) c[1 + cnt] = n;
            ++cnt;
        }
    }
    c[0] = cnt <= kCap ? cnt : -1;
}

// k nearest nodes of `pos` (ascending distance, ties by node index = stable argsort of the
// squared distances; what KDTree.query(pos, k+1)[1][:-1] yields, core/fusion.py:175-176).
// All 256 threads of the block must call this (LDS staging + barriers).
template <int KS>
__device__ __forceinline__ void block_knn(const double *__restrict__ node_pos, const int *__restrict__ c, int N,
                                          double px, double py, double pz, bool active,
                                          double (&bd)[KS], int (&bi)[KS]) {
    __shared__ double spos[kCap * 3];
    __shared__ int sidx[kCap];
    const int cnt = c[0];
    const int total = cnt >= 0 ? cnt : N;
#pragma unroll
    for (int i = 0; i < KS; ++i) { bd[i] = __builtin_huge_val(); bi[i] = -1; }
    for (int base = 0; base < total; base += kCap) {
        const int n = min(kCap, total - base);
        if ((int)threadIdx.x < n) {
            const int gi = cnt >= 0 ? c[1 + base + threadIdx.x] : base + (int)threadIdx.x;
            sidx[threadIdx.x] = gi;
            spos[3 * threadIdx.x + 0] = node_pos[3 * gi + 0];
            spos[3 * threadIdx.x + 1] = node_pos[3 * gi + 1];
            spos[3 * threadIdx.x + 2] = node_pos[3 * gi + 2];
        }
        __syncthreads();
        if (active) {
            for (int i = 0; i < n; ++i) {
                const double dx = px - spos[3 * i], dy = py - spos[3 * i + 1], dz = pz - spos[3 * i + 2];
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < bd[KS - 1]) topk_insert<KS>(bd, bi, d2, sidx[i]);
            }
        }
        __syncthreads();
    }
}

// Fusion.dq_blend + warp (core/fusion.py:502-551) for one point whose k nearest nodes are
// (bd, bi).  Returns the point warped by the blended DQ and then by m_lw (x1 is re-rounded to
// float32 inside the second dqb_warp, core/util.py:69); *wi_out = mean node distance (:180-183).
template <int KS>
__device__ __forceinline__ void dqb_weights(const double *__restrict__ node_w, const double (&bd)[KS], const int (&bi)[KS], int k,
                                            double (&wg)[KS], double &wi) {
    wi = 0.0;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        wg[j] = 0.0;
        if (j < k) {
            const double dist = sqrt(bd[j]);
            const double t = dist / (2.0 * node_w[bi[j]]);
            wg[j] = exp(-1.0 * (t * t));                                 // :537
            wi = wi + dist / (double)k;                                  // mean node distance (:180-183)
        }
    }
}

template <int KS>
__device__ __forceinline__ D3 dqb_blend_warp(const double *__restrict__ node_dq, const double (&wg)[KS], const int (&bi)[KS], int k,
                                             const double *lw, double px, double py, double pz) {
    double b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        if (j < k) {
            const int gi = bi[j];
#pragma unroll
            for (int c = 0; c < 8; ++c) b[c] = b[c] + wg[j] * node_dq[8 * gi + c];   // :538
        }
    }
    // 8-norm (:551), pairwise like numpy's reduction of 8 contiguous values
    const double n2 = ((b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3])) +
                      ((b[4] * b[4] + b[5] * b[5]) + (b[6] * b[6] + b[7] * b[7]));
    const double n = sqrt(n2);
    if (n == 0.0) {                                                       // :544-549
        b[0] = 1.0;
#pragma unroll
        for (int c = 1; c < 8; ++c) b[c] = 0.0;
    } else {
        const double inv = 1.0 / n;                  // one division; each component within 1 ulp of b/n
#pragma unroll
        for (int c = 0; c < 8; ++c) b[c] = b[c] * inv;
    }
    const D3 x1 = dqb_warp_exact(b, px, py, pz);                          // :510
    return dqb_warp_exact(lw, round_f32(x1.x), round_f32(x1.y), round_f32(x1.z));   // :512
}

// MODE 0: search the brick's candidates; 1: search and store per voxel the k indices (and, with a weight cache, the
// k blend weights and the integration weight wi); 2: load the stored indices; 3: load indices and weights.
// The k nearest nodes of a voxel centre, their distances and hence the blend weights depend on the node positions
// and radii only, which stay put while the graph is unchanged (only node_dq moves from frame to frame), so every
// frame after the first skips the search (mode 2: 2*k bytes per voxel read instead of the LDS-staged scan) and the
// sqrt / divide / exp chain (mode 3: another 8*(k+1) bytes).  Stored values are the ones the search path computes,
// so the result is bit-identical.
template <typename VolT, typename LiveT, int KS, int MODE>
__global__ __launch_bounds__(256) void fuse_volume_dqb_kernel(VolT *__restrict__ tsdf, VolT *__restrict__ tsdf_w,
                                                               const LiveT *__restrict__ live,
                                                               const double *__restrict__ node_pos,
                                                               const double *__restrict__ node_dq,
                                                               const double *__restrict__ node_w,
                                                               const int *__restrict__ cand,
                                                               unsigned short *__restrict__ knn_cache,
                                                               double *__restrict__ w_cache, const DqbParams p) {
    const size_t nvox = (size_t)p.nx * p.Y * p.Z;
    const long brick = blockIdx.x;
    int xl, y, z;
    bool inb;
    if (MODE >= 2) {            // no search, no bricks: threads run along z (whole 128-B lines of every per-voxel array)
        const size_t lin = (size_t)blockIdx.x * 256 + threadIdx.x;
        inb = lin < nvox;
        z = (int)(lin % (size_t)p.Z);
        y = (int)((lin / (size_t)p.Z) % (size_t)p.Y);
        xl = (int)(lin / ((size_t)p.Z * p.Y));
    } else {
        const int bz = (int)(brick % p.nbz);
        const int by = (int)((brick / p.nbz) % p.nby);
        const int bx = (int)(brick / ((long)p.nbz * p.nby));
        const int lz = threadIdx.x & (kBZ - 1);
        const int ly = (threadIdx.x >> 4) & (kBY - 1);
        const int lx = threadIdx.x >> 6;
        xl = bx * kBX + lx; y = by * kBY + ly; z = bz * kBZ + lz;
        inb = (xl < p.nx) && (y < p.Y) && (z < p.Z);
    }
    const double px = (double)(p.x0 + xl), py = (double)y, pz = (double)z;
    const size_t off = ((size_t)xl * p.Y + y) * p.Z + z;
    double bd[KS];
    int bi[KS];
    double wg[KS];
    double wi;
    if (MODE >= 2) {
        if (!inb) return;
        unsigned short id[KS];
        if (KS == 4 && p.k == 4) {
            const uint2 v = *reinterpret_cast<const uint2 *>(knn_cache + off * 4);
            id[0] = (unsigned short)(v.x & 0xffffu); id[1] = (unsigned short)(v.x >> 16);
            id[2] = (unsigned short)(v.y & 0xffffu); id[3] = (unsigned short)(v.y >> 16);
        } else {
#pragma unroll
            for (int j = 0; j < KS; ++j) id[j] = j < p.k ? knn_cache[off * p.k + j] : (unsigned short)0;
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) bi[j] = min((int)id[j], p.N - 1);          // a stale or foreign workspace must not fault
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < KS; ++j) wg[j] = j < p.k ? w_cache[(size_t)j * nvox + off] : 0.0;
            wi = w_cache[(size_t)p.k * nvox + off];
        } else {
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const int gi = bi[j];
                const double dx = px - node_pos[3 * gi], dy = py - node_pos[3 * gi + 1], dz = pz - node_pos[3 * gi + 2];
                bd[j] = (dx * dx + dy * dy) + dz * dz;
            }
            dqb_weights<KS>(node_w, bd, bi, p.k, wg, wi);
        }
    } else {
        block_knn<KS>(node_pos, cand + brick * (kCap + 1), p.N, px, py, pz, inb, bd, bi);
        if (!inb) return;
        if (MODE == 1) {
            if (KS == 4 && p.k == 4) {
                uint2 v;
                v.x = (unsigned)bi[0] | ((unsigned)bi[1] << 16);
                v.y = (unsigned)bi[2] | ((unsigned)bi[3] << 16);
                *reinterpret_cast<uint2 *>(knn_cache + off * 4) = v;
            } else {
#pragma unroll
                for (int j = 0; j < KS; ++j) if (j < p.k) knn_cache[off * p.k + j] = (unsigned short)bi[j];
            }
        }
        dqb_weights<KS>(node_w, bd, bi, p.k, wg, wi);
        if (MODE == 1 && w_cache) {
#pragma unroll
            for (int j = 0; j < KS; ++j) if (j < p.k) w_cache[(size_t)j * nvox + off] = wg[j];
            w_cache[(size_t)p.k * nvox + off] = wi;
        }
    }
    const D3 q = dqb_blend_warp<KS>(node_dq, wg, bi, p.k, p.lw.q, px, py, pz);                          // fusion.py:178
    double s;
    if (!interpolate_exact(live, p.LX, p.LY, p.LZ, q.x, q.y, q.z, s)) return;
    if (!(s > -1.0 * p.tdist)) return;                                                              // :179
    double wt = (double)tsdf_w[off];
    if (wt == 0.0) wt = wi;                                                                         // :186-187
    const double m = s < p.tdist ? s : p.tdist;
    tsdf[off] = (VolT)(((double)tsdf[off] * wt + m * wi) / (wi + wt));                              // :189
    const double nw = wi + wt;
    tsdf_w[off] = (VolT)(nw < p.wmax ? nw : p.wmax);                                                // :190
}

// ------------------------------------------------------------------------------------------
// K3 fast path: float32 volumes, knn = 4 (round 3).
// The exact kernel above spends ~350 fp64 operations per voxel on the reference's chain (8-norm with sqrt and a division, two
// dual-quaternion sandwich products, the sampler, an IEEE division in the update) and -- with stored neighbourhoods -- reads
// 48 B of cache per voxel (four 16-bit node indices, four fp64 blend weights, the fp64 integration weight): 3.7-4.0 x the
// algorithmic bytes.  For float32 volumes only the DECISIONS have to be the reference's (which voxels are updated: bit-exact);
// the values carry a float32 rounding anyway (bar: 2 n eps32 (1 + |T|)).  So:
//   * dq_blend is invariant to a common scale of the weights: the cache keeps THREE weights normalised to sum 1 (fp64; the
//     fourth is 1 - their sum) and the integration weight as float32: 28 B per voxel + the indices = 36 B instead of 48.
//     (32-bit fixed-point weights -- 24 B in all -- were considered and rejected: their 2^-33 quantisation moves the warped
//     point by ~1e-8 voxel, and x1 is rounded to float32 inside dqb_warp (core/util.py:69): the rounding decision of 0.1-5 %
//     of the voxels would no longer provably be the reference's.)
//   * the sandwich product is its closed form x' = (M(b) p + t(b)) / |b|_8^2 (M = r P r*, t = 2 vec(d r*)) with one
//     corrected reciprocal; the second warp (m_lw, fixed) is an affine map folded on the host, as in K2;
//   * decisions are guarded and fall back to the exact chain (weights recomputed from the node positions): the float32
//     re-rounding of x1 within 1e-11 relative of a rounding tie (this path's x1 is good to ~1e-13); the sample position
//     within 1e-9 of a face of the live volume or of a cell boundary (the reference's sampler is DIS-continuous across y / z
//     cells: its y fraction blends along z); s within 1e-9 of -tdist; |b|_8 == 0 or weights that do not normalise.
// Every mode (search, search + store, stored) runs THIS arithmetic on the same normalised weights: bit-identical results.
struct DqbNorm {
    double w0, w1, w2;          // w_j / sum, j = 0..2 (the fourth is 1 - (w0 + w1 + w2))
    float wi;                   // mean node distance; negative = "exact chain only" (the weights do not normalise)
};

__device__ __forceinline__ DqbNorm dqb_normalise(const double (&wg)[4], double wi) {
    DqbNorm e;
    const double sum = (wg[0] + wg[1]) + (wg[2] + wg[3]);
    e.wi = (float)wi;
    if (!(sum > 1e-280) || !(sum < __builtin_huge_val())) {          // all-zero blend (core/fusion.py:544-549), overflow, NaN
        e.w0 = e.w1 = e.w2 = 0.0;
        e.wi = -1.0f;
        return e;
    }
    const double inv = 1.0 / sum;
    e.w0 = wg[0] * inv; e.w1 = wg[1] * inv; e.w2 = wg[2] * inv;
    return e;
}

// is `v` within 2^-36 |v| (1.5e-11) of a float32 rounding tie?  The 29 mantissa bits a conversion to float32 drops sit at the
// bottom of the double's low word; a tie is 0x10000000 there, and this path's x1 is good to ~1e-13 relative (a few hundred
// units of the last place), so a window of +-2^16 units decides on integers.  Below 2^-10 the float32 grid is no coarser than
// 1e4 times that error: those few voxels (and NaN / inf) go the exact way.
__device__ __forceinline__ bool near_f32_tie(double v) {
    const unsigned lo = (unsigned)__double2loint(v);
    const bool tie = ((lo & 0x1fffffffu) - (0x10000000u - 0x10000u)) <= 0x20000u;
    return tie | !(fabs(v) > 0x1p-10) | !(fabs(v) < 0x1p+100);
}

// The running average of the float32 paths (core/fusion.py:186-190): ONE expression for every float32 path -- the fast finish,
// the exact chain of a guarded voxel, the constant-live stream -- so that which of them takes a voxel never shows in its bits.
// wi_f: the integration weight rounded to float32 (what the stored neighbourhoods keep).
__device__ __forceinline__ void dqb_update_f32(float t_old, float w_old, float wi_f, double m, double wmax, float &t_new, float &w_new) {
    const double wi = (double)wi_f;
    double wt = (double)w_old;
    if (wt == 0.0) wt = wi;                                                                         // fusion.py:186-187
    const double den = wi + wt;
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
    t_new = (float)(__builtin_fma((double)t_old, wt, m * wi) * r);                                  // :189
    w_new = (float)fmin(den, wmax);                                                                 // :190
}

// interpolate_exact with one rule on top, for float32 volumes only: eight EQUAL corners give that value (exact in real
// arithmetic; the reference's c (1 - d) + c d may land one ulp of the DOUBLE beside it, far inside the float32 bar).  It makes
// the sample of a position whose corners all hold the truncation value independent of the position -- what the constant-live
// skip below relies on -- in the exact chain as in the fast one (whose FMA lerps have the property anyway).
template <typename LiveT>
__device__ __forceinline__ bool interpolate_exact_eq(const LiveT *__restrict__ vol, int RX, int RY, int RZ,
                                                     double px, double py, double pz, double &out) {
    const double mn = fmin(fmin(px, py), pz);
    if (!(mn >= 0.0) || !(px <= (double)(RX - 1)) || !(py <= (double)(RY - 1)) || !(pz <= (double)(RZ - 1))) return false;
    const double fx = floor(px), fy = floor(py), fz = floor(pz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const int x1 = (int)ceil(px), y1 = (int)ceil(py), z1 = (int)ceil(pz);
    const double xd = px - fx, yd = py - fy, zd = pz - fz;
    const size_t sx = (size_t)RY * RZ, sy = (size_t)RZ;
    const double c000 = (double)vol[x0 * sx + y0 * sy + z0], c100 = (double)vol[x1 * sx + y0 * sy + z0];
    const double c001 = (double)vol[x0 * sx + y1 * sy + z0], c101 = (double)vol[x1 * sx + y1 * sy + z0];
    const double c010 = (double)vol[x0 * sx + y0 * sy + z1], c110 = (double)vol[x1 * sx + y0 * sy + z1];
    const double c011 = (double)vol[x0 * sx + y1 * sy + z1], c111 = (double)vol[x1 * sx + y1 * sy + z1];
    if (c000 == c100 && c000 == c001 && c000 == c101 && c000 == c010 && c000 == c110 && c000 == c011 && c000 == c111) {
        out = c000;
        return true;
    }
    const double c00 = c000 * (1.0 - xd) + c100 * xd;
    const double c01 = c001 * (1.0 - xd) + c101 * xd;
    const double c10 = c010 * (1.0 - xd) + c110 * xd;
    const double c11 = c011 * (1.0 - xd) + c111 * xd;
    const double c0 = c00 * (1.0 - yd) + c10 * yd;
    const double c1 = c01 * (1.0 - yd) + c11 * yd;
    out = c0 * (1.0 - zd) + c1 * zd;
    return true;
}

// the reference's chain for one voxel from scratch (weights from the node positions): its decisions and its sample; the running
// average is dqb_update_f32 with the stored float32 integration weight (wi_f < 0: none stored -- the weights did not normalise --
// the one computed here, rounded)
template <typename LiveT>
__device__ __forceinline__ void dqb_exact_voxel(float *__restrict__ tsdf, float *__restrict__ tsdf_w, const LiveT *__restrict__ live,
                                                const double *__restrict__ node_pos, const double *__restrict__ node_dq,
                                                const double *__restrict__ node_w, const int (&bi)[4], const DqbParams &p,
                                                double px, double py, double pz, size_t off, float wi_f) {
    double bd[4], wg[4], wi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gi = bi[j];
        const double dx = px - node_pos[3 * gi], dy = py - node_pos[3 * gi + 1], dz = pz - node_pos[3 * gi + 2];
        bd[j] = (dx * dx + dy * dy) + dz * dz;
    }
    dqb_weights<4>(node_w, bd, bi, 4, wg, wi);
    const D3 q = dqb_blend_warp<4>(node_dq, wg, bi, 4, p.lw.q, px, py, pz);
    double s;
    if (!interpolate_exact_eq(live, p.LX, p.LY, p.LZ, q.x, q.y, q.z, s)) return;
    if (!(s > -1.0 * p.tdist)) return;
    const double m = s < p.tdist ? s : p.tdist;
    float tn, wn;
    dqb_update_f32(tsdf[off], tsdf_w[off], wi_f >= 0.0f ? wi_f : (float)wi, m, p.wmax, tn, wn);
    tsdf[off] = tn;
    tsdf_w[off] = wn;
}

#ifndef DFH_K3_STUB                  // experiment builds (tools/build_variant.sh ... -DDFH_K3_STUB=bits): parts of the steady-state kernel
#define DFH_K3_STUB 0                // left out to see what it is bound by: 1 live gathers, 2 stores, 4 weight loads, 8 DQ rows, 16 index loads
#endif

// where the blend reads a node's eight DQ components from: global memory (a 64-byte row per node) or an LDS copy of the table
struct DqGlobal {
    const double *__restrict__ q;
    __device__ __forceinline__ const double *row(int n) const { return q + 8 * (size_t)n; }
};
// doubles per node in LDS: 80-byte rows spread the 16-byte reads of different nodes over the banks (graphs up to 819 nodes: 64 KB,
// two workgroups per CU); plain 64-byte rows let graphs up to 2 304 nodes (144 KB, one workgroup per CU) use the table too
constexpr int kDqLdsStride = 10, kDqLdsStrideBig = 8;
// sub-lists of the constant-live skip (below): one hot counter serialises ~4 ns per returning atomic at the memory side
constexpr int kSkipLists = 64;
template <int STRIDE>
struct DqLds {
    const double *q;                        // (the caller indexes its __shared__ array directly: the address space survives inlining)
    __device__ __forceinline__ const double *row(int n) const { return q + STRIDE * n; }
};

// The fast path in three steps (the steady-state kernel calls them itself, with deferred exact re-evaluation; dqb_fast_voxel
// runs them back to back with the exact chain inline).  Same operations in the same order either way: the same bits.
struct DqbWarped {
    double qx, qy, qz;          // sample position in the live volume
    double fx, fy, fz;          // its floor (valid when ok)
    bool ok;                    // inside the live volume
    bool redo;                  // a decision is too close to call: this voxel goes through the exact chain
};

// blend -> x1 (closed form) -> float32 re-rounding -> m_lw (affine) -> inside / cell-boundary tests
template <typename DqSrc>
__device__ __forceinline__ DqbWarped dqb_stage_warp(const DqSrc dqs, const int (&bi)[4], const DqbNorm e, const DqbParams &p,
                                                    const RigidFastParams &f, double px, double py, double pz) {
    DqbWarped o;
    bool redo = !(e.wi >= 0.0f);
    double wq[4];
    wq[0] = e.w0; wq[1] = e.w1; wq[2] = e.w2;
    wq[3] = 1.0 - ((wq[0] + wq[1]) + wq[2]);
    double b[8];
    if (DFH_K3_STUB & 8) {
#pragma unroll
        for (int c = 0; c < 8; ++c) b[c] = (c == 0 ? 1.0 : 1e-3 * c) * wq[c & 3] + 1e-4 * bi[c & 3];
    } else {
        const double *d0 = dqs.row(bi[0]);
#pragma unroll
        for (int c = 0; c < 8; ++c) b[c] = wq[0] * d0[c];
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const double *dj = dqs.row(bi[j]);
#pragma unroll
            for (int c = 0; c < 8; ++c) b[c] = __builtin_fma(wq[j], dj[c], b[c]);
        }
    }
    const double n2 = __builtin_fma(b[0], b[0], __builtin_fma(b[1], b[1], __builtin_fma(b[2], b[2], b[3] * b[3]))) +
                      __builtin_fma(b[4], b[4], __builtin_fma(b[5], b[5], __builtin_fma(b[6], b[6], b[7] * b[7])));
    redo = redo | !(n2 > 1e-200);
    double inv = __builtin_amdgcn_rcp(n2);
    inv = __builtin_fma(inv, __builtin_fma(-n2, inv, 1.0), inv);
    inv = __builtin_fma(inv, __builtin_fma(-n2, inv, 1.0), inv);
    // x1 = (r P r* + 2 vec(d r*)) / |b|^2
    const double w = b[0], x = b[1], y = b[2], z = b[3], d0 = b[4], d1 = b[5], d2 = b[6], d3 = b[7];
    const double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
    const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    const double m00 = (ww + xx) - (yy + zz), m11 = (ww + yy) - (xx + zz), m22 = (ww + zz) - (xx + yy);
    const double t0 = __builtin_fma(d3, y, __builtin_fma(-d2, z, __builtin_fma(d1, w, -d0 * x)));
    const double t1 = __builtin_fma(-d3, x, __builtin_fma(d2, w, __builtin_fma(d1, z, -d0 * y)));
    const double t2 = __builtin_fma(d3, w, __builtin_fma(d2, x, __builtin_fma(-d1, y, -d0 * z)));
    const double u0 = __builtin_fma(m00, px, 2.0 * __builtin_fma(xy - wz, py, __builtin_fma(xz + wy, pz, t0)));
    const double u1 = __builtin_fma(m11, py, 2.0 * __builtin_fma(xy + wz, px, __builtin_fma(yz - wx, pz, t1)));
    const double u2 = __builtin_fma(m22, pz, 2.0 * __builtin_fma(xz - wy, px, __builtin_fma(yz + wx, py, t2)));
    const double x1d = u0 * inv, y1d = u1 * inv, z1d = u2 * inv;
    // the float32 re-rounding of x1 inside the second dqb_warp (core/util.py:69)
    const float x1f = (float)x1d, y1f = (float)y1d, z1f = (float)z1d;
    redo = redo | near_f32_tie(x1d) | near_f32_tie(y1d) | near_f32_tie(z1d);
    const double ax = (double)x1f, ay = (double)y1f, az = (double)z1f;
    o.qx = __builtin_fma(f.M[2], az, __builtin_fma(f.M[1], ay, __builtin_fma(f.M[0], ax, f.t[0])));
    o.qy = __builtin_fma(f.M[5], az, __builtin_fma(f.M[4], ay, __builtin_fma(f.M[3], ax, f.t[1])));
    o.qz = __builtin_fma(f.M[8], az, __builtin_fma(f.M[7], ay, __builtin_fma(f.M[6], ax, f.t[2])));
    const double hx = (double)(p.LX - 1), hy = (double)(p.LY - 1), hz = (double)(p.LZ - 1);
    const double lo = fmin(fmin(o.qx, o.qy), o.qz);
    const double hi = fmin(fmin(hx - o.qx, hy - o.qy), hz - o.qz);
    o.ok = (lo >= 0.0) & (hi >= 0.0);
    // (NaN-safe: every comparison is written so that a NaN lands in `redo`)
    redo = redo | !(fmin(fabs(lo), fabs(hi)) > 1e-9);
    o.fx = o.fy = o.fz = 0.0;
    if (o.ok) {
        // cell boundaries: the sampler's y / z fractions blend along the other axis, so it jumps where a coordinate crosses an integer
        o.fx = floor(o.qx); o.fy = floor(o.qy); o.fz = floor(o.qz);
        const double fx = o.qx - o.fx, fy = o.qy - o.fy, fz = o.qz - o.fz;
        redo = redo | !(fmin(fmin(fmin(fx, fy), fz), fmin(fmin(1.0 - fx, 1.0 - fy), 1.0 - fz)) > 1e-9);
    }
    o.redo = redo;
    return o;
}

// the eight corner samples of sample_fast (requested here, used in dqb_stage_finish); voxel 0 when the position is outside
template <typename LiveT>
__device__ __forceinline__ void dqb_stage_gather(const LiveT *__restrict__ live, const DqbParams &p, const DqbWarped &wp, LiveT (&c)[8]) {
    if (DFH_K3_STUB & 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = (LiveT)(0.25 * i + wp.qz * 1e-3);
        return;
    }
    // (outside: fx = fy = fz = 0 and the "ceil" corners are taken at +0: voxel 0, never used)
    const int x0 = (int)wp.fx, y0 = (int)wp.fy, z0 = (int)wp.fz;
    // ceil(q) = floor(q) + (q > floor(q)); a position inside the volume has ceil <= R - 1
    const int dxs = (wp.ok && wp.qx > wp.fx) ? 1 : 0, dys = (wp.ok && wp.qy > wp.fy) ? 1 : 0, dzs = (wp.ok && wp.qz > wp.fz) ? 1 : 0;
    if ((long)p.LX * p.LY * p.LZ < (1L << 31)) {           // (uniform) 32-bit element offsets
        const int sy = p.LZ, sx = p.LY * p.LZ;
        const int b000 = (x0 * p.LY + y0) * p.LZ + z0;
        const int ox = dxs ? sx : 0, oy = dys ? sy : 0, oz = dzs;
        c[0] = live[b000]; c[1] = live[b000 + ox];                                          // c000, c100 of sample_fast
        c[2] = live[b000 + oy]; c[3] = live[b000 + oy + ox];                                // c001, c101  (y1)
        c[4] = live[b000 + oz]; c[5] = live[b000 + oz + ox];                                // c010, c110  (z1)
        c[6] = live[b000 + oy + oz]; c[7] = live[b000 + oy + oz + ox];                      // c011, c111
    } else {
        const size_t sx = (size_t)p.LY * p.LZ, sy = (size_t)p.LZ;
        const size_t x1 = x0 + dxs, y1 = y0 + dys, z1 = z0 + dzs;
        c[0] = live[x0 * sx + y0 * sy + z0]; c[1] = live[x1 * sx + y0 * sy + z0];
        c[2] = live[x0 * sx + y1 * sy + z0]; c[3] = live[x1 * sx + y1 * sy + z0];
        c[4] = live[x0 * sx + y0 * sy + z1]; c[5] = live[x1 * sx + y0 * sy + z1];
        c[6] = live[x0 * sx + y1 * sy + z1]; c[7] = live[x1 * sx + y1 * sy + z1];
    }
}

// `redo` voxels of the steady-state kernel are not re-evaluated in place -- the exact chain inlined into that kernel costs it
// 40 VGPRs, i.e. half its waves -- but appended to a list (their linear voxel offsets) that dqb_redo_kernel works off right
// behind it.  {count, blocks done} live in front of the list; the redo kernel's last block clears them for the next call.
struct DqbRedoList {
    unsigned *count;            // [0] entries, [1] blocks of the redo kernel that have finished
    unsigned *offs;             // capacity: one entry per voxel of the slab (cannot overflow)
};

// sample (sample_fast's lerps), the s > -tdist decision, the running average; a `redo` voxel goes through the exact chain here
// (DEFER = false) or onto the redo list
template <typename LiveT, bool DEFER = false>
__device__ __forceinline__ void dqb_stage_finish(float *__restrict__ tsdf, float *__restrict__ tsdf_w, const LiveT *__restrict__ live,
                                                 const double *__restrict__ node_pos, const double *__restrict__ node_dq,
                                                 const double *__restrict__ node_w, const int (&bi)[4], float wi_f, const DqbParams &p,
                                                 const DqbWarped &wp, const LiveT (&c)[8], double px, double py, double pz, size_t off,
                                                 float t_old, float w_old, const DqbRedoList redo_list = DqbRedoList{nullptr, nullptr}) {
    bool redo = wp.redo;
    bool ok = wp.ok;
    double sv = 0.0;
    if (ok) {
        const double xd = wp.qx - wp.fx, yd = wp.qy - wp.fy, zd = wp.qz - wp.fz;
        const double c000 = (double)c[0], c100 = (double)c[1], c001 = (double)c[2], c101 = (double)c[3];
        const double c010 = (double)c[4], c110 = (double)c[5], c011 = (double)c[6], c111 = (double)c[7];
        const double c00 = __builtin_fma(xd, c100 - c000, c000), c01 = __builtin_fma(xd, c101 - c001, c001);
        const double c10 = __builtin_fma(xd, c110 - c010, c010), c11 = __builtin_fma(xd, c111 - c011, c011);
        const double c0 = __builtin_fma(yd, c10 - c00, c00);          // y fraction blends the z1 samples (util.py:135)
        const double c1 = __builtin_fma(yd, c11 - c01, c01);
        sv = __builtin_fma(zd, c1 - c0, c0);                          // z fraction blends the y1 samples (util.py:137)
        const double margin = sv + p.tdist;
        ok = margin > 0.0;
        redo = redo | !(fabs(margin) > 1e-9 * (1.0 + fabs(sv)));
    }
    if (DEFER) {
        // one atomic per WAVE (an identity warp field puts every sample on a lattice point, i.e. every voxel on this list:
        // 64 x fewer atomics on the one counter); all lanes of the wave reach this point together
        const unsigned long long m = __ballot(redo);
        if (m != 0ull) {
            const int lane = (int)(threadIdx.x & 63);
            unsigned base = 0;
            if (lane == __builtin_ctzll(m)) base = atomicAdd(redo_list.count, (unsigned)__builtin_popcountll(m));
            base = (unsigned)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
            if (redo) redo_list.offs[base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (unsigned)off;
        }
        if (redo) return;
    }
    if (__builtin_expect(redo, 0)) {
        if (DEFER) {
            return;
        } else {
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));              // keep the exact chain's arithmetic inside its branch
            dqb_exact_voxel(tsdf, tsdf_w, live, node_pos, node_dq, node_w, bi, p, px, py, pz, off, wi_f);
        }
        return;
    }
    if (!ok) return;
    float tn, wn;
    dqb_update_f32(t_old, w_old, wi_f, fmin(sv, p.tdist), p.wmax, tn, wn);
    if ((DFH_K3_STUB & 2) && wn != 12345.678f) return;
    tsdf[off] = tn;
    tsdf_w[off] = wn;
}

template <typename LiveT, typename DqSrc>
__device__ __forceinline__ void dqb_fast_voxel(float *__restrict__ tsdf, float *__restrict__ tsdf_w, const LiveT *__restrict__ live,
                                               const double *__restrict__ node_pos, const double *__restrict__ node_dq, const DqSrc dqs,
                                               const double *__restrict__ node_w, const int (&bi)[4], const DqbNorm e,
                                               const DqbParams &p, const RigidFastParams &f, double px, double py, double pz, size_t off,
                                               float t_old, float w_old) {
    const DqbWarped wp = dqb_stage_warp(dqs, bi, e, p, f, px, py, pz);
    LiveT c[8];
    dqb_stage_gather(live, p, wp, c);
    dqb_stage_finish(tsdf, tsdf_w, live, node_pos, node_dq, node_w, bi, e.wi, p, wp, c, px, py, pz, off, t_old, w_old);
}

// MODE as in fuse_volume_dqb_kernel; the cache of modes 1 / 3 holds, behind the indices, three planes of normalised weights
// (fp64, z-contiguous like the exact path's) and one float32 plane of integration weights.
template <typename LiveT, int MODE>
__global__ __launch_bounds__(256) void fuse_volume_dqb_fast_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                    const LiveT *__restrict__ live,
                                                                    const double *__restrict__ node_pos,
                                                                    const double *__restrict__ node_dq,
                                                                    const double *__restrict__ node_w,
                                                                    const int *__restrict__ cand,
                                                                    unsigned short *__restrict__ knn_cache,
                                                                    double *__restrict__ w_cache, const DqbParams p,
                                                                    const RigidFastParams f) {
    const size_t nvox = (size_t)p.nx * p.Y * p.Z;
    float *__restrict__ wi_cache = reinterpret_cast<float *>(w_cache + 3 * nvox);
    const long brick = blockIdx.x;
    int xl, y, z;
    bool inb;
    if (MODE >= 2) {
        const size_t lin = (size_t)blockIdx.x * 256 + threadIdx.x;
        inb = lin < nvox;
        z = (int)(lin % (size_t)p.Z);
        y = (int)((lin / (size_t)p.Z) % (size_t)p.Y);
        xl = (int)(lin / ((size_t)p.Z * p.Y));
    } else {
        const int bz = (int)(brick % p.nbz);
        const int by = (int)((brick / p.nbz) % p.nby);
        const int bx = (int)(brick / ((long)p.nbz * p.nby));
        const int lz = threadIdx.x & (kBZ - 1);
        const int ly = (threadIdx.x >> 4) & (kBY - 1);
        const int lx = threadIdx.x >> 6;
        xl = bx * kBX + lx; y = by * kBY + ly; z = bz * kBZ + lz;
        inb = (xl < p.nx) && (y < p.Y) && (z < p.Z);
    }
    const double px = (double)(p.x0 + xl), py = (double)y, pz = (double)z;
    const size_t off = ((size_t)xl * p.Y + y) * p.Z + z;
    int bi[4];
    DqbNorm e;
    if (MODE >= 2) {
        if (!inb) return;
        const uint2 v = *reinterpret_cast<const uint2 *>(knn_cache + off * 4);
        bi[0] = min((int)(v.x & 0xffffu), p.N - 1); bi[1] = min((int)(v.x >> 16), p.N - 1);          // a stale or foreign workspace must not fault
        bi[2] = min((int)(v.y & 0xffffu), p.N - 1); bi[3] = min((int)(v.y >> 16), p.N - 1);
    } else {
        double bd[4];
        block_knn<4>(node_pos, cand + brick * (kCap + 1), p.N, px, py, pz, inb, bd, bi);
        if (!inb) return;
        if (MODE == 1) {
            uint2 v;
            v.x = (unsigned)bi[0] | ((unsigned)bi[1] << 16);
            v.y = (unsigned)bi[2] | ((unsigned)bi[3] << 16);
            *reinterpret_cast<uint2 *>(knn_cache + off * 4) = v;
        }
        double wg[4], wi;
        dqb_weights<4>(node_w, bd, bi, 4, wg, wi);
        e = dqb_normalise(wg, wi);
        if (MODE == 1 && w_cache) {
            w_cache[off] = e.w0; w_cache[nvox + off] = e.w1; w_cache[2 * nvox + off] = e.w2;
            wi_cache[off] = e.wi;
        }
    }
    if (MODE == 3) {
        e.w0 = w_cache[off]; e.w1 = w_cache[nvox + off]; e.w2 = w_cache[2 * nvox + off];
        e.wi = wi_cache[off];
    } else if (MODE == 2) {
        double bd[4], wg[4], wi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gi = bi[j];
            const double dx = px - node_pos[3 * gi], dy = py - node_pos[3 * gi + 1], dz = pz - node_pos[3 * gi + 2];
            bd[j] = (dx * dx + dy * dy) + dz * dz;
        }
        dqb_weights<4>(node_w, bd, bi, 4, wg, wi);
        e = dqb_normalise(wg, wi);
    }
    dqb_fast_voxel(tsdf, tsdf_w, live, node_pos, node_dq, DqGlobal{node_dq}, node_w, bi, e, p, f, px, py, pz, off, tsdf[off], tsdf_w[off]);
}

// Stored neighbourhoods, steady state (Z a multiple of 64, the node table fits the LDS): persistent workgroups that keep the
// whole node_dq table in LDS and walk 64-voxel runs of z rows.  What the one-run-per-wave kernel above is bound by
// (profiles/r3_k3_experiments.txt): not arithmetic and not cache bytes -- the fp64 chain and the fast path took the same
// 330 us -- but a chain of dependent memory round trips per run at 4-7 waves per SIMD: per-voxel inputs 0.77 us, the four
// nodes' DQ rows (16 gathers of 16 B per lane: 256 B per voxel through a 64 B/clk L1, the texture addresser 93 % busy),
// the eight live-volume samples 2.3 us, the store drain 1.0 us (in-kernel stamps of every phase, -DDFH_K3_TRACE builds of
// round 3).  Here the DQ rows come from LDS (16 ds_read on a path of their own), the coordinates are wave-uniform scalar
// arithmetic (the kernel above divides a 64-bit linear index per lane), and undecidable voxels go onto a list instead of through
// an inlined exact chain (63 instead of 102 VGPRs): 340 -> 248 us at 256^3 / 512 nodes.  (A software pipeline over three runs
// -- inputs of run i+2 and live samples of run i+1 in flight while run i is finished -- was built and measured slower, 267 us:
// with the memory operations stubbed out the kernel still takes 172 us; it is bound by its ~220 fp64 instructions per run.)
struct DqbRunInputs {
    uint2 idx;
    double w0, w1, w2;
    float wi, t, w;
};

// LIST: only the bricks on the skip's sub-lists are taken (the constant-live skip below leaves the bricks near the live surface).
template <typename LiveT, int TPB, int STRIDE, bool LIST = false>
__global__ __launch_bounds__(TPB) void fuse_volume_dqb_lds_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w,
                                                                   const LiveT *__restrict__ live,
                                                                   const double *__restrict__ node_pos,
                                                                   const double *__restrict__ node_dq,
                                                                   const double *__restrict__ node_w,
                                                                   const unsigned short *__restrict__ knn_cache,
                                                                   const double *__restrict__ w_cache, const DqbParams p,
                                                                   const RigidFastParams f, int n_runs, const DqbRedoList redo_list,
                                                                   const unsigned *__restrict__ list_count = nullptr,
                                                                   const unsigned *__restrict__ list = nullptr, unsigned sub_cap = 0) {
    extern __shared__ double sdq[];                                           // N rows of STRIDE doubles
    for (int i = threadIdx.x; i < p.N * 8; i += TPB) sdq[(i >> 3) * STRIDE + (i & 7)] = node_dq[i];
    __syncthreads();
    const size_t nvox = (size_t)p.nx * p.Y * p.Z;
    const float *__restrict__ wi_cache = reinterpret_cast<const float *>(w_cache + 3 * nvox);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (TPB / 64) + (threadIdx.x >> 6)));
    const int n_waves = (int)gridDim.x * (TPB / 64);
    const int runs_per_row = p.Z >> 6;
    auto fetch = [&](int r, DqbRunInputs &in) {
        const size_t off = (size_t)r * 64 + lane;
        if (DFH_K3_STUB & 16) in.idx = make_uint2((unsigned)(r & 255) | 0x10000u * ((r + 7) & 255), (unsigned)((r + 3) & 255) | 0x10000u * ((r + 11) & 255));
        else in.idx = *reinterpret_cast<const uint2 *>(knn_cache + off * 4);
        if (DFH_K3_STUB & 4) { in.w0 = 0.4 + 1e-9 * lane; in.w1 = 0.3; in.w2 = 0.2; in.wi = 30.0f; }
        else {
            in.w0 = w_cache[off]; in.w1 = w_cache[nvox + off]; in.w2 = w_cache[2 * nvox + off];
            in.wi = wi_cache[off];
        }
        in.t = tsdf[off]; in.w = tsdf_w[off];
    };
    auto coords = [&](int r, double &px, double &py, double &pz) {
        const int row = r / runs_per_row;                                      // (wave-uniform arithmetic)
        const int xl = row / p.Y;
        px = (double)(p.x0 + xl); py = (double)(row - xl * p.Y); pz = (double)((r - row * runs_per_row) * 64 + lane);
    };
    auto nodes_of = [&](uint2 idx, int (&bi)[4]) {
        bi[0] = min((int)(idx.x & 0xffffu), p.N - 1); bi[1] = min((int)(idx.x >> 16), p.N - 1);   // a stale or foreign workspace must not fault
        bi[2] = min((int)(idx.y & 0xffffu), p.N - 1); bi[3] = min((int)(idx.y >> 16), p.N - 1);
    };
    if (LIST) {
        // The bricks dqb_bound_kernel left to this kernel, on kSkipLists sub-lists in brick order (z fastest).  A pass takes ONE
        // row (of a brick's 16) of FOUR consecutive entries: lanes 16 g .. 16 g + 15 the 16 voxels of entry 4 q + g -- mostly
        // neighbours along z, i.e. up to 64 contiguous voxels (four rows of one brick per pass, 64-byte pieces four rows apart,
        // measured 11.5 us per pass against 9.2: the texture addresser pays per cache line).  Wave w serves sub-list
        // w % kSkipLists and takes the items w / kSkipLists, + n_waves / kSkipLists, ...: the same number of passes per wave, +-1.
        const int l = wave % kSkipLists, first = wave / kSkipLists, step = max(n_waves / kSkipLists, 1);
        if (wave >= step * kSkipLists) return;                                 // (a grid that is not a multiple of the list count)
        const unsigned *__restrict__ mine = list + (size_t)l * sub_cap;
        const int n_entries = (int)min(list_count[16 * l], sub_cap);
        const int n_items = 16 * ((n_entries + 3) >> 2);                       // item = (four consecutive entries, one of a brick's 16 rows)
        const unsigned last_brick = (unsigned)(p.nbx * p.nby * p.nbz - 1);
        const int g = lane >> 4;
        auto entry_of = [&](int it) { const int e = 4 * (it >> 4) + g; return e < n_entries ? min(mine[e], last_brick) : 0xffffffffu; };
        unsigned entry = first < n_items ? entry_of(first) : 0xffffffffu;      // (the next item's entries are asked for a whole item ahead)
        for (int it = first; it < n_items; it += step) {
            const unsigned brick = entry;
            if (it + step < n_items) entry = entry_of(it + step);
            const int rr = it & 15;
            const bool have = brick != 0xffffffffu;
            const unsigned bq = have ? brick : 0u;
            const int bz = (int)(bq % (unsigned)p.nbz), bxy = (int)(bq / (unsigned)p.nbz);
            const int by = bxy % p.nby, bx = bxy / p.nby;
            const int xl = bx * kBX + (rr >> 2), y = by * kBY + (rr & 3);
            const bool valid = have && xl < p.nx && y < p.Y;
            const size_t off = ((size_t)xl * p.Y + y) * p.Z + (size_t)(bz * kBZ + (lane & 15));
            const double px = (double)(p.x0 + xl), py = (double)y, pz = (double)(bz * kBZ + (lane & 15));
            DqbRunInputs in;
            in.idx = make_uint2(0u, 0u); in.w0 = 1.0; in.w1 = in.w2 = 0.0; in.wi = 1.0f; in.t = in.w = 0.0f;
            if (valid) {
                in.idx = *reinterpret_cast<const uint2 *>(knn_cache + off * 4);
                in.w0 = w_cache[off]; in.w1 = w_cache[nvox + off]; in.w2 = w_cache[2 * nvox + off];
                in.wi = wi_cache[off];
                in.t = tsdf[off]; in.w = tsdf_w[off];
            }
            int bi[4];
            nodes_of(in.idx, bi);
            DqbNorm e;
            e.w0 = in.w0; e.w1 = in.w1; e.w2 = in.w2; e.wi = in.wi;
            DqbWarped wp = dqb_stage_warp(DqLds<STRIDE>{sdq}, bi, e, p, f, px, py, pz);
            if (!valid) { wp.ok = false; wp.redo = false; wp.fx = wp.fy = wp.fz = 0.0; }
            LiveT c[8];
            dqb_stage_gather(live, p, wp, c);
            dqb_stage_finish<LiveT, true>(tsdf, tsdf_w, live, node_pos, node_dq, node_w, bi, e.wi, p, wp, c, px, py, pz, off, in.t, in.w, redo_list);
        }
        return;
    }
    if (wave >= n_runs) return;
    for (int r = wave; r < n_runs; r += n_waves) {                            // consecutive waves take consecutive runs
        DqbRunInputs in;
        fetch(r, in);
        double px, py, pz;
        coords(r, px, py, pz);
        int bi[4];
        nodes_of(in.idx, bi);
        DqbNorm e;
        e.w0 = in.w0; e.w1 = in.w1; e.w2 = in.w2; e.wi = in.wi;
        const DqbWarped wp = dqb_stage_warp(DqLds<STRIDE>{sdq}, bi, e, p, f, px, py, pz);
        LiveT c[8];
        dqb_stage_gather(live, p, wp, c);
        dqb_stage_finish<LiveT, true>(tsdf, tsdf_w, live, node_pos, node_dq, node_w, bi, e.wi, p, wp, c, px, py, pz, (size_t)r * 64 + lane,
                                      in.t, in.w, redo_list);
    }
}

// The voxels the steady-state kernel could not decide (a float32 rounding tie of x1, a sample on a cell or volume boundary, s
// at -tdist: ~1e-4 of them), through the reference's chain.  Fixed grid; the last block to finish clears the list's header.
// (fixed grid: the list's length is known on the device only.  64 blocks: 12.6 us at the usual ~0.5 % of a 256^3 volume; 512 blocks,
// one voxel per thread, took 16.7 us -- the launch is the header's round trip and the last block's hand-shake, not the voxels)
constexpr int kRedoBlocks = 64;
template <typename LiveT>
__global__ __launch_bounds__(256) void dqb_redo_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w, const LiveT *__restrict__ live,
                                                        const double *__restrict__ node_pos, const double *__restrict__ node_dq,
                                                        const double *__restrict__ node_w, const unsigned short *__restrict__ knn_cache,
                                                        const float *__restrict__ wi_cache, const DqbParams p, const DqbRedoList redo_list) {
    const unsigned n = redo_list.count[0];
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const size_t off = redo_list.offs[i];
        const int z = (int)(off % (size_t)p.Z);
        const int y = (int)((off / (size_t)p.Z) % (size_t)p.Y);
        const int xl = (int)(off / ((size_t)p.Z * p.Y));
        const uint2 v = *reinterpret_cast<const uint2 *>(knn_cache + off * 4);
        int bi[4];
        bi[0] = min((int)(v.x & 0xffffu), p.N - 1); bi[1] = min((int)(v.x >> 16), p.N - 1);
        bi[2] = min((int)(v.y & 0xffffu), p.N - 1); bi[3] = min((int)(v.y >> 16), p.N - 1);
        dqb_exact_voxel(tsdf, tsdf_w, live, node_pos, node_dq, node_w, bi, p, (double)(p.x0 + xl), (double)y, (double)z, off, wi_cache[off]);
    }
    __syncthreads();                                                          // (this block's reads of count[0] are done)
    if (threadIdx.x == 0 && atomicAdd(redo_list.count + 1, 1u) == gridDim.x - 1) {
        redo_list.count[0] = 0u;
        redo_list.count[1] = 0u;
    }
}

// ------------------------------------------------------------------------------------------
// Constant-live skip (round 4; profiles/r3_k3_experiments.txt section 6 left it open for want of a bound).
// The live volume holds the truncation value everywhere except within tdist of the live surface; a voxel all of whose possible
// sample corners hold exactly tdist takes s = tdist WHATEVER its warped position: the update needs T, w and the stored
// integration weight (12-20 B per voxel moved instead of 64, a dozen instructions instead of ~220) -- no indices, no blend
// weights, no node rows, no gathers.  What it takes is a PROVEN bound on |warp(p) - p|.
//
// Bound.  b = sum_j w_j dq_j with w_j >= 0, sum 1 (the stored weights are normalised; the exact chain's are a positive multiple
// and the 8-norm normalisation removes any common factor).  With r, d its real and dual parts the reference's warp (dq_blend +
// dqb_warp, core/fusion.py:527-551, core/util.py:68-72) is x1 = (r p r* + 2 vec(d r*)) / (|r|^2 + |d|^2), so
//     x1 - p = N(b, p) / (|r|^2 + |d|^2),   N(b, p) = (r p r* - |r|^2 p) + 2 vec(d r*) - |d|^2 p,
// a quadratic form in b.  (The last term is the 8-norm's: a unit dual quaternion with translation t has |d| = |t| / 2 and the
// reference's normalisation SCALES the point by 1 / (1 + |t|^2 / 4); in the frame loop it moves voxels far from the surface by up
// to 15 voxels -- tools/k3_skip_probe.py.)  With G its symmetric bilinear form, G(a, a, p) = N(a, p),
//     G(a, b, p) = (a_r p b_r* + b_r p a_r*) / 2 - (a_r . b_r + a_d . b_d) p + vec(a_d b_r* + b_d a_r*),
// a convex combination gives N(b, p) = sum_jk w_j w_k G(dq_j, dq_k, p), hence |N(b, p)| <= max_jk |G(dq_j, dq_k, p)|, and likewise
// |r|^2 + |d|^2 >= |r|^2 >= min_jk r_j . r_k -- the maxima over the pairs of nodes that can blend in.  Over a brick with centre c
// and half diagonal rho_b:  G(a, b, c + p') - G(a, b, c) = (v_a + v_b) x p' + [(da p' db* + db p' da*) / 2 - (da . db) p'] - (a_d . b_d) p'
// (r = 1 + delta, v = its vector part), of norm <= (|v_a| + |v_b| + 2 |da| |db| + |a_d . b_d|) rho_b.  So
//     D = max_jk [ |G_jk(c)| + (|v_j| + |v_k| + 2 |delta_j| |delta_k| + |d_j . d_k|) rho_b ] / min_jk r_j . r_k
// bounds |x1 - p| for every voxel of the brick (cross-checked against the oracle's dq_blend + dqb_warp on random blends,
// tests/test_k3_skip_bound.py; on the device against every voxel, tests/test_gpu_fuse_volume.py::test_dqb_skip_bound_holds_for_every_voxel;
// on the bench scene the largest displacement / bound is 0.98).  On top:
// the float32 rounding of x1 (core/util.py:69), the paths' own error and 1e-3 voxel of slack.  m_lw must be the identity (the
// frame loop's case); otherwise no skip.  The sample's corners floor / ceil lie within D + 1 of p on every axis.
// The nodes that can blend into a brick: the union of its voxels' stored k nearest nodes (`used`, built with the stored
// neighbourhoods; static while the graph stays) -- 6-12 nodes where the candidate lists hold 23.
//
// Masks.  U: one bit per 4x4x4 cell of the LIVE volume, set iff its 64 voxels all equal tdist (cells cut by a face: clear).  S: one
// bit per cell of the canonical slab, set for the four cells of a brick iff every live cell within reach = ceil((D + 1) / 4) cells
// (Chebyshev) of them exists and is set in U -- then every corner of every voxel of the brick holds tdist, the position is inside
// the live volume, s = tdist exactly (eight equal corners: the FMA lerps return the value, interpolate_exact_eq makes the exact
// chain do the same), s > -tdist holds and the update is dqb_update_f32(T, w, wi, tdist).  Such bricks are taken by
// dqb_stream_kernel (16-byte packs, no warp); the LDS kernel takes the other bricks, four of a brick's 16-voxel rows per pass.  Bits: identical to the kernels without the skip (test_dqb_constant_live_skip), because every path ends in
// dqb_update_f32.
struct DqbSkip {
    unsigned long long *U;          // word (cx * CY + cy) * WZ + (cz >> 6), bit cz & 63
    unsigned long long *S;          // one BYTE per K3 brick: 1 = its voxels take the constant-live stream
    unsigned char *mb;              // per K3 brick (4 x 4 x 16 = four cells along z): cells to look around, 1 or 2; 255 = no bound
    float *db;                      // per K3 brick: D (diagnostics and tests)
    unsigned short *used;           // per K3 brick: kSkipUsed node ids, ascending, 0xffff = none; [0] = 0xfffe: more than fit
    unsigned *sub_count;            // kSkipLists counters, 64 B apart (index 16 l): bricks on sub-list l
    unsigned *sub_list;             // kSkipLists sub-lists of sub_cap entries: the warp kernel's bricks; brick b goes on list b % kSkipLists
    unsigned sub_cap;
    int CX, CY, CZ, WZ;             // cells of the live volume, 64-bit words per cell row
    int SCX, SCY;                   // cell rows of the slab
};
constexpr int kSkipMaxWords = 8;    // LZ <= 2048
constexpr int kSkipUsed = 16;

template <typename LiveT>
__global__ __launch_bounds__(1024) void dqb_live_mask_kernel(const LiveT *__restrict__ live, int LX, int LY, int LZ, double tdist,
                                                              const DqbSkip k, int cx_lo) {
    __shared__ unsigned long long part[16][kSkipMaxWords];
    const int cell_row = cx_lo * k.CY + blockIdx.x;                            // (only the cell planes within reach of the slab)
    const int cx = cell_row / k.CY, cy = cell_row - cx * k.CY;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = 4 * cx + (wave >> 2), y = 4 * cy + (wave & 3);
    const bool row_ok = x < LX && y < LY;                                      // (a cell cut by the volume's face stays clear)
    for (int wz = 0; wz < k.WZ; ++wz) {
        const int cz = wz * 64 + lane;
        bool f = false;
        if (row_ok && cz < k.CZ) {
            const VPack<LiveT, 4> v = *reinterpret_cast<const VPack<LiveT, 4> *>(live + ((size_t)x * LY + y) * LZ + 4 * (size_t)cz);
            f = (double)v.v[0] == tdist && (double)v.v[1] == tdist && (double)v.v[2] == tdist && (double)v.v[3] == tdist;
        }
        const unsigned long long b = __ballot(f);
        if (lane == 0) part[wave][wz] = b;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + kSkipLists) k.sub_count[16 * (threadIdx.x - 64)] = 0u;   // (filled by dqb_bound_kernel, behind this launch)
    if ((int)threadIdx.x < k.WZ) {
        unsigned long long u = part[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < 16; ++w) u &= part[w][threadIdx.x];
        k.U[(size_t)cell_row * k.WZ + threadIdx.x] = u;
    }
}

// The nodes a brick's voxels blend (union of their stored k nearest nodes), ascending: one block per brick, a voxel per thread.
// Every stored index is on the brick's candidate list (sorted by node id), so a binary search finds its slot.
__global__ __launch_bounds__(256) void dqb_used_nodes_kernel(const unsigned short *__restrict__ knn_cache, const int *__restrict__ cand,
                                                              const DqbParams p, const DqbSkip k) {
    __shared__ int slist[kCap];
    __shared__ unsigned char flag[kCap];
    __shared__ int wave_cnt[4];
    const long brick = blockIdx.x;
    const int bz = (int)(brick % p.nbz);
    const int by = (int)((brick / p.nbz) % p.nby);
    const int bx = (int)(brick / ((long)p.nbz * p.nby));
    const int lz = threadIdx.x & (kBZ - 1), ly = (threadIdx.x >> 4) & (kBY - 1), lx = threadIdx.x >> 6;
    const int xl = bx * kBX + lx, y = by * kBY + ly, z = bz * kBZ + lz;
    const bool inb = xl < p.nx && y < p.Y && z < p.Z;
    const int *c = cand + brick * (kCap + 1);
    const int cnt = c[0];
    unsigned short *out = k.used + brick * kSkipUsed;
    if (cnt < 0) {                                                             // overflowed list: any node may blend in -> no bound
        if (threadIdx.x < kSkipUsed) out[threadIdx.x] = threadIdx.x == 0 ? (unsigned short)0xfffe : (unsigned short)0xffff;
        return;
    }
    if ((int)threadIdx.x < cnt) slist[threadIdx.x] = c[1 + threadIdx.x];
    flag[threadIdx.x] = 0;
    __syncthreads();
    if (inb) {
        const size_t off = ((size_t)xl * p.Y + y) * p.Z + z;
        const uint2 v = *reinterpret_cast<const uint2 *>(knn_cache + off * 4);
        const int id[4] = {(int)(v.x & 0xffffu), (int)(v.x >> 16), (int)(v.y & 0xffffu), (int)(v.y >> 16)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int lo = 0, hi = cnt - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (slist[mid] < id[j]) lo = mid + 1; else hi = mid;
            }
            if (slist[lo] == id[j]) flag[lo] = 1;
        }
    }
    __syncthreads();
    const bool f = (int)threadIdx.x < cnt && flag[threadIdx.x] != 0;
    const unsigned long long m = __ballot(f);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wave_cnt[wave] = __builtin_popcountll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_cnt[w];
    const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (total > kSkipUsed) {
        if (threadIdx.x < kSkipUsed) out[threadIdx.x] = threadIdx.x == 0 ? (unsigned short)0xfffe : (unsigned short)0xffff;
        return;
    }
    if ((int)threadIdx.x >= total && threadIdx.x < kSkipUsed) out[threadIdx.x] = (unsigned short)0xffff;
    if (f) out[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (unsigned short)slist[threadIdx.x];
}

// bit i of the result = bit i + s of the cell row (cur = this word, nxt = the next one along z); s in 1..63
__device__ __forceinline__ unsigned long long row_shr(unsigned long long cur, unsigned long long nxt, int s) { return (cur >> s) | (nxt << (64 - s)); }
__device__ __forceinline__ unsigned long long row_shl(unsigned long long cur, unsigned long long prv, int s) { return (cur << s) | (prv >> (64 - s)); }
__device__ __forceinline__ unsigned long long wave_and_u64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo &= (unsigned)__shfl_xor((int)lo, o, 64);
        hi &= (unsigned)__shfl_xor((int)hi, o, 64);
    }
    return ((unsigned long long)hi << 32) | lo;
}

// How far around a brick the live volume is constant: rs[brick] = 2 / 1 / 0 = every live cell within two / one cell(s)
// (Chebyshev) of the brick's four cells exists and is set in U / not even that.  One wave per slab cell row, a lane per
// neighbouring cell row (5 x 5), the words eroded along z and AND-ed over the wave.  (Written into mb; dqb_bound_kernel reads it
// there and replaces it by the reach the bound asks for.)
__global__ __launch_bounds__(256) void dqb_reach_kernel(const DqbParams p, const DqbSkip k) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= k.SCX * k.SCY) return;
    const int lane = threadIdx.x & 63;
    const int scx = t / k.SCY, cy = t - scx * k.SCY;
    const int cx = p.x0 / 4 + scx;
    const int dx = lane / 5 - 2, dy = lane % 5 - 2;
    const bool nb = lane < 25;
    const int ax = cx + dx, ay = cy + dy;
    const bool inside = nb && ax >= 0 && ax < k.CX && ay >= 0 && ay < k.CY;   // outside the live volume: never constant
    const unsigned long long *urow = k.U + ((size_t)(inside ? ax : 0) * k.CY + (inside ? ay : 0)) * k.WZ;
    unsigned char *rs = k.mb + ((long)scx * p.nby + cy) * p.nbz;
    for (int wz = 0; wz < k.WZ; ++wz) {
        unsigned long long e1 = ~0ull, e2 = ~0ull;
        if (nb) {
            const unsigned long long cur = inside ? urow[wz] : 0ull, prv = inside && wz > 0 ? urow[wz - 1] : 0ull,
                                     nxt = inside && wz + 1 < k.WZ ? urow[wz + 1] : 0ull;
            const unsigned long long a1 = cur & row_shr(cur, nxt, 1) & row_shl(cur, prv, 1);
            e2 = a1 & row_shr(cur, nxt, 2) & row_shl(cur, prv, 2);
            if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1) e1 = a1;
        }
        e1 = wave_and_u64(e1);
        e2 = wave_and_u64(e2);
        if (lane < 16) {
            const int bz = wz * 16 + lane;
            const unsigned long long grp = 0xfull << (4 * lane);
            if (bz < p.nbz) rs[bz] = (e2 & grp) == grp ? 2 : ((e1 & grp) == grp ? 1 : 0);
        }
    }
}

// Per brick: the bound D, the reach it asks for and the verdict (Sb = 1: its voxels take the constant-live stream; 0: the warp
// kernel, which finds it on one of kSkipLists sub-lists: one list behind ONE counter cost 100 us for 26 k bricks -- same-address
// returning atomics serialise at the memory side -- and flags read by statically assigned waves left the launch as long as
// without the skip, 215 us: the live surface's shell falls on few of them).  16 lanes per brick: lane j keeps node used[j]'s dual
// quaternion (LDS); the pairs (j <= k) are dealt densely over the lanes.  all_bounds: compute D also where the live volume
// rules the skip out anyway (tests).
__global__ __launch_bounds__(256) void dqb_bound_kernel(const double *__restrict__ node_dq, const DqbParams p, const DqbSkip k, int all_bounds) {
    // float32 from here on: the pair loop is the kernel (28 pairs of ~130 operations per brick), and what it produces is a BOUND.
    // Its own rounding: products of size <= |c| |q|^2 ~ 500 carry 2^-24 relative each, ~40 of them add up along a component:
    // < 2.4e-6 |c| voxel; the node values rounded to float32: another 1e-7 |c|.  4e-6 |c| + 2e-3 voxel are added below for both.
    __shared__ float sq[16][kSkipUsed * 15 + 1];                               // per node: dq (8), a = r (0, c) (4), |v|, |r - 1|; 15 floats, 241 per brick
    const long nbricks = (long)p.nbx * p.nby * p.nbz;
    const long brick_raw = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool real = brick_raw < nbricks;
    const long brick = real ? brick_raw : 0;
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int bz = (int)(brick % p.nbz);
    const int by = (int)((brick / p.nbz) % p.nby);
    const int bx = (int)(brick / ((long)p.nbz * p.nby));
    const double cxd = (double)(p.x0 + bx * kBX) + 0.5 * (kBX - 1), cyd = (double)(by * kBY) + 0.5 * (kBY - 1),
                 czd = (double)(bz * kBZ) + 0.5 * (kBZ - 1);
    const float cx = (float)cxd, cy = (float)cyd, cz = (float)czd;             // (half-integers below 2^24: exact)
    const int rs = real ? (int)k.mb[brick] : 0;                                // dqb_reach_kernel's answer (this kernel overwrites it)
    const bool want = real && (rs > 0 || all_bounds);                          // (uniform over the brick's 16 lanes)
    const unsigned short id = want ? k.used[brick * kSkipUsed + sub] : (unsigned short)0xffff;
    const bool have = id < 0xfffe && (int)id < p.N;
    bool bad = false;
    float me[8] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (have) {
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) me[c8] = (float)node_dq[8 * (size_t)id + c8];
    }
    float *mine = &sq[grp][sub * 15];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        mine[c8] = me[c8];
        bad = bad | !(fabsf(me[c8]) < 1e6f);                                   // (NaN / inf land here)
    }
    mine[8] = -(me[1] * cx + me[2] * cy + me[3] * cz);                         // a = r (0, c)
    mine[9] = me[0] * cx + me[2] * cz - me[3] * cy;
    mine[10] = me[0] * cy + me[3] * cx - me[1] * cz;
    mine[11] = me[0] * cz + me[1] * cy - me[2] * cx;
    const float vv = (me[1] * me[1] + me[2] * me[2]) + me[3] * me[3];
    mine[12] = sqrtf(vv);
    mine[13] = sqrtf((me[0] - 1.0f) * (me[0] - 1.0f) + vv);
    const int n = __builtin_popcount((unsigned)(__ballot(have) >> (16 * (grp & 3))) & 0xffffu);       // ids are packed to the front
    const bool overflow = (__shfl((int)id, (int)(threadIdx.x & 48), 64) & 0xffff) == 0xfffe;   // the group's lane 0
    // (a group reads only what its own 16 lanes wrote, all in one wave: the wave's LDS operations execute in order -- no workgroup
    // barrier in this kernel, its waves run independently: 200 -> 150 us at 512^3)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    float num = 0.0f, n2min = __builtin_huge_valf();
    const int npairs = n * (n + 1) / 2;
    for (int q = sub; q < npairs; q += 16) {
        int j = 0, rem = q;
        while (rem >= n - j) { rem -= n - j; ++j; }
        const int kk = j + rem;
        const float *a = &sq[grp][j * 15], *b = &sq[grp][kk * 15];
        const float aw = a[0], ax = a[1], ay = a[2], az = a[3], a0 = a[4], a1 = a[5], a2 = a[6], a3 = a[7];
        const float bw = b[0], bx_ = b[1], by_ = b[2], bz_ = b[3], b0 = b[4], b1 = b[5], b2 = b[6], b3 = b[7];
        const float q1w = a[8], q1x = a[9], q1y = a[10], q1z = a[11], q2w = b[8], q2x = b[9], q2y = b[10], q2z = b[11];
        // vec(q b*) for b* = (bw, -bx, -by, -bz): x: -qw bx + qx bw - qy bz + qz by, ...
        const float s1x = -q1w * bx_ + q1x * bw - q1y * bz_ + q1z * by_, s1y = -q1w * by_ + q1y * bw - q1z * bx_ + q1x * bz_, s1z = -q1w * bz_ + q1z * bw - q1x * by_ + q1y * bx_;
        const float s2x = -q2w * ax + q2x * aw - q2y * az + q2z * ay, s2y = -q2w * ay + q2y * aw - q2z * ax + q2x * az, s2z = -q2w * az + q2z * aw - q2x * ay + q2y * ax;
        // vec(a_d b_r*) + vec(b_d a_r*)
        const float t1x = -a0 * bx_ + a1 * bw - a2 * bz_ + a3 * by_, t1y = -a0 * by_ + a2 * bw - a3 * bx_ + a1 * bz_, t1z = -a0 * bz_ + a3 * bw - a1 * by_ + a2 * bx_;
        const float t2x = -b0 * ax + b1 * aw - b2 * az + b3 * ay, t2y = -b0 * ay + b2 * aw - b3 * ax + b1 * az, t2z = -b0 * az + b3 * aw - b1 * ay + b2 * ax;
        const float rr = (aw * bw + ax * bx_) + (ay * by_ + az * bz_), dd = (a0 * b0 + a1 * b1) + (a2 * b2 + a3 * b3);
        const float gx = 0.5f * (s1x + s2x) - (rr + dd) * cx + (t1x + t2x);
        const float gy = 0.5f * (s1y + s2y) - (rr + dd) * cy + (t1y + t2y);
        const float gz = 0.5f * (s1z + s2z) - (rr + dd) * cz + (t1z + t2z);
        const float lam = a[12] + b[12] + 2.0f * a[13] * b[13] + fabsf(dd);
        num = fmaxf(num, sqrtf(gx * gx + gy * gy + gz * gz) + lam * (float)DFH_BRICK_RADIUS);
        n2min = fminf(n2min, rr);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        num = fmaxf(num, __shfl_xor(num, o, 16));
        n2min = fminf(n2min, __shfl_xor(n2min, o, 16));
        bad = bad | (__shfl_xor(bad ? 1 : 0, o, 16) != 0);
    }
    float D = __builtin_huge_valf();
    int m = 255;
    if (want && !overflow && n >= 1 && !bad && n2min > 0.25f && num < 1e6f) {
        const double cn = sqrt((cxd * cxd + cyd * cyd) + czd * czd) + DFH_BRICK_RADIUS;
        double d = (double)num / ((double)n2min * (1.0 - 1e-5)) * (1.0 + 1e-5);
        d = d + (cn + d) * 0x1p-23 + cn * 4e-6 + 2e-3;                         // + float32 rounding of x1 + this kernel's and the paths' own error + slack
        D = (float)(d * (1.0 + 0x1p-22));                                      // (rounded up: the stored value is never below d)
        const double need = d + 1.0;                                           // the corners: floor / ceil of the position
        m = need <= 4.0 ? 1 : (need <= 8.0 ? 2 : 255);
    }
    const bool safe = m <= rs && all_bounds != 2;                              // (uniform over the 16 lanes; all_bounds = 2: experiment, every brick to the warp kernel)
    if (sub == 0 && real) {
        k.db[brick] = want ? D : -1.0f;                                        // -1: not computed (the live volume ruled the skip out)
        k.mb[brick] = (unsigned char)m;
        reinterpret_cast<unsigned char *>(k.S)[brick] = safe ? 1 : 0;
    }
    // the warp kernel's bricks of this WAVE (four of them, in brick order: z fastest, neighbours along z stay neighbours) onto
    // sub-list hash(block) behind one atomic
    // (not blockIdx % kSkipLists: with 64 columns per x plane that is "the columns at y = l", a plane, and the sphere's planes
    // hold anything from none to most of the shell -- 135 us instead of 100)
    const unsigned l = (blockIdx.x ^ (blockIdx.x >> 6) ^ (blockIdx.x >> 12) ^ (blockIdx.x >> 18)) % kSkipLists;
    const unsigned long long unsafe = __ballot(sub == 0 && real && !safe);      // bits 0, 16, 32, 48
    if (unsafe == 0ull) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == __builtin_ctzll(unsafe)) base = atomicAdd(k.sub_count + 16 * l, (unsigned)__builtin_popcountll(unsafe));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(unsafe));
    if (sub == 0 && real && !safe)
        k.sub_list[(size_t)l * k.sub_cap + base + (unsigned)__builtin_popcountll(unsafe & ((1ull << lane) - 1ull))] = (unsigned)brick;
}

// the bricks with Sb = 1: T, w, wi in 16-byte packs, s = tdist; a pack nothing changes in (T already at tdist, w saturated: most
// of the free space) is not written back
template <bool NT>
__global__ __launch_bounds__(256) void dqb_stream_kernel(float *__restrict__ tsdf, float *__restrict__ tsdf_w, const float *__restrict__ wi_cache,
                                                          const DqbParams p, const DqbSkip k, const DqbRedoList redo_list, unsigned npacks, int sat_ok) {
    const unsigned pack = blockIdx.x * 256u + threadIdx.x;
    if (pack >= npacks) return;
    const unsigned zp = (unsigned)p.Z >> 2;
    const unsigned row = pack / zp, cz = pack - row * zp;
    const unsigned xl = row / (unsigned)p.Y, y = row - xl * (unsigned)p.Y;
    if (!reinterpret_cast<const unsigned char *>(k.S)[((size_t)(xl >> 2) * p.nby + (y >> 2)) * p.nbz + (cz >> 2)]) return;
    const size_t off = (size_t)pack * 4;
    // (NT: volumes beyond the Infinity Cache -- whole 1-KiB rows per wave instruction, nothing re-read: K1's non-temporal policy)
    typedef float v4f __attribute__((ext_vector_type(4)));
    float4 t0, w0;
    if (NT) {
        const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(tsdf + off)), b = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(tsdf_w + off));
        t0 = make_float4(a.x, a.y, a.z, a.w); w0 = make_float4(b.x, b.y, b.z, b.w);
    } else {
        t0 = *reinterpret_cast<const float4 *>(tsdf + off);
        w0 = *reinterpret_cast<const float4 *>(tsdf_w + off);
    }
    // T already at tdist and w at wmax (the free space after a few frames): the update changes nothing whatever wi >= 0 is,
    // PROVIDED tdist is a power of two (the frame loop's 4 voxels; sat_ok): then tdist wi and tdist wt are exact, their sum is
    // tdist (wi + wt) exactly (two float32 values of similar size add exactly in fp64), times the reciprocal, good to an ulp of
    // the double, rounds to the float32 tdist; and wt = wmax gives min(wi + wmax, wmax) = wmax.  No wi load, no store.
    const float td = (float)p.tdist, wm = (float)p.wmax;
    if (sat_ok && t0.x == td && t0.y == td && t0.z == td && t0.w == td && w0.x == wm && w0.y == wm && w0.z == wm && w0.w == wm) return;
    float4 wi;
    if (NT) { const v4f c = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(wi_cache + off)); wi = make_float4(c.x, c.y, c.z, c.w); }
    else wi = *reinterpret_cast<const float4 *>(wi_cache + off);
    float4 t = t0, w = w0;
    float *tv = &t.x, *wv = &w.x;
    const float *wiv = &wi.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (wiv[j] >= 0.0f) {
            dqb_update_f32(tv[j], wv[j], wiv[j], p.tdist, p.wmax, tv[j], wv[j]);
        } else {                                   // no stored weight (the blend weights did not normalise): the exact chain's turn
            redo_list.offs[atomicAdd(redo_list.count, 1u)] = (unsigned)(off + j);
        }
    }
    if (t.x != t0.x || t.y != t0.y || t.z != t0.z || t.w != t0.w) {
        if (NT) { v4f a; a.x = t.x; a.y = t.y; a.z = t.z; a.w = t.w; __builtin_nontemporal_store(a, reinterpret_cast<v4f *>(tsdf + off)); }
        else *reinterpret_cast<float4 *>(tsdf + off) = t;
    }
    if (w.x != w0.x || w.y != w0.y || w.z != w0.z || w.w != w0.w) {
        if (NT) { v4f a; a.x = w.x; a.y = w.y; a.z = w.z; a.w = w.w; __builtin_nontemporal_store(a, reinterpret_cast<v4f *>(tsdf_w + off)); }
        else *reinterpret_cast<float4 *>(tsdf_w + off) = w;
    }
}

// Regions of the skip inside a level-2 workspace: behind the fast path's 28 B per voxel and the redo list's 4 B + header, in the
// 8 B per voxel the fp64-weight layout leaves unused.  Returns false when they do not fit (tiny slabs of a large live volume).
static bool skip_layout(const DqbParams &p, double *w_cache, DqbSkip &k, size_t *used = nullptr) {
    const size_t nv = (size_t)p.nx * p.Y * p.Z;
    k.CX = (p.LX + 3) / 4; k.CY = (p.LY + 3) / 4; k.CZ = p.LZ / 4; k.WZ = (k.CZ + 63) / 64;
    k.SCX = p.nx / 4; k.SCY = (p.Y + 3) / 4;
    const long nbricks = (long)p.nbx * p.nby * p.nbz;
    char *base = reinterpret_cast<char *>(w_cache);
    size_t at = (nv * 32 + 16 + 63) & ~(size_t)63;
    auto take = [&](size_t bytes) { char *q = base ? base + at : nullptr; at = (at + bytes + 63) & ~(size_t)63; return q; };
    k.used = reinterpret_cast<unsigned short *>(take((size_t)nbricks * kSkipUsed * 2));        // (static while the graph stays: first)
    k.U = reinterpret_cast<unsigned long long *>(take((size_t)k.CX * k.CY * k.WZ * 8));
    k.S = reinterpret_cast<unsigned long long *>(take((size_t)nbricks));                        // (one byte per brick)
    k.mb = reinterpret_cast<unsigned char *>(take((size_t)nbricks));
    // blocks of 16 bricks, dealt by a hash of the block index: a bijection on every aligned group of kSkipLists blocks (the low
    // six bits are xor-ed with bits that are constant inside the group), so no list gets more than this
    k.sub_cap = (unsigned)((((nbricks + 15) / 16 + kSkipLists - 1) / kSkipLists) * 16);
    k.sub_count = reinterpret_cast<unsigned *>(take((size_t)kSkipLists * 64));
    k.sub_list = reinterpret_cast<unsigned *>(take((size_t)kSkipLists * k.sub_cap * 4));
    k.db = reinterpret_cast<float *>(take((size_t)nbricks * 4));
    if (used) *used = at;
    return at <= nv * 40;
}

// sizes for which the skip's tables exist at all (what the store pass and the steady state both ask)
static bool skip_sizes_ok(const DqbParams &p) {
    const size_t nv = (size_t)p.nx * p.Y * p.Z;
    return p.x0 % 4 == 0 && p.nx % 4 == 0 && p.LZ % 4 == 0 && p.LZ <= 64 * 4 * kSkipMaxWords && p.Z % 64 == 0 && p.Z <= 2048 &&
           nv / 4 < (1ull << 32);
}

template <typename LiveT>
static int launch_dqb_fast(void *tsdf, void *tsdf_w, const void *live, const double *node_pos, const double *node_dq,
                           const double *node_w, int *cand, unsigned short *knn_cache, double *w_cache, int mode, const DqbParams &p,
                           hipStream_t s) {
    RigidFastParams f;
    fold_rigid(p.lw.q, f);
    // the redo list of the steady-state kernel sits in the cache's spare room: the fast path uses 28 of the 40 B per voxel of
    // a level-2 workspace (three fp64 planes, one float32 plane), the list one unsigned per voxel behind them + 16 B of header
    DqbRedoList redo_list{nullptr, nullptr};
    if (w_cache) {
        const size_t nv = (size_t)p.nx * p.Y * p.Z;
        unsigned *base = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(w_cache) + nv * 28);
        redo_list.count = base;
        redo_list.offs = base + 4;
        if (mode == 1) DFH_HIP_CHECK(hipMemsetAsync(base, 0, 16, s));        // (rebuild: the header starts clean)
    }
    const long nbricks = (long)p.nbx * p.nby * p.nbz;
    const long nblocks = mode >= 2 ? ((long)p.nx * p.Y * p.Z + 255) / 256 : nbricks;
#define DFH_K3F(MODE) hipLaunchKernelGGL((fuse_volume_dqb_fast_kernel<LiveT, MODE>), dim3((unsigned)nblocks), dim3(256), 0, s, (float *)tsdf, \
                                         (float *)tsdf_w, (const LiveT *)live, node_pos, node_dq, node_w, cand, knn_cache, w_cache, p, f)
    const long n_runs = (long)p.nx * p.Y * (p.Z / 64);
    const bool big = (size_t)p.N * kDqLdsStride * sizeof(double) > 64 * 1024;     // (64 KB: what a kernel may ask for without an attribute)
    const size_t lds = (size_t)p.N * (big ? kDqLdsStrideBig : kDqLdsStride) * sizeof(double);
    bool lds_path = mode == 3 && p.Z % 64 == 0 && n_runs < (1L << 25) && lds <= 144 * 1024 && !on(opt().k3_no_lds);
    if (lds_path && big) {
        // more than 64 KB of dynamic LDS has to be asked for, once per kernel (the attribute sticks to the function); a device that
        // refuses takes the gather kernel below
        static std::once_flag asked;
        static hipError_t asked_rc = hipSuccess;
        std::call_once(asked, [] {
            asked_rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&fuse_volume_dqb_lds_kernel<LiveT, 1024, kDqLdsStrideBig, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (asked_rc == hipSuccess)
                asked_rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&fuse_volume_dqb_lds_kernel<LiveT, 1024, kDqLdsStrideBig, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (asked_rc != hipSuccess) (void)hipGetLastError();
        });
        lds_path = asked_rc == hipSuccess;
    }
    if (lds_path) {
        // persistent grid: as many workgroups as fit the chip with this much LDS (160 KB per CU)
        int dev = 0;
        DFH_HIP_CHECK(hipGetDevice(&dev));
        DeviceInfo &di = device_info(dev);
        if (di.n_cu == 0) DFH_HIP_CHECK(hipDeviceGetAttribute(&di.n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        // workgroups of 1024 threads: sixteen waves share one copy of the table (measured 248 us against 263 for 512 / 256)
        const int tpb = (opt().k3_tpb == 256 || opt().k3_tpb == 512) && !big ? (int)opt().k3_tpb : 1024;
        long per_cu = (long)(160 * 1024 / (lds + 512));
        const long max_per_cu = 2048 / tpb;
        if (per_cu > max_per_cu) per_cu = max_per_cu;
        if (opt().k3_wg_per_cu > 0 && opt().k3_wg_per_cu < per_cu) per_cu = opt().k3_wg_per_cu;
        long wgs = per_cu * di.n_cu;
        const int wpb = tpb / 64;
        if (wgs * wpb > n_runs) wgs = (n_runs + wpb - 1) / wpb;
        // constant-live skip (above): steady state, m_lw the identity, slab and live volume cut into whole 4-cells, tdist a float32
        DqbSkip sk = {};
        const size_t nv = (size_t)p.nx * p.Y * p.Z;
        // k3_skip: 0 off, 1 on, 2 on + every brick's bound computed (tests), unset: on for grids beyond 256^3 -- measured on the
        // bench scenes (tools/k3_skip_probe.py): 512^3 / 2 048 nodes 1.91-1.96 -> 1.20-1.25 ms (72 % of the bricks skip), 256^3 /
        // 512 nodes 235 -> 232-250 us (61 % skip; the pre-passes and the warp kernel's poorer coalescing on 16-voxel rows eat it)
        const bool skip_wanted = opt().k3_skip > 0 || (opt().k3_skip < 0 && (long)p.X * p.Y * p.Z > (1L << 24));
        bool skip = skip_wanted && p.tdist > 0.0 && (double)(float)p.tdist == p.tdist && skip_sizes_ok(p) &&
                    (uintptr_t)live % (4 * sizeof(LiveT)) == 0 && (uintptr_t)tsdf % 16 == 0 && (uintptr_t)tsdf_w % 16 == 0 &&
                    p.lw.q[0] == 1.0 && p.lw.q[1] == 0.0 && p.lw.q[2] == 0.0 && p.lw.q[3] == 0.0 && p.lw.q[4] == 0.0 && p.lw.q[5] == 0.0 &&
                    p.lw.q[6] == 0.0 && p.lw.q[7] == 0.0;
        if (skip) skip = skip_layout(p, w_cache, sk);
        const float *wi_cache = reinterpret_cast<const float *>(w_cache + 3 * nv);
        if (skip) {
            const long nbr = (long)p.nbx * p.nby * p.nbz;
            int ex = 0;
            // (the stream's "nothing changes" shortcut: exact only for a power-of-two tdist -- see dqb_stream_kernel)
            const int sat_ok = frexp(p.tdist, &ex) == 0.5 && p.wmax > 0.0 && (double)(float)p.wmax == p.wmax ? 1 : 0;
            const int cx_lo = std::max(0, p.x0 / 4 - 2), cx_hi = std::min(sk.CX, (p.x0 + p.nx) / 4 + 2);       // reach <= 2 cells
            if (cx_hi > cx_lo)
                hipLaunchKernelGGL((dqb_live_mask_kernel<LiveT>), dim3((unsigned)((cx_hi - cx_lo) * sk.CY)), dim3(1024), 0, s, (const LiveT *)live, p.LX,
                                   p.LY, p.LZ, p.tdist, sk, cx_lo);
            hipLaunchKernelGGL(dqb_reach_kernel, dim3((unsigned)((sk.SCX * sk.SCY + 3) / 4)), dim3(256), 0, s, p, sk);
            hipLaunchKernelGGL(dqb_bound_kernel, dim3((unsigned)((nbr + 15) / 16)), dim3(256), 0, s, node_dq, p, sk, opt().k3_skip == 2 ? 1 : (opt().k3_skip == 3 ? 2 : 0));
            if (nv * 8 > ((size_t)256 << 20))
                hipLaunchKernelGGL(dqb_stream_kernel<true>, dim3((unsigned)((nv / 4 + 255) / 256)), dim3(256), 0, s, (float *)tsdf, (float *)tsdf_w, wi_cache,
                                   p, sk, redo_list, (unsigned)(nv / 4), sat_ok);
            else
                hipLaunchKernelGGL(dqb_stream_kernel<false>, dim3((unsigned)((nv / 4 + 255) / 256)), dim3(256), 0, s, (float *)tsdf, (float *)tsdf_w, wi_cache,
                                   p, sk, redo_list, (unsigned)(nv / 4), sat_ok);
        }
#define DFH_K3L(TPB, STRIDE)                                                                                                                          \
        do {                                                                                                                                          \
            if (skip) hipLaunchKernelGGL((fuse_volume_dqb_lds_kernel<LiveT, TPB, STRIDE, true>), dim3((unsigned)wgs), dim3(TPB), lds, s, (float *)tsdf,  \
                                         (float *)tsdf_w, (const LiveT *)live, node_pos, node_dq, node_w, knn_cache, w_cache, p, f, (int)n_runs,        \
                                         redo_list, (const unsigned *)sk.sub_count, (const unsigned *)sk.sub_list, sk.sub_cap);                       \
            else hipLaunchKernelGGL((fuse_volume_dqb_lds_kernel<LiveT, TPB, STRIDE, false>), dim3((unsigned)wgs), dim3(TPB), lds, s, (float *)tsdf,     \
                                    (float *)tsdf_w, (const LiveT *)live, node_pos, node_dq, node_w, knn_cache, w_cache, p, f, (int)n_runs, redo_list, \
                                    (const unsigned *)nullptr, (const unsigned *)nullptr, 0u);                                                       \
        } while (0)
        if (big) DFH_K3L(1024, kDqLdsStrideBig);
        else if (tpb == 256) DFH_K3L(256, kDqLdsStride); else if (tpb == 512) DFH_K3L(512, kDqLdsStride); else DFH_K3L(1024, kDqLdsStride);
        // ... and right behind it the voxels it put on its redo list, through the exact chain
        hipLaunchKernelGGL((dqb_redo_kernel<LiveT>), dim3(kRedoBlocks), dim3(256), 0, s, (float *)tsdf, (float *)tsdf_w, (const LiveT *)live, node_pos, node_dq,
                           node_w, knn_cache, reinterpret_cast<const float *>(w_cache + 3 * ((size_t)p.nx * p.Y * p.Z)), p, redo_list);
#undef DFH_K3L
    } else if (mode == 0) DFH_K3F(0); else if (mode == 1) DFH_K3F(1); else if (mode == 2) DFH_K3F(2); else DFH_K3F(3);
#undef DFH_K3F
    if (mode == 1 && w_cache && skip_sizes_ok(p)) {
        // the store pass also leaves, per brick, the nodes its voxels blend: what the constant-live skip's bound runs over
        DqbSkip sk = {};
        if (skip_layout(p, w_cache, sk))
            hipLaunchKernelGGL(dqb_used_nodes_kernel, dim3((unsigned)nbricks), dim3(256), 0, s, (const unsigned short *)knn_cache, (const int *)cand, p, sk);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

template <typename VolT, typename LiveT>
static int launch_dqb(void *tsdf, void *tsdf_w, const void *live, const double *node_pos, const double *node_dq,
                      const double *node_w, int *cand, unsigned short *knn_cache, double *w_cache, int mode, const DqbParams &p,
                      hipStream_t s) {
    const long nbricks = (long)p.nbx * p.nby * p.nbz;
    const long nblocks = mode >= 2 ? ((long)p.nx * p.Y * p.Z + 255) / 256 : nbricks;
#define DFH_K3(KS, MODE) hipLaunchKernelGGL((fuse_volume_dqb_kernel<VolT, LiveT, KS, MODE>), dim3((unsigned)nblocks), dim3(256), 0, s, \
                                            (VolT *)tsdf, (VolT *)tsdf_w, (const LiveT *)live, node_pos, node_dq, node_w, cand, knn_cache, w_cache, p)
    if (p.k <= 4) {                                   // 4 register slots suffice: half the insertion work
        if (mode == 0) DFH_K3(4, 0); else if (mode == 1) DFH_K3(4, 1); else if (mode == 2) DFH_K3(4, 2); else DFH_K3(4, 3);
    } else {
        if (mode == 0) DFH_K3(kKMax, 0); else if (mode == 1) DFH_K3(kKMax, 1); else if (mode == 2) DFH_K3(kKMax, 2); else DFH_K3(kKMax, 3);
    }
#undef DFH_K3
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

static size_t cand_bytes(const int res[3], int x0, int x1) {
    int nbx, nby, nbz;
    nbx = (x1 - x0 + kBX - 1) / kBX;
    nby = (res[1] + kBY - 1) / kBY;
    nbz = (res[2] + kBZ - 1) / kBZ;
    return (((size_t)nbx * nby * nbz * (kCap + 1) * sizeof(int)) + 15) & ~(size_t)15;
}

// k nearest nodes + blend weights of arbitrary points through the bricks' candidate lists (what sample_knn_kernel of
// dfh_solve.hip computes, same expressions, same tie order: candidates are kept in node order and the insertion is
// stable).  A point outside the slab's lattice, or in a brick whose list overflowed, scans every node.
template <int KS>
__global__ __launch_bounds__(256) void sample_knn_bricks_kernel(const double *__restrict__ spos, int S,
                                                                 const double *__restrict__ node_pos,
                                                                 const double *__restrict__ node_w,
                                                                 const int *__restrict__ cand, const DqbParams p,
                                                                 int *__restrict__ nbr, double *__restrict__ wts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const double px = spos[3 * (size_t)i], py = spos[3 * (size_t)i + 1], pz = spos[3 * (size_t)i + 2];
    const double rx = rint(px), ry = rint(py), rz = rint(pz);
    const bool inside = rx >= (double)p.x0 && rx < (double)(p.x0 + p.nx) && ry >= 0.0 && ry < (double)p.Y && rz >= 0.0 &&
                        rz < (double)p.Z;                                        // (false for NaN)
    const int *c = nullptr;
    int cnt = -1;
    if (inside) {
        const long brick = ((long)(((int)rx - p.x0) / kBX) * p.nby + (int)ry / kBY) * p.nbz + (int)rz / kBZ;
        c = cand + brick * (kCap + 1);
        cnt = c[0];
    }
    double bd[KS];
    int bi[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) { bd[j] = __builtin_huge_val(); bi[j] = -1; }
    const int total = cnt >= 0 ? cnt : p.N;
    constexpr int U = 8;                          // the loop is a chain of dependent loads (list entry -> node position):
    for (int j0 = 0; j0 < total; j0 += U) {       // issue U of them before the first use
        int g[U];
        double x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = min(j0 + u, total - 1);
            g[u] = cnt >= 0 ? c[1 + jj] : jj;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { x[u] = node_pos[3 * g[u]]; y[u] = node_pos[3 * g[u] + 1]; z[u] = node_pos[3 * g[u] + 2]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double dx = px - x[u], dy = py - y[u], dz = pz - z[u];
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            if (j0 + u < total && d2 < bd[KS - 1]) topk_insert<KS>(bd, bi, d2, g[u]);
        }
    }
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        if (j < p.k) {
            const int gi = bi[j];
            nbr[(size_t)i * p.k + j] = gi;
            const double t = sqrt(bd[j]) / (2.0 * node_w[gi]);
            wts[(size_t)i * p.k + j] = exp(-1.0 * (t * t));
        }
    }
}

static void brick_counts(const int res[3], int x0, int x1, int &nbx, int &nby, int &nbz) {
    nbx = (x1 - x0 + kBX - 1) / kBX;
    nby = (res[1] + kBY - 1) / kBY;
    nbz = (res[2] + kBZ - 1) / kBZ;
}

}  // namespace dfh

extern "C" int dfh_fuse_volume_rigid(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int x0, int x1,
                                     const void *live, int live_dtype, const int live_res[3],
                                     const double lw_dq[8], double tdist, double wmax, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(tsdf && tsdf_w && live && res && live_res && lw_dq, "dfh_fuse_volume_rigid: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_fuse_volume_rigid: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(live_dtype == DFH_F32 || live_dtype == DFH_F64, "dfh_fuse_volume_rigid: bad live_dtype %d", live_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0, "dfh_fuse_volume_rigid: bad grid");
    DFH_REQUIRE(live_res[0] > 0 && live_res[1] > 0 && live_res[2] > 0, "dfh_fuse_volume_rigid: bad live grid");
    DFH_REQUIRE((long)live_res[0] * live_res[1] * live_res[2] < (1L << 40), "dfh_fuse_volume_rigid: live grid too large");
    DFH_REQUIRE(0 <= x0 && x0 <= x1 && x1 <= res[0], "dfh_fuse_volume_rigid: slab [%d,%d) outside [0,%d)", x0, x1, res[0]);
    DFH_REQUIRE(x1 - x0 <= 65535, "dfh_fuse_volume_rigid: slab has more than 65535 planes");
    if (x1 == x0) return DFH_OK;
    RigidParams p;
    for (int i = 0; i < 8; ++i) p.lw.q[i] = lw_dq[i];
    p.tdist = tdist; p.wmax = wmax;
    p.X = res[0]; p.Y = res[1]; p.Z = res[2];
    p.LX = live_res[0]; p.LY = live_res[1]; p.LZ = live_res[2];
    p.x0 = x0; p.nx = x1 - x0;
    const size_t esz = vol_dtype == DFH_F32 ? 4 : 8;
    const bool vec4 = (res[2] % 4 == 0) && ((uintptr_t)tsdf % (4 * esz) == 0) && ((uintptr_t)tsdf_w % (4 * esz) == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vol_dtype == DFH_F32 && vec4 && !on(opt().k2_exact)) {
        RigidFastParams f;
        fold_rigid(lw_dq, f);
        p.zpacks = p.Z / 4;
        p.zp_shift = -1;
        for (int b = 0; b < 31; ++b) if (p.zpacks == (1 << b)) p.zp_shift = b;
        dim3 grid((unsigned)(((long)p.Y * p.zpacks + 255) / 256), (unsigned)p.nx), block(256);
        const bool strided = p.zpacks % 64 == 0 && !on(opt().k2_no_strided);
#define DFH_K2(LT, ST, NT_) hipLaunchKernelGGL((fuse_volume_rigid_fast_kernel<LT, ST, NT_>), grid, block, 0, s, (float *)tsdf, \
                                               (float *)tsdf_w, (const LT *)live, p, f)
        const bool nt = strided && (opt().k2_nt > 0 || (opt().k2_nt < 0 && (size_t)p.nx * p.Y * p.Z * 8 > ((size_t)256 << 20)));
        if (live_dtype == DFH_F32) { if (strided) { if (nt) DFH_K2(float, true, true); else DFH_K2(float, true, false); } else DFH_K2(float, false, false); }
        else { if (strided) { if (nt) DFH_K2(double, true, true); else DFH_K2(double, true, false); } else DFH_K2(double, false, false); }
#undef DFH_K2
        DFH_HIP_CHECK(hipGetLastError());
        return DFH_OK;
    }
    if (vol_dtype == DFH_F32) {
        if (live_dtype == DFH_F32) return launch_rigid<float, float>(tsdf, tsdf_w, live, p, vec4, s);
        return launch_rigid<float, double>(tsdf, tsdf_w, live, p, vec4, s);
    }
    if (live_dtype == DFH_F32) return launch_rigid<double, float>(tsdf, tsdf_w, live, p, vec4, s);
    return launch_rigid<double, double>(tsdf, tsdf_w, live, p, vec4, s);
}

extern "C" size_t dfh_dqb_workspace_bytes(const int res[3], int x0, int x1) {
    if (!res || x1 <= x0) return 0;
    return dfh::cand_bytes(res, x0, x1);
}

extern "C" size_t dfh_dqb_workspace_bytes_cached(const int res[3], int x0, int x1, int knn, int n_nodes, int level) {
    if (!res || x1 <= x0) return 0;
    const size_t base = dfh::cand_bytes(res, x0, x1);
    if (level < 1 || knn < 1 || knn > dfh::kKMax || n_nodes > 65536) return base;         // indices are kept as 16-bit
    const size_t nvox = (size_t)(x1 - x0) * res[1] * res[2];
    const size_t idx = (nvox * knn * sizeof(unsigned short) + 15) & ~(size_t)15;
    return base + idx + (level >= 2 ? nvox * (knn + 1) * sizeof(double) : 0);
}

extern "C" int dfh_fuse_volume_dqb(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int x0, int x1,
                                   const void *live, int live_dtype, const int live_res[3],
                                   const double *node_pos, const double *node_dq, const double *node_w, int n_nodes,
                                   int knn, const double lw_dq[8], double tdist, double wmax,
                                   void *workspace, size_t workspace_bytes, int rebuild_candidates, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(tsdf && tsdf_w && live && res && live_res && lw_dq && node_pos && node_dq && node_w,
                "dfh_fuse_volume_dqb: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_fuse_volume_dqb: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(live_dtype == DFH_F32 || live_dtype == DFH_F64, "dfh_fuse_volume_dqb: bad live_dtype %d", live_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0, "dfh_fuse_volume_dqb: bad grid");
    DFH_REQUIRE(live_res[0] > 0 && live_res[1] > 0 && live_res[2] > 0, "dfh_fuse_volume_dqb: bad live grid");
    DFH_REQUIRE(0 <= x0 && x0 <= x1 && x1 <= res[0], "dfh_fuse_volume_dqb: slab [%d,%d) outside [0,%d)", x0, x1, res[0]);
    DFH_REQUIRE(knn >= 1 && knn <= kKMax, "dfh_fuse_volume_dqb: knn=%d outside [1,%d]", knn, kKMax);
    DFH_REQUIRE(n_nodes >= knn, "dfh_fuse_volume_dqb: %d nodes < knn=%d", n_nodes, knn);
    if (x1 == x0) return DFH_OK;
    DFH_REQUIRE(workspace && workspace_bytes >= dfh_dqb_workspace_bytes(res, x0, x1),
                "dfh_fuse_volume_dqb: workspace too small (need %zu bytes)", dfh_dqb_workspace_bytes(res, x0, x1));
    DqbParams p;
    for (int i = 0; i < 8; ++i) p.lw.q[i] = lw_dq[i];
    p.tdist = tdist; p.wmax = wmax;
    p.X = res[0]; p.Y = res[1]; p.Z = res[2];
    p.LX = live_res[0]; p.LY = live_res[1]; p.LZ = live_res[2];
    p.x0 = x0; p.nx = x1 - x0; p.N = n_nodes; p.k = knn;
    brick_counts(res, x0, x1, p.nbx, p.nby, p.nbz);
    const long nbricks = (long)p.nbx * p.nby * p.nbz;
    DFH_REQUIRE(nbricks < (1L << 31), "dfh_fuse_volume_dqb: too many bricks");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int *cand = static_cast<int *>(workspace);
    // a workspace of dfh_dqb_workspace_bytes_cached() also keeps every voxel's k node indices
    const size_t base = cand_bytes(res, x0, x1);
    const size_t cached1 = dfh_dqb_workspace_bytes_cached(res, x0, x1, knn, n_nodes, 1);
    const size_t cached2 = dfh_dqb_workspace_bytes_cached(res, x0, x1, knn, n_nodes, 2);
    const bool has_idx = cached1 > base && workspace_bytes >= cached1 && !on(opt().k3_no_cache);
    const bool has_w = has_idx && workspace_bytes >= cached2;
    unsigned short *knn_cache = has_idx ? reinterpret_cast<unsigned short *>(static_cast<char *>(workspace) + base) : nullptr;
    double *w_cache = has_w ? reinterpret_cast<double *>(static_cast<char *>(workspace) + cached1) : nullptr;
    int mode = !has_idx ? 0 : (rebuild_candidates ? 1 : (has_w ? 3 : 2));
    // float32 volumes with knn = 4 take the fast path; its cache (28 B per voxel) lives where the exact path keeps its fp64
    // weights, so a workspace remembers which of the two wrote it: stored weights of the other kind are not used (the stored
    // indices serve both; the weights are then recomputed, mode 2)
    const bool fast = vol_dtype == DFH_F32 && knn == 4 && !on(opt().k3_exact);
    if (has_w) {
        static std::mutex mu;
        static std::unordered_map<const void *, int> format;           // workspace -> 1 (fast entries) / 2 (fp64 weights)
        std::lock_guard<std::mutex> lock(mu);
        if (mode == 1) {
            if (format.size() > 4096) format.clear();
            format[workspace] = fast ? 1 : 2;
        } else if (mode == 3) {
            const auto it = format.find(workspace);
            if (it == format.end() || it->second != (fast ? 1 : 2)) mode = 2;
        }
    }
    if (rebuild_candidates) {
        hipLaunchKernelGGL(dqb_candidates_kernel, dim3((unsigned)((nbricks + 255) / 256)), dim3(256), 0, s, node_pos, cand, p);
        DFH_HIP_CHECK(hipGetLastError());
    }
    if (fast) {
        if (live_dtype == DFH_F32) return launch_dqb_fast<float>(tsdf, tsdf_w, live, node_pos, node_dq, node_w, cand, knn_cache, w_cache, mode, p, s);
        return launch_dqb_fast<double>(tsdf, tsdf_w, live, node_pos, node_dq, node_w, cand, knn_cache, w_cache, mode, p, s);
    }
    if (vol_dtype == DFH_F32) {
        if (live_dtype == DFH_F32) return launch_dqb<float, float>(tsdf, tsdf_w, live, node_pos, node_dq, node_w, cand, knn_cache, w_cache, mode, p, s);
        return launch_dqb<float, double>(tsdf, tsdf_w, live, node_pos, node_dq, node_w, cand, knn_cache, w_cache, mode, p, s);
    }
    if (live_dtype == DFH_F32) return launch_dqb<double, float>(tsdf, tsdf_w, live, node_pos, node_dq, node_w, cand, knn_cache, w_cache, mode, p, s);
    return launch_dqb<double, double>(tsdf, tsdf_w, live, node_pos, node_dq, node_w, cand, knn_cache, w_cache, mode, p, s);
}

extern "C" int dfh_dqb_skip_layout(const int res[3], int x0, int x1, const int live_res[3], int knn, int n_nodes, size_t out[13]) {
    using namespace dfh;
    DFH_REQUIRE(res && live_res && out, "dfh_dqb_skip_layout: null pointer");
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && 0 <= x0 && x0 < x1 && x1 <= res[0], "dfh_dqb_skip_layout: bad grid / slab");
    DFH_REQUIRE(knn == 4, "dfh_dqb_skip_layout: the skip belongs to the knn = 4 float32 path");
    DqbParams p = {};
    p.X = res[0]; p.Y = res[1]; p.Z = res[2];
    p.LX = live_res[0]; p.LY = live_res[1]; p.LZ = live_res[2];
    p.x0 = x0; p.nx = x1 - x0; p.N = n_nodes; p.k = knn;
    brick_counts(res, x0, x1, p.nbx, p.nby, p.nbz);
    const size_t cached1 = dfh_dqb_workspace_bytes_cached(res, x0, x1, knn, n_nodes, 1);
    const size_t cached2 = dfh_dqb_workspace_bytes_cached(res, x0, x1, knn, n_nodes, 2);
    DFH_REQUIRE(cached2 > cached1, "dfh_dqb_skip_layout: no level-2 workspace for these sizes");
    DqbSkip k = {};
    size_t used = 0;
    const bool fits = skip_layout(p, reinterpret_cast<double *>(cached1), k, &used);        // (pointer arithmetic on offsets only)
    out[0] = reinterpret_cast<size_t>(k.U); out[1] = reinterpret_cast<size_t>(k.S); out[2] = reinterpret_cast<size_t>(k.mb);
    out[3] = reinterpret_cast<size_t>(k.db); out[4] = 0; out[5] = 0;
    out[6] = (size_t)k.CX; out[7] = (size_t)k.CY; out[8] = (size_t)k.WZ; out[9] = (size_t)k.SCX; out[10] = (size_t)k.SCY;
    out[11] = fits && skip_sizes_ok(p) ? 1 : 0;
    out[12] = reinterpret_cast<size_t>(k.used);
    return DFH_OK;
}

extern "C" int dfh_dqb_build_candidates(const int res[3], int x0, int x1, const double *node_pos, int n_nodes, int knn,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(res && node_pos, "dfh_dqb_build_candidates: null pointer");
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0, "dfh_dqb_build_candidates: bad grid");
    DFH_REQUIRE(0 <= x0 && x0 <= x1 && x1 <= res[0], "dfh_dqb_build_candidates: slab [%d,%d) outside [0,%d)", x0, x1, res[0]);
    DFH_REQUIRE(knn >= 1 && knn <= kKMax && n_nodes >= knn, "dfh_dqb_build_candidates: knn=%d, %d nodes", knn, n_nodes);
    if (x1 == x0) return DFH_OK;
    DFH_REQUIRE(workspace && workspace_bytes >= dfh_dqb_workspace_bytes(res, x0, x1),
                "dfh_dqb_build_candidates: workspace too small (need %zu bytes)", dfh_dqb_workspace_bytes(res, x0, x1));
    DqbParams p = {};
    p.X = res[0]; p.Y = res[1]; p.Z = res[2];
    p.x0 = x0; p.nx = x1 - x0; p.N = n_nodes; p.k = knn;
    brick_counts(res, x0, x1, p.nbx, p.nby, p.nbz);
    const long nbricks = (long)p.nbx * p.nby * p.nbz;
    DFH_REQUIRE(nbricks < (1L << 31), "dfh_dqb_build_candidates: too many bricks");
    hipLaunchKernelGGL(dqb_candidates_kernel, dim3((unsigned)((nbricks + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       node_pos, static_cast<int *>(workspace), p);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

extern "C" int dfh_sample_knn_bricks(const double *sample_pos, int n_samples, const double *node_pos, const double *node_w,
                                     int n_nodes, int knn, const int res[3], int x0, int x1, const void *workspace,
                                     size_t workspace_bytes, int *nbr_out, double *weights_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_samples >= 0, "dfh_sample_knn_bricks: negative sample count");
    if (n_samples == 0) return DFH_OK;
    DFH_REQUIRE(sample_pos && node_pos && node_w && res && nbr_out && weights_out, "dfh_sample_knn_bricks: null pointer");
    DFH_REQUIRE(knn >= 1 && knn <= kKMax && n_nodes >= knn, "dfh_sample_knn_bricks: knn=%d, %d nodes", knn, n_nodes);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && 0 <= x0 && x0 < x1 && x1 <= res[0], "dfh_sample_knn_bricks: bad grid / slab");
    DFH_REQUIRE(workspace && workspace_bytes >= dfh_dqb_workspace_bytes(res, x0, x1),
                "dfh_sample_knn_bricks: workspace too small (need %zu bytes)", dfh_dqb_workspace_bytes(res, x0, x1));
    DqbParams p = {};
    p.X = res[0]; p.Y = res[1]; p.Z = res[2];
    p.x0 = x0; p.nx = x1 - x0; p.N = n_nodes; p.k = knn;
    brick_counts(res, x0, x1, p.nbx, p.nby, p.nbz);
    dim3 grid((unsigned)((n_samples + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (knn <= 4)
        hipLaunchKernelGGL(sample_knn_bricks_kernel<4>, grid, block, 0, s, sample_pos, n_samples, node_pos, node_w,
                           static_cast<const int *>(workspace), p, nbr_out, weights_out);
    else
        hipLaunchKernelGGL(sample_knn_bricks_kernel<kKMax>, grid, block, 0, s, sample_pos, n_samples, node_pos, node_w,
                           static_cast<const int *>(workspace), p, nbr_out, weights_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}
