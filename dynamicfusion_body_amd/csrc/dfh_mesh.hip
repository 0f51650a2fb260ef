// Marching cubes on the device: `_vertices`, `_faces`, `_normals` from a TSDF volume, the call the
// reference makes through skimage (measure.marching_cubes_lewiner at core/fusion_dm.py:319-331,342
// and core/fusion.py:554-568).  skimage's Lewiner tables are a third-party
// dependency that is absent here, so the triangle table is derived by tools/gen_mc_table.py (same
// topology rules for every cube, watertight, consistently oriented); what IS pinned by the
// reference's own output mesh (meshes/original.obj -> tests/golden/g9_mesh.npz): vertices on lattice
// edges at the linearly interpolated crossing in array-index coordinates, unit normals pointing down
// the gradient, faces wound with their right-hand normal up the gradient, zero-area faces dropped
// (allow_degenerate=False).  oracle/mc_np.py states the same computation in numpy.
//
// Layout: one thread per lattice point p = (x, y, z) of the (step-subsampled) volume, z fastest,
// 256 consecutive points per workgroup.  Point p owns the three edges leaving it along +axis 0/1/2
// and the cube whose corner 0 it is.  Four launches, no atomics, deterministic output order
// (vertices by owner point then axis; faces by cube then table order):
//   mc_count_kernel   per-workgroup {vertices, faces}
//   mc_scan_kernel    exclusive scan of those pairs (one workgroup), totals
//   mc_vertex_kernel  code[p] = crossing mask << 29 | index of p's first vertex; positions, normals
//   mc_face_kernel    faces, looking up the owners' codes
#include "dfh_common.h"
#include "dfh_mc_table.h"

namespace dfh {

constexpr int kMcBlock = 256;
constexpr unsigned kMcBaseMask = (1u << 29) - 1u;

struct McParams {
    int Y, Z;                 // strides of the full volume (elements): x*Y*Z + y*Z + z
    int s;                    // step_size
    int NX, NY, NZ;           // lattice dims = ceil(dim / s)
    double level;
    long npts;
};

template <typename VolT>
__device__ __forceinline__ double mc_val(const VolT *__restrict__ vol, const McParams &p, int x, int y, int z) {
    return (double)vol[((size_t)(x * p.s) * p.Y + (size_t)(y * p.s)) * p.Z + (size_t)(z * p.s)];
}

struct McPoint {
    int x, y, z;
    bool inside;              // p < npts
    double f[8];              // corner values (only those inside the lattice are meaningful)
    unsigned above;           // bit c: corner c exists and value > level
    unsigned eq;              // bit c: corner c exists and value == level
    unsigned cross;           // bit a: owned edge along axis a exists and is crossed
    bool cell;                // the cube with corner 0 = p exists
};

template <typename VolT>
__device__ __forceinline__ void mc_load(const VolT *__restrict__ vol, const McParams &p, long idx, McPoint &q) {
    q.inside = idx < p.npts;
    q.above = q.eq = q.cross = 0u;
    q.cell = false;
    q.x = q.y = q.z = 0;
    if (!q.inside) return;
    q.z = (int)(idx % p.NZ);
    q.y = (int)((idx / p.NZ) % p.NY);
    q.x = (int)(idx / ((long)p.NZ * p.NY));
    const bool hx = q.x + 1 < p.NX, hy = q.y + 1 < p.NY, hz = q.z + 1 < p.NZ;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ox = c & 1, oy = (c >> 1) & 1, oz = (c >> 2) & 1;
        const bool have = (!ox || hx) && (!oy || hy) && (!oz || hz);
        q.f[c] = have ? mc_val(vol, p, q.x + ox, q.y + oy, q.z + oz) : 0.0;
        if (have && q.f[c] > p.level) q.above |= 1u << c;
        if (have && q.f[c] == p.level) q.eq |= 1u << c;
    }
    const unsigned a0 = q.above & 1u;
    if (hx && (((q.above >> 1) & 1u) != a0)) q.cross |= 1u;
    if (hy && (((q.above >> 2) & 1u) != a0)) q.cross |= 2u;
    if (hz && (((q.above >> 4) & 1u) != a0)) q.cross |= 4u;
    q.cell = hx && hy && hz;
}

// edge e = 4*a + o1 + 2*o2  ->  axis a and the corner (offset bits) where the edge starts
__device__ __forceinline__ void mc_edge(int e, int &a, int &c0) {
    a = e >> 2;
    const int o1 = e & 1, o2 = (e >> 1) & 1;
    // the two other axes in increasing order: a=0 -> (1,2), a=1 -> (0,2), a=2 -> (0,1)
    const int u = a == 0 ? 1 : 0, v = a == 2 ? 1 : 2;
    c0 = (o1 << u) | (o2 << v);
}

// a triangle is dropped when two of its vertices sit on the same corner (value == level there)
__device__ __forceinline__ bool mc_keep(const signed char *row, int t, unsigned eq) {
    if (eq == 0u) return true;
    int col[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int a, c0;
        mc_edge(row[1 + 3 * t + k], a, c0);
        const int c1 = c0 | (1 << a);
        col[k] = ((eq >> c0) & 1u) ? c0 : (((eq >> c1) & 1u) ? c1 : -1 - k);
    }
    return col[0] != col[1] && col[1] != col[2] && col[0] != col[2];
}

__device__ __forceinline__ int mc_tri_count(const McPoint &q) {
    if (!q.cell || q.above == 0u || q.above == 255u) return 0;
    const signed char *row = kMcTable + kMcRow * (int)q.above;
    const int n = row[0];
    if (q.eq == 0u) return n;
    int kept = 0;
    for (int t = 0; t < n; ++t) kept += mc_keep(row, t, q.eq) ? 1 : 0;
    return kept;
}

// exclusive scan of one int per thread over the 256-thread workgroup; *total = workgroup sum
__device__ __forceinline__ int mc_block_scan(int v, int *lds /* 4 ints */, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kMcBlock / 64; ++w) {
        const int t = lds[w];
        before += w < wave ? t : 0;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return before + inc - v;
}

template <typename VolT>
__global__ __launch_bounds__(kMcBlock) void mc_count_kernel(const VolT *__restrict__ vol, const McParams p,
                                                             unsigned *__restrict__ counts) {
    __shared__ int lds[4];
    McPoint q;
    mc_load(vol, p, (long)blockIdx.x * kMcBlock + threadIdx.x, q);
    int tv, tf;
    mc_block_scan(__popc(q.cross), lds, &tv);
    mc_block_scan(mc_tri_count(q), lds, &tf);
    if (threadIdx.x == 0) { counts[2 * blockIdx.x] = (unsigned)tv; counts[2 * blockIdx.x + 1] = (unsigned)tf; }
}

// counts[2b], counts[2b+1] -> exclusive prefix sums (in place); totals[0..1] = sums
__global__ __launch_bounds__(1024) void mc_scan_kernel(unsigned *__restrict__ counts, long nblocks, long *__restrict__ totals) {
    __shared__ unsigned long long wsum[2][16];
    __shared__ unsigned long long carry[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) carry[threadIdx.x] = 0ull;
    __syncthreads();
    for (long base = 0; base < nblocks; base += 1024) {
        const long b = base + threadIdx.x;
        unsigned long long v[2], inc[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            v[k] = b < nblocks ? counts[2 * b + k] : 0u;
            inc[k] = v[k];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long t = __shfl_up(inc[k], o, 64);
                if (lane >= o) inc[k] += t;
            }
            if (lane == 63) wsum[k][wave] = inc[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            unsigned long long before = carry[k];
            for (int w = 0; w < wave; ++w) before += wsum[k][w];
            // exclusive prefixes must fit the 29-bit vertex index / 31-bit face index (checked on the host from the totals)
            if (b < nblocks) counts[2 * b + k] = (unsigned)(before + inc[k] - v[k]);
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned long long tot = carry[k];
                for (int w = 0; w < 16; ++w) tot += wsum[k][w];
                carry[k] = tot;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = (long)carry[0]; totals[1] = (long)carry[1]; }
}

// gradient of the lattice function at lattice point (x,y,z): central differences, one-sided on the faces
template <typename VolT>
__device__ __forceinline__ void mc_grad(const VolT *__restrict__ vol, const McParams &p, int x, int y, int z, double *g) {
    const int n[3] = {p.NX, p.NY, p.NZ};
    const int c[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int lo[3] = {x, y, z}, hi[3] = {x, y, z};
        double v = 0.0;
        if (n[a] > 1) {
            if (c[a] == 0) {
                hi[a] = 1;
                v = mc_val(vol, p, hi[0], hi[1], hi[2]) - mc_val(vol, p, lo[0], lo[1], lo[2]);
            } else if (c[a] == n[a] - 1) {
                lo[a] = c[a] - 1;
                v = mc_val(vol, p, hi[0], hi[1], hi[2]) - mc_val(vol, p, lo[0], lo[1], lo[2]);
            } else {
                lo[a] = c[a] - 1;
                hi[a] = c[a] + 1;
                v = (mc_val(vol, p, hi[0], hi[1], hi[2]) - mc_val(vol, p, lo[0], lo[1], lo[2])) * 0.5;
            }
        }
        g[a] = v;
    }
}

template <typename VolT>
__global__ __launch_bounds__(kMcBlock) void mc_vertex_kernel(const VolT *__restrict__ vol, const McParams p,
                                                              const unsigned *__restrict__ offsets, unsigned *__restrict__ code,
                                                              float *__restrict__ verts, float *__restrict__ normals,
                                                              float *__restrict__ values, long cap) {
    __shared__ int lds[4];
    const long idx = (long)blockIdx.x * kMcBlock + threadIdx.x;
    McPoint q;
    mc_load(vol, p, idx, q);
    int tot;
    const int rank = mc_block_scan(__popc(q.cross), lds, &tot);
    if (!q.inside) return;
    const unsigned base = offsets[2 * blockIdx.x] + (unsigned)rank;
    code[idx] = (q.cross << 29) | (base & kMcBaseMask);
    if (q.cross == 0u) return;
    double g0[3];
    mc_grad(vol, p, q.x, q.y, q.z, g0);
    unsigned k = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!((q.cross >> a) & 1u)) continue;
        const long vi = (long)base + k;
        ++k;
        if (vi >= cap) continue;
        const double f0 = q.f[0], f1 = q.f[1 << a];
        const double t = (p.level - f0) / (f1 - f0);
        double pos[3] = {(double)q.x, (double)q.y, (double)q.z};
        pos[a] = pos[a] + t;
        double g1[3];
        mc_grad(vol, p, q.x + (a == 0), q.y + (a == 1), q.z + (a == 2), g1);
        const double gx = g0[0] + t * (g1[0] - g0[0]), gy = g0[1] + t * (g1[1] - g0[1]), gz = g0[2] + t * (g1[2] - g0[2]);
        const double n2 = (gx * gx + gy * gy) + gz * gz;
        const double nrm = sqrt(n2);
        const double inv = nrm > 0.0 ? -1.0 / nrm : 0.0;
        verts[3 * vi + 0] = (float)(pos[0] * (double)p.s);
        verts[3 * vi + 1] = (float)(pos[1] * (double)p.s);
        verts[3 * vi + 2] = (float)(pos[2] * (double)p.s);
        normals[3 * vi + 0] = (float)(gx * inv);
        normals[3 * vi + 1] = (float)(gy * inv);
        normals[3 * vi + 2] = (float)(gz * inv);
        if (values) values[vi] = (float)(f0 > f1 ? f0 : f1);
    }
}

template <typename VolT>
__global__ __launch_bounds__(kMcBlock) void mc_face_kernel(const VolT *__restrict__ vol, const McParams p,
                                                            const unsigned *__restrict__ offsets, const unsigned *__restrict__ code,
                                                            int *__restrict__ faces, long cap) {
    __shared__ int lds[4];
    const long idx = (long)blockIdx.x * kMcBlock + threadIdx.x;
    McPoint q;
    mc_load(vol, p, idx, q);
    const int nt = mc_tri_count(q);
    int tot;
    const int rank = mc_block_scan(nt, lds, &tot);
    if (nt == 0) return;
    long fi = (long)offsets[2 * blockIdx.x + 1] + rank;
    const signed char *row = kMcTable + kMcRow * (int)q.above;
    const int n = row[0];
    for (int t = 0; t < n; ++t) {
        if (!mc_keep(row, t, q.eq)) continue;
        if (fi < cap) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int a, c0;
                mc_edge(row[1 + 3 * t + k], a, c0);
                const long owner = idx + (long)(c0 & 1) * p.NY * p.NZ + (long)((c0 >> 1) & 1) * p.NZ + (long)((c0 >> 2) & 1);
                const unsigned cd = code[owner];
                const unsigned mask = cd >> 29;
                faces[3 * fi + k] = (int)((cd & kMcBaseMask) + (unsigned)__popc(mask & ((1u << a) - 1u)));
            }
        }
        ++fi;
    }
}

// ---- reference vertex order ------------------------------------------------------------------------
// skimage numbers vertices as its faces create them (cube by cube) and flips the face rows afterwards,
// so in its output vertex ids increase with their first use when the rows are read right-to-left
// (checked on meshes/original.obj).  Same here: key of slot (f, k) = 3 f + (2 - k); first[v] = smallest
// key that uses v (atomicMin on integers: order-independent); the keys that are first uses are
// counted, scanned and become the new ids; vertices no face uses are dropped.
constexpr int kMcKeysPerBlock = 1024;

__global__ __launch_bounds__(256) void mc_first_use_kernel(const int *__restrict__ faces, long nslots, unsigned *__restrict__ first) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= nslots) return;
    const long f = s / 3;
    const int k = (int)(s - 3 * f);
    atomicMin(first + faces[s], (unsigned)(3 * f + (2 - k)));
}

__device__ __forceinline__ int mc_key_vertex(const int *__restrict__ faces, long key) {
    const long f = key / 3;
    const int pos = (int)(key - 3 * f);
    return faces[3 * f + (2 - pos)];
}

__global__ __launch_bounds__(256) void mc_rank_count_kernel(const int *__restrict__ faces, long nkeys,
                                                            const unsigned *__restrict__ first, unsigned *__restrict__ bsum) {
    __shared__ int lds[4];
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long key = (long)blockIdx.x * kMcKeysPerBlock + threadIdx.x * 4 + j;
        if (key < nkeys) c += first[mc_key_vertex(faces, key)] == (unsigned)key ? 1 : 0;
    }
    int tot;
    mc_block_scan(c, lds, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = (unsigned)tot;
}

// exclusive scan of n unsigned values in place (one workgroup), *total = sum
__global__ __launch_bounds__(1024) void mc_scan1_kernel(unsigned *__restrict__ a, long n, long *__restrict__ total) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (long base = 0; base < n; base += 1024) {
        const long b = base + threadIdx.x;
        const unsigned long long v = b < n ? a[b] : 0u;
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned long long before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (b < n) a[b] = (unsigned)(before + inc - v);
        __syncthreads();
        if (threadIdx.x == 1023) {
            unsigned long long tot = carry;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            carry = tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = (long)carry;
}

__global__ __launch_bounds__(256) void mc_rank_apply_kernel(const int *__restrict__ faces, long nkeys,
                                                            const unsigned *__restrict__ first, const unsigned *__restrict__ boff,
                                                            unsigned *__restrict__ newid) {
    __shared__ int lds[4];
    int flag[4], v[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long key = (long)blockIdx.x * kMcKeysPerBlock + threadIdx.x * 4 + j;
        flag[j] = 0;
        v[j] = 0;
        if (key < nkeys) {
            v[j] = mc_key_vertex(faces, key);
            flag[j] = first[v[j]] == (unsigned)key ? 1 : 0;
        }
        c += flag[j];
    }
    int tot;
    unsigned rank = boff[blockIdx.x] + (unsigned)mc_block_scan(c, lds, &tot);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (flag[j]) newid[v[j]] = rank++;
    }
}

__global__ __launch_bounds__(256) void mc_permute_kernel(const float *__restrict__ verts, const float *__restrict__ normals,
                                                         const float *__restrict__ values, const unsigned *__restrict__ first,
                                                         const unsigned *__restrict__ newid, long nv, float *__restrict__ verts_out,
                                                         float *__restrict__ normals_out, float *__restrict__ values_out) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv || first[v] == 0xFFFFFFFFu) return;
    const size_t d = newid[v];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        verts_out[3 * d + c] = verts[3 * v + c];
        normals_out[3 * d + c] = normals[3 * v + c];
    }
    if (values && values_out) values_out[d] = values[v];
}

__global__ __launch_bounds__(256) void mc_relabel_kernel(int *__restrict__ faces, long nslots, const unsigned *__restrict__ newid) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s < nslots) faces[s] = (int)newid[faces[s]];
}

static bool mc_params(const int res[3], int step, double level, McParams &p) {
    p.Y = res[1]; p.Z = res[2]; p.s = step; p.level = level;
    p.NX = (res[0] + step - 1) / step; p.NY = (res[1] + step - 1) / step; p.NZ = (res[2] + step - 1) / step;
    p.npts = (long)p.NX * p.NY * p.NZ;
    return true;
}

static void mc_workspace_layout(const McParams &p, size_t &code_bytes, size_t &counts_bytes, long &nblocks) {
    nblocks = (p.npts + kMcBlock - 1) / kMcBlock;
    code_bytes = ((size_t)p.npts * sizeof(unsigned) + 15) & ~(size_t)15;
    counts_bytes = ((size_t)nblocks * 2 * sizeof(unsigned) + 15) & ~(size_t)15;
}

}  // namespace dfh

extern "C" {

size_t dfh_mc_workspace_bytes(const int res[3], int step) {
    using namespace dfh;
    if (!res || res[0] <= 0 || res[1] <= 0 || res[2] <= 0 || step < 1) return 0;
    McParams p;
    mc_params(res, step, 0.0, p);
    size_t cb, nb;
    long nblocks;
    mc_workspace_layout(p, cb, nb, nblocks);
    return cb + nb;
}

int dfh_mc_count(const void *vol, int vol_dtype, const int res[3], int step, double level, void *workspace,
                 size_t workspace_bytes, long *totals_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(vol && res && workspace && totals_out, "dfh_mc_count: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_mc_count: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && step >= 1, "dfh_mc_count: bad grid / step");
    DFH_REQUIRE(level == level, "dfh_mc_count: level is NaN");
    DFH_REQUIRE(workspace_bytes >= dfh_mc_workspace_bytes(res, step), "dfh_mc_count: workspace too small");
    McParams p;
    mc_params(res, step, level, p);
    DFH_REQUIRE(p.npts < (1L << 31) * (long)kMcBlock, "dfh_mc_count: grid too large");
    size_t cb, nb;
    long nblocks;
    mc_workspace_layout(p, cb, nb, nblocks);
    unsigned *counts = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + cb);
    hipStream_t s = (hipStream_t)stream;
    if (vol_dtype == DFH_F32)
        hipLaunchKernelGGL(mc_count_kernel<float>, dim3((unsigned)nblocks), dim3(kMcBlock), 0, s, (const float *)vol, p, counts);
    else
        hipLaunchKernelGGL(mc_count_kernel<double>, dim3((unsigned)nblocks), dim3(kMcBlock), 0, s, (const double *)vol, p, counts);
    hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, s, counts, nblocks, totals_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_mc_emit(const void *vol, int vol_dtype, const int res[3], int step, double level, void *workspace,
                size_t workspace_bytes, float *verts, float *normals, float *values, int *faces, long cap_verts, long cap_faces,
                void *stream) {
    using namespace dfh;
    DFH_REQUIRE(vol && res && workspace, "dfh_mc_emit: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_mc_emit: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && step >= 1, "dfh_mc_emit: bad grid / step");
    DFH_REQUIRE(workspace_bytes >= dfh_mc_workspace_bytes(res, step), "dfh_mc_emit: workspace too small");
    DFH_REQUIRE(cap_verts >= 0 && cap_faces >= 0, "dfh_mc_emit: negative capacity");
    DFH_REQUIRE(cap_verts <= (long)kMcBaseMask, "dfh_mc_emit: more than 2^29-1 vertices");
    DFH_REQUIRE((cap_verts == 0 || (verts && normals)) && (cap_faces == 0 || faces), "dfh_mc_emit: null output");
    McParams p;
    mc_params(res, step, level, p);
    size_t cb, nb;
    long nblocks;
    mc_workspace_layout(p, cb, nb, nblocks);
    unsigned *code = static_cast<unsigned *>(workspace);
    unsigned *counts = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + cb);
    hipStream_t s = (hipStream_t)stream;
    if (vol_dtype == DFH_F32) {
        hipLaunchKernelGGL(mc_vertex_kernel<float>, dim3((unsigned)nblocks), dim3(kMcBlock), 0, s, (const float *)vol, p, counts, code,
                           verts, normals, values, cap_verts);
        hipLaunchKernelGGL(mc_face_kernel<float>, dim3((unsigned)nblocks), dim3(kMcBlock), 0, s, (const float *)vol, p, counts, code,
                           faces, cap_faces);
    } else {
        hipLaunchKernelGGL(mc_vertex_kernel<double>, dim3((unsigned)nblocks), dim3(kMcBlock), 0, s, (const double *)vol, p, counts,
                           code, verts, normals, values, cap_verts);
        hipLaunchKernelGGL(mc_face_kernel<double>, dim3((unsigned)nblocks), dim3(kMcBlock), 0, s, (const double *)vol, p, counts, code,
                           faces, cap_faces);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

size_t dfh_mc_reorder_workspace_bytes(long n_verts, long n_faces) {
    if (n_verts < 0 || n_faces < 0) return 0;
    const size_t nb = (size_t)((3 * n_faces + dfh::kMcKeysPerBlock - 1) / dfh::kMcKeysPerBlock);
    return ((2 * (size_t)n_verts + nb + 1) * sizeof(unsigned) + 15) & ~(size_t)15;
}

int dfh_mc_reorder(const float *verts_in, const float *normals_in, const float *values_in, int *faces, long n_verts, long n_faces,
                   float *verts_out, float *normals_out, float *values_out, long *used_out, void *workspace,
                   size_t workspace_bytes, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_verts >= 0 && n_faces >= 0 && n_verts < (1L << 31) && 3 * n_faces < (1L << 32) - 1, "dfh_mc_reorder: bad sizes");
    DFH_REQUIRE(used_out && workspace, "dfh_mc_reorder: null pointer");
    DFH_REQUIRE(workspace_bytes >= dfh_mc_reorder_workspace_bytes(n_verts, n_faces), "dfh_mc_reorder: workspace too small");
    DFH_REQUIRE(n_verts == 0 || (verts_in && normals_in && verts_out && normals_out), "dfh_mc_reorder: null vertex array");
    DFH_REQUIRE(n_faces == 0 || faces, "dfh_mc_reorder: null face array");
    hipStream_t s = (hipStream_t)stream;
    unsigned *first = static_cast<unsigned *>(workspace);
    unsigned *newid = first + n_verts;
    unsigned *bsum = newid + n_verts;
    const long nkeys = 3 * n_faces;
    const long nb = (nkeys + kMcKeysPerBlock - 1) / kMcKeysPerBlock;
    DFH_HIP_CHECK(hipMemsetAsync(first, 0xFF, sizeof(unsigned) * (size_t)n_verts, s));
    if (nkeys > 0) {
        hipLaunchKernelGGL(mc_first_use_kernel, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, s, faces, nkeys, first);
        hipLaunchKernelGGL(mc_rank_count_kernel, dim3((unsigned)nb), dim3(256), 0, s, faces, nkeys, first, bsum);
    }
    hipLaunchKernelGGL(mc_scan1_kernel, dim3(1), dim3(1024), 0, s, bsum, nb, used_out);
    if (nkeys > 0) {
        hipLaunchKernelGGL(mc_rank_apply_kernel, dim3((unsigned)nb), dim3(256), 0, s, faces, nkeys, first, bsum, newid);
        hipLaunchKernelGGL(mc_permute_kernel, dim3((unsigned)((n_verts + 255) / 256)), dim3(256), 0, s, verts_in, normals_in, values_in,
                           first, newid, n_verts, verts_out, normals_out, values_out);
        hipLaunchKernelGGL(mc_relabel_kernel, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, s, faces, nkeys, newid);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

}  // extern "C"
