// Marching cubes on the device: `_vertices`, `_faces`, `_normals` from a TSDF volume, the call the
// reference makes through skimage (measure.marching_cubes_lewiner at core/fusion_dm.py:319-331,342
// and core/fusion.py:554-568).  skimage's Lewiner tables are a third-party
// dependency that is absent here, so the triangle table is derived by tools/gen_mc_table.py
// (watertight, consistently oriented, and for the 88 configurations that occur in the reference's own
// output mesh meshes/original.obj triangulated exactly as there).  Pinned by that mesh
// (tests/golden/g9_mesh.npz): vertices on lattice edges at the linearly interpolated crossing in
// array-index coordinates, unit normals pointing down the gradient, faces wound with their
// right-hand normal up the gradient, zero-area faces dropped (allow_degenerate=False), face order by
// cube, vertex order by first use (dfh_mc_reorder); its face array is reproduced bit for bit.
// oracle/mc_np.py states the same computation in numpy.
//
// Work decomposition: lattice point p = (x, y, z) of the (step-subsampled) volume owns the three
// edges leaving it along +axis 0/1/2 and the cube whose corner 0 it is.  A tile = one z row (x, y, all z),
// tiles numbered in C order of the points (the count pass covers 4 rows per workgroup, the emit passes
// one active row per workgroup: they are latency-bound, so many short workgroups).  No atomics,
// deterministic output order (vertices by owner point then axis; faces by cube then table order):
//   mc_count_kernel        per-tile {vertices, faces, active}; signs compared in the volume's own type
//   mc_scan_chunk_kernel   exclusive scan inside chunks of 2048 tiles   } hierarchical scan
//   mc_scan_top_kernel     scan of the chunk totals, grand totals       }
//   mc_active_kernel       compacted list of tiles that emit anything (the emit passes launch only those)
//   mc_vertex_kernel       code[p] = crossing mask << 29 | index of p's first vertex; positions, normals
//   mc_face_kernel         faces, looking up the owners' codes
#include "dfh_common.h"
#include "dfh_mc_table.h"

namespace dfh {

constexpr int kMcBlock = 256;
constexpr int kMcRows = 4;                  // z rows (tiles) per workgroup of the count pass
constexpr int kMcChunk = 2048;              // tiles per scan chunk (256 threads x 8)
constexpr unsigned kMcBaseMask = (1u << 29) - 1u;

struct McParams {
    int Y, Z;                 // strides of the full volume (elements): x*Y*Z + y*Z + z
    int s;                    // step_size
    int NX, NY, NZ;           // lattice dims = ceil(dim / s)
    int nseg;                 // 256-point segments per z row
    int nty;                  // count-pass workgroups along y
    double level;
    float lo_f, eq_f;         // fp32 volumes: (double)v > level <=> v > lo_f (largest float <= level);
                              //               (double)v == level <=> v == eq_f (NaN when level is no float)
    long ntiles;
};

template <typename VolT>
__device__ __forceinline__ size_t mc_off(const McParams &p, int x, int y, int z) {
    return ((size_t)(x * p.s) * p.Y + (size_t)(y * p.s)) * p.Z + (size_t)(z * p.s);
}
template <typename VolT>
__device__ __forceinline__ double mc_val(const VolT *__restrict__ vol, const McParams &p, int x, int y, int z) {
    return (double)vol[mc_off<VolT>(p, x, y, z)];
}
__device__ __forceinline__ bool mc_above(float v, const McParams &p) { return v > p.lo_f; }
__device__ __forceinline__ bool mc_above(double v, const McParams &p) { return v > p.level; }
__device__ __forceinline__ bool mc_equal(float v, const McParams &p) { return v == p.eq_f; }
__device__ __forceinline__ bool mc_equal(double v, const McParams &p) { return v == p.level; }

// ---- sign classification: one wave per z row, four consecutive points per lane ----------------------
// The four z rows a point's cube touches: (x,y), (x+1,y), (x,y+1), (x+1,y+1).  Wave-uniform, so the address
// arithmetic is scalar; rows outside the lattice alias row (x,y) and are masked out.
template <typename VolT>
struct McRows {
    const VolT *r[4];         // index = ox + 2*oy  (bit 0: +x, bit 1: +y)
    bool hx, hy;
    bool vec;                 // 4 consecutive samples can be fetched with one 16-byte (fp32) / two (fp64) loads
};

template <typename VolT>
__device__ __forceinline__ McRows<VolT> mc_rows(const VolT *__restrict__ vol, const McParams &p, int x, int y) {
    McRows<VolT> w;
    w.hx = x + 1 < p.NX;
    w.hy = y + 1 < p.NY;
    const size_t sx = (size_t)p.s * p.Y * p.Z, sy = (size_t)p.s * p.Z;
    const VolT *b = vol + (size_t)x * sx + (size_t)y * sy;
    w.r[0] = b;
    w.r[1] = w.hx ? b + sx : b;
    w.r[2] = w.hy ? b + sy : b;
    w.r[3] = (w.hx && w.hy) ? b + sx + sy : b;
    w.vec = p.s == 1 && (p.Z & 3) == 0 && ((uintptr_t)vol & 15) == 0;
    return w;
}

// bit j (0..4) of the result: sample z0+j of the row is above the level / equals it (0 beyond the row's end)
template <typename VolT>
__device__ __forceinline__ void mc_row_bits(const VolT *__restrict__ rp, const McParams &p, bool vec, int z0, unsigned &ab, unsigned &eb) {
    VolT v[5];
    if (vec && z0 + 3 < p.NZ) {
        if constexpr (sizeof(VolT) == 4) {
            const float4 q = *reinterpret_cast<const float4 *>(rp + z0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            const double2 q0 = *reinterpret_cast<const double2 *>(rp + z0), q1 = *reinterpret_cast<const double2 *>(rp + z0 + 2);
            v[0] = q0.x; v[1] = q0.y; v[2] = q1.x; v[3] = q1.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rp[(size_t)min(z0 + j, p.NZ - 1) * p.s];
    }
    // sample z0+4 is the next lane's first sample (lanes hold consecutive groups of four); lane 63 fetches it
    {
        const VolT nxt = __shfl_down(v[0], 1, 64);
        v[4] = (threadIdx.x & 63) == 63 ? rp[(size_t)min(z0 + 4, p.NZ - 1) * p.s] : nxt;
    }
    ab = eb = 0u;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const bool in = z0 + j < p.NZ;
        if (in && mc_above(v[j], p)) ab |= 1u << j;
        if (in && mc_equal(v[j], p)) eb |= 1u << j;
    }
}

struct McSigns {
    unsigned above;           // bit c: corner c exists and value > level   (corner c: offset (c&1, (c>>1)&1, (c>>2)&1))
    unsigned eq;              // bit c: corner c exists and value == level
    unsigned cross;           // bit a: owned edge along axis a exists and is crossed
    bool cell;                // the cube with corner 0 = p exists
};

// signs of the four points z0..z0+3 of a row from the 5-bit sample masks of its four cube rows
// (index ox + 2*oy; rows outside the lattice: zero masks)
__device__ __forceinline__ void mc_assemble4(const unsigned (&ab)[4], const unsigned (&eb)[4], bool hx, bool hy, const McParams &p, int z0,
                                             McSigns (&q)[4]) {
    // all twenty samples on one side of the level (the bulk of the volume): nothing is crossed, no cube is cut
    const unsigned any = ab[0] | ab[1] | ab[2] | ab[3], all = ab[0] & ab[1] & ab[2] & ab[3];
    if (any == 0u || (all == 31u && hx && hy && z0 + 4 < p.NZ)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { q[j].above = q[j].eq = q[j].cross = 0u; q[j].cell = false; }
        return;
    }
    // bit j of the four rows -> one nibble (row ri at bit ri): bytes packed, bit picked, gathered by a multiply
    const unsigned pa = ab[0] | (ab[1] << 8) | (ab[2] << 16) | (ab[3] << 24);
    const unsigned pe = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (eb[3] << 24);
    unsigned na[5], ne[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        na[j] = ((((pa >> j) & 0x01010101u) * 0x01020408u) >> 24) & 15u;
        ne[j] = pe ? ((((pe >> j) & 0x01010101u) * 0x01020408u) >> 24) & 15u : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool inside = z0 + j < p.NZ, hz = z0 + j + 1 < p.NZ;
        const unsigned above = na[j] | (na[j + 1] << 4);
        const unsigned diff = above ^ ((above & 1u) ? 255u : 0u);          // corners on the other side than corner 0
        unsigned cross = 0u;
        if (hx) cross |= (diff >> 1) & 1u;
        if (hy) cross |= ((diff >> 2) & 1u) << 1;
        if (hz) cross |= ((diff >> 4) & 1u) << 2;
        q[j].above = inside ? above : 0u;
        q[j].eq = inside ? (ne[j] | (ne[j + 1] << 4)) : 0u;
        q[j].cross = inside ? cross : 0u;
        q[j].cell = inside && hx && hy && hz;
    }
}

// signs of the four points z0..z0+3 of row (x,y)
template <typename VolT>
__device__ __forceinline__ void mc_signs4(const McRows<VolT> &w, const McParams &p, int z0, McSigns (&q)[4]) {
    unsigned ab[4], eb[4];
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) {
        const bool have = (!(ri & 1) || w.hx) && (!(ri & 2) || w.hy);
        ab[ri] = eb[ri] = 0u;
        if (have) mc_row_bits(w.r[ri], p, w.vec, z0, ab[ri], eb[ri]);
    }
    mc_assemble4(ab, eb, w.hx, w.hy, p, z0, q);
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

__device__ __forceinline__ int mc_tri_count(const McSigns &q) {
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

// exclusive scan over the 64 lanes of a wave (no barrier); *total = wave sum
__device__ __forceinline__ int mc_wave_scan(int v, int *total) {
    const int lane = threadIdx.x & 63;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    *total = __shfl(inc, 63, 64);
    return inc - v;
}

// 256-point segment `seg` of a row in the row's activity mask; segments beyond the 31st share bit 31.
__device__ __forceinline__ unsigned mc_chunk_bit(int seg) { return 1u << (seg < 31 ? seg : 31); }

// count pass: one wave per kMcRows consecutive rows (x, y0..y0+3): the ten sample rows they touch are fetched
// back to back, then the four rows are classified; no barriers.  Workgroup = 4 waves = 16 rows.
// Plane x+1 is read again by the workgroups of plane x+1: workgroups are dealt round-robin to the 8 XCDs
// (separate L2s), so each XCD gets a contiguous slab of x planes and finds that re-read in its own L2.
template <typename VolT>
__global__ __launch_bounds__(kMcBlock) void mc_count_kernel(const VolT *__restrict__ vol, const McParams p,
                                                             uint4 *__restrict__ entries) {
    const int lane = threadIdx.x & 63;
    const int yg = (p.nty + kMcBlock / 64 - 1) / (kMcBlock / 64);      // workgroups per x plane
    int x, g;
    if ((p.NX & 7) == 0) {
        const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
        x = xcd * (p.NX >> 3) + j / yg;
        g = j % yg;
    } else {
        x = (int)blockIdx.x / yg;
        g = (int)blockIdx.x % yg;
    }
    const int y0 = (g * (kMcBlock / 64) + (int)(threadIdx.x >> 6)) * kMcRows;
    if (y0 >= p.NY) return;
    const bool hx = x + 1 < p.NX;
    const size_t sx = (size_t)p.s * p.Y * p.Z, sy = (size_t)p.s * p.Z;
    const VolT *base = vol + (size_t)x * sx + (size_t)y0 * sy;
    const bool vec = p.s == 1 && (p.Z & 3) == 0 && ((uintptr_t)vol & 15) == 0;
    int cnt[kMcRows];                                   // vertices | faces << 16 per row (<= 768 and <= 1280 per segment and lane group)
    unsigned mask[kMcRows];
#pragma unroll
    for (int r = 0; r < kMcRows; ++r) { cnt[r] = 0; mask[r] = 0u; }
    for (int seg = 0; seg < p.nseg; ++seg) {
        const int z0 = seg * kMcBlock + 4 * lane;
        unsigned A[2][kMcRows + 1], E[2][kMcRows + 1];
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
#pragma unroll
            for (int yy = 0; yy <= kMcRows; ++yy) {
                A[ox][yy] = E[ox][yy] = 0u;
                if (z0 < p.NZ && y0 + yy < p.NY && (ox == 0 || hx)) mc_row_bits(base + ox * sx + yy * sy, p, vec, z0, A[ox][yy], E[ox][yy]);
            }
#pragma unroll
        for (int r = 0; r < kMcRows; ++r) {
            int c = 0;
            if (z0 < p.NZ && y0 + r < p.NY) {
                const unsigned ab[4] = {A[0][r], A[1][r], A[0][r + 1], A[1][r + 1]};
                const unsigned eb[4] = {E[0][r], E[1][r], E[0][r + 1], E[1][r + 1]};
                McSigns q[4];
                mc_assemble4(ab, eb, hx, y0 + r + 1 < p.NY, p, z0, q);
#pragma unroll
                for (int j = 0; j < 4; ++j) c += __popc(q[j].cross) | (mc_tri_count(q[j]) << 16);
            }
            // per-row totals can exceed 16 bits over many segments: fold into 64-bit lanes later; here per segment only
            if (__any(c != 0)) mask[r] |= mc_chunk_bit(seg);
            cnt[r] += c;
        }
        // flush the packed counters before they can overflow (every 16 segments: 16 * 20 triangles per lane < 2^16)
        if ((seg & 15) == 15 || seg == p.nseg - 1) {
#pragma unroll
            for (int r = 0; r < kMcRows; ++r) {
                int c = cnt[r];                          // low half <= 64 * 192, no carry into the high half
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
                cnt[r] = 0;
                if (lane == 0 && y0 + r < p.NY) {
                    uint4 *e = entries + (long)x * p.NY + y0 + r;
                    const unsigned v = (unsigned)(c & 0xffff), f = (unsigned)(c >> 16) & 0xffffu;
                    if (seg < 16) *e = make_uint4(v, f, 0u, 0u);
                    else { e->x += v; e->y += f; }
                }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < kMcRows; ++r)
            if (y0 + r < p.NY) {
                uint4 *e = entries + (long)x * p.NY + y0 + r;
                e->z = (e->x | e->y) != 0u ? 1u : 0u;
                e->w = mask[r];
            }
    }
}

// entries[t] = {vertices, faces, active, -} -> exclusive prefixes inside the chunk; chunk sums to chunk_tot
__global__ __launch_bounds__(kMcBlock) void mc_scan_chunk_kernel(uint4 *__restrict__ entries, long ntiles, uint4 *__restrict__ chunk_tot) {
    __shared__ int lds[4];
    const long t0 = (long)blockIdx.x * kMcChunk + (long)threadIdx.x * 8;
    uint4 e[8];
    unsigned sv = 0, sf = 0, sa = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        e[j] = t0 + j < ntiles ? entries[t0 + j] : make_uint4(0u, 0u, 0u, 0u);
        const uint4 pre = make_uint4(sv, sf, sa, e[j].w);             // .w keeps the activity mask (non-zero = active)
        sv += e[j].x; sf += e[j].y; sa += e[j].z;
        e[j] = pre;
    }
    int tv, tf, ta;
    const unsigned bv = (unsigned)mc_block_scan((int)sv, lds, &tv);
    const unsigned bf = (unsigned)mc_block_scan((int)sf, lds, &tf);
    const unsigned ba = (unsigned)mc_block_scan((int)sa, lds, &ta);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (t0 + j < ntiles) entries[t0 + j] = make_uint4(e[j].x + bv, e[j].y + bf, e[j].z + ba, e[j].w);
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = make_uint4((unsigned)tv, (unsigned)tf, (unsigned)ta, 0u);
}

// chunk totals -> exclusive prefixes (few hundred at most: one thread), grand totals {vertices, faces, active tiles}
__global__ void mc_scan_top_kernel(uint4 *__restrict__ chunk_tot, long nchunks, long *__restrict__ totals) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long v = 0, f = 0, a = 0;
    for (long c = 0; c < nchunks; ++c) {
        const uint4 t = chunk_tot[c];
        chunk_tot[c] = make_uint4((unsigned)v, (unsigned)f, (unsigned)a, 0u);
        v += t.x; f += t.y; a += t.z;
    }
    totals[0] = (long)v; totals[1] = (long)f; totals[2] = (long)a;
}

__global__ __launch_bounds__(kMcBlock) void mc_active_kernel(const uint4 *__restrict__ entries, const uint4 *__restrict__ chunk_tot,
                                                             long ntiles, unsigned *__restrict__ list) {
    const long t = (long)blockIdx.x * kMcBlock + threadIdx.x;
    if (t >= ntiles) return;
    const uint4 e = entries[t];
    if (e.w) list[chunk_tot[t / kMcChunk].z + e.z] = (unsigned)t;
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

// The emit passes run on the active rows only, one wave per row: wave w of workgroup b takes entry 4 b + w of
// `list` (or row 4 b + w itself when list == NULL).  Everything inside a row is wave-synchronous: no barriers.
__device__ __forceinline__ long mc_tile(const unsigned *__restrict__ list, long nrows) {
    const long i = (long)blockIdx.x * (kMcBlock / 64) + (threadIdx.x >> 6);
    if (i >= nrows) return -1;
    return list ? (long)list[i] : i;
}

// One vertex: interpolated crossing of the edge leaving (x,y,z) along axis a, normal from the lattice gradient.
template <typename VolT>
__device__ __forceinline__ void mc_emit_vertex(const VolT *__restrict__ vol, const McParams &p, int x, int y, int z, int a, long vi,
                                               float *__restrict__ verts, float *__restrict__ normals, float *__restrict__ values) {
    const int x1 = x + (a == 0), y1 = y + (a == 1), z1 = z + (a == 2);
    const double f0 = mc_val(vol, p, x, y, z), f1 = mc_val(vol, p, x1, y1, z1);
    const double tt = (p.level - f0) / (f1 - f0);
    const double px = (double)x + (a == 0 ? tt : 0.0), py = (double)y + (a == 1 ? tt : 0.0), pz = (double)z + (a == 2 ? tt : 0.0);
    double g0[3], g1[3];
    mc_grad(vol, p, x, y, z, g0);
    mc_grad(vol, p, x1, y1, z1, g1);
    const double gx = g0[0] + tt * (g1[0] - g0[0]), gy = g0[1] + tt * (g1[1] - g0[1]), gz = g0[2] + tt * (g1[2] - g0[2]);
    const double n2 = (gx * gx + gy * gy) + gz * gz;
    const double nrm = sqrt(n2);
    const double inv = nrm > 0.0 ? -1.0 / nrm : 0.0;
    verts[3 * vi + 0] = (float)(px * (double)p.s);
    verts[3 * vi + 1] = (float)(py * (double)p.s);
    verts[3 * vi + 2] = (float)(pz * (double)p.s);
    normals[3 * vi + 0] = (float)(gx * inv);
    normals[3 * vi + 1] = (float)(gy * inv);
    normals[3 * vi + 2] = (float)(gz * inv);
    if (values) values[vi] = (float)(f0 > f1 ? f0 : f1);
}

template <typename VolT>
__global__ __launch_bounds__(kMcBlock) void mc_vertex_kernel(const VolT *__restrict__ vol, const McParams p,
                                                              const uint4 *__restrict__ entries, const uint4 *__restrict__ chunk_tot,
                                                              const unsigned *__restrict__ list, long nrows, unsigned *__restrict__ code,
                                                              float *__restrict__ verts, float *__restrict__ normals,
                                                              float *__restrict__ values, long cap) {
    __shared__ unsigned queue_all[kMcBlock / 64][3 * kMcBlock];   // per wave: crossed edges of one segment, z << 2 | axis
    const long t = mc_tile(list, nrows);
    if (t < 0) return;
    const int lane = threadIdx.x & 63;
    unsigned *queue = queue_all[threadIdx.x >> 6];
    const int x = (int)(t / p.NY), y = (int)(t - (long)x * p.NY);
    const uint4 ent = entries[t];
    unsigned carry = ent.x + chunk_tot[t / kMcChunk].x;
    const McRows<VolT> rows = mc_rows(vol, p, x, y);
    unsigned *code_row = code + ((long)x * p.NY + y) * p.NZ;
    for (int seg = 0; seg < p.nseg; ++seg) {
        if (list && !(ent.w & mc_chunk_bit(seg))) continue;          // nothing emitted here, nobody reads these codes
        const int z0 = seg * kMcBlock + 4 * lane;
        McSigns q[4];
        int cnt = 0;
        if (z0 < p.NZ) {
            mc_signs4(rows, p, z0, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) cnt += __popc(q[j].cross);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j].cross = 0u;
        }
        int tot;
        int rank = mc_wave_scan(cnt, &tot);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (z0 + j < p.NZ) code_row[z0 + j] = (q[j].cross << 29) | ((carry + (unsigned)rank) & kMcBaseMask);
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if ((q[j].cross >> a) & 1u) queue[rank++] = ((unsigned)(z0 + j) << 2) | (unsigned)a;
        }
        __builtin_amdgcn_wave_barrier();
        // light lanes enqueued, all lanes emit: the fp64 interpolation / gradient work runs on full waves
        for (int i = lane; i < tot; i += 64) {
            const long vi = (long)carry + i;
            const unsigned e = queue[i];
            if (vi < cap) mc_emit_vertex(vol, p, x, y, (int)(e >> 2), (int)(e & 3u), vi, verts, normals, values);
        }
        __builtin_amdgcn_wave_barrier();
        carry += (unsigned)tot;
    }
}

template <typename VolT>
__global__ __launch_bounds__(kMcBlock) void mc_face_kernel(const VolT *__restrict__ vol, const McParams p,
                                                            const uint4 *__restrict__ entries, const uint4 *__restrict__ chunk_tot,
                                                            const unsigned *__restrict__ list, long nrows, const unsigned *__restrict__ code,
                                                            int *__restrict__ faces, long cap) {
    // per wave: the kept triangles of one segment, (z - seg*256) << 12 | case << 4 | triangle number
    __shared__ unsigned queue_all[kMcBlock / 64][kMcMaxTris * kMcBlock];
    const long t = mc_tile(list, nrows);
    if (t < 0) return;
    const int lane = threadIdx.x & 63;
    unsigned *queue = queue_all[threadIdx.x >> 6];
    const int x = (int)(t / p.NY), y = (int)(t - (long)x * p.NY);
    const uint4 ent = entries[t];
    long carry = (long)ent.y + chunk_tot[t / kMcChunk].y;
    const McRows<VolT> rows = mc_rows(vol, p, x, y);
    const long row_idx = ((long)x * p.NY + y) * p.NZ;
    for (int seg = 0; seg < p.nseg; ++seg) {
        if (list && !(ent.w & mc_chunk_bit(seg))) continue;
        const int z0 = seg * kMcBlock + 4 * lane;
        McSigns q[4];
        int nt[4] = {0, 0, 0, 0};
        int cnt = 0;
        if (z0 < p.NZ) {
            mc_signs4(rows, p, z0, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) { nt[j] = mc_tri_count(q[j]); cnt += nt[j]; }
        }
        int tot;
        int rank = mc_wave_scan(cnt, &tot);
        if (tot == 0) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (nt[j] == 0) continue;
            const signed char *row = kMcTable + kMcRow * (int)q[j].above;
            const int n = row[0];
            for (int tr = 0; tr < n; ++tr)
                if (mc_keep(row, tr, q[j].eq)) queue[rank++] = ((unsigned)(4 * lane + j) << 12) | (q[j].above << 4) | (unsigned)tr;
        }
        __builtin_amdgcn_wave_barrier();
        // one triangle per lane: three code look-ups each
        for (int i = lane; i < tot; i += 64) {
            const long fi = carry + i;
            if (fi >= cap) continue;
            const unsigned e = queue[i];
            const long idx = row_idx + seg * kMcBlock + (long)(e >> 12);
            const signed char *row = kMcTable + kMcRow * (int)((e >> 4) & 255u);
            const int tr = (int)(e & 15u);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int a, c0;
                mc_edge(row[1 + 3 * tr + k], a, c0);
                const long owner = idx + (long)(c0 & 1) * p.NY * p.NZ + (long)((c0 >> 1) & 1) * p.NZ + (long)((c0 >> 2) & 1);
                const unsigned cd = code[owner];
                const unsigned mask = cd >> 29;
                faces[3 * fi + k] = (int)((cd & kMcBaseMask) + (unsigned)__popc(mask & ((1u << a) - 1u)));
            }
        }
        __builtin_amdgcn_wave_barrier();
        carry += tot;
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
    p.nseg = (p.NZ + kMcBlock - 1) / kMcBlock;
    p.nty = (p.NY + kMcRows - 1) / kMcRows;
    p.ntiles = (long)p.NX * p.NY;
    float lo = (float)level;                                // round to nearest, then step down if above
    if ((double)lo > level) lo = nextafterf(lo, -HUGE_VALF);
    p.lo_f = lo;
    p.eq_f = (double)(float)level == level ? (float)level : NAN;
    return true;
}

struct McWorkspace {
    unsigned *code;
    uint4 *entries, *chunk_tot;
    unsigned *list;
    long nchunks;
    size_t bytes;
};

static McWorkspace mc_workspace(const McParams &p, void *base) {
    McWorkspace w;
    const size_t npts = (size_t)p.NX * p.NY * p.NZ;
    w.nchunks = (p.ntiles + kMcChunk - 1) / kMcChunk;
    size_t off = 0;
    char *b = static_cast<char *>(base);
    auto take = [&](size_t n) { char *r = b ? b + off : nullptr; off += (n + 15) & ~(size_t)15; return r; };
    w.code = reinterpret_cast<unsigned *>(take(npts * sizeof(unsigned)));
    w.entries = reinterpret_cast<uint4 *>(take((size_t)p.ntiles * sizeof(uint4)));
    w.chunk_tot = reinterpret_cast<uint4 *>(take((size_t)w.nchunks * sizeof(uint4)));
    w.list = reinterpret_cast<unsigned *>(take((size_t)p.ntiles * sizeof(unsigned)));
    w.bytes = off;
    return w;
}

}  // namespace dfh

extern "C" {

size_t dfh_mc_workspace_bytes(const int res[3], int step) {
    using namespace dfh;
    if (!res || res[0] <= 0 || res[1] <= 0 || res[2] <= 0 || step < 1) return 0;
    McParams p;
    mc_params(res, step, 0.0, p);
    return mc_workspace(p, nullptr).bytes;
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
    DFH_REQUIRE(mc_params(res, step, level, p), "dfh_mc_count: more than 65535 lattice planes");
    const McWorkspace w = mc_workspace(p, workspace);
    hipStream_t s = (hipStream_t)stream;
    const long ncount = (long)((p.nty + kMcBlock / 64 - 1) / (kMcBlock / 64)) * p.NX;
    DFH_REQUIRE(ncount < (1L << 31), "dfh_mc_count: grid too large");
    const dim3 grid((unsigned)ncount);
    if (vol_dtype == DFH_F32)
        hipLaunchKernelGGL(mc_count_kernel<float>, grid, dim3(kMcBlock), 0, s, (const float *)vol, p, w.entries);
    else
        hipLaunchKernelGGL(mc_count_kernel<double>, grid, dim3(kMcBlock), 0, s, (const double *)vol, p, w.entries);
    hipLaunchKernelGGL(mc_scan_chunk_kernel, dim3((unsigned)w.nchunks), dim3(kMcBlock), 0, s, w.entries, p.ntiles, w.chunk_tot);
    hipLaunchKernelGGL(mc_scan_top_kernel, dim3(1), dim3(64), 0, s, w.chunk_tot, w.nchunks, totals_out);
    hipLaunchKernelGGL(mc_active_kernel, dim3((unsigned)((p.ntiles + kMcBlock - 1) / kMcBlock)), dim3(kMcBlock), 0, s, w.entries,
                       w.chunk_tot, p.ntiles, w.list);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_mc_emit(const void *vol, int vol_dtype, const int res[3], int step, double level, void *workspace,
                size_t workspace_bytes, float *verts, float *normals, float *values, int *faces, long cap_verts, long cap_faces,
                long n_active_tiles, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(vol && res && workspace, "dfh_mc_emit: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_mc_emit: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && step >= 1, "dfh_mc_emit: bad grid / step");
    DFH_REQUIRE(workspace_bytes >= dfh_mc_workspace_bytes(res, step), "dfh_mc_emit: workspace too small");
    DFH_REQUIRE(cap_verts >= 0 && cap_faces >= 0, "dfh_mc_emit: negative capacity");
    DFH_REQUIRE(cap_verts <= (long)kMcBaseMask, "dfh_mc_emit: more than 2^29-1 vertices");
    DFH_REQUIRE((cap_verts == 0 || (verts && normals)) && (cap_faces == 0 || faces), "dfh_mc_emit: null output");
    McParams p;
    DFH_REQUIRE(mc_params(res, step, level, p), "dfh_mc_emit: more than 65535 lattice planes");
    DFH_REQUIRE(n_active_tiles <= p.ntiles, "dfh_mc_emit: n_active_tiles exceeds the tile count");
    const McWorkspace w = mc_workspace(p, workspace);
    const long nrows = n_active_tiles >= 0 ? n_active_tiles : p.ntiles;          // < 0: visit every row, no list
    const unsigned *list = n_active_tiles >= 0 ? w.list : nullptr;
    if (nrows == 0) return DFH_OK;
    const long nlaunch = (nrows + kMcBlock / 64 - 1) / (kMcBlock / 64);          // one wave per row
    DFH_REQUIRE(nlaunch < (1L << 31), "dfh_mc_emit: too many rows");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nlaunch);
    if (vol_dtype == DFH_F32) {
        hipLaunchKernelGGL(mc_vertex_kernel<float>, grid, dim3(kMcBlock), 0, s, (const float *)vol, p, w.entries, w.chunk_tot, list, nrows,
                           w.code, verts, normals, values, cap_verts);
        hipLaunchKernelGGL(mc_face_kernel<float>, grid, dim3(kMcBlock), 0, s, (const float *)vol, p, w.entries, w.chunk_tot, list, nrows,
                           w.code, faces, cap_faces);
    } else {
        hipLaunchKernelGGL(mc_vertex_kernel<double>, grid, dim3(kMcBlock), 0, s, (const double *)vol, p, w.entries, w.chunk_tot, list, nrows,
                           w.code, verts, normals, values, cap_verts);
        hipLaunchKernelGGL(mc_face_kernel<double>, grid, dim3(kMcBlock), 0, s, (const double *)vol, p, w.entries, w.chunk_tot, list, nrows,
                           w.code, faces, cap_faces);
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
