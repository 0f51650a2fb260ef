// Surface-sample extraction from the canonical TSDF: the step in front of the warp solve.
// The reference gets its samples from skimage's marching cubes (core/fusion.py:554-568,
// core/fusion_dm.py:319-331: `_vertices`, `_normals`), which is outside this hot path
// (SURVEY.md §8(f) rank 1).  The stand-in used by the solve: every band voxel (w > 0, |T| < band,
// T in voxel units as fuseDepths stores it) yields one sample -- the voxel centre moved onto the
// zero level set along the TSDF gradient, with the normalised gradient as normal.  Three launches:
// per-block counts, a single-block exclusive scan, ordered emission (samples come out in voxel
// order [x][y][z], deterministically, with no atomics).
#include "dfh_common.h"

namespace dfh {

constexpr int kExVox = 1024;          // voxels per block (256 threads x 4 along z)

struct ExtractParams {
    int X, Y, Z;                      // slab dims (planes of this buffer)
    int x0;                           // global index of plane 0
    double band;
    long nvox;
};

template <typename VolT>
__device__ __forceinline__ bool band_sample(const VolT *__restrict__ T, const VolT *__restrict__ W, const ExtractParams &p,
                                            long v, double *pos, double *nrm) {
    if (v >= p.nvox) return false;
    const double t = (double)T[v];
    if (!((double)W[v] > 0.0) || !(fabs(t) < p.band)) return false;
    const int z = (int)(v % p.Z);
    const int y = (int)((v / p.Z) % p.Y);
    const int x = (int)(v / ((long)p.Z * p.Y));
    const long sx = (long)p.Y * p.Z, sy = p.Z;
    // central differences inside the slab, one-sided at its faces
    const int xl = x > 0 ? x - 1 : 0, xh = x < p.X - 1 ? x + 1 : p.X - 1;
    const int yl = y > 0 ? y - 1 : 0, yh = y < p.Y - 1 ? y + 1 : p.Y - 1;
    const int zl = z > 0 ? z - 1 : 0, zh = z < p.Z - 1 ? z + 1 : p.Z - 1;
    const double gx = xh > xl ? ((double)T[xh * sx + y * sy + z] - (double)T[xl * sx + y * sy + z]) / (double)(xh - xl) : 0.0;
    const double gy = yh > yl ? ((double)T[x * sx + yh * sy + z] - (double)T[x * sx + yl * sy + z]) / (double)(yh - yl) : 0.0;
    const double gz = zh > zl ? ((double)T[x * sx + y * sy + zh] - (double)T[x * sx + y * sy + zl]) / (double)(zh - zl) : 0.0;
    const double n = sqrt((gx * gx + gy * gy) + gz * gz);
    if (!(n > 1e-6)) return false;
    if (pos) {
        const double nx = gx / n, ny = gy / n, nz = gz / n;
        nrm[0] = nx; nrm[1] = ny; nrm[2] = nz;
        // one Newton step onto the zero level set: centre - T grad / |grad|^2  (a projective TSDF has |grad| > 1 on
        // surfaces oblique to the camera; stepping by T along the unit normal would overshoot there).  Where the
        // gradient is flatter than a distance field's (|grad| < 1: borders of the observed region, noise) the step is
        // |T| and no more, so that a sample never leaves its voxel's band neighbourhood
        const double st = t / fmax(n, 1.0);
        pos[0] = (double)(x + p.x0) - st * nx;
        pos[1] = (double)y - st * ny;
        pos[2] = (double)z - st * nz;
    }
    return true;
}

template <typename VolT>
__global__ __launch_bounds__(256) void surface_count_kernel(const VolT *__restrict__ T, const VolT *__restrict__ W,
                                                             const ExtractParams p, int *__restrict__ block_count) {
    __shared__ int red[4];
    // thread t takes voxels t, t+256, t+512, t+768 of the block's 1024: consecutive lanes read consecutive voxels
    const long v0 = (long)blockIdx.x * kExVox + threadIdx.x;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c += band_sample<VolT>(T, W, p, v0 + j * 256, nullptr, nullptr) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---- float32 volumes whose z rows are a multiple of 4 long: a thread takes FOUR CONSECUTIVE voxels along z ------------------
// One 16-byte load of T and of W per thread; a thread that finds one of its voxels in the band fetches the x and y neighbours
// of all four with four more 16-byte loads (the neighbours of consecutive voxels are consecutive) and the two z neighbours
// outside the pack -- one round trip, where band_sample makes seven dependent-free but separate 4-byte loads per voxel behind
// three 64-bit divisions.  band_pack_eval evaluates band_sample's expressions on those values: the same doubles, the same
// decisions, the same samples.
struct BandPack {
    float t[4], w[4];            // the thread's voxels (x, y, z0 .. z0 + 3)
    float xm[4], xp[4], ym[4], yp[4];
    float zm, zp;                // T at z0 - 1 and z0 + 4 (clamped to the pack's own ends at the faces)
    int x, y, z0;
    int dx, dy;                  // xh - xl, yh - yl of the central differences (2 inside, 1 at a face, 0 on a one-voxel-thick slab)
    unsigned in_band;            // bit j: voxel j has w > 0 and |T| < band
};

__device__ __forceinline__ void ld4(const float *a, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4 *>(a);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// loads the pack at voxel v0 (a multiple of 4, inside the volume); neighbours only if a voxel lies in the band
__device__ __forceinline__ void band_pack_load(const float *__restrict__ T, const float *__restrict__ W, const ExtractParams &p, long v0, BandPack &k) {
    ld4(T + v0, k.t);
    k.in_band = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (fabs((double)k.t[j]) < p.band) k.in_band |= 1u << j;
    if (!k.in_band) return;
    // the weights only of packs with a voxel inside the band (a twentieth of them): the sweep reads 4 B per voxel, not 8
    ld4(W + v0, k.w);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (!((double)k.w[j] > 0.0)) k.in_band &= ~(1u << j);
    if (!k.in_band) return;
    long row;                    // index of the z row
    if (p.nvox < (1L << 31)) {   // (32-bit divisions where the volume allows them)
        const unsigned r = (unsigned)v0 / (unsigned)p.Z;
        k.z0 = (int)((unsigned)v0 - r * (unsigned)p.Z);
        k.x = (int)(r / (unsigned)p.Y);
        k.y = (int)(r - (unsigned)k.x * (unsigned)p.Y);
        row = r;
    } else {
        row = v0 / p.Z;
        k.z0 = (int)(v0 - row * p.Z);
        k.x = (int)(row / p.Y);
        k.y = (int)(row - (long)k.x * p.Y);
    }
    (void)row;
    const long sx = (long)p.Y * p.Z, sy = p.Z;
    const int xl = k.x > 0 ? k.x - 1 : 0, xh = k.x < p.X - 1 ? k.x + 1 : p.X - 1;
    const int yl = k.y > 0 ? k.y - 1 : 0, yh = k.y < p.Y - 1 ? k.y + 1 : p.Y - 1;
    k.dx = xh - xl; k.dy = yh - yl;
    ld4(T + (xl * sx + k.y * sy + k.z0), k.xm);
    ld4(T + (xh * sx + k.y * sy + k.z0), k.xp);
    ld4(T + (k.x * sx + yl * sy + k.z0), k.ym);
    ld4(T + (k.x * sx + yh * sy + k.z0), k.yp);
    k.zm = k.z0 > 0 ? T[v0 - 1] : k.t[0];
    k.zp = k.z0 + 4 < p.Z ? T[v0 + 4] : k.t[3];
}

// band_sample for voxel j of a loaded pack
__device__ __forceinline__ bool band_pack_eval(const BandPack &k, const ExtractParams &p, int j, double *pos, double *nrm) {
    if (!((k.in_band >> j) & 1u)) return false;
    const double t = (double)k.t[j];
    const int z = k.z0 + j;
    const int zl = z > 0 ? z - 1 : 0, zh = z < p.Z - 1 ? z + 1 : p.Z - 1;
    const float tzl = zl == z ? k.t[j] : (j > 0 ? k.t[j - 1] : k.zm);
    const float tzh = zh == z ? k.t[j] : (j < 3 ? k.t[j + 1] : k.zp);
    const double gx = k.dx > 0 ? ((double)k.xp[j] - (double)k.xm[j]) / (double)k.dx : 0.0;
    const double gy = k.dy > 0 ? ((double)k.yp[j] - (double)k.ym[j]) / (double)k.dy : 0.0;
    const double gz = zh > zl ? ((double)tzh - (double)tzl) / (double)(zh - zl) : 0.0;
    const double n = sqrt((gx * gx + gy * gy) + gz * gz);
    if (!(n > 1e-6)) return false;
    if (pos) {
        const double nx = gx / n, ny = gy / n, nz = gz / n;
        nrm[0] = nx; nrm[1] = ny; nrm[2] = nz;
        const double st = t / fmax(n, 1.0);              // (the bounded Newton step of band_sample)
        pos[0] = (double)(k.x + p.x0) - st * nx;
        pos[1] = (double)k.y - st * ny;
        pos[2] = (double)z - st * nz;
    }
    return true;
}

__global__ __launch_bounds__(256) void surface_count_vec_kernel(const float *__restrict__ T, const float *__restrict__ W,
                                                                 const ExtractParams p, int *__restrict__ block_count) {
    __shared__ int red[4];
    const long v0 = (long)blockIdx.x * kExVox + 4 * (long)threadIdx.x;
    int c = 0;
    if (v0 < p.nvox) {                                   // nvox % 4 == 0: the pack is inside the volume
        BandPack k;
        band_pack_load(T, W, p, v0, k);
        if (k.in_band) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c += band_pack_eval(k, p, j, nullptr, nullptr) ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// the emit pass on packs: samples in voxel order = thread order, then the thread's own four
__global__ __launch_bounds__(256) void surface_emit_vec_kernel(const float *__restrict__ T, const float *__restrict__ W,
                                                                const ExtractParams p, const int *__restrict__ block_offset,
                                                                double *__restrict__ pos_out, double *__restrict__ nrm_out,
                                                                long capacity) {
    __shared__ int wave_cnt[4];
    if (block_offset[blockIdx.x + 1] == block_offset[blockIdx.x]) return;       // nothing to emit: do not re-read the voxels
    const long v0 = (long)blockIdx.x * kExVox + 4 * (long)threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double pos[4][3], nrm[4][3];
    bool ok[4] = {false, false, false, false};
    if (v0 < p.nvox) {
        BandPack k;
        band_pack_load(T, W, p, v0, k);
        if (k.in_band) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ok[j] = band_pack_eval(k, p, j, pos[j], nrm[j]);
        }
    }
    int before = 0, wave_total = 0;                      // samples of lower lanes of this wave; of the whole wave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long m = __ballot(ok[j]);
        before += __popcll(m & ((1ull << lane) - 1ull));
        wave_total += __popcll(m);
    }
    if (lane == 0) wave_cnt[wv] = wave_total;
    __syncthreads();
    long mine = (long)block_offset[blockIdx.x] + before;
    for (int w_ = 0; w_ < wv; ++w_) mine += wave_cnt[w_];
    const long total = (long)block_offset[gridDim.x];    // sentinel written by the scan: number of samples of the volume
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!ok[j]) continue;
        long dst = mine;
        bool keep = true;
        if (capacity < total) {                          // an EVEN subsample in voxel order (see surface_emit_kernel)
            dst = (long)(((__int128)mine * capacity) / total);
            keep = mine == 0 || (long)(((__int128)(mine - 1) * capacity) / total) != dst;
        }
        if (keep) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { pos_out[3 * dst + a] = pos[j][a]; nrm_out[3 * dst + a] = nrm[j][a]; }
        }
        ++mine;
    }
}

// Exclusive scan of the block counts in place, total -> *total_out and the sentinel block_count[nblocks].
// Two launches: a workgroup scans ITS 4 096 counts from zero (a thread takes four consecutive ones with one 16-byte load, the
// waves scan the threads' sums with six 32-bit shuffles, sixteen wave totals go through LDS) and leaves the chunk's total; then
// every 256 counts add the sum of the chunk totals before theirs.  (Rounds 2-4 scanned 16 K counts per workgroup with sixteen
// 64-bit shuffle scans per thread: 192 ds_bpermute per wave, 3 072 through ONE CU's LDS pipe -- 29 us at 256^3 where this takes
// 5 + 4; a single workgroup walking a 512^3 volume's list took 235 us.  Tried with it and dropped: one wave per block, four
// blocks per workgroup in the count and emit passes (a quarter of the workgroups, no barrier) -- count 37 -> 50 us, emit 54 -> 75 us
// at 256^3: the packs inside the band are what these passes take, and a wave then walks four of them in turn.)
constexpr int kScanChunk = 4096;
template <bool SINGLE>
__global__ __launch_bounds__(1024) void surface_scan_chunk_kernel(int *__restrict__ block_count, int nblocks, long *__restrict__ chunk_tot,
                                                                   long *__restrict__ total_out) {
    __shared__ int wtot[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long base = (long)blockIdx.x * kScanChunk + 4 * (long)t;
    int v[4] = {0, 0, 0, 0};
    if (base + 3 < nblocks && ((uintptr_t)block_count & 15) == 0) {
        const int4 q = *reinterpret_cast<const int4 *>(block_count + base);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = base + j < nblocks ? block_count[base + j] : 0;
    }
    const int sum = (v[0] + v[1]) + (v[2] + v[3]);      // (a chunk holds at most 4 096 x 1 024 samples: int is enough)
    int x = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wtot[wv] = x;
    __syncthreads();
    int off = 0, all = 0;
#pragma unroll
    for (int w_ = 0; w_ < 16; ++w_) {
        const int q = wtot[w_];
        off += w_ < wv ? q : 0;
        all += q;
    }
    const int ex = off + x - sum;
    const int o4[4] = {ex, ex + v[0], ex + v[0] + v[1], ex + v[0] + v[1] + v[2]};
    if (base + 3 < nblocks && ((uintptr_t)block_count & 15) == 0) {
        *reinterpret_cast<int4 *>(block_count + base) = make_int4(o4[0], o4[1], o4[2], o4[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j < nblocks) block_count[base + j] = o4[j];
    }
    if (t == 0) {
        if (SINGLE) {
            *total_out = (long)all;
            block_count[nblocks] = all;               // sentinel: block b emits offset[b+1] - offset[b] samples
        } else {
            chunk_tot[blockIdx.x] = (long)all;
        }
    }
}

// 256 counts per workgroup, all of one chunk: + the totals of the chunks before it (added up in index order by the workgroup
// itself: a few hundred values at most); the last workgroup also leaves the volume's total and the sentinel
__global__ __launch_bounds__(256) void surface_scan_add_kernel(int *__restrict__ block_count, int nblocks, const long *__restrict__ chunk_tot,
                                                                long *__restrict__ total_out) {
    __shared__ long part[4];
    const int c = (int)(((long)blockIdx.x * 256) / kScanChunk);
    const bool last = blockIdx.x == gridDim.x - 1;
    const int upto = last ? c + 1 : c;                // (the last workgroup lies in the last chunk: + its own total = the volume's)
    long v = 0;
    for (int j = threadIdx.x; j < upto; j += 256) v += chunk_tot[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    const long sum = (part[0] + part[1]) + (part[2] + part[3]);
    const long base = last ? sum - chunk_tot[c] : sum;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nblocks && base != 0) block_count[i] += (int)base;
    if (last && threadIdx.x == 0) {
        *total_out = sum;
        block_count[nblocks] = (int)sum;
    }
}

template <typename VolT>
__global__ __launch_bounds__(256) void surface_emit_kernel(const VolT *__restrict__ T, const VolT *__restrict__ W,
                                                            const ExtractParams p, const int *__restrict__ block_offset,
                                                            double *__restrict__ pos_out, double *__restrict__ nrm_out,
                                                            long capacity) {
    __shared__ int wave_cnt[4][4];               // [chunk j][wave]: samples of voxels j*256 + 64*wave .. +63
    if (block_offset[blockIdx.x + 1] == block_offset[blockIdx.x]) return;       // nothing to emit: do not re-read the voxels
    // thread t takes voxels t, t+256, t+512, t+768 of the block (coalesced); samples keep voxel order: chunk j, then thread
    const long v0 = (long)blockIdx.x * kExVox + threadIdx.x;
    double pos[4][3], nrm[4][3];
    bool ok[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int before[4];                               // samples of lower lanes of this wave in chunk j
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ok[j] = band_sample<VolT>(T, W, p, v0 + j * 256, pos[j], nrm[j]);
        const unsigned long long m = __ballot(ok[j]);
        before[j] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[j][wv] = __popcll(m);
    }
    __syncthreads();
    long at = (long)block_offset[blockIdx.x];
    const long total = (long)block_offset[gridDim.x];      // sentinel written by the scan: number of samples of the volume
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long mine = at + before[j];
        for (int w_ = 0; w_ < 4; ++w_) {
            if (w_ < wv) mine += wave_cnt[j][w_];
            at += wave_cnt[j][w_];               // after the loop: start of chunk j+1
        }
        // capacity < total: an EVEN subsample in voxel order -- sample i goes to slot floor(i * capacity / total) and is kept
        // iff it is the first one of its slot (every slot gets exactly one), instead of the first `capacity` samples,
        // which would all lie in the lowest x planes
        long dst = mine;
        bool keep = ok[j];
        if (capacity < total) {
            dst = (long)(((__int128)mine * capacity) / total);
            keep = keep && (mine == 0 || (long)(((__int128)(mine - 1) * capacity) / total) != dst);
        }
        if (keep) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { pos_out[3 * dst + a] = pos[j][a]; nrm_out[3 * dst + a] = nrm[j][a]; }
        }
    }
}

}  // namespace dfh

extern "C" {

size_t dfh_surface_workspace_bytes(const int res[3]) {
    if (!res || res[0] <= 0 || res[1] <= 0 || res[2] <= 0) return 0;
    const long nvox = (long)res[0] * res[1] * res[2];
    const size_t nb = (size_t)((nvox + dfh::kExVox - 1) / dfh::kExVox);
    const size_t nchunks = (nb + dfh::kScanChunk - 1) / dfh::kScanChunk;
    return ((sizeof(int) * (nb + 1) + 15) & ~(size_t)15) + sizeof(long) * (nchunks + 2);          // counts + sentinel | chunk totals of the scan
}

int dfh_surface_count(const void *tsdf, const void *tsdf_w, int vol_dtype, const int res[3], double band, void *workspace,
                      size_t workspace_bytes, long *total_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(tsdf && tsdf_w && res && workspace && total_out, "dfh_surface_count: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_surface_count: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && band > 0.0, "dfh_surface_count: bad grid / band");
    DFH_REQUIRE(workspace_bytes >= dfh_surface_workspace_bytes(res), "dfh_surface_count: workspace too small");
    ExtractParams p{res[0], res[1], res[2], 0, band, (long)res[0] * res[1] * res[2]};
    const long nb = (p.nvox + kExVox - 1) / kExVox;
    DFH_REQUIRE(nb < (1L << 31), "dfh_surface_count: grid too large");
    hipStream_t s = (hipStream_t)stream;
    int *bc = static_cast<int *>(workspace);
    if (vol_dtype == DFH_F32 && res[2] % 4 == 0 && ((uintptr_t)tsdf & 15) == 0 && ((uintptr_t)tsdf_w & 15) == 0) {
        hipLaunchKernelGGL(surface_count_vec_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const float *)tsdf, (const float *)tsdf_w, p, bc);
    } else if (vol_dtype == DFH_F32) {
        hipLaunchKernelGGL(surface_count_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float *)tsdf, (const float *)tsdf_w, p, bc);
    } else {
        hipLaunchKernelGGL(surface_count_kernel<double>, dim3((unsigned)nb), dim3(256), 0, s, (const double *)tsdf, (const double *)tsdf_w, p, bc);
    }
    if (nb <= kScanChunk) {
        hipLaunchKernelGGL(surface_scan_chunk_kernel<true>, dim3(1), dim3(1024), 0, s, bc, (int)nb, (long *)nullptr, total_out);
    } else {
        const int nchunks = (int)((nb + kScanChunk - 1) / kScanChunk);
        long *chunk_tot = reinterpret_cast<long *>(static_cast<char *>(workspace) + ((sizeof(int) * ((size_t)nb + 1) + 15) & ~(size_t)15));
        hipLaunchKernelGGL(surface_scan_chunk_kernel<false>, dim3((unsigned)nchunks), dim3(1024), 0, s, bc, (int)nb, chunk_tot, total_out);
        hipLaunchKernelGGL(surface_scan_add_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, bc, (int)nb, (const long *)chunk_tot, total_out);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_surface_emit(const void *tsdf, const void *tsdf_w, int vol_dtype, const int res[3], int x0, double band,
                     const void *workspace, double *pos_out, double *nrm_out, long capacity, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(tsdf && tsdf_w && res && workspace, "dfh_surface_emit: null pointer");
    DFH_REQUIRE(vol_dtype == DFH_F32 || vol_dtype == DFH_F64, "dfh_surface_emit: bad vol_dtype %d", vol_dtype);
    DFH_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && band > 0.0 && capacity >= 0, "dfh_surface_emit: bad arguments");
    if (capacity == 0) return DFH_OK;
    DFH_REQUIRE(pos_out && nrm_out, "dfh_surface_emit: null output");
    ExtractParams p{res[0], res[1], res[2], x0, band, (long)res[0] * res[1] * res[2]};
    const long nb = (p.nvox + kExVox - 1) / kExVox;
    hipStream_t s = (hipStream_t)stream;
    const int *bo = static_cast<const int *>(workspace);
    if (vol_dtype == DFH_F32 && res[2] % 4 == 0 && ((uintptr_t)tsdf & 15) == 0 && ((uintptr_t)tsdf_w & 15) == 0) {
        hipLaunchKernelGGL(surface_emit_vec_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const float *)tsdf, (const float *)tsdf_w, p, bo, pos_out, nrm_out, capacity);
    } else if (vol_dtype == DFH_F32) {
        hipLaunchKernelGGL(surface_emit_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float *)tsdf, (const float *)tsdf_w, p, bo, pos_out, nrm_out, capacity);
    } else {
        hipLaunchKernelGGL(surface_emit_kernel<double>, dim3((unsigned)nb), dim3(256), 0, s, (const double *)tsdf, (const double *)tsdf_w, p, bo, pos_out, nrm_out, capacity);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

}  // extern "C"
