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

// The same counts for float32 volumes whose z rows are a multiple of 4 long: a thread takes FOUR CONSECUTIVE voxels (one 16-byte
// load of T and of W; the count of a block does not depend on which thread looks at which voxel) and runs band_sample -- the
// neighbour loads and the gradient test -- only for the few per cent that lie in the band.
__global__ __launch_bounds__(256) void surface_count_vec_kernel(const float *__restrict__ T, const float *__restrict__ W,
                                                                 const ExtractParams p, int *__restrict__ block_count) {
    __shared__ int red[4];
    const long v0 = (long)blockIdx.x * kExVox + 4 * (long)threadIdx.x;
    int c = 0;
    if (v0 < p.nvox) {                                   // nvox % 4 == 0: the pack is inside the volume
        const float4 t = *reinterpret_cast<const float4 *>(T + v0);
        const float4 w = *reinterpret_cast<const float4 *>(W + v0);
        const float tt[4] = {t.x, t.y, t.z, t.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((double)ww[j] > 0.0 && fabs((double)tt[j]) < p.band) c += band_sample<float>(T, W, p, v0 + j, nullptr, nullptr) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of the block counts in place (single workgroup), total -> *total_out.  16 K counts per round: every
// thread takes 16 of them 1024 apart (coalesced, all loads in flight at once), waves scan by shuffles, wave 0 scans the
// 256 wave totals, three barriers per round.
constexpr int kScanE = 16;
__global__ __launch_bounds__(1024) void surface_scan_kernel(int *__restrict__ block_count, int nblocks, long *__restrict__ total_out) {
    __shared__ long wtot[kScanE * 16];          // [chunk e][wave w] inclusive totals, chunk-major = scan order
    __shared__ long woff[kScanE * 16 + 1];      // exclusive offsets of the same, [256] = the round's total
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    long carry = 0;
    for (long base = 0; base < nblocks; base += 1024 * kScanE) {
        int v[kScanE];
        long inc[kScanE];
#pragma unroll
        for (int e = 0; e < kScanE; ++e) {
            const long idx = base + (long)e * 1024 + t;
            v[e] = idx < nblocks ? block_count[idx] : 0;
        }
#pragma unroll
        for (int e = 0; e < kScanE; ++e) {
            long x = v[e];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const long y = __shfl_up(x, o, 64);
                if (lane >= o) x += y;
            }
            inc[e] = x;
            if (lane == 63) wtot[e * 16 + wv] = x;
        }
        __syncthreads();
        if (wv == 0) {                            // 256 partials, 4 consecutive ones per lane
            long p0 = wtot[4 * lane], p1 = wtot[4 * lane + 1], p2 = wtot[4 * lane + 2], p3 = wtot[4 * lane + 3];
            long x = p0 + p1 + p2 + p3;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const long y = __shfl_up(x, o, 64);
                if (lane >= o) x += y;
            }
            const long ex = x - (p0 + p1 + p2 + p3);
            woff[4 * lane] = ex;
            woff[4 * lane + 1] = ex + p0;
            woff[4 * lane + 2] = ex + p0 + p1;
            woff[4 * lane + 3] = ex + p0 + p1 + p2;
            if (lane == 63) woff[kScanE * 16] = x;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kScanE; ++e) {
            const long idx = base + (long)e * 1024 + t;
            if (idx < nblocks) block_count[idx] = (int)(carry + woff[e * 16 + wv] + inc[e] - v[e]);   // capacity is checked by the host against the total
        }
        carry += woff[kScanE * 16];
        __syncthreads();                          // wtot / woff are rewritten in the next round
    }
    if (t == 0) {
        *total_out = carry;
        block_count[nblocks] = (int)carry;        // sentinel: block b emits offset[b+1] - offset[b] samples
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
    return sizeof(int) * (size_t)((nvox + dfh::kExVox - 1) / dfh::kExVox + 1) + sizeof(long);      // counts + sentinel
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
    hipLaunchKernelGGL(surface_scan_kernel, dim3(1), dim3(1024), 0, s, bc, (int)nb, total_out);
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
    if (vol_dtype == DFH_F32) {
        hipLaunchKernelGGL(surface_emit_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float *)tsdf, (const float *)tsdf_w, p, bo, pos_out, nrm_out, capacity);
    } else {
        hipLaunchKernelGGL(surface_emit_kernel<double>, dim3((unsigned)nb), dim3(256), 0, s, (const double *)tsdf, (const double *)tsdf_w, p, bo, pos_out, nrm_out, capacity);
    }
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

}  // extern "C"
