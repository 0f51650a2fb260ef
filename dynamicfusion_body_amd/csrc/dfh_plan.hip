// Per-frame bookkeeping of the warp-field solve on the device: samples sorted by node tuple, and the static plan of
// dfh_gn_build_planned (rows = runs of equal tuples inside kGnTile-sample tiles; per 6x6 block and per node the list of
// (row, slot) entries that contribute to it).  The host used to assemble all this from ~35 torch launches per frame
// (pack, sort, cumsum, searchsorted, ...): host-bound at ~0.6 ms.  Here it is a handful of launches behind three C
// entry points; the two key sorts are rocPRIM's device radix sort (stable, so every list comes out in ascending entry
// order -- the fixed summation order the bit-reproducible build relies on), everything else is hand-written.
#include "dfh_common.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace dfh {

// rocPRIM takes its merge sort below a million items; its default block sort handles 1 024 items, i.e. nine merge passes of
// two small launches each for the 520 000 sample keys of a frame (23 launches, ~6 us each: bound by launches, not bytes).
// 4 096 items per sorted block: seven passes.  (The Onesweep radix path was slower at this size: profiles/r2_gn_experiments.txt.)
#ifndef DFH_SORT_BLOCK_ITEMS
#define DFH_SORT_BLOCK_ITEMS 8
#endif
using SortCfg = rocprim::radix_sort_config<rocprim::default_config,
                                           rocprim::merge_sort_config<512, 512, DFH_SORT_BLOCK_ITEMS, 128, 256, 8>,
                                           rocprim::default_config>;
constexpr int kPlanTile = kGnTile;      // == kTile of dfh_solve.hip: a row never spans two tiles
constexpr int kPlanWaves = kPlanTile / 64;
constexpr int kKMaxP = 8;

__device__ __forceinline__ int plan_find_block(const int *__restrict__ row_ptr, const int *__restrict__ col, int a, int b) {
    int lo = row_ptr[a], hi = row_ptr[a + 1] - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int c = col[mid];
        if (c == b) return mid;
        if (c < b) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// ---- samples sorted by node tuple ------------------------------------------------------------------------------
// key = the tuple read as a k-digit number in base N (lexicographic order of the tuples), value = sample index
__global__ __launch_bounds__(256) void plan_pack_keys_kernel(const int *__restrict__ nbr, int S, int k, unsigned long long N,
                                                              unsigned long long *__restrict__ key, int *__restrict__ idx) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    unsigned long long kk = 0;
    for (int j = 0; j < k; ++j) kk = kk * N + (unsigned long long)nbr[(size_t)s * k + j];
    key[s] = kk;
    idx[s] = s;
}

__global__ __launch_bounds__(256) void plan_permute_kernel(const int *__restrict__ order, int S, int k, const double *__restrict__ pos,
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

// ---- rows ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool plan_is_head(const int *__restrict__ nbr, int s, int k) {
    if ((s & (kPlanTile - 1)) == 0) return true;          // a row never spans two tiles
    bool h = false;
    for (int j = 0; j < k; ++j) h = h || nbr[(size_t)s * k + j] != nbr[(size_t)(s - 1) * k + j];
    return h;
}

// rows per tile
__global__ __launch_bounds__(kPlanTile) void plan_heads_kernel(const int *__restrict__ nbr, int S, int k, int *__restrict__ tile_rows) {
    __shared__ int cnt[kPlanWaves];
    const int s = blockIdx.x * kPlanTile + threadIdx.x;
    const bool head = s < S && plan_is_head(nbr, s, k);
    const unsigned long long b = __ballot(head);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int w = 0; w < kPlanWaves; ++w) n += cnt[w];
        tile_rows[blockIdx.x] = n;
    }
}

// exclusive scan of the tile counts in place (one workgroup; tile_rows[n_tiles] = total = *n_rows_out)
__global__ __launch_bounds__(1024) void plan_scan_kernel(int *__restrict__ tile_rows, int n_tiles, int *__restrict__ n_rows_out) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_tiles ? tile_rows[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (i < n_tiles) tile_rows[i] = off + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) { tile_rows[n_tiles] = carry_s; *n_rows_out = carry_s; }
}

// run_id of every sample and the first sample of every row
__global__ __launch_bounds__(kPlanTile) void plan_runid_kernel(const int *__restrict__ nbr, int S, int k, const int *__restrict__ tile_off,
                                                          int *__restrict__ run_id, int *__restrict__ row_first) {
    __shared__ int cnt[kPlanWaves];
    const int s = blockIdx.x * kPlanTile + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool head = s < S && plan_is_head(nbr, s, k);
    const unsigned long long b = __ballot(head);
    if (lane == 0) cnt[wv] = __popcll(b);
    __syncthreads();
    int r = tile_off[blockIdx.x] + __popcll(b & ((2ull << lane) - 1ull)) - 1;        // heads up to and including this lane
    for (int w = 0; w < wv; ++w) r += cnt[w];
    if (s < S) {
        run_id[s] = r;
        if (head) row_first[r] = s;
    }
}

// ---- list keys -----------------------------------------------------------------------------------------------------
// entry e = row * k^2 + sa * k + sb -> key = index of block (node[sa], node[sb]) in the pattern (B if it is not there);
// entries with sa == 0 also emit the node entry row * k + sb -> key = node[sb]
// One returning atomic per DISTINCT key of a wave instead of one per lane: the lanes that hold the same key elect the lowest of
// them, it adds the group's size to the key's count, every lane takes base + its rank inside the group.  The counts are the same;
// the slots inside a list come out in another order, which does not matter (the lists are sorted afterwards).  Rows are sorted by
// node tuple, and a wave holds 64 consecutive rows of ONE slot pair (see the kernel), so long stretches share their key: the
// hottest counters (a node's diagonal block, its node list: ~200 entries) receive a handful of atomics instead of ~200 that
// serialise on one address at the memory side.
__device__ __forceinline__ int plan_count_slot(int *__restrict__ cnt, int key, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    int leader = lane, rank = 0, size = 0;
    while (todo) {                                               // (wave-uniform loop over the distinct keys)
        const int l0 = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(key, l0, 64);
        const unsigned long long same = __ballot(active && key == k0) & todo;
        if (active && key == k0) {
            leader = l0;
            rank = __popcll(same & ((1ull << lane) - 1ull));
            size = __popcll(same);
        }
        todo &= ~same;
    }
    int base = 0;
    if (active && leader == lane) base = atomicAdd(cnt + key, size);
    base = __shfl(base, leader, 64);
    return base + rank;
}

// Launch: blockIdx.y = the slot pair pr = sa * k + sb, blockIdx.x * 256 + threadIdx.x = the row.
__global__ __launch_bounds__(256) void plan_keys_kernel(const int *__restrict__ nbr, const int *__restrict__ row_first, int n_rows, int k,
                                                         const int *__restrict__ row_ptr, const int *__restrict__ col, int B,
                                                         int *__restrict__ blk_key, int *__restrict__ blk_val, int *__restrict__ node_key,
                                                         int *__restrict__ node_val, int *__restrict__ uncovered,
                                                         int *__restrict__ blk_cnt, int *__restrict__ node_cnt) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int pr = blockIdx.y, kk = k * k;
    const int sa = pr / k, sb = pr - sa * k;
    const bool in = row < n_rows;
    int blk = B, nb = 0;
    if (in) {
        const size_t s0 = (size_t)row_first[row] * k;
        const int na = nbr[s0 + sa];
        nb = nbr[s0 + sb];
        blk = plan_find_block(row_ptr, col, na, nb);
        if (blk < 0) { blk = B; atomicOr(uncovered, 1); }
    }
    // counting pass of the lists: the count's old value (+ the rank inside the wave's group) is this entry's slot inside its
    // list (any order: the lists are sorted afterwards, each on its own)
    const int slot = plan_count_slot(blk_cnt, blk, in && blk < B);
    if (in) {
        const size_t e = (size_t)row * kk + pr;
        blk_key[e] = blk;
        blk_val[e] = blk < B ? slot : 0;
    }
    if (sa == 0) {                                               // (uniform per workgroup)
        const int nslot = plan_count_slot(node_cnt, nb, in);
        if (in) {
            node_key[(size_t)row * k + sb] = nb;
            node_val[(size_t)row * k + sb] = nslot;
        }
    }
}

// zero two integer ranges in one launch
__global__ __launch_bounds__(256) void plan_zero2_kernel(int *__restrict__ a, int na, int *__restrict__ b, int nb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < na) a[i] = 0;
    else if (i - na < nb) b[i - na] = 0;
}

__global__ void plan_flag_out_kernel(const int *__restrict__ flag_in, int *__restrict__ flag_out) { *flag_out = *flag_in; }

// exclusive scans of the two count arrays in place (workgroup 0: a[0..na), total to a[na]; workgroup 1: b likewise).
// Every thread takes a contiguous chunk of up to 16 counts (one pass for up to 16 K lists, three barriers; more lists: rounds).
// flag_in -> *flag_out: the "pattern must grow" flag of the launch before (set there by atomics, in the workspace) goes to the
// caller's word by ONE plain store, so that word may be pinned host memory the host spins on.
__global__ __launch_bounds__(1024) void plan_scan2_kernel(int *__restrict__ a, int na, int *__restrict__ b, int nb,
                                                           const int *__restrict__ flag_in, int *__restrict__ flag_out) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *flag_out = *flag_in;
    int *v_ = blockIdx.x == 0 ? a : b;
    const int n = blockIdx.x == 0 ? na : nb;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    constexpr int C = 16;
    for (int base = 0; base < n; base += 1024 * C) {
        const int per = min(C, (min(n - base, 1024 * C) + 1023) / 1024);       // counts per thread in this round
        const int i0 = base + (int)threadIdx.x * per;
        int v[C];
        int sum = 0;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            v[j] = (j < per && i0 + j < n) ? v_[i0 + j] : 0;
            sum += v[j];
        }
        int x = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wv] = x;
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        int run = off + x - sum;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            if (j < per && i0 + j < n) v_[i0 + j] = run;
            run += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) v_[n] = carry_s;
}

// every entry to its slot: ent[ptr[key] + rank]
__global__ __launch_bounds__(256) void plan_fill_kernel(const int *__restrict__ blk_key, const int *__restrict__ blk_rank, long E, int B,
                                                         const int *__restrict__ blk_ptr, int *__restrict__ blk_ent,
                                                         const int *__restrict__ node_key, const int *__restrict__ node_rank, long E2,
                                                         const int *__restrict__ node_ptr, int *__restrict__ node_ent) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < E) {
        const int blk = blk_key[e];
        if (blk < B) blk_ent[blk_ptr[blk] + blk_rank[e]] = (int)e;
    }
    if (e < E2) node_ent[node_ptr[node_key[e]] + node_rank[e]] = (int)e;
}

// Every list sorted ascending (= the order a stable sort of the entries by key gives, the fixed summation order of the
// bit-reproducible build): one WAVE per list, lists [0, B) = blocks, [B, B + N) = nodes.  Up to 64 entries: a bitonic network
// on registers; up to kListLds: the same in the wave's LDS; longer (few nodes and very many rows): every entry's rank counted
// against a copy of the list in `tmp` (quadratic, a fallback).
constexpr int kListLds = 2048;
__global__ __launch_bounds__(256) void plan_sort_lists_kernel(const int *__restrict__ blk_ptr, int *__restrict__ blk_ent, int B,
                                                               const int *__restrict__ node_ptr, int *__restrict__ node_ent, int N,
                                                               int *__restrict__ blk_tmp, int *__restrict__ node_tmp) {
    __shared__ int sl[4][kListLds];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = blockIdx.x * 4 + wv;
    if (L >= B + N) return;
    const bool isb = L < B;
    const int id = isb ? L : L - B;
    const int *ptr = isb ? blk_ptr : node_ptr;
    int *ent = isb ? blk_ent : node_ent;
    int *tmp = isb ? blk_tmp : node_tmp;
    const int beg = ptr[id], n = ptr[id + 1] - beg;
    if (n <= 1) return;
    constexpr int kBig = 0x7fffffff;
    if (n <= 64) {
        int v = lane < n ? ent[beg + lane] : kBig;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int o = __shfl_xor(v, j, 64);
                const bool up = (lane & k) == 0, low = (lane & j) == 0;
                v = (low == up) ? min(v, o) : max(v, o);
            }
        if (lane < n) ent[beg + lane] = v;
        return;
    }
    if (n <= kListLds) {
        int P = 128;
        while (P < n) P <<= 1;
        int *a = sl[wv];
        for (int i = lane; i < P; i += 64) a[i] = i < n ? ent[beg + i] : kBig;
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
                for (int t = lane; t < P / 2; t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // t with a 0 inserted at bit log2(j)
                    const int x = a[i], y = a[i | j];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[i | j] = x; }
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        for (int i = lane; i < n; i += 64) ent[beg + i] = a[i];
        return;
    }
    for (int i = lane; i < n; i += 64) tmp[beg + i] = ent[beg + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");              // (the wave reads other lanes' copies below)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int i = lane; i < n; i += 64) {
        const int x = tmp[beg + i];
        int r = 0;
        for (int q = 0; q < n; ++q) r += tmp[beg + q] < x ? 1 : 0;
        ent[beg + r] = x;                                           // entries are distinct: ranks are a permutation
    }
}

// values 0..n-1 of the two radix sorts (A/B path)
__global__ __launch_bounds__(256) void plan_iota_kernel(int *__restrict__ a, long na, int *__restrict__ b, long nb) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < na) a[i] = (int)i;
    if (i < nb) b[i] = (int)i;
}

// ptr[v] = first position of a key >= v in the sorted keys, v = 0..n_keys (CSR offsets of the lists)
__global__ __launch_bounds__(256) void plan_ptr_kernel(const int *__restrict__ sorted, int n, int n_keys, int *__restrict__ ptr) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v > n_keys) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    ptr[v] = lo;
}

static unsigned bits_for(unsigned long long max_value) {
    unsigned b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

static size_t align16(size_t x) { return (x + 15) / 16 * 16; }

static size_t sort_temp_bytes_u64(int n) {
    size_t t = 0;
    if (rocprim::radix_sort_pairs<SortCfg>(nullptr, t, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int *)nullptr, (int *)nullptr,
                                  (size_t)n) != hipSuccess) t = 0;
    return t;
}
static size_t sort_temp_bytes_i32(long n) {
    size_t t = 0;
    if (rocprim::radix_sort_pairs<SortCfg>(nullptr, t, (int *)nullptr, (int *)nullptr, (int *)nullptr, (int *)nullptr, (size_t)n) != hipSuccess) t = 0;
    return t;
}

}  // namespace dfh

extern "C" {

size_t dfh_gn_sort_workspace_bytes(int n_samples) {
    using namespace dfh;
    if (n_samples <= 0) return 0;
    return align16(sizeof(unsigned long long) * (size_t)n_samples) + align16(sizeof(int) * (size_t)n_samples) +
           align16(sort_temp_bytes_u64(n_samples));
}

int dfh_gn_sort_samples(const double *pos, const double *nrm, const int *nbr, const double *weights, int n_samples, int knn,
                        int n_nodes, double *pos_out, double *nrm_out, int *nbr_out, double *weights_out, long *key_out,
                        int *order_out, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_samples >= 0 && knn >= 1 && knn <= kKMaxP && n_nodes >= 1, "dfh_gn_sort_samples: bad sizes");
    if (n_samples == 0) return DFH_OK;
    DFH_REQUIRE(pos && nrm && nbr && weights && pos_out && nrm_out && nbr_out && weights_out && key_out && order_out && workspace,
                "dfh_gn_sort_samples: null pointer");
    DFH_REQUIRE(workspace_bytes >= dfh_gn_sort_workspace_bytes(n_samples), "dfh_gn_sort_samples: workspace too small");
    double span = 1.0;
    for (int j = 0; j < knn; ++j) span *= (double)n_nodes;
    DFH_REQUIRE(span < 9.0e18, "dfh_gn_sort_samples: %d digits in base %d do not fit 64 bits", knn, n_nodes);
    hipStream_t s = (hipStream_t)stream;
    char *w = static_cast<char *>(workspace);
    unsigned long long *key_in = reinterpret_cast<unsigned long long *>(w); w += align16(sizeof(unsigned long long) * (size_t)n_samples);
    int *idx_in = reinterpret_cast<int *>(w); w += align16(sizeof(int) * (size_t)n_samples);
    int *order = order_out;
    size_t temp = sort_temp_bytes_u64(n_samples);
    const dim3 grid((unsigned)((n_samples + 255) / 256)), block(256);
    hipLaunchKernelGGL(plan_pack_keys_kernel, grid, block, 0, s, nbr, n_samples, knn, (unsigned long long)n_nodes, key_in, idx_in);
    const unsigned end_bit = bits_for((unsigned long long)(span - 1.0) + 1ull);
    DFH_HIP_CHECK(rocprim::radix_sort_pairs<SortCfg>(w, temp, key_in, reinterpret_cast<unsigned long long *>(key_out), idx_in, order,
                                            (size_t)n_samples, 0u, end_bit > 64u ? 64u : end_bit, s));
    hipLaunchKernelGGL(plan_permute_kernel, grid, block, 0, s, order, n_samples, knn, pos, nrm, nbr, weights, pos_out, nrm_out, nbr_out,
                       weights_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

int dfh_gn_tile_samples(void) { return dfh::kGnTile; }

int dfh_gn_plan_count(const int *nbr, int n_samples, int knn, int *tile_off, int *n_rows_out, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_samples >= 1 && knn >= 1 && knn <= kKMaxP, "dfh_gn_plan_count: bad sizes");
    DFH_REQUIRE(nbr && tile_off && n_rows_out, "dfh_gn_plan_count: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = (n_samples + kPlanTile - 1) / kPlanTile;
    hipLaunchKernelGGL(plan_heads_kernel, dim3(n_tiles), dim3(kPlanTile), 0, s, nbr, n_samples, knn, tile_off);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, s, tile_off, n_tiles, n_rows_out);
    DFH_HIP_CHECK(hipGetLastError());
    return DFH_OK;
}

size_t dfh_gn_plan_workspace_bytes(int n_rows, int knn) {
    using namespace dfh;
    if (n_rows <= 0 || knn < 1 || knn > kKMaxP) return 0;
    const size_t E = (size_t)n_rows * knn * knn, E2 = (size_t)n_rows * knn;
    return 3 * align16(sizeof(int) * E) + 3 * align16(sizeof(int) * E2) + align16(sort_temp_bytes_i32((long)E)) + 16;
}

int dfh_gn_plan_build(const int *nbr, int n_samples, int knn, int n_nodes, const int *tile_off, int n_rows, const int *row_ptr,
                      const int *col, int n_blocks, int *run_id, int *row_first, int *blk_ptr, int *blk_ent, int *node_ptr,
                      int *node_ent, int *uncovered_out, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace dfh;
    DFH_REQUIRE(n_samples >= 1 && knn >= 1 && knn <= kKMaxP && n_nodes >= 1 && n_blocks >= 1 && n_rows >= 1, "dfh_gn_plan_build: bad sizes");
    DFH_REQUIRE(nbr && tile_off && row_ptr && col && run_id && row_first && blk_ptr && blk_ent && node_ptr && node_ent && uncovered_out &&
                workspace, "dfh_gn_plan_build: null pointer");
    DFH_REQUIRE(workspace_bytes >= dfh_gn_plan_workspace_bytes(n_rows, knn), "dfh_gn_plan_build: workspace too small");
    DFH_REQUIRE((long)n_rows * knn * knn < (1L << 31), "dfh_gn_plan_build: too many rows for 32-bit entries");
    hipStream_t s = (hipStream_t)stream;
    const size_t E = (size_t)n_rows * knn * knn, E2 = (size_t)n_rows * knn;
    char *w = static_cast<char *>(workspace);
    int *bk_in = reinterpret_cast<int *>(w); w += align16(sizeof(int) * E);
    int *bv_in = reinterpret_cast<int *>(w); w += align16(sizeof(int) * E);
    int *bk_out = reinterpret_cast<int *>(w); w += align16(sizeof(int) * E);
    int *nk_in = reinterpret_cast<int *>(w); w += align16(sizeof(int) * E2);
    int *nv_in = reinterpret_cast<int *>(w); w += align16(sizeof(int) * E2);
    int *nk_out = reinterpret_cast<int *>(w); w += align16(sizeof(int) * E2);
    size_t temp = sort_temp_bytes_i32((long)E);
    int *unc_ws = reinterpret_cast<int *>(w + align16(temp));     // (the 16 spare bytes of the workspace) the flag while atomics set it
    const int n_tiles = (n_samples + kPlanTile - 1) / kPlanTile;
    DFH_HIP_CHECK(hipMemsetAsync(unc_ws, 0, sizeof(int), s));
    hipLaunchKernelGGL(plan_runid_kernel, dim3(n_tiles), dim3(kPlanTile), 0, s, nbr, n_samples, knn, tile_off, run_id, row_first);
    if (on(opt().plan_radix)) {              // the lists through two stable device radix sorts (round 2, first half): kept for A/B
        hipLaunchKernelGGL(plan_zero2_kernel, dim3((unsigned)((n_blocks + n_nodes + 2 + 255) / 256)), dim3(256), 0, s, blk_ptr, n_blocks + 1,
                           node_ptr, n_nodes + 1);
        hipLaunchKernelGGL(plan_keys_kernel, dim3((unsigned)((n_rows + 255) / 256), (unsigned)(knn * knn)), dim3(256), 0, s, nbr, row_first, n_rows, knn,
                           row_ptr, col, n_blocks, bk_in, bk_out, nk_in, nk_out, unc_ws, blk_ptr, node_ptr);
        hipLaunchKernelGGL(plan_flag_out_kernel, dim3(1), dim3(1), 0, s, unc_ws, uncovered_out);
        hipLaunchKernelGGL(plan_iota_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, bv_in, (long)E, nv_in, (long)E2);
        DFH_HIP_CHECK(hipGetLastError());
        DFH_HIP_CHECK(rocprim::radix_sort_pairs<SortCfg>(w, temp, bk_in, bk_out, bv_in, blk_ent, E, 0u, bits_for((unsigned long long)n_blocks + 1ull), s));
        size_t temp2 = sort_temp_bytes_i32((long)E2);
        DFH_REQUIRE(temp2 <= temp, "dfh_gn_plan_build: scratch of the node sort exceeds the block sort's");
        DFH_HIP_CHECK(rocprim::radix_sort_pairs<SortCfg>(w, temp2, nk_in, nk_out, nv_in, node_ent, E2, 0u, bits_for((unsigned long long)n_nodes), s));
        hipLaunchKernelGGL(plan_ptr_kernel, dim3((unsigned)((n_blocks + 1 + 255) / 256)), dim3(256), 0, s, bk_out, (int)E, n_blocks, blk_ptr);
        hipLaunchKernelGGL(plan_ptr_kernel, dim3((unsigned)((n_nodes + 1 + 255) / 256)), dim3(256), 0, s, nk_out, (int)E2, n_nodes, node_ptr);
        DFH_HIP_CHECK(hipGetLastError());
        return DFH_OK;
    }
    // The lists by counting: every entry takes a slot in its list while the lists are counted (atomic increments: any order),
    // an exclusive scan turns the counts into offsets, the entries go to offset + slot, and every list is then sorted on its
    // own by one wave (lists are short: ~36 entries per block, ~200 per node).  Six launches where the two library sorts
    // took 26; the result is the same as a stable sort of the entries by key, element for element.
    hipLaunchKernelGGL(plan_zero2_kernel, dim3((unsigned)((n_blocks + n_nodes + 2 + 255) / 256)), dim3(256), 0, s, blk_ptr, n_blocks + 1,
                       node_ptr, n_nodes + 1);
    hipLaunchKernelGGL(plan_keys_kernel, dim3((unsigned)((n_rows + 255) / 256), (unsigned)(knn * knn)), dim3(256), 0, s, nbr, row_first, n_rows, knn,
                       row_ptr, col, n_blocks, bk_in, bv_in, nk_in, nv_in, unc_ws, blk_ptr, node_ptr);
    hipLaunchKernelGGL(plan_scan2_kernel, dim3(2), dim3(1024), 0, s, blk_ptr, n_blocks, node_ptr, n_nodes, unc_ws, uncovered_out);
    hipLaunchKernelGGL(plan_fill_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, bk_in, bv_in, (long)E, n_blocks, blk_ptr, blk_ent,
                       nk_in, nv_in, (long)E2, node_ptr, node_ent);
    hipLaunchKernelGGL(plan_sort_lists_kernel, dim3((unsigned)((n_blocks + n_nodes + 3) / 4)), dim3(256), 0, s, blk_ptr, blk_ent, n_blocks,
                       node_ptr, node_ent, n_nodes, bk_out, nk_out);
    DFH_HIP_CHECK(hipGetLastError());
    (void)temp;
    return DFH_OK;
}

}  // extern "C"
