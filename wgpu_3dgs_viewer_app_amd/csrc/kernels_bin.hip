// kernels_bin.hip — tile binning for gfx950: per-splat tile counts in depth order, exclusive scan,
// duplicate emission of (tile id, Gaussian index) pairs, and per-tile ranges of the tile-sorted list.
//
// The reference has no screen tiles (it draws one instanced quad per surviving Gaussian and lets the
// ROP blend, src/tab/scene.rs:2306-2313); this stage is the build's replacement for that rasteriser
// front-end.  Pairs are emitted in front-to-back depth order, so a STABLE sort by tile id alone
// (kernels_sort.hip, 2 passes for <= 65536 tiles) leaves every tile's list depth-ordered.
// Integer-only; must be bit-exact against oracle/gsx_oracle.c:gsxo_tile_lists.
// Algorithmic bytes: N_vis*44 + D*12.
#include "gsx_internal.h"

namespace gsx {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanThreads * kScanItems;  // 4096 entries per workgroup

size_t scan_blocks(uint64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile); }

// Multi-GPU: a rank bins only the tile rows it owns (row % world == rank); world = 1 owns every row.
__device__ inline uint32_t first_owned_row(uint32_t y0, uint32_t world, uint32_t rank) {
    return y0 + ((rank + world - (y0 % world)) % world);
}

__device__ inline uint32_t rect_area(float4 a, uint32_t world, uint32_t rank) {
    uint32_t rx = __float_as_uint(a.z), ry = __float_as_uint(a.w);
    uint32_t y0 = ry & 0xFFFFu, y1 = ry >> 16;
    uint32_t first = first_owned_row(y0, world, rank);
    uint32_t rows = first < y1 ? (y1 - 1u - first) / world + 1u : 0u;
    return ((rx >> 16) - (rx & 0xFFFFu)) * rows;
}

__device__ inline uint32_t block_reduce_sum(uint32_t v, uint32_t* smem4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63u) == 0) smem4[threadIdx.x >> 6] = v;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// tiles touched by the j-th splat in depth order -> cnt[j]; per-workgroup sums -> block_sums
__global__ __launch_bounds__(kScanThreads) void k_tile_counts(const uint32_t* __restrict__ d_n_vis,
                                                               const uint32_t* __restrict__ sorted_idx,
                                                               const float4* __restrict__ rec_a,
                                                               uint32_t* __restrict__ cnt,
                                                               uint32_t* __restrict__ block_sums, uint32_t world,
                                                               uint32_t rank) {
    __shared__ uint32_t red[4];
    uint32_t sum = 0;
    const uint32_t n_vis = *d_n_vis;
    const uint32_t base = blockIdx.x * kScanTile;
#pragma unroll 4
    for (int r = 0; r < kScanItems; ++r) {
        uint32_t j = base + r * kScanThreads + threadIdx.x;
        if (j < n_vis) {
            uint32_t c = rect_area(rec_a[sorted_idx[j]], world, rank);
            cnt[j] = c;
            sum += c;
        }
    }
    uint32_t tot = block_reduce_sum(sum, red);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of block_sums in place, grand total -> *d_total
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* __restrict__ sums, uint32_t nblocks,
                                                           uint32_t* __restrict__ d_total) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        uint32_t i = base + tid;
        uint32_t v = i < nblocks ? sums[i] : 0u, x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wsum[w];
        uint32_t carry = carry_s;
        if (i < nblocks) sums[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) *d_total = carry_s;
}

// emit (tile id, Gaussian index) pairs for the j-th splat at offset = block_sums[wg] + local exclusive scan
__global__ __launch_bounds__(kScanThreads) void k_tile_emit(uint32_t n_vis, const uint32_t* __restrict__ sorted_idx,
                                                             const float4* __restrict__ rec_a,
                                                             const uint32_t* __restrict__ cnt,
                                                             const uint32_t* __restrict__ block_offs, uint32_t tiles_x,
                                                             uint32_t* __restrict__ tkey, uint32_t* __restrict__ tval,
                                                             uint32_t world, uint32_t rank) {
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // each lane owns kScanItems CONSECUTIVE entries so the local scan is a serial prefix + one block scan
    const uint32_t j0 = blockIdx.x * kScanTile + tid * kScanItems;
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < kScanItems; ++r) {
        uint32_t j = j0 + r;
        mine += j < n_vis ? cnt[j] : 0u;
    }
    uint32_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t off = block_offs[blockIdx.x] + x - mine;
    for (uint32_t w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll 1
    for (int r = 0; r < kScanItems; ++r) {
        uint32_t j = j0 + r;
        if (j >= n_vis) break;
        const uint32_t cr = cnt[j];  // re-read (L1/L2 hit) instead of a runtime-indexed register array
        if (cr == 0) continue;
        uint32_t idx = sorted_idx[j];
        float4 a = rec_a[idx];
        uint32_t rx = __float_as_uint(a.z), ry = __float_as_uint(a.w);
        uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = ry & 0xFFFFu, y1 = ry >> 16;
        uint32_t o = off;
        for (uint32_t ty = first_owned_row(y0, world, rank); ty < y1; ty += world)
            for (uint32_t tx = x0; tx < x1; ++tx) {
                tkey[o] = ty * tiles_x + tx;
                tval[o] = idx;
                ++o;
            }
        off += cr;
    }
}

// ranges[t] = [first, last+1) of tile t in the tile-sorted pair list (ranges pre-zeroed)
__global__ __launch_bounds__(256) void k_tile_ranges(uint32_t D, const uint32_t* __restrict__ tkey,
                                                      uint2* __restrict__ ranges) {
    uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= D) return;
    uint32_t t = tkey[e];
    if (e == 0 || tkey[e - 1] != t) ranges[t].x = e;
    if (e == D - 1 || tkey[e + 1] != t) ranges[t].y = e + 1;
}

hipError_t launch_tile_counts(hipStream_t s, uint32_t n_upper, const uint32_t* d_n_vis, const uint32_t* sorted_idx,
                              const Records& rec, uint32_t* cnt, uint32_t* block_sums, uint32_t* d_total, uint32_t world,
                              uint32_t rank) {
    uint32_t nb = (uint32_t)scan_blocks(n_upper);
    if (nb)
        hipLaunchKernelGGL(k_tile_counts, dim3(nb), dim3(kScanThreads), 0, s, d_n_vis, sorted_idx, rec.a, cnt, block_sums,
                           world, rank);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nb, d_total);
    return hipGetLastError();
}

hipError_t launch_tile_emit(hipStream_t s, uint32_t n_vis, const uint32_t* sorted_idx, const Records& rec,
                            const uint32_t* cnt, const uint32_t* block_sums, uint32_t tiles_x, uint32_t* tkey,
                            uint32_t* tval, uint32_t world, uint32_t rank) {
    uint32_t nb = (uint32_t)scan_blocks(n_vis);
    if (!nb) return hipSuccess;
    hipLaunchKernelGGL(k_tile_emit, dim3(nb), dim3(kScanThreads), 0, s, n_vis, sorted_idx, rec.a, cnt, block_sums,
                       tiles_x, tkey, tval, world, rank);
    return hipGetLastError();
}

hipError_t launch_tile_ranges(hipStream_t s, uint32_t D, const uint32_t* tkey_sorted, uint32_t n_tiles,
                              uint2* ranges) {
    hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)n_tiles, s);
    if (e != hipSuccess) return e;
    if (D == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_ranges, dim3((D + 255) / 256), dim3(256), 0, s, D, tkey_sorted, ranges);
    return hipGetLastError();
}

}  // namespace gsx
