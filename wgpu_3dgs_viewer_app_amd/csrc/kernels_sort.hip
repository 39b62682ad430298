// kernels_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 (wave64).
//
// Replaces the reference's K2 (`radix_sorter.sort(encoder, bind_group, indirect_args)`,
// src/tab/scene.rs:865-869: key = depth, value = Gaussian index) and also orders the tile-binning
// pairs (key = tile id).  Integer-only, HBM-bound: per 8-bit pass each element is read twice as a key
// (histogram + scatter), once as a value, and written once: 20 B/element/pass.
//
// Structure per pass (three launches, no inter-workgroup spinning):
//   k_radix_hist    each workgroup histograms its 4096-element tile   -> table[digit][workgroup]
//   k_radix_rowscan one workgroup per digit scans its table row        -> exclusive offsets + digit totals
//   k_radix_scatter re-reads the tile, ranks every element stably inside the workgroup with
//                   wave-level digit matching (8 ballots), and scatters to its final position.
// Stability (and therefore determinism, ties broken by input order) comes from the ranking order
// (wave, round, lane) == memory order inside a tile, tiles in workgroup order, digits in table order.
#include <algorithm>

#include "gsx_internal.h"

namespace gsx {

constexpr int kRadixThreads = 256;
constexpr int kRadixWaves = kRadixThreads / 64;
constexpr int kRadixRounds = 16;                                    // elements per lane
constexpr int kRadixTile = kRadixThreads * kRadixRounds;            // 4096 elements per workgroup
constexpr int kWaveChunk = 64 * kRadixRounds;                       // 1024 contiguous elements per wave

constexpr uint32_t kRadixGrid = 1024;  // workgroups per pass at most (4 per CU); fixes the histogram table size

static inline uint32_t radix_blocks(uint64_t n) { return (uint32_t)((n + kRadixTile - 1) / kRadixTile); }
size_t radix_table_entries(uint64_t) { return (size_t)256 * kRadixGrid + 256; }

// lanes of this wave holding the same 8-bit digit (among `valid` lanes)
__device__ inline unsigned long long wave_match8(uint32_t digit, bool valid) {
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        bool bit = (digit >> b) & 1u;
        unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

__device__ inline unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63u)) - 1ull;
}

// Work split: the ceil(n/4096) tiles are dealt to the workgroups in CONTIGUOUS runs (workgroup b owns tiles
// [b*tpb, (b+1)*tpb)), so the grid and the histogram table have a fixed size (<= kRadixGrid) that does not
// depend on n.  That lets n live on the device (d_n, the tile-pair count of a depth slab) with no host
// round trip: n_cap only bounds the launch.
__device__ inline void radix_my_tiles(uint32_t n, uint32_t& t0, uint32_t& t1) {
    const uint32_t tiles = (n + kRadixTile - 1) / kRadixTile;
    const uint32_t tpb = (tiles + gridDim.x - 1) / gridDim.x;
    t0 = min(blockIdx.x * tpb, tiles);
    t1 = min(t0 + tpb, tiles);
}

__global__ __launch_bounds__(kRadixThreads) void k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n_cap,
                                                               const uint32_t* __restrict__ d_n, int shift,
                                                               uint32_t* __restrict__ table) {
    __shared__ uint32_t hist[256];
    const uint32_t n = d_n ? min(*d_n, n_cap) : n_cap;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    hist[tid] = 0;
    __syncthreads();
    uint32_t t0, t1;
    radix_my_tiles(n, t0, t1);
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t base = t * kRadixTile + wave * kWaveChunk;
#pragma unroll 4
        for (int r = 0; r < kRadixRounds; ++r) {
            uint32_t e = base + r * 64 + lane;
            bool valid = e < n;
            uint32_t digit = valid ? (keys[e] >> shift) & 255u : 0u;
            unsigned long long m = wave_match8(digit, valid);
            // the lowest lane of every digit group adds the group's population
            if (valid && (m & lanemask_lt()) == 0) atomicAdd(&hist[digit], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    table[tid * gridDim.x + blockIdx.x] = hist[tid];
}

// one workgroup per digit: exclusive scan of table[digit][0..nblocks) in place; total -> totals[digit]
__global__ __launch_bounds__(256) void k_radix_rowscan(uint32_t* __restrict__ table, uint32_t nblocks,
                                                        uint32_t* __restrict__ totals) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    uint32_t* row = table + (size_t)blockIdx.x * nblocks;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 256) {
        uint32_t i = base + tid;
        uint32_t v = i < nblocks ? row[i] : 0u;
        // inclusive wave scan
        uint32_t x = v;
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
        if (i < nblocks) row[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == 255) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry_s;
}

template <bool IOTA>
__global__ __launch_bounds__(kRadixThreads) void k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                                  const uint32_t* __restrict__ vals_in,
                                                                  uint32_t* __restrict__ keys_out,
                                                                  uint32_t* __restrict__ vals_out, uint32_t n_cap,
                                                                  const uint32_t* __restrict__ d_n, int shift,
                                                                  const uint32_t* __restrict__ table,
                                                                  const uint32_t* __restrict__ totals) {
    __shared__ uint32_t cnt[kRadixWaves][256];  // per-wave digit counters, then absolute output offsets
    __shared__ uint32_t wtot[4];
    const uint32_t n = d_n ? min(*d_n, n_cap) : n_cap;
    uint32_t t0, t1;
    radix_my_tiles(n, t0, t1);
    if (t0 >= t1) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    // thread = digit: running output offset of this digit for this workgroup
    uint32_t run;
    {   // exclusive scan of the 256 digit totals (each workgroup redoes this tiny scan)
        uint32_t v = totals[tid], x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wtot[w];
        run = woff + x - v + table[tid * gridDim.x + blockIdx.x];
    }
    volatile uint32_t* mycnt = cnt[wave];
    for (uint32_t t = t0; t < t1; ++t) {
#pragma unroll
        for (int w = 0; w < kRadixWaves; ++w) cnt[w][tid] = 0;
        __syncthreads();
        const uint32_t base = t * kRadixTile + wave * kWaveChunk;
        uint32_t key[kRadixRounds], val[kRadixRounds], rank[kRadixRounds];
#pragma unroll
        for (int r = 0; r < kRadixRounds; ++r) {
            uint32_t e = base + r * 64 + lane;
            bool valid = e < n;
            key[r] = valid ? keys_in[e] : 0xFFFFFFFFu;
            val[r] = valid ? (IOTA ? e : vals_in[e]) : 0u;
            uint32_t digit = (key[r] >> shift) & 255u;
            unsigned long long m = wave_match8(digit, valid);
            uint32_t before = (uint32_t)__popcll(m & lanemask_lt());
            uint32_t old = valid ? mycnt[digit] : 0u;   // every lane of a digit group reads the same counter
            rank[r] = old + before;
            __builtin_amdgcn_wave_barrier();
            if (valid && before == 0) mycnt[digit] = old + (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        {   // thread = digit: per-wave counts -> absolute output offsets; advance the running offset
#pragma unroll
            for (int w = 0; w < kRadixWaves; ++w) {
                uint32_t c = cnt[w][tid];
                cnt[w][tid] = run;
                run += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRadixRounds; ++r) {
            uint32_t e = base + r * 64 + lane;
            if (e < n) {
                uint32_t digit = (key[r] >> shift) & 255u;
                uint32_t o = cnt[wave][digit] + rank[r];
                keys_out[o] = key[r];
                vals_out[o] = val[r];
            }
        }
        __syncthreads();
    }
}

hipError_t launch_rowscan(hipStream_t s, uint32_t* table, uint32_t nrows, uint32_t nblocks, uint32_t* totals) {
    if (nrows) hipLaunchKernelGGL(k_radix_rowscan, dim3(nrows), dim3(256), 0, s, table, nblocks, totals);
    return hipGetLastError();
}

hipError_t launch_radix_sort(hipStream_t s, const RadixBuffers& buf, uint32_t n, const uint32_t* d_n, int bits,
                             bool iota_values, bool* result_in_b) {
    *result_in_b = false;
    if (n == 0) return hipSuccess;
    const uint32_t nb = std::min<uint32_t>(kRadixGrid, radix_blocks(n));
    uint32_t* totals = buf.table + (size_t)256 * kRadixGrid;
    const int passes = (bits + 7) / 8;
    const uint32_t *kin = buf.keys_src, *vin = buf.vals_src;
    uint32_t *kout = buf.keys_a, *vout = buf.vals_a;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(kRadixThreads), 0, s, kin, n, d_n, shift, buf.table);
        hipLaunchKernelGGL(k_radix_rowscan, dim3(256), dim3(256), 0, s, buf.table, nb, totals);
        if (p == 0 && iota_values)
            hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nb), dim3(kRadixThreads), 0, s, kin, vin, kout, vout, n, d_n,
                               shift, buf.table, totals);
        else
            hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nb), dim3(kRadixThreads), 0, s, kin, vin, kout, vout, n, d_n,
                               shift, buf.table, totals);
        *result_in_b = (kout == buf.keys_b);
        kin = kout;
        vin = vout;
        kout = *result_in_b ? buf.keys_a : buf.keys_b;
        vout = *result_in_b ? buf.vals_a : buf.vals_b;
    }
    return hipGetLastError();
}

}  // namespace gsx
