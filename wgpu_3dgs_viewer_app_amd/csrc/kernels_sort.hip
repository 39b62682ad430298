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
#include "gsx_internal.h"

namespace gsx {

constexpr int kRadixThreads = 256;
constexpr int kRadixWaves = kRadixThreads / 64;
constexpr int kRadixRounds = 16;                                    // elements per lane
constexpr int kRadixTile = kRadixThreads * kRadixRounds;            // 4096 elements per workgroup
constexpr int kWaveChunk = 64 * kRadixRounds;                       // 1024 contiguous elements per wave

static inline uint32_t radix_blocks(uint64_t n) { return (uint32_t)((n + kRadixTile - 1) / kRadixTile); }
size_t radix_table_entries(uint64_t n) { return (size_t)256 * radix_blocks(n) + 256; }

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

__global__ __launch_bounds__(kRadixThreads) void k_radix_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                               uint32_t* __restrict__ table, uint32_t nblocks) {
    __shared__ uint32_t hist[256];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRadixTile + wave * kWaveChunk;
#pragma unroll 4
    for (int r = 0; r < kRadixRounds; ++r) {
        uint32_t e = base + r * 64 + lane;
        bool valid = e < n;
        uint32_t digit = valid ? (keys[e] >> shift) & 255u : 0u;
        unsigned long long m = wave_match8(digit, valid);
        // the lowest lane of every digit group adds the group's population
        if (valid && (m & lanemask_lt()) == 0) atomicAdd(&hist[digit], (uint32_t)__popcll(m));
    }
    __syncthreads();
    table[tid * nblocks + blockIdx.x] = hist[tid];
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
                                                                  uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                                  const uint32_t* __restrict__ table, uint32_t nblocks,
                                                                  const uint32_t* __restrict__ totals) {
    __shared__ uint32_t cnt[kRadixWaves][256];  // per-wave digit counters, later absolute output offsets
    __shared__ uint32_t dbase[256];             // exclusive scan of the digit totals
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
#pragma unroll
    for (int w = 0; w < kRadixWaves; ++w) cnt[w][tid] = 0;
    {   // exclusive scan of the 256 digit totals (each workgroup redoes this tiny scan)
        uint32_t v = totals[tid], x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        __shared__ uint32_t wtot[4];
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wtot[w];
        dbase[tid] = woff + x - v;
    }
    __syncthreads();

    const uint32_t base = blockIdx.x * kRadixTile + wave * kWaveChunk;
    uint32_t key[kRadixRounds], val[kRadixRounds], rank[kRadixRounds];
    volatile uint32_t* mycnt = cnt[wave];
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
    {   // thread = digit: turn the per-wave counts into absolute output offsets
        uint32_t run = dbase[tid] + table[tid * nblocks + blockIdx.x];
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
}

hipError_t launch_rowscan(hipStream_t s, uint32_t* table, uint32_t nrows, uint32_t nblocks, uint32_t* totals) {
    if (nrows) hipLaunchKernelGGL(k_radix_rowscan, dim3(nrows), dim3(256), 0, s, table, nblocks, totals);
    return hipGetLastError();
}

hipError_t launch_radix_sort(hipStream_t s, const RadixBuffers& buf, uint32_t n, int bits, bool iota_values,
                             bool* result_in_b) {
    *result_in_b = false;
    if (n == 0) return hipSuccess;
    const uint32_t nb = radix_blocks(n);
    uint32_t* totals = buf.table + (size_t)256 * nb;
    const int passes = (bits + 7) / 8;
    const uint32_t *kin = buf.keys_src, *vin = buf.vals_src;
    uint32_t *kout = buf.keys_a, *vout = buf.vals_a;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_radix_hist, dim3(nb), dim3(kRadixThreads), 0, s, kin, n, shift, buf.table, nb);
        hipLaunchKernelGGL(k_radix_rowscan, dim3(256), dim3(256), 0, s, buf.table, nb, totals);
        if (p == 0 && iota_values)
            hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nb), dim3(kRadixThreads), 0, s, kin, vin, kout, vout, n, shift,
                               buf.table, nb, totals);
        else
            hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nb), dim3(kRadixThreads), 0, s, kin, vin, kout, vout, n,
                               shift, buf.table, nb, totals);
        *result_in_b = (kout == buf.keys_b);
        kin = kout;
        vin = vout;
        kout = *result_in_b ? buf.keys_a : buf.keys_b;
        vout = *result_in_b ? buf.vals_a : buf.vals_b;
    }
    return hipGetLastError();
}

}  // namespace gsx
