// kernels_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 (wave64).
//
// Replaces the reference's K2 (`radix_sorter.sort(encoder, bind_group, indirect_args)`,
// src/tab/scene.rs:865-869: key = depth, value = Gaussian index) and also orders the tile-binning
// pairs (key = tile id).  Integer-only, HBM-bound.
//
// One launch per 8-bit digit ("onesweep" structure, written for CDNA4):
//   k_radix_global_hist   one read of the keys builds the digit histograms of ALL passes.
//   k_radix_onesweep      persistent workgroups take 8192-element tiles in ticket order; per tile:
//                         stable ranks by wave-level digit matching (8 ballots) -> per-digit tile counts
//                         -> published as one 64-bit {epoch|flag, count} word per (tile, digit);
//                         thread d looks back over the preceding tiles' words of digit d (decoupled
//                         look-back) to get the tile's global offset; the tile is reordered by digit in
//                         LDS and written out so that consecutive lanes hit consecutive addresses.
// Per pass every element is read once (8 B) and written once (8 B); intermediate passes move interleaved
// {key,value} pairs, the last pass writes split key / value arrays for the consumers.
// Inter-workgroup protocol (guide §6 G16, form R2 "the data IS the flag"): each status word is written by
// ONE relaxed agent-scope 64-bit atomic store and polled with relaxed agent-scope loads; nothing else is
// exchanged between workgroups, so no fence is needed.  Tickets guarantee that every tile a workgroup can
// wait for is already held by a running workgroup (no dependence on dispatch order or XCD placement).
// The status words carry an epoch (one per launch) so they never need clearing; the ticket counter is
// reset by the last workgroup to finish.
// The element count may live on the device (d_n: tile pairs of a depth slab) — no host round trip.
// Stability (ties keep input order) makes the depth order deterministic: ties break by Gaussian index.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "gsx_internal.h"

namespace gsx {

constexpr int kRadixThreads = 256;                        // histogram / scan kernels
constexpr int kRadixWaves = kRadixThreads / 64;
// The digit pass runs 512-thread workgroups (8 waves), one 8192-element tile each; threads 0..255 also own one digit each.
// What bounds a pass at millions of pairs is the chain of tile prefixes: ~40 tiles per microsecond whatever a tile holds
// (1024 / 2048 / 4096 / 8192 / 16384 elements per tile: 190 / 105 / 67 / 56 / 54 us per pass at 8.46 M pairs; a 16384-element
// tile needs 1024 threads and costs the small sorts 15 %).  tools/bench_sort.hip, round 3.
constexpr int kSweepThreads = 512;
constexpr int kSweepWaves = kSweepThreads / 64;
constexpr int kRadixRounds = 16;                          // elements per lane
constexpr int kRadixTile = kSweepThreads * kRadixRounds;  // 8192 elements per tile
// Small sorts (the ~0.3 M admitted pairs of a speculated frame, its repair round, the block lists): a few dozen 8192-element tiles
// leave most of the chip idle and every tile is a 16-round chain per lane — the pass is one tile's latency.  At or below
// kRadixSmallN elements (decided ON THE DEVICE from *d_n: the host only knows an upper bound) a pass runs 2048-element tiles,
// four rounds per lane: four times the workgroups, a quarter of the chain each.  Same ranks, same order (tile boundaries are not
// part of the result).  (Round 5 swept the threshold with an environment switch; settled, the switch is gone.)
constexpr int kRadixRoundsSmall = 4;
constexpr int kRadixTileSmall = kSweepThreads * kRadixRoundsSmall;  // 2048
constexpr uint32_t kRadixSmallN = 1u << 19;
constexpr uint32_t kRadixGrid = 512;                      // persistent workgroups = resident capacity (73 KB of LDS each: 2 per CU)
constexpr uint32_t kMaxPasses = 4;
constexpr int kLook = 4;                                  // predecessors examined per look-back round trip (2 and 8 measured)

constexpr unsigned long long kFlagAggregate = 1, kFlagPrefix = 2;

#ifdef GSX_SORT_PROFILE  // tools/bench_sort.hip: per-tile phase timestamps (100 MHz wall clock), 8 slots per tile
__device__ long long* g_sort_prof = nullptr;
#define GSX_PROF(ph)                                                                              \
    do {                                                                                          \
        if (g_sort_prof && threadIdx.x == 0) g_sort_prof[(size_t)tile * 8 + (ph)] = wall_clock64(); \
    } while (0)
#else
#define GSX_PROF(ph)
#endif

static uint32_t radix_small_n() {
    return kRadixSmallN;
}
// tiles a pass over at most n elements may use (status words, grid): the larger of the two tilings
static inline uint32_t radix_tiles(uint64_t n) {
    const uint64_t big = (n + kRadixTile - 1) / kRadixTile;
    const uint64_t small_ = (std::min<uint64_t>(n, radix_small_n()) + kRadixTileSmall - 1) / kRadixTileSmall;
    return (uint32_t)std::max(big, small_);
}

// workspace (u32 words): [0, 1024) global histograms of up to 4 passes | 1024 ticket | 1025 finished |
// from 1032: 2 words (one 64-bit status) per (tile, digit)
size_t radix_workspace_words(uint64_t n) { return 1032 + (size_t)radix_tiles(std::max<uint64_t>(n, 1)) * 256 * 2; }

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

// lanes whose `theirs` equals THIS lane's `mine` (8-bit values)
__device__ inline unsigned long long wave_match8_pair(uint32_t theirs, uint32_t mine) {
    unsigned long long m = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot((theirs >> b) & 1u);
        m &= ((mine >> b) & 1u) ? bal : ~bal;
    }
    return m;
}

__device__ inline unsigned long long lanemask_lt() { return (1ull << (threadIdx.x & 63u)) - 1ull; }

// ---- histograms of every pass in one read of the keys ----
// skip_key: elements whose key is 0xFFFFFFFF do not exist (a projection's culled records: the sort compacts on the way)
template <int KEY_STRIDE /* 1: key array, 2: interleaved {key,value} pairs */>
__global__ __launch_bounds__(kRadixThreads) void k_radix_global_hist(const uint32_t* __restrict__ keys, uint32_t n_cap,
                                                                      const uint32_t* __restrict__ d_n, int passes, int dbits,
                                                                      uint32_t* __restrict__ ghist, int skip_key) {
    __shared__ uint32_t hist[kMaxPasses][256];
    const uint32_t n = d_n ? min(*d_n, n_cap) : n_cap;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (int p = 0; p < passes; ++p) hist[p][tid] = 0;
    __syncthreads();
    // Every workgroup takes ONE contiguous share of the keys, at least 4096 of them: the launch is sized by an upper bound, and
    // with the elements dealt round-robin every one of 768 workgroups held a few of a 0.4 M-entry block sort's keys and flushed
    // ~250 bins each — 100 K global atomics on 256 addresses were most of that kernel's 14 us.  Inside its share a wave takes
    // 64-element rounds, four per trip with their loads issued together (one load in flight per wave made the kernel a chain
    // of memory round trips).
    constexpr int kHistUnroll = 4;
    const uint32_t share = max(4096u, (((n + gridDim.x - 1u) / gridDim.x) + 1023u) & ~1023u);
    const uint32_t lo = blockIdx.x * share, hi = min(n, lo + share);
    if (lo >= n) return;  // (uniform per workgroup, nothing to flush)
    for (uint32_t base = lo + (tid >> 6) * (64u * kHistUnroll); base < hi; base += kRadixWaves * (64u * kHistUnroll)) {
        uint32_t key[kHistUnroll];
#pragma unroll
        for (int u = 0; u < kHistUnroll; ++u) {
            const uint32_t e = base + u * 64u + lane;
            key[u] = e < hi ? keys[(size_t)e * KEY_STRIDE] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kHistUnroll; ++u) {
            const uint32_t e = base + u * 64u + lane;
            const bool valid = e < hi && !(skip_key && key[u] == 0xFFFFFFFFu);
            const unsigned long long vmask = __ballot(valid);
            for (int p = 0; p < passes; ++p) {
                const uint32_t digit = (key[u] >> (dbits * p)) & ((1u << dbits) - 1u);
                // only counts are needed here: plain LDS atomics, except when the whole wave shares one digit
                // (the exponent byte of depth keys), where 64 same-address atomics would serialise
                const uint32_t first = __builtin_amdgcn_readfirstlane(digit);
                if (__ballot(valid && digit != first) == 0) {
                    if (vmask && lane == (uint32_t)__ffsll((long long)vmask) - 1u) atomicAdd(&hist[p][first], (uint32_t)__popcll(vmask));
                } else if (valid) {
                    atomicAdd(&hist[p][digit], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p)
        if (hist[p][tid]) atomicAdd(&ghist[p * 256 + tid], hist[p][tid]);
}

// one workgroup per row: exclusive scan of table[row][0..nblocks) in place; total -> totals[row]
// (used by the multi-GPU pack, kernels_shard.hip)
__global__ __launch_bounds__(256) void k_radix_rowscan(uint32_t* __restrict__ table, uint32_t nblocks,
                                                        uint32_t* __restrict__ totals, const uint32_t* __restrict__ d_n, uint32_t tile,
                                                        const uint32_t* __restrict__ d_skip) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    uint32_t* row = table + (size_t)blockIdx.x * nblocks;  // (the row stride stays the full grid's)
    if (d_n) nblocks = min(nblocks, (*d_n + tile - 1u) / tile);
    if (d_skip && *d_skip == 0u) nblocks = 0;  // the producer wrote nothing (or zeros): the total is 0
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 256) {
        uint32_t i = base + tid;
        uint32_t v = i < nblocks ? row[i] : 0u;
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

// ---- one digit pass ----
// IN : 0 = keys array, value = element index | 1 = split key / value arrays | 2 = interleaved pairs
// OUT: 0 = interleaved pairs                 | 1 = split key / value arrays
typedef unsigned long long u64;

// SKIP (IN == 0 only): input elements whose key is 0xFFFFFFFF do not exist — the first pass of a depth sort reads the
// projection's key plane as it lies (culled records carry that key) and writes dense pairs: no compaction pass before the
// sort (the unspeculated frame paid 40 MB + 68 MB + 68 MB for one).  The number of elements that do exist is the sum of this
// pass's histogram; workgroup 0 writes it to *d_n_out, which is the element count of the passes that follow.
// MSD: the digit of a key is s_map[msd_fine(key)] — the partition pass of the bucket sort (gsx_internal.h): `ghist` is the 2048-bin fine
// histogram, cut here, by every workgroup alike, into 256 coarse buckets of equal population (consecutive fine bins: the digit is
// monotone in the key); ranges_out receives the buckets.
template <int IN, int OUT, bool LANE_ORDERED, bool SKIP, int ROUNDS, bool MSD = false>
__device__ __forceinline__ void radix_sweep_tiles(const uint32_t* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   const uint2* __restrict__ pairs_in,
                                                                   uint32_t* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out,
                                                          uint2* __restrict__ pairs_out, const uint32_t n, int shift, uint32_t dmask,
                                                          const uint32_t* __restrict__ ghist /* this pass */,
                                                          uint32_t* __restrict__ ticket /* [0] ticket, [1] finished */,
                                                          u64* __restrict__ status, uint32_t epoch,
                                                          uint32_t* __restrict__ ghist_clear /* last pass: all rows */,
                                                          uint32_t ghist_clear_words, uint32_t* __restrict__ d_n_out,
                                                          uint2* __restrict__ ranges_out,
                                                          const uint4* __restrict__ payload_in, uint4* __restrict__ payload_out,
                                                          const uint8_t* __restrict__ code_in, uint8_t* __restrict__ code_out,
                                                          uint2* __restrict__ s_pairs /* LDS: kRadixTile */, uint32_t (*__restrict__ cnt)[256] /* LDS: [kSweepWaves][256] */,
                                                          uint32_t* __restrict__ s_gbase /* LDS: 256 */, uint32_t* __restrict__ s_wtot /* LDS: kSweepWaves */,
                                                          uint32_t* __restrict__ s_misc /* LDS: [0] tile, [1] last, [2] tile_n */,
                                                          uint8_t* __restrict__ s_map = nullptr /* MSD, LDS: kMsdFine */, uint32_t* __restrict__ s_span = nullptr /* MSD, LDS: 512 */,
                                                          const MsdMap msd_map = MsdMap{0u, 0u, 0u}) {
    constexpr int kTile = kSweepThreads * ROUNDS;  // elements per tile
    auto digit_of = [&](uint32_t k) -> uint32_t {
        if constexpr (MSD) return s_map[msd_fine(k, msd_map)];
        else return (k >> shift) & dmask;
    };
    constexpr int kChunk = 64 * ROUNDS;            // contiguous elements per wave
    uint32_t& s_tile = s_misc[0];
    uint32_t& s_last = s_misc[1];
    uint32_t& s_tile_n = s_misc[2];               // elements of the tile that exist (SKIP)
    const uint32_t n_tiles = (n + kTile - 1) / kTile;
    // the launch is sized by an upper bound; with a device-side n only min(grid, n_tiles) workgroups have anything to
    // do — the others leave without touching the ticket (768 same-address atomics alone cost ~9 us)
    const uint32_t participants = min(gridDim.x, n_tiles);
    if (blockIdx.x >= participants) return;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const bool dig = tid < 256u;  // this thread also owns digit `tid` (the scans below run over the first four waves; the others add 0)

    // exclusive scan of this pass's global digit histogram: thread d -> first output slot of digit d
    uint32_t dbase;
    if constexpr (MSD) {
        // fine histogram -> exclusive prefix (thread t: bins 4t .. 4t + 3) -> coarse bucket of every fine bin = prefix / ceil(n / 256):
        // a bucket holds consecutive bins whose prefixes fall into one stretch of that length — at most that many elements plus its
        // last bin's.  [first, end) of every bucket: min / max over its bins (LDS atomics; the prefix is monotone).
        const uint4 h4 = reinterpret_cast<const uint4*>(ghist)[tid];
        const uint32_t sum4 = h4.x + h4.y + h4.z + h4.w;
        uint32_t x = sum4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) s_wtot[wave] = x;
        s_span[tid] = tid < 256u ? 0xFFFFFFFFu : 0u;  // [0, 256): first slot of a bucket (min) | [256, 512): its end (max)
        __syncthreads();
        uint32_t woff = 0, total = 0;
        for (uint32_t w = 0; w < (uint32_t)kSweepWaves; ++w) {
            if (w < wave) woff += s_wtot[w];
            total += s_wtot[w];
        }
        const uint32_t target = max(1u, (total + kMsdBuckets - 1u) / kMsdBuckets);
        uint32_t pre = woff + x - sum4;
        const uint32_t hh[4] = {h4.x, h4.y, h4.z, h4.w};
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t c = min(kMsdBuckets - 1u, pre / target);
            packed |= c << (8 * j);
            if (hh[j]) {
                atomicMin(&s_span[c], pre);
                atomicMax(&s_span[256u + c], pre + hh[j]);
            }
            pre += hh[j];
        }
        reinterpret_cast<uint32_t*>(s_map)[tid] = packed;
        __syncthreads();
        const uint32_t first = dig ? s_span[tid] : 0xFFFFFFFFu, end = dig ? s_span[256u + tid] : 0u;
        const uint32_t v = first != 0xFFFFFFFFu ? end - first : 0u;
        dbase = v ? first : 0u;
        if (ranges_out && blockIdx.x == 0 && dig) ranges_out[tid] = v ? make_uint2(dbase, dbase + v) : make_uint2(0u, 0u);
        __syncthreads();
    } else {
        const uint32_t v = dig ? ghist[tid] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) s_wtot[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += s_wtot[w];
        dbase = woff + x - v;
        if (SKIP && blockIdx.x == 0 && tid == 255u && d_n_out) *d_n_out = woff + x;  // the histogram's total: what exists
        // a one-pass sort (the block sort: the digit IS the key): [first slot, end) of every key is the scan just made —
        // what a k_tile_ranges launch over the sorted keys would find
        if (ranges_out && blockIdx.x == 0 && dig) ranges_out[tid] = v ? make_uint2(dbase, dbase + v) : make_uint2(0u, 0u);
        __syncthreads();
    }

    for (;;) {
        // one tile per ticket: batching consecutive tiles would chain every workgroup behind the LAST tile
        // of its predecessor and serialise the look-back
        if (tid == 0) s_tile = atomicAdd(&ticket[0], 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= n_tiles) break;
        GSX_PROF(0);
        for (uint32_t i = tid; i < (uint32_t)kSweepWaves * 256u; i += kSweepThreads) (&cnt[0][0])[i] = 0;
        __syncthreads();

        // load + stable ranks inside the wave's contiguous 1024-element chunk
        const uint32_t base = tile * kTile + wave * kChunk;
        uint32_t key[ROUNDS], val[ROUNDS], rank[ROUNDS];
        // all of the lane's loads first: inside the ranking loop (LDS counters, wave barriers) the compiler kept every load
        // next to its use, and a tile paid 16 serial memory round trips (10-12 us of its ~24)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = base + r * 64 + lane;
            const bool valid = e < n;
            if (IN == 2) {
                uint2 kv = valid ? pairs_in[e] : make_uint2(0xFFFFFFFFu, 0u);
                key[r] = kv.x;
                val[r] = kv.y;
            } else {
                key[r] = valid ? keys_in[e] : 0xFFFFFFFFu;
                val[r] = valid ? (IN == 0 ? e : vals_in[e]) : 0u;
                // (Records::code8 rides in the value's top byte — element indices below 2^24 — from the first pass, which reads it
                //  coalesced, to the last, which splits it off again: a gather of it by depth order cost a 64-byte sector an element)
                if (IN == 0 && code_in && valid) val[r] |= (uint32_t)code_in[e] << 24;
            }
        }
        if (LANE_ORDERED) {
            // Stable ranks straight from the LDS: one returning add per element on its (wave, digit) counter.  Lanes of one
            // instruction that hit the same counter are served in ascending lane order on this hardware — undocumented,
            // so radix_lane_ordered_adds() checks it on the device before this variant is ever chosen — and the LDS
            // executes a wave's instructions in order, so the old value IS the element's rank among the wave's
            // 1024-element chunk.  (The ballot-matching path below spends ~7 us per tile here, this one well under 1.)
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const bool valid = SKIP ? key[r] != 0xFFFFFFFFu : base + r * 64 + lane < n;
                rank[r] = valid ? atomicAdd(&cnt[wave][digit_of(key[r])], 1u) : 0u;
            }
        } else {
            // Stable ranks in three straight-line phases, so that nothing waits for an LDS round trip per round (a read ->
            // write -> read chain through the counters cost 16 x 2 LDS latencies per tile): (a) the digit-match masks of all
            // rounds, independent of each other; (b) one returning LDS add per digit group and round, issued back to back by
            // the group's first lane — the LDS executes a wave's operations in order, which is exactly the sequential
            // semantics the counters need; (c) the group's base handed to its other lanes by a lane permute.
            unsigned long long mm[ROUNDS];
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const bool valid = SKIP ? key[r] != 0xFFFFFFFFu : base + r * 64 + lane < n;
                mm[r] = wave_match8(digit_of(key[r]), valid);
            }
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const bool valid = SKIP ? key[r] != 0xFFFFFFFFu : base + r * 64 + lane < n;
                const uint32_t before = (uint32_t)__popcll(mm[r] & lanemask_lt());
                rank[r] = 0;
                if (valid && before == 0) rank[r] = atomicAdd(&cnt[wave][digit_of(key[r])], (uint32_t)__popcll(mm[r]));
            }
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t first = ((uint32_t)__ffsll((long long)mm[r]) - 1u) & 63u;
                rank[r] = (uint32_t)__shfl((int)rank[r], (int)first, 64) + (uint32_t)__popcll(mm[r] & lanemask_lt());
            }
        }
        __syncthreads();
        GSX_PROF(1);

        // thread d: tile count of digit d, per-wave exclusive offsets, publish, look back
        uint32_t tile_cnt = 0;
        if (dig) {
#pragma unroll
            for (int w = 0; w < kSweepWaves; ++w) {
                const uint32_t c = cnt[w][tid];
                cnt[w][tid] = tile_cnt;  // exclusive offset of wave w inside digit d's run
                tile_cnt += c;
            }
        }
        u64* my_status = status + ((size_t)tile * 256 + tid);
        const u64 tag = (u64)epoch << 34;
        const bool live = dig && tid <= dmask;  // digits this pass can produce: the others have nothing to publish or look up
        if (live)
            __hip_atomic_store(my_status, tag | ((tile == 0 ? kFlagPrefix : kFlagAggregate) << 32) | (u64)tile_cnt,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // local exclusive scan over digits: first local slot of digit d in the reordered tile
        uint32_t lstart;
        {
            uint32_t x = tile_cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t y = __shfl_up(x, o, 64);
                if (lane >= (uint32_t)o) x += y;
            }
            if (lane == 63) s_wtot[wave] = x;
            __syncthreads();
            uint32_t woff = 0;
            for (uint32_t w = 0; w < wave; ++w) woff += s_wtot[w];
            lstart = woff + x - tile_cnt;
            if (SKIP && tid == 255u) s_tile_n = woff + x;  // (threads past the 256 digits add nothing: this is the tile's total)
        }
        GSX_PROF(2);
        // The tile is reordered by digit in LDS BEFORE the look-back: the reorder needs local offsets only, and the 2 us it
        // takes are 2 us the predecessors have had to publish (measured: 329 -> 320 us per 8.46 M-pair sort).  What the
        // look-back costs on this part is not the walk but visibility: an agent-scope store of one XCD reaches a reader on
        // another after ~3 us, and a tile needs one or two such hops — a two-level (grouped) look-back that reads 10x fewer
        // status words and polling back-offs of 30 ns ... 3 us all left its 7.5 us per tile unchanged (tools/bench_sort.hip).
        if (dig) {
#pragma unroll
            for (int w = 0; w < kSweepWaves; ++w) cnt[w][tid] += lstart;  // local slot = cnt[w][d] + rank
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = base + r * 64 + lane;
            if (SKIP ? key[r] != 0xFFFFFFFFu : e < n) {
                const uint32_t digit = digit_of(key[r]);
                s_pairs[cnt[wave][digit] + rank[r]] = make_uint2(key[r], val[r]);
            }
        }
        GSX_PROF(3);
        uint32_t excl = 0;
        if (tile > 0 && live) {
            // Look-back, kLook predecessors per round trip: the status loads of a batch are independent and issued
            // together (one load at a time made every predecessor a full L2 round trip: with hundreds of tiles in flight
            // the look-back, not the data movement, set the pass time).  Words are consumed nearest first; an
            // unpublished one ends the batch and is polled again.
            int32_t k = (int32_t)tile - 1;
            bool found = false;
            while (!found) {
                u64 w[kLook];
#pragma unroll
                for (int i = 0; i < kLook; ++i) {
                    const int32_t kk = max(k - i, 0);
                    w[i] = __hip_atomic_load(status + ((size_t)kk * 256 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int32_t used = 0;
#pragma unroll
                for (int i = 0; i < kLook; ++i) {
                    if (!found && used == i && k - i >= 0) {
                        const uint32_t flag = (uint32_t)(w[i] >> 32) & 3u;
                        if ((uint32_t)(w[i] >> 34) == epoch && flag != 0) {  // published in this launch
                            excl += (uint32_t)w[i];
                            used = i + 1;
                            found = flag == kFlagPrefix || k - i == 0;
                        }
                    }
                }
                k -= used;
                if (!found && used == 0) __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(my_status, tag | (kFlagPrefix << 32) | (u64)(excl + tile_cnt), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (dig) s_gbase[tid] = dbase + excl - lstart;
        __syncthreads();
        GSX_PROF(4);
        // write out: consecutive lanes -> consecutive addresses inside every digit run
        const uint32_t tile_n = SKIP ? s_tile_n : min((uint32_t)kTile, n - tile * kTile);
#pragma unroll 4
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t slot = r * kSweepThreads + tid;
            if (slot < tile_n) {
                const uint2 kv = s_pairs[slot];
                const uint32_t o = s_gbase[digit_of(kv.x)] + slot;
                if (OUT == 0) {
                    pairs_out[o] = kv;
                } else {
                    keys_out[o] = kv.x;
                    vals_out[o] = code_out ? (kv.y & 0xFFFFFFu) : kv.y;
                    // (the block sort: the 16-byte record its value names travels with it, so that whoever walks the sorted list
                    //  reads records side by side instead of one dependent gather per entry — k_composite_blocks)
                    if (payload_out) payload_out[o] = payload_in[kv.y];
                    if (code_out) code_out[o] = (uint8_t)(kv.y >> 24);   // the value's top byte is the element's code (packed by the first pass)
                }
            }
        }
        __syncthreads();  // s_tile, cnt, s_pairs are reused by the next tile
        GSX_PROF(5);
    }
    // the last workgroup to leave re-arms the ticket for the next launch and, after the final pass, clears the
    // digit histograms so that the next sort on this workspace needs no memset
    // (its own LDS word: lanes that have not yet read the final ticket out of s_tile must not see this flag instead — with
    // s_tile reused here, a late lane took 0 or 1 for its ticket, went back into the loop alone and hung the pass)
    if (tid == 0) {
        const uint32_t fin = atomicAdd(&ticket[1], 1u);
        s_last = (fin == participants - 1) ? 1u : 0u;
        if (s_last) {
            ticket[1] = 0;
            __hip_atomic_store(&ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_last)
        for (uint32_t k = tid; k < ghist_clear_words; k += kSweepThreads) ghist_clear[k] = 0;
}

// The pass as a kernel: tile size chosen from the element count that exists on the device.
template <int IN, int OUT, bool LANE_ORDERED, bool SKIP = false>
__global__ __launch_bounds__(kSweepThreads) void k_radix_onesweep(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                   const uint2* __restrict__ pairs_in, uint32_t* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out, uint2* __restrict__ pairs_out, uint32_t n_cap,
                                                                   const uint32_t* __restrict__ d_n, int shift, uint32_t dmask,
                                                                   const uint32_t* __restrict__ ghist, uint32_t* __restrict__ ticket,
                                                                   u64* __restrict__ status, uint32_t epoch, uint32_t* __restrict__ ghist_clear,
                                                                   uint32_t ghist_clear_words, uint32_t* __restrict__ d_n_out,
                                                                   uint2* __restrict__ ranges_out, const uint4* __restrict__ payload_in,
                                                                   uint4* __restrict__ payload_out, uint32_t small_n, const uint8_t* __restrict__ code_in,
                                                                   uint8_t* __restrict__ code_out) {
    __shared__ uint2 s_pairs[kRadixTile];       // tile reordered by digit
    __shared__ uint32_t cnt[kSweepWaves][256];  // per-wave digit counts, then per-wave local offsets
    __shared__ uint32_t s_gbase[256];           // global slot of the tile's local slot 0, per digit
    __shared__ uint32_t s_wtot[kSweepWaves];
    __shared__ uint32_t s_misc[4];
    const uint32_t n = d_n ? min(*d_n, n_cap) : n_cap;
    if (!SKIP && n <= small_n)
        radix_sweep_tiles<IN, OUT, LANE_ORDERED, SKIP, kRadixRoundsSmall>(keys_in, vals_in, pairs_in, keys_out, vals_out, pairs_out, n, shift, dmask, ghist, ticket,
                                                                           status, epoch, ghist_clear, ghist_clear_words, d_n_out, ranges_out, payload_in,
                                                                           payload_out, code_in, code_out, s_pairs, cnt, s_gbase, s_wtot, s_misc);
    else
        radix_sweep_tiles<IN, OUT, LANE_ORDERED, SKIP, kRadixRounds>(keys_in, vals_in, pairs_in, keys_out, vals_out, pairs_out, n, shift, dmask, ghist, ticket,
                                                                      status, epoch, ghist_clear, ghist_clear_words, d_n_out, ranges_out, payload_in, payload_out,
                                                                      code_in, code_out, s_pairs, cnt, s_gbase, s_wtot, s_misc);
}

hipError_t launch_rowscan(hipStream_t s, uint32_t* table, uint32_t nrows, uint32_t nblocks, uint32_t* totals, const uint32_t* d_n, uint32_t tile,
                          const uint32_t* d_skip) {
    if (nrows) GSX_LAUNCH(k_radix_rowscan, dim3(nrows), dim3(256), 0, s, table, nblocks, totals, d_n, tile, d_skip);
    return hipGetLastError();
}

static std::atomic<uint32_t> g_epoch{1};  // distinguishes the status words of successive launches (any stream, any viewer, any thread)

// ---- is a returning LDS add served in ascending lane order?  (see k_radix_onesweep<.., LANE_ORDERED>) ----
// The probe has the shape of the kernel that relies on the answer: 512-thread workgroups (eight waves sharing the LDS), every
// lane issuing kRadixRounds back-to-back returning adds on its wave's 256 counters, one workgroup per resident slot of the
// persistent sort grid (two per CU) so the LDS is contended the way it is in production, and address patterns from "no
// two lanes collide" to "all 64 lanes on one counter".  Expected value of each add: the counter before the instruction
// (the wave's earlier rounds, tracked by ballot matching — the documented path) plus the number of LOWER lanes that hit
// the same counter in the same instruction.
__global__ __launch_bounds__(kSweepThreads) void k_lane_order_probe(uint32_t seed, uint32_t iterations, uint32_t* __restrict__ violations) {
    __shared__ uint32_t c[kSweepWaves][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t bad = 0;
    for (uint32_t it = 0; it < iterations; ++it) {
        for (uint32_t i = tid; i < (uint32_t)kSweepWaves * 256u; i += kSweepThreads) (&c[0][0])[i] = 0;
        __syncthreads();
        const uint32_t span = 1u << ((blockIdx.x + it + wave) % 9u);  // 1, 2, 4 ... 256 distinct counters
        uint32_t a[kRadixRounds], got[kRadixRounds];
#pragma unroll
        for (int r = 0; r < kRadixRounds; ++r) {
            uint32_t x = (seed + blockIdx.x * 9781u + it * 6271u + (uint32_t)r * 7919u + wave * 104729u) * 2654435761u + lane * 40503u;
            x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            a[r] = x & (span - 1u);
        }
#pragma unroll
        for (int r = 0; r < kRadixRounds; ++r) got[r] = atomicAdd(&c[wave][a[r]], 1u);  // chained, like the ranking loop
        // reference ranks: per round, lower lanes with the same counter + that counter's total over the earlier rounds
#pragma unroll
        for (int r = 0; r < kRadixRounds; ++r) {
            const unsigned long long m = wave_match8(a[r], true);
            uint32_t earlier = 0;
            for (int q = 0; q < r; ++q) earlier += (uint32_t)__popcll(wave_match8_pair(a[q], a[r]));
            bad += got[r] != earlier + (uint32_t)__popcll(m & lanemask_lt());
        }
        __syncthreads();
    }
    if (bad) atomicAdd(violations, bad);
}

// One answer per device (the property belongs to the silicon a viewer runs on): -1 not probed yet (the ballot-matching ranks
// are used), 0 no, 1 yes.  Probed on the CURRENT device; gsx_viewer_create calls it after hipSetDevice.
constexpr int kMaxDevices = 64;
static std::atomic<int> g_lane_ordered[kMaxDevices];
static std::atomic<bool> g_lane_init{false};
static std::atomic<int> g_rank_override{-1};  // gsx_debug_set_radix_rank_mode: -1 none, 0 force matching, 1 force lane-ordered

static int lane_ordered_slot(int* dev_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
    *dev_out = dev;
    bool expected = false;
    if (g_lane_init.compare_exchange_strong(expected, true))
        for (auto& x : g_lane_ordered) x.store(-1);
    return dev < 0 ? 0 : g_lane_ordered[dev].load();
}

bool radix_lane_ordered_adds() {
    int dev;
    const int known = lane_ordered_slot(&dev);
    if (dev < 0) return false;
    if (known >= 0) return known == 1;
    if (getenv("GSX_RADIX_MATCH_RANKS")) {  // force the documented-behaviour path
        g_lane_ordered[dev].store(0);
        return false;
    }
    uint32_t* d = nullptr;
    uint32_t h = 1;
    bool ok = hipMalloc(&d, 4) == hipSuccess && gsx::op::Memset(d, 0, 4) == hipSuccess;
    if (ok) {
        GSX_LAUNCH(k_lane_order_probe, dim3(kRadixGrid), dim3(kSweepThreads), 0, 0, 12345u, 8u, d);
        ok = hipGetLastError() == hipSuccess && gsx::op::Memcpy(&h, d, 4, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (d) (void)gsx::op::Free(d);
    g_lane_ordered[dev].store((ok && h == 0) ? 1 : 0);
    return ok && h == 0;
}

void radix_set_rank_override(int mode) { g_rank_override.store(mode < 0 ? -1 : (mode ? 1 : 0)); }

static bool use_lane_ordered() {
    const int o = g_rank_override.load(std::memory_order_relaxed);
    if (o >= 0) return o == 1;
    int dev;
    return lane_ordered_slot(&dev) == 1;
}

hipError_t launch_radix_sort(hipStream_t s, const RadixBuffers& buf, uint32_t n, uint32_t* d_n, int bits, bool iota_values,
                             bool skip_culled, uint2* ranges_out, const uint4* payload_in, uint4* payload_out, bool hist_done,
                             const uint8_t* code_in, uint8_t* code_out) {
    if (n == 0) return hipSuccess;
    const int passes = (bits + 7) / 8;
    if (ranges_out && passes != 1) return hipErrorInvalidValue;  // key ranges fall out of a ONE-digit sort only
    // equal digit widths (13 tile-key bits sort as 7 + 6, not 8 + 5): fewer digits mean shorter status rows to publish
    // and look back over, and longer runs per digit in the scattered writes
    const int dbits = (bits + passes - 1) / passes;
    const uint32_t dmask = (1u << dbits) - 1u;
    uint32_t* ghist = buf.workspace;
    uint32_t* ticket = buf.workspace + 1024;
    u64* status = reinterpret_cast<u64*>(buf.workspace + 1032);
    const uint32_t tiles = radix_tiles(n);
    const uint32_t grid_limit = kRadixGrid;   // (768 / 512 / 256 workgroups swept in rounds 2-3: the resident capacity wins)
    const uint32_t grid = std::min<uint32_t>(grid_limit, tiles);
    // skip_culled: the keys are a projection's key plane (iota values); records whose key is 0xFFFFFFFF do not exist.  The
    // histogram and the first pass run over all n of them, *d_n receives how many exist, the later passes run over those.
    const bool skip = skip_culled && buf.keys_src && !buf.pairs_src && iota_values && passes > 1 && d_n;
    // ghist is zero here: the workspace is cleared at allocation and every sort's last pass clears it again
    // histogram workgroups: ~one per 1024..4096 elements up to the persistent grid (sorts of a few hundred thousand elements
    // are latency-bound: 300 k keys took 89 us with 4096 x 4 elements per workgroup, 67 us with 4096)
    const uint32_t hper = 4096u;
    const uint32_t hgrid = std::max<uint32_t>(1u, std::min<uint32_t>(768u, (uint32_t)(((uint64_t)n + hper - 1) / hper)));
    if (hist_done) {
        // (the kernel that wrote the pairs counted their digits into ghist: k_block_bin)
    } else if (buf.pairs_src)
        GSX_LAUNCH(k_radix_global_hist<2>, dim3(hgrid), dim3(kRadixThreads), 0, s,
                           reinterpret_cast<const uint32_t*>(buf.pairs_src), n, d_n, passes, dbits, ghist, 0);
    else
        GSX_LAUNCH(k_radix_global_hist<1>, dim3(hgrid), dim3(kRadixThreads), 0, s, buf.keys_src, n, skip ? nullptr : d_n, passes, dbits,
                           ghist, skip ? 1 : 0);
    const uint2* pin = buf.pairs_src;
    uint2* pout = buf.pairs_a;
    const bool lane_ordered = use_lane_ordered();
    for (int p = 0; p < passes; ++p) {
        const int shift = dbits * p;
        const bool first = p == 0, last = p == passes - 1;
        const uint32_t epoch = (g_epoch.fetch_add(1, std::memory_order_relaxed) & 0x1FFFFFFFu) | (1u << 29);  // 30 bits, never 0
#define GSX_SWEEP_ARGS(DN, DNOUT)                                                                                        \
    dim3(grid), dim3(kSweepThreads), 0, s, buf.keys_src, buf.vals_src, pin, buf.keys_out, buf.vals_out, pout, n, DN, shift, dmask, \
        ghist + 256 * p, ticket, status, epoch, ghist, last ? 256u * (uint32_t)passes : 0u, DNOUT, ranges_out,                  \
        last ? payload_in : nullptr, last ? payload_out : nullptr, radix_small_n(), (first && code_out) ? code_in : nullptr, last ? code_out : nullptr
#define GSX_SWEEP(IN, OUT)                                                                                               \
    do {                                                                                                                 \
        if (lane_ordered)                                                                                                \
            GSX_LAUNCH((k_radix_onesweep<IN, OUT, true>), GSX_SWEEP_ARGS(d_n, nullptr));                          \
        else                                                                                                             \
            GSX_LAUNCH((k_radix_onesweep<IN, OUT, false>), GSX_SWEEP_ARGS(d_n, nullptr));                         \
    } while (0)
        if (first && skip) {  // (more than one pass: the first one writes interleaved pairs)
            if (lane_ordered)
                GSX_LAUNCH((k_radix_onesweep<0, 0, true, true>), GSX_SWEEP_ARGS(nullptr, d_n));
            else
                GSX_LAUNCH((k_radix_onesweep<0, 0, false, true>), GSX_SWEEP_ARGS(nullptr, d_n));
        } else if (first && buf.pairs_src) {
            if (last) GSX_SWEEP(2, 1); else GSX_SWEEP(2, 0);
        } else if (first && last) {
            if (iota_values) GSX_SWEEP(0, 1); else GSX_SWEEP(1, 1);
        } else if (first) {
            if (iota_values) GSX_SWEEP(0, 0); else GSX_SWEEP(1, 0);
        } else if (last) {
            GSX_SWEEP(2, 1);
        } else {
            GSX_SWEEP(2, 0);
        }
#undef GSX_SWEEP
#undef GSX_SWEEP_ARGS
        pin = pout;
        pout = (pout == buf.pairs_a) ? buf.pairs_b : buf.pairs_a;
    }
    return hipGetLastError();
}

// =====================================================================================================================
// Bucket sort (gsx_internal.h "bucket sort"): fine histogram -> MSD partition (the onesweep kernel, digit = balanced bucket
// table) -> every bucket sorted in LDS.  Three launches at most (two when the producer of the pairs counted the histogram)
// for what the LSD sort does in five; the depth sort of a speculated frame, of its repair round, of a multi-GPU rank's band.
// =====================================================================================================================
uint32_t next_sort_epoch() { return (g_epoch.fetch_add(1, std::memory_order_relaxed) & 0x1FFFFFFFu) | (1u << 29); }

// the fine histogram + key range of sort `seq` from keys that already lie in memory (KEY_STRIDE 1: key array, 2: pairs)
template <int KEY_STRIDE>
__global__ __launch_bounds__(kRadixThreads) void k_msd_hist(const uint32_t* __restrict__ keys, uint32_t n_cap, const uint32_t* __restrict__ d_n,
                                                             uint32_t* __restrict__ fine, const uint32_t* __restrict__ hint, uint32_t* __restrict__ acc) {
    __shared__ uint32_t hist[kMsdFine];
    __shared__ uint32_t s_mn[kRadixWaves], s_mx[kRadixWaves];
    const uint32_t n = d_n ? min(*d_n, n_cap) : n_cap;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t share = max(4096u, (((n + gridDim.x - 1u) / gridDim.x) + 1023u) & ~1023u);
    const uint32_t lo_e = blockIdx.x * share, hi_e = min(n, lo_e + share);
    if (lo_e >= n) return;
    for (uint32_t i = tid; i < kMsdFine; i += kRadixThreads) hist[i] = 0;
    const MsdMap map = msd_mapping(hint);
    __syncthreads();
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    constexpr int kU = 4;
    for (uint32_t base = lo_e + wave * (64u * kU); base < hi_e; base += kRadixWaves * (64u * kU)) {
        uint32_t key[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t e = base + u * 64u + lane;
            key[u] = e < hi_e ? keys[(size_t)e * KEY_STRIDE] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t e = base + u * 64u + lane;
            if (e < hi_e) {
                atomicAdd(&hist[msd_fine(key[u], map)], 1u);
                mn = min(mn, key[u]);
                mx = max(mx, key[u]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_down((int)mn, o, 64));
        mx = max(mx, (uint32_t)__shfl_down((int)mx, o, 64));
    }
    if (lane == 0) {
        s_mn[wave] = mn;
        s_mx[wave] = mx;
    }
    __syncthreads();
    for (uint32_t i = tid; i < kMsdFine; i += kRadixThreads)
        if (hist[i]) atomicAdd(&fine[i], hist[i]);
    if (tid == 0) {
        for (int w = 1; w < kRadixWaves; ++w) {
            mn = min(mn, s_mn[w]);
            mx = max(mx, s_mx[w]);
        }
        if (mn <= mx) {
            atomicMin(&acc[0], mn);
            atomicMax(&acc[1], mx);
        }
    }
}

// the partition pass: k_radix_onesweep with the bucket table as its digit
template <int IN, bool LANE_ORDERED>
__global__ __launch_bounds__(kSweepThreads) void k_msd_sweep(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                              const uint2* __restrict__ pairs_in, uint2* __restrict__ pairs_out, uint32_t n_cap,
                                                              const uint32_t* __restrict__ d_n, uint32_t* __restrict__ fine, const uint32_t* __restrict__ hint,
                                                              uint32_t* __restrict__ ticket, u64* __restrict__ status, uint32_t epoch,
                                                              uint2* __restrict__ ranges_out, uint32_t small_n) {
    __shared__ uint2 s_pairs[kRadixTile];
    __shared__ uint32_t cnt[kSweepWaves][256];
    __shared__ uint32_t s_gbase[256];
    __shared__ uint32_t s_wtot[kSweepWaves];
    __shared__ uint32_t s_misc[4];
    __shared__ uint32_t s_span[512];
    __shared__ __attribute__((aligned(16))) uint8_t s_map[kMsdFine];
    const uint32_t n = d_n ? min(*d_n, n_cap) : n_cap;
    if (n == 0) {  // no tile, no workgroup takes part: the buckets of the sort before must not stay behind
        if (blockIdx.x == 0 && threadIdx.x < kMsdBuckets) ranges_out[threadIdx.x] = make_uint2(0u, 0u);
        return;
    }
    const MsdMap map = msd_mapping(hint);
    if (n <= small_n)
        radix_sweep_tiles<IN, 0, LANE_ORDERED, false, kRadixRoundsSmall, true>(keys_in, vals_in, pairs_in, nullptr, nullptr, pairs_out, n, 0, 255u, fine, ticket, status,
                                                                                epoch, fine, kMsdFine, nullptr, ranges_out, nullptr, nullptr, nullptr, nullptr, s_pairs, cnt, s_gbase,
                                                                                s_wtot, s_misc, s_map, s_span, map);
    else
        radix_sweep_tiles<IN, 0, LANE_ORDERED, false, kRadixRounds, true>(keys_in, vals_in, pairs_in, nullptr, nullptr, pairs_out, n, 0, 255u, fine, ticket, status, epoch,
                                                                           fine, kMsdFine, nullptr, ranges_out, nullptr, nullptr, nullptr, nullptr, s_pairs, cnt, s_gbase, s_wtot, s_misc,
                                                                           s_map, s_span, map);
}

// ---- every bucket sorted on the key bits that vary inside it: stable LSD passes of up to 8 bits, per-wave counters ----
constexpr int kBucketRounds = 16;
constexpr uint32_t kBucketCap = kSweepThreads * kBucketRounds;  // 8192 pairs in LDS

// stable ranks of the wave's elements among the wave's chunk (one returning LDS add per element, or ballot matching)
template <bool LANE_ORDERED, int ROUNDS>
__device__ __forceinline__ void bucket_ranks(const uint32_t (&dg)[ROUNDS], const bool (&ok)[ROUNDS], uint32_t (&rank)[ROUNDS], uint32_t* __restrict__ wcnt /* LDS: this wave's 256 counters */,
                                             int rounds) {
    if (LANE_ORDERED) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) rank[r] = (r < rounds && ok[r]) ? atomicAdd(&wcnt[dg[r]], 1u) : 0u;
    } else {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            rank[r] = 0;
            if (r < rounds) {  // (uniform)
                const unsigned long long mm = wave_match8(dg[r], ok[r]);
                const uint32_t before = (uint32_t)__popcll(mm & lanemask_lt());
                uint32_t base = 0;
                if (ok[r] && before == 0) base = atomicAdd(&wcnt[dg[r]], (uint32_t)__popcll(mm));
                const uint32_t first = ((uint32_t)__ffsll((long long)mm) - 1u) & 63u;
                rank[r] = (uint32_t)__shfl((int)base, (int)first, 64) + before;
            }
        }
    }
}

// after the ranks: thread d turns cnt[w][d] into wave w's first slot of digit d (+ base[d]); returns nothing — callers sync
__device__ __forceinline__ void bucket_offsets(uint32_t (*__restrict__ cnt)[256], uint32_t* __restrict__ s_wtot, const uint32_t* __restrict__ s_base /* LDS, nullable: per-digit start */,
                                               uint32_t* __restrict__ s_base_next /* LDS, nullable: += the tile's count */) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool dig = tid < 256u;
    uint32_t tile_cnt = 0;
    if (dig) {
#pragma unroll
        for (int w = 0; w < kSweepWaves; ++w) {
            const uint32_t c = cnt[w][tid];
            cnt[w][tid] = tile_cnt;
            tile_cnt += c;
        }
    }
    uint32_t start;
    if (s_base) {  // several tiles: the digit's run begins where the tiles before left it
        start = dig ? s_base[tid] : 0u;
        if (dig && s_base_next) s_base_next[tid] = start + tile_cnt;
    } else {       // one tile: exclusive scan over the digits
        uint32_t x = tile_cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) s_wtot[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += s_wtot[w];
        start = woff + x - tile_cnt;
    }
    if (dig) {
#pragma unroll
        for (int w = 0; w < kSweepWaves; ++w) cnt[w][tid] += start;
    }
}

template <bool LANE_ORDERED>
__global__ __launch_bounds__(kSweepThreads) void k_bucket_sort(uint2* src /* the partitioned pairs (large buckets ping-pong between src and tmp) */, uint2* tmp,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, const uint2* __restrict__ ranges,
                                                                uint32_t cap, uint32_t* __restrict__ hint, const uint32_t* __restrict__ acc) {
    uint2* const src_rw = src;
    // the last kernel of a sort: the union of the key ranges seen so far becomes the range the NEXT sort's kernels map keys by (they all
    // run behind this kernel: the snapshot is stable while they read it)
    if (blockIdx.x == 0 && threadIdx.x == 0 && acc[0] <= acc[1]) {
        hint[0] = acc[0];
        hint[1] = acc[1];
    }
    __shared__ uint2 s_pairs[kBucketCap];
    __shared__ uint32_t cnt[kSweepWaves][256];
    __shared__ uint32_t s_wtot[kSweepWaves];
    __shared__ uint32_t s_mn[kSweepWaves], s_mx[kSweepWaves];
    __shared__ uint32_t s_base[256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t b = blockIdx.x; b < kMsdBuckets; b += gridDim.x) {
        const uint2 range = ranges[b];
        const uint32_t start = range.x, m = range.y - range.x;
        if (m == 0) continue;  // (uniform)
        if (m <= cap) {
            // ---- the bucket in registers / LDS ----
            const int rounds = (int)((m + kSweepThreads - 1u) / kSweepThreads);   // rounds per lane; wave w owns [w * 64 * rounds, ...)
            const uint32_t chunk = 64u * (uint32_t)rounds;
            uint32_t key[kBucketRounds], val[kBucketRounds];
            bool ok[kBucketRounds];
            uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
            for (int r = 0; r < kBucketRounds; ++r) {
                const uint32_t e = wave * chunk + (uint32_t)r * 64u + lane;
                ok[r] = r < rounds && e < m;
                uint2 kv = make_uint2(0u, 0u);
                if (ok[r]) kv = src[start + e];
                key[r] = kv.x;
                val[r] = kv.y;
                if (ok[r]) {
                    mn = min(mn, kv.x);
                    mx = max(mx, kv.x);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
                mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
            }
            if (lane == 0) {
                s_mn[wave] = mn;
                s_mx[wave] = mx;
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kSweepWaves; ++w) {
                mn = min(mn, s_mn[w]);
                mx = max(mx, s_mx[w]);
            }
            const int bits = mx > mn ? 32 - __clz((int)(mx - mn)) : 0;   // key bits that vary inside the bucket
            const int passes = (bits + 7) / 8;
            const int dbits = passes ? (bits + passes - 1) / passes : 0;
            const uint32_t dmask = (1u << dbits) - 1u;
            for (int p = 0; p < passes; ++p) {
                for (uint32_t i = tid; i < (uint32_t)kSweepWaves * 256u; i += kSweepThreads) (&cnt[0][0])[i] = 0;
                __syncthreads();
                uint32_t dg[kBucketRounds], rank[kBucketRounds];
#pragma unroll
                for (int r = 0; r < kBucketRounds; ++r) dg[r] = ((key[r] - mn) >> (dbits * p)) & dmask;
                bucket_ranks<LANE_ORDERED, kBucketRounds>(dg, ok, rank, cnt[wave], rounds);
                __syncthreads();
                bucket_offsets(cnt, s_wtot, nullptr, nullptr);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kBucketRounds; ++r)
                    if (ok[r]) s_pairs[cnt[wave][dg[r]] + rank[r]] = make_uint2(key[r], val[r]);
                __syncthreads();
                if (p + 1 < passes) {
#pragma unroll
                    for (int r = 0; r < kBucketRounds; ++r) {
                        if (ok[r]) {
                            const uint2 kv = s_pairs[wave * chunk + (uint32_t)r * 64u + lane];
                            key[r] = kv.x;
                            val[r] = kv.y;
                        }
                    }
                }
            }
            if (passes == 0) {  // every key of the bucket is the same: the partition pass left them in index order
#pragma unroll
                for (int r = 0; r < kBucketRounds; ++r) {
                    if (ok[r]) {
                        const uint32_t e = wave * chunk + (uint32_t)r * 64u + lane;
                        keys_out[start + e] = key[r];
                        vals_out[start + e] = val[r];
                    }
                }
            } else {
                for (uint32_t slot = tid; slot < m; slot += kSweepThreads) {
                    const uint2 kv = s_pairs[slot];
                    keys_out[start + slot] = kv.x;
                    vals_out[start + slot] = kv.y;
                }
            }
            __syncthreads();  // s_pairs, cnt, s_mn / s_mx are reused by the next bucket
        } else {
            // ---- a bucket that does not fit: the same stable passes through global memory, tile by tile (one workgroup: slow, correct) ----
            uint32_t mn = 0xFFFFFFFFu, mx = 0u;
            for (uint32_t e = tid; e < m; e += kSweepThreads) {
                const uint32_t k = src[start + e].x;
                mn = min(mn, k);
                mx = max(mx, k);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
                mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
            }
            if (lane == 0) {
                s_mn[wave] = mn;
                s_mx[wave] = mx;
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kSweepWaves; ++w) {
                mn = min(mn, s_mn[w]);
                mx = max(mx, s_mx[w]);
            }
            const int bits = mx > mn ? 32 - __clz((int)(mx - mn)) : 0;
            const int passes = (bits + 7) / 8;
            const int dbits = passes ? (bits + passes - 1) / passes : 0;
            const uint32_t dmask = (1u << dbits) - 1u;
            if (passes == 0) {
                for (uint32_t e = tid; e < m; e += kSweepThreads) {
                    const uint2 kv = src[start + e];
                    keys_out[start + e] = kv.x;
                    vals_out[start + e] = kv.y;
                }
            }
            const uint2* from = src + start;
            uint2* to = tmp + start;
            for (int p = 0; p < passes; ++p) {
                const bool last = p + 1 == passes;
                // the pass's histogram over the whole bucket -> first slot of every digit
                if (tid < 256u) s_base[tid] = 0;
                __syncthreads();
                for (uint32_t e = tid; e < m; e += kSweepThreads) atomicAdd(&s_base[((from[e].x - mn) >> (dbits * p)) & dmask], 1u);
                __syncthreads();
                {
                    const uint32_t c = tid < 256u ? s_base[tid] : 0u;
                    uint32_t x = c;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        uint32_t y = __shfl_up(x, o, 64);
                        if (lane >= (uint32_t)o) x += y;
                    }
                    if (lane == 63) s_wtot[wave] = x;
                    __syncthreads();
                    uint32_t woff = 0;
                    for (uint32_t w = 0; w < wave; ++w) woff += s_wtot[w];
                    if (tid < 256u) s_base[tid] = woff + x - c;
                    __syncthreads();
                }
                for (uint32_t t0 = 0; t0 < m; t0 += kBucketCap) {
                    const uint32_t tm = min(kBucketCap, m - t0);
                    const int rounds = (int)((tm + kSweepThreads - 1u) / kSweepThreads);
                    const uint32_t chunk = 64u * (uint32_t)rounds;
                    for (uint32_t i = tid; i < (uint32_t)kSweepWaves * 256u; i += kSweepThreads) (&cnt[0][0])[i] = 0;
                    __syncthreads();
                    uint32_t key[kBucketRounds], val[kBucketRounds], dg[kBucketRounds], rank[kBucketRounds];
                    bool ok[kBucketRounds];
#pragma unroll
                    for (int r = 0; r < kBucketRounds; ++r) {
                        const uint32_t e = wave * chunk + (uint32_t)r * 64u + lane;
                        ok[r] = r < rounds && e < tm;
                        uint2 kv = make_uint2(0u, 0u);
                        if (ok[r]) kv = from[t0 + e];
                        key[r] = kv.x;
                        val[r] = kv.y;
                        dg[r] = ((kv.x - mn) >> (dbits * p)) & dmask;
                    }
                    bucket_ranks<LANE_ORDERED, kBucketRounds>(dg, ok, rank, cnt[wave], rounds);
                    __syncthreads();
                    bucket_offsets(cnt, s_wtot, s_base, s_base);   // thread d reads s_base[d] and writes it back advanced: its own word only
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < kBucketRounds; ++r) {
                        if (ok[r]) {
                            const uint32_t o = cnt[wave][dg[r]] + rank[r];
                            if (last) {
                                keys_out[start + o] = key[r];
                                vals_out[start + o] = val[r];
                            } else {
                                to[o] = make_uint2(key[r], val[r]);
                            }
                        }
                    }
                    __syncthreads();
                }
                // what this pass wrote is what the next pass reads: visible to the whole workgroup behind the barrier
                __threadfence_block();
                __syncthreads();
                const uint2* nf = to;
                to = (to == tmp + start) ? src_rw + start : tmp + start;
                from = nf;
            }
            __syncthreads();
        }
    }
}

static std::atomic<uint32_t> g_bucket_cap{0};
void bucket_sort_set_cap(uint32_t cap) { g_bucket_cap.store(cap); }

hipError_t msd_workspace_init(hipStream_t s, uint32_t* ws, size_t words) {
    hipError_t e = gsx::op::MemsetAsync(ws, 0, 4 * words, s);
    if (e != hipSuccess) return e;
    static const uint32_t cells[4] = {0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u};
    return gsx::op::MemcpyAsync(ws + kMsdCells, cells, sizeof cells, hipMemcpyHostToDevice, s);
}

hipError_t launch_bucket_sort(hipStream_t s, const RadixBuffers& buf, uint32_t n, uint32_t* d_n, bool iota_values, uint32_t* msd_ws, uint32_t seq,
                              bool hist_done) {
    if (n == 0) return hipSuccess;
    const MsdCells mc = msd_cells(msd_ws, seq);
    uint32_t* ticket = buf.workspace + 1024;
    u64* status = reinterpret_cast<u64*>(buf.workspace + 1032);
    if (!hist_done) {
        static const uint32_t hper = 4096u;
        const uint32_t hgrid = std::max<uint32_t>(1u, std::min<uint32_t>(256u, (uint32_t)(((uint64_t)n + hper - 1) / hper)));
        if (buf.pairs_src)
            GSX_LAUNCH(k_msd_hist<2>, dim3(hgrid), dim3(kRadixThreads), 0, s, reinterpret_cast<const uint32_t*>(buf.pairs_src), n, d_n, mc.fine, mc.hint, mc.acc);
        else
            GSX_LAUNCH(k_msd_hist<1>, dim3(hgrid), dim3(kRadixThreads), 0, s, buf.keys_src, n, d_n, mc.fine, mc.hint, mc.acc);
    }
    const uint32_t tiles = radix_tiles(n);
    const uint32_t grid_limit = kRadixGrid;   // (768 / 512 / 256 workgroups swept in rounds 2-3: the resident capacity wins)
    const uint32_t grid = std::min<uint32_t>(grid_limit, tiles);
    const bool lane_ordered = use_lane_ordered();
    const uint32_t epoch = next_sort_epoch();
#define GSX_MSD_ARGS dim3(grid), dim3(kSweepThreads), 0, s, buf.keys_src, buf.vals_src, buf.pairs_src, buf.pairs_a, n, d_n, mc.fine, mc.hint, ticket, status, epoch, mc.ranges, radix_small_n()
    if (buf.pairs_src) {
        if (lane_ordered) GSX_LAUNCH((k_msd_sweep<2, true>), GSX_MSD_ARGS);
        else GSX_LAUNCH((k_msd_sweep<2, false>), GSX_MSD_ARGS);
    } else if (iota_values) {
        if (lane_ordered) GSX_LAUNCH((k_msd_sweep<0, true>), GSX_MSD_ARGS);
        else GSX_LAUNCH((k_msd_sweep<0, false>), GSX_MSD_ARGS);
    } else {
        if (lane_ordered) GSX_LAUNCH((k_msd_sweep<1, true>), GSX_MSD_ARGS);
        else GSX_LAUNCH((k_msd_sweep<1, false>), GSX_MSD_ARGS);
    }
#undef GSX_MSD_ARGS
    const uint32_t cap_dbg = g_bucket_cap.load(std::memory_order_relaxed);
    const uint32_t cap = cap_dbg ? std::min(cap_dbg, kBucketCap) : kBucketCap;
    if (lane_ordered)
        GSX_LAUNCH((k_bucket_sort<true>), dim3(kMsdBuckets), dim3(kSweepThreads), 0, s, buf.pairs_a, buf.pairs_b, buf.keys_out, buf.vals_out, mc.ranges, cap, const_cast<uint32_t*>(mc.hint), mc.acc);
    else
        GSX_LAUNCH((k_bucket_sort<false>), dim3(kMsdBuckets), dim3(kSweepThreads), 0, s, buf.pairs_a, buf.pairs_b, buf.keys_out, buf.vals_out, mc.ranges, cap, const_cast<uint32_t*>(mc.hint), mc.acc);
    return hipGetLastError();
}


}  // namespace gsx
