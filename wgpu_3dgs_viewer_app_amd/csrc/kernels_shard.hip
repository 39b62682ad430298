// kernels_shard.hip — multi-GPU exchange support for gfx950: pack projected records by destination
// rank, and import received records as a rank's per-frame record set.
//
// No reference counterpart (the reference is single-device, src/main.rs:85-98).  The Gaussian array is
// sharded by splat index; the screen is cut into `world` contiguous bands of tile rows, band g = rank g (BandEdges: equal, or
// balanced by the previous frame's per-row work).  After the projection
// pass each rank sends every visible record to the rank(s) whose tile rows its rectangle touches
// (RCCL all-to-all, done by the host layer), so that compositing — which is order dependent per
// pixel — happens with ALL splats of a pixel on one GPU, in global depth order.
// The exchange is speculative: every tile has a depth-key WINDOW [lo, hi) (host layer: hi = a margin behind the
// depth at which the tile's neighbourhood saturated last frame, unbounded for tiles expected to stay open); a
// record travels to a band only if some tile it touches there has its key inside the window, and the importing
// side bins it into exactly those tiles (kernels_bin.hip, same predicate).  A verification step sends what a
// wrongly predicted tile still misses as a second set of windows [hi, inf).  Whatever the windows, every tile
// composites a gap-free depth prefix, so the pixels come out identical.
// Packing is an order-preserving multi-destination stream compaction (records keep ascending local
// index inside every destination group), which keeps the depth-tie order = global Gaussian index.
// Record = 48 bytes: {mean.x, mean.y, rect.x, rect.y | conic a, b, c, opacity | r, g, b, depth}.
#include <algorithm>

#include "gsx_internal.h"
#include "window_scan.h"

namespace gsx {

constexpr int kPackThreads = 256;
constexpr int kMaxWorld = 64;
// Records per workgroup = 256 x ROUNDS.  16 rounds (4096 records) when the whole shard is scanned; 2 rounds (512) when only a
// candidate list is — a few per cent of the shard, compacted at the front: with 4096-record tiles 0.3 M candidates kept 73
// workgroups busy on 256 CUs (47 + 51 us for count + scatter); the grid is sized for the shard, the workgroups past the list
// return at once.
// A shard of at most kPackSmallShard records (an 8-GPU rank's share of cfg5's 6 M-Gaussian models: 0.75 M) is scanned in 1024-record
// tiles: at 4096 its 183 workgroups left three CUs in four idle and the repair round's count + scatter took 70 + 46 us (round 5).
constexpr uint32_t kPackRoundsFull = 16, kPackRoundsSmall = 4, kPackRoundsList = 2;
constexpr uint64_t kPackSmallShard = 1u << 21;

size_t pack_blocks(uint64_t n, uint32_t rounds) { return (size_t)((n + kPackThreads * rounds - 1) / (kPackThreads * rounds)); }
uint32_t pack_rounds(bool candidate_list, uint64_t n) { return candidate_list ? kPackRoundsList : (n <= kPackSmallShard ? kPackRoundsSmall : kPackRoundsFull); }

// per record: destination mask (stored for the scatter pass); per workgroup and destination: record count
// list (nullable): pack only the candidates list[0 .. *d_list_n) = (key, index) pairs (the records a lazily projected
// shard admitted); element e of every per-record array below then refers to list position e.
// travellers (nullable): ballots of the elements that travel anywhere + their count per workgroup (feeds the shading of
// a lazily projected shard's repair round).
template <int kPackRounds>
__global__ __launch_bounds__(kPackThreads) void k_pack_count(const uint32_t* __restrict__ key,
                                                              const float4* __restrict__ rec_a, uint32_t n,
                                                              const BandEdges bands,
                                                              const uint2* __restrict__ window, uint32_t tiles_x,
                                                              unsigned long long* __restrict__ masks,
                                                              uint32_t* __restrict__ table, uint32_t nblocks,
                                                              const uint2* __restrict__ list, const uint32_t* __restrict__ d_list_n,
                                                              unsigned long long* __restrict__ travellers,
                                                              uint32_t* __restrict__ traveller_counts,
                                                              const uint32_t* __restrict__ gate, uint32_t gate_row_words,
                                                              const WindowPyramid pyr, const uint32_t* __restrict__ rect8,
                                                              const uint32_t* __restrict__ d_skip) {
    __shared__ uint32_t cnt[kMaxWorld];
    __shared__ uint32_t tcnt[kPackThreads / 64];
    constexpr uint32_t kPackTile = kPackThreads * kPackRounds, kPackWaveChunk = 64 * kPackRounds;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t world = bands.world;
    // an always-enqueued repair round with nothing to repair (*d_skip == 0: no tile needs anything): nothing is read, nothing
    // written — the row scan, the headers and the scatter behind look at the same word and take every total for 0
    if (d_skip && *d_skip == 0u) return;
    if (tid < kMaxWorld) cnt[tid] = 0;
    __syncthreads();
    if (list) n = min(n, *d_list_n);
    const uint32_t base = blockIdx.x * kPackTile + wave * kPackWaveChunk;
    if (blockIdx.x * kPackTile >= n) {  // (uniform per workgroup) nothing here — a candidate list is a small fraction of the shard: a zero table column
        if (tid < world) table[tid * nblocks + blockIdx.x] = 0u;
        if (travellers && tid == 0) traveller_counts[blockIdx.x] = 0u;
        return;
    }
    uint32_t tc = 0;
    // The loads of all rounds are issued before any is consumed: keys / list entries, then the rectangles they point at.  With
    // one round at a time a wave paid three dependent memory round trips per round, sixteen times over — and a candidate list
    // (a few per cent of the shard) leaves too few waves on the chip to hide any of it: 200-270 us for 0.7 M candidates.
    uint32_t kks[kPackRounds], srcs[kPackRounds], rxs[kPackRounds], rys[kPackRounds];
#pragma unroll
    for (int r = 0; r < kPackRounds; ++r) {
        const uint32_t e = base + r * 64 + lane;
        kks[r] = kCulledKey;
        srcs[r] = e;
        if (e < n) {
            if (list) {
                const uint2 p = list[e];
                kks[r] = p.x;
                srcs[r] = p.y;
            } else {
                kks[r] = key[e];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kPackRounds; ++r) {
        rxs[r] = rys[r] = 0;
        if (kks[r] != kCulledKey) rec_rect(rec_a, rect8, srcs[r], rxs[r], rys[r]);
    }
#pragma unroll
    for (int r = 0; r < kPackRounds; ++r) {
        const uint32_t e = base + r * 64 + lane;
        const uint32_t kk = kks[r], rx = rxs[r], ry = rys[r];
        const unsigned long long m = pyr.data ? dest_mask_pyramid(pyr, kk, rx, ry, bands)
                                              : wave_dest_mask(window, tiles_x, kk, rx, ry, bands, gate, gate_row_words);
        if (e < n) masks[e] = m;
        if (travellers) {
            const unsigned long long any = __ballot(m != 0ull);
            if (lane == 0 && base + r * 64 < n) travellers[(base + r * 64) >> 6] = any;
            tc += (uint32_t)__popcll(any);
        }
        for (uint32_t g = 0; g < world; ++g) {
            unsigned long long bal = __ballot((m >> g) & 1ull);
            if (lane == 0 && bal) atomicAdd(&cnt[g], (uint32_t)__popcll(bal));
        }
    }
    if (travellers && lane == 0) tcnt[wave] = tc;
    __syncthreads();
    if (tid < world) table[tid * nblocks + blockIdx.x] = cnt[tid];
    if (travellers && tid == 0) traveller_counts[blockIdx.x] = tcnt[0] + tcnt[1] + tcnt[2] + tcnt[3];
}

// table rows were scanned exclusively in place (k_radix_rowscan), totals[g] = records for destination g
template <int kPackRounds>
__global__ __launch_bounds__(kPackThreads) void k_pack_scatter(const unsigned long long* __restrict__ masks,
                                                                const float4* __restrict__ rec_a,
                                                                const float4* __restrict__ rec_b,
                                                                const float4* __restrict__ rec_c, uint32_t n,
                                                                uint32_t world, const uint32_t* __restrict__ table,
                                                                uint32_t nblocks, const uint32_t* __restrict__ totals,
                                                                float4* __restrict__ send, uint64_t capacity,
                                                                const uint2* __restrict__ list, const uint32_t* __restrict__ d_list_n,
                                                                const SlotSpans sp, uint32_t slotted, const uint32_t* __restrict__ d_skip) {
    if (d_skip && *d_skip == 0u) return;  // (uniform) nothing was counted: the masks are stale
    __shared__ uint32_t run[kPackThreads / 64][kMaxWorld];  // per-wave running counts -> absolute offsets
    __shared__ uint32_t dbase[kMaxWorld];
    __shared__ uint32_t dend[kMaxWorld];   // first position past what destination g may hold
    constexpr uint32_t kPackTile = kPackThreads * kPackRounds, kPackWaveChunk = 64 * kPackRounds;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    if (tid < kMaxWorld) {
        if (slotted) {      // slots (device-resident exchange): slot g = [header | up to sp.cap[g] records] at record sp.off[g]
            dbase[tid] = sp.off[tid] + 1u;
            dend[tid] = sp.off[tid] + 1u + sp.cap[tid];
        } else {            // packed groups in rank order (exact split sizes known to the host)
            uint32_t b = 0;
            for (uint32_t g = 0; g < tid && g < world; ++g) b += totals[g];
            dbase[tid] = b;
            dend[tid] = 0xFFFFFFFFu;
        }
    }
    for (int w = 0; w < kPackThreads / 64; ++w)
        if (tid < kMaxWorld) run[w][tid] = 0;
    __syncthreads();
    if (list) n = min(n, *d_list_n);
    if (blockIdx.x * kPackTile >= n) return;  // uniform per workgroup (before any barrier that matters: the ones above were passed)
    const uint32_t base = blockIdx.x * kPackTile + wave * kPackWaveChunk;
    // the destination masks of all rounds first (independent loads), then pass 1: per-wave counts
    unsigned long long ms[kPackRounds];
#pragma unroll
    for (int r = 0; r < kPackRounds; ++r) {
        const uint32_t e = base + r * 64 + lane;
        ms[r] = e < n ? masks[e] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < kPackRounds; ++r) {
        const unsigned long long m = ms[r];
        for (uint32_t g = 0; g < world; ++g) {
            unsigned long long bal = __ballot((m >> g) & 1ull);
            if (lane == 0) run[wave][g] += (uint32_t)__popcll(bal);
        }
    }
    __syncthreads();
    if (tid < world) {  // wave-exclusive prefix + workgroup offset + destination base
        uint32_t o = dbase[tid] + table[tid * nblocks + blockIdx.x];
        for (int w = 0; w < kPackThreads / 64; ++w) {
            uint32_t c = run[w][tid];
            run[w][tid] = o;
            o += c;
        }
    }
    __syncthreads();
    // pass 2: ranks inside the wave, in (round, lane) = memory order
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kPackRounds; ++r) {
        uint32_t e = base + r * 64 + lane;
        const unsigned long long m = ms[r];
        float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
        if (m) {
            const uint32_t src = list ? list[e].y : e;
            a = rec_a[src];
            b = rec_b[src];
            c = rec_c[src];
        }
        for (uint32_t g = 0; g < world; ++g) {
            const bool hit = (m >> g) & 1ull;
            unsigned long long bal = __ballot(hit);
            uint32_t o = run[wave][g];
            __builtin_amdgcn_wave_barrier();
            if (hit) {
                uint64_t pos = (uint64_t)o + (uint32_t)__popcll(bal & lt);
                if (pos < capacity && pos < dend[g]) {
                    send[3 * pos + 0] = a;
                    send[3 * pos + 1] = b;
                    send[3 * pos + 2] = c;
                }
            }
            if (lane == 0) run[wave][g] = o + (uint32_t)__popcll(bal);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// received AoS records -> the record planes of a model; key = bit pattern of the depth
__global__ __launch_bounds__(256) void k_import_records(const float4* __restrict__ recv, uint32_t n, Records rec) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 a = recv[3 * (size_t)i], b = recv[3 * (size_t)i + 1], c = recv[3 * (size_t)i + 2];
    rec.a[i] = a;
    rec.b[i] = b;
    rec.c[i] = c;
    rec.key[i] = __float_as_uint(c.w);
}

hipError_t launch_pack_count(hipStream_t s, const Records& rec, uint32_t n, const BandEdges& bands,
                             const uint2* window, uint32_t tiles_x, unsigned long long* masks, uint32_t* table,
                             const uint2* list, const uint32_t* d_list_n, unsigned long long* travellers, uint32_t* traveller_counts,
                             const uint32_t* gate, uint32_t gate_row_words, const WindowPyramid* pyramid, const uint32_t* d_skip) {
    const uint32_t rounds = pack_rounds(list != nullptr, n), nb = (uint32_t)pack_blocks(n, rounds);
    if (nb) {
        auto kernel = rounds == kPackRoundsList ? k_pack_count<(int)kPackRoundsList>
                      : (rounds == kPackRoundsSmall ? k_pack_count<(int)kPackRoundsSmall> : k_pack_count<(int)kPackRoundsFull>);
        GSX_LAUNCH(kernel, dim3(nb), dim3(kPackThreads), 0, s, rec.key, rec.a, n, bands, window,
                           tiles_x, masks, table, nb, list, d_list_n, travellers, traveller_counts, gate, gate_row_words,
                           pyramid ? *pyramid : WindowPyramid{}, rec.rect8, d_skip);
    }
    return hipGetLastError();
}

hipError_t launch_pack_scatter(hipStream_t s, const Records& rec, uint32_t n, uint32_t world,
                               const unsigned long long* masks, const uint32_t* table, const uint32_t* totals, void* d_send,
                               uint64_t capacity, const uint2* list, const uint32_t* d_list_n, const SlotSpans* slots, const uint32_t* d_skip) {
    const uint32_t rounds = pack_rounds(list != nullptr, n), nb = (uint32_t)pack_blocks(n, rounds);
    if (nb) {
        auto kernel = rounds == kPackRoundsList ? k_pack_scatter<(int)kPackRoundsList>
                      : (rounds == kPackRoundsSmall ? k_pack_scatter<(int)kPackRoundsSmall> : k_pack_scatter<(int)kPackRoundsFull>);
        GSX_LAUNCH(kernel, dim3(nb), dim3(kPackThreads), 0, s, masks, rec.a, rec.b, rec.c, n, world, table, nb,
                           totals, reinterpret_cast<float4*>(d_send), capacity, list, d_list_n, slots ? *slots : SlotSpans{}, slots ? 1u : 0u, d_skip);
    }
    return hipGetLastError();
}

// ---- device-resident exchange: slots, counts in the slot headers (no host round trip) ----
// send / recv buffer of a round: `world` slots; slot p = a header record followed by up to cap[p] records of 48 bytes, at record
// off[p] (SlotSpans, gsx_internal.h).  The sizes are the same on both ends of a pair because every rank derives them from gathered
// data (uniform: the busiest pair of the last frame; pair by pair: the last frame's count matrix).  Header: word 0 = records the
// sender HAD for this destination, word 1 = records it sent = min(word 0, cap).
__global__ __launch_bounds__(64) void k_pack_headers(const uint32_t* __restrict__ totals, uint32_t world, const SlotSpans sp,
                                                      float4* __restrict__ send, SlabStats* __restrict__ stats, uint32_t round,
                                                      const uint32_t* __restrict__ d_skip) {
    const uint32_t g = threadIdx.x;
    const bool skip = d_skip && *d_skip == 0u;  // an always-enqueued repair round with nothing to repair: empty slots
    uint32_t cnt = (g < world && !skip) ? totals[g] : 0u;
    uint32_t over = 0u;
    if (g < world) {
        float4* h = send + 3ull * (size_t)sp.off[g];
        h[0] = make_float4(__uint_as_float(cnt), __uint_as_float(min(cnt, sp.cap[g])), 0.0f, 0.0f);
        h[1] = make_float4(0, 0, 0, 0);
        h[2] = make_float4(0, 0, 0, 0);
        over = cnt > sp.cap[g] ? 1u : 0u;
        if (round == 0u) stats->slot_want[g] = cnt;     // (the gathered count matrix sizes the next frame's slots pair by pair)
    }
    uint32_t mx = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        over |= (uint32_t)__shfl_xor((int)over, o, 64);
    }
    if (g == 0) {
        stats->slot_max[round & 1u] = mx;           // what this rank WANTED to send to its busiest destination
        stats->slot_over[round & 1u] = over;        // ... and whether some destination's slot was too small for what it was owed
    }
}

// received slots -> the record planes of the model, compacted in (source rank, source index) order = global Gaussian
// index order (what makes the stable depth sort break ties exactly as on one GPU); the record count stays on the device
__global__ __launch_bounds__(256) void k_import_slots(const float4* __restrict__ recv, uint32_t world, const SlotSpans sp, Records rec,
                                                       SlabStats* __restrict__ stats) {
    __shared__ uint32_t pre[kMaxWorld + 1];    // records imported from the sources before s
    __shared__ uint32_t capre[kMaxWorld + 1];  // thread index space: capacities of the sources before s
    if (threadIdx.x == 0) {
        uint32_t acc = 0, cacc = 0;
        for (uint32_t s = 0; s < world; ++s) {
            pre[s] = acc;
            capre[s] = cacc;
            acc += min(__float_as_uint(recv[3ull * (size_t)sp.off[s]].y), sp.cap[s]);
            cacc += sp.cap[s];
        }
        pre[world] = acc;
        capre[world] = cacc;
        if (blockIdx.x == 0) {
            stats->n_visible = acc;  // every imported record is visible by construction
            stats->n_sorted = acc;
        }
    }
    __syncthreads();
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= capre[world]) return;
    uint32_t s = 0;
    while (s + 1u < world && e >= capre[s + 1u]) ++s;
    const uint32_t j = (uint32_t)(e - capre[s]);
    if (j >= pre[s + 1] - pre[s]) return;
    const float4* r = recv + 3ull * ((size_t)sp.off[s] + 1u + j);
    const uint32_t i = pre[s] + j;
    const float4 a = r[0], b = r[1], c = r[2];
    rec.a[i] = a;
    rec.b[i] = b;
    rec.c[i] = c;
    rec.key[i] = __float_as_uint(c.w);
}

// per-tile limits -> round-1 windows [0, limit)
__global__ __launch_bounds__(256) void k_limits_to_windows(const uint32_t* __restrict__ limit, uint32_t n_tiles, uint2* __restrict__ win) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_tiles) win[t] = make_uint2(0u, limit[t]);
}

// Verification on the device, from the all-gathered feedback (layout: gsx_internal.h, feedback_*): a tile whose window was
// bounded and that is still open gets, in the second exchange, what it was refused: [limit, inf); every other tile nothing.
// limit == nullptr (the round had no windows): nothing to repair.
// The band edges arrive as a by-value kernel argument.  A workgroup copies them to LDS once (edges_to_lds: blockDim.x > kMaxRanks) and
// every lookup indexes LDS: indexing the argument itself with a per-lane index can compile into a select chain over all 65 words PER
// ACCESS, with the scalar registers spilled around it — k_shard_verify was 33 000 instructions and 78 us at 3840x2160 that way (round 5).
__device__ inline void edges_to_lds(uint32_t* s_e, const BandEdges& b) {
    if (threadIdx.x <= (uint32_t)kMaxRanks) s_e[threadIdx.x] = b.e[threadIdx.x];
}
__device__ inline uint32_t band_of_edges(const uint32_t* e, uint32_t world, uint32_t ty) {  // (band_of, gsx_internal.h)
    uint32_t g = 0;
    while (g + 1u < world && ty >= e[g + 1u]) ++g;
    return g;
}
// where the gathered saturation keys of tile row ty start
__device__ inline uint32_t sat_row_base(const uint32_t* e, uint32_t world, uint32_t ty, uint32_t tiles_x, uint32_t stride) {
    const uint32_t g = band_of_edges(e, world, ty);
    return g * stride + kShardExtraWords + (ty - e[g]) * tiles_x;
}

// this rank's feedback piece: statistics, the saturation keys of its band, the work of its tile rows.
// done_before (nullable; layered models): the tiles nearer models had saturated before this model was composited.  They say
// nothing about THIS model's depths: reported as saturated at the smallest depth key, so that the model's next limit there
// is whatever its neighbourhood needs and nothing more (the single-GPU rule, k_spec_next).
__global__ __launch_bounds__(256) void k_shard_feedback(const uint32_t* __restrict__ tile_sat, const uint32_t* __restrict__ row_work, uint32_t tiles_x,
                                                         uint32_t tiles_y, uint32_t row_lo, uint32_t rows, uint32_t* __restrict__ out,
                                                         const SlabStats* __restrict__ stats, const uint32_t* __restrict__ done_before,
                                                         uint32_t row_words, uint32_t gather_root_plus1, uint32_t policy_flags,
                                                         uint32_t* __restrict__ za, uint32_t nza, uint32_t* __restrict__ zb, uint32_t nzb) {
    // (za / zb: the verification's counters and its need bitmap, zeroed on the way for the launch that follows the gather — one launch less)
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < max(nza, nzb); k += gridDim.x * blockDim.x) {
        if (k < nza) za[k] = 0u;
        if (k < nzb) zb[k] = 0u;
    }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_sat = rows * tiles_x;
    if (i >= kShardExtraWords + n_sat + rows) return;
    if (i < kShardExtraWords) {  // round 0's figures (k_pack_headers) + what this rank binned
        uint32_t x = 0;
        if (i == 0u) x = stats->slot_max[0];
        else if (i == 1u) x = stats->slot_over[0];
        else if (i == 2u) x = gather_root_plus1;
        else if (i == 3u) x = stats->n_entries_total;
        else if (i == 4u) x = stats->slot_max[1];    // the repair round's figures (meaningful in the feedback gathered AFTER that round)
        else if (i == 5u) x = stats->slot_over[1];
        else if (i == 6u) x = policy_flags;          // how this rank sizes slots and bands: ranks that disagree must not exchange
        else if (i >= 8u) x = stats->slot_want[i - 8u];
        out[i] = x;
        return;
    }
    if (i >= kShardExtraWords + n_sat) {
        const uint32_t ty = row_lo + (i - kShardExtraWords - n_sat);
        out[i] = (row_work && ty < tiles_y) ? row_work[ty] : 0u;
        return;
    }
    const uint32_t k = i - kShardExtraWords;
    const uint32_t ty = row_lo + k / tiles_x, tx = k % tiles_x;
    uint32_t s = ty < tiles_y ? tile_sat[ty * tiles_x + tx] : 0u;
    if (done_before && ty < tiles_y && ((done_before[ty * row_words + (tx >> 5)] >> (tx & 31u)) & 1u)) s = 1u;
    out[i] = s;
}

// The verdict arithmetic by ONE thread, row by row: the statement the parallel tail of k_shard_verify must reproduce, and what runs for
// a frame of more than 1024 tile rows.
// (the edges through a pointer — LDS in k_shard_verify's last workgroup: indexing the by-value kernel argument with a computed index from
//  a function that is not inlined would make the compiler copy the whole struct to scratch memory)
__device__ inline uint32_t work_at_edges(const uint32_t* __restrict__ sat, uint32_t ty, uint32_t tiles_x, uint32_t world, const uint32_t* e, uint32_t stride) {
    const uint32_t g = band_of_edges(e, world, ty);
    return sat[(size_t)g * stride + kShardExtraWords + (e[g + 1u] - e[g]) * tiles_x + (ty - e[g])];
}
__device__ __noinline__ void verdict_tail_serial(const uint32_t* __restrict__ sat, uint32_t tiles_x, uint32_t tiles_y, uint32_t world, const uint32_t* e,
                                                 uint32_t stride, uint32_t* __restrict__ hv, unsigned long long* __restrict__ host_verdict,
                                                 uint32_t seq, const uint32_t* __restrict__ d_need, uint32_t balance) {
    {
        auto work = [&](uint32_t ty) -> uint32_t { return work_at_edges(sat, ty, tiles_x, world, e, stride); };
        const uint32_t total = __hip_atomic_load(d_need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t gmax = 0, over = 0, root_bad = 0, ent_sum = 0, ent_max = 0;
        const uint32_t root0 = sat[2], flags0 = sat[6];
        for (uint32_t g = 0; g < world; ++g) {
            const uint32_t* x = sat + (size_t)g * stride;
            gmax = max(gmax, x[0]);
            over |= x[1];
            root_bad |= x[2] != root0 ? 1u : 0u;
            root_bad |= x[6] != flags0 ? 2u : 0u;  // (bit 1: the ranks size their slots / bands by different policies)
            ent_sum += x[3];
            ent_max = max(ent_max, x[3]);
        }
        hv[4] = root_bad;
        hv[5] = ent_sum;
        hv[6] = ent_max;
        {   // how evenly THIS frame's bands shared the work: busiest rank x world x 1000 / all
            unsigned long long all = 0, busiest = 0;
            for (uint32_t g = 0; g < world; ++g) {
                unsigned long long wg = 0;
                for (uint32_t ty = e[g]; ty < e[g + 1u] && ty < tiles_y; ++ty) wg += work(ty);
                all += wg;
                busiest = wg > busiest ? wg : busiest;
            }
            hv[7] = all ? (uint32_t)(busiest * world * 1000ull / all) : 1000u;
        }
        // next frame's bands: contiguous runs of tile rows of (as nearly as rows allow) equal work.  A row weighs what its tiles
        // cost this frame (tile_work, gsx_internal.h) plus kTileWork for every tile (the ones that were not composited at all).
        // Hysteresis: an edge that moves shifts what every pair exchanges (the slots sized pair by pair from this frame's counts
        // would overflow for nothing), and bands a few rows tall cannot be tuned finer than a row.  New edges are adopted only
        // when the busiest rank carries more than kBalanceKeep x the mean AND they would have shared THIS frame's work at
        // least a tenth better.
        uint32_t* ne = hv + kVerdictEdges;
        bool adopt = false;
        if (balance && hv[7] > kBalanceKeepPermille) {
            unsigned long long W = 0;
            for (uint32_t ty = 0; ty < tiles_y; ++ty) W += (unsigned long long)work(ty) + (unsigned long long)kTileWork * tiles_x;
            unsigned long long acc = 0, band_w = 0, worst = 0;
            uint32_t g = 1;
            ne[0] = 0u;
            for (uint32_t ty = 0; ty < tiles_y; ++ty) {
                const unsigned long long w = (unsigned long long)work(ty) + (unsigned long long)kTileWork * tiles_x;
                // edge g goes in front of row ty if that is at least as close to g / world of the work as behind it
                while (g < world && (acc * world >= g * W || 2ull * (g * W - acc * world) <= w * world)) {
                    ne[g++] = ty;
                    worst = band_w > worst ? band_w : worst;
                    band_w = 0;
                }
                acc += w;
                band_w += w;
            }
            worst = band_w > worst ? band_w : worst;
            while (g < world) ne[g++] = tiles_y;
            ne[world] = tiles_y;
            // the same measure for the bands in force (rows' weights as above)
            unsigned long long cur_worst = 0;
            for (uint32_t gg = 0; gg < world; ++gg) {
                unsigned long long wg = 0;
                for (uint32_t ty = e[gg]; ty < e[gg + 1u] && ty < tiles_y; ++ty)
                    wg += (unsigned long long)work(ty) + (unsigned long long)kTileWork * tiles_x;
                cur_worst = wg > cur_worst ? wg : cur_worst;
            }
            adopt = worst * 10ull <= cur_worst * 9ull;
        }
        if (!adopt)
            for (uint32_t g = 0; g <= world; ++g) ne[g] = e[g];
        __hip_atomic_store(host_verdict + 1, ((unsigned long long)gmax << 32) | (over ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_verdict, ((unsigned long long)seq << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void k_shard_verify(const uint32_t* __restrict__ limit, const uint32_t* __restrict__ sat, uint32_t n_tiles,
                                                       uint32_t tiles_x, const BandEdges bands, uint32_t stride, uint2* __restrict__ win2,
                                                       uint32_t* __restrict__ d_need, uint32_t* __restrict__ ticket,
                                                       unsigned long long* __restrict__ host_verdict, uint32_t seq,
                                                       uint32_t* __restrict__ need_bits, uint32_t balance) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t world = bands.world;
    // where the gathered saturation keys of the tile rows this workgroup touches start (one walk over the band edges per ROW, by the
    // first threads, instead of one per tile: the walk is a chain of dependent loads)
    __shared__ uint32_t s_base[40], s_e[kMaxRanks + 1];
    edges_to_lds(s_e, bands);
    __syncthreads();
    const uint32_t first = (blockIdx.x * 256u) / tiles_x;
    if (threadIdx.x < 40u) {
        const uint32_t y = first + threadIdx.x;
        s_base[threadIdx.x] = y * tiles_x < n_tiles ? sat_row_base(s_e, world, y, tiles_x, stride) : 0u;
    }
    __syncthreads();
    bool need = false;
    if (t < n_tiles) {
        const uint32_t lim = limit ? limit[t] : 0xFFFFFFFFu;
        const uint32_t tx = t % tiles_x, ty = t / tiles_x;
        const uint32_t k = ty - first;
        need = lim < 0xFFFFFFFFu && sat[(k < 40u ? s_base[k] : sat_row_base(s_e, world, ty, tiles_x, stride)) + tx] == 0u;
        win2[t] = need ? make_uint2(lim, 0xFFFFFFFFu) : make_uint2(0u, 0u);
        if (need) atomicOr(&need_bits[ty * ((tiles_x + 31u) / 32u) + (tx >> 5)], 1u << (tx & 31u));
    }
    __shared__ uint32_t s_need, s_last;
    if (threadIdx.x == 0) s_need = 0;
    __syncthreads();
    const unsigned long long bal = __ballot(need);
    if ((threadIdx.x & 63u) == 0 && bal) atomicAdd(&s_need, (uint32_t)__popcll(bal));
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_need) atomicAdd(d_need, s_need);
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // The last block posts the verdict of round 0 (layout: gsx_internal.h, kVerdict*):
    //   word 1 = {largest per-destination record count over all ranks | any rank's slot overflowed}
    //   word 0 = {seq | tiles that need the repair round}     (release store: a host that waits polls this one)
    //   and behind them: the count matrix (slot sizes of the next frame, pair by pair), the band edges of the next frame
    //   (balanced by the rows' work), whether the ranks agree about the gather root and the slot policy, the ranks' list entries.
    // Every input is globally gathered, so every rank posts the same verdict and takes the same decisions.  All 256 threads fetch —
    // the rows' work into LDS, the matrix straight through — and one thread does the arithmetic on what they fetched (round 5: the
    // one thread walking ~200 dependent loads was 16 us on the critical path of every sharded frame).
    uint32_t* hv = reinterpret_cast<uint32_t*>(host_verdict);
    const uint32_t tiles_y = n_tiles / tiles_x;
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = tid; k < world * world; k += 256u) hv[kVerdictMatrix + k] = sat[(size_t)(k / world) * stride + 8u + k % world];
    if (tiles_y > 1024u || (balance & 2u)) {  // (a frame taller than 16384 pixels: the rows do not fit the tables below; or a test asks)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) verdict_tail_serial(sat, tiles_x, tiles_y, world, s_e, stride, hv, host_verdict, seq, d_need, balance & 1u);
        if (tid == 0) *ticket = 0;
        return;
    }
    // Rounds 1-4 had one thread walk the rows four times (the work shares, the total, the new edges, the bands in force): 14 us at
    // 1080p and 61 us at 3840x2160 on the critical path of every model of every sharded frame.  Now: the rows' work and its running
    // sum in LDS (s_pre[ty] = work of the rows before ty), and every figure a difference of two entries; edge g of the next frame
    // is the FIRST row in front of which the serial rule would put it (the rule is monotone in g: a row that takes edge g has taken
    // every edge before it), found by all threads at once.  Same 64-bit integer arithmetic, same edges.
    __shared__ unsigned long long s_pre[1025], s_tot[2][256], s_band[kMaxRanks], s_new[kMaxRanks];
    __shared__ uint32_t s_edge[kMaxRanks + 1], s_stat[8];
    {
        unsigned long long loc[4], run = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t ty = 4u * tid + u;
            run += ty < tiles_y ? (unsigned long long)work_at_edges(sat, ty, tiles_x, world, s_e, stride) : 0ull;
            loc[u] = run;
        }
        s_tot[0][tid] = run;
        __syncthreads();
        int cur = 0;
        for (uint32_t d = 1; d < 256u; d <<= 1) {  // inclusive scan of the threads' totals
            s_tot[cur ^ 1][tid] = s_tot[cur][tid] + (tid >= d ? s_tot[cur][tid - d] : 0ull);
            cur ^= 1;
            __syncthreads();
        }
        const unsigned long long before = tid ? s_tot[cur][tid - 1u] : 0ull;
        if (tid == 0) s_pre[0] = 0ull;
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) s_pre[4u * tid + u + 1u] = before + loc[u];
    }
    if (tid < 64u) {  // the ranks' statistics (wave 0)
        const bool in = tid < world;
        const uint32_t* x = sat + (size_t)(in ? tid : 0u) * stride;
        const uint32_t root0 = sat[2], flags0 = sat[6];
        uint32_t gmax = in ? x[0] : 0u, over = in ? x[1] : 0u;
        uint32_t bad = in ? ((x[2] != root0 ? 1u : 0u) | (x[6] != flags0 ? 2u : 0u)) : 0u;  // (bit 1: the ranks size their slots / bands by different policies)
        uint32_t ent_sum = in ? x[3] : 0u, ent_max = ent_sum;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gmax = max(gmax, (uint32_t)__shfl_xor((int)gmax, o, 64));
            over |= (uint32_t)__shfl_xor((int)over, o, 64);
            bad |= (uint32_t)__shfl_xor((int)bad, o, 64);
            ent_sum += (uint32_t)__shfl_xor((int)ent_sum, o, 64);
            ent_max = max(ent_max, (uint32_t)__shfl_xor((int)ent_max, o, 64));
        }
        if (tid == 0) {
            s_stat[0] = gmax;
            s_stat[1] = over;
            hv[4] = bad;
            hv[5] = ent_sum;
            hv[6] = ent_max;
        }
    }
    if (tid <= world) s_edge[tid] = tid == 0 ? 0u : tiles_y;
    __syncthreads();
    const unsigned long long kw = (unsigned long long)kTileWork * tiles_x;  // every tile of a row weighs kTileWork besides what it cost
    auto pre_w = [&](uint32_t ty) -> unsigned long long { return s_pre[ty] + kw * ty; };  // weight of the rows before ty
    // how evenly THIS frame's bands shared the work: busiest rank x world x 1000 / all; the same with the rows' weights
    if (tid < world) {
        const uint32_t lo = min(s_e[tid], tiles_y), hi = max(lo, min(s_e[tid + 1u], tiles_y));
        s_band[tid] = s_pre[hi] - s_pre[lo];
        s_new[tid] = pre_w(hi) - pre_w(lo);
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long busiest = 0, cur_worst = 0;
        const unsigned long long all = s_pre[tiles_y];
        for (uint32_t g = 0; g < world; ++g) {
            busiest = s_band[g] > busiest ? s_band[g] : busiest;
            cur_worst = s_new[g] > cur_worst ? s_new[g] : cur_worst;
        }
        const uint32_t share = all ? (uint32_t)(busiest * world * 1000ull / all) : 1000u;
        hv[7] = share;
        s_stat[2] = share;
        s_band[0] = cur_worst;
    }
    __syncthreads();
    // next frame's bands: contiguous runs of tile rows of (as nearly as rows allow) equal work.  A row weighs what its tiles
    // cost this frame (tile_work, gsx_internal.h) plus kTileWork for every tile (the ones that were not composited at all).
    // Hysteresis: an edge that moves shifts what every pair exchanges (the slots sized pair by pair from this frame's counts
    // would overflow for nothing), and bands a few rows tall cannot be tuned finer than a row.  New edges are adopted only
    // when the busiest rank carries more than kBalanceKeep x the mean AND they would have shared THIS frame's work at
    // least a tenth better.
    const bool rebalance = (balance & 1u) && s_stat[2] > kBalanceKeepPermille;
    if (rebalance) {
        const unsigned long long W = pre_w(tiles_y), cur_worst = s_band[0];
        for (uint32_t ty = tid; ty < tiles_y; ty += 256u) {
            const unsigned long long acc = pre_w(ty), w = pre_w(ty + 1u) - acc;
            // edge g goes in front of row ty if that is at least as close to g / world of the work as behind it
            for (uint32_t g = 1; g < world && (acc * world >= g * W || 2ull * (g * W - acc * world) <= w * world); ++g) atomicMin(&s_edge[g], ty);
        }
        __syncthreads();
        if (tid < world) s_new[tid] = pre_w(s_edge[tid + 1u]) - pre_w(s_edge[tid]);
        __syncthreads();
        if (tid == 0) {
            unsigned long long worst = 0;
            for (uint32_t g = 0; g < world; ++g) worst = s_new[g] > worst ? s_new[g] : worst;
            s_stat[3] = worst * 10ull <= cur_worst * 9ull ? 1u : 0u;
        }
        __syncthreads();
    }
    const bool adopt = rebalance && s_stat[3] != 0u;
    if (tid <= world) hv[kVerdictEdges + tid] = adopt ? s_edge[tid] : s_e[tid];
    __threadfence_system();  // (the block may go to pinned host memory: visible before the word a host polls)
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = __hip_atomic_load(d_need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(host_verdict + 1, ((unsigned long long)s_stat[0] << 32) | (s_stat[1] ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_verdict, ((unsigned long long)seq << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        *ticket = 0;
    }
}

// Sizing the repair exchange: counts_all = per rank {records it has for its busiest destination, 0, 0, 0} (all-gathered);
// the global maximum goes to the host — the exact slot size of the repair round, the same on every rank.
__global__ __launch_bounds__(64) void k_shard_post_counts(const uint32_t* __restrict__ counts_all, uint32_t world,
                                                           unsigned long long* __restrict__ host_verdict, uint32_t seq) {
    uint32_t mx = threadIdx.x < world ? counts_all[4u * threadIdx.x] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (threadIdx.x == 0) {
        __hip_atomic_store(host_verdict + 1, (unsigned long long)mx << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_verdict, (unsigned long long)seq << 32, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The verdict of a frame whose repair round was enqueued unasked (gsx_shard_frame.cpp: no host look inside a frame).  k_shard_verify
// left the round-0 verdict in DEVICE memory (staged: same layout as the host block); this kernel, behind the repair round's feedback
// gather (sat: piece g at word g * stride; nullptr: the frame had no repair round), adds what that round reports — the largest count
// any rank had for one destination, whether a repair slot overflowed — and posts the whole block to the frame's slot of the host ring.
// Every input is gathered data: every rank posts the same block.
__global__ __launch_bounds__(256) void k_shard_post_verdict(const uint32_t* __restrict__ staged, const uint32_t* __restrict__ sat, uint32_t world,
                                                             uint32_t stride, uint32_t* __restrict__ host_block, uint32_t seq);

// local maximum of the per-destination totals -> out[0..3] = {max, 0, 0, 0}
__global__ __launch_bounds__(64) void k_shard_max_count(const uint32_t* __restrict__ totals, uint32_t world, uint32_t* __restrict__ out) {
    uint32_t mx = threadIdx.x < world ? totals[threadIdx.x] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (threadIdx.x < 4) out[threadIdx.x] = threadIdx.x == 0 ? mx : 0u;
}

// Next frame's per-tile depth-key limit from this frame's saturation keys: (1 + margin) x the deepest saturation depth in
// the tile's (2 radius + 1)^2 neighbourhood — the camera moves — and unbounded if any tile of the neighbourhood stayed open
// (parallel.next_limits is the numpy statement of the same policy).  Outside the frame counts as nothing.
// win_next (nullable): also the next frame's round-0 windows [0, limit) (saves that frame a launch); staged / host_block (nullable): block 0
// posts the frame's verdict on its way (post_verdict_body: what k_shard_post_verdict does as a launch of its own)
__device__ inline unsigned long long need_of_staged(const uint32_t* __restrict__ staged) { return (unsigned long long)staged[0]; }  // low half of word 0

__device__ inline void post_verdict_body(const uint32_t* __restrict__ staged, const uint32_t* __restrict__ sat, uint32_t world, uint32_t stride,
                                         uint32_t* __restrict__ host_block, uint32_t seq) {
    const uint32_t n = kVerdictMatrix + world * world;
    for (uint32_t i = 4u + threadIdx.x; i < n; i += 256u)
        if (i != kVerdictRepairMax && i != kVerdictRepairOver) host_block[i] = staged[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t mx = 0, over = 0;
        const unsigned long long need = need_of_staged(staged);
        if (sat && need)  // (a round that repaired nothing left its statistics untouched: the previous frame's)
            for (uint32_t g = 0; g < world; ++g) {
                mx = max(mx, sat[(size_t)g * stride + 4u]);
                over |= sat[(size_t)g * stride + 5u];
            }
        host_block[kVerdictRepairMax] = mx;
        host_block[kVerdictRepairOver] = over;
        __threadfence_system();
        unsigned long long* hv = reinterpret_cast<unsigned long long*>(host_block);
        const unsigned long long* sv = reinterpret_cast<const unsigned long long*>(staged);
        __hip_atomic_store(hv + 1, sv[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hv, ((unsigned long long)seq << 32) | need, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void k_shard_post_verdict(const uint32_t* __restrict__ staged, const uint32_t* __restrict__ sat, uint32_t world,
                                                             uint32_t stride, uint32_t* __restrict__ host_block, uint32_t seq) {
    post_verdict_body(staged, sat, world, stride, host_block, seq);
}

__global__ __launch_bounds__(256) void k_shard_next_limits(const uint32_t* __restrict__ sat, uint32_t tiles_x, uint32_t tiles_y, float gain,
                                                            int radius, uint32_t* __restrict__ limit, const BandEdges bands, uint32_t stride,
                                                            uint2* __restrict__ win_next, const uint32_t* __restrict__ staged,
                                                            const uint32_t* __restrict__ sat_verdict, uint32_t* __restrict__ host_block, uint32_t seq) {
    if (host_block && blockIdx.x == 0) post_verdict_body(staged, sat_verdict, bands.world, stride, host_block, seq);
    // where the tile rows this workgroup's neighbourhoods reach lie in the gathered pieces: one lookup per row instead of a walk over
    // the band edges per load (49 loads per tile at radius 3: the walk was most of the kernel's 14 us)
    __shared__ uint32_t s_row[48], s_e[kMaxRanks + 1];
    edges_to_lds(s_e, bands);
    __syncthreads();
    const uint32_t t0 = blockIdx.x * 256u, row0 = t0 / tiles_x;
    const int first = max((int)row0 - radius, 0);
    if (threadIdx.x < 48u) {
        const uint32_t y = (uint32_t)first + threadIdx.x;
        s_row[threadIdx.x] = y < tiles_y ? sat_row_base(s_e, bands.world, y, tiles_x, stride) : 0u;
    }
    __syncthreads();
    const uint32_t t = t0 + threadIdx.x;
    if (t >= tiles_x * tiles_y) return;
    const int tx = (int)(t % tiles_x), ty = (int)(t / tiles_x);
    float deepest = 0.0f;
    bool open = false;
    for (int y = max(ty - radius, 0); y <= min(ty + radius, (int)tiles_y - 1); ++y) {
        // (a workgroup's 256 tiles span at most 256 / tiles_x + 2 rows; with the radius on both sides that fits the table for every grid
        //  of at least 8 tiles a row — narrower frames take the walk)
        const uint32_t k = (uint32_t)(y - first);
        const uint32_t base = k < 48u ? s_row[k] : sat_row_base(s_e, bands.world, (uint32_t)y, tiles_x, stride);
        for (int x = max(tx - radius, 0); x <= min(tx + radius, (int)tiles_x - 1); ++x) {
            const uint32_t s = sat[base + (uint32_t)x];
            if (s == 0u) open = true;
            else deepest = fmaxf(deepest, __uint_as_float(s));
        }
    }
    uint32_t out = 0xFFFFFFFFu;
    if (!open) {
        const float lim = deepest * gain;
        out = (lim < 3.0e38f) ? max(__float_as_uint(lim), 1u) : 0xFFFFFFFFu;
    }
    limit[t] = out;
    if (win_next) win_next[t] = make_uint2(0u, out);
}

hipError_t launch_pack_headers(hipStream_t s, const uint32_t* totals, uint32_t world, const SlotSpans& slots, void* d_send, SlabStats* stats, uint32_t round,
                               const uint32_t* d_skip) {
    GSX_LAUNCH(k_pack_headers, dim3(1), dim3(64), 0, s, totals, world, slots, reinterpret_cast<float4*>(d_send), stats, round, d_skip);
    return hipGetLastError();
}

hipError_t launch_import_slots(hipStream_t s, const void* d_recv, uint32_t world, const SlotSpans& slots, const Records& rec, SlabStats* stats) {
    uint64_t total = 0;
    for (uint32_t p = 0; p < world; ++p) total += slots.cap[p];
    GSX_LAUNCH(k_import_slots, dim3((unsigned)std::max<uint64_t>((total + 255) / 256, 1)), dim3(256), 0, s,
               reinterpret_cast<const float4*>(d_recv), world, slots, rec, stats);
    return hipGetLastError();
}

hipError_t launch_limits_to_windows(hipStream_t s, const uint32_t* limit, uint32_t n_tiles, uint2* win) {
    GSX_LAUNCH(k_limits_to_windows, dim3((n_tiles + 255) / 256), dim3(256), 0, s, limit, n_tiles, win);
    return hipGetLastError();
}

hipError_t launch_shard_verify(hipStream_t s, const uint32_t* limit, const uint32_t* sat, uint32_t tiles_x, uint32_t tiles_y, const BandEdges& bands,
                               uint2* win2, uint32_t* d_need, uint32_t* d_ticket, unsigned long long* host_verdict, uint32_t seq,
                               uint32_t* need_bits, uint32_t balance) {
    const uint32_t n_tiles = tiles_x * tiles_y;
    // GSX_SHARD_VERIFY_SERIAL=1 (tests): the verdict arithmetic by the one-thread walk that frames of more than 1024 tile rows take
    // (bit 1 of `balance`): tests/test_gpu_shard_lib.py checks that both statements post the same edges and figures
    const char* serial = getenv("GSX_SHARD_VERIFY_SERIAL");
    if (serial && *serial == '1') balance |= 2u;
    GSX_LAUNCH(k_shard_verify, dim3((n_tiles + 255) / 256), dim3(256), 0, s, limit, sat, n_tiles, tiles_x, bands, feedback_stride(bands, tiles_x), win2,
               d_need, d_ticket, host_verdict, seq, need_bits, balance);
    return hipGetLastError();
}

hipError_t launch_shard_feedback(hipStream_t s, const uint32_t* tile_sat, const uint32_t* row_work, uint32_t tiles_x, uint32_t tiles_y, const BandEdges& bands,
                                 uint32_t rank, uint32_t* out, const SlabStats* stats, const uint32_t* done_before, uint32_t row_words, uint32_t gather_root_plus1,
                                 uint32_t policy_flags, uint32_t* za, uint32_t nza, uint32_t* zb, uint32_t nzb) {
    const uint32_t rows = bands.e[rank + 1u] - bands.e[rank], n = feedback_words(bands, tiles_x, rank);
    GSX_LAUNCH(k_shard_feedback, dim3((n + 255) / 256), dim3(256), 0, s, tile_sat, row_work, tiles_x, tiles_y, bands.e[rank], rows, out, stats, done_before,
               row_words, gather_root_plus1, policy_flags, za, nza, zb, nzb);
    return hipGetLastError();
}

hipError_t launch_shard_post_counts(hipStream_t s, const uint32_t* counts_all, uint32_t world, unsigned long long* host_verdict, uint32_t seq) {
    GSX_LAUNCH(k_shard_post_counts, dim3(1), dim3(64), 0, s, counts_all, world, host_verdict, seq);
    return hipGetLastError();
}

hipError_t launch_shard_post_verdict(hipStream_t s, const uint32_t* staged, const uint32_t* sat, uint32_t world, uint32_t stride, uint32_t* host_block, uint32_t seq) {
    GSX_LAUNCH(k_shard_post_verdict, dim3(1), dim3(256), 0, s, staged, sat, world, stride, host_block, seq);
    return hipGetLastError();
}

hipError_t launch_shard_max_count(hipStream_t s, const uint32_t* totals, uint32_t world, uint32_t* out4) {
    GSX_LAUNCH(k_shard_max_count, dim3(1), dim3(64), 0, s, totals, world, out4);
    return hipGetLastError();
}

hipError_t launch_shard_next_limits(hipStream_t s, const uint32_t* sat, uint32_t tiles_x, uint32_t tiles_y, float margin, uint32_t radius,
                                    uint32_t* limit, const BandEdges& bands, uint2* win_next, const uint32_t* staged, const uint32_t* sat_verdict,
                                    uint32_t* host_block, uint32_t seq) {
    GSX_LAUNCH(k_shard_next_limits, dim3((tiles_x * tiles_y + 255) / 256), dim3(256), 0, s, sat, tiles_x, tiles_y, 1.0f + margin,
                       (int)std::min<uint32_t>(radius, 16u), limit, bands, feedback_stride(bands, tiles_x), win_next, staged, sat_verdict, host_block, seq);
    return hipGetLastError();
}

hipError_t launch_import_records(hipStream_t s, const void* d_recv, uint32_t n, const Records& rec) {
    if (n) GSX_LAUNCH(k_import_records, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float4*>(d_recv), n, rec);
    return hipGetLastError();
}

}  // namespace gsx
