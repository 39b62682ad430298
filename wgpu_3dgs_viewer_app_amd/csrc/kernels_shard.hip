// kernels_shard.hip — multi-GPU exchange support for gfx950: pack projected records by destination
// rank, and import received records as a rank's per-frame record set.
//
// No reference counterpart (the reference is single-device, src/main.rs:85-98).  The Gaussian array is
// sharded by splat index; the screen is cut into `world` contiguous bands of tile rows, band g = rank g.  After the projection
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
#include "gsx_internal.h"
#include "window_scan.h"

namespace gsx {

constexpr int kPackThreads = 256;
constexpr int kPackRounds = 16;
constexpr int kPackTile = kPackThreads * kPackRounds;  // 4096 records per workgroup
constexpr int kPackWaveChunk = 64 * kPackRounds;
constexpr int kMaxWorld = 64;

size_t pack_blocks(uint64_t n) { return (size_t)((n + kPackTile - 1) / kPackTile); }

// per record: destination mask (stored for the scatter pass); per workgroup and destination: record count
// list (nullable): pack only the candidates list[0 .. *d_list_n) = (key, index) pairs (the records a lazily projected
// shard admitted); element e of every per-record array below then refers to list position e.
// travellers (nullable): ballots of the elements that travel anywhere + their count per workgroup (feeds the shading of
// a lazily projected shard's repair round).
__global__ __launch_bounds__(kPackThreads) void k_pack_count(const uint32_t* __restrict__ key,
                                                              const float4* __restrict__ rec_a, uint32_t n,
                                                              uint32_t world, uint32_t rpr,
                                                              const uint2* __restrict__ window, uint32_t tiles_x,
                                                              unsigned long long* __restrict__ masks,
                                                              uint32_t* __restrict__ table, uint32_t nblocks,
                                                              const uint2* __restrict__ list, const uint32_t* __restrict__ d_list_n,
                                                              unsigned long long* __restrict__ travellers,
                                                              uint32_t* __restrict__ traveller_counts) {
    __shared__ uint32_t cnt[kMaxWorld];
    __shared__ uint32_t tcnt[kPackThreads / 64];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    if (tid < kMaxWorld) cnt[tid] = 0;
    __syncthreads();
    if (list) n = min(n, *d_list_n);
    const uint32_t base = blockIdx.x * kPackTile + wave * kPackWaveChunk;
    uint32_t tc = 0;
    for (int r = 0; r < kPackRounds; ++r) {
        uint32_t e = base + r * 64 + lane;
        uint32_t kk = kCulledKey, src = e;
        if (e < n) {
            if (list) {
                const uint2 p = list[e];
                kk = p.x;
                src = p.y;
            } else {
                kk = key[e];
            }
        }
        uint32_t rx = 0, ry = 0;
        if (kk != kCulledKey) {
            const float4 a = rec_a[src];
            rx = __float_as_uint(a.z);
            ry = __float_as_uint(a.w);
        }
        const unsigned long long m = wave_dest_mask(window, tiles_x, kk, rx, ry, rpr, world);
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
__global__ __launch_bounds__(kPackThreads) void k_pack_scatter(const unsigned long long* __restrict__ masks,
                                                                const float4* __restrict__ rec_a,
                                                                const float4* __restrict__ rec_b,
                                                                const float4* __restrict__ rec_c, uint32_t n,
                                                                uint32_t world, const uint32_t* __restrict__ table,
                                                                uint32_t nblocks, const uint32_t* __restrict__ totals,
                                                                float4* __restrict__ send, uint64_t capacity,
                                                                const uint2* __restrict__ list, const uint32_t* __restrict__ d_list_n) {
    __shared__ uint32_t run[kPackThreads / 64][kMaxWorld];  // per-wave running counts -> absolute offsets
    __shared__ uint32_t dbase[kMaxWorld];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    if (tid < kMaxWorld) {
        uint32_t b = 0;
        for (uint32_t g = 0; g < tid && g < world; ++g) b += totals[g];
        dbase[tid] = b;
    }
    for (int w = 0; w < kPackThreads / 64; ++w)
        if (tid < kMaxWorld) run[w][tid] = 0;
    __syncthreads();
    if (list) n = min(n, *d_list_n);
    const uint32_t base = blockIdx.x * kPackTile + wave * kPackWaveChunk;
    // pass 1: per-wave counts
    for (int r = 0; r < kPackRounds; ++r) {
        uint32_t e = base + r * 64 + lane;
        const unsigned long long m = e < n ? masks[e] : 0ull;
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
    for (int r = 0; r < kPackRounds; ++r) {
        uint32_t e = base + r * 64 + lane;
        const unsigned long long m = e < n ? masks[e] : 0ull;
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
                if (pos < capacity) {
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

hipError_t launch_pack_count(hipStream_t s, const Records& rec, uint32_t n, uint32_t world, uint32_t rows_per_rank,
                             const uint2* window, uint32_t tiles_x, unsigned long long* masks, uint32_t* table,
                             const uint2* list, const uint32_t* d_list_n, unsigned long long* travellers, uint32_t* traveller_counts) {
    uint32_t nb = (uint32_t)pack_blocks(n);
    if (nb)
        hipLaunchKernelGGL(k_pack_count, dim3(nb), dim3(kPackThreads), 0, s, rec.key, rec.a, n, world, rows_per_rank, window,
                           tiles_x, masks, table, nb, list, d_list_n, travellers, traveller_counts);
    return hipGetLastError();
}

hipError_t launch_pack_scatter(hipStream_t s, const Records& rec, uint32_t n, uint32_t world,
                               const unsigned long long* masks, const uint32_t* table, const uint32_t* totals, void* d_send,
                               uint64_t capacity, const uint2* list, const uint32_t* d_list_n) {
    uint32_t nb = (uint32_t)pack_blocks(n);
    if (nb)
        hipLaunchKernelGGL(k_pack_scatter, dim3(nb), dim3(kPackThreads), 0, s, masks, rec.a, rec.b, rec.c, n, world, table, nb,
                           totals, reinterpret_cast<float4*>(d_send), capacity, list, d_list_n);
    return hipGetLastError();
}

hipError_t launch_import_records(hipStream_t s, const void* d_recv, uint32_t n, const Records& rec) {
    if (n) hipLaunchKernelGGL(k_import_records, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float4*>(d_recv), n, rec);
    return hipGetLastError();
}

}  // namespace gsx
