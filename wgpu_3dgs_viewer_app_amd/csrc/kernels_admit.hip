// kernels_admit.hip — admission pass for gfx950: compacts the (depth key, Gaussian index) pairs of the records that
// take part in this frame's depth sort, in ascending index order (so the stable sort breaks depth ties by index).
//
// No reference counterpart: the reference sorts every surviving Gaussian (radix_sorter.sort, src/tab/scene.rs:865-869).
// Without windows every visible record is admitted, and the sort then moves N_vis pairs instead of N keys.  With
// per-tile depth-key windows (temporal occlusion speculation, gsx_frame.cpp) a record is admitted only if some tile of
// its rectangle still takes it; the rest stay in the record planes, untouched, for the verification round.
// Order-preserving stream compaction: per-wave ballots + per-workgroup counts, a row scan, then a scatter.
#include "gsx_internal.h"
#include "window_scan.h"

namespace gsx {

constexpr int kAdmitThreads = 256;
constexpr int kAdmitRounds = 16;
constexpr int kAdmitTile = kAdmitThreads * kAdmitRounds;  // 4096 records per workgroup
constexpr int kAdmitWaveChunk = 64 * kAdmitRounds;

size_t admit_blocks(uint64_t n) { return (size_t)((n + kAdmitTile - 1) / kAdmitTile); }

// ballots[e / 64] = admitted lanes of the 64 records e..e+63; counts[workgroup] = admitted records
__global__ __launch_bounds__(kAdmitThreads) void k_admit_count(const uint32_t* __restrict__ key, const float4* __restrict__ rec_a,
                                                                uint32_t n, const uint2* __restrict__ window, uint32_t tiles_x,
                                                                const uint32_t* __restrict__ gate, uint32_t row_words,
                                                                const WindowPyramid pyr,
                                                                const uint32_t* __restrict__ d_skip,
                                                                unsigned long long* __restrict__ ballots,
                                                                uint32_t* __restrict__ counts, const uint32_t* __restrict__ rect8) {
    __shared__ uint32_t wcnt[kAdmitThreads / 64];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const bool skip = d_skip && *d_skip == 0;  // verification round with nothing to repair: admit nothing
    if (skip) return;  // (uniform) the scan and the scatter behind look at the same word: total 0, no ballot read
    const uint32_t base = blockIdx.x * kAdmitTile + wave * kAdmitWaveChunk;
    uint32_t c = 0;
    // every key and rectangle of the wave's 16 rounds first: the loop below is a chain of dependent loads per round otherwise
    // (key -> rectangle -> four pyramid cells), and a wave that walks 16 such chains one after the other is latency, not
    // bandwidth (70 us for 80 MB at 10 M records)
    uint32_t kks[kAdmitRounds], rxs[kAdmitRounds], rys[kAdmitRounds];
#pragma unroll
    for (int r = 0; r < kAdmitRounds; ++r) {
        const uint32_t e = base + r * 64 + lane;
        kks[r] = kCulledKey;
        if (!skip && e < n) kks[r] = key[e];
    }
#pragma unroll
    for (int r = 0; r < kAdmitRounds; ++r) {
        const uint32_t e = base + r * 64 + lane;
        rxs[r] = rys[r] = 0;
        if ((window || pyr.data) && !skip && e < n) rec_rect(rec_a, rect8, e, rxs[r], rys[r]);  // (culled records too: no wait for the key)
    }
#pragma unroll
    for (int r = 0; r < kAdmitRounds; ++r) {
        const uint32_t kk = kks[r], rx = rxs[r], ry = rys[r];
        bool adm;
        if (pyr.data) adm = kk != kCulledKey && pyramid_admits(pyr, kk, rx, ry);
        else if (window) adm = wave_dest_mask(window, tiles_x, kk, rx, ry, one_band(), gate, row_words) & 1ull;
        else adm = kk != kCulledKey;
        const unsigned long long bal = __ballot(adm);
        if (lane == 0 && base + r * 64 < n) ballots[(base + r * 64) >> 6] = bal;
        c += (uint32_t)__popcll(bal);
    }
    if (lane == 0) wcnt[wave] = c;
    __syncthreads();
    if (tid == 0) counts[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// pairs[offset ..] = (key, index) of the admitted records of the workgroup.  offsets: the per-workgroup counts scanned exclusively
// (d_total == nullptr), or — d_total != nullptr — the RAW counts: every workgroup sums the counts in front of it itself (a few
// thousand L2-resident words at most) and the last one writes the total: the single-workgroup scan in between was a launch of
// its own on the critical path of every speculated frame's repair round.
template <int kRounds /* 256 x kRounds records per workgroup: the tile of the pass that wrote the ballots and the counts */>
__global__ __launch_bounds__(kAdmitThreads) void k_admit_scatter(const uint32_t* __restrict__ key, uint32_t n,
                                                                  const unsigned long long* __restrict__ ballots,
                                                                  const uint32_t* __restrict__ offsets,
                                                                  uint2* __restrict__ pairs, const uint32_t* __restrict__ d_skip,
                                                                  uint32_t* __restrict__ d_total) {
    __shared__ uint32_t wcnt[kAdmitThreads / 64];
    __shared__ uint32_t wpre[kAdmitThreads / 64];
    if (d_skip && *d_skip == 0) {  // nothing was admitted and no ballot was written (k_admit_count)
        if (d_total && blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) *d_total = 0u;
        return;
    }
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    uint32_t before = 0;
    if (d_total) {
        uint32_t x = 0;
        for (uint32_t i = tid; i < blockIdx.x; i += kAdmitThreads) x += offsets[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if (lane == 0) wpre[wave] = x;
        __syncthreads();
        for (uint32_t w = 0; w < kAdmitThreads / 64; ++w) before += wpre[w];
        if (blockIdx.x == gridDim.x - 1u && tid == 0) *d_total = before + offsets[blockIdx.x];
    } else {
        before = offsets[blockIdx.x];
    }
    const uint32_t base = blockIdx.x * (kAdmitThreads * kRounds) + wave * (64 * kRounds);
    uint32_t c = 0;
    for (int r = 0; r < kRounds; ++r)
        if (base + r * 64 < n) c += (uint32_t)__popcll(ballots[(base + r * 64) >> 6]);
    if (lane == 0) wcnt[wave] = c;
    __syncthreads();
    uint32_t o = before;
    for (uint32_t w = 0; w < wave; ++w) o += wcnt[w];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int r = 0; r < kRounds; ++r) {
        const uint32_t e0 = base + r * 64;
        if (e0 >= n) break;
        const unsigned long long bal = ballots[e0 >> 6];
        if ((bal >> lane) & 1ull) pairs[o + (uint32_t)__popcll(bal & lt)] = make_uint2(key[e0 + lane], e0 + lane);
        o += (uint32_t)__popcll(bal);
    }
}

// ---- compaction straight from the projection pass (its ballots: one word per wave, counts per 256-Gaussian workgroup) ----
// one workgroup: exclusive scan of counts[0..nblocks) in place, total -> *d_total
// (counts stay as the projection pass wrote them: a model may be sorted again without being projected again)
// visible / d_n_visible (nullable): the projection pass's per-workgroup VISIBLE counts are summed on the way (N_vis is a
// statistic only; when a sort follows the projection this saves k_sum_counts' launch)
__global__ __launch_bounds__(1024) void k_admit_scan(const uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets, uint32_t nblocks,
                                                     uint32_t* __restrict__ d_total, const uint32_t* __restrict__ visible,
                                                     uint32_t* __restrict__ d_n_visible) {
    constexpr uint32_t kTiles = 10;
    __shared__ uint32_t wtot[kTiles][16], wpre[kTiles][16];
    __shared__ uint32_t carry_s;
    __shared__ uint32_t vis_s[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    // Super-tiles of kTiles coalesced 4096-wide tiles (one uint4 per lane each; the buffers are padded to a multiple of 4):
    // ALL loads of a super-tile are issued before the first is used — one memory round trip per 40 K counts instead of one
    // per tile (the kernel is a single workgroup: ten serial round trips and thirty barriers were its 19 us).  (A contiguous
    // run per thread instead of coalesced tiles moved 10x the cache lines through the one CU and was no faster.)
    uint32_t vis = 0;
    for (uint32_t base = 0; base < nblocks; base += 4096u * kTiles) {
        uint4 v[kTiles], q[kTiles];
#pragma unroll
        for (uint32_t k = 0; k < kTiles; ++k) {
            const uint32_t i = base + 4096u * k + 4u * tid;
            v[k] = make_uint4(0, 0, 0, 0);
            q[k] = make_uint4(0, 0, 0, 0);
            if (i < nblocks) {
                v[k] = *reinterpret_cast<const uint4*>(counts + i);
                if (visible) q[k] = *reinterpret_cast<const uint4*>(visible + i);
            }
        }
        // every tile's wave-level scans first (registers + shuffles), ONE exchange of the 10 x 16 wave totals through LDS, then
        // the offsets: three barriers per super-tile instead of three per tile
        uint32_t x[kTiles], s4[kTiles];
#pragma unroll
        for (uint32_t k = 0; k < kTiles; ++k) {
            const uint32_t i = base + 4096u * k + 4u * tid;
            if (i + 1 >= nblocks) v[k].y = q[k].y = 0;
            if (i + 2 >= nblocks) v[k].z = q[k].z = 0;
            if (i + 3 >= nblocks) v[k].w = q[k].w = 0;
            vis += q[k].x + q[k].y + q[k].z + q[k].w;
            s4[k] = v[k].x + v[k].y + v[k].z + v[k].w;
            uint32_t y = s4[k];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t z = __shfl_up(y, o, 64);
                if (lane >= (uint32_t)o) y += z;
            }
            x[k] = y;
            if (lane == 63) wtot[k][wave] = y;
        }
        __syncthreads();
        if (tid < kTiles * 16u) {  // exclusive prefix of the wave totals in (tile, wave) order
            uint32_t p = 0;
            for (uint32_t j = 0; j < tid; ++j) p += wtot[j >> 4][j & 15u];
            wpre[tid >> 4][tid & 15u] = p;
        }
        __syncthreads();
        const uint32_t carry = carry_s;
#pragma unroll
        for (uint32_t k = 0; k < kTiles; ++k) {
            const uint32_t i = base + 4096u * k + 4u * tid;
            if (i < nblocks) {
                const uint32_t off = carry + wpre[k][wave] + x[k] - s4[k];
                const uint4 o4 = make_uint4(off, off + v[k].x, off + v[k].x + v[k].y, off + v[k].x + v[k].y + v[k].z);
                if (i + 3 < nblocks) {
                    *reinterpret_cast<uint4*>(offsets + i) = o4;
                } else {
                    offsets[i] = o4.x;
                    if (i + 1 < nblocks) offsets[i + 1] = o4.y;
                    if (i + 2 < nblocks) offsets[i + 2] = o4.z;
                }
            }
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wpre[kTiles - 1][15] + wtot[kTiles - 1][15];
        __syncthreads();
    }
    if (visible) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vis += __shfl_down(vis, o, 64);
        if (lane == 0) vis_s[wave] = vis;
        __syncthreads();
        if (tid == 0) {
            uint32_t t = 0;
            for (int w = 0; w < 16; ++w) t += vis_s[w];
            *d_n_visible = t;
        }
    }
    if (tid == 0) *d_total = carry_s;
}

// One lane per ballot word (64 Gaussians, one wave of k_project): a few per cent of the Gaussians are admitted, so a
// thread per Gaussian mostly dispatched workgroups that had nothing to do (39 K workgroups at 10 M: 27 us of dispatch).
// words = ceil(n / 64); word w belongs to projection workgroup w / 4, whose first output slot is offsets[w / 4].
__global__ __launch_bounds__(256) void k_admit_scatter256(const uint32_t* __restrict__ key, uint32_t words,
                                                           const unsigned long long* __restrict__ ballots,
                                                           const uint32_t* __restrict__ offsets, uint2* __restrict__ pairs) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= words) return;
    unsigned long long mine = ballots[w];
    if (!mine) return;
    uint32_t o = offsets[w >> 2];
    for (uint32_t q = w & ~3u; q < w; ++q) o += (uint32_t)__popcll(ballots[q]);
    while (mine) {  // ascending lanes: index order inside the word
        const uint32_t i = w * 64u + (uint32_t)__ffsll((long long)mine) - 1u;
        mine &= mine - 1ull;
        pairs[o++] = make_uint2(key[i], i);
    }
}

// dense admission (unspeculated frames: every visible Gaussian): one thread per Gaussian, coalesced key reads and pair
// writes — a lane walking the 64 bits of a full ballot word writes 64 scattered pairs one after the other (205 vs ~30 us)
__global__ __launch_bounds__(256) void k_admit_scatter_dense(const uint32_t* __restrict__ key, uint32_t n,
                                                              const unsigned long long* __restrict__ ballots,
                                                              const uint32_t* __restrict__ offsets, uint2* __restrict__ pairs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned long long* b = ballots + blockIdx.x * 4u;
    const unsigned long long mine = b[wave];
    if (!((mine >> lane) & 1ull)) return;
    uint32_t o = offsets[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) o += (uint32_t)__popcll(b[w]);
    pairs[o + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull))] = make_uint2(key[i], i);
}

// ---- the same compaction as ONE launch: decoupled look-back over tiles of 1024 ballot words (65536 Gaussians) ----
// Replaces k_admit_scan (one workgroup walking every per-workgroup count: 15 us at 10 M Gaussians) + k_admit_scatter256 on speculated
// frames, and counts, where the keys pass through anyway, what the depth sort wants to know about them: the 2048-bin fine histogram
// and the key range of the bucket sort (gsx_internal.h).  A tile: every thread takes four consecutive ballot words; popcounts -> block
// scan -> the tile's total is published as one 64-bit {epoch, flag, count} word (kernels_sort.hip's protocol: relaxed agent-scope
// store / loads, "the data is the flag") and wave 0 looks back over the tiles before it, 64 at a time; wave 1 does the same for the
// projection's visible counts (N_vis: a statistic).  The admitted indices go to LDS first, then every thread gathers keys for whole
// runs of them: independent loads, coalesced pair stores (a lane walking the bits of its own words waited for each key in turn).
// Tiles are taken in ticket order, so a tile only ever waits for tiles that are running.
constexpr int kCompactThreads = 256;
constexpr uint32_t kCompactList = 8192;   // admitted indices of a tile that lie in LDS at a time (a page)
typedef unsigned long long u64c;

// WPT ballot words per thread: 4 (tiles of 1024 words = 65 536 Gaussians) for large models, 1 (256 words = 16 384 Gaussians) for models of
// fewer than kCompactSmallWords words — a 1 M-Gaussian model was 16 tiles on 256 CUs, 56 us against the 17 us of a 10 M one (round 6)
template <uint32_t WPT>
__global__ __launch_bounds__(kCompactThreads) void k_admit_compact(const uint32_t* __restrict__ key, uint32_t words,
                                                                    const unsigned long long* __restrict__ ballots,
                                                                    const uint32_t* __restrict__ block_visible, uint32_t nblocks,
                                                                    uint2* __restrict__ pairs, uint32_t* __restrict__ d_total,
                                                                    uint32_t* __restrict__ d_n_visible, uint32_t* __restrict__ ticket,
                                                                    u64c* __restrict__ status, uint32_t epoch, uint32_t* __restrict__ fine,
                                                                    const uint32_t* __restrict__ hint, uint32_t* __restrict__ acc,
                                                                    const uint32_t* __restrict__ d_skip) {
    __shared__ uint32_t s_hist[kMsdFine];
    __shared__ uint32_t s_list[kCompactList];
    if (d_skip && *d_skip == 0u) {  // a repair round with nothing to repair: no ballot was written, nothing is admitted (uniform)
        if (blockIdx.x == 0 && threadIdx.x == 0) *d_total = 0u;
        return;
    }
    __shared__ uint32_t s_wsum[kCompactThreads / 64], s_vsum[kCompactThreads / 64], s_mn[kCompactThreads / 64], s_mx[kCompactThreads / 64];
    __shared__ uint32_t s_tile, s_before, s_vis_before;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t kTileWords = WPT * (uint32_t)kCompactThreads;
    const uint32_t n_tiles = (words + kTileWords - 1u) / kTileWords;   // == gridDim.x: one tile per workgroup
    if (tid == 0) s_tile = atomicAdd(&ticket[0], 1u);
    if (fine)
        for (uint32_t i = tid; i < kMsdFine; i += kCompactThreads) s_hist[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    // the thread's ballot words, its count, the tile's scan
    const uint32_t w0 = tile * kTileWords + WPT * tid;
    unsigned long long b[WPT];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < WPT; ++j) {
        b[j] = w0 + j < words ? ballots[w0 + j] : 0ull;
        mine += (uint32_t)__popcll(b[j]);
    }
    uint32_t vis = 0;
    if (block_visible && tid < kTileWords / 4u) {
        const uint32_t bi = tile * (kTileWords / 4u) + tid;   // 256-Gaussian projection workgroups: four ballot words each
        if (bi < nblocks) vis = block_visible[bi];
    }
    uint32_t x = mine, vx = vis;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= (uint32_t)o) x += y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vx += __shfl_xor(vx, o, 64);
    if (lane == 63) s_wsum[wave] = x;
    if (lane == 0) s_vsum[wave] = vx;
    __syncthreads();
    uint32_t woff = 0, total = 0, vtotal = 0;
#pragma unroll
    for (uint32_t w = 0; w < kCompactThreads / 64; ++w) {
        if (w < wave) woff += s_wsum[w];
        total += s_wsum[w];
        vtotal += s_vsum[w];
    }
    const uint32_t local = woff + x - mine;   // first slot of this thread's pairs inside the tile
    if (wave < 2) {   // wave 0: the pair counts; wave 1: the visible counts (words 2 * tile and 2 * tile + 1)
        const uint32_t excl = tile_scan_publish(status + wave, 2u, tile, epoch, lane, wave == 0 ? total : vtotal);
        if (lane == 0) (wave == 0 ? s_before : s_vis_before) = excl;
    }
    __syncthreads();
    const uint32_t before = s_before;
    if (tile == n_tiles - 1u && tid == 0) {
        *d_total = before + total;
        if (block_visible) *d_n_visible = s_vis_before + vtotal;
    }
    MsdMap map{0u, 0u, 0u};
    if (fine) map = msd_mapping(hint);
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    // the tile's admitted indices go through LDS a page of kCompactList at a time (a speculated frame's tile holds ~2000: one page;
    // a tile where most is admitted — stale windows after a camera jump, a repair round behind them — takes up to eight): every
    // lane hands in the indices whose slots fall into the page, in order, then all lanes gather keys for whole runs of them
    {
        uint32_t o = local;   // slot, inside the tile, of this lane's next index
        for (uint32_t page_lo = 0; page_lo < total; page_lo += kCompactList) {
            const uint32_t page_hi = page_lo + kCompactList, page_n = min(kCompactList, total - page_lo);
#pragma unroll
            for (uint32_t j = 0; j < WPT; ++j) {
                while (b[j] && o < page_hi) {
                    s_list[o - page_lo] = (w0 + j) * 64u + (uint32_t)__ffsll((long long)b[j]) - 1u;
                    b[j] &= b[j] - 1ull;
                    ++o;
                }
            }
            __syncthreads();
            constexpr int kU = 4;
            for (uint32_t q0 = 0; q0 < page_n; q0 += kCompactThreads * kU) {
                uint32_t idx[kU], kk[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const uint32_t q = q0 + (uint32_t)u * kCompactThreads + tid;
                    idx[u] = q < page_n ? s_list[q] : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) kk[u] = idx[u] != 0xFFFFFFFFu ? key[idx[u]] : 0u;
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    if (idx[u] != 0xFFFFFFFFu) {
                        const uint32_t q = q0 + (uint32_t)u * kCompactThreads + tid;
                        pairs[before + page_lo + q] = make_uint2(kk[u], idx[u]);
                        if (fine) {
                            atomicAdd(&s_hist[msd_fine(kk[u], map)], 1u);
                            mn = min(mn, kk[u]);
                            mx = max(mx, kk[u]);
                        }
                    }
                }
            }
            __syncthreads();   // the next page overwrites the list
        }
    }
    if (fine) {
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
        for (uint32_t i = tid; i < kMsdFine; i += kCompactThreads)
            if (s_hist[i]) atomicAdd(&fine[i], s_hist[i]);
        if (tid == 0) {
            for (uint32_t w = 1; w < kCompactThreads / 64; ++w) {
                mn = min(mn, s_mn[w]);
                mx = max(mx, s_mx[w]);
            }
            if (mn <= mx) {
                atomicMin(&acc[0], mn);
                atomicMax(&acc[1], mx);
            }
        }
    }
    // the last workgroup to finish re-arms the ticket for the next launch
    if (tid == 0) {
        const uint32_t fin = atomicAdd(&ticket[1], 1u);
        if (fin == n_tiles - 1u) {
            ticket[1] = 0;
            __hip_atomic_store(&ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

hipError_t launch_admit_compact(hipStream_t s, const uint32_t* key, uint32_t n, const unsigned long long* ballots, uint32_t* d_total, uint2* pairs,
                                const uint32_t* block_visible, uint32_t* d_n_visible, uint32_t* msd_ws, uint32_t seq, const uint32_t* d_skip, bool histogram) {
    const uint32_t words = (n + 63u) / 64u;
    if (!words) {
        if (block_visible) (void)gsx::op::MemsetAsync(d_n_visible, 0, 4, s);
        return gsx::op::MemsetAsync(d_total, 0, 4, s);
    }
    const MsdCells mc = msd_cells(msd_ws, seq);
    if (words < kCompactSmallWords)
        GSX_LAUNCH(k_admit_compact<1>, dim3((words + kCompactThreads - 1u) / kCompactThreads), dim3(kCompactThreads), 0, s, key, words, ballots, block_visible, (n + 255u) / 256u, pairs,
                   d_total, d_n_visible, msd_ws + kMsdTicket, reinterpret_cast<u64c*>(msd_ws + kMsdStatus), next_sort_epoch(), histogram ? mc.fine : nullptr, mc.hint, mc.acc, d_skip);
    else
        GSX_LAUNCH(k_admit_compact<4>, dim3((words + kCompactWordsPerTile - 1u) / kCompactWordsPerTile), dim3(kCompactThreads), 0, s, key, words, ballots, block_visible, (n + 255u) / 256u,
                   pairs, d_total, d_n_visible, msd_ws + kMsdTicket, reinterpret_cast<u64c*>(msd_ws + kMsdStatus), next_sort_epoch(), histogram ? mc.fine : nullptr, mc.hint, mc.acc, d_skip);
    return hipGetLastError();
}

hipError_t launch_admit_from_project(hipStream_t s, const uint32_t* key, uint32_t n, const unsigned long long* ballots,
                                     const uint32_t* block_counts, uint32_t* block_offsets, uint32_t* d_total, uint2* pairs,
                                     bool sparse, const uint32_t* block_visible, uint32_t* d_n_visible) {
    const uint32_t nb = (n + 255) / 256;
    if (!nb) {
        if (block_visible) (void)gsx::op::MemsetAsync(d_n_visible, 0, 4, s);
        return gsx::op::MemsetAsync(d_total, 0, 4, s);
    }
    GSX_LAUNCH(k_admit_scan, dim3(1), dim3(1024), 0, s, block_counts, block_offsets, nb, d_total, block_visible, d_n_visible);
    const uint32_t words = (n + 63) / 64;
    if (sparse)
        GSX_LAUNCH(k_admit_scatter256, dim3((words + 255) / 256), dim3(256), 0, s, key, words, ballots, block_offsets, pairs);
    else
        GSX_LAUNCH(k_admit_scatter_dense, dim3(nb), dim3(256), 0, s, key, n, ballots, block_offsets, pairs);
    return hipGetLastError();
}

hipError_t launch_admit_scatter(hipStream_t s, const uint32_t* key, uint32_t n, const unsigned long long* ballots,
                                const uint32_t* offsets, uint2* pairs, const uint32_t* d_skip, uint32_t* d_total, uint32_t rounds) {
    if (rounds == 4u) {
        const uint32_t nb = (uint32_t)((n + 1023u) / 1024u);
        if (nb) GSX_LAUNCH(k_admit_scatter<4>, dim3(nb), dim3(kAdmitThreads), 0, s, key, n, ballots, offsets, pairs, d_skip, d_total);
        return hipGetLastError();
    }
    const uint32_t nb = (uint32_t)admit_blocks(n);
    if (nb) GSX_LAUNCH(k_admit_scatter<kAdmitRounds>, dim3(nb), dim3(kAdmitThreads), 0, s, key, n, ballots, offsets, pairs, d_skip, d_total);
    return hipGetLastError();
}

hipError_t launch_admit(hipStream_t s, const Records& rec, uint32_t n, const uint2* window, uint32_t tiles_x,
                        const uint32_t* gate, uint32_t row_words, const WindowPyramid& pyramid, const uint32_t* d_skip, unsigned long long* ballots, uint32_t* counts, uint32_t* d_total, uint2* pairs,
                        uint32_t* msd_ws, uint32_t seq) {
    const uint32_t nb = (uint32_t)admit_blocks(n);
    if (!nb) return gsx::op::MemsetAsync(d_total, 0, 4, s);
    GSX_LAUNCH(k_admit_count, dim3(nb), dim3(kAdmitThreads), 0, s, rec.key, rec.a, n, window, tiles_x, gate, row_words, pyramid, d_skip, ballots, counts, rec.rect8);
    if (msd_ws)  // the compaction by look-back, counting the bucket sort's histogram on its way: launch_bucket_sort(..., hist_done = true) follows
        return launch_admit_compact(s, rec.key, n, ballots, d_total, pairs, nullptr, nullptr, msd_ws, seq, d_skip, true);
    GSX_LAUNCH(k_admit_scatter<kAdmitRounds>, dim3(nb), dim3(kAdmitThreads), 0, s, rec.key, n, ballots, counts, pairs, d_skip, d_total);  // (scans the raw counts itself)
    return hipGetLastError();
}

}  // namespace gsx
