// gsx_internal.h — shared between the host code (gsx_frame.cpp, gsx_api*.cpp) and the gfx950 kernels.
// Everything here is build-internal; the public surface is include/gsx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsx.h"
#include "gsx_launch.h"

namespace gsx {

constexpr int kTile = GSX_TILE;          // 16x16 px screen tiles
constexpr int kShPlanes4 = 11;           // 45 SH floats = 11 float4 planes + 1 float plane
constexpr uint32_t kCulledKey = 0xFFFFFFFFu;

// Per-(camera, model, gaussian-transform) constants, derived once per preprocess on the host in
// float32 with the operation order of spec/RENDER_SPEC.md §3 and passed to the kernels by value.
struct FrameConsts {
    float T[9];       // row-major 3x3: W * R_m * diag(s_m)
    float vt[3];      // W * t_m + view translation
    float P[16];      // projection, column-major
    float cam_m[3];   // camera in unscaled model space
    float s_m[3];     // model scale
    float fx, fy, limx, limy;
    float width, height;
    float size2, k, k2;
    float low_pass, cull_margin, alpha_max, alpha_min, point_radius, t_eps;
    uint32_t w_px, h_px, tiles_x, tiles_y;
    uint32_t sh_deg, no_sh0, display_mode;
    uint32_t band_lo, band_hi;  // tile rows this viewer renders (gsx_viewer_set_band; 0 .. tiles_y = everything): a Gaussian
                                // whose rectangle misses the band is culled
};

struct ModelTransform {
    float pos[3] = {0, 0, 0};
    float quat[4] = {0, 0, 0, 1};
    float scale[3] = {1, 1, 1};
};

void frame_consts_setup(const float view[16], const float proj[16], uint32_t width, uint32_t height,
                        const ModelTransform& mt, float size, uint32_t display_mode, uint32_t sh_deg, uint32_t no_sh0,
                        const gsx_spec_params& sp, FrameConsts* out);

// Multi-GPU: the screen is cut into `world` contiguous bands of tile rows, band g = rank g: rows [e[g], e[g + 1]).  e[0] = 0,
// e[world] >= tiles_y (rows past the frame hold nothing); a band may be empty.  Equal bands (e[g] = g * ceil(tiles_y / world)) or
// bands balanced by the previous frame's per-row work (gsx_shard_frame.cpp).  Passed to the kernels by value: every frame in
// flight carries the edges it was enqueued with.
constexpr uint32_t kMaxRanks = 64;
struct BandEdges {
    uint32_t world;
    uint32_t e[kMaxRanks + 1];
};
__host__ __device__ inline uint32_t band_of(const BandEdges& b, uint32_t ty) {  // owner of tile row ty (rows past the last edge: the last rank)
    uint32_t g = 0;
    while (g + 1u < b.world && ty >= b.e[g + 1u]) ++g;
    return g;
}
__host__ __device__ inline uint32_t band_rows_max(const BandEdges& b) {
    uint32_t m = 0;
    for (uint32_t g = 0; g < b.world; ++g) m = b.e[g + 1u] - b.e[g] > m ? b.e[g + 1u] - b.e[g] : m;
    return m;
}
// One rank's feedback after a round (gsx_shard_feedback), in u32 words:
//   [0] records it wanted to send to its busiest destination   [1] a slot overflowed   [2] gather root + 1 (0: every rank receives)
//   [3] list entries it binned this round   [4] / [5] the same as [0] / [1] for the repair round   [6] policy flags (how it sizes slots
//   and bands: kPolicy*)   [7] 0   [8 + d] records it wanted to send to destination d (round 0)
//   then rows x tiles_x saturation depth keys of its band (0 = open), then rows words of per-tile-row work (per tile: list entries
//   tile_work(): what the next frame's bands are balanced by); rows = its band's height.
// The pieces are gathered at a common stride (the tallest band's piece); a rank sends only what its own band needs.
constexpr uint32_t kShardExtraWords = 8 + kMaxRanks;
constexpr uint32_t kBalanceKeepPermille = 1150;  // bands whose busiest rank carries at most 1.15 x the mean work are left alone
// Band balancing: what a tile cost this frame.  Unit: one list candidate looked at and passed by, in whole-chip throughput — by the
// block compositor's instruction counts a candidate that is TAKEN (blended into 256 pixels) costs ~50 of those; an entry of the tile's
// list has cost its share of the depth sort, the binning and the list sort before any tile looks at it (cfg4: ~300 us for 0.87 M block
// entries against 223 us for the compositor's ~20 M candidates and ~5 M takers: ~440 units an entry, spread over the tiles that share
// the list); a tile that finds nothing still reads and writes its pixels.  But a band is not done when its share of the chip's
// throughput is used up: it is done when its LONGEST tile is, and a tile's walk is serial (0.95 us per 128 candidates against 0.11 us
// per taker, tools/tile_profile.py).  So the walk counts kWalkWork times: the rows whose tiles walk long lists and take little (an open
// horizon) then weigh what they cost in time, the cut goes through them, and a band cut through a block row has shorter lists
// (entries are clipped to the band).  Measured with every rank alone on the GPU (tools/rank_alone.py, world 8, open sky): slowest
// rank 1.11 ms with kWalkWork = 1 (worse than equal bands: 0.94), 0.96-0.97 with 12, 0.91 / 1.08 with 24 / 48; unspeculated
// 1.37 -> 0.96 ms (equal bands: 1.39).
constexpr uint32_t kTileWork = 256;
constexpr uint32_t kWalkWork = 12;
__host__ __device__ inline uint32_t tile_work(uint32_t walked, uint32_t taken, uint32_t list_len, uint32_t tiles_per_list) {
    return kTileWork + kWalkWork * walked + 50u * taken + (440u * list_len) / (tiles_per_list ? tiles_per_list : 1u);
}
__host__ __device__ inline uint32_t feedback_stride(const BandEdges& b, uint32_t tiles_x) { return kShardExtraWords + band_rows_max(b) * (tiles_x + 1u); }
__host__ __device__ inline uint32_t feedback_words(const BandEdges& b, uint32_t tiles_x, uint32_t g) { return kShardExtraWords + (b.e[g + 1u] - b.e[g]) * (tiles_x + 1u); }
// The verdict block in pinned host memory (u32 words; words 0..3 are the two 64-bit verdict words the host polls):
//   [4] the ranks disagree about the gather root   [5] list entries of all ranks   [6] largest of any rank   [7] work of the busiest
//   rank x world x 1000 / work of all (how evenly this frame's bands shared it)   [8 .. 8 + world] band edges for
//   the next frame   [80 + s * world + d] records rank s wanted to send to rank d in round 0
//   [76] (frames whose repair round is always enqueued) most records any rank had for ONE destination in the repair round
//   [77] ... and whether a repair slot overflowed
constexpr uint32_t kVerdictEdges = 8, kVerdictRepairMax = 76, kVerdictRepairOver = 77, kVerdictMatrix = 80, kVerdictWords = kVerdictMatrix + kMaxRanks * kMaxRanks;

// Resident pod planes of one model (SoA, every plane contiguous over the model's N Gaussians).
// The reference's 8-way pod choice (scene.rs:23-81) selects which SH / cov3d planes exist:
//   Sh Single : sh4 (11 float4 planes) + sh1          180 B     Cov3d Single: cov_a (float4) + cov_b (float2)  24 B
//   Sh Half   : sh_h (6 uint4 planes = 48 f16, 3 pad)   96 B     Cov3d Half  : cov_h (uint2 = 4 f16) + cov_h2 (u32 = 2 f16)  12 B
//   Sh Norm8  : sh_q (3 uint4 planes = 48 snorm8, 3 pad) 48 B
//   Sh None   : nothing
struct PodPlanes {
    float4* pc;      // N   : x, y, z, bitcast(rgba8)
    float4* cov_a;   // N   : xx, xy, xz, yy
    float2* cov_b;   // N   : yz, zz
    float4* sh4;     // 11*N: plane p holds SH floats 4p..4p+3 of every Gaussian (float index = 3*coeff + channel)
    float* sh1;      // N   : SH float 44
    uint4* sh_h;     // 6*N : plane p holds SH floats 8p..8p+7 as f16
    uint4* sh_q;     // 3*N : plane p holds SH floats 16p..16p+15 as snorm8
    uint4* sh_aos;   // P*N : the shade record, P = aos_stride consecutive words per Gaussian: the SH words in plane order (f32: 12, word
                     //       11 = {float 44, 0, 0, 0}; f16: 6; snorm8: 3), then pc and the covariance (aos_geo) — everything the sparse
                     //       shading pass reads about a Gaussian, in whole cache lines: 256 B (f32 SH), 192 / 128 B (f16 SH with f32 / f16
                     //       covariance), 128 B (snorm8 SH).  nullptr for Sh None.
    uint32_t aos_stride;  // words (uint4) per record: aos_layout()
    uint32_t aos_geo;     // word of the record that holds pc; the covariance follows (f32: two words {xx,xy,xz,yy} {yz,zz,0,0}; f16: one
                          // word {xx|xy, xz|yy, yz|zz, 0}).  The record holds EVERYTHING the sparse shading reads about a Gaussian.
    uint2* cov_h;    // N   : xx, xy, xz, yy as f16
    uint32_t* cov_h2;  // N : yz, zz as f16
    uint32_t* mask;  // ceil(N/32) words, bit = keep (nullptr: keep all)
    int sh_kind;     // gsx_sh_kind
    int cov_kind;    // gsx_cov3d_kind
};

// words per shade record and the word that holds pc, by pod kind (PodPlanes::sh_aos)
inline void aos_layout(int sh_kind, int cov_kind, uint32_t* stride, uint32_t* geo) {
    switch (sh_kind) {
        case GSX_SH_SINGLE: *stride = 16u; *geo = 12u; break;
        case GSX_SH_HALF: *stride = cov_kind == GSX_COV3D_SINGLE ? 12u : 8u; *geo = 6u; break;
        case GSX_SH_NORM8: *stride = 8u; *geo = 3u; break;
        default: *stride = 0u; *geo = 0u; break;
    }
}

// Projected records of one model for the current frame (valid where key != kCulledKey).
struct Records {
    uint32_t* key;   // N : f32 bit pattern of view depth, kCulledKey when culled
    float4* a;       // N : mean.x, mean.y, bitcast(x0 | x1<<16), bitcast(y0 | y1<<16)   (tile rect, max exclusive)
    float4* b;       // N : conic a, b, c, opacity
    float4* c;       // N : r, g, b, view depth
    uint32_t* rect8; // N or nullptr.  Non-null in a LAZILY projected frame on a grid of at most 255 x 255 tiles: the geometry-only
                     // projection wrote the tile rectangle of EVERY Gaussian here as four bytes x0 | y0<<8 | x1<<16 | y1<<24 (0 =
                     // culled) instead of the 16-byte `a` record — a write costs this part twice a read — and `a` (like b, c) then
                     // exists only for the Gaussians k_shade has visited.  Whoever needs the rectangle of an arbitrary visible
                     // Gaussian (the repair admission, the repair exchange's pack) reads it through rec_rect().
    uint8_t* code8;  // N or nullptr.  Slab shading: where the rectangle lies on a 16 x 16 grid of coarse screen cells — cy << 4 | cx of its
                     // first tile's cell when it reaches at most one cell further in x and in y, 0xFF ("look at the rectangle") otherwise
                     // (coarse_code()).  The depth sort carries the byte into depth order in its values' top bits, and a later depth
                     // slab refuses, without looking its rectangle up, every record none of whose (up to four) cells holds a tile that is
                     // still open (k_block_bin) — the rectangle gather by depth order is a 64-byte sector a record.
};
// coarse cell of tile t on an axis of `tiles` tiles (16 cells); the code of a tile rectangle; does a code touch a live cell?
// live: 256 bits, bit (cy * 16 + cx) — eight words
__host__ __device__ inline uint32_t coarse_cell(uint32_t t, uint32_t tiles) { return min(t * 16u / max(tiles, 1u), 15u); }
__host__ __device__ inline uint32_t coarse_code(uint32_t rx, uint32_t ry, uint32_t tiles_x, uint32_t tiles_y) {
    const uint32_t cx0 = coarse_cell(rx & 0xFFFFu, tiles_x), cx1 = coarse_cell(max(rx >> 16, 1u) - 1u, tiles_x);
    const uint32_t cy0 = coarse_cell(ry & 0xFFFFu, tiles_y), cy1 = coarse_cell(max(ry >> 16, 1u) - 1u, tiles_y);
    return (cx1 <= cx0 + 1u && cy1 <= cy0 + 1u) ? (cy0 << 4 | cx0) : 0xFFu;
}
__host__ __device__ inline bool coarse_hit(uint32_t code, const uint32_t* live) {
    if (code == 0xFFu) return true;
    const uint32_t cx = code & 15u, cy = code >> 4;
    uint32_t rows = (live[cy >> 1] >> ((cy & 1u) * 16u)) & 0xFFFFu;
    if (cy < 15u) rows |= (live[(cy + 1u) >> 1] >> (((cy + 1u) & 1u) * 16u)) & 0xFFFFu;
    return (rows & ((3u << cx) & 0xFFFFu)) != 0u;
}
// tile rectangle of record i in the packed (x0 | x1<<16, y0 | y1<<16) form
__device__ inline void rec_rect(const float4* __restrict__ rec_a, const uint32_t* __restrict__ rect8, uint32_t i, uint32_t& rx, uint32_t& ry) {
    if (rect8) {
        const uint32_t p = rect8[i];
        rx = (p & 0xFFu) | ((p >> 16 & 0xFFu) << 16);
        ry = (p >> 8 & 0xFFu) | ((p >> 24) << 16);
    } else {
        const float4 a = rec_a[i];
        rx = __float_as_uint(a.z);
        ry = __float_as_uint(a.w);
    }
}

// ---- launch wrappers (one per kernel family); all enqueue on `s` and return the launch status ----
hipError_t launch_convert(hipStream_t s, const gsx_gaussian* d_src, uint64_t n, uint64_t start, uint64_t model_n,
                          const PodPlanes& pod);
hipError_t launch_pack_pod(hipStream_t s, const float* d_pos, const uint32_t* d_color, const float* d_sh,
                           const float* d_cov, uint64_t n, uint64_t start, uint64_t model_n, const PodPlanes& pod);
hipError_t launch_unpack_pod(hipStream_t s, const PodPlanes& pod, uint64_t model_n, float* d_pos, uint32_t* d_color,
                             float* d_sh, float* d_cov);
// Max-pyramid over the per-tile window ends (launch_window_pyramid): level l holds, per cell of 2^l x 2^l tiles, the
// largest end.  A rectangle of extent <= 2^l tiles lies under at most 2x2 cells of level l, so four loads bound the
// largest window end under it from above: a conservative admission test (a hierarchical-Z test on depth keys).
struct WindowPyramid {
    const uint32_t* data;  // nullptr: no pyramid
    uint32_t off[9];       // first element of level l
    uint32_t wx[9], wy[9]; // cells per row / column of level l
    uint32_t levels;       // level (levels - 1) is a single cell
    uint32_t min_of_starts; // 0: cells hold the largest window END (admit key < it); 1: the smallest START among non-empty
                            // windows, KEY_ALL if none (admit key >= it) — the repair round's windows [hi, inf)
    // min_of_starts pyramids also carry ONE 64-bit word behind their levels: bit (cy * 8 + cx) of an 8 x 8 grid of screen cells
    // (tile >> cell_sx, tile >> cell_sy) is set iff some tile of the cell has a non-empty window.  A repair touches few tiles:
    // most records are refused by this word — shifts and an AND in registers — before any pyramid cell is loaded.
    uint32_t cells_off;     // word offset of that (8-byte aligned) word in `data`
    uint32_t cell_sx, cell_sy;
};
// What the projection kernel needs to decide admission in place (see kernels_admit.hip): ballots[i / 64] = admitted
// lanes of Gaussians i..i+63, block_counts[i / 256] = admitted per workgroup.
// A rect / brush / texture query answered inside the geometry-only projection (the reference's K1 answers its query in the
// preprocess pass too, scene.rs:856-863): one flag bit per Gaussian from the projected centre the kernel holds anyway, so a
// frame with such a query (every frame while a selection tool is dragged) stays a lazily shaded frame.
struct ProjectQuery {
    gsx_query q;
    const uint8_t* texture;  // GSX_QUERY_TEXTURE: viewport-sized, non-zero = inside
    uint32_t tex_w, tex_h;
    uint32_t* flags;         // ceil(N / 32) words; nullptr: no query in this launch
};
struct ProjectAdmission {
    WindowPyramid pyramid;        // data == nullptr: every visible Gaussian is admitted
    unsigned long long* ballots;  // ceil(N / 64) words
    uint32_t* block_counts;       // ceil(N / 256) words
    uint32_t lazy;                // 1: geometry only; launch_shade writes the conic / colour records of the admitted Gaussians
    ProjectQuery query;           // lazy launches only
};
// the records a repair round admitted; those not shaded by the (lazy) projection pass are completed
struct LateProjection {
    const uint2* pairs;                 // (key, index)
    const uint32_t* d_n;                // number of pairs, on the device
    const unsigned long long* shaded;   // nullable: ballots of the records to skip (shaded already)
    bool write_a = false;               // the projection left no `a` record (Records::rect8 mode): k_shade writes it as well
};
// d_block_visible: one count per 256-Gaussian workgroup (project_blocks(n) entries); launch_sum_counts
// reduces them into *d_n_visible.
hipError_t launch_project(hipStream_t s, const FrameConsts& f, uint32_t n, const PodPlanes& pod, const Records& rec,
                          uint32_t* d_block_visible, const ProjectAdmission& adm);
hipError_t launch_shade(hipStream_t s, const FrameConsts& f, uint32_t n, const PodPlanes& pod, const Records& rec,
                        const LateProjection& late);
hipError_t launch_sum_counts(hipStream_t s, const uint32_t* d_block_visible, uint32_t n, uint32_t* d_n_visible);
size_t project_blocks(uint64_t n);

// Stable LSD radix sort of (key,value) u32 pairs, 8-bit digits (kernels_sort.hip).  `bits` = number of
// significant key bits.  Reads (keys_src, vals_src) — never written; iota_values: the value of element i is
// i — and leaves the sorted keys / values in (keys_out, vals_out).  pairs_a / pairs_b: interleaved scratch.
struct RadixBuffers {
    const uint32_t *keys_src, *vals_src;
    const uint2* pairs_src;  // alternative input: interleaved {key,value} pairs (then keys_src/vals_src unused)
    uint32_t *keys_out, *vals_out;
    uint2 *pairs_a, *pairs_b;
    uint32_t* workspace;  // radix_workspace_words(n) u32, zero-initialised once at allocation
};
size_t radix_workspace_words(uint64_t n);
// One-time probe (synchronous, current device): may the sort take its stable ranks from returning LDS adds?  Until it has
// been called the sort uses the ballot-matching ranks.  GSX_RADIX_MATCH_RANKS=1 forces those.
bool radix_lane_ordered_adds();
void radix_set_rank_override(int mode);  // -1 none | 0 ballot matching | 1 lane-ordered LDS adds (debug: gsx_debug_set_radix_rank_mode)
// n sizes the launch; d_n (nullable) is the real element count on the device (<= n).
// ranges_out (one-digit sorts only: bits <= 8): [first, end) of every key value in the sorted output, (0, 0) for absent ones
hipError_t launch_radix_sort(hipStream_t s, const RadixBuffers& buf, uint32_t n, uint32_t* d_n, int bits, bool iota_values,
                             bool skip_culled = false, uint2* ranges_out = nullptr,  // skip_culled: keys == 0xFFFFFFFF do not exist; *d_n receives the count that do
                             const uint4* payload_in = nullptr, uint4* payload_out = nullptr,  // the last pass also writes payload_out[sorted position] = payload_in[value]
                             bool hist_done = false,  // the digit histograms are in the workspace already (k_block_bin counted them): no histogram launch
                             const uint8_t* code_in = nullptr, uint8_t* code_out = nullptr);  // the last pass also writes code_out[sorted position] = code_in[value] (Records::code8)

// ---- bucket sort of (depth key, Gaussian index) pairs: one MSD partition + one launch of in-LDS bucket sorts (kernels_sort.hip) ----
// The order the spec defines is (key, index) and the index travels with the key; what the depth sort of a speculated frame (a few
// hundred thousand admitted pairs) paid was not bytes but launches: a histogram + four digit passes, each one tile's latency plus the
// chain of tile prefixes.  Here: a 2048-bin "fine" histogram of the keys — counted by the kernel that makes the pairs
// (k_admit_compact) or by k_msd_hist — is cut into 256 coarse buckets of equal population by every workgroup of the partition pass
// alike; the partition pass is the onesweep kernel with that table as its digit (stable: ties keep index order); one more launch
// sorts every bucket on its remaining key bits, in LDS (stable LSD passes; a bucket that does not fit runs the same passes through
// global memory, one workgroup per bucket — slower, never wrong).  Which keys share a fine bin is a guess — from the minimum and
// maximum of the model's previous sort — and only the balance depends on it: keys outside the guessed range land in the first /
// last bin.
constexpr uint32_t kMsdFine = 2048;      // fine bins
constexpr uint32_t kMsdBuckets = 256;    // coarse buckets = the partition pass's digit
// workspace (u32 words, zeroed at allocation except the cells): [0, 2048) fine histogram | [2048, 2050) {min, max} of every key sorted
// so far (the running union), [2050, 2052) its snapshot: what the kernels of a sort map keys by | [2056] ticket, [2057] finished
// (k_admit_compact) | [2064, 2064 + 512) bucket ranges (uint2 x 256) | from 2576: k_admit_compact's status words, 4 per tile
constexpr uint32_t kMsdCells = 2048, kMsdTicket = 2056, kMsdRanges = 2064, kMsdStatus = 2576;
constexpr uint32_t kCompactWordsPerTile = 1024;  // ballot words (64 Gaussians each) per k_admit_compact workgroup — 256 for models of fewer than
constexpr uint32_t kCompactSmallWords = 65536;   // ... this many ballot words (4.2 M Gaussians)
inline size_t msd_workspace_words(uint64_t n_gaussians) {
    const uint64_t words = (n_gaussians + 63) / 64;
    const uint64_t tiles = words < kCompactSmallWords ? (words + 255) / 256 : (words + kCompactWordsPerTile - 1) / kCompactWordsPerTile;
    return kMsdStatus + 4 * (size_t)(tiles + 1);
}
#ifdef __HIPCC__
// Decoupled look-back of a single-pass scan, one wave per call: status[k * stride] is tile k's 64-bit word {epoch << 34 | flag << 32 |
// count}, flag 1 = the tile's own count, 2 = its inclusive prefix; written with one relaxed agent-scope store, read with relaxed
// agent-scope loads ("the data is the flag": kernels_sort.hip).  Returns the exclusive prefix of `tile`: 64 predecessors per round
// trip, nearest first; tiles must have been taken in ticket order (a tile only ever waits for tiles that are running).
__device__ inline uint32_t tile_lookback(const unsigned long long* __restrict__ status, uint32_t stride, uint32_t tile, uint32_t epoch, uint32_t lane) {
    uint32_t excl = 0;
    int32_t k = (int32_t)tile - 1;
    while (k >= 0) {
        const int32_t kk = k - (int32_t)lane;
        unsigned long long w = 0;
        if (kk >= 0) w = __hip_atomic_load(status + (size_t)kk * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t flag = (uint32_t)(w >> 32) & 3u;
        const bool there = kk >= 0 && (uint32_t)(w >> 34) == epoch && flag != 0;
        const unsigned long long missing = __ballot(kk >= 0 && !there);
        const unsigned long long prefix = __ballot(there && flag == 2u);
        const uint32_t first_missing = missing ? (uint32_t)__ffsll((long long)missing) - 1u : 64u;
        const uint32_t first_prefix = prefix ? (uint32_t)__ffsll((long long)prefix) - 1u : 64u;
        const uint32_t take = first_prefix < first_missing ? first_prefix + 1u : first_missing;   // lanes [0, take) are consumed
        uint32_t x = (lane < take && kk >= 0) ? (uint32_t)w : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        excl += x;
        if (first_prefix < first_missing) break;
        k -= (int32_t)take;
        if (take == 0) __builtin_amdgcn_s_sleep(1);
    }
    return excl;
}
// publish + look back + publish the inclusive prefix: the calling wave's lane 0 owns the tile's word
__device__ inline uint32_t tile_scan_publish(unsigned long long* __restrict__ status, uint32_t stride, uint32_t tile, uint32_t epoch, uint32_t lane, uint32_t total) {
    const unsigned long long tag = (unsigned long long)epoch << 34;
    unsigned long long* my = status + (size_t)tile * stride;
    if (lane == 0) __hip_atomic_store(my, tag | ((tile == 0 ? 2ull : 1ull) << 32) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    if (tile > 0) {
        excl = tile_lookback(status, stride, tile, epoch, lane);
        if (lane == 0) __hip_atomic_store(my, tag | (2ull << 32) | (unsigned long long)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return excl;
}
#endif

// Fine bin of a key.  The keys of a sort are GUESSED to lie in [lo, hi] — the union of the key ranges of the model's earlier sorts (or,
// before any, depths 2^-7 ... 2^14) — and that stretch gets bins 128 ... 1919 at 2^fs keys a bin; whatever lies outside still gets bins
// of its own, a sixteenth of an octave wide, 128 on either side (beyond eight octaves: the outermost bin).  The guess decides the
// balance only: a frame whose keys left the guessed range (a camera jump, a repair round that admits the far half of the scene) keeps
// thousands of keys a bin instead of all of them in one.
struct MsdMap {
    uint32_t lo, hi, fs;
};
__device__ inline MsdMap msd_mapping(const uint32_t* __restrict__ hint) {
    uint32_t mn = 0x3C000000u, mx = 0x46800000u;
    const uint32_t a = hint[0], b = hint[1];
    if (a <= b) {
        mn = a;
        mx = b;
    }
    const uint32_t span = mx - mn;
    const int bits = 32 - __clz((int)(span | 1u));
    uint32_t fs = bits > 11 ? (uint32_t)(bits - 11) : 0u;
    if ((span >> fs) >= 1792u) fs += 1u;
    return MsdMap{mn, mx, fs};
}
__device__ inline uint32_t msd_fine(uint32_t key, const MsdMap& m) {
    if (key < m.lo) return 127u - min((m.lo >> 19) - (key >> 19), 127u);
    if (key > m.hi) return 1920u + min((key >> 19) - (m.hi >> 19), 127u);
    return 128u + ((key - m.lo) >> m.fs);
}
struct MsdCells {   // pointers into workspace `ws`
    uint32_t* fine;
    const uint32_t* hint;   // {min, max}: the snapshot every kernel of a sort maps keys by
    uint32_t* acc;          // {min, max}: the running union, widened by the kernel that counts the histogram; the bucket kernel copies it to `hint`
    uint2* ranges;
};
inline MsdCells msd_cells(uint32_t* ws, uint32_t /*seq*/) {
    return MsdCells{ws, ws + kMsdCells + 2, ws + kMsdCells, reinterpret_cast<uint2*>(ws + kMsdRanges)};
}
// the workspace's initial contents (host): zeros, both cells at {0xFFFFFFFF, 0} = "no keys seen"
hipError_t msd_workspace_init(hipStream_t s, uint32_t* ws, size_t words);
// n sizes the launches; *d_n is the element count.  hist_done: the fine histogram (and the cells) of sort `seq` were made by the
// kernel that wrote buf.pairs_src (launch_admit_compact with the same ws and seq); otherwise k_msd_hist runs first.
// buf: pairs_src, or keys_src (+ vals_src / iota_values); pairs_a, pairs_b scratch; keys_out / vals_out the sorted result;
// workspace = the radix workspace (ticket + status words of the partition pass).
hipError_t launch_bucket_sort(hipStream_t s, const RadixBuffers& buf, uint32_t n, uint32_t* d_n, bool iota_values, uint32_t* msd_ws, uint32_t seq,
                              bool hist_done);
// k_admit_compact (kernels_admit.hip): the admitted (key, index) pairs of a projection pass in index order from its ballots — one
// launch (decoupled look-back over 65536-Gaussian tiles) instead of k_admit_scan + k_admit_scatter256 — and, msd_ws != nullptr, the fine
// histogram + key range of sort `seq`.  *d_total = pairs; block_visible (nullable): the projection's per-workgroup visible counts are
// summed into *d_n_visible on the way.
hipError_t launch_admit_compact(hipStream_t s, const uint32_t* key, uint32_t n, const unsigned long long* ballots, uint32_t* d_total, uint2* pairs,
                                const uint32_t* block_visible, uint32_t* d_n_visible, uint32_t* msd_ws, uint32_t seq,
                                const uint32_t* d_skip = nullptr /* points at 0: nothing is admitted, no ballot is read */,
                                bool histogram = true /* false: the compaction alone (msd_ws still holds its ticket and status words) */);
uint32_t next_sort_epoch();  // status-word epochs of every look-back kernel of the process (kernels_sort.hip)
void block_bin_set_big_rect(uint32_t blocks);    // tests / tuning: rectangles of more blocks than this are walked by the whole wave in k_block_bin (0: the default)
void block_bin_set_big_slab(uint32_t records);  // tests: slabs of this many records and more take eight records per lane in k_block_bin (0: the default)
void bucket_sort_set_cap(uint32_t cap);  // tests: buckets above `cap` pairs take the global-memory path (0: the LDS capacity)

// Device-resident per-model frame statistics; the host mirrors them lazily (no sync inside a frame).
struct SlabStats {
    uint32_t n_visible;        // N_vis (projection pass / import)
    uint32_t n_sorted;         // records in the depth order of this frame (admission pass / import); <= n_visible
    uint32_t n_candidates;     // lazily projected shard: records the admission let through (gsx_shard_set_windows)
    uint32_t overflow_events;  // slabs, over the model's lifetime, whose tile entries did not fit the pair buffers (never reset:
                               // the host compares with the count it has seen and grows the buffers for the frames to come)
    uint32_t max_needed_ever;  // largest slab D ever seen
    uint32_t slot_max[2];      // device-resident exchange: most records this rank had for ONE destination in the last round 0 / 1
    uint32_t slot_over[2];     // ... and whether that exceeded the slot (the verdict tells every rank; round 0 is then redone)
    uint32_t shard_need;       // tiles of the whole frame that needed the repair round (gsx_shard_verify)
    uint32_t shard_ticket;     // k_shard_verify: blocks done (the last one posts the verdict); directly behind shard_need: zeroed together
    uint32_t slot_want[64];    // records this rank wanted to send to each destination in the last round 0 (k_pack_headers): the
                               // gathered count matrix sizes the next frame's slots per (source, destination) pair
    uint32_t walk_max;         // block compositor: the longest walk of any tile in the model's frame BEFORE the one that wrote it, in chunks
                               // of 128 list candidates (tile_order_job) — the host picks the block size by it (gsx_frame.cpp)
    // ---- from here on: zeroed at the start of every frame ----
    uint32_t n_entries;        // D of the slab being processed: what was binned into the pair buffers (<= their capacity)
    uint32_t n_entries_total;  // sum of slab D over the frame (including entries the spill compositor handled without pairs)
    uint32_t overflow;         // a slab of THIS frame needed more pair capacity than allocated (its tail went to the spill compositor)
    uint32_t max_needed;       // largest slab D of this frame
    uint32_t slabs_used;       // number of slabs that still found a live tile (progressive mode)
    uint32_t n_sorted2;        // speculation: records admitted in the repair round
    uint32_t spec_need;        // speculation: tiles that needed the repair round
    uint32_t verify_ticket;    // k_spec_verify: blocks that have added their share of spec_need (the last one posts the verdict)
    uint32_t live_cells[8][8]; // slab shading: per depth slab (index min(slab, 7)), 256 bits: the coarse cells that hold a tile still open (k_block_table; coarse_hit)
    uint32_t n_shaded_total;   // slab shading: records shaded over the frame's slabs (the host stops slab shading on a scene where most visible records are: translucent)
    uint32_t n_slab_shade;     // slab shading (gsx_render_options): records of the current slab some block takes = the slab's shading list (k_block_bin)
    uint32_t slab_cut;         // depth-order position up to which the current slab was binned into pairs: the whole slab unless
                               // its entries overflowed the pair buffers; k_composite_spill composites [slab_cut, slab end)
};

// Two runs of words some kernel of the frame zeroes on its way (the frame's saturation state + the first model's counters:
// folded into the first slab's block-table kernel instead of a launch of their own).
struct ZeroJob {
    uint32_t* a = nullptr;
    uint32_t na = 0;
    uint32_t* b = nullptr;
    uint32_t nb = 0;
    // ... and the block compositor's dispatch order, as one more workgroup of the slab's k_block_counts (tile_order_job):
    // order_buf = {threshold, tile_cost[order_tiles], tile_order[order_tiles]}
    uint32_t* order_buf = nullptr;
    uint32_t order_tiles = 0;
    // ... and a run of words copied on the same way (a layered model's speculated frame keeps the saturation bitmap the models in front
    // of it left: k_spec_next must not read their tiles as this model's)
    const uint32_t* copy_src = nullptr;
    uint32_t* copy_dst = nullptr;
    uint32_t n_copy = 0;
};
// Frames of more tiles keep index order: at 3840 x 2160 (32 400 tiles: ten dispatch rounds, the tail is a smaller share of the launch) the
// order gains little and the job's one workgroup takes ~20 us per model — cfg5 (four models) measured 6 % slower with it.
// (tile_order_job keeps 3 words of LDS per 64 tiles: within k_block_counts' 4096)
constexpr uint32_t kTileOrderMax = 16384;
constexpr uint32_t tile_order_lds_words(uint32_t n_tiles) { return 3u * ((n_tiles + 63u) / 64u) + 32u; }

#ifdef __HIPCC__
// The block compositor's dispatch order (k_composite_blocks): the tiles whose cost in the model's frame before (tile_cost, summed over
// that frame's compositor launches; zeroed here) was above that frame's threshold come first, the others after, each class in index
// order (neighbours share block lists and records: they still run together).  threshold <- this frame's average cost.
// One workgroup of THREADS lanes; a wave takes 64 consecutive tiles at a time: one coalesced load, one ballot, no second look at memory.
template <uint32_t THREADS>
__device__ __forceinline__ void tile_order_job(uint32_t* __restrict__ order_buf, const uint32_t n_tiles, uint32_t* __restrict__ lds /* tile_order_lds_words(n_tiles), 8-byte aligned */,
                                               uint32_t* __restrict__ walk_max_out /* <- the longest walk of any tile in that frame, in chunks (SlabStats::walk_max) */) {
    constexpr uint32_t kWaves = THREADS / 64u, kBatch = 16u;
    const uint32_t groups = (n_tiles + 63u) / 64u;
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(lds);   // bit l of group g: tile 64 g + l is expensive
    uint32_t* s_before = lds + 2u * groups;                                     // expensive tiles in the groups before g
    uint32_t *s_sum = s_before + groups, *s_cnt = s_sum + kWaves;
    uint32_t* __restrict__ tile_cost = order_buf + 1;
    uint32_t* __restrict__ tile_order = tile_cost + n_tiles;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t thr = order_buf[0];
    uint32_t sum = 0, walk = 0;
    for (uint32_t gb = wave; gb < groups; gb += kBatch * kWaves) {  // kBatch loads in flight per lane (one workgroup: latency is all there is)
        uint32_t c[kBatch];
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t tile = (gb + k * kWaves) * 64u + lane;
            c[k] = tile < n_tiles ? tile_cost[tile] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {  // {chunks walked << 16 | takers} (summed over the frame's launches) -> the cost the order is made by
            walk = max(walk, c[k] >> 16);
            c[k] = (9u * (c[k] >> 16) + (c[k] & 0xFFFFu)) >> 2;
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t g = gb + k * kWaves, tile = g * 64u + lane;
            if (tile < n_tiles) tile_cost[tile] = 0u;
            sum += c[k];
            const unsigned long long m = __ballot(c[k] > thr);
            if (lane == 0u && g < groups) s_mask[g] = m;
        }
    }
#pragma unroll
    for (uint32_t d = 32u; d; d >>= 1) {
        sum += __shfl_xor(sum, d);
        walk = max(walk, (uint32_t)__shfl_xor(walk, d));
    }
    if (lane == 0u) {
        s_sum[wave] = sum;
        s_cnt[wave] = walk;  // (s_cnt is written again only behind the next barrier)
    }
    __syncthreads();
    if (tid == 0u && walk_max_out) {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t k = 0; k < kWaves; ++k) w = max(w, s_cnt[k]);
        *walk_max_out = w;
    }
    __syncthreads();
    // exclusive scan of the groups' counts: every lane a contiguous run of groups, then the runs
    const uint32_t per = (groups + THREADS - 1u) / THREADS, g0 = min(tid * per, groups), g1 = min(g0 + per, groups);
    uint32_t mine = 0;
    for (uint32_t g = g0; g < g1; ++g) mine += (uint32_t)__popcll(s_mask[g]);
    uint32_t incl = mine;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63u) s_cnt[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; ++w) {
        run += w < wave ? s_cnt[w] : 0u;
        total += s_cnt[w];
    }
    for (uint32_t g = g0; g < g1; ++g) {
        s_before[g] = run;
        run += (uint32_t)__popcll(s_mask[g]);
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t g = wave; g < groups; g += kWaves) {
        const uint32_t tile = g * 64u + lane;
        const unsigned long long m = s_mask[g];
        const uint32_t before = s_before[g];
        if (tile < n_tiles) {
            const bool exp_ = (m >> lane) & 1ull;
            // (a cheap tile: after all the expensive ones, behind the cheap tiles of the groups before and of this group's lower lanes)
            const uint32_t at = exp_ ? before + (uint32_t)__popcll(m & lt) : total + (g * 64u - before) + (uint32_t)__popcll(~m & lt);
            tile_order[at] = tile;
        }
    }
    if (tid == 0u) {
        uint32_t all = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) all += s_sum[w];
        order_buf[0] = all / n_tiles;
    }
}
#endif

// Block lists (kernels_bin.hip): the tile rows [row_lo, row_hi) a viewer composites — the whole screen, or the band of a multi-GPU
// rank — in at most 256 blocks of 2^bsx x 2^bsy tiles.  Block rows count from row_lo: a rank that owns an eighth of the rows gets
// blocks an eighth the size for the same one-pass block sort, and its tiles walk lists that much shorter (round 5: until then the
// grid covered the screen whoever owned it, and a world-8 rank's 1020 tiles shared 32 lists).
struct BlockGrid {
    uint32_t bsx, bsy;       // log2 of the block size in tiles
    uint32_t blocks_x, blocks_y;
};
inline BlockGrid block_grid(uint32_t bsx, uint32_t bsy, uint32_t tiles_x, uint32_t row_lo, uint32_t row_hi) {
    const uint32_t rows = row_hi > row_lo ? row_hi - row_lo : 1u;
    return BlockGrid{bsx, bsy, (tiles_x + (1u << bsx) - 1u) >> bsx, (rows + (1u << bsy) - 1u) >> bsy};
}
#ifdef __HIPCC__
// One WAVE computes table[b] = {min window start, max window end (of the non-empty windows of the block's live tiles), live} and
// zeroes ranges[b]; a lane per tile.  done (nullable): saturated tiles are not live; win (nullable): no windows = takes every key.
__device__ inline void wave_block_table_entry(const BlockGrid& g, uint32_t b, uint32_t tiles_x, uint32_t tiles_y, uint32_t row_lo, uint32_t row_hi,
                                              const uint32_t* __restrict__ done, uint32_t row_words, const uint2* __restrict__ win,
                                              uint4* __restrict__ table, uint2* __restrict__ ranges, uint32_t* __restrict__ live_cells = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t bx = b % g.blocks_x, by = b / g.blocks_x;
    const uint32_t x0 = bx << g.bsx, x1 = min(x0 + (1u << g.bsx), tiles_x);
    const uint32_t y0 = row_lo + (by << g.bsy), y1 = min(min(row_lo + ((by + 1u) << g.bsy), tiles_y), row_hi);
    uint32_t lo = 0xFFFFFFFFu, hi = 0u, live = 0u;
    const uint32_t w = x1 - x0, total = y1 > y0 ? w * (y1 - y0) : 0u;
    for (uint32_t k0 = 0; k0 < total; k0 += 64) {   // (uniform trip count: the ballots below want every lane)
        const uint32_t k = k0 + lane;
        bool tile_live = false;
        uint32_t tx = 0, ty = 0;
        if (k < total) {
            tx = x0 + k % w;
            ty = y0 + k / w;
            tile_live = !(done && ((done[ty * row_words + (tx >> 5)] >> (tx & 31u)) & 1u));
            if (tile_live && win) {
                const uint2 ww = win[ty * tiles_x + tx];
                if (ww.x >= ww.y) {
                    tile_live = false;
                } else {
                    lo = min(lo, ww.x);
                    hi = max(hi, ww.y);
                }
            }
        }
        if (tile_live) live = 1u;
        if (live_cells) {   // the coarse cells of the live tiles: one atomic per distinct cell of the wave (a block reaches over a few at most)
            const uint32_t cell = tile_live ? coarse_cell(ty, tiles_y) * 16u + coarse_cell(tx, tiles_x) : 0xFFFFFFFFu;
            unsigned long long pending = __ballot(cell != 0xFFFFFFFFu);
            while (pending) {
                const int src = __ffsll((long long)pending) - 1;
                const uint32_t c = (uint32_t)__shfl((int)cell, src, 64);
                if ((int)lane == src) atomicOr(&live_cells[c >> 5], 1u << (c & 31u));
                pending &= ~__ballot(cell == c);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor(lo, o, 64));
        hi = max(hi, (uint32_t)__shfl_xor(hi, o, 64));
        live |= (uint32_t)__shfl_xor(live, o, 64);
    }
    if (!win) {
        lo = 0u;
        hi = 0xFFFFFFFFu;
    }
    if (lane == 0) {
        table[b] = make_uint4(lo, hi, live, 0u);
        ranges[b] = make_uint2(0u, 0u);
    }
}
#endif

// Tile binning.
// Splats [j0, min(j1, *d_n_vis)) of the depth order; cnt / block_sums are indexed relative to j0.
// [row_lo, row_hi): the band of tile rows this rank bins (0, tiles_y on one GPU).
// done (nullable): bitmap of saturated tiles (row_words u32 per tile row) that receive no more entries.
// srect[j - j0]: the splat's packed tile rectangle, gathered once by the count pass and re-read by emit.
hipError_t launch_tile_counts(hipStream_t s, uint32_t j0, uint32_t j1, const uint32_t* d_n_vis, const uint32_t* sorted_idx,
                              const Records& rec, uint2* srect, uint32_t* cnt, uint32_t* block_sums, SlabStats* stats,
                              uint32_t capacity, uint32_t row_lo, uint32_t row_hi, const uint32_t* done, uint32_t row_words,
                              const uint32_t* d_done_count, uint32_t owned_tiles, uint32_t slab_index,
                              const uint2* window, const uint32_t* sorted_keys, uint32_t tiles_x,
                              const WindowPyramid* min_ends = nullptr /* of `window`, when every window starts at 0 */);
hipError_t launch_tile_emit(hipStream_t s, uint32_t j0, uint32_t j1, const uint32_t* sorted_idx, const uint2* srect,
                            const uint32_t* cnt, const uint32_t* block_sums, uint32_t tiles_x, uint2* tpairs,
                            uint32_t row_lo, uint32_t row_hi, const uint32_t* done, uint32_t row_words,
                            const uint32_t* d_n_vis, const uint32_t* d_entries, uint32_t capacity,
                            const uint2* window, const uint32_t* sorted_keys, const uint32_t* d_cut /* SlabStats::slab_cut */);
// ranges_clean: the whole table (table_tiles entries = its allocation) is already all-zero (the previous composite
// cleared what it used); otherwise it is zeroed here, all of it
hipError_t launch_tile_ranges(hipStream_t s, uint32_t capacity, const uint32_t* d_n, const uint32_t* tkey_sorted,
                              uint32_t table_tiles, uint2* ranges, bool ranges_clean);
size_t scan_blocks(uint64_t n);
// Block lists (progressive frames): bin by blocks of 2^bsx x 2^bsy tiles (<= 256 blocks), one 8-bit sort pass, and
// k_composite_blocks applies the exact per-tile decision.  brec: uint4 per slab record; table: 1024 uint4; ranges: the block
// range table (zeroed here, filled by launch_tile_ranges).
// the same as ONE launch behind the block table (k_block_bin: counts, slots by decoupled look-back, entries, and the block sort's digit
// histograms into sort_ghist = the block sort's workspace): launch_radix_sort(..., hist_done = true) follows.  bin_ws: bin_workspace_words(records) u32, zeroed once.
size_t bin_workspace_words(uint64_t n_records);
hipError_t launch_block_bin_fused(hipStream_t s, uint32_t j0, uint32_t j1, const uint32_t* d_n_vis, const uint32_t* sorted_idx,
                                  const Records& rec, const uint32_t* sorted_keys, uint4* brec, SlabStats* stats, uint32_t capacity,
                                  uint32_t row_lo, uint32_t row_hi, const uint32_t* done, uint32_t row_words, const uint32_t* d_done_count,
                                  uint32_t owned_tiles, uint32_t slab_index, const uint2* window, uint32_t tiles_x, uint32_t tiles_y,
                                  uint32_t bsx, uint32_t bsy, uint4* table, uint2* pairs, uint2* ranges, const ZeroJob& zero, bool table_ready,
                                  uint32_t* bin_ws, uint32_t* sort_ghist, int block_bits,
                                  uint2* shade_pairs = nullptr /* slab shading: (key, index) of the slab's records some block takes -> stats->n_slab_shade; the
                                                                  rectangles are then read from rec.rect8 (the records are not shaded yet) */,
                                  const uint8_t* sorted_code = nullptr /* Records::code8 in depth order: later slabs skip records whose coarse cells hold no open tile */);
hipError_t launch_block_bin(hipStream_t s, uint32_t j0, uint32_t j1, const uint32_t* d_n_vis, const uint32_t* sorted_idx,
                            const Records& rec, const uint32_t* sorted_keys, uint4* brec, uint32_t* cnt, uint32_t* block_sums,
                            SlabStats* stats, uint32_t capacity, uint32_t row_lo, uint32_t row_hi, const uint32_t* done,
                            uint32_t row_words, const uint32_t* d_done_count, uint32_t owned_tiles, uint32_t slab_index,
                            const uint2* window, uint32_t tiles_x, uint32_t tiles_y, uint32_t bsx, uint32_t bsy, uint4* table,
                            uint2* pairs, uint2* ranges, const ZeroJob& zero = ZeroJob{} /* words the table kernel zeroes on the way */,
                            bool table_ready = false /* table and ranges are in place already (launch_spec_verify built them) */);
hipError_t launch_composite_blocks(hipStream_t s, const FrameConsts& f, const uint2* ranges, const uint32_t* list /* nullptr: brec is in list order */, const uint4* brec,
                                   const Records& rec, float4* fb, bool carry, uint32_t* done, uint32_t row_words,
                                   uint32_t* d_done_count, uint32_t* tile_sat, const uint2* window, uint32_t row_lo,
                                   uint32_t row_hi, uint32_t bsx, uint32_t bsy, uint32_t* row_work /* as launch_composite */,
                                   const SlabStats* stats, uint32_t j1, const uint32_t* d_n, const uint32_t* sorted_idx,
                                   const uint32_t* sorted_keys /* the slab's tail behind stats->slab_cut is composited pair-free by the same launch */,
                                   uint4* tile_prof = nullptr /* development: per tile {start, duration (10 ns ticks), chunks walked | list chunks << 16, takers} */,
                                   const uint32_t* tile_order = nullptr /* the tile workgroup i composites (nullptr: tile i) */,
                                   uint32_t* tile_cost = nullptr /* += what each tile cost: tile_order_job's input for the model's next frame */, const uint32_t* rect8 = nullptr /* slab shading: the pair-free tail reads rectangles from the packed plane */);

// Selection / edits / queries (kernels_edit.hip).
hipError_t launch_edit_prepare(hipStream_t s, uint32_t n, const uint32_t* selection, uint32_t* edited, float4* edit_a,
                               float4* edit_b, const gsx_gaussian_edit& sel_edit, const uint32_t* mask, uint32_t* keep);
hipError_t launch_edit_apply(hipStream_t s, uint32_t n, const Records& rec, const uint32_t* selection, const uint32_t* edited,
                             const float4* edit_a, const float4* edit_b, const float highlight[4]);
hipError_t launch_edit_apply_list(hipStream_t s, uint32_t n, const Records& rec, const uint2* pairs, const uint32_t* d_n,
                                  const unsigned long long* skip, const uint32_t* selection, const uint32_t* edited, const float4* edit_a,
                                  const float4* edit_b, const float highlight[4]);
hipError_t launch_query(hipStream_t s, uint32_t n, const Records& rec, const gsx_query& q, const uint8_t* texture, uint32_t tex_w,
                        uint32_t tex_h, const FrameConsts& f, uint32_t* flags, gsx_query_hit* hits, uint32_t* hit_count,
                        uint32_t hit_capacity);
hipError_t launch_selection_op(hipStream_t s, uint32_t n_words, uint32_t op, const uint32_t* flags, uint32_t* selection);

WindowPyramid window_pyramid_layout(uint32_t tiles_x, uint32_t tiles_y, const uint32_t* data);  // total words: off[levels]... see .hip
size_t window_pyramid_words(uint32_t tiles_x, uint32_t tiles_y);
hipError_t launch_window_pyramid(hipStream_t s, const uint2* window, uint32_t tiles_x, uint32_t tiles_y, uint32_t* data,
                                 bool min_of_starts = false, const uint32_t* d_skip = nullptr, uint32_t* min_ends = nullptr);

// exclusive scan of the projection pass's per-workgroup counts (total -> *d_total), then the compaction of the admitted
// (key, index) pairs in ascending index order
hipError_t launch_admit_from_project(hipStream_t s, const uint32_t* key, uint32_t n, const unsigned long long* ballots,
                                     const uint32_t* block_counts, uint32_t* block_offsets, uint32_t* d_total, uint2* pairs,
                                     bool sparse /* a few per cent admitted (windows): one lane per ballot word */,
                                     const uint32_t* block_visible = nullptr /* + sum these into *d_n_visible (saves k_sum_counts) */,
                                     uint32_t* d_n_visible = nullptr);

// Admission pass (kernels_admit.hip): compacts the (key, index) pairs of the records the depth sort takes, ascending
// index; window == nullptr admits every visible record.  *d_total = number of pairs.  d_skip (nullable): when it
// points at 0 nothing is admitted (verification round with nothing to repair).  gate (nullable): tile bitmap; records
// whose rectangle holds no gated tile are refused before the windows are looked at.  pyramid (data != nullptr): the
// windows are [0, hi) and admission is CONSERVATIVE — every record some tile admits is admitted, plus a few more; the
// binning applies the exact per-tile windows, so the surplus only rides through the depth sort.
size_t admit_blocks(uint64_t n);
// compaction from ballots over workgroups of 256 x rounds records (rounds = 16, or 4: the tile of the pass that counted — pack_rounds)
// (offsets = exclusively scanned per-workgroup counts)
// d_total != nullptr: `offsets` holds the RAW per-workgroup counts — every workgroup sums the ones in front of it itself and the last
// writes the total (no scan launch in between)
hipError_t launch_admit_scatter(hipStream_t s, const uint32_t* key, uint32_t n, const unsigned long long* ballots,
                                const uint32_t* offsets, uint2* pairs, const uint32_t* d_skip = nullptr, uint32_t* d_total = nullptr, uint32_t rounds = 16);
// (rec.rect8 != nullptr: rectangles are read from the packed plane)
hipError_t launch_admit(hipStream_t s, const Records& rec, uint32_t n, const uint2* window, uint32_t tiles_x,
                        const uint32_t* gate, uint32_t row_words, const WindowPyramid& pyramid, const uint32_t* d_skip, unsigned long long* ballots, uint32_t* counts, uint32_t* d_total, uint2* pairs,
                        uint32_t* msd_ws = nullptr, uint32_t seq = 0 /* msd_ws: compaction by k_admit_compact, which also counts the bucket sort's histogram of sort `seq` */);

// Temporal occlusion speculation (kernels_spec.hip): verification of this frame's windows, windows of the next frame.
// pyr2_data (nullable): + the min-pyramid of the repair windows' starts; grid / table / ranges (nullable): + the repair slab's block
// table, its ranges zeroed — both only when some tile needs the repair round (k_spec_verify_fused)
hipError_t launch_spec_verify(hipStream_t s, const uint2* win1, const uint32_t* done, uint32_t row_words, uint32_t tiles_x,
                              uint32_t tiles_y, uint2* win2, uint32_t* need_bits, uint32_t* d_need, uint32_t band_lo, uint32_t band_hi,
                              unsigned long long* host_verdict /* pinned host word or null */, uint32_t seq, uint32_t* pyr2_data,
                              const BlockGrid* grid, uint4* table, uint2* ranges);
hipError_t launch_zero_words(hipStream_t s, uint32_t* a, uint32_t na, uint32_t* b, uint32_t nb);
hipError_t launch_validate_tiles(hipStream_t s, const uint2* ranges, uint32_t n_tiles, const uint32_t* list, const uint32_t* d_entries,
                                 uint32_t capacity, uint32_t n_records, uint32_t* report);
hipError_t launch_spec_next(hipStream_t s, const uint32_t* tile_sat, const uint32_t* done, const uint32_t* done_before,
                            uint32_t row_words, uint32_t tiles_x, uint32_t tiles_y, float margin, uint32_t radius, uint2* win_next,
                            uint32_t band_lo, uint32_t band_hi);

// Multi-GPU exchange support (kernels_shard.hip).
// d_n (nullable) / tile: only the first ceil(*d_n / tile) columns of every row hold anything (a candidate list shorter than the grid)
hipError_t launch_spin(hipStream_t s, uint32_t microseconds);  // one wave that does nothing for that long (stream / queue probe)
hipError_t launch_rowscan(hipStream_t s, uint32_t* table, uint32_t nrows, uint32_t nblocks, uint32_t* totals, const uint32_t* d_n = nullptr,
                          uint32_t tile = 1, const uint32_t* d_skip = nullptr /* points at 0: the totals are 0, nothing is read */);
size_t pack_blocks(uint64_t n, uint32_t rounds);  // workgroups of a pack pass whose tiles hold 256 x rounds records
uint32_t pack_rounds(bool candidate_list, uint64_t n);  // 2 (a candidate list), 4 (a whole shard of <= 2 M records), 16
// bands: rank g owns the tile rows [bands.e[g], bands.e[g + 1]).  A record travels to g if its rectangle touches g's band and some
// tile of it there has the record's key inside its window.
// window: uint2 [lo, hi) depth-key window per tile (tiles_y * tiles_x, row-major) or nullptr = every tile takes everything.
// list / d_list_n (nullable): pack only these (key, index) candidates; the per-element arrays are then indexed by list position.
// travellers / traveller_counts (nullable): ballots + per-workgroup counts of the elements that travel anywhere.
hipError_t launch_pack_count(hipStream_t s, const Records& rec, uint32_t n, const BandEdges& bands,
                             const uint2* window, uint32_t tiles_x, unsigned long long* masks, uint32_t* table,
                             const uint2* list, const uint32_t* d_list_n, unsigned long long* travellers, uint32_t* traveller_counts,
                             const uint32_t* gate = nullptr /* tile bitmap: records whose rectangle holds no gated tile go nowhere */,
                             uint32_t gate_row_words = 0,
                             const WindowPyramid* pyramid = nullptr /* decide by the windows' pyramid alone: a conservative superset */,
                             const uint32_t* d_skip = nullptr /* points at 0: nothing travels, nothing is read or written (an always-enqueued repair round) */);
// The slots of an exchange buffer, in records: slot p = one header record at off[p], then up to cap[p] records.
struct SlotSpans {
    uint32_t off[kMaxRanks], cap[kMaxRanks];
};
inline SlotSpans uniform_slots(uint32_t world, uint32_t cap) {
    SlotSpans sp{};
    for (uint32_t p = 0; p < world; ++p) {
        sp.off[p] = p * (cap + 1u);
        sp.cap[p] = cap;
    }
    return sp;
}
hipError_t launch_pack_scatter(hipStream_t s, const Records& rec, uint32_t n, uint32_t world,
                               const unsigned long long* masks, const uint32_t* table, const uint32_t* totals, void* d_send,
                               uint64_t capacity, const uint2* list, const uint32_t* d_list_n,
                               const SlotSpans* slots = nullptr /* != nullptr: into these slots, headers first, at most cap[p] records each */,
                               const uint32_t* d_skip = nullptr);
// device-resident exchange (kernels_shard.hip): slot headers, import from slots, windows / verification / next limits
hipError_t launch_pack_headers(hipStream_t s, const uint32_t* totals, uint32_t world, const SlotSpans& slots, void* d_send, SlabStats* stats, uint32_t round,
                               const uint32_t* d_skip = nullptr);
// staged: the round-0 verdict block in device memory (launch_shard_verify with a device pointer); sat (nullable): the feedback gathered
// after the repair round; host_block: the frame's slot of the pinned ring
hipError_t launch_shard_post_verdict(hipStream_t s, const uint32_t* staged, const uint32_t* sat, uint32_t world, uint32_t stride, uint32_t* host_block, uint32_t seq);
hipError_t launch_import_slots(hipStream_t s, const void* d_recv, uint32_t world, const SlotSpans& slots, const Records& rec, SlabStats* stats);
hipError_t launch_limits_to_windows(hipStream_t s, const uint32_t* limit, uint32_t n_tiles, uint2* win);
// sat: the all-gathered feedback, piece g at word g * feedback_stride(bands, tiles_x) (layout: feedback_* below)
hipError_t launch_shard_verify(hipStream_t s, const uint32_t* limit, const uint32_t* sat, uint32_t tiles_x, uint32_t tiles_y, const BandEdges& bands,
                               uint2* win2, uint32_t* d_need, uint32_t* d_ticket, unsigned long long* host_verdict, uint32_t seq,
                               uint32_t* need_bits /* zeroed; bit per tile that needs the repair round */, uint32_t balance /* post balanced edges for the next frame */);
hipError_t launch_shard_post_counts(hipStream_t s, const uint32_t* counts_all, uint32_t world, unsigned long long* host_verdict, uint32_t seq);
hipError_t launch_shard_max_count(hipStream_t s, const uint32_t* totals, uint32_t world, uint32_t* out4);
// win_next (nullable): also the next frame's round-0 windows [0, limit); staged + host_block (nullable): block 0 posts the frame's verdict on
// its way (what launch_shard_post_verdict does as a launch of its own; sat_verdict: the feedback gathered after the repair round, or nullptr)
hipError_t launch_shard_next_limits(hipStream_t s, const uint32_t* sat, uint32_t tiles_x, uint32_t tiles_y, float margin, uint32_t radius,
                                    uint32_t* limit, const BandEdges& bands, uint2* win_next = nullptr, const uint32_t* staged = nullptr,
                                    const uint32_t* sat_verdict = nullptr, uint32_t* host_block = nullptr, uint32_t seq = 0);
// this rank's feedback piece (layout: feedback_* above)
hipError_t launch_shard_feedback(hipStream_t s, const uint32_t* tile_sat, const uint32_t* row_work, uint32_t tiles_x, uint32_t tiles_y, const BandEdges& bands,
                                 uint32_t rank, uint32_t* out, const SlabStats* stats, const uint32_t* done_before, uint32_t row_words, uint32_t gather_root_plus1,
                                 uint32_t policy_flags = 0, uint32_t* za = nullptr, uint32_t nza = 0, uint32_t* zb = nullptr, uint32_t nzb = 0 /* words zeroed on the way */);
constexpr uint32_t kPolicyBalance = 1u, kPolicyPairSlots = 2u;  // feedback word [6]; bits 8.. : the model's forced slot size (gsx_shard_set_slot_records)
hipError_t launch_import_records(hipStream_t s, const void* d_recv, uint32_t n, const Records& rec);

// Mask evaluation (kernels_mask.hip); passed to the kernel by value.
struct MaskShapeConsts {
    uint32_t kind;
    float pos[3];
    float rot[9];  // row-major rotation of the shape
    float scale[3];
    float box_lim[3];  // box: |d / scale| <= 1 decided without the division (mask_box_limit)
};
// The box test of the reference's mask (|d / s| <= 1 per axis, d / s correctly rounded) as ONE compare |d| <= lim:
// RN(|d| / |s|) <= 1  <=>  |d| / |s| <= 1 + 2^-24 (the tie rounds to even, 1.0)  <=>  |d| <= |s|, because the float next above |s| is
// |s| + ulp(|s|) with ulp(|s|) / |s| > 2^-24 (denormal |s| included).  s = 0 or NaN: the quotient is inf or NaN, never inside (-1);
// s = +-inf: every finite d gives 0, inside (FLT_MAX; d = inf gives NaN: outside either way).
inline float mask_box_limit(float s) {
    if (s != s || s == 0.0f) return -1.0f;
    const float a = s < 0.0f ? -s : s;
    return a > 3.402823466e+38f ? 3.402823466e+38f : a;
}
struct MaskProgram {
    float m_rot[9], m_pos[3], m_scale[3];  // model transform
    uint32_t n_shapes, n_ops;
    MaskShapeConsts shapes[GSX_MASK_MAX_SHAPES];
    gsx_mask_op ops[GSX_MASK_MAX_OPS];
};
hipError_t launch_mask_evaluate(hipStream_t s, const float4* pc, uint32_t n, const MaskProgram& prog, uint32_t* mask);
void quat_to_rows(const float q[4], float r[9]);

// Compositing and resolve.
// carry: continue from the (C, T) already in fb (later slabs / models behind); done: saturated-tile bitmap
// (read to skip tiles when carrying, updated when a tile saturates; nullable).
// depth feedback (nullable): when a tile saturates, *depth_needed = max(., sorted_keys[min(slab_end, *d_n_vis) - 1])
hipError_t launch_composite(hipStream_t s, const FrameConsts& f, uint2* ranges, const uint32_t* list,
                            const Records& rec, float4* fb, bool carry, uint32_t* done, uint32_t row_words,
                            uint32_t* d_done_count, bool clear_ranges, uint32_t* tile_sat, uint32_t* row_work = nullptr /* per tile row: += list entries walked */);
// The splats [stats->slab_cut, min(j1, *d_n)) of the depth order, composited WITHOUT tile pairs (an overflowing slab's tail):
// one workgroup per live tile scans them, keeps those whose rectangle (and window) takes the tile, blends them like
// k_composite.  Falls through when the slab was not cut.
hipError_t launch_composite_spill(hipStream_t s, const FrameConsts& f, const SlabStats* stats, uint32_t j1, const uint32_t* d_n,
                                  const uint32_t* sorted_idx, const uint32_t* sorted_keys, const Records& rec, float4* fb,
                                  uint32_t* done, uint32_t row_words, uint32_t* d_done_count, uint32_t* tile_sat, uint32_t row_lo,
                                  uint32_t row_hi, const uint2* window);
hipError_t launch_clear_fb(hipStream_t s, float4* fb, uint32_t n_px);
hipError_t launch_resolve_rgba8(hipStream_t s, const float4* fb, uint32_t n_px, float bg_r, float bg_g, float bg_b,
                                uint32_t* out_rgba8);

}  // namespace gsx
