// tools/bench_sort.hip — micro-benchmark + phase profile of the onesweep radix pass (csrc/kernels_sort.hip, included as
// source with GSX_SORT_PROFILE).  Not part of the product.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -DGSX_SORT_PROFILE tools/bench_sort.hip \
//         -Iwgpu_3dgs_viewer_app_amd/csrc -Iinclude -o tools/bench_sort
//   tools/bench_sort [n=5300000] [bits=13]
#define GSX_LAUNCH_STANDALONE 1  // csrc/gsx_launch.h: launches submit at once, nothing of libgsx is linked
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../wgpu_3dgs_viewer_app_amd/csrc/kernels_sort.hip"

using namespace gsx;
#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 5300000u;
    const int bits = argc > 2 ? atoi(argv[2]) : 13;
    printf("returning LDS adds lane-ordered on this device: %d\n", (int)radix_lane_ordered_adds());
    std::mt19937 rng(7);
    std::vector<uint2> h(n);
    const bool depth = argc > 3;  // third argument: keys are float bits of depths in [0.2, 12) (one or two exponent bytes)
    const bool dup = argc > 3 && !strcmp(argv[3], "dup");  // "dup": only 997 distinct depths -> every key is shared by thousands
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t k = bits >= 32 ? rng() : rng() % ((1u << bits) - 31u);
        if (depth) {
            const float f = dup ? 0.2f + 11.8f * (float)(rng() % 997u) / 997.0f : 0.2f + 11.8f * (float)(rng() >> 8) / 16777216.0f;
            memcpy(&k, &f, 4);
        }
        h[i] = make_uint2(k, i);
    }
    uint2 *src, *pa, *pb;
    uint32_t *ko, *vo, *ws, *dn;
    CK(hipMalloc(&src, 8ull * n));
    CK(hipMalloc(&pa, 8ull * n));
    CK(hipMalloc(&pb, 8ull * n));
    CK(hipMalloc(&ko, 4ull * n));
    CK(hipMalloc(&vo, 4ull * n));
    CK(hipMalloc(&dn, 4));
    const size_t wsw = radix_workspace_words(n);
    CK(hipMalloc(&ws, 4 * wsw));
    CK(hipMemset(ws, 0, 4 * wsw));
    CK(hipMemcpy(src, h.data(), 8ull * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dn, &n, 4, hipMemcpyHostToDevice));
    RadixBuffers rb{nullptr, nullptr, src, ko, vo, pa, pb, ws};
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(launch_radix_sort(s, rb, n, dn, bits, false));
    CK(hipStreamSynchronize(s));
    const int reps = 20;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(launch_radix_sort(s, rb, n, dn, bits, false));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float ms_pairs = ms;
    const int passes = (bits + 7) / 8;
    printf("n %u bits %d: %.1f us per sort (%d passes + hist), %.2f TB/s algorithmic (16 B per element per pass)\n", n, bits,
           1000.0 * ms_pairs / reps, passes, 16.0 * n * passes / (ms_pairs / reps * 1e-3) / 1e12);
    if (argc > 3 && !strcmp(argv[3], "skip")) {
        // the depth sort of an unspeculated frame: the projection's key plane as it lies, 15 % of it culled (0xFFFFFFFF): the first
        // pass compacts on the way, *dn receives the number of records that exist
        std::vector<uint32_t> keys(n);
        std::vector<uint2> ref2;
        for (uint32_t i = 0; i < n; ++i) {
            const float f = 0.2f + 11.8f * (float)(rng() >> 8) / 16777216.0f;
            uint32_t kk;
            memcpy(&kk, &f, 4);
            keys[i] = (rng() % 100u) < 15u ? 0xFFFFFFFFu : kk;
            if (keys[i] != 0xFFFFFFFFu) ref2.push_back(make_uint2(keys[i], i));
        }
        uint32_t* dkeys;
        CK(hipMalloc(&dkeys, 4ull * n));
        CK(hipMemcpy(dkeys, keys.data(), 4ull * n, hipMemcpyHostToDevice));
        RadixBuffers rk{dkeys, nullptr, nullptr, ko, vo, pa, pb, ws};
        for (int i = 0; i < 3; ++i) CK(launch_radix_sort(s, rk, n, dn, 32, true, true));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(launch_radix_sort(s, rk, n, dn, 32, true, true));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t got_n = 0;
        CK(hipMemcpy(&got_n, dn, 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> k2(n), v2(n);
        CK(hipMemcpy(k2.data(), ko, 4ull * n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(v2.data(), vo, 4ull * n, hipMemcpyDeviceToHost));
        std::stable_sort(ref2.begin(), ref2.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
        size_t bad2 = got_n != ref2.size();
        for (size_t i = 0; i < ref2.size() && i < got_n; ++i) bad2 += (ref2[i].x != k2[i]) || (ref2[i].y != v2[i]);
        printf("skip mode: n %u, %zu exist (device says %u): %.1f us per sort; mismatches vs std::stable_sort of the existing ones: %zu\n", n,
               ref2.size(), got_n, 1000.0 * ms / reps, bad2);
        CK(hipMemcpy(dn, &n, 4, hipMemcpyHostToDevice));
        CK(launch_radix_sort(s, rb, n, dn, bits, false));  // (the pair sort's outputs again, for the check below)
        CK(hipStreamSynchronize(s));
    }
    // ---- the bucket sort (fine histogram -> MSD partition -> in-LDS bucket sorts) on the same pairs: three launches ----
    // GSX_BUCKET_CAP=<pairs> forces buckets above that size through the global-memory path
    {
        if (getenv("GSX_BUCKET_CAP")) bucket_sort_set_cap((uint32_t)atoi(getenv("GSX_BUCKET_CAP")));
        uint32_t* mws;
        const size_t mwords = msd_workspace_words(n);
        CK(hipMalloc(&mws, 4 * mwords));
        CK(msd_workspace_init(s, mws, mwords));
        uint32_t *ko2, *vo2;
        CK(hipMalloc(&ko2, 4ull * n + 4));
        CK(hipMalloc(&vo2, 4ull * n + 4));
        RadixBuffers rb2{nullptr, nullptr, src, ko2, vo2, pa, pb, ws};
        uint32_t seq = 0;
        std::vector<uint2> ref = h;
        std::stable_sort(ref.begin(), ref.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
        std::vector<uint32_t> k2(n), v2(n);
        size_t bad_total = 0;
        for (int round = 0; round < 3; ++round) {  // round 0: no key range known yet; 1, 2: the range of the sort before
            CK(hipMemset(ko2, 0xEE, 4ull * n));
            CK(launch_bucket_sort(s, rb2, n, dn, false, mws, seq++, false));
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(k2.data(), ko2, 4ull * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(v2.data(), vo2, 4ull * n, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (uint32_t i = 0; i < n; ++i) bad += (ref[i].x != k2[i]) || (ref[i].y != v2[i]);
            printf("bucket sort round %d: mismatches vs std::stable_sort: %zu\n", round, bad);
            bad_total += bad;
        }
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(launch_bucket_sort(s, rb2, n, dn, false, mws, seq++, false));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("bucket sort: n %u: %.1f us per sort (histogram + partition + bucket launch) against %.1f us for the LSD sort\n", n, 1000.0 * ms / reps,
               1000.0 * ms_pairs / reps);
        // a device-side count below the launch bound, and an empty sort
        for (uint32_t part : {n / 3u, 0u}) {
            CK(hipMemcpy(dn, &part, 4, hipMemcpyHostToDevice));
            CK(launch_bucket_sort(s, rb2, n, dn, false, mws, seq++, false));
            CK(hipStreamSynchronize(s));
            std::vector<uint2> refp(h.begin(), h.begin() + part);
            std::stable_sort(refp.begin(), refp.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
            CK(hipMemcpy(k2.data(), ko2, 4ull * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(v2.data(), vo2, 4ull * n, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (uint32_t i = 0; i < part; ++i) bad += (refp[i].x != k2[i]) || (refp[i].y != v2[i]);
            printf("bucket sort of the first %u pairs (count on the device): mismatches %zu\n", part, bad);
            bad_total += bad;
        }
        CK(hipMemcpy(dn, &n, 4, hipMemcpyHostToDevice));
        // keys as they lie + iota values (an imported band)
        {
            std::vector<uint32_t> keys(n);
            for (uint32_t i = 0; i < n; ++i) keys[i] = h[i].x;
            uint32_t* dkeys;
            CK(hipMalloc(&dkeys, 4ull * n + 4));
            CK(hipMemcpy(dkeys, keys.data(), 4ull * n, hipMemcpyHostToDevice));
            RadixBuffers rk{dkeys, nullptr, nullptr, ko2, vo2, pa, pb, ws};
            CK(launch_bucket_sort(s, rk, n, dn, true, mws, seq++, false));
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(k2.data(), ko2, 4ull * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(v2.data(), vo2, 4ull * n, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (uint32_t i = 0; i < n; ++i) bad += (ref[i].x != k2[i]) || (ref[i].y != v2[i]);
            printf("bucket sort from a key array: mismatches %zu\n", bad);
            bad_total += bad;
        }
        // keys that have LEFT the range the earlier sorts saw (a camera jump; a repair round that admits the far half of the scene): the
        // guessed range only decides the balance — same order, and no bucket may swallow everything
        {
            std::vector<uint2> h2(n);
            for (uint32_t i = 0; i < n; ++i) {
                const float f = (i & 7u) == 0u ? 0.001f + 0.01f * (float)(rng() >> 8) / 16777216.0f : 20.0f + 2980.0f * (float)(rng() >> 8) / 16777216.0f;
                uint32_t kk;
                memcpy(&kk, &f, 4);
                h2[i] = make_uint2(kk, i);
            }
            uint2* src2;
            CK(hipMalloc(&src2, 8ull * n + 8));
            CK(hipMemcpy(src2, h2.data(), 8ull * n, hipMemcpyHostToDevice));
            RadixBuffers rb3{nullptr, nullptr, src2, ko2, vo2, pa, pb, ws};
            CK(hipEventRecord(e0, s));
            CK(launch_bucket_sort(s, rb3, n, dn, false, mws, seq++, false));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(k2.data(), ko2, 4ull * n, hipMemcpyDeviceToHost));
            CK(hipMemcpy(v2.data(), vo2, 4ull * n, hipMemcpyDeviceToHost));
            std::stable_sort(h2.begin(), h2.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
            size_t bad = 0;
            for (uint32_t i = 0; i < n; ++i) bad += (h2[i].x != k2[i]) || (h2[i].y != v2[i]);
            printf("bucket sort of keys outside every range seen before: %.1f us, mismatches %zu\n", 1000.0 * ms, bad);
            bad_total += bad;
        }
        printf("bucket sort mismatches in all: %zu\n", bad_total);
        CK(launch_radix_sort(s, rb, n, dn, bits, false));  // (the pair sort's outputs again, for the check below)
        CK(hipStreamSynchronize(s));
    }
    // correctness
    std::vector<uint32_t> k(n), v(n);
    CK(hipMemcpy(k.data(), ko, 4ull * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(v.data(), vo, 4ull * n, hipMemcpyDeviceToHost));
    std::vector<uint2> ref = h;
    std::stable_sort(ref.begin(), ref.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
    size_t bad = 0;
    for (uint32_t i = 0; i < n; ++i) bad += (ref[i].x != k[i]) || (ref[i].y != v[i]);
    printf("mismatches vs std::stable_sort: %zu\n", bad);
    // phase profile of one sort (all passes write the same slots: the last pass remains)
    const uint32_t tile_elems = n <= radix_small_n() ? kRadixTileSmall : kRadixTile;
    const uint32_t tiles = (n + tile_elems - 1) / tile_elems;
    long long* prof;
    CK(hipMalloc(&prof, 64ull * tiles));
    CK(hipMemset(prof, 0, 64ull * tiles));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_sort_prof), &prof, sizeof(prof)));
    CK(launch_radix_sort(s, rb, n, dn, bits, false));
    CK(hipStreamSynchronize(s));
    std::vector<long long> hp(8ull * tiles);
    CK(hipMemcpy(hp.data(), prof, 64ull * tiles, hipMemcpyDeviceToHost));
    long long t0 = hp[0], t1 = 0;
    double ph[5] = {0, 0, 0, 0, 0};
    for (uint32_t t = 0; t < tiles; ++t) {
        t0 = std::min(t0, hp[8ull * t]);
        t1 = std::max(t1, hp[8ull * t + 5]);
        for (int p = 0; p < 5; ++p) ph[p] += (double)(hp[8ull * t + p + 1] - hp[8ull * t + p]);
    }
    printf("last pass: %u tiles, span %.1f us; mean per tile (us): load+rank %.2f, publish %.2f, LDS reorder %.2f, look-back %.2f, write %.2f\n",
           tiles, (t1 - t0) / 100.0, ph[0] / tiles / 100, ph[1] / tiles / 100, ph[2] / tiles / 100, ph[3] / tiles / 100, ph[4] / tiles / 100);
    for (uint32_t t : {0u, 1u, tiles / 4, tiles / 2, tiles - 1}) {
        printf("  tile %5u: start +%.1f us:", t, (hp[8ull * t] - t0) / 100.0);
        for (int p = 0; p < 5; ++p) printf(" %.2f", (hp[8ull * t + p + 1] - hp[8ull * t + p]) / 100.0);
        printf("\n");
    }
    return 0;
}
