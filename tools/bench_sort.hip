// tools/bench_sort.hip — micro-benchmark + phase profile of the onesweep radix pass (csrc/kernels_sort.hip, included as
// source with GSX_SORT_PROFILE).  Not part of the product.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -DGSX_SORT_PROFILE tools/bench_sort.hip \
//         -Iwgpu_3dgs_viewer_app_amd/csrc -Iinclude -o tools/bench_sort
//   tools/bench_sort [n=5300000] [bits=13]
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
    const int passes = (bits + 7) / 8;
    printf("n %u bits %d: %.1f us per sort (%d passes + hist), %.2f TB/s algorithmic (16 B per element per pass)\n", n, bits,
           1000.0 * ms / reps, passes, 16.0 * n * passes / (ms / reps * 1e-3) / 1e12);
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
    const uint32_t tiles = (n + kRadixTile - 1) / kRadixTile;
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
