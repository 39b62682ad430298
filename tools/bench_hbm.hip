// tools/bench_hbm.hip — what this box's HBM delivers for the projection pass's access SHAPE, without its arithmetic:
// N = 10 M "Gaussians", one per lane, 256 per workgroup; R float4 planes read (non-temporal), W float4 planes written.
// The projection pass reads 13.75 planes (220 B) and writes 3.25 (52 B) per visible Gaussian.  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bench_hbm.hip -o tools/bench_hbm && tools/bench_hbm
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));

// R planes read, W planes written; NT: non-temporal loads; NTS: non-temporal stores; PER: elements per lane (256 apart)
template <int R, int W, bool NT, bool NTS, int PER>
__global__ __launch_bounds__(256) void k_planes(const f4v* __restrict__ src, f4v* __restrict__ dst, uint32_t n) {
    f4v acc[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) acc[k] = f4v{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = blockIdx.x * (256u * PER) + 256u * k + threadIdx.x;
        if (i < n) {
#pragma unroll
            for (int p = 0; p < R; ++p) {
                const f4v v = NT ? __builtin_nontemporal_load(&src[(size_t)p * n + i]) : src[(size_t)p * n + i];
                acc[k] += v;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = blockIdx.x * (256u * PER) + 256u * k + threadIdx.x;
        if (i < n) {
#pragma unroll
            for (int p = 0; p < W; ++p) {
                const f4v v = acc[k] + (float)p;
                if (NTS) __builtin_nontemporal_store(v, &dst[(size_t)p * n + i]);
                else dst[(size_t)p * n + i] = v;
            }
            if (W == 0 && acc[k].x == 1234.5f) dst[i] = acc[k];
        }
    }
}

// dependent stages like the projection: plane 0 first, then 2 planes, then the rest
template <int R, int W>
__global__ __launch_bounds__(256) void k_staged(const f4v* __restrict__ src, f4v* __restrict__ dst, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    f4v a = __builtin_nontemporal_load(&src[i]);
    const uint32_t j = a.x == 777.0f ? 0u : i;  // data dependence, never taken
    f4v b = __builtin_nontemporal_load(&src[(size_t)1 * n + j]) + __builtin_nontemporal_load(&src[(size_t)2 * n + j]);
    const uint32_t k = b.x == 777.0f ? 0u : i;
    f4v acc = a + b;
#pragma unroll
    for (int p = 3; p < R; ++p) acc += __builtin_nontemporal_load(&src[(size_t)p * n + k]);
#pragma unroll
    for (int p = 0; p < W; ++p) dst[(size_t)p * n + i] = acc + (float)p;
    if (W == 0 && acc.x == 1234.5f) dst[i] = acc;
}

template <class F> static float time_ms(F launch, int reps = 12) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000u;
    f4v *src, *dst;
    CK(hipMalloc(&src, 16ull * n * 14));
    CK(hipMalloc(&dst, 16ull * n * 4));
    CK(hipMemset(src, 0, 16ull * n * 14));
    CK(hipMemset(dst, 0, 16ull * n * 4));
    const uint32_t nb = (n + 255) / 256;
#define RUN(name, R, W, ...)                                                                                       \
    {                                                                                                              \
        const float ms = time_ms([&] { __VA_ARGS__; });                                                             \
        const double gb = 16.0 * n * ((R) + (W)) / 1e9;                                                             \
        printf("%-52s %6.3f ms  %5.2f GB  %5.2f TB/s\n", name, ms, gb, gb / ms);                                   \
    }
    RUN("read 14 planes, NT", 14, 0, hipLaunchKernelGGL((k_planes<14, 0, true, false, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 14 planes, default loads", 14, 0, hipLaunchKernelGGL((k_planes<14, 0, false, false, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 14 planes, NT, 2 per lane", 14, 0, hipLaunchKernelGGL((k_planes<14, 0, true, false, 2>), dim3((nb + 1) / 2), dim3(256), 0, 0, src, dst, n));
    RUN("read 14, write 3, NT loads", 14, 3, hipLaunchKernelGGL((k_planes<14, 3, true, false, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 14, write 3, NT loads + NT stores", 14, 3, hipLaunchKernelGGL((k_planes<14, 3, true, true, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 14, write 3, default", 14, 3, hipLaunchKernelGGL((k_planes<14, 3, false, false, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 14 write 3, NT, staged 1 -> 2 -> 11 (dependent)", 14, 3, hipLaunchKernelGGL((k_staged<14, 3>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 14, staged, no writes", 14, 0, hipLaunchKernelGGL((k_staged<14, 0>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 4 write 4 (copy), NT loads", 4, 4, hipLaunchKernelGGL((k_planes<4, 4, true, false, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 4 write 4 (copy), NT both", 4, 4, hipLaunchKernelGGL((k_planes<4, 4, true, true, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 2 planes write 1 (geometry-only shape), NT", 2, 1, hipLaunchKernelGGL((k_planes<2, 1, true, false, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 2 write 1, NT both", 2, 1, hipLaunchKernelGGL((k_planes<2, 1, true, true, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 2 write 1, NT, 4 per lane", 2, 1, hipLaunchKernelGGL((k_planes<2, 1, true, false, 4>), dim3((nb + 3) / 4), dim3(256), 0, 0, src, dst, n));
    RUN("read 1 plane, NT", 1, 0, hipLaunchKernelGGL((k_planes<1, 0, true, false, 1>), dim3(nb), dim3(256), 0, 0, src, dst, n));
    RUN("read 1 plane, NT, 4 per lane", 1, 0, hipLaunchKernelGGL((k_planes<1, 0, true, false, 4>), dim3((nb + 3) / 4), dim3(256), 0, 0, src, dst, n));
    return 0;
}
