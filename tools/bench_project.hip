// tools/bench_project.hip — micro-benchmark of memory-access structures for the projection pass.
// Not part of the product; it exists to choose the HBM layout / load schedule of kernels_project.hip by
// measurement (within-process interleaved A/B, same arithmetic from csrc/project_math.h).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/bench_project.hip \
//         -Iwgpu_3dgs_viewer_app_amd/csrc -Lwgpu_3dgs_viewer_app_amd -lgsx -Wl,-rpath,'$ORIGIN/../wgpu_3dgs_viewer_app_amd' -o tools/bench_project
#define GSX_LAUNCH_STANDALONE 1  // csrc/gsx_launch.h: launches submit at once, nothing of libgsx is linked
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>

#include "project_math.h"

using namespace gsx;

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

// ---- layouts ------------------------------------------------------------------------------------
// SoA planes: pc[N] cov_a[N] cov_b[N] sh4[11][N] sh1[N]
// Chunked:   per 256 Gaussians one contiguous 56,320-byte block: pc | cov_a | cov_b | sh4[0..10] | sh1
constexpr uint32_t kChunk = 256, kChunkBytes = 256 * 220, kRecChunkBytes = 256 * 52;

struct In {
    const char* base;  // chunked
    const float4 *pc, *cov_a, *sh4;
    const float2* cov_b;
    const float* sh1;
    uint32_t n;
    size_t sh_stride;  // float4 elements between consecutive SH planes (product: n)
};
struct Out {
    char* base;  // chunked
    float4 *a, *b, *c;
    uint32_t* key;
};

__device__ inline float4 nt4(const float4* p) {
    return make_float4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y), __builtin_nontemporal_load(&p->z),
                       __builtin_nontemporal_load(&p->w));
}
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ inline float4 nt4v(const float4* p) {
    float4v v = __builtin_nontemporal_load((const float4v*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int LAYOUT> __device__ inline float4 ld_pc(const In& in, uint32_t i) {
    if (LAYOUT == 2) return nt4v(&in.pc[i]);
    if (LAYOUT == 0) return in.pc[i];
    return *(const float4*)(in.base + (size_t)(i >> 8) * kChunkBytes + (i & 255u) * 16u);
}
template <int LAYOUT> __device__ inline float4 ld_cova(const In& in, uint32_t i) {
    if (LAYOUT == 2) return nt4v(&in.cov_a[i]);
    if (LAYOUT == 0) return in.cov_a[i];
    return *(const float4*)(in.base + (size_t)(i >> 8) * kChunkBytes + 4096u + (i & 255u) * 16u);
}
template <int LAYOUT> __device__ inline float2 ld_covb(const In& in, uint32_t i) {
    if (LAYOUT == 0 || LAYOUT == 2) return in.cov_b[i];
    return *(const float2*)(in.base + (size_t)(i >> 8) * kChunkBytes + 8192u + (i & 255u) * 8u);
}
template <int LAYOUT> __device__ inline float4 ld_sh4(const In& in, int p, uint32_t i) {
    if (LAYOUT == 2) return nt4v(&in.sh4[(size_t)p * in.sh_stride + i]);
    if (LAYOUT == 0) return in.sh4[(size_t)p * in.sh_stride + i];
    return *(const float4*)(in.base + (size_t)(i >> 8) * kChunkBytes + 10240u + p * 4096u + (i & 255u) * 16u);
}
template <int LAYOUT> __device__ inline float ld_sh1(const In& in, uint32_t i) {
    if (LAYOUT == 0 || LAYOUT == 2) return in.sh1[i];
    return *(const float*)(in.base + (size_t)(i >> 8) * kChunkBytes + 55296u + (i & 255u) * 4u);
}
template <int LAYOUT> __device__ inline void st_rec(const Out& o, uint32_t i, bool vis, uint32_t key, float4 a, float4 b, float4 c) {
    if (LAYOUT == 0 || LAYOUT == 2) {
        o.key[i] = key;
        if (vis) { o.a[i] = a; o.b[i] = b; o.c[i] = c; }
    } else {
        char* p = o.base + (size_t)(i >> 8) * kRecChunkBytes;
        *(uint32_t*)(p + 12288u + (i & 255u) * 4u) = key;
        if (vis) {
            *(float4*)(p + (i & 255u) * 16u) = a;
            *(float4*)(p + 4096u + (i & 255u) * 16u) = b;
            *(float4*)(p + 8192u + (i & 255u) * 16u) = c;
        }
    }
}

// EAGER 0: load pc -> cull -> load cov -> rect -> load sh -> colour (loads only for survivors)
// EAGER 1: issue every load first, then compute
// EAGER 2: pc first (cull), then cov + sh together for frustum survivors
template <int LAYOUT, int EAGER, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_var(const FrameConsts f, In in, Out out, uint32_t* nvis) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool vis = i < in.n;
    float4 pc = make_float4(0, 0, 0, 0), cva = pc;
    float2 cvb = make_float2(0, 0);
    float s[48];
    if (EAGER == 1 && vis) {
        pc = ld_pc<LAYOUT>(in, i);
        cva = ld_cova<LAYOUT>(in, i);
        cvb = ld_covb<LAYOUT>(in, i);
#pragma unroll
        for (int p = 0; p < 11; ++p) {
            float4 v = ld_sh4<LAYOUT>(in, p, i);
            s[4 * p] = v.x; s[4 * p + 1] = v.y; s[4 * p + 2] = v.z; s[4 * p + 3] = v.w;
        }
        s[44] = ld_sh1<LAYOUT>(in, i);
    } else if (vis) {
        pc = ld_pc<LAYOUT>(in, i);
    }
    ViewClip vc;
    vis = pm_view_cull(f, pc.x, pc.y, pc.z, vc) && vis;
    if (EAGER == 2 && vis) {
        cva = ld_cova<LAYOUT>(in, i);
        cvb = ld_covb<LAYOUT>(in, i);
#pragma unroll
        for (int p = 0; p < 11; ++p) {
            float4 v = ld_sh4<LAYOUT>(in, p, i);
            s[4 * p] = v.x; s[4 * p + 1] = v.y; s[4 * p + 2] = v.z; s[4 * p + 3] = v.w;
        }
        s[44] = ld_sh1<LAYOUT>(in, i);
    }
    Splat2D sp{};
    if (vis) {
        if (EAGER == 0) {
            cva = ld_cova<LAYOUT>(in, i);
            cvb = ld_covb<LAYOUT>(in, i);
        }
        vis = pm_cov2d_rect(f, vc, cva.x, cva.y, cva.z, cva.w, cvb.x, cvb.y, sp);
    }
    float r = 0, g = 0, b = 0;
    const uint32_t color = __float_as_uint(pc.w);
    if (vis) {
        if (EAGER == 0) {
#pragma unroll
            for (int p = 0; p < 11; ++p) {
                float4 v = ld_sh4<LAYOUT>(in, p, i);
                s[4 * p] = v.x; s[4 * p + 1] = v.y; s[4 * p + 2] = v.z; s[4 * p + 3] = v.w;
            }
            s[44] = ld_sh1<LAYOUT>(in, i);
        }
        pm_color<3>(f, pc.x, pc.y, pc.z, color, s, r, g, b);
    }
    if (i < in.n)
        st_rec<LAYOUT>(out, i, vis, vis ? __float_as_uint(vc.d) : kCulledKey,
                       make_float4(sp.mx, sp.my, __uint_as_float(sp.rx), __uint_as_float(sp.ry)),
                       make_float4(sp.con_a, sp.con_b, sp.con_c, (float)(color >> 24) * (1.0f / 255.0f)),
                       make_float4(r, g, b, vc.d));
    // per-workgroup count, no same-address atomics (156 K wave atomics on one word cost 1.8 ms by themselves)
    __shared__ uint32_t wc[4];
    unsigned long long bal = __ballot(vis);
    if ((threadIdx.x & 63u) == 0) wc[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) nvis[1 + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ void k_sum(uint32_t* nvis, uint32_t nb) {
    __shared__ uint32_t red[256];
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < nb; i += 256) s += nvis[1 + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) nvis[0] = red[0];
}

// plain streaming copy of the same byte volume: the practical HBM ceiling for this read/write mix
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n_read, size_t n_write) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t k = i; k < n_read; k += stride) {
        float4 v = src[k];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        if (k < n_write) dst[k] = v;
    }
    if (acc.x == 1234.5f) dst[0] = acc;
}

__device__ inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ inline float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// same synthetic values in both layouts
__global__ void k_fill(In soa, char* chunked, uint32_t n) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float v[55];
    for (int k = 0; k < 55; ++k) v[k] = u01(hash32(i * 64u + k));
    float4 pc = make_float4(v[0] * 8 - 4, v[1] * 8 - 4, v[2] * 8 - 4, __uint_as_float(hash32(i) | 0x80000000u));
    float sx = 0.002f + 0.03f * v[3], sy = 0.002f + 0.03f * v[4], sz = 0.002f + 0.03f * v[5];
    float4 ca = make_float4(sx * sx, 0.1f * sx * sy, 0.0f, sy * sy);
    float2 cb = make_float2(0.1f * sy * sz, sz * sz);
    ((float4*)soa.pc)[i] = pc;
    ((float4*)soa.cov_a)[i] = ca;
    ((float2*)soa.cov_b)[i] = cb;
    char* c = chunked + (size_t)(i >> 8) * kChunkBytes;
    uint32_t l = i & 255u;
    if (chunked) {
        *(float4*)(c + l * 16) = pc;
        *(float4*)(c + 4096 + l * 16) = ca;
        *(float2*)(c + 8192 + l * 8) = cb;
    }
    for (int p = 0; p < 11; ++p) {
        float4 s = make_float4(v[6 + 4 * p] * 0.3f - 0.15f, v[7 + 4 * p] * 0.3f - 0.15f, v[8 + 4 * p] * 0.3f - 0.15f, v[9 + 4 * p] * 0.3f - 0.15f);
        ((float4*)soa.sh4)[(size_t)p * soa.sh_stride + i] = s;
        if (chunked) *(float4*)(c + 10240 + p * 4096 + l * 16) = s;
    }
    ((float*)soa.sh1)[i] = v[50] * 0.3f - 0.15f;
    if (chunked) *(float*)(c + 55296 + l * 4) = v[50] * 0.3f - 0.15f;
}

static void look_at(const float e[3], float view[16]) {
    float f[3] = {-e[0], -e[1], -e[2]};
    float l = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (float& x : f) x /= l;
    float up[3] = {0, 1, 0};
    float s[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    l = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for (float& x : s) x /= l;
    float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    float m[16] = {s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0,
                   -(e[0] * s[0] + e[1] * s[1] + e[2] * s[2]), -(e[0] * u[0] + e[1] * u[1] + e[2] * u[2]),
                   (e[0] * f[0] + e[1] * f[1] + e[2] * f[2]), 1};
    for (int i = 0; i < 16; ++i) view[i] = m[i];
}

// Placement sweep: the 15 input planes and 4 output planes carved from ONE slab, plane p starting at
// p * (round_up(N*16, 2 MiB) + skew).  Shows how much the relative alignment of the 19 concurrent
// streams matters for HBM channel balance.
static void placement_sweep(uint32_t n, const FrameConsts& f, uint32_t* nvis, int rounds) {
    const size_t plane = (((size_t)n * 16 + 255) >> 8) << 8;
    const size_t r2m = ((((size_t)n * 16 + (2u << 20) - 1) >> 21) << 21) - plane;  // pad up to a 2 MiB multiple
    const size_t skews[] = {0, 4096, r2m, r2m + 256};
    char* slab;
    const size_t max_skew = skews[sizeof(skews) / sizeof(skews[0]) - 1];
    CK(hipMalloc((void**)&slab, 21 * (plane + max_skew)));
    CK(hipMemset(slab, 0, 21 * (plane + max_skew)));
    uint32_t nchunks = (n + 255) / 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    char* sep[20];
    for (int p = 0; p < 20; ++p) CK(hipMalloc((void**)&sep[p], p == 4 ? 11 * plane : plane));
    for (int mode = 0; mode < 3; ++mode)
    for (size_t skew : skews) {
        if (mode > 0 && skew != 0) continue;
        auto at = [&](int p) { return mode == 0 ? slab + (size_t)p * (plane + skew) : sep[p]; };
        In in{};
        in.n = n;
        in.pc = (const float4*)at(0); in.cov_a = (const float4*)at(1); in.cov_b = (const float2*)at(2);
        in.sh1 = (const float*)at(3);
        in.sh4 = (const float4*)at(4);
        in.sh_stride = mode == 0 ? (plane + skew) / 16 : (mode == 1 ? plane / 16 : n);
        Out out{nullptr, (float4*)at(16), (float4*)at(17), (float4*)at(18), (uint32_t*)at(19)};
        if (mode) printf("separate hipMalloc per stream, sh stride %zu B: ", in.sh_stride * 16);
        hipLaunchKernelGGL(k_fill, dim3(nchunks), dim3(256), 0, 0, in, (char*)nullptr, n);
        CK(hipDeviceSynchronize());
        std::vector<float> ms;
        for (int r = 0; r < rounds + 2; ++r) {
            if (getenv("GSX_SLEEP_US")) usleep(atoi(getenv("GSX_SLEEP_US")));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_var<0, 0, 1>), dim3(nchunks), dim3(256), 0, 0, f, in, out, nvis);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("placement skew %8zu B: median %.3f ms  min %.3f ms\n", skew, ms[ms.size() / 2], ms[0]);
    }
}

int main(int argc, char** argv) {
    uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000u;
    int rounds = argc > 2 ? atoi(argv[2]) : 15;
    uint32_t nchunks = (n + 255) / 256;
    In in{};
    in.n = n;
    void *pc, *ca, *cb, *sh4, *sh1, *chunked, *rk, *ra, *rb, *rc, *rchunk;
    uint32_t* nvis;
    CK(hipMalloc(&pc, 16ull * n)); CK(hipMalloc(&ca, 16ull * n)); CK(hipMalloc(&cb, 8ull * n));
    CK(hipMalloc(&sh4, 176ull * n)); CK(hipMalloc(&sh1, 4ull * n));
    CK(hipMalloc(&chunked, (size_t)nchunks * kChunkBytes));
    CK(hipMalloc(&rk, 4ull * n)); CK(hipMalloc(&ra, 16ull * n)); CK(hipMalloc(&rb, 16ull * n)); CK(hipMalloc(&rc, 16ull * n));
    CK(hipMalloc(&rchunk, (size_t)nchunks * kRecChunkBytes));
    CK(hipMalloc(&nvis, 4 * (size_t)(nchunks + 1)));
    in.pc = (float4*)pc; in.cov_a = (float4*)ca; in.cov_b = (float2*)cb; in.sh4 = (float4*)sh4; in.sh1 = (float*)sh1;
    in.base = (char*)chunked;
    in.sh_stride = n;
    Out out{(char*)rchunk, (float4*)ra, (float4*)rb, (float4*)rc, (uint32_t*)rk};
    hipLaunchKernelGGL(k_fill, dim3(nchunks), dim3(256), 0, 0, in, (char*)chunked, n);
    CK(hipDeviceSynchronize());

    float eye[3] = {0.0f, 1.5f, -6.0f}, view[16], proj[16] = {0};
    look_at(eye, view);
    float h = 1.0f / std::tan(0.5f * 1.0471975512f), asp = 1920.0f / 1080.0f, zn = 0.1f, zf = 1e4f;
    proj[0] = h / asp; proj[5] = h; proj[10] = zf / (zn - zf); proj[11] = -1; proj[14] = proj[10] * zn;
    gsx_spec_params sp;
    gsx_spec_params_default(&sp);
    FrameConsts f;
    ModelTransform mt;
    frame_consts_setup(view, proj, 1920, 1080, mt, 1.0f, 0, 3, 0, sp, &f);

    if (argc > 3) {
        placement_sweep(n, f, nvis, rounds);
        return 0;
    }
    struct Var { const char* name; void (*launch)(const FrameConsts&, In, Out, uint32_t*, uint32_t); };
#define VAR(L, E, W)                                                                                           \
    Var{"layout=" #L " eager=" #E " lb=" #W, [](const FrameConsts& f, In in, Out out, uint32_t* nv, uint32_t nb) { \
            hipLaunchKernelGGL((k_var<L, E, W>), dim3(nb), dim3(256), 0, 0, f, in, out, nv);                        \
        }}
    std::vector<Var> vars = {VAR(0, 0, 1), VAR(2, 0, 1), VAR(1, 0, 1), VAR(2, 1, 1), VAR(1, 1, 1), VAR(2, 0, 2), VAR(1, 0, 2), VAR(2, 0, 1)};
    std::vector<std::vector<float>> ms(vars.size() + 1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t hv = 0;
    std::vector<uint32_t> nv(vars.size());
    for (int r = 0; r < rounds + 2; ++r) {
        for (size_t v = 0; v < vars.size(); ++v) {
            CK(hipMemset(nvis, 0, 4));
            CK(hipEventRecord(e0));
            vars[v].launch(f, in, out, nvis, nchunks);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r >= 2) ms[v].push_back(t);
            hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, nvis, nchunks);
            CK(hipMemcpy(&hv, nvis, 4, hipMemcpyDeviceToHost));
            nv[v] = hv;
        }
        // streaming-copy ceiling with the same byte counts: read 220 B/G, write 52 B/G
        size_t n_read = (size_t)nchunks * kChunkBytes / 16, n_write = std::min<size_t>(n_read, (size_t)n * 52 / 16);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const float4*)chunked, (float4*)sh4, n_read, n_write);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms[vars.size()].push_back(t);
    }
    {   // same arrays, tight back-to-back loop of variant 0, then with a streaming copy in between
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<float> t;
            for (int r = 0; r < rounds; ++r) {
                if (mode == 1)
                    hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const float4*)chunked, (float4*)rchunk, (size_t)n * 3, (size_t)n * 3);
                if (mode == 2) usleep(2000);
                CK(hipEventRecord(e0));
                vars[0].launch(f, in, out, nvis, nchunks);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float x; CK(hipEventElapsedTime(&x, e0, e1));
                t.push_back(x);
            }
            std::sort(t.begin(), t.end());
            printf("variant 0 on the same arrays, %s: median %.3f ms min %.3f\n",
                   mode == 0 ? "back-to-back" : (mode == 1 ? "after a 480 MB streaming copy" : "after 2 ms idle"), t[t.size() / 2], t[0]);
        }
    }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mn = [](std::vector<float> v) { return *std::min_element(v.begin(), v.end()); };
    printf("N=%u rounds=%d\n", n, rounds);
    for (size_t v = 0; v < vars.size(); ++v) {
        double bytes = (double)n * 220 + (double)nv[v] * 40;
        printf("%-28s nvis=%u  median %.3f ms  min %.3f ms  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)\n", vars[v].name, nv[v],
               med(ms[v]), mn(ms[v]), bytes / med(ms[v]) / 1e6, bytes / med(ms[v]) / 1e6 / 80.0);
    }
    {
        double bytes = (double)n * 272;
        printf("%-28s median %.3f ms -> %.0f GB/s (read 220 B + write 52 B per Gaussian)\n", "streaming copy", med(ms[vars.size()]),
               bytes / med(ms[vars.size()]) / 1e6);
    }
    return 0;
}
