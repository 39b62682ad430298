// tools/bench_atomic.hip — what does an atomic on ONE address cost when every workgroup of a launch issues it?
// (development aid; `hipcc -O3 --offload-arch=gfx950 tools/bench_atomic.hip -o tools/bench_atomic`)
// Why: round 4 tried to let the block compositor's 8160 workgroups append themselves to a list with atomicAdd (three atomics per
// workgroup on four adjacent words) and the launch went from 187 to 707 us.  The sort's tile tickets and "last workgroup" counters
// are atomics of the same shape; this measures their rate.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_ret(unsigned* ctr, unsigned* out, unsigned stride_words) {   // returning: the workgroup waits for the old value
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(&ctr[(blockIdx.x % 8u) * stride_words * 0u], 1u);
        if (t == 0xFFFFFFFFu) out[0] = t;
    }
}
__global__ void k_ret_spread(unsigned* ctr, unsigned* out, unsigned ways, unsigned stride_words) {  // `ways` addresses, stride apart
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(&ctr[(blockIdx.x % ways) * stride_words], 1u);
        if (t == 0xFFFFFFFFu) out[0] = t;
    }
}
__global__ void k_noret(unsigned* ctr) {   // fire and forget
    if (threadIdx.x == 0) atomicAdd(ctr, 1u);
}
__global__ void k_none(unsigned* ctr) {
    if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFu) ctr[0] = 1;
}
// a persistent loop that takes tickets, as the onesweep does: `grid` resident workgroups share `tickets` tickets
__global__ void k_ticket_loop(unsigned* ctr, unsigned tickets, unsigned* out, unsigned spin) {
    __shared__ unsigned s_t;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(ctr, 1u);
        __syncthreads();
        const unsigned t = s_t;
        __syncthreads();
        if (t >= tickets) break;
        unsigned acc = t;
        for (unsigned k = 0; k < spin; ++k) acc = acc * 1664525u + 1013904223u;
        if (acc == 0x12345u) out[0] = acc;
    }
}
// the sort histogram's flush: every workgroup adds its 1024 LDS bins to 1024 global bins, `stride_words` apart
__global__ __launch_bounds__(256) void k_flush(unsigned* bins, unsigned stride_words) {
    for (unsigned p = 0; p < 4u; ++p) atomicAdd(&bins[(size_t)(p * 256u + threadIdx.x) * stride_words], blockIdx.x + 1u);
}
template <class F> float run(hipStream_t s, int n, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1000.0f * ms / n;
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *ctr, *out; hipMalloc(&ctr, 1 << 20); hipMemset(ctr, 0, 1 << 20); hipMalloc(&out, 4096);
    const int n = 200;
    for (unsigned grid : {256u, 2048u, 8160u, 32768u}) {
        const float base = run(s, n, [&] { hipLaunchKernelGGL(k_none, dim3(grid), dim3(128), 0, s, ctr); });
        const float nr = run(s, n, [&] { hipLaunchKernelGGL(k_noret, dim3(grid), dim3(128), 0, s, ctr); });
        const float rt = run(s, n, [&] { hipLaunchKernelGGL(k_ret, dim3(grid), dim3(128), 0, s, ctr, out, 0u); });
        printf("grid %6u x 128: no atomic %7.2f us | one address, no return %7.2f us (%5.1f ns each) | one address, returning %7.2f us (%5.1f ns each)\n",
               grid, base, nr, 1000.f * (nr - base) / grid, rt, 1000.f * (rt - base) / grid);
        for (unsigned ways : {8u, 64u}) {
            const float sp = run(s, n, [&] { hipLaunchKernelGGL(k_ret_spread, dim3(grid), dim3(128), 0, s, ctr, out, ways, 64u); });
            printf("                 returning, %2u addresses 256 B apart: %7.2f us (%5.1f ns each)\n", ways, sp, 1000.f * (sp - base) / grid);
        }
    }
    for (unsigned grid : {256u, 768u})
        for (unsigned stride : {1u, 4u, 16u, 64u}) {
            const float t = run(s, n, [&] { hipLaunchKernelGGL(k_flush, dim3(grid), dim3(256), 0, s, ctr, stride); });
            printf("histogram flush: %4u workgroups x 1024 bins, bins %3u B apart: %7.2f us (%.2f ns per atomic)\n", grid, 4 * stride, t,
                   1000.f * t / (grid * 1024.f));
        }
    // ticket loop: 2048 resident workgroups (8 per CU) x 256 threads, like the sort; tickets = tiles of an 8.3 M key pass
    for (unsigned spin : {0u, 2000u, 20000u}) {
        for (unsigned tickets : {2036u, 8144u}) {
            const float t = run(s, 50, [&] {
                hipMemsetAsync(ctr, 0, 4, s);
                hipLaunchKernelGGL(k_ticket_loop, dim3(2048), dim3(256), 0, s, ctr, tickets, out, spin);
            });
            printf("ticket loop, 2048 workgroups, %5u tickets, %5u multiply-adds of work per ticket: %7.2f us\n", tickets, spin, t);
        }
    }
    return 0;
}
