// tools/bench_launch.hip — what does one dependent kernel boundary cost on this box?  (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float f[96]; };
__global__ void k_empty(unsigned* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_read(const unsigned* d_n, unsigned* p) { if (*d_n == 12345u && threadIdx.x == 0) p[blockIdx.x] = 1; }
__global__ void k_big(Big b, const unsigned* d_n, unsigned* p) { if (*d_n == 12345u && threadIdx.x == 0) p[blockIdx.x] = (unsigned)b.f[3]; }
template <class F> float run(hipStream_t s, int n, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
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
    unsigned *p, *dn; hipMalloc(&p, 1 << 20); hipMalloc(&dn, 4); hipMemset(dn, 0, 4);
    Big big{};
    const int n = 2000;
    printf("empty 1x64          : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }));
    printf("empty 256x256       : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p); }));
    printf("empty 8160x128      : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_empty, dim3(8160), dim3(128), 0, s, p); }));
    printf("read d_n 1x64       : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_read, dim3(1), dim3(64), 0, s, dn, p); }));
    printf("read d_n 6144x256   : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_read, dim3(6144), dim3(256), 0, s, dn, p); }));
    printf("read d_n 39063x256  : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_read, dim3(39063), dim3(256), 0, s, dn, p); }));
    printf("384 B kernarg 768x256: %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_big, dim3(768), dim3(256), 0, s, big, dn, p); }));
    return 0;
}
