// tools/bench_launch.hip — what does one dependent kernel boundary cost on this box, and does a HIP graph make it cheaper?
// (development aid; `hipcc -O3 --offload-arch=gfx950 tools/bench_launch.hip -o tools/bench_launch`)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
struct Big { float f[96]; };
__global__ void k_empty(unsigned* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_read(const unsigned* d_n, unsigned* p) { if (*d_n == 12345u && threadIdx.x == 0) p[blockIdx.x] = 1; }
__global__ void k_big(Big b, const unsigned* d_n, unsigned* p) { if (*d_n == 12345u && threadIdx.x == 0) p[blockIdx.x] = (unsigned)b.f[3]; }
// a link of a dependent chain shaped like the small stages of a speculated frame: reads a device-side count the previous
// kernel wrote, touches `work` words per thread, writes the count its successor reads
__global__ void k_link(const unsigned* d_in, unsigned* d_out, unsigned* buf, unsigned work) {
    const unsigned n = *d_in;
    unsigned acc = 0;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned k = 0; k < work; ++k) acc += buf[(i + k * 65536u) & 0xFFFFFu];
    if (acc == 0xDEADBEEFu) buf[i & 0xFFFFFu] = acc;
    if (i == 0) *d_out = n + 1;
}
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
    unsigned *p, *dn, *buf; hipMalloc(&p, 1 << 20); hipMalloc(&dn, 4096); hipMemset(dn, 0, 4096); hipMalloc(&buf, 4 << 20); hipMemset(buf, 0, 4 << 20);
    Big big{};
    const int n = 2000;
    printf("empty 1x64          : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }));
    printf("empty 256x256       : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p); }));
    printf("empty 8160x128      : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_empty, dim3(8160), dim3(128), 0, s, p); }));
    printf("read d_n 1x64       : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_read, dim3(1), dim3(64), 0, s, dn, p); }));
    printf("read d_n 6144x256   : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_read, dim3(6144), dim3(256), 0, s, dn, p); }));
    printf("read d_n 39063x256  : %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_read, dim3(39063), dim3(256), 0, s, dn, p); }));
    printf("384 B kernarg 768x256: %.2f us/launch\n", run(s, n, [&] { hipLaunchKernelGGL(k_big, dim3(768), dim3(256), 0, s, big, dn, p); }));

    // ---- a frame-shaped chain: 40 dependent links (grid sizes of the speculated frame's small stages), stream vs graph ----
    const int links = 40;
    const unsigned grids[8] = {1, 32, 1200, 64, 1, 256, 600, 8};
    auto chain = [&](hipStream_t st) {
        for (int k = 0; k < links; ++k)
            hipLaunchKernelGGL(k_link, dim3(grids[k & 7]), dim3(256), 0, st, dn + (k & 63), dn + ((k + 1) & 63), buf, (k & 3) == 2 ? 4u : 1u);
    };
    for (int sync_each = 0; sync_each < 2; ++sync_each) {
        // sync_each = 1: the host waits for every chain (the synchronised metric: enqueue time is exposed)
        auto timed = [&](auto&& submit) {
            for (int i = 0; i < 10; ++i) { submit(); }
            hipStreamSynchronize(s);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            const int reps = 300;
            hipEventRecord(a, s);
            for (int i = 0; i < reps; ++i) { submit(); if (sync_each) hipStreamSynchronize(s); }
            hipEventRecord(b, s);
            hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, a, b);
            return 1000.0f * ms / reps;
        };
        const float t_stream = timed([&] { chain(s); });
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        chain(s);
        hipStreamEndCapture(s, &g);
        hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (e != hipSuccess) { printf("graph instantiate failed: %s\n", hipGetErrorString(e)); return 1; }
        const float t_graph = timed([&] { hipGraphLaunch(ge, s); });
        // a graph whose kernel parameters are patched before every launch (camera constants change per frame): 40 nodes
        size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
        std::vector<hipGraphNode_t> nodes(nn); hipGraphGetNodes(g, nodes.data(), &nn);
        const float t_graph_patch = timed([&] {
            for (size_t k = 0; k < nn; ++k) {
                hipKernelNodeParams kp{};
                if (hipGraphKernelNodeGetParams(nodes[k], &kp) == hipSuccess) hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp);
            }
            hipGraphLaunch(ge, s);
        });
        // the robust way to keep a graph current: capture the frame's launches again every time (no submission: the host
        // logic runs as it is), let hipGraphExecUpdate carry the new parameters into the instantiated graph, launch that
        int update_failures = 0;
        const float t_recapture = timed([&] {
            hipGraph_t g2 = nullptr;
            hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
            chain(s);
            hipStreamEndCapture(s, &g2);
            hipGraphNode_t bad = nullptr; hipGraphExecUpdateResult res;
            if (hipGraphExecUpdate(ge, g2, &bad, &res) != hipSuccess) update_failures += 1;
            hipGraphDestroy(g2);
            hipGraphLaunch(ge, s);
        });
        // host cost alone of the three ways to submit one chain (no device wait inside the loop)
        auto host_us = [&](auto&& submit) {
            hipStreamSynchronize(s);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 50; ++i) submit();
            const auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(s);
            return std::chrono::duration<double, std::micro>(t1 - t0).count() / 50.0;
        };
        if (!sync_each) {
            const double h_stream = host_us([&] { chain(s); });
            const double h_graph = host_us([&] { hipGraphLaunch(ge, s); });
            const double h_recap = host_us([&] {
                hipGraph_t g2 = nullptr;
                hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
                chain(s);
                hipStreamEndCapture(s, &g2);
                hipGraphNode_t bad = nullptr; hipGraphExecUpdateResult res;
                (void)hipGraphExecUpdate(ge, g2, &bad, &res);
                hipGraphDestroy(g2);
                hipGraphLaunch(ge, s);
            });
            printf("host time per chain (50 chains enqueued back to back): stream %.1f us, graph launch %.1f us, capture + update + launch %.1f us\n", h_stream, h_graph, h_recap);
        }
        printf("  re-captured + hipGraphExecUpdate every chain: %.1f us (%d update failures)\n", t_recapture, update_failures);
        printf("chain of %d dependent links, %s: stream %.1f us (%.2f per link), graph %.1f us (%.2f), graph + %zu node patches %.1f us\n", links,
               sync_each ? "host waits for every chain" : "free running", t_stream, t_stream / links, t_graph, t_graph / links, nn, t_graph_patch);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
