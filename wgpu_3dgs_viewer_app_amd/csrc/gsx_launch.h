// gsx_launch.h — every kernel launch of libgsx goes through GSX_LAUNCH.  Build-internal.
//
// Why: a speculated frame is ~40 DEPENDENT launches, most of them a few microseconds of work, and on this part a kernel boundary
// on a stream costs 3.3 us where the same boundary inside a HIP graph costs 1.75 (tools/bench_launch.hip,
// profiles/r04_bench_launch.txt: a 40-link chain 131 us as stream launches, 70 us as a graph).  The frame's launch sequence is
// the same from frame to frame and every branch is taken on the device, so the frame can be a graph — but the host logic around
// the launches (tuner, slab plan, buffer growth, options, edits ...) decides arguments every frame, and a cached graph that is
// merely TRUSTED to still be right is a silent-corruption machine.  So nothing is trusted:
//
//   * while a LaunchTrace is active on the calling thread (TraceScope, one per frame-level entry point), GSX_LAUNCH on the
//     trace's stream does not submit: it records {kernel, grid, block, arguments by value};
//   * any other operation on that stream (copy, memset, event, host wait — trace_flush() in front of it) and the end of the
//     scope close the SEGMENT recorded so far: the segment is compared with the cached graph of the same position in the same
//     entry point — same kernels in the same order? — every node whose grid or argument bytes differ is patched
//     (hipGraphExecKernelNodeSetParams, ~1 us; camera constants and sort epochs change every frame: ~15 nodes) and the graph is
//     launched; another kernel sequence instantiates (and caches) another graph; segments of fewer than kMinGraphNodes launches
//     are simply launched.
//
// Every frame therefore executes exactly the launches its host logic asked for, with exactly those arguments; the graph only
// changes how they reach the device.  The host pays one pass of its own logic + memcmp + a handful of patches + one
// hipGraphLaunch instead of ~30 hipLaunchKernel calls (95 -> 42 us per cfg4 frame).
//
// What round 4 measured next (bench.py `launch_graphs`, tools/graph_probe.py): that is ALL it buys.  A real kernel boundary costs the
// same inside a graph as on a stream — the 3.3 -> 1.75 us above is the command processor's rate for EMPTY kernels; a frame's
// kernels run long enough for the next packet to be fetched meanwhile — and a frame that arrives as one graph launch starts a few
// microseconds later than one whose first kernel is already queued: 1607 vs 1633 frames/s with one frame in flight.  So the recording
// is OFF by default (every GSX_LAUNCH submits at once) and a switch for hosts whose time is what counts:
// gsx_debug_set_launch_graphs(1) / GSX_GRAPH=1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <tuple>
#include <type_traits>
#include <utility>

namespace gsx {

struct LaunchTrace;
#ifdef GSX_LAUNCH_STANDALONE  // tools/*.hip compile a kernel file as source into a small driver: no trace, no library behind it
inline thread_local LaunchTrace* t_trace = nullptr;
inline std::atomic<uint64_t> g_launch_count{0};
inline bool trace_record(const void*, dim3, dim3, uint32_t, hipStream_t, void* const*, const uint32_t*, uint32_t) { return false; }
inline void trace_flush() {}
#else
extern thread_local LaunchTrace* t_trace;  // the calling thread's active trace (nullptr: launches submit at once)
// every launch (recorded or submitted at once), process-wide: bench.py prints launches per frame from it
extern std::atomic<uint64_t> g_launch_count;

// records one launch if `s` is the active trace's stream; false: not recorded, the caller submits it
bool trace_record(const void* fn, dim3 grid, dim3 block, uint32_t shmem, hipStream_t s, void* const* params, const uint32_t* sizes, uint32_t n_params);
// closes the segment recorded so far on the calling thread's trace (no trace: nothing).  In front of every non-kernel operation
// on a stream a trace may be recording for, and of everything that makes the host wait for the device.
void trace_flush();
#endif

template <class... KArgs, class... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, uint32_t shmem, hipStream_t s, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "GSX_LAUNCH: argument count does not match the kernel's parameters");
    std::tuple<std::decay_t<KArgs>...> params(std::forward<Args>(args)...);  // converted to the kernel's parameter types, by value
    void* ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
    uint32_t sizes[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
    {
        size_t i = 0;
        std::apply([&](auto&... p) { ((ptrs[i] = const_cast<void*>(static_cast<const void*>(&p)), sizes[i] = (uint32_t)sizeof(p), ++i), ...); }, params);
    }
    g_launch_count.fetch_add(1, std::memory_order_relaxed);
    if (t_trace && trace_record(reinterpret_cast<const void*>(kernel), grid, block, shmem, s, ptrs, sizes, (uint32_t)sizeof...(KArgs))) return;
    (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, shmem, s);
}

// Everything else the library does to a stream, and everything that makes the host wait for the device, closes the segment a
// trace may be recording first (trace_flush is a thread-local load when nothing records).  tests/test_oracle_cpu.py greps
// csrc/ for raw calls of these.
namespace op {
inline hipError_t MemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st) { trace_flush(); return hipMemcpyAsync(d, s, n, k, st); }
inline hipError_t Memcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { trace_flush(); return hipMemcpy(d, s, n, k); }
inline hipError_t MemsetAsync(void* d, int v, size_t n, hipStream_t st) { trace_flush(); return hipMemsetAsync(d, v, n, st); }
inline hipError_t Memset(void* d, int v, size_t n) { trace_flush(); return hipMemset(d, v, n); }
inline hipError_t MemsetD32Async(hipDeviceptr_t d, int v, size_t n, hipStream_t st) { trace_flush(); return hipMemsetD32Async(d, v, n, st); }
inline hipError_t EventRecord(hipEvent_t e, hipStream_t st) { trace_flush(); return hipEventRecord(e, st); }
inline hipError_t StreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned flags) { trace_flush(); return hipStreamWaitEvent(st, e, flags); }
inline hipError_t StreamSynchronize(hipStream_t st) { trace_flush(); return hipStreamSynchronize(st); }
inline hipError_t StreamQuery(hipStream_t st) { trace_flush(); return hipStreamQuery(st); }
inline hipError_t EventSynchronize(hipEvent_t e) { trace_flush(); return hipEventSynchronize(e); }
inline hipError_t Free(void* p) { trace_flush(); return hipFree(p); }
}  // namespace op

}  // namespace gsx

// GSX_LAUNCH(kernel, grid, block, shared_bytes, stream, kernel arguments...)
#define GSX_LAUNCH(...) ::gsx::launch(__VA_ARGS__)
