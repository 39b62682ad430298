// gsx_graph.cpp — the launch trace behind GSX_LAUNCH (gsx_launch.h): records the kernel launches of a frame-level entry point,
// cuts them into segments at every other stream operation, and submits each segment as a cached HIP graph whose nodes are
// patched to this frame's arguments.  No reference counterpart (the reference records a wgpu command encoder per frame,
// src/tab/scene.rs:856-873: the same idea — build the frame's commands, submit once — with wgpu's own validation).
#include <atomic>
#include <cstdlib>

#include "gsx_state.h"

namespace gsx {

thread_local LaunchTrace* t_trace = nullptr;
std::atomic<uint64_t> g_launch_count{0};

namespace {

constexpr size_t kMinGraphNodes = 3;   // shorter segments are launched directly (a graph launch costs the host ~5 us)
constexpr size_t kMaxVariants = 6;     // cached graphs per (entry point, segment position): speculated / plain / repairing ...

std::atomic<int> g_graphs_enabled{-1};  // -1: not decided (GSX_GRAPH, default off: see TraceScope)

struct Launch {
    const void* fn;
    dim3 grid, block;
    uint32_t shmem;
    uint32_t first_param, n_params;  // into Recording::param_off / param_size
};

struct Recording {
    std::vector<Launch> launches;
    std::vector<uint32_t> param_off, param_size;  // offsets into blob, 16-byte aligned
    std::vector<unsigned char> blob;
    void clear() {
        launches.clear();
        param_off.clear();
        param_size.clear();
        blob.clear();
    }
};

struct Variant {
    Recording rec;  // what the instantiated graph currently holds
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> nodes;
    uint64_t last_use = 0;
    void destroy() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr;
        graph = nullptr;
    }
};

}  // namespace

struct LaunchTrace {
    hipStream_t stream = nullptr;
    bool active = false;
    bool broken = false;     // a graph call failed on this viewer: direct launches from now on
    uint32_t scope = 0, ordinal = 0;
    uint64_t tick = 0;
    Recording cur;
    std::map<uint64_t, std::vector<Variant>> cache;  // key: scope << 32 | ordinal
    gsx_launch_stats stats{};
    std::vector<void*> ptrs;  // scratch: kernelParams of one node
    hipError_t first_error = hipSuccess;  // of the scope in progress: a recorded launch that failed when it was finally submitted
};

namespace {

void fill_params(const Recording& r, const Launch& l, std::vector<void*>& ptrs, hipKernelNodeParams* kp) {
    ptrs.resize(std::max<uint32_t>(l.n_params, 1u));
    for (uint32_t i = 0; i < l.n_params; ++i) ptrs[i] = const_cast<unsigned char*>(r.blob.data()) + r.param_off[l.first_param + i];
    *kp = hipKernelNodeParams{};
    kp->func = const_cast<void*>(l.fn);
    kp->gridDim = l.grid;
    kp->blockDim = l.block;
    kp->sharedMemBytes = l.shmem;
    kp->kernelParams = ptrs.data();
    kp->extra = nullptr;
}

void replay(LaunchTrace* t) {
    for (const Launch& l : t->cur.launches) {
        hipKernelNodeParams kp;
        fill_params(t->cur, l, t->ptrs, &kp);
        const hipError_t e = hipLaunchKernel(l.fn, l.grid, l.block, kp.kernelParams, l.shmem, t->stream);
        if (e != hipSuccess && t->first_error == hipSuccess) t->first_error = e;
    }
    t->stats.direct_launches += t->cur.launches.size();
}

bool same_kernels(const Recording& a, const Recording& b) {
    if (a.launches.size() != b.launches.size()) return false;
    for (size_t i = 0; i < a.launches.size(); ++i)
        if (a.launches[i].fn != b.launches[i].fn || a.launches[i].n_params != b.launches[i].n_params) return false;
    return true;
}

bool same_launch(const Recording& a, const Launch& la, const Recording& b, const Launch& lb) {
    if (la.grid.x != lb.grid.x || la.grid.y != lb.grid.y || la.grid.z != lb.grid.z || la.block.x != lb.block.x || la.block.y != lb.block.y ||
        la.block.z != lb.block.z || la.shmem != lb.shmem)
        return false;
    for (uint32_t i = 0; i < la.n_params; ++i) {
        const uint32_t sa = a.param_size[la.first_param + i], sb = b.param_size[lb.first_param + i];
        if (sa != sb || memcmp(a.blob.data() + a.param_off[la.first_param + i], b.blob.data() + b.param_off[lb.first_param + i], sa) != 0) return false;
    }
    return true;
}

bool build(LaunchTrace* t, Variant* v) {
    v->rec = t->cur;
    if (hipGraphCreate(&v->graph, 0) != hipSuccess) return false;
    v->nodes.resize(v->rec.launches.size());
    for (size_t i = 0; i < v->rec.launches.size(); ++i) {
        hipKernelNodeParams kp;
        fill_params(v->rec, v->rec.launches[i], t->ptrs, &kp);
        // a chain: stream order, node by node
        if (hipGraphAddKernelNode(&v->nodes[i], v->graph, i ? &v->nodes[i - 1] : nullptr, i ? 1 : 0, &kp) != hipSuccess) return false;
    }
    if (hipGraphInstantiate(&v->exec, v->graph, nullptr, nullptr, 0) != hipSuccess) return false;
    t->stats.graphs_built += 1;
    return true;
}

// the recorded segment through a graph; false: something failed (the caller replays it directly)
bool submit_graph(LaunchTrace* t) {
    const uint64_t key = (uint64_t)t->scope << 32 | t->ordinal;
    std::vector<Variant>& vs = t->cache[key];
    Variant* hit = nullptr;
    for (Variant& v : vs)
        if (same_kernels(v.rec, t->cur)) {
            hit = &v;
            break;
        }
    if (!hit) {
        if (vs.size() >= kMaxVariants) {  // the least recently used goes
            size_t lru = 0;
            for (size_t i = 1; i < vs.size(); ++i)
                if (vs[i].last_use < vs[lru].last_use) lru = i;
            vs[lru].destroy();
            vs.erase(vs.begin() + (long)lru);
        }
        vs.emplace_back();
        hit = &vs.back();
        if (!build(t, hit)) {
            hit->destroy();
            vs.pop_back();
            return false;
        }
    } else {
        for (size_t i = 0; i < t->cur.launches.size(); ++i) {
            if (same_launch(hit->rec, hit->rec.launches[i], t->cur, t->cur.launches[i])) continue;
            hipKernelNodeParams kp;
            fill_params(t->cur, t->cur.launches[i], t->ptrs, &kp);
            if (hipGraphExecKernelNodeSetParams(hit->exec, hit->nodes[i], &kp) != hipSuccess) {
                (void)hipGetLastError();
                hit->destroy();
                vs.erase(vs.begin() + (hit - vs.data()));
                return false;
            }
            t->stats.nodes_patched += 1;
        }
        hit->rec = t->cur;
    }
    hit->last_use = ++t->tick;
    if (hipGraphLaunch(hit->exec, t->stream) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    t->stats.graph_launches += 1;
    t->stats.graph_nodes += t->cur.launches.size();
    return true;
}

void close_segment(LaunchTrace* t) {
    if (t->cur.launches.empty()) return;
    if (t->broken || t->cur.launches.size() < kMinGraphNodes || !submit_graph(t)) {
        if (t->cur.launches.size() >= kMinGraphNodes && !t->broken) {  // (only reached when submit_graph failed)
            t->broken = true;
            t->stats.broken = 1;
        }
        replay(t);
    }
    t->cur.clear();
    t->ordinal += 1;
}

}  // namespace

bool launch_graphs_enabled() {
    int e = g_graphs_enabled.load(std::memory_order_relaxed);
    if (e < 0) {
        const char* s = getenv("GSX_GRAPH");
        e = (s && *s) ? std::max(0, std::min(2, atoi(s))) : 0;  // 1: record while the stream is busy; 2: record even when it is idle (tests)
        g_graphs_enabled.store(e, std::memory_order_relaxed);
    }
    return e != 0;
}

bool trace_record(const void* fn, dim3 grid, dim3 block, uint32_t shmem, hipStream_t s, void* const* params, const uint32_t* sizes, uint32_t n_params) {
    LaunchTrace* t = t_trace;
    if (!t || !t->active || s != t->stream) return false;
    Recording& r = t->cur;
    Launch l{fn, grid, block, shmem, (uint32_t)r.param_off.size(), n_params};
    for (uint32_t i = 0; i < n_params; ++i) {
        const size_t off = (r.blob.size() + 15u) & ~(size_t)15u;
        r.blob.resize(off + sizes[i]);
        memcpy(r.blob.data() + off, params[i], sizes[i]);
        r.param_off.push_back((uint32_t)off);
        r.param_size.push_back(sizes[i]);
    }
    r.launches.push_back(l);
    return true;
}

void trace_flush() {
    if (t_trace && t_trace->active) close_segment(t_trace);
}

TraceScope::TraceScope(gsx_viewer* v, uint32_t scope_id) {
    if (!v || t_trace || !launch_graphs_enabled() || v->validate) return;  // (a scope inside a scope belongs to the outer one)
    if (!v->trace) v->trace = new LaunchTrace();
    LaunchTrace* t = v->trace;
    if (t->broken) return;
    // Recording buys HOST time (cfg4: 95 -> 42 us inside gsx_render_frame); on the device a real kernel boundary costs the same
    // inside a graph as on a stream (the 3.3 -> 1.75 us of tools/bench_launch.hip is the command processor's rate for EMPTY
    // kernels: a frame's kernels run long enough for the next packet to be fetched meanwhile), and a frame that arrives as one
    // graph launch starts a few microseconds later than one whose first kernel is already queued (profiles/r04_bench.json,
    // launch_graphs: 1607 vs 1633 fps with one frame in flight) — which is why the recording is OFF by default and a switch for
    // hosts whose time is what counts (gsx_debug_set_launch_graphs(1) / GSX_GRAPH=1).  And a graph reaches the device only when
    // the whole segment has been recorded: a host that WAITS for every frame (the app's own protocol) would leave the device idle
    // meanwhile (661 vs 616 us per synchronised frame, tools/graph_probe.py).  So, when switched on: record while the device
    // still has work of this stream queued, submit launch by launch when the stream is idle, the first kernel starts at once.
    if (g_graphs_enabled.load(std::memory_order_relaxed) != 2 && hipStreamQuery(v->stream) == hipSuccess) {
        t->stats.idle_direct_scopes += 1;
        return;
    }
    t->stream = v->stream;
    t->scope = scope_id;
    t->ordinal = 0;
    t->first_error = hipSuccess;
    t->active = true;
    t->cur.clear();
    t_trace = t;
    mine = t;
}

TraceScope::~TraceScope() { (void)finish(); }

gsx_status TraceScope::finish() {
    if (!mine) return GSX_OK;
    LaunchTrace* t = mine;
    mine = nullptr;
    close_segment(t);
    t->active = false;
    t_trace = nullptr;
    if (t->first_error != hipSuccess) {
        const hipError_t e = t->first_error;
        t->first_error = hipSuccess;
        (void)hipGetLastError();
        return fail(GSX_ERR_HIP, "a recorded kernel launch failed when it was submitted: %s", hipGetErrorString(e));
    }
    return GSX_OK;
}

void trace_destroy(LaunchTrace* t) {
    if (!t) return;
    if (t_trace == t) t_trace = nullptr;
    for (auto& kv : t->cache)
        for (Variant& v : kv.second) v.destroy();
    delete t;
}

void trace_stats(const LaunchTrace* t, gsx_launch_stats* out) { *out = t ? t->stats : gsx_launch_stats{}; }

}  // namespace gsx

extern "C" {

uint64_t gsx_debug_launch_count(void) { return gsx::g_launch_count.load(); }
uint64_t gsx_debug_device_bytes(void) { return gsx::g_dev_bytes.load(); }

gsx_status gsx_debug_tile_profile(gsx_viewer* v, uint32_t* out4, uint64_t n_tiles) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    gsx_viewer* l = result_lane(v);
    if (!out4 || !l->tile_profile || l->tile_prof.bytes < 16 * n_tiles)
        return fail(GSX_ERR_INVALID_ARG, "gsx_debug_tile_profile: the viewer was not created under GSX_TILE_PROFILE=1, or no block-list frame yet");
    HIPCHK(gsx::op::MemcpyAsync(out4, l->tile_prof.p, 16 * n_tiles, hipMemcpyDeviceToHost, l->stream));
    HIPCHK(gsx::op::StreamSynchronize(l->stream));
    return GSX_OK;
}

void gsx_debug_set_launch_graphs(int32_t enabled) { gsx::g_graphs_enabled.store(enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled)); }

gsx_status gsx_viewer_launch_stats(gsx_viewer* v, gsx_launch_stats* out, uint32_t reset) {
    if (!v || !out) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_launch_stats: null argument");
    gsx::trace_stats(v->trace, out);
    for (gsx_viewer* l : v->lanes) {  // the lanes' frames belong to the viewer
        gsx_launch_stats ls{};
        gsx::trace_stats(l->trace, &ls);
        out->graph_launches += ls.graph_launches;
        out->graph_nodes += ls.graph_nodes;
        out->nodes_patched += ls.nodes_patched;
        out->direct_launches += ls.direct_launches;
        out->graphs_built += ls.graphs_built;
        out->broken |= ls.broken;
        out->idle_direct_scopes += ls.idle_direct_scopes;
        if (reset && l->trace) l->trace->stats = gsx_launch_stats{};
    }
    if (reset && v->trace) v->trace->stats = gsx_launch_stats{};
    return GSX_OK;
}

}  // extern "C"
