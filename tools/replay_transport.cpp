// tools/replay_transport.cpp — the transport of tools/rank_alone.py's replay phase as native code: the two callbacks a caller hands to
// gsx_viewer_comm_init_custom_v, serving the pieces ONE rank received in a recorded N-rank run as device copies, in the order the
// library asks for them.  No Python runs inside the timed rank.  Not part of the product: a tool.
//   hipcc -O2 -std=c++17 -fPIC -shared tools/replay_transport.cpp -o tools/libreplay_transport.so
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <vector>

namespace {
// One kernel moves every piece of a collective, as RCCL's one kernel per grouped call does (a hipMemcpyAsync per piece was 8 launches
// per collective at world 8: the replayed rank paid ~25 launch gaps per frame that a real node does not).
struct CopyJob {
    const char* src[64];
    char* dst[64];
    unsigned long long bytes[64];
    unsigned int n;
};
// (16-byte words when both ends are 16-byte aligned, else 4-byte words when both are 4-byte aligned — the feedback pieces: a band's
//  words start wherever the band before ended — else bytes.  Round 5: the byte path had been taken for every feedback gather, 39 us
//  for 130 KB at 3840x2160.)
__global__ __launch_bounds__(256) void k_multi_copy(CopyJob job) {
    const unsigned int piece = blockIdx.y;
    if (piece >= job.n) return;
    const unsigned long long nb = job.bytes[piece];
    const char* __restrict__ s = job.src[piece];
    char* __restrict__ d = job.dst[piece];
    const unsigned long long both = reinterpret_cast<unsigned long long>(s) | reinterpret_cast<unsigned long long>(d);
    const unsigned long long first = (unsigned long long)blockIdx.x * 256ull + threadIdx.x, step = (unsigned long long)gridDim.x * 256ull;
    unsigned long long done = 0;
    if (!(both & 15ull)) {
        const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(s);
        uint4* __restrict__ d4 = reinterpret_cast<uint4*>(d);
        for (unsigned long long i = first; i < nb / 16ull; i += step) d4[i] = s4[i];
        done = nb / 16ull * 16ull;
    } else if (!(both & 3ull)) {
        const unsigned int* __restrict__ s1 = reinterpret_cast<const unsigned int*>(s);
        unsigned int* __restrict__ d1 = reinterpret_cast<unsigned int*>(d);
        for (unsigned long long i = first; i < nb / 4ull; i += step) d1[i] = s1[i];
        done = nb / 4ull * 4ull;
    }
    for (unsigned long long i = done + first; i < nb; i += step) d[i] = s[i];
}
int run_job(const CopyJob& job, hipStream_t s) {
    if (!job.n) return 0;
    unsigned long long mx = 0;
    for (unsigned int p = 0; p < job.n; ++p) mx = job.bytes[p] > mx ? job.bytes[p] : mx;
    const unsigned int gx = (unsigned int)((mx / 4ull + 256ull * 8ull - 1ull) / (256ull * 8ull));  // <= 8 words of 4 bytes per thread (2 of 16)
    hipLaunchKernelGGL(k_multi_copy, dim3(gx < 1u ? 1u : (gx > 512u ? 512u : gx), job.n), dim3(256), 0, s, job);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

struct Piece {
    uint64_t off, bytes;
    const void* keep;  // device copy of what the peer sent in the recording (nullptr: this rank's own piece)
};
struct Call {
    int kind;  // 0 all-to-all, 1 gather
    std::vector<Piece> pieces;  // world of them (gather: empty when this rank received nothing)
};
struct Ctx {
    uint32_t world, rank;
    std::vector<Call> calls;
    size_t at = 0;
    uint64_t wire = 0;
    int failed = 0;
};
}  // namespace

extern "C" {

void* rt_create(uint32_t world, uint32_t rank) { return new Ctx{world, rank}; }
void rt_destroy(void* c) { delete static_cast<Ctx*>(c); }
// one recorded collective: n = world pieces (or 0), in peer order
void rt_add_call(void* c, int kind, uint32_t n, const uint64_t* off, const uint64_t* bytes, const uint64_t* keep) {
    Call call;
    call.kind = kind;
    for (uint32_t p = 0; p < n; ++p) call.pieces.push_back(Piece{off[p], bytes[p], reinterpret_cast<const void*>(keep[p])});
    static_cast<Ctx*>(c)->calls.push_back(std::move(call));
}
void rt_rewind(void* c) {
    Ctx* x = static_cast<Ctx*>(c);
    x->at = 0;
    x->wire = 0;
    x->failed = 0;
}
uint64_t rt_wire(void* c) { return static_cast<Ctx*>(c)->wire; }
uint64_t rt_position(void* c) { return static_cast<Ctx*>(c)->at; }
int rt_failed(void* c) { return static_cast<Ctx*>(c)->failed; }

static const Call* next(Ctx* x, int kind) {
    if (x->at >= x->calls.size() || x->calls[x->at].kind != kind) {
        fprintf(stderr, "replay out of step: call %zu of %zu is %s, the library asks for %s\n", x->at, x->calls.size(),
                x->at < x->calls.size() ? (x->calls[x->at].kind ? "a gather" : "an all-to-all") : "past the recording", kind ? "a gather" : "an all-to-all");
        x->failed = 1;
        return nullptr;
    }
    return &x->calls[x->at++];
}

// gsx_comm_all_to_all_v_fn
int rt_all_to_all_v(void* ctx, const void* d_send, const uint64_t* so, const uint64_t* sb, void* d_recv, const uint64_t* ro, const uint64_t* rb,
                    hipStream_t s) {
    Ctx* x = static_cast<Ctx*>(ctx);
    const Call* c = next(x, 0);
    if (!c) return 7;  // GSX_ERR_RCCL
    CopyJob job{};
    for (uint32_t p = 0; p < x->world; ++p) {
        const Piece& pc = c->pieces[p];
        if (pc.off != ro[p] || pc.bytes != rb[p]) {
            fprintf(stderr, "replay: the rank sizes its slots differently from the recording (peer %u: %llu bytes at %llu, recorded %llu at %llu)\n", p,
                    (unsigned long long)rb[p], (unsigned long long)ro[p], (unsigned long long)pc.bytes, (unsigned long long)pc.off);
            x->failed = 1;
            return 7;
        }
        if (!pc.bytes) continue;
        const void* src = p == x->rank ? static_cast<const char*>(d_send) + so[p] : pc.keep;  // its own slot: a device copy, as over RCCL
        job.src[job.n] = static_cast<const char*>(src);
        job.dst[job.n] = static_cast<char*>(d_recv) + pc.off;
        job.bytes[job.n++] = pc.bytes;
        if (p != x->rank) x->wire += sb[p];
    }
    return run_job(job, s);
}

// gsx_comm_gather_v_fn
int rt_gather_v(void* ctx, const void* d_send, uint64_t n, void* d_recv, const uint64_t* ro, const uint64_t* rb, int32_t root, hipStream_t s) {
    Ctx* x = static_cast<Ctx*>(ctx);
    const Call* c = next(x, 1);
    if (!c) return 7;
    CopyJob job{};
    if (root < 0 || (uint32_t)root == x->rank) {
        for (uint32_t p = 0; p < x->world && p < c->pieces.size(); ++p) {
            const Piece& pc = c->pieces[p];
            if (pc.off != ro[p] || pc.bytes != rb[p]) {
                x->failed = 1;
                return 7;
            }
            if (!pc.bytes) continue;
            const void* src = p == x->rank ? d_send : pc.keep;
            char* dst = static_cast<char*>(d_recv) + pc.off;
            if (src == dst) continue;
            job.src[job.n] = static_cast<const char*>(src);
            job.dst[job.n] = dst;
            job.bytes[job.n++] = pc.bytes;
        }
    }
    x->wire += root < 0 ? (uint64_t)(x->world - 1u) * n : ((uint32_t)root == x->rank ? 0u : n);
    return run_job(job, s);
}

}  // extern "C"
