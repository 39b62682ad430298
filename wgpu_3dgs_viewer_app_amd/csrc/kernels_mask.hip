// kernels_mask.hip — mask evaluation (the reference's K5: `gs::MaskEvaluator::evaluate`,
// src/tab/scene.rs:2124-2131, 2201-2209) for gfx950.  One Gaussian per lane: world position from the model
// transform, membership in up to 32 box / ellipsoid shapes, then the set-algebra tree evaluated as a
// postfix program on a bit stack held in one register; a wave writes its 64 result bits as one 8-byte store.
// HBM-bound (16 B read per Gaussian, 1 bit written); operation order mirrors oracle/gsx_oracle.c:gsxo_mask_evaluate.
#include "gsx_internal.h"

namespace gsx {

// read once, 160 MB at 10 M Gaussians: non-temporal, like the projection pass's plane loads
__device__ inline float4 mk_ld_stream(const float4* p) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ inline float mk_dot3(float a0, float a1, float a2, float b0, float b1, float b2) { return (a0 * b0 + a1 * b1) + a2 * b2; }

// One Gaussian's verdict: membership in the shapes, then the postfix program (operation order = gsxo_mask_evaluate).
__device__ inline bool mk_keep(const float4 p, const MaskProgram& prog) {
    const float sx = prog.m_scale[0] * p.x, sy = prog.m_scale[1] * p.y, sz = prog.m_scale[2] * p.z;
    const float wx = mk_dot3(prog.m_rot[0], prog.m_rot[1], prog.m_rot[2], sx, sy, sz) + prog.m_pos[0];
    const float wy = mk_dot3(prog.m_rot[3], prog.m_rot[4], prog.m_rot[5], sx, sy, sz) + prog.m_pos[1];
    const float wz = mk_dot3(prog.m_rot[6], prog.m_rot[7], prog.m_rot[8], sx, sy, sz) + prog.m_pos[2];
    uint32_t inside = 0;
    for (uint32_t s = 0; s < prog.n_shapes; ++s) {
        const MaskShapeConsts& sh = prog.shapes[s];
        const float rx = wx - sh.pos[0], ry = wy - sh.pos[1], rz = wz - sh.pos[2];
        // inverse rotation = transpose: local_c = column c of R . rel
        const float d0 = mk_dot3(sh.rot[0], sh.rot[3], sh.rot[6], rx, ry, rz);
        const float d1 = mk_dot3(sh.rot[1], sh.rot[4], sh.rot[7], rx, ry, rz);
        const float d2 = mk_dot3(sh.rot[2], sh.rot[5], sh.rot[8], rx, ry, rz);
        bool in;
        if (sh.kind == GSX_MASK_BOX) {
            // |d / scale| <= 1 without the division — the same verdict bit for bit (mask_box_limit, gsx_internal.h); the kernel is
            // bound by vector issue as much as by HBM, and an IEEE division is ten instructions
            in = fabsf(d0) <= sh.box_lim[0] && fabsf(d1) <= sh.box_lim[1] && fabsf(d2) <= sh.box_lim[2];
        } else {
            const float q0 = d0 / sh.scale[0], q1 = d1 / sh.scale[1], q2 = d2 / sh.scale[2];
            in = (q0 * q0 + q1 * q1) + q2 * q2 <= 1.0f;
        }
        inside |= (in ? 1u : 0u) << s;
    }
    // postfix evaluation, the stack is a bit string (depth <= 32)
    uint32_t stack = 0;
    for (uint32_t k = 0; k < prog.n_ops; ++k) {
        const uint32_t op = prog.ops[k].opcode, arg = prog.ops[k].arg;
        if (op == GSX_MASK_OP_SHAPE) {
            stack = (stack << 1) | ((inside >> arg) & 1u);
        } else if (op == GSX_MASK_OP_COMPLEMENT) {
            stack ^= 1u;
        } else {
            const uint32_t b = stack & 1u, a = (stack >> 1) & 1u;
            uint32_t r = op == GSX_MASK_OP_UNION ? (a | b)
                         : op == GSX_MASK_OP_INTERSECTION ? (a & b)
                         : op == GSX_MASK_OP_DIFFERENCE ? (a & ~b & 1u) : (a ^ b);
            stack = ((stack >> 2) << 1) | r;
        }
    }
    return prog.n_ops == 0 ? true : (stack & 1u);
}

// A workgroup takes 1024 consecutive Gaussians, a lane four of them 256 apart: the four non-temporal 16-byte loads are in
// flight before the first verdict is computed (one load per lane left the kernel waiting on HBM latency: 2.05 TB/s, round 3),
// and a wave's 64 verdicts leave as ONE 8-byte store.
constexpr uint32_t kMaskPerLane = 4;
__global__ __launch_bounds__(256) void k_mask_evaluate(const float4* __restrict__ pc, uint32_t n, MaskProgram prog,
                                                        uint32_t* __restrict__ mask) {
    const uint32_t base = blockIdx.x * (256u * kMaskPerLane) + threadIdx.x;
    float4 p[kMaskPerLane];
#pragma unroll
    for (uint32_t k = 0; k < kMaskPerLane; ++k) {
        const uint32_t i = base + k * 256u;
        p[k] = i < n ? mk_ld_stream(pc + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t k = 0; k < kMaskPerLane; ++k) {
        const uint32_t i = base + k * 256u;
        const bool keep = i < n && mk_keep(p[k], prog);
        const unsigned long long bal = __ballot(keep);
        const uint32_t first = i - lane;  // the wave's first Gaussian: a multiple of 64
        if (lane == 0 && first < n) {
            if (first + 32u < n) *reinterpret_cast<uint2*>(mask + (first >> 5)) = make_uint2((uint32_t)bal, (uint32_t)(bal >> 32));
            else mask[first >> 5] = (uint32_t)bal;  // the model's last word
        }
    }
}

hipError_t launch_mask_evaluate(hipStream_t s, const float4* pc, uint32_t n, const MaskProgram& prog, uint32_t* mask) {
    const uint32_t per = 256u * kMaskPerLane;
    if (n) GSX_LAUNCH(k_mask_evaluate, dim3((n + per - 1) / per), dim3(256), 0, s, pc, n, prog, mask);
    return hipGetLastError();
}

}  // namespace gsx
