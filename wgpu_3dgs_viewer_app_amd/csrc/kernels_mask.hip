// kernels_mask.hip — mask evaluation (the reference's K5: `gs::MaskEvaluator::evaluate`,
// src/tab/scene.rs:2124-2131, 2201-2209) for gfx950.  One Gaussian per lane: world position from the model
// transform, membership in up to 32 box / ellipsoid shapes, then the set-algebra tree evaluated as a
// postfix program on a bit stack held in one register; a wave writes its 64 result bits as two words.
// HBM-bound (16 B read per Gaussian, 1 bit written); operation order mirrors oracle/gsx_oracle.c:gsxo_mask_evaluate.
#include "gsx_internal.h"

namespace gsx {

__device__ inline float mk_dot3(float a0, float a1, float a2, float b0, float b1, float b2) { return (a0 * b0 + a1 * b1) + a2 * b2; }

__global__ __launch_bounds__(256) void k_mask_evaluate(const float4* __restrict__ pc, uint32_t n, MaskProgram prog,
                                                        uint32_t* __restrict__ mask) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const float4 p = pc[i];
        const float sx = prog.m_scale[0] * p.x, sy = prog.m_scale[1] * p.y, sz = prog.m_scale[2] * p.z;
        const float wx = mk_dot3(prog.m_rot[0], prog.m_rot[1], prog.m_rot[2], sx, sy, sz) + prog.m_pos[0];
        const float wy = mk_dot3(prog.m_rot[3], prog.m_rot[4], prog.m_rot[5], sx, sy, sz) + prog.m_pos[1];
        const float wz = mk_dot3(prog.m_rot[6], prog.m_rot[7], prog.m_rot[8], sx, sy, sz) + prog.m_pos[2];
        uint32_t inside = 0;
        for (uint32_t s = 0; s < prog.n_shapes; ++s) {
            const MaskShapeConsts& sh = prog.shapes[s];
            const float rx = wx - sh.pos[0], ry = wy - sh.pos[1], rz = wz - sh.pos[2];
            // inverse rotation = transpose: local_c = column c of R . rel
            const float q0 = mk_dot3(sh.rot[0], sh.rot[3], sh.rot[6], rx, ry, rz) / sh.scale[0];
            const float q1 = mk_dot3(sh.rot[1], sh.rot[4], sh.rot[7], rx, ry, rz) / sh.scale[1];
            const float q2 = mk_dot3(sh.rot[2], sh.rot[5], sh.rot[8], rx, ry, rz) / sh.scale[2];
            bool in = sh.kind == GSX_MASK_BOX ? (fabsf(q0) <= 1.0f && fabsf(q1) <= 1.0f && fabsf(q2) <= 1.0f)
                                              : ((q0 * q0 + q1 * q1) + q2 * q2 <= 1.0f);
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
        keep = prog.n_ops == 0 ? true : (stack & 1u);
    }
    const unsigned long long bal = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t word = (blockIdx.x * 256u + (threadIdx.x & ~63u)) >> 5;
    if (lane == 0 && (blockIdx.x * 256u + (threadIdx.x & ~63u)) < n) mask[word] = (uint32_t)bal;
    if (lane == 32 && (blockIdx.x * 256u + (threadIdx.x & ~63u) + 32u) < n) mask[word + 1] = (uint32_t)(bal >> 32);
}

hipError_t launch_mask_evaluate(hipStream_t s, const float4* pc, uint32_t n, const MaskProgram& prog, uint32_t* mask) {
    if (n) hipLaunchKernelGGL(k_mask_evaluate, dim3((n + 255) / 256), dim3(256), 0, s, pc, n, prog, mask);
    return hipGetLastError();
}

}  // namespace gsx
