"""Synthetic scenes and the CPU-side ``gs::Gaussian`` record.

The generator is the one BASELINE.md §3 / SURVEY.md §8(d) fix for the benchmark configs:
``numpy.random.Generator(PCG64(1234 + cfg))``; 70 % of the positions uniform in [-4,4]^3, 30 % in 64
clusters (sigma 0.25); log-scales N(-4.0, 0.7) clamped to [-7,-1]; rotations normalised N(0,I)_4;
opacity logits N(0.5, 1.5); f_dc ~ N(0,1); f_rest ~ N(0, 0.15) (zero for SH-0 configs).
The PLY-domain values are then converted exactly as the reference converts a PLY vertex into a
``gs::Gaussian`` before upload (``g.map(gs::Gaussian::from)``, src/app.rs:1064): scale = exp,
opacity = sigmoid, DC colour = clamp(0.5 + SH_C0 * f_dc), both stored as UNORM8.
"""
from __future__ import annotations

import numpy as np

SH_C0 = 0.28209479177387814

#: numpy mirror of ``gsx_gaussian`` (include/gsx.h) = ``gs::Gaussian`` {rot, pos, color, sh, scale}; 224 bytes
GAUSSIAN_DTYPE = np.dtype(
    [("rot", "<f4", (4,)), ("pos", "<f4", (3,)), ("color", "u1", (4,)), ("sh", "<f4", (15, 3)), ("scale", "<f4", (3,))]
)
assert GAUSSIAN_DTYPE.itemsize == 224

#: ``gs::PlyGaussianPod``: x y z nx ny nz f_dc[3] f_rest[45] opacity scale[3] rot[4] = 62 f32 = 248 bytes
PLY_DTYPE = np.dtype(
    [("pos", "<f4", (3,)), ("n", "<f4", (3,)), ("f_dc", "<f4", (3,)), ("f_rest", "<f4", (45,)),
     ("opacity", "<f4"), ("scale", "<f4", (3,)), ("rot", "<f4", (4,))]
)
assert PLY_DTYPE.itemsize == 248


def unorm8(x: np.ndarray) -> np.ndarray:
    """clamp to [0,1] and quantise to 8 bits, round half up."""
    return np.floor(np.clip(x, 0.0, 1.0).astype(np.float32) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)


def gaussians_from_ply(ply: np.ndarray) -> np.ndarray:
    """``gs::Gaussian::from(PlyGaussianPod)`` (src/app.rs:1064): PLY-domain vertex -> render-ready Gaussian.

    rot: PLY stores w,x,y,z -> normalised x,y,z,w;  scale: exp;  colour: clamp(0.5 + C0 f_dc) and
    sigmoid(opacity) as UNORM8;  f_rest: channel-major [3][15] -> coefficient-major [15][3]."""
    n = ply.shape[0]
    g = np.zeros(n, dtype=GAUSSIAN_DTYPE)
    q = ply["rot"].astype(np.float32)
    q = q / np.sqrt(np.sum(q * q, axis=1, keepdims=True, dtype=np.float32))
    g["rot"] = q[:, [1, 2, 3, 0]]
    g["pos"] = ply["pos"]
    g["scale"] = np.exp(ply["scale"].astype(np.float32))
    g["color"][:, :3] = unorm8(np.float32(0.5) + np.float32(SH_C0) * ply["f_dc"])
    g["color"][:, 3] = unorm8(1.0 / (1.0 + np.exp(-ply["opacity"].astype(np.float32))))
    g["sh"] = ply["f_rest"].reshape(n, 3, 15).transpose(0, 2, 1)
    return g


#: scene variants (``synthetic_ply(variant=...)``).  "default" is the benchmark generator of BASELINE.md §3.  The other two exist to
#: put the temporal occlusion speculation where it does NOT shine (bench.py ``robustness``, tests/test_gpu_speculation.py):
#:   "translucent"  the default scene with the opacity logits drawn from N(-5.5, 1.5) instead of N(0.5, 1.5): median opacity 0.004,
#:                  next to no tile of a 1080p frame ever saturates — nothing to speculate on, the viewer must notice and stop trying
#:   "surfaces"     a captured-scene stand-in (the INRIA garden PLY is not in the image): Gaussians ON a handful of large surfaces —
#:                  six planes through the cube and four spheres — flattened along the surface normal, log-scales N(-3, 1.2)
#:                  clamped to [-7, 1] (a heavy tail: a few splats are metres wide and cover hundreds of tiles), opacity logits
#:                  N(2, 1.5) (surfaces are mostly opaque); list entries per visible Gaussian are ~10x the default scene's
VARIANTS = ("default", "translucent", "surfaces")


def _surface_points(rng, m: int, seed: int):
    """positions on six planes / four spheres (which ones: a function of `seed` alone) + the surface normal at each"""
    geo = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 0x5AFE])))
    normals = geo.standard_normal((6, 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    offsets = geo.uniform(-2.5, 2.5, 6).astype(np.float32)
    centres = geo.uniform(-2.5, 2.5, (4, 3)).astype(np.float32)
    radii = geo.uniform(0.6, 1.6, 4).astype(np.float32)
    which = rng.integers(0, 10, size=m)
    pos = np.zeros((m, 3), np.float32)
    nrm = np.zeros((m, 3), np.float32)
    p = rng.uniform(-4.0, 4.0, size=(m, 3)).astype(np.float32)
    d = rng.standard_normal((m, 3), dtype=np.float32)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), np.float32(1e-6))
    for k in range(6):
        sel = which == k
        nk = normals[k]
        pos[sel] = p[sel] - ((p[sel] @ nk) - offsets[k])[:, None] * nk   # projected onto the plane n.x = offset
        nrm[sel] = nk
    for k in range(4):
        sel = which == 6 + k
        pos[sel] = centres[k] + radii[k] * d[sel]
        nrm[sel] = d[sel]
    return pos, nrm


def _quat_z_to(nrm: np.ndarray) -> np.ndarray:
    """w, x, y, z (the PLY's order) of the shortest rotation that takes +z to `nrm`"""
    z = np.array([0.0, 0.0, 1.0], np.float32)
    w = np.float32(1.0) + nrm @ z
    xyz = np.cross(np.broadcast_to(z, nrm.shape), nrm).astype(np.float32)
    flip = w < 1e-6                                   # opposite: any axis perpendicular to z
    xyz[flip] = np.array([1.0, 0.0, 0.0], np.float32)
    w = np.where(flip, np.float32(0.0), w)
    q = np.concatenate([w[:, None], xyz], axis=1).astype(np.float32)
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def _ply_block(n: int, seed: int, sh_degree: int, b: int, centres: np.ndarray, block: int, variant: str = "default") -> np.ndarray:
    lo, hi = b * block, min((b + 1) * block, n)
    m = hi - lo
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, b + 1])))
    v = np.zeros(m, dtype=PLY_DTYPE)
    if variant == "surfaces":
        pos, nrm = _surface_points(rng, m, seed)
        v["pos"] = pos + np.float32(0.01) * rng.standard_normal((m, 3), dtype=np.float32)
        ls = np.clip(np.float32(-3.0) + np.float32(1.2) * rng.standard_normal((m, 3), dtype=np.float32), -7.0, 1.0)
        ls[:, 2] = np.minimum(ls[:, 2], np.float32(-5.0))         # thin along the normal
        v["scale"] = ls
        v["rot"] = _quat_z_to(nrm)
        v["opacity"] = np.float32(2.0) + np.float32(1.5) * rng.standard_normal(m, dtype=np.float32)
        v["f_dc"] = rng.standard_normal((m, 3), dtype=np.float32)
        if sh_degree > 0:
            v["f_rest"] = np.float32(0.15) * rng.standard_normal((m, 45), dtype=np.float32)
        return v
    uniform = rng.uniform(-4.0, 4.0, size=(m, 3)).astype(np.float32)
    which = rng.integers(0, 64, size=m)
    clustered = centres[which] + np.float32(0.25) * rng.standard_normal((m, 3), dtype=np.float32)
    in_cluster = rng.random(m) < 0.3
    v["pos"] = np.where(in_cluster[:, None], clustered, uniform)
    v["scale"] = np.clip(np.float32(-4.0) + np.float32(0.7) * rng.standard_normal((m, 3), dtype=np.float32), -7.0, -1.0)
    v["rot"] = rng.standard_normal((m, 4), dtype=np.float32)
    v["opacity"] = np.float32(-5.5 if variant == "translucent" else 0.5) + np.float32(1.5) * rng.standard_normal(m, dtype=np.float32)
    v["f_dc"] = rng.standard_normal((m, 3), dtype=np.float32)
    if sh_degree > 0:
        v["f_rest"] = np.float32(0.15) * rng.standard_normal((m, 45), dtype=np.float32)
    return v


def synthetic_ply(n: int, seed: int, sh_degree: int = 3, start: int = 0, count: int | None = None,
                  workers: int | None = None, variant: str = "default") -> np.ndarray:
    """PLY-domain synthetic scene.  ``start``/``count`` select a contiguous shard of the
    same scene: the scene is defined in fixed blocks of 65536 Gaussians, block b drawn from
    ``PCG64(SeedSequence([seed, b + 1]))`` and the 64 cluster centres from ``PCG64(seed)``, so a shard
    costs only its own blocks and blocks generate in parallel (numpy releases the GIL)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    if variant not in VARIANTS:
        raise ValueError(f"variant {variant!r}: one of {VARIANTS}")
    if count is None:
        count = n - start
    block = 65536
    centres = np.random.Generator(np.random.PCG64(seed)).uniform(-3.0, 3.0, size=(64, 3)).astype(np.float32)
    out = np.zeros(count, dtype=PLY_DTYPE)
    if count == 0:
        return out
    b0, b1 = start // block, (start + count + block - 1) // block

    def work(b):
        v = _ply_block(n, seed, sh_degree, b, centres, block, variant)
        lo, hi = b * block, min((b + 1) * block, n)
        s, e = max(lo, start), min(hi, start + count)
        out[s - start : e - start] = v[s - lo : e - lo]

    workers = workers or min(32, os.cpu_count() or 1)
    if b1 - b0 == 1 or workers == 1:
        for b in range(b0, b1):
            work(b)
    else:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(work, range(b0, b1)))
    return out


def synthetic_gaussians(n: int, seed: int, sh_degree: int = 3, start: int = 0, count: int | None = None,
                        variant: str = "default") -> np.ndarray:
    """Render-ready ``gs::Gaussian`` array of the synthetic scene (or of the shard [start, start+count))."""
    return gaussians_from_ply(synthetic_ply(n, seed, sh_degree, start, count, variant=variant))


#: benchmark configs of BASELINE.json / BASELINE.md §3: name -> (N, sh_degree, width, height, seed)
CONFIGS = {
    "cfg1": (50_000, 0, 640, 480, 1235),
    "cfg2": (1_000_000, 3, 1920, 1080, 1236),
    "cfg3": (5_800_000, 3, 1920, 1080, 1237),  # garden-sized synthetic (the INRIA PLY is not in the container)
    "cfg4": (10_000_000, 3, 1920, 1080, 1238),
    "cfg5": (24_000_000, 3, 3840, 2160, 1239),  # 4 models x 6 M
}
