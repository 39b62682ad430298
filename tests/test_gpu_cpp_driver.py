"""GPU: the C++ gs:: facade (include/gsx.hpp) driven by tools/frame_driver.cpp replays the app's frame protocol;
the same LCG scene pushed through the Python mirror must give the same frame (to the ulps of the two host-side camera
restatements), the same cull counts, paint order and selection."""
import os
import re
import subprocess

import numpy as np
import pytest

from wgpu_3dgs_viewer_app_amd import camera
from wgpu_3dgs_viewer_app_amd.scene import GAUSSIAN_DTYPE
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lcg_scene(n, seed):
    s = np.uint32(seed)
    vals = np.empty(n * 59, np.uint32)
    state = int(seed)
    for i in range(vals.size):  # the driver's generator, value by value
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[i] = state
    v = vals.reshape(n, 59)
    u = (v >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
    g = np.zeros(n, GAUSSIAN_DTYPE)
    q = u[:, 0:4] * np.float32(2) - np.float32(1)
    l = np.sqrt(((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3], dtype=np.float32)
    g["rot"] = q / l[:, None]
    g["pos"] = u[:, 4:7] * np.float32(6) - np.float32(3)
    g["color"] = (v[:, 7:11] >> 24).astype(np.uint8)
    g["sh"] = ((u[:, 11:56] - np.float32(0.5)) * np.float32(0.3)).reshape(n, 15, 3)
    g["scale"] = np.float32(0.02) + np.float32(0.2) * u[:, 56:59]
    return g


def fnv1a(data: bytes) -> int:
    h = 1469598103934665603
    for chunk in np.frombuffer(data, np.uint8):
        h = ((h ^ int(chunk)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_facade_matches_python_mirror(tmp_path):
    exe = os.path.join(ROOT, "tools", "frame_driver")
    assert os.path.exists(exe), "tools/frame_driver missing: run __graft_entry__.build()"
    n = 3000
    dump = str(tmp_path / "fb.bin")
    out = subprocess.run([exe, str(n), dump], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    m = re.search(r"vis_a=(\d+) n_b=(\d+) vis_b=(\d+) order=(\w),(\w) fnv=([0-9a-f]{16})", out.stdout)
    assert m, out.stdout
    a, b = lcg_scene(n, 12345), lcg_scene(n // 2, 777)
    w, h = 320, 200
    cam = camera.CameraOrbitControl(pos=np.array([2.0, 1.5, -6.0], np.float32))
    mtb = camera.ModelTransform(pos=np.array([0.5, 0.2, 2.0], np.float32), rot=np.array([10, 30, -20], np.float32),
                                scale=np.array([1.1, 0.9, 1.0], np.float32))
    with MultiModelViewer() as v:
        for key, g in (("a", a), ("b", b)):
            v.add_model(key, g.shape[0])
            v.models[key].gaussian_buffers.gaussians_buffer.update_range(0, g)
        v.update_camera(cam, (w, h))
        v.update_model_transform("a", (0, 0, 0), (0, 0, 0, 1), (1, 1, 1))
        # same float32 quaternion as the C++ facade's quat_from_euler_zyx
        d2r = np.float32(0.017453292519943295)
        v.update_model_transform("b", mtb.pos, mtb.quat(), mtb.scale)
        v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
        for key in ("a", "b"):
            v.preprocessor.preprocess(key)
            v.radix_sorter.sort(key)
        v.poll()
        keys = camera.model_render_order(cam.pos, {"a": np.zeros(3), "b": mtb.pos})
        v.renderer.render(keys)
        fb = v.download_framebuffer()
        va, vb = v.frame_stats("a")["n_visible"], v.frame_stats("b")["n_visible"]
        # the second frame of the driver: rect selection through the same protocol
        from wgpu_3dgs_viewer_app_amd import query

        v.update_query(query.QueryPod.rect((60.0, 40.0), (250.0, 160.0), query.QuerySelectionOp.Set))
        v.update_selection_highlight((1.0, 0.0, 1.0, 0.5))
        for key in ("a", "b"):
            v.preprocessor.preprocess(key)
            v.radix_sorter.sort(key)
        v.renderer.render(keys)
        for key in ("a", "b"):
            v.postprocessor.postprocess(key)
        v.poll()
        selected = sum(int(np.unpackbits(v.models[k].gaussian_buffers.selection_buffer.download().view(np.uint8)).sum()) for k in ("a", "b"))
    assert [m.group(4), m.group(5)] == keys
    assert (int(m.group(1)), int(m.group(3))) == (va, vb)
    cpp = np.fromfile(dump, np.float32).reshape(h, w, 4)
    assert int(m.group(6), 16) == fnv1a(cpp.tobytes())
    # glam's look_at_rh / perspective_rh / from_euler are restated twice on the host (C++ float32 vs numpy float32, different
    # operation order): the matrices agree to an ulp or two, so do the frames — not bit for bit
    err = float(np.abs(cpp - fb).max())
    assert err <= 2e-4, f"the C++ facade's frame differs from the Python mirror's: L-inf {err}"
    ms = re.search(r"selected=(\d+)", out.stdout)
    assert ms and int(ms.group(1)) == selected and selected > 50, (out.stdout, selected)
    assert "frame_driver clones=ok" in out.stdout, out.stdout   # cloned buffer handles downloaded on spawned threads (app.rs:769-816)
    # camera matrices / the model quaternion are float32 on both sides but come from different libm calls (last-bit
    # differences), so the frames agree to rounding rather than bit-for-bit
    assert np.abs(cpp - fb).max() <= 2e-4
