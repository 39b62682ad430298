"""SURVEY 8 (f) rows, measured: what the callers either side of the render path cost on cfg4-sized data (10 M Gaussians).
  upload      gsx_model_upload_range (H2D + k_convert), PCIe-inclusive, three pod kinds
  mask        gsx_mask_evaluate ('0 - 1', box minus ellipsoid): one pass over the positions
  query       a rect query riding on the projection pass + gsx_postprocess (selection Set)
  edit frame  a frame while a selection edit is active (k_edit_prepare / k_edit_apply, unlazy projection)
  PLY         gsx_ply_write (one thread) / gsx_ply_read_gaussians (one call for the file: several host threads) on the host, 1 M Gaussians
One JSON line.  usage: python tools/bench_rows.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, query, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind  # noqa: E402
from wgpu_3dgs_viewer_app_amd.ply import Gaussians  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

n, sh, w, h, seed = scene.CONFIGS["cfg4"]
g = scene.synthetic_gaussians(n, seed, sh)
res = {"n_gaussians": n}


def timed(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t))


for name, (shk, covk) in {"single/single": (0, 0), "norm8/half": (2, 1)}.items():
    v = MultiModelViewer(sh=shk, cov3d=covk)
    v.add_model("m", n)
    buf = v.models["m"].gaussian_buffers.gaussians_buffer
    s = timed(lambda: (buf.update_range(0, g), v.poll()), reps=2)
    res[f"upload_GBps_pcie_inclusive[{name}]"] = round(g.nbytes / s / 1e9, 1)
    if name != "single/single":
        v.close()
        continue
    shapes = [MaskShape(MaskShapeKind.Box, pos=np.zeros(3, np.float32), scale=np.array([3.0, 3.0, 3.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.zeros(3, np.float32), scale=np.array([1.5, 1.5, 1.5], np.float32))]
    ev = MaskEvaluator(v)
    s = timed(lambda: (ev.evaluate(MaskOp.parse("0 - 1"), "m", shapes), v.poll()))
    res["mask_evaluate_call_and_wait_us"] = round(1e6 * s, 1)   # one call + the host's wait for it (launch latency and Python included)
    prog = MaskOp.parse("0 - 1")
    s = timed(lambda: ([ev.evaluate(prog, "m", shapes) for _ in range(50)], v.poll())) / 50.0   # 50 calls enqueued back to back: the kernel
    res["mask_evaluate_us"] = round(1e6 * s, 1)
    res["mask_evaluate_GBps_of_positions"] = round(16.0 * n / s / 1e9, 1)
    ev.evaluate(None, "m")
    cam = camera.orbit_pose(10)

    def frame():
        v.update_camera(cam, (w, h))
        v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
        v.render_frame(["m"])
        v.postprocessor.postprocess("m")
        v.poll()

    base = timed(frame, reps=10)
    v.update_query(query.QueryPod.rect((600.0, 300.0), (1300.0, 800.0), query.QuerySelectionOp.Set))
    res["frame_with_rect_query_and_postprocess_ms"] = round(1e3 * timed(frame, reps=10), 3)
    v.update_query(query.QueryPod.none())
    v.update_selection_edit_with_pod(query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.5, 1.0, 1.2), 0.1, 0.2, 1.0, 0.9))
    v.update_selection_highlight((1.0, 0.5, 0.0, 0.3))
    res["frame_with_selection_edit_and_highlight_ms"] = round(1e3 * timed(frame, reps=10), 3)
    res["frame_plain_synchronised_ms"] = round(1e3 * base, 3)
    sel = v.models["m"].gaussian_buffers.selection_buffer.download()
    res["selected_gaussians"] = int(np.unpackbits(sel.view(np.uint8)).sum())
    v.close()

g1 = Gaussians(g[:1_000_000])
t0 = time.perf_counter()
blob = g1.write_ply_array()
res["ply_write_MBps_host"] = round(blob.nbytes / (time.perf_counter() - t0) / 1e6, 0)
t0 = time.perf_counter()
back = Gaussians.read_ply(blob)
res["ply_read_MBps_host"] = round(blob.nbytes / (time.perf_counter() - t0) / 1e6, 0)
assert back.gaussians.shape[0] == 1_000_000
print(json.dumps(res))
