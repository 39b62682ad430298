"""BASELINE.json configs[4] on ONE GPU: 4 models x 6 M Gaussians (24 M, SH-3), each with its own TRS, a `0 - 1` mask on one,
a stored rect selection with an HSV edit on another, 3840x2160, models layered far -> near by camera distance every frame
(scene.rs:533-558).  Prints one JSON line: frames/s of the default schedule (1 and 2 frames in flight; the edited model
keeps every frame on the viewer itself, so the second number shows what overlapping is NOT allowed to do here) and
unspeculated.  usage: python tools/bench_cfg5.py [--steps 60]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, parallel, query, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--edit", type=int, default=1, help="0: no selection / edit (every frame may overlap)")
ap.add_argument("--shard", type=int, default=0, help="1: every frame through gsx_shard_render_frame_keys (one rank over the in-process "
                "group transport: the whole exchange protocol per model, nothing on a link)")
ap.add_argument("--only-default", action="store_true", help="the default schedule alone (for a kernel trace)")
args = ap.parse_args()
n_total, sh, w, h, seed = scene.CONFIGS["cfg5"]
n = n_total // 4
tr = {"a": camera.ModelTransform(pos=np.array([0.0, 0.0, 2.5], np.float32)),
      "b": camera.ModelTransform(pos=np.array([2.0, 0.2, -1.0], np.float32), rot=np.array([0, 35, 0], np.float32)),
      "c": camera.ModelTransform(pos=np.array([-2.5, -0.1, -0.5], np.float32), scale=np.array([0.9, 0.9, 0.9], np.float32)),
      "d": camera.ModelTransform(pos=np.array([0.3, -0.2, 0.5], np.float32), rot=np.array([20, -35, 50], np.float32),
                                 scale=np.array([1.2, 0.9, 1.1], np.float32))}
v = MultiModelViewer()
for i, k in enumerate(tr):
    g = scene.synthetic_gaussians(n, seed + i, sh)
    v.add_model(k, n)
    v.models[k].gaussian_buffers.gaussians_buffer.update_range(0, g)
    v.update_model_transform(k, tr[k].pos, tr[k].quat(), tr[k].scale)
    del g
shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([3.0, 3.0, 3.0], np.float32)),
          MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([1.5, 1.5, 1.5], np.float32))]
MaskEvaluator(v).evaluate(MaskOp.parse("0 - 1"), "a", shapes)
orbit = [camera.orbit_pose(k) for k in range(240)]
keys_of = [parallel.model_render_keys(c.pos, tr) for c in orbit]


if args.shard:
    from wgpu_3dgs_viewer_app_amd.viewer import CommGroup  # noqa: E402

    group = CommGroup(1)
    v.comm_init_group(group, 0)


def frame(i):
    v.update_camera(orbit[i % 240], (w, h))
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
    if args.shard:
        v.shard_render_frame_keys(keys_of[i % 240], [n] * 4)
    else:
        v.render_frame(keys_of[i % 240])


if args.edit:  # a rect selection on the frame of pose 0, then an HSV edit of what it selected (stored: GaussianEditPod per Gaussian)
    v.update_query(query.QueryPod.rect((1200.0, 600.0), (2600.0, 1500.0), query.QuerySelectionOp.Set))
    frame(0)
    for k in keys_of[0]:
        v.postprocessor.postprocess(k)
    v.update_query(query.QueryPod.none())
    v.update_selection_edit_with_pod(query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.5, 1.0, 1.2), 0.1, 0.2, 1.0, 0.9))


def loop(**opts):
    v.set_render_options(**opts)
    for i in range(args.warmup):
        frame(i)
    v.poll()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        frame(i)
    v.poll()
    return args.steps / (time.perf_counter() - t0)


res = {"path": "gsx_shard_render_frame_keys, world 1 (in-process group)" if args.shard else "gsx_render_frame",
       "workload": f"cfg5: 4 x {n} Gaussians SH-3, {w}x{h}, TRS per model, mask '0 - 1' on one"
                   + (", stored selection + HSV edit on the models the rectangle hit" if args.edit else ""),
       "fps_default_schedule": round(loop(), 1), "steps": args.steps}
if not args.only_default:
    res["fps_two_frames_in_flight"] = round(loop(frames_in_flight=2), 1)
    res["fps_unspeculated"] = round(loop(speculative=0), 1)
st = {k: v.frame_stats(k) for k in tr}
res["n_visible"] = int(sum(s["n_visible"] for s in st.values()))
res["overflow_slabs"] = int(sum(s["overflow_slabs"] for s in st.values()))
if args.shard:
    res["shard_stats"] = v.shard_stats()
print(json.dumps(res))
v.close()
