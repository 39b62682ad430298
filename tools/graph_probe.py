"""Development: what do the cached HIP graphs (csrc/gsx_launch.h) do to a cfg4 frame?  Free-running and synchronised loops with
graphs on / off in one process: wall time per frame, host time inside gsx_render_frame, launch statistics.
usage: python tools/graph_probe.py [frames]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, scene, viewer as viewer_mod  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, sh, w, h, seed = scene.CONFIGS[os.environ.get("PROBE_WORKLOAD", "cfg4")]
g = scene.synthetic_gaussians(n, seed, sh)
v = MultiModelViewer()
v.add_model("m", n)
v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
del g
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)


def loop(sync, spec):
    v.set_render_options(speculative=spec)
    for i in range(40):
        v.update_camera(orbit[i % 240], (w, h))
        v.render_frame(["m"])
    v.poll()
    v.launch_stats(reset=True)
    in_call = 0.0
    t0 = time.perf_counter()
    for i in range(40, 40 + frames):
        v.update_camera(orbit[i % 240], (w, h))
        a = time.perf_counter()
        v.render_frame(["m"])
        in_call += time.perf_counter() - a
        if sync:
            v.poll()
    v.poll()
    el = time.perf_counter() - t0
    return dict(us_per_frame=round(1e6 * el / frames, 1), host_us_in_render_frame=round(1e6 * in_call / frames, 1), stats=v.launch_stats())


res = {}
for rep in range(2):
    for graphs in (1, 0):
        viewer_mod.set_launch_graphs(bool(graphs))
        for spec in (1, 0):
            for sync in (0, 1):
                res[f"rep{rep} graphs={graphs} speculative={spec} {'synchronised' if sync else 'free-running'}"] = loop(sync, spec)
for k, r in res.items():
    print(k, json.dumps(r))
