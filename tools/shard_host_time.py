"""Dev aid: where a sharded frame's time goes at world 1 over RCCL — host enqueue time per gsx_shard_render_frame call against the
frame time, for 1 / 2 / 3 frames in flight, and the same for gsx_render_frame.  usage: python tools/shard_host_time.py [cfg4]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import _lib, camera, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

n, sh, w, h, seed = scene.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
g = scene.synthetic_gaussians(n, seed, sh)
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
K = 480


def run(v, call, lanes):
    v.set_render_options(frames_in_flight=lanes)
    for i in range(40):
        v.update_camera(orbit[i % 240], (w, h))
        call(v)
    v.poll()
    host = 0.0
    t0 = time.perf_counter()
    for i in range(40, 40 + K):
        v.update_camera(orbit[i % 240], (w, h))
        a = time.perf_counter()
        call(v)
        host += time.perf_counter() - a
    t_enq = time.perf_counter() - t0
    v.poll()
    t_all = time.perf_counter() - t0
    return dict(lanes=lanes, fps=round(K / t_all, 1), ms_frame=round(1e3 * t_all / K, 4), host_ms_in_call=round(1e3 * host / K, 4),
                enqueue_loop_ms=round(1e3 * t_enq / K, 4))


v = MultiModelViewer()
v.add_model("m", n)
v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
for lanes in (() if os.environ.get("NO_SINGLE") else (1, 2, 3)):
    print("single ", run(v, lambda vv: vv.render_frame(["m"]), lanes), flush=True)
v.close()
from wgpu_3dgs_viewer_app_amd.viewer import CommGroup  # noqa: E402

for transport in (os.environ.get("TRANSPORTS", "group,rccl").split(",")):
    v = MultiModelViewer()
    v.add_model("m", n)
    v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
    if transport == "rccl":
        uid = (C.c_uint8 * 128)()
        _lib.check(v._L.gsx_comm_unique_id(uid))
        v.comm_init_rccl(1, 0, bytes(uid))
    else:
        grp = CommGroup(1)
        v.comm_init_group(grp, 0)
    margin, radius = float(os.environ.get("MARGIN", "0.25")), int(os.environ.get("RADIUS", "3"))
    for lanes in (1, 2, 3):
        r = run(v, lambda vv: vv.shard_render_frame("m", n, True, margin, radius), lanes)
        st = v.shard_stats(reset=True)
        r.update(repair_frac=round(st["repair_frames"] / max(st["frames"], 1), 3), verdict_wait_ms_per_frame=round(st["verdict_wait_ns"] / 1e6 / max(st["frames"], 1), 4),
                 slot=st["last_slot_records"])
        print("sharded", transport, r, flush=True)
    v.close()
