"""Development: gsx_render_options.host_verify (0: the repair round always enqueued, decided on the device; 1: the host asks for the
verdict of every speculated frame; 2: automatic) against how the host drives the frames — streaming with one / two frames in flight, or
waiting for every frame.  cfg4 orbit, same process, same resident scene.  usage: python tools/ab_host_verify.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd import viewer as viewer_mod  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 240
n, sh, w, h, seed = scene.CONFIGS["cfg4"]
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
g = scene.synthetic_gaussians(n, seed, sh)
v = MultiModelViewer()
v.add_model("m", n)
v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)


def loop(sync, count, start):
    l0 = viewer_mod.launch_count()
    t0 = time.perf_counter()
    for i in range(start, start + count):
        v.update_camera(orbit[i % 240], (w, h))
        v.render_frame(["m"])
        if sync:
            v.poll()
    v.poll()
    return count / (time.perf_counter() - t0), (viewer_mod.launch_count() - l0) / count


for rep in range(2):
    for hv in (2, 0, 1):
        row = []
        for lanes, sync in ((2, False), (1, False), (1, True)):
            v.set_render_options(host_verify=hv, frames_in_flight=lanes)
            loop(sync, 60, 0)
            fps, launches = loop(sync, frames, 60)
            row.append(f"{'sync' if sync else 'stream'} x{lanes}: {fps:7.1f} fps {launches:5.2f} launches")
        print(f"host_verify {hv}: " + " | ".join(row), flush=True)
v.close()
