import sys, time, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import torch
from wgpu_3dgs_viewer_app_amd import camera, parallel, scene
n, sh, w, h, seed = scene.CONFIGS["cfg3"]
g = scene.synthetic_gaussians(n, seed, sh, 0, n)
r = parallel.ShardedViewer(device=0, world=1, rank=0, use_dist=False, sh=0, cov3d=0, mode="index", gather="float", overlap_gather=False)
r.stages.viewer.set_render_options(frames_in_flight=2)
r.load_shard(g, 0, n); r.poll()
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
for rep in range(5):
    torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]; t0 = time.perf_counter()
    for i in range(3000):
        r.render_frame(orbit[i % 240], (w, h))
    r.poll(); torch.cuda.synchronize()
    print(rep, round(3000 / (time.perf_counter() - t0), 1), "fps; free GPU memory change MB:", (torch.cuda.mem_get_info()[0] - free0) / 1e6, flush=True)
