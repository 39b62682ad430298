for m in 0.5 0.35 0.25; do
python - <<PY
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from wgpu_3dgs_viewer_app_amd import camera, parallel, scene
n, sh, w, h, seed = scene.CONFIGS["cfg4"]
g = scene.synthetic_gaussians(n, seed, sh)
v = parallel.ShardedViewer(world=1, rank=0, use_dist=True, comm="lib")
v.margin = $m
v.load_shard(g, 0, n)
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
for i in range(12): v.render_frame(orbit[i], (w, h))
v.poll(); t0 = time.perf_counter()
for i in range(12, 132): v.render_frame(orbit[i], (w, h))
v.poll(); dt = time.perf_counter() - t0
print("margin", $m, "fps", round(120 / dt, 1), flush=True)
v.close()
PY
done
