"""Experiment: L independent viewers (own stream, own copy of the scene) on ONE GPU, frame i rendered by viewer i % L.
Does the device overlap the latency-bound tail of one frame with the bandwidth-bound head of the next?  (Each viewer's
speculation then looks L poses back.)  usage: python tools/two_lanes.py [--lanes 2] [--steps 200] [--workload cfg4]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from wgpu_3dgs_viewer_app_amd import camera, parallel, scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=2)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=40)
ap.add_argument("--workload", default="cfg4")
ap.add_argument("--speculative", type=int, default=1)
ap.add_argument("--in-flight", type=int, default=1, help="gsx_render_options.frames_in_flight of each viewer")
args = ap.parse_args()
n, sh, w, h, seed = scene.CONFIGS[args.workload]
g = scene.synthetic_gaussians(n, seed, sh, 0, n)
lanes = []
for _ in range(args.lanes):
    r = parallel.ShardedViewer(device=0, world=1, rank=0, use_dist=False, sh=0, cov3d=0, mode="index", gather="float", overlap_gather=False)
    r.stages.viewer.set_render_options(speculative=args.speculative, frames_in_flight=args.in_flight)
    r.load_shard(g, 0, n)
    r.poll()
    lanes.append(r)
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
for i in range(args.warmup):
    lanes[i % args.lanes].render_frame(orbit[i % 240], (w, h))
for r in lanes:
    r.poll()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.warmup, args.warmup + args.steps):
    lanes[i % args.lanes].render_frame(orbit[i % 240], (w, h))
t_host = time.perf_counter() - t0
for r in lanes:
    r.poll()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"lanes {args.lanes} speculative {args.speculative} {args.workload}: {args.steps / dt:.1f} fps ({1e3 * dt / args.steps:.3f} ms/frame; host enqueue {1e3 * t_host / args.steps:.3f} ms/frame)")
