#!/usr/bin/env python3
"""Development: the device-resident exchange on one GPU over real RCCL (world 1), frame by frame — the verdicts (tiles needing
repair, the busiest pair's record count, overflow) and the slot size the library chooses.  usage: tools/slot_debug.py [parts]
(parts: render only the first 1/parts of cfg4, i.e. one rank's shard of a parts-way partition)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, parallel, scene  # noqa: E402

n, sh, w, h, seed = scene.CONFIGS["cfg4"]
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = scene.synthetic_gaussians(n, seed, sh, 0, n // parts)
v = parallel.ShardedViewer(world=1, rank=0, use_dist=True, comm="lib")
v.load_shard(g, 0, n // parts)
v.profile = {}   # the Python-driven path (same stage calls as gsx_shard_render_frame), which keeps the verdict
for i in range(48):
    slot = v.stages.slot_records(v.KEY, 1, n // parts) if i else None
    v.render_frame(camera.orbit_pose(i if i < 40 else 100 + i), (w, h))
    print("frame", i, "slot", slot, "verdict", v.last_verdict, "rounds", v.rounds, flush=True)
v.poll()
print({k: round(1e3 * x / 48, 3) for k, x in v.profile.items()}, "ms per frame by section (with syncs)")
