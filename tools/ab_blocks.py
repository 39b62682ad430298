"""Development: block-list granularity against the long-list scenes.  For each scene (cfg4 orbit, open sky, translucent) and each
GSX_BLOCKS_MAX (read when a viewer is created) times the speculated and the unspeculated loop, one frame in flight, same process.
usage: python tools/ab_blocks.py [256,1024] [orbit,open_sky,translucent]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

settings = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,1024").split(",")]
scenes = (sys.argv[2] if len(sys.argv) > 2 else "orbit,open_sky,translucent").split(",")
n, sh, w, h, seed = scene.CONFIGS["cfg4"]
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
FRAMES = int(os.environ.get("AB_FRAMES", "120"))
for sc in scenes:
    g = scene.synthetic_gaussians(n, seed, sh, 0, n, variant="translucent") if sc == "translucent" else scene.synthetic_gaussians(n, seed, sh)
    sums = {}
    for rep in range(2):
        for bm in settings:
            if bm:
                os.environ["GSX_BLOCKS_MAX"] = str(bm)
            else:
                os.environ.pop("GSX_BLOCKS_MAX", None)   # 0: the library's own choice (256, or 1024 while some tile's walk is long)
            v = MultiModelViewer()
            v.add_model("m", n)
            v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
            v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
            if sc == "open_sky":
                from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind
                MaskEvaluator(v).evaluate(MaskOp.parse("0"), "m", [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, -4.5, 0.0], np.float32), scale=np.array([10.0, 5.0, 10.0], np.float32))])
            res = {}
            for spec in (1, 0):
                v.set_render_options(speculative=spec)
                for i in range(100 if spec else 20):
                    v.update_camera(orbit[i % 240], (w, h))
                    v.render_frame(["m"])
                v.poll()
                t0 = time.perf_counter()
                for i in range(100, 100 + FRAMES):
                    v.update_camera(orbit[i % 240], (w, h))
                    v.render_frame(["m"])
                v.poll()
                res[spec] = FRAMES / (time.perf_counter() - t0)
            fb = v.download_framebuffer()
            cs = int(np.frombuffer(fb.tobytes(), np.uint32).astype(np.uint64).sum() & 0xFFFFFFFFFFFF)
            sums.setdefault(cs, []).append(bm)
            print(f"{sc:12s} rep {rep} blocks_max {bm:5d}: speculated {res[1]:8.1f} fps, unspeculated {res[0]:8.1f} fps, checksum {cs:x}", flush=True)
            v.close()
    print(f"{sc}: frames identical across settings: {len(sums) == 1}")
    del g
