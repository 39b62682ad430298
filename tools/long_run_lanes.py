import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from wgpu_3dgs_viewer_app_amd import camera, scene
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer
n, sh, w, h, seed = scene.CONFIGS["cfg3"]
v = MultiModelViewer()
v.add_model("m", n)
v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, scene.synthetic_gaussians(n, seed, sh))
v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
ref = MultiModelViewer(); ref.add_model("m", n)
ref.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, scene.synthetic_gaussians(n, seed, sh))
ref.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
ref.set_render_options(speculative=0, progressive=0)
rng = np.random.default_rng(5)
for rep, lanes in enumerate((1, 2, 3, 2, 1)):
    v.set_render_options(frames_in_flight=lanes)
    torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]; t0 = time.perf_counter()
    for i in range(4000):
        size = (w, h) if (i // 500) % 2 == 0 else (1280, 720)
        v.update_camera(camera.orbit_pose((i * 3) % 240 if i % 97 else int(rng.integers(0, 240))), size)
        v.render_frame(["m"])
    v.poll(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pose = int(rng.integers(0, 240))
    for x in (v, ref):
        x.update_camera(camera.orbit_pose(pose), (w, h)); x.render_frame(["m"])
    same = np.array_equal(v.download_framebuffer(), ref.download_framebuffer())
    print(rep, "lanes", lanes, round(4000 / dt, 1), "fps; free GPU memory change MB:", (torch.cuda.mem_get_info()[0] - free0) / 1e6, "frame equals the single-pass frame:", same, flush=True)
    assert same
print("long run OK")
