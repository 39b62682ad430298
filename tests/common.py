"""Shared helpers for the parity tests: seeded scenes sized so the oracle finishes in seconds."""
import numpy as np

import oracle
from wgpu_3dgs_viewer_app_amd import camera, scene


def small_scene(n, seed, sh_degree=3, scale_mul=6.0):
    """Synthetic scene with enlarged splats so a small frame sees real overdraw."""
    g = scene.synthetic_gaussians(n, seed, sh_degree)
    g["scale"] *= np.float32(scale_mul)
    return g


def default_transform():
    return camera.ModelTransform()


def odd_transform():
    return camera.ModelTransform(pos=np.array([0.3, -0.2, 0.5], np.float32), rot=np.array([20, -35, 50], np.float32),
                                 scale=np.array([1.2, 0.9, 1.1], np.float32))


def oracle_frame(cam, w, h, mt=None, size=1.0, display_mode=0, sh_deg=3, no_sh0=0, params=None):
    mt = mt or camera.ModelTransform()
    return oracle.frame_setup(cam.view(), cam.projection(w / h), w, h, mt.pos, mt.quat(), mt.scale, size, display_mode,
                              sh_deg, no_sh0, params)


def oracle_model_frame(g, cam, w, h, mt=None, fb=None, mask=None, **kw):
    """Oracle pipeline for one model; returns (frame, projection dict, sorted idx, n_vis, fb)."""
    f = oracle_frame(cam, w, h, mt, **kw)
    pos, color, sh, cov = oracle.convert(g)
    pr = oracle.project(f, pos, color, sh, cov, mask)
    idx, nvis = oracle.depth_sort(pr["key"])
    if fb is None:
        fb = oracle.new_framebuffer(f)
    oracle.rasterize(f, pr, idx, nvis, fb)
    return f, pr, idx, nvis, fb


class ThreadHub:
    """Shared state of ``ThreadComm``: `world` ranks of ONE process (threads), each with its own viewer + stream."""

    def __init__(self, world, timeout=120.0):
        import threading

        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.slots = [None] * world


class ThreadComm:
    """Stand-in for ``parallel.TorchComm`` that lets `world` ranks run the real exchange protocol on one device:
    the collectives are device-to-device copies between the ranks' buffers, delivered exactly in the order
    ``all_to_all_single`` / ``all_gather_into_tensor`` deliver them (by source rank).  TEST ONLY."""

    def __init__(self, hub, rank):
        self.hub, self.rank = hub, rank
        self.bytes_sent = 0

    @staticmethod
    def _sync(t):
        import torch

        if t.is_cuda:
            torch.cuda.current_stream().synchronize()

    def all_to_all_slots(self, recv, send):
        """Fixed-size slots [world, 1 + T, 12]: slot p of `send` goes to rank p.  ``bytes_sent`` counts the PAYLOAD the
        headers announce (word 1 of a slot's first record = records sent), not the slot size."""
        import numpy as np

        h = self.hub
        self._sync(send)
        h.slots[self.rank] = send
        sent = send[:, 0, 1].cpu().numpy().view(np.uint32)
        self.bytes_sent += int(sum(int(c) for g, c in enumerate(sent) if g != self.rank)) * send.shape[2] * 4
        h.barrier.wait()
        for src in range(h.world):
            s = h.slots[src]
            assert s.shape == send.shape, f"rank {src} sized its slots {tuple(s.shape)}, rank {self.rank} {tuple(send.shape)}"
            recv[src].copy_(s[self.rank])
        self._sync(recv)
        h.barrier.wait()

    def all_gather(self, out, inp):
        h = self.hub
        self._sync(inp)
        h.slots[self.rank] = inp
        h.barrier.wait()
        n = inp.numel()
        for src in range(h.world):
            out[src * n:(src + 1) * n].copy_(h.slots[src])
        self._sync(out)
        h.barrier.wait()


def run_ranks(world, fn):
    """Run fn(rank, comm) on `world` threads; re-raises the first failure (and releases the others)."""
    import threading

    hub = ThreadHub(world)
    errors, results = [], [None] * world

    def body(r):
        try:
            results[r] = fn(r, ThreadComm(hub, r))
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            hub.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    real = [e for e in errors if not isinstance(e, threading.BrokenBarrierError)]
    if real or errors:
        raise (real or errors)[0]
    return results
