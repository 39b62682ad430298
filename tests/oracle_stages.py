"""Oracle-backed stage implementation for ``parallel.ShardedViewer`` — TEST ONLY.

It lets the routing / exchange / merge logic of the multi-GPU path run on CPU with gloo (world_size 2)
and be compared bit-for-bit with a single-process oracle frame.  Record format is the HIP one (48 bytes:
mean.xy, rect bits, conic, opacity, rgb, depth) so the GPU pack kernel can be checked against it too.
"""
import numpy as np
import torch

import oracle
from wgpu_3dgs_viewer_app_amd import camera


def records_from_projection(pr):
    """All visible records of an oracle projection, ascending index, in the 48-byte exchange format."""
    vis = np.nonzero(pr["key"] != 0xFFFFFFFF)[0]
    rec = np.zeros((vis.size, 12), np.float32)
    rect = pr["rect"][vis]
    rec[:, 0:2] = pr["mean2d"][vis]
    rec[:, 2] = (rect[:, 0] | (rect[:, 2] << 16)).astype(np.uint32).view(np.float32)
    rec[:, 3] = (rect[:, 1] | (rect[:, 3] << 16)).astype(np.uint32).view(np.float32)
    rec[:, 4:8] = pr["conic_opacity"][vis]
    rec[:, 8:11] = pr["rgb"][vis]
    rec[:, 11] = pr["key"][vis].view(np.float32)
    return rec, rect


def pack_by_destination(pr, world):
    """Records grouped by destination rank (tile row % world), ascending index inside each group."""
    rec, rect = records_from_projection(pr)
    y0, y1 = rect[:, 1].astype(np.int64), rect[:, 3].astype(np.int64)
    groups, counts = [], []
    for g in range(world):
        first = y0 + ((g - y0) % world)
        sel = first < y1
        groups.append(rec[sel])
        counts.append(int(sel.sum()))
    return (np.concatenate(groups) if groups else rec[:0]), counts


def projection_from_records(rec):
    n = rec.shape[0]
    bits2 = rec[:, 2].view(np.uint32)
    bits3 = rec[:, 3].view(np.uint32)
    rect = np.stack([bits2 & 0xFFFF, bits3 & 0xFFFF, bits2 >> 16, bits3 >> 16], 1).astype(np.uint32)
    return dict(key=np.ascontiguousarray(rec[:, 11]).view(np.uint32).copy(), rect=np.ascontiguousarray(rect),
                mean2d=np.ascontiguousarray(rec[:, 0:2]), conic_opacity=np.ascontiguousarray(rec[:, 4:8]),
                rgb=np.ascontiguousarray(rec[:, 8:11]), n_visible=n)


class OracleStages:
    def __init__(self):
        self._fb = None

    def stream_ctx(self):
        import contextlib

        return contextlib.nullcontext()

    def load_shard(self, key, gaussians, start, n_total):
        self.pod = oracle.convert(gaussians)

    def set_uniforms(self, key, cam, size, model_transform=None, gaussian_transform=None):
        mt = model_transform or camera.ModelTransform()
        w, h = size
        self.frame = oracle.frame_setup(cam.view(), cam.projection(w / h), w, h, mt.pos, mt.quat(), mt.scale)
        self.size = (w, h)

    def _render_projection(self, pr, world=1, rank=0):
        f = self.frame
        idx, nvis = oracle.depth_sort(pr["key"])
        off, lst = oracle.tile_lists(f, idx, nvis, pr["rect"])
        fb = oracle.new_framebuffer(f)
        oracle.composite_tiles(f, pr, off, lst, fb)
        for ty in range(f.tiles_y):
            if ty % world != rank:
                fb[ty * 16:(ty + 1) * 16] = (0, 0, 0, 1)
        self._fb = fb
        self._stats = dict(n_gaussians=pr["key"].size, n_visible=nvis, n_tile_entries=int(lst.size))
        return self._stats

    def stats(self, key):
        return self._stats

    def render_local(self, key):
        pos, color, sh, cov = self.pod
        return self._render_projection(oracle.project(self.frame, pos, color, sh, cov))

    def project_and_pack(self, key, world):
        pos, color, sh, cov = self.pod
        pr = oracle.project(self.frame, pos, color, sh, cov)
        send, counts = pack_by_destination(pr, world)
        return torch.from_numpy(np.ascontiguousarray(send)), counts

    def alloc_records(self, n):
        return torch.empty((n, 12), dtype=torch.float32)

    def render_records(self, key, recv, n, world, rank):
        return self._render_projection(projection_from_records(recv.numpy()[:n]), world, rank)

    def _rows_per_rank(self, world):
        return (self.frame.tiles_y + world - 1) // world

    def own_strip(self, world, rank):
        w, h = self.size
        rpr = self._rows_per_rank(world)
        strip = np.zeros((rpr, 16, w, 4), np.float32)
        strip[..., 3] = 1
        for r in range(rpr):
            ty = rank + r * world
            y0, y1 = ty * 16, min(ty * 16 + 16, h)
            if y0 < h:
                strip[r, : y1 - y0] = self._fb[y0:y1]
        return torch.from_numpy(strip.reshape(-1))

    def gather_buffer(self, strip, world):
        return torch.empty(world * strip.numel(), dtype=strip.dtype)

    def assemble(self, gathered, world):
        w, h = self.size
        rpr = self._rows_per_rank(world)
        fb = np.zeros((h, w, 4), np.float32)
        parts = gathered.numpy().reshape(world, -1)
        for g, p in enumerate(parts):
            s = p.reshape(rpr, 16, w, 4)
            for r in range(rpr):
                ty = g + r * world
                y0, y1 = ty * 16, min(ty * 16 + 16, h)
                if y0 < h:
                    fb[y0:y1] = s[r, : y1 - y0]
        self._fb = fb

    def framebuffer(self):
        return self._fb

    def poll(self):
        pass

    def set_pass_timing(self, on):
        pass

    def get_pass_timing(self):
        return {}

    def close(self):
        pass
