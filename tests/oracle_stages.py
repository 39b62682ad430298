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


def pack_by_destination(pr, world, tiles_x, tiles_y, window=None):
    """``gsx_shard_pack`` restated: a visible record goes to rank g (band = tile rows [g*rpr, (g+1)*rpr)) if its
    rectangle holds, inside the band, a tile whose window [lo, hi) (uint32 [tiles_y, tiles_x, 2]; None = everything)
    contains the record's depth key.  Groups in rank order, ascending index."""
    rec, rect = records_from_projection(pr)
    keys = np.ascontiguousarray(rec[:, 11]).view(np.uint32).astype(np.int64)
    rpr = (tiles_y + world - 1) // world
    groups, counts = [], []
    for g in range(world):
        y0 = np.maximum(rect[:, 1].astype(np.int64), g * rpr)
        y1 = np.minimum(rect[:, 3].astype(np.int64), (g + 1) * rpr)
        sel = y0 < y1
        if window is not None:
            lo, hi = window[..., 0].astype(np.int64), window[..., 1].astype(np.int64)
            for i in np.nonzero(sel)[0]:
                sl = (slice(y0[i], y1[i]), slice(rect[i, 0], rect[i, 2]))
                sel[i] = bool(((lo[sl] <= keys[i]) & (keys[i] < hi[sl])).any())
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
        self.pods, self.frames, self._prs, self._imp = {}, {}, {}, {}

    @property
    def frame(self):  # tile grid / viewport are the same for every model
        return next(iter(self.frames.values()))

    def stream_ctx(self):
        import contextlib

        return contextlib.nullcontext()

    def load_shard(self, key, gaussians, start, n_total):
        self.pods[key] = oracle.convert(gaussians)

    def set_uniforms(self, key, cam, size, model_transform=None, gaussian_transform=None):
        mt = model_transform or camera.ModelTransform()
        w, h = size
        self.frames[key] = oracle.frame_setup(cam.view(), cam.projection(w / h), w, h, mt.pos, mt.quat(), mt.scale)
        self.size = (w, h)

    def _band(self, world, rank):
        rpr = (self.frame.tiles_y + world - 1) // world
        return rpr, min(rank * rpr, self.frame.tiles_y), min((rank + 1) * rpr, self.frame.tiles_y)

    def _lists(self, key, pr, window=None):
        """(projection, tile offsets, tile list) of a record set, window-filtered"""
        f = self.frames[key]
        idx, nvis = oracle.depth_sort(pr["key"])
        off, lst = oracle.tile_lists(f, idx, nvis, pr["rect"])
        tile_of = np.repeat(np.arange(off.size - 1), np.diff(off).astype(np.int64))
        if window is not None:  # a tile bins only the records its window admits
            k = pr["key"][lst].astype(np.int64)
            w = window.reshape(-1, 2).astype(np.int64)
            keep = (w[tile_of, 0] <= k) & (k < w[tile_of, 1])
            lst = np.ascontiguousarray(lst[keep])
            tile_of = tile_of[keep]
            cnt = np.bincount(tile_of, minlength=off.size - 1)
            off = np.concatenate([[0], np.cumsum(cnt)]).astype(off.dtype)
        # deepest key binned into each tile: an upper bound of the depth at which a saturated tile saturated
        deepest = np.zeros(off.size - 1, np.uint32)
        np.maximum.at(deepest, tile_of, pr["key"][lst])
        return dict(pr=pr, off=off, lst=lst, nvis=nvis, deepest=deepest.reshape(f.tiles_y, f.tiles_x))

    def _composite(self, keys, world=1, rank=0, more=False):
        """layers far -> near: each model is composited IN FRONT of what the framebuffer holds (composite_tiles)"""
        f = self.frame
        layer = oracle.new_framebuffer(f)
        for key in keys:
            L = self._imp[key]
            oracle.composite_tiles(self.frames[key], L["pr"], L["off"], L["lst"], layer)
        near = self._imp[keys[-1]]
        if more:  # this set lies BEHIND what the framebuffer holds
            front = self._fb
            fb = np.empty_like(front)
            fb[..., :3] = front[..., :3] + front[..., 3:4] * layer[..., :3]
            fb[..., 3] = front[..., 3] * layer[..., 3]
            self._deepest = np.maximum(self._deepest, near["deepest"])
        else:
            fb = layer
            self._deepest = near["deepest"]
        _, lo, hi = self._band(world, rank)
        fb[: lo * 16] = (0, 0, 0, 1)
        fb[hi * 16:] = (0, 0, 0, 1)
        self._fb = fb
        self._stats = {k: dict(n_gaussians=self._imp[k]["pr"]["key"].size, n_visible=self._imp[k]["nvis"],
                               n_tile_entries=int(self._imp[k]["lst"].size)) for k in keys}

    def stats(self, key):
        return dict(self._stats[key], n_repair_tiles=getattr(self, "_need_count", 0))

    # ---- device-resident protocol of the product (include/gsx.h), restated on the host: fixed slots whose headers carry
    #      the counts, windows / verification / next limits kept by the stage backend ----
    EXTRA = 4  # statistics words behind every rank's band of saturation keys

    def _m(self, key):
        """per-model protocol state (limits, windows, slot figures): models are exchanged one after the other"""
        self._ms = getattr(self, "_ms", {})
        return self._ms.setdefault(key, {})

    def frame_begin(self, key, world, rank, speculate=True, limit=None):
        from wgpu_3dgs_viewer_app_amd import parallel

        f = self.frame
        m = self._m(key)
        if limit is not None:
            m["limit"] = np.array(limit, np.uint32)
        lim = m.get("limit")
        m["limited"] = bool(speculate and lim is not None and lim.shape == (f.tiles_y, f.tiles_x))
        m["win1"] = parallel.windows_first(lim) if m["limited"] else None
        m["win2"] = np.zeros((f.tiles_y, f.tiles_x, 2), np.uint32)
        self._prs[key] = self._project(key)
        self._world, self._rank = world, rank
        m["slot_max"], m["slot_over"] = [0, 0], [0, 0]
        self._need_count = 0

    def slot_records(self, key, world, shard_max):
        return int(shard_max)   # no adaptive policy here: the safe size, identical on every rank

    def pack_slots(self, key, world, rnd, slot):
        m = self._m(key)
        window = m["win1"] if rnd == 0 else m["win2"]
        send, counts = pack_by_destination(self._prs[key], world, self.frame.tiles_x, self.frame.tiles_y, window)
        out = np.zeros((world, slot + 1, 12), np.float32)
        o = 0
        for g, c in enumerate(counts):
            sent = min(c, slot)
            out[g, 0, 0:2] = np.array([c, sent], np.uint32).view(np.float32)
            out[g, 1:1 + sent] = send[o:o + sent]
            o += c
        m["slot_max"][rnd] = max(counts) if counts else 0
        m["slot_over"][rnd] = int(m["slot_max"][rnd] > slot)
        return torch.from_numpy(out)

    def alloc_slots(self, world, slot, rnd):
        return torch.zeros((world, slot + 1, 12), dtype=torch.float32)

    def _saturated(self):
        """bool [tiles_y, tiles_x]: every pixel of the tile has T < 1e-4 (padding pixels do not exist: partial tiles count
        their real pixels)"""
        f = self.frame
        out = np.zeros((f.tiles_y, f.tiles_x), bool)
        for ty in range(f.tiles_y):
            for tx in range(f.tiles_x):
                tile = self._fb[ty * 16: ty * 16 + 16, tx * 16: tx * 16 + 16, 3]
                out[ty, tx] = bool(tile.size) and bool((tile < 1e-4).all())
        return out

    def import_slots(self, key, recv, world, rank, rnd, slot, behind=False):
        r = recv.numpy()
        parts = []
        for s in range(world):
            sent = int(r[s, 0, 1:2].view(np.uint32)[0])
            parts.append(r[s, 1:1 + sent])
        recs = np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros((0, 12), np.float32)
        m = self._m(key)
        window = m["win1"] if rnd == 0 else m["win2"]
        self._world, self._rank = world, rank
        self._imp[key] = self._lists(key, projection_from_records(recs), window)
        if rnd == 0:  # a layered frame: the tiles nearer models saturated say nothing about this model's depths
            m["done_before"] = self._saturated() if behind else None
        self._fb_key = key
        self._composite([key], world, rank, more=(rnd == 1 or behind))

    def alloc_sat(self, world, mine):
        return torch.zeros(world * mine.numel(), dtype=torch.int32)

    def _sat_map(self, sat_all, world):
        f = self.frame
        rpr = (f.tiles_y + world - 1) // world
        a = sat_all.numpy().view(np.uint32).reshape(world, rpr * f.tiles_x + self.EXTRA)
        return a[:, : rpr * f.tiles_x].reshape(world * rpr, f.tiles_x)[: f.tiles_y], a[:, rpr * f.tiles_x:]

    def verify(self, key, world, sat_all):
        from wgpu_3dgs_viewer_app_amd import parallel

        sat, extra = self._sat_map(sat_all, world)
        m = self._m(key)
        if m["limited"]:
            need = (m["limit"] < parallel.KEY_ALL) & (sat == 0)
            m["win2"] = parallel.windows_second(m["limit"], need)
            self._need_count = int(need.sum())
        else:
            m["win2"] = np.zeros(sat.shape + (2,), np.uint32)
            self._need_count = 0
        self._seq = getattr(self, "_seq", 0) + 1
        self._verdicts = getattr(self, "_verdicts", {})
        self._verdicts[self._seq] = dict(need_tiles=self._need_count, overflow=bool(extra[:, 1].any()), max_records=int(extra[:, 0].max()))
        return self._seq

    def wait_verdict(self, key, seq):
        return self._verdicts.pop(seq)

    def repair_count(self, key, world):
        _, counts = pack_by_destination(self._prs[key], world, self.frame.tiles_x, self.frame.tiles_y, self._m(key)["win2"])
        return torch.tensor([max(counts) if counts else 0, 0, 0, 0], dtype=torch.int32)

    def alloc_counts(self, world):
        return torch.zeros(4 * world, dtype=torch.int32)

    def post_counts(self, world, counts_all):
        self._seq = getattr(self, "_seq", 0) + 1
        self._verdicts[self._seq] = dict(need_tiles=0, overflow=False, max_records=int(counts_all.numpy().reshape(world, 4)[:, 0].max()))
        return self._seq

    def next_windows(self, key, world, sat_all, margin, radius):
        from wgpu_3dgs_viewer_app_amd import parallel

        sat, _ = self._sat_map(sat_all, world)
        self._m(key)["limit_next"] = parallel.next_limits(sat, margin, radius)

    def frame_end(self, key):
        m = self._m(key)
        m["limit"] = m["limit_next"]

    def limits(self, key):
        return self._m(key)["limit"]

    def _project(self, key):
        pos, color, sh, cov = self.pods[key]
        return oracle.project(self.frames[key], pos, color, sh, cov)

    def render_local(self, key):
        self.render_local_keys([key])

    def render_local_keys(self, keys):
        for k in keys:
            self._imp[k] = self._lists(k, self._project(k))
        self._world, self._rank = 1, 0
        self._composite(keys)

    def render_band(self, keys, world, rank):
        """screen-band mode: the whole scene is here; composite only this rank's band of tile rows"""
        for k in keys:
            self._imp[k] = self._lists(k, self._project(k))
        self._world, self._rank = world, rank
        self._composite(list(keys), world, rank)

    def begin_frame(self, key, world, rank, window=None):
        self._prs[key] = self._project(key)
        self._set_win = window

    def pack(self, key, world, window=None):
        if window is None:
            window = self._set_win  # the windows announced at begin_frame
        send, counts = pack_by_destination(self._prs[key], world, self.frame.tiles_x, self.frame.tiles_y, window)
        return torch.from_numpy(np.ascontiguousarray(send)), counts

    def alloc_records(self, n):
        return torch.empty((n, 12), dtype=torch.float32)

    def import_records(self, key, recv, n, world, rank, window=None):
        self._world, self._rank = world, rank
        self._imp[key] = self._lists(key, projection_from_records(recv.numpy()[:n]), window)

    def render_keys(self, keys, more=False):
        self._composite(list(keys), self._world, self._rank, more)

    def render_records(self, key, recv, n, world, rank, more=False, window=None):
        self.import_records(key, recv, n, world, rank, window)
        self.render_keys([key], more)

    def feedback(self, key, world, rank):
        """Saturation depth keys of this rank's band (rows_per_rank x tiles_x, 0 = open): a tile is saturated when all
        its pixels have T < 1e-4; its key is the deepest one binned into it."""
        f = self.frame
        rpr, lo, hi = self._band(world, rank)
        out = np.zeros((rpr, f.tiles_x), np.uint32)
        m = self._m(key)
        before = m.get("done_before")
        sat = self._saturated()
        for ty in range(lo, hi):
            for tx in range(f.tiles_x):
                if sat[ty, tx]:
                    out[ty - lo, tx] = 1 if (before is not None and before[ty, tx]) else max(int(self._deepest[ty, tx]), 1)
        extra = np.zeros(self.EXTRA, np.uint32)
        extra[0], extra[1] = m.get("slot_max", [0, 0])[0], m.get("slot_over", [0, 0])[0]
        return torch.from_numpy(np.concatenate([out.reshape(-1), extra]).view(np.int32).copy())

    def own_band(self):
        w, h = self.size
        rpr, lo, hi = self._band(self._world, self._rank)
        band = np.zeros((rpr * 16, w, 4), np.float32)
        rows = self._fb[lo * 16: min(hi * 16, h)]
        band[: rows.shape[0]] = rows
        return torch.from_numpy(band.reshape(-1))

    @staticmethod
    def resolve_rgba8(fb, background):
        """(rgb, T) over a background -> RGBA8 words: fma(T, bg, C) clamped, alpha = 1 - T, round half up (k_resolve_rgba8)"""
        p = fb.astype(np.float64)
        out = np.zeros(fb.shape[:-1], np.uint32)
        for c in range(3):
            v = np.clip((p[..., 3] * np.float64(np.float32(background[c])) + p[..., c]).astype(np.float32), 0.0, 1.0)
            out |= np.floor(v * np.float32(255.0) + np.float32(0.5)).astype(np.uint32) << np.uint32(8 * c)
        a = np.clip(np.float32(1.0) - fb[..., 3], 0.0, 1.0).astype(np.float32)
        out |= np.floor(a * np.float32(255.0) + np.float32(0.5)).astype(np.uint32) << np.uint32(24)
        return out

    def own_band_rgba8(self, background=(0.0, 0.0, 0.0), slot=0):
        w, h = self.size
        rpr, lo, hi = self._band(self._world, self._rank)
        band = np.zeros((rpr * 16, w, 4), np.float32)
        band[..., 3] = 0.0  # padding rows: the device framebuffer is zero-initialised there
        rows = self._fb[lo * 16: min(hi * 16, h)]
        band[: rows.shape[0]] = rows
        return torch.from_numpy(self.resolve_rgba8(band, background).view(np.int32).reshape(-1).copy())

    def own_frame_rgba8(self, background=(0.0, 0.0, 0.0), slot=0):
        return torch.from_numpy(self.resolve_rgba8(self._fb, background).view(np.int32).reshape(-1).copy())

    def gather_target_frames_rgba8(self, world):
        w, h = self.size
        self._frames8 = torch.zeros(world * w * h, dtype=torch.int32)
        return self._frames8

    def frames_rgba8(self):
        w, h = self.size
        return self._frames8.numpy().view(np.uint8).reshape(-1, h, w, 4).copy()

    def gather_target_rgba8(self):
        w, h = self.size
        rpr, _, _ = self._band(self._world, self._rank)
        self._gather8 = torch.zeros(self._world * rpr * 16 * w, dtype=torch.int32)
        return self._gather8

    def frame_rgba8(self):
        w, h = self.size
        return self._gather8.numpy().view(np.uint8).reshape(-1, w, 4)[:h].copy()

    def gather_target(self):
        w, h = self.size
        rpr, _, _ = self._band(self._world, self._rank)
        self._gather = torch.zeros(self._world * rpr * 16 * w * 4, dtype=torch.float32)
        return self._gather

    def framebuffer(self):
        if getattr(self, "_gather", None) is not None:
            w, h = self.size
            return self._gather.numpy().reshape(-1, w, 4)[:h].copy()
        return self._fb

    def poll(self):
        pass

    def set_pass_timing(self, on, passes=None):
        pass

    def get_pass_timing(self):
        return {}

    def close(self):
        pass
