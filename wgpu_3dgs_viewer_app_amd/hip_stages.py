"""Stage backend of ``parallel.ShardedViewer`` over libgsx.so (the product path; HIP kernels only).

PyTorch is used for the device buffers that RCCL moves (send / receive records, framebuffer strips) and
for stream identity; every byte in those buffers is produced and consumed by the HIP kernels in csrc/.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np

from . import _lib
from .camera import ModelTransform
from .viewer import Cov3dKind, GaussianDisplayMode, GaussianShDegree, MultiModelViewer, ShKind

RECORD_FLOATS = 12


class HipStages:
    def __init__(self, device: int = 0, stream=None, use_torch: bool = False, sh: int = 0, cov3d: int = 0):
        self.device = device
        self.torch_stream = None
        if use_torch:
            import torch

            # a dedicated non-default stream shared by torch (RCCL staging) and libgsx, so launch order = data order
            self.torch_stream = torch.cuda.Stream(device=device)
            stream = self.torch_stream.cuda_stream
        self.viewer = MultiModelViewer(size=(1, 1), device=device, stream=stream, sh=ShKind(sh), cov3d=Cov3dKind(cov3d))
        self._size = (1, 1)
        self._send = None
        self._fb_t = None
        self._fbk = None

    def stream_ctx(self):
        if self.torch_stream is None:
            return contextlib.nullcontext()
        import torch

        return torch.cuda.stream(self.torch_stream)

    # -- scene --
    def load_shard(self, key: str, gaussians: np.ndarray, start: int, n_total: int) -> None:
        self.viewer.add_model(key, gaussians.shape[0])
        self.viewer.models[key].gaussian_buffers.gaussians_buffer.update_range(0, gaussians)
        self._n_local = max(getattr(self, "_n_local", 0), gaussians.shape[0])  # sizes the shared send buffer

    def set_uniforms(self, key, camera, size, model_transform=None, gaussian_transform=None) -> None:
        mt = model_transform or ModelTransform()
        gt = gaussian_transform or (1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
        self.viewer.update_camera(camera, size)
        # every frame, like the app (scene.rs:796-809): three host-side setters, no device work
        self.viewer.update_model_transform(key, mt.pos, mt.quat(), mt.scale)
        self.viewer.update_gaussian_transform(*gt)
        self._size = (int(size[0]), int(size[1]))

    # -- single GPU --
    def render_local(self, key: str) -> None:
        """Enqueue one whole frame; no host synchronisation (statistics are fetched lazily by ``stats``)."""
        self.viewer.render_frame([key])

    def render_local_keys(self, keys) -> None:
        self.viewer.render_frame(list(keys))

    def stats(self, key: str) -> dict:
        return self.viewer.frame_stats(key)

    # -- multi GPU (stage split of include/gsx.h) --
    def begin_frame(self, key: str, world: int, rank: int, window=None) -> None:
        """Project the resident shard and make sure the library renders into the padded framebuffer RCCL gathers into.
        ``window``: the windows of the coming exchange (``gsx_shard_set_windows``): the projection then shades only what can
        travel, and ``pack(key, world, None)`` packs from that candidate list."""
        import torch

        v = self.viewer
        lay = _lib.ShardLayout()
        _lib.check(v._L.gsx_shard_layout(v._h, world, rank, C.byref(lay)))
        n_floats = lay.padded_framebuffer_bytes // 4
        if self._fb_t is None or self._fb_t.numel() != n_floats:
            self._fb_t = torch.zeros(n_floats, dtype=torch.float32, device=f"cuda:{self.device}")
            self._band_t = torch.empty(lay.band_bytes // 4, dtype=torch.float32, device=f"cuda:{self.device}")
            self._fbk = None
            _lib.check(v._L.gsx_viewer_set_external_framebuffer(v._h, self._fb_t.data_ptr(), lay.padded_framebuffer_bytes))
        self._lay = lay
        _lib.check(v._L.gsx_shard_set_windows(v._h, key.encode(), self._window_ptr(window, "_set_win_t")))
        v.preprocessor.preprocess(key)

    # -- multi GPU, device-resident protocol (include/gsx.h): nothing below waits for the device or reads a count --
    def _padded_framebuffer(self, world: int, rank: int):
        import torch

        v = self.viewer
        lay = _lib.ShardLayout()
        _lib.check(v._L.gsx_shard_layout(v._h, world, rank, C.byref(lay)))
        n_floats = lay.padded_framebuffer_bytes // 4
        if self._fb_t is None or self._fb_t.numel() != n_floats:
            self._fb_t = torch.zeros(n_floats, dtype=torch.float32, device=f"cuda:{self.device}")
            self._band_t = torch.empty(lay.band_bytes // 4, dtype=torch.float32, device=f"cuda:{self.device}")
            self._fbk = None
            _lib.check(v._L.gsx_viewer_set_external_framebuffer(v._h, self._fb_t.data_ptr(), lay.padded_framebuffer_bytes))
        self._lay = lay

    def frame_begin(self, key: str, world: int, rank: int, speculate: bool = True, limit=None) -> None:
        """``gsx_shard_frame_begin``: windows from the limits the last frame left on the device (``limit``: numpy uint32
        [tiles_y, tiles_x] to use instead — tests), then the projection of the resident shard."""
        import torch

        self._padded_framebuffer(world, rank)
        v = self.viewer
        d_lim = None
        if limit is not None:
            self._lim_t = torch.from_numpy(np.ascontiguousarray(limit, np.uint32).view(np.int32)).to(f"cuda:{self.device}", non_blocking=True)
            d_lim = self._lim_t.data_ptr()
        _lib.check(v._L.gsx_shard_frame_begin(v._h, key.encode(), world, rank, 1 if speculate else 0, d_lim))

    def slot_records(self, key: str, world: int, shard_max: int) -> int:
        """``gsx_shard_slot_records``: round 0's slot size — the same number on every rank (``shard_max`` = the largest shard
        of the model; the figure behind the policy is the last verdict's global maximum)."""
        t = C.c_uint32()
        _lib.check(self.viewer._L.gsx_shard_slot_records(self.viewer._h, key.encode(), world, int(shard_max), C.byref(t)))
        return int(t.value)

    def pack_slots(self, key: str, world: int, rnd: int, slot: int):
        """``gsx_shard_pack_slots`` -> float32 tensor [world, 1 + slot, 12]: slot g for rank g, header first."""
        import torch

        pool = self.__dict__.setdefault("_send_slots", {})
        if pool.get(rnd) is None or pool[rnd].shape != (world, slot + 1, RECORD_FLOATS):
            pool[rnd] = torch.empty((world, slot + 1, RECORD_FLOATS), dtype=torch.float32, device=f"cuda:{self.device}")
        t = pool[rnd]
        _lib.check(self.viewer._L.gsx_shard_pack_slots(self.viewer._h, key.encode(), world, rnd, t.data_ptr(), slot))
        return t

    def alloc_slots(self, world: int, slot: int, rnd: int):
        import torch

        pool = self.__dict__.setdefault("_recv_slots", {})
        if pool.get(rnd) is None or pool[rnd].shape != (world, slot + 1, RECORD_FLOATS):
            pool[rnd] = torch.empty((world, slot + 1, RECORD_FLOATS), dtype=torch.float32, device=f"cuda:{self.device}")
        return pool[rnd]

    def import_slots(self, key: str, recv, world: int, rank: int, rnd: int, slot: int, behind: bool = False) -> None:
        """``gsx_shard_import_slots``: import (counts from the slot headers, on the device) + depth sort + render; ``behind``:
        a layered frame's model that is not the nearest (GSX_SHARD_BEHIND)."""
        _lib.check(self.viewer._L.gsx_shard_import_slots(self.viewer._h, key.encode(), recv.data_ptr(), world, rank, rnd | (2 if behind else 0), slot))

    def alloc_sat(self, world: int, mine):
        import torch

        if getattr(self, "_sat_all", None) is None or self._sat_all.numel() != world * mine.numel():
            self._sat_all = torch.zeros(world * mine.numel(), dtype=mine.dtype, device=mine.device)
        return self._sat_all

    def verify(self, key: str, world: int, sat_all) -> int:
        """``gsx_shard_verify``: repair windows on the device; returns the sequence number of the verdict it will post."""
        seq = C.c_uint32()
        _lib.check(self.viewer._L.gsx_shard_verify(self.viewer._h, key.encode(), world, sat_all.data_ptr(), C.byref(seq)))
        return int(seq.value)

    def wait_verdict(self, key, seq: int) -> dict:
        """``gsx_shard_wait_verdict``: the one host wait of a frame (two pinned words; no stream synchronisation)."""
        out = _lib.ShardVerdict()
        _lib.check(self.viewer._L.gsx_shard_wait_verdict(self.viewer._h, key.encode() if key else None, int(seq), C.byref(out)))
        return dict(need_tiles=int(out.need_tiles), overflow=bool(out.overflow), max_records=int(out.max_records))

    def repair_count(self, key: str, world: int):
        """``gsx_shard_repair_count`` -> int32 tensor [4]: {records for this rank's busiest destination under the repair windows, 0, 0, 0}"""
        import torch

        if getattr(self, "_cnt4", None) is None:
            self._cnt4 = torch.zeros(4, dtype=torch.int32, device=f"cuda:{self.device}")
        _lib.check(self.viewer._L.gsx_shard_repair_count(self.viewer._h, key.encode(), world, self._cnt4.data_ptr()))
        return self._cnt4

    def alloc_counts(self, world: int):
        import torch

        if getattr(self, "_cnt_all", None) is None or self._cnt_all.numel() != 4 * world:
            self._cnt_all = torch.zeros(4 * world, dtype=torch.int32, device=f"cuda:{self.device}")
        return self._cnt_all

    def post_counts(self, world: int, counts_all) -> int:
        seq = C.c_uint32()
        _lib.check(self.viewer._L.gsx_shard_post_counts(self.viewer._h, world, counts_all.data_ptr(), C.byref(seq)))
        return int(seq.value)

    def frame_end(self, key: str) -> None:
        _lib.check(self.viewer._L.gsx_shard_frame_end(self.viewer._h, key.encode()))

    def next_windows(self, key: str, world: int, sat_all, margin: float, radius: int) -> None:
        _lib.check(self.viewer._L.gsx_shard_next_windows(self.viewer._h, key.encode(), world, sat_all.data_ptr(), float(margin), int(radius)))

    def limits(self, key: str) -> np.ndarray:
        """The per-tile limits the next frame will use (parity / debugging; synchronises)."""
        w, h = self._size
        tx, ty = (w + 15) // 16, (h + 15) // 16
        out = np.empty((ty, tx), np.uint32)
        _lib.check(self.viewer._L.gsx_shard_download_limits(self.viewer._h, key.encode(), out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size))
        return out

    def render_frame_lib(self, keys, shard_max, speculate: bool, margin: float, radius: int) -> None:
        """``gsx_shard_render_frame_keys``: the whole index-sharded frame inside the library (collectives over RCCL or the
        in-process group); ``keys`` far -> near, ``shard_max`` per key."""
        v = self.viewer
        if self._fb_t is not None:   # the library gathers into a padded framebuffer of its own
            _lib.check(v._L.gsx_viewer_set_external_framebuffer(v._h, None, 0))
            self._fb_t = None
        v.shard_render_frame_keys(list(keys), list(shard_max), speculate, margin, radius)

    def render_band(self, keys, world: int, rank: int) -> None:
        """Screen-band mode (the whole scene is resident on every GPU): render band `rank` of `world` into the padded
        framebuffer the bands are all-gathered into.  No record exchange; enqueued without host synchronisation."""
        import torch

        v = self.viewer
        lay = _lib.ShardLayout()
        _lib.check(v._L.gsx_shard_layout(v._h, world, rank, C.byref(lay)))
        n_floats = lay.padded_framebuffer_bytes // 4
        if self._fb_t is None or self._fb_t.numel() != n_floats:
            self._fb_t = torch.zeros(n_floats, dtype=torch.float32, device=f"cuda:{self.device}")
            self._band_t = torch.empty(lay.band_bytes // 4, dtype=torch.float32, device=f"cuda:{self.device}")
            _lib.check(v._L.gsx_viewer_set_external_framebuffer(v._h, self._fb_t.data_ptr(), lay.padded_framebuffer_bytes))
        self._lay = lay
        _lib.check(v._L.gsx_viewer_set_band(v._h, lay.row_lo, lay.row_hi))
        v.render_frame(list(keys))

    def _window_ptr(self, window, slot):
        """numpy uint32 [tiles_y, tiles_x, 2] -> device pointer (the library copies it, the tensor is kept until reused)."""
        import torch

        if window is None:
            return None
        t = torch.from_numpy(np.ascontiguousarray(window, np.uint32).view(np.int32)).to(f"cuda:{self.device}", non_blocking=True)
        setattr(self, slot, t)
        return t.data_ptr()

    def pack(self, key: str, world: int, window=None):
        """``gsx_shard_pack``; ``window``: per-tile depth-key windows, numpy uint32 [tiles_y, tiles_x, 2]; None = the windows
        given to ``begin_frame`` (candidate list), or everything if there were none."""
        import torch

        v = self.viewer
        cap = max(self._n_local * min(world, 3), 1)  # a record reaches every band its rectangle touches; 3 is generous
        if self._send is None or self._send.shape[0] < cap:
            self._send = torch.empty((cap, RECORD_FLOATS), dtype=torch.float32, device=f"cuda:{self.device}")
        d_win = self._window_ptr(window, "_pack_win_t")
        counts = (C.c_uint64 * world)()
        try:
            _lib.check(v._L.gsx_shard_pack(v._h, key.encode(), world, d_win, self._send.data_ptr(), cap, counts))
        except _lib.GsxError:
            need = sum(int(c) for c in counts)
            if need <= cap:
                raise
            self._send = torch.empty((need, RECORD_FLOATS), dtype=torch.float32, device=f"cuda:{self.device}")
            _lib.check(v._L.gsx_shard_pack(v._h, key.encode(), world, d_win, self._send.data_ptr(), need, counts))
        return self._send, [int(c) for c in counts]

    def alloc_records(self, n: int):
        import torch

        return torch.empty((n, RECORD_FLOATS), dtype=torch.float32, device=f"cuda:{self.device}")

    def import_records(self, key: str, recv, n: int, world: int, rank: int, window=None) -> None:
        """``gsx_shard_import`` + ``gsx_sort``; ``window``: the same per-tile windows that were given to ``pack``."""
        v = self.viewer
        d_win = self._window_ptr(window, "_imp_win_t")
        _lib.check(v._L.gsx_shard_import(v._h, key.encode(), recv.data_ptr() if n else None, n, world, rank, d_win))
        v.radix_sorter.sort(key)

    def render_keys(self, keys, more: bool = False) -> None:
        """``gsx_render`` / ``gsx_render_more`` over the imported record sets, ``keys`` far -> near (scene.rs:533-558)."""
        v = self.viewer
        arr = (C.c_char_p * len(keys))(*[k.encode() for k in keys])
        _lib.check((v._L.gsx_render_more if more else v._L.gsx_render)(v._h, arr, len(keys)))

    def render_records(self, key: str, recv, n: int, world: int, rank: int, more: bool = False, window=None) -> None:
        self.import_records(key, recv, n, world, rank, window)
        self.render_keys([key], more)

    def feedback(self, key: str, world: int, rank: int):
        """Device tensor int32[rows_per_rank * tiles_x] (u32 bit patterns): saturation depth key of every tile of this
        rank's band, 0 = open."""
        import torch

        v = self.viewer
        nw = C.c_uint32()
        _lib.check(v._L.gsx_shard_feedback_words(v._h, world, C.byref(nw)))
        if self._fbk is None or self._fbk.numel() != nw.value:
            self._fbk = torch.zeros(nw.value, dtype=torch.int32, device=f"cuda:{self.device}")
        _lib.check(v._L.gsx_shard_feedback(v._h, key.encode(), world, rank, self._fbk.data_ptr()))
        return self._fbk

    # -- screen-band mode, RGBA8 delivery: the bands travel resolved (4 bytes a pixel instead of 16) --
    def own_band_rgba8(self, background=(0.0, 0.0, 0.0), slot: int = 0):
        """Resolve this rank's band (``gsx_resolve_rgba8_device``) into one of two band buffers (`slot`: the caller
        alternates them so that a gather still in flight on another stream keeps its source).  Enqueued, no sync.
        int32 tensor [rows_per_rank * 16 * width], one RGBA8 pixel per word."""
        import torch

        v, lay = self.viewer, self._lay
        w = self._size[0]
        rows = lay.rows_per_rank * 16
        if getattr(self, "_band8", None) is None or self._band8[0].numel() != rows * w:
            self._band8 = [torch.zeros(rows * w, dtype=torch.int32, device=f"cuda:{self.device}") for _ in range(2)]
            self._frame8 = None
        bg = (C.c_float * 3)(*[float(x) for x in background])
        y0 = lay.band_offset_bytes // (16 * w)
        _lib.check(v._L.gsx_resolve_rgba8_device(v._h, bg, y0, y0 + rows, self._band8[slot].data_ptr()))
        return self._band8[slot]

    def gather_target_rgba8(self):
        """int32 tensor [world * rows_per_rank * 16 * width]: every rank's resolved band, in place."""
        import torch

        n = self._lay.padded_framebuffer_bytes // 16
        if getattr(self, "_frame8", None) is None or self._frame8.numel() != n:
            self._frame8 = torch.zeros(n, dtype=torch.int32, device=f"cuda:{self.device}")
        return self._frame8

    def frame_rgba8(self) -> np.ndarray:
        """(height, width, 4) uint8 of the gathered frame (synchronises the caller's streams first)."""
        w, h = self._size
        return self._frame8.cpu().numpy().view(np.uint8).reshape(-1, w, 4)[:h].copy()

    # -- frame-parallel mode: this rank renders whole frames; the resolved frames of all ranks are all-gathered --
    def own_frame_rgba8(self, background=(0.0, 0.0, 0.0), slot: int = 0):
        """Resolve the whole frame into one of two frame buffers (int32 [height * width]).  Enqueued, no sync."""
        import torch

        v = self.viewer
        w, h = self._size
        if getattr(self, "_own8", None) is None or self._own8[0].numel() != w * h:
            self._own8 = [torch.zeros(w * h, dtype=torch.int32, device=f"cuda:{self.device}") for _ in range(2)]
            self._frames8 = None
        bg = (C.c_float * 3)(*[float(x) for x in background])
        _lib.check(v._L.gsx_resolve_rgba8_device(v._h, bg, 0, h, self._own8[slot].data_ptr()))
        return self._own8[slot]

    def gather_target_frames_rgba8(self, world: int):
        """int32 tensor [world * height * width]: the frames the ranks rendered in this round, by rank."""
        import torch

        w, h = self._size
        if getattr(self, "_frames8", None) is None or self._frames8.numel() != world * w * h:
            self._frames8 = torch.zeros(world * w * h, dtype=torch.int32, device=f"cuda:{self.device}")
        return self._frames8

    def frames_rgba8(self) -> np.ndarray:
        """(world, height, width, 4) uint8 of the last gathered round."""
        w, h = self._size
        return self._frames8.cpu().numpy().view(np.uint8).reshape(-1, h, w, 4).copy()

    def own_band(self):
        """This rank's band of the framebuffer, copied out so the all-gather never aliases its own output."""
        lay = self._lay
        lo = lay.band_offset_bytes // 4
        self._band_t.copy_(self._fb_t[lo: lo + lay.band_bytes // 4])
        return self._band_t

    def gather_target(self):
        """The padded framebuffer itself: RCCL all-gathers every rank's band into place."""
        return self._fb_t

    # -- common --
    def framebuffer(self) -> np.ndarray:
        return self.viewer.download_framebuffer()

    def poll(self) -> None:
        self.viewer.poll()

    def set_pass_timing(self, on: bool, passes=None) -> None:
        self.viewer.set_pass_timing(on, passes)

    def get_pass_timing(self) -> dict:
        return self.viewer.get_pass_timing()

    def close(self) -> None:
        self.viewer.close()
