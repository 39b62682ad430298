"""Multi-GPU rendering: one process per MI355X, the Gaussian array sharded by splat index.

The reference is single-device (src/main.rs:85-98); this layer is new.  "over" blending is order
dependent per pixel, so per-GPU images of index shards cannot simply be summed.  The
path therefore has ONE real exchange step, sort-middle by screen rows:

  stage P  every rank projects / culls / colours its resident shard          (no communication)
  stage X  projected 48-byte records are routed to the rank(s) owning the tile rows they touch (the screen
           is cut into `world` contiguous bands of tile rows): RCCL all-to-all over xGMI, all 7 links busy
  stage C  every rank depth-sorts what it received, bins it into ITS band and composites it in
           progressive depth slabs; received records arrive ordered by (source rank, local index) =
           global index, so the stable sort breaks depth ties exactly like the single-GPU path
  stage M  the disjoint (rgb, T) bands are all-gathered straight into every rank's padded framebuffer.

Opaque scenes hide most splats, so stage X is speculative: every 16x16 tile has a depth-key limit — (1 + margin) x
the deepest depth at which its neighbourhood saturated last frame, unbounded where a neighbour stayed open — and a
record travels to a band only if some tile it touches there still takes it; the importing rank bins it into exactly
those tiles.  One small all-gather of the per-tile saturation depths verifies the limits and yields the next frame's;
only if a limited tile is still open do the records it was refused travel in a second exchange (`gsx_render_more`
composites them behind).  Whatever the limits, every tile composites a gap-free depth prefix, so the pixels are
bit-identical to the single-GPU frame; the limits only decide how many bytes cross xGMI (one link per GPU pair: at
2 GPUs the full 110 MB per frame would take ~2 ms).

A second partitioning is offered for scenes that fit one GPU's 288 GB many times over (10 M Gaussians = 8 GB):
``mode="screen"`` keeps the whole scene on every GPU and gives rank g the band g of tile rows; the frame then needs no
exchange step at all, only stage M.  Projection no longer scales with the GPU count (it is 0.12 ms of a 1.2 ms frame),
everything behind it does, and nothing waits on the host.

``torch.distributed`` (backend nccl = RCCL) is plumbing only: it moves buffers the HIP kernels packed.
The stage implementation is injectable (``stages=``) so the routing / merge logic is covered on CPU
with gloo at world_size 2 (tests/test_parallel_cpu.py) using the oracle as a checker.
"""
from __future__ import annotations

import numpy as np

RECORD_FLOATS = 12  # mean.xy, rect.xy (bits), conic.abc, opacity, rgb, depth  = 48 bytes
KEY_ALL = 0xFFFFFFFF  # exclusive upper bound of every valid depth key


def next_limits(sat: np.ndarray, margin: float, radius: int) -> np.ndarray:
    """Next frame's per-tile depth-key limit from this frame's saturation keys (uint32 [tiles_y, tiles_x], 0 = open):
    (1 + margin) x the deepest saturation depth in the tile's (2 radius + 1)^2 neighbourhood — the camera moves —
    and KEY_ALL (unbounded) if any tile of the neighbourhood stayed open.  Outside the frame counts as nothing."""
    sat = np.asarray(sat, np.uint32)
    d = np.where(sat != 0, sat.view(np.float32), np.float32(np.inf)).astype(np.float32)
    ty, tx = d.shape
    p = np.pad(d, ((radius, radius), (0, 0)), constant_values=np.float32(0.0))   # separable max filter: rows, then columns
    m = p[radius: radius + ty].copy()
    for dy in range(2 * radius + 1):
        np.maximum(m, p[dy: dy + ty], out=m)
    p = np.pad(m, ((0, 0), (radius, radius)), constant_values=np.float32(0.0))
    for dx in range(2 * radius + 1):
        np.maximum(m, p[:, dx: dx + tx], out=m)
    lim = (m * np.float32(1.0 + margin)).astype(np.float32)
    out = lim.view(np.uint32).copy()
    out[~np.isfinite(lim)] = KEY_ALL
    return np.maximum(out, 1)


def windows_first(limit: np.ndarray) -> np.ndarray:
    """Round 1: tile t admits the keys [0, limit[t])."""
    w = np.zeros(limit.shape + (2,), np.uint32)
    w[..., 1] = limit
    return w


def windows_second(limit: np.ndarray, need: np.ndarray) -> np.ndarray:
    """Round 2: the tiles in `need` admit what they were refused, [limit[t], KEY_ALL); every other tile nothing."""
    w = np.zeros(limit.shape + (2,), np.uint32)
    w[..., 0] = np.where(need, limit, 0)
    w[..., 1] = np.where(need, KEY_ALL, 0)
    return w


def model_render_keys(camera_pos, transforms: dict) -> list:
    """The app's far -> near model order: descending squared distance from the camera to each model's world centre,
    which is the model position (scene.rs:533-558, app.rs:1038-1046: `center` stays Vec3::ZERO)."""
    cp = np.asarray(camera_pos, np.float32)
    d = {k: float(((np.asarray(mt.pos, np.float32) - cp) ** 2).sum()) for k, mt in transforms.items()}
    return sorted(d, key=lambda k: -d[k])


def shard_range(n: int, rank: int, world: int):
    """Contiguous index shard [start, start+count) of rank; sizes differ by at most one."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class TorchComm:
    """The three collectives of the path on ``torch.distributed`` (nccl = RCCL over xGMI; gloo in the CPU tests)."""

    def __init__(self, group=None):
        self.group = group

    def all_to_all_slots(self, recv, send):
        """Fixed-size slots, equal split: slot p of `send` goes to rank p, slot p of `recv` comes from rank p."""
        import torch.distributed as dist

        dist.all_to_all_single(recv, send, group=self.group)

    def all_gather(self, out, inp):
        import torch.distributed as dist

        dist.all_gather_into_tensor(out, inp, group=self.group)


class LibComm(TorchComm):
    """The product transport: the collectives of the index-sharded frame run INSIDE libgsx over RCCL
    (``gsx_comm_all_to_all`` / ``gsx_comm_all_gather``, include/gsx.h), one communicator per viewer, bootstrapped like any
    NCCL program — rank 0 draws the unique id, ``torch.distributed`` only carries those 128 bytes to the other ranks.
    With it ``ShardedViewer`` renders an index-sharded frame through ONE library call, ``gsx_shard_render_frame``."""

    def __init__(self, stages, world, rank, group=None):
        super().__init__(group)
        import ctypes as C

        from . import _lib

        self._v = stages.viewer
        self._stream = stages.torch_stream
        L = self._v._L
        ident = [None]
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            _lib.check(L.gsx_comm_unique_id(buf))
            ident[0] = bytes(buf)
        if world > 1:
            import torch.distributed as dist

            dist.broadcast_object_list(ident, src=0, group=group)
        buf = (C.c_uint8 * 128).from_buffer_copy(ident[0])
        _lib.check(L.gsx_viewer_comm_init(self._v._h, world, rank, buf))

    def _on_viewer_stream(self):
        import torch

        return self._stream is None or torch.cuda.current_stream().cuda_stream == self._stream.cuda_stream

    def all_to_all_slots(self, recv, send):
        from . import _lib

        per_peer = send.numel() * send.element_size() // send.shape[0]
        _lib.check(self._v._L.gsx_comm_all_to_all(self._v._h, send.data_ptr(), recv.data_ptr(), per_peer))

    def all_gather(self, out, inp):
        from . import _lib

        if not self._on_viewer_stream():   # an overlapped gather on the caller's second stream: torch's communicator
            return super().all_gather(out, inp)
        _lib.check(self._v._L.gsx_comm_all_gather(self._v._h, inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size()))


class GroupComm(TorchComm):
    """The in-process transport of the library (``gsx_comm_group_*``, csrc/gsx_comm_group.cpp): `world` ranks of ONE process,
    one host thread and one viewer each; collectives are device copies inside libgsx.  ``group``: a ``viewer.CommGroup``
    shared by the ranks."""

    def __init__(self, stages, group, rank):
        super().__init__(None)
        self._v = stages.viewer
        self._v.comm_init_group(group, rank)

    def all_to_all_slots(self, recv, send):
        from . import _lib

        per_peer = send.numel() * send.element_size() // send.shape[0]
        _lib.check(self._v._L.gsx_comm_all_to_all(self._v._h, send.data_ptr(), recv.data_ptr(), per_peer))

    def all_gather(self, out, inp):
        from . import _lib

        _lib.check(self._v._L.gsx_comm_all_gather(self._v._h, inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size()))


class ShardedViewer:
    """One rank's view of a scene rendered by ``world`` GPUs.  With world == 1 (and use_dist False) this is
    exactly the single-GPU ``MultiModelViewer`` protocol."""

    KEY = "shard"

    def __init__(self, device: int = 0, world: int = 1, rank: int = 0, use_dist: bool = False, stream=None,
                 stages=None, group=None, sh: int = 0, cov3d: int = 0, comm=None, mode: str = "index",
                 gather: str = "float", overlap_gather: bool = False, background=(0.0, 0.0, 0.0)):
        """mode "index": every rank holds an index shard of the Gaussians, projected records are exchanged (module doc).
        mode "screen": every rank holds the WHOLE scene (``load_shard(all, 0, n)``) and renders one band of tile rows;
        the only collective is the all-gather of the bands.
        gather (screen mode): "float" — the (rgb, T) bands, 16 bytes a pixel, ``framebuffer()`` as on one GPU; "rgba8" — every
        rank resolves its band against ``background`` first (the app's blit to its Rgba8Unorm surface) and 4 bytes a pixel
        travel; the frame is ``frame_rgba8()``.  overlap_gather (rgba8): the all-gather runs on a second stream, under the
        next frame's projection and sorting — the gathered frame is complete after ``poll()``.
        mode "frames" (frame-parallel): the whole scene on every rank, and every rank renders WHOLE frames — in a round of
        ``world`` consecutive frames rank g renders frame g (``render_frame`` gets this rank's own camera).  The resolved
        RGBA8 frames of the round are all-gathered (``frames_rgba8()``: any rank, e.g. the one that owns the display, has
        every frame).  Throughput scales with the GPU count, the latency of one frame stays that of one GPU; the temporal
        speculation of a rank looks ``world`` frames back instead of one."""
        if mode not in ("index", "screen", "frames"):
            raise ValueError(mode)
        if mode == "frames":
            gather = "rgba8"
        if gather not in ("float", "rgba8") or (gather == "rgba8" and mode == "index"):
            raise ValueError(f"gather={gather!r} with mode={mode!r}")
        self.mode, self.gather, self.overlap_gather, self.background = mode, gather, bool(overlap_gather), tuple(background)
        self._frame_no = 0
        self._comm_stream = None
        self._gather_done = [None, None]
        self.world, self.rank, self.use_dist = world, rank, use_dist
        if stages is None:
            from .hip_stages import HipStages  # the product path: libgsx.so, fails loudly if missing

            stages = HipStages(device=device, stream=stream, use_torch=use_dist, sh=sh, cov3d=cov3d)
        self.stages = stages
        # the transport is injectable (tests drive `world` ranks as threads of one process, or over gloo on CPU); "lib" = the
        # collectives inside libgsx over RCCL, the index-sharded frame as one library call
        if comm == "lib":
            self.comm = LibComm(stages, world, rank, group)
        elif hasattr(comm, "_h") and hasattr(comm, "world"):   # a viewer.CommGroup: the library's in-process transport
            self.comm = GroupComm(stages, comm, rank)
        else:
            self.comm = comm if comm is not None else TorchComm(group)
        if mode == "frames" and world >= 4 and hasattr(stages, "viewer"):
            # a rank's previous frame is `world` poses back: wider windows (measured on cfg4, one GPU rendering every 8th / 4th
            # pose: margin 0.25 / radius 3 -> 638 / 869 fps, every frame needs the repair round at 8; 0.5 / 6 -> 765 / 936)
            stages.viewer.set_render_options(spec_margin=0.5, spec_radius=6)
        self._limit = None  # uint32 [tiles_y, tiles_x]: limits to use INSTEAD of the ones the last frame left on the device (tests); consumed by the next frame
        self.speculate = True
        # measured on cfg4 at 8 ranks (tools/emulate_ranks.py): margin 0.5 / radius 3 moves 3.6 MB per rank and frame instead
        # of 51.6 MB and needs the second exchange in 18 % of the frames (tiles that open up from nothing: unpredictable)
        self.margin = 0.25  # how far behind last frame's saturation depth a tile still takes records (world 1 over RCCL, cfg4:
                            # 962 / 1044 / 1066 fps at 0.5 / 0.35 / 0.25 — the repair round is sized exactly, so a tighter
                            # window costs a repair now and then, never a truncated slot)
        self.radius = 3     # tiles; neighbourhood over which the saturation depth is maximised (camera motion)
        self._rounds_host = None
        self.force_slot = None   # tests: a round-0 slot size instead of the library's policy (to provoke the overflow redo)
        self.last_verdict = None
        self.keys = []      # models in load order
        self.profile = None  # set to {} to collect host wall time per protocol section (adds device syncs)
        self._size = (1, 1)

    # -- scene --
    def load_shard(self, gaussians: np.ndarray, start: int, n_total: int, key: str | None = None) -> None:
        """This rank's index shard [start, start + len) of model ``key`` (several models: call once per key)."""
        key = key or self.KEY
        self._shard_max = getattr(self, "_shard_max", {})
        self._shard_max[key] = (int(n_total) + self.world - 1) // self.world if self.mode == "index" else int(n_total)
        self.stages.load_shard(key, gaussians, start, n_total)
        if key not in self.keys:
            self.keys.append(key)

    # -- frame --
    def render_frame(self, camera, size, model_transform=None, gaussian_transform=None, keys=None, transforms=None):
        """One frame; afterwards ``framebuffer()`` holds the complete (rgb, T) image on every rank.

        One model: ``model_transform`` is its TRS.  Several models: ``keys`` lists them far -> near — the app's
        ``model_render_keys`` (scene.rs:533-558), see ``model_render_keys`` below — and ``transforms[key]`` their TRS;
        models are layered, never merged, exactly as on one GPU."""
        st = self.stages
        keys = list(keys) if keys is not None else list(self.keys[:1] or [self.KEY])
        transforms = dict(transforms or {})
        if model_transform is not None and len(keys) == 1:
            transforms.setdefault(keys[0], model_transform)
        for k in keys:
            st.set_uniforms(k, camera, size, transforms.get(k), gaussian_transform)
        self._size = (int(size[0]), int(size[1]))
        if not self.use_dist:
            if len(keys) == 1:
                st.render_local(keys[0])
            else:
                st.render_local_keys(keys)
            return
        with st.stream_ctx():
            if self.mode == "frames":
                st.render_local_keys(keys)
                self.rounds = 0
                self._gather_rgba8()
            elif self.mode == "screen":
                # every rank has every Gaussian: render band `rank`, gather the bands — nothing else crosses the links and
                # nothing waits on the host (speculation, layered models, edits: all as on one GPU, per band)
                st.render_band(keys, self.world, self.rank)
                self.rounds = 0
                if self.gather == "rgba8":
                    self._gather_rgba8()
                else:
                    self.comm.all_gather(st.gather_target(), st.own_band())
            else:
                self._render_frame_dist(keys)

    def _tick(self, name):
        """Dev aid (self.profile = {}): host wall time per protocol section, with a device sync at every boundary."""
        if self.profile is None:
            return
        import time

        self.stages.poll()
        now = time.perf_counter()
        self.profile[name] = self.profile.get(name, 0.0) + (now - self._t_last)
        self._t_last = now

    def _render_frame_dist(self, keys):
        """One index-sharded frame, device-resident protocol (include/gsx.h "multi-GPU, device-resident protocol"): every stage
        call enqueues, every collective moves fixed-size buffers, and what the host waits for is one verdict per model — two
        pinned words every rank derives from the same gathered data.  ``keys`` far -> near; models are layered, never merged
        (scene.rs:533-558, 2302-2314): they are exchanged and composited nearest first, each behind the ones before it, and a
        model's repair round comes before the next model's records.  With the library transport the whole sequence is ONE
        call, gsx_shard_render_frame_keys (csrc/gsx_shard_frame.cpp), which this method mirrors step for step."""
        st, comm = self.stages, self.comm
        world, rank = self.world, self.rank
        if self.profile is not None:
            import time

            st.poll()
            self._t_last = time.perf_counter()
        limit, self._limit = self._limit, None
        self._rounds_host = None
        order = list(reversed(keys))                         # compositing order: nearest model first
        shard_max = [self._shard_max[k] for k in order]
        if isinstance(comm, (LibComm, GroupComm)) and limit is None and self.profile is None and self.force_slot is None:
            st.render_frame_lib(list(keys), [self._shard_max[k] for k in keys], self.speculate, self.margin, self.radius)
            return

        def exchange_round(i, rnd, slot):
            key = order[i]
            send = st.pack_slots(key, world, rnd, slot)
            self._tick("pack")
            recv = st.alloc_slots(world, slot, rnd)
            comm.all_to_all_slots(recv, send)                # stage X: fixed slots, the counts ride in the headers
            self._tick("exchange")
            st.import_slots(key, recv, world, rank, rnd, slot, behind=i > 0)   # stage C: import + depth sort + this rank's band
            self._tick("import_sort_render")
            mine = st.feedback(key, world, rank)
            sat_all = st.alloc_sat(world, mine)
            comm.all_gather(sat_all, mine)                   # verification AND the source of the next frame's limits
            self._tick("feedback")
            return sat_all

        def round0(i, slot):
            sat_all = exchange_round(i, 0, slot)
            seq = st.verify(order[i], world, sat_all)        # repair windows on the device; posts the verdict
            st.next_windows(order[i], world, sat_all, self.margin, self.radius)   # enqueued BEFORE the wait: what follows when all is well
            return seq

        def settle(i, seq):
            """the verdict of model i's round 0: True = a slot overflowed; a needed repair round is run here"""
            verdict = st.wait_verdict(order[i], seq)
            self.last_verdict = verdict
            if verdict["overflow"]:
                return True
            if verdict["need_tiles"]:
                mine = st.repair_count(order[i], world)      # the repair round is sized exactly: count, gather, post, wait
                counts_all = st.alloc_counts(world)
                comm.all_gather(counts_all, mine)
                sized = st.wait_verdict(None, st.post_counts(world, counts_all))
                sat_all = exchange_round(i, 1, max(sized["max_records"], 1))
                st.next_windows(order[i], world, sat_all, self.margin, self.radius)
                self._rounds_host = 2
            return False

        # stage P: windows [0, limit) from each model's last frame (first frame / speculation off: none), project the shards
        for k in order:
            st.frame_begin(k, world, rank, self.speculate, limit.get(k) if isinstance(limit, dict) else limit)
        self._tick("project")
        slots = [self.force_slot if self.force_slot is not None else st.slot_records(k, world, m) for k, m in zip(order, shard_max)]
        self._rounds_host = 1
        for attempt in (0, 1):
            overflow = False
            for i in range(len(order)):
                overflow = settle(i, round0(i, slots[i]))
                if overflow:
                    break
            if not overflow:
                break
            if attempt == 1:
                raise RuntimeError(f"an exchange slot of {slots} records (whole shards) overflowed")
            slots = list(shard_max)                          # a destination can be sent at most a whole shard: always fits
        # stage M: the disjoint bands of tile rows are all-gathered straight into the framebuffer
        comm.all_gather(st.gather_target(), st.own_band())
        self._tick("gather")
        for k in order:
            st.frame_end(k)

    @property
    def rounds(self):
        """Exchange rounds of the last index-sharded frame (the library call does not tell the host: statistics, synchronises)."""
        if not self.use_dist or self.mode != "index":
            return 0
        if self._rounds_host is not None:
            return self._rounds_host
        return 2 if self.last_stats().get("n_repair_tiles", 0) > 0 else 1

    @rounds.setter
    def rounds(self, value):
        self._rounds_host = value if value else None

    def skip_frame(self) -> None:
        """mode "frames": this rank has no frame in the round (the job's frame count is not a multiple of the GPU count) but
        the round's gather is a collective: join it with the frame this rank rendered last."""
        if self.mode != "frames" or not self.use_dist:
            return
        with self.stages.stream_ctx():
            self._gather_rgba8()

    def _gather_rgba8(self):
        """Resolve this rank's band and all-gather the RGBA8 bands; with ``overlap_gather`` on a second stream, so that the
        links work under the next frame's projection (two band buffers: a band is rewritten two frames later, after the
        gather that read it)."""
        st = self.stages
        slot = self._frame_no & 1
        self._frame_no += 1
        overlap = self.overlap_gather and getattr(st, "torch_stream", None) is not None
        frames = self.mode == "frames"
        own = (lambda: st.own_frame_rgba8(self.background, slot)) if frames else (lambda: st.own_band_rgba8(self.background, slot))
        target = (lambda: st.gather_target_frames_rgba8(self.world)) if frames else st.gather_target_rgba8
        if not overlap:
            self.comm.all_gather(target(), own())
            return
        import torch

        main = st.torch_stream
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=st.device)
        if self._gather_done[slot] is not None:
            main.wait_event(self._gather_done[slot])   # the gather of two frames ago has read this band buffer
        band = own()
        out = target()
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self._comm_stream):
            self._comm_stream.wait_event(ready)
            self.comm.all_gather(out, band)
            done = torch.cuda.Event()
            done.record(self._comm_stream)
        self._gather_done[slot] = done

    def framebuffer(self) -> np.ndarray:
        return self.stages.framebuffer()

    def frame_rgba8(self) -> np.ndarray:
        """(height, width, 4) uint8: the gathered frame of ``gather="rgba8"`` (synchronises)."""
        self.poll()
        return self.stages.frame_rgba8()

    def frames_rgba8(self) -> np.ndarray:
        """mode "frames": (world, height, width, 4) uint8, the frames of the last round by rank (synchronises)."""
        self.poll()
        return self.stages.frames_rgba8()

    def poll(self) -> None:
        self.stages.poll()
        if self._comm_stream is not None:
            self._comm_stream.synchronize()

    def last_stats(self, key: str | None = None) -> dict:
        """Statistics of the last frame (synchronises)."""
        return dict(self.stages.stats(key or (self.keys[0] if self.keys else self.KEY)))

    def set_pass_timing(self, on: bool, passes=None) -> None:
        self.stages.set_pass_timing(on, passes)

    def get_pass_timing(self) -> dict:
        return self.stages.get_pass_timing()

    def close(self) -> None:
        self.stages.close()
