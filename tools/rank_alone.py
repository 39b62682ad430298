"""What would ONE rank of an N-GPU sharded frame cost if it had a GPU to itself?  (a falsifiable prediction
of the 2 / 4 / 8-GPU rate from a one-GPU box.)

No reference counterpart (the reference renders on one wgpu device, src/main.rs:85-98); the design is SURVEY.md 8(e).

Phase 1, record: `world` ranks run as host threads on the one GPU, each with its index shard, each calling the library's own
frame loop (gsx_shard_render_frame) over a transport of THIS file (gsx_viewer_comm_init_custom_v: device copies between the ranks'
buffers, delivered in rank order like RCCL).  Besides delivering, the transport keeps, per rank, every piece that rank RECEIVED in
every collective of every frame — slots, feedback, repair counts, bands.
Phase 2, replay: one rank at a time, ALONE on the GPU, runs the same frames through the same library call; its transport now
serves the recorded pieces as device copies (the rank's own pieces are copied as the library asks).  Its behaviour is that of the
N-rank run — same shard, same limits, same slots, same verdicts, same band — and the frames it renders are compared with the
recording's (band checksums).  What is timed is the rank's wall time per frame with nothing else on the device: projection of
its shard, pack, import, sort, bin, composite of its band, verification — everything but the time the bytes spend on the links,
which is added as wire bytes / (7 links x 153 GB/s) (MI355X_MICROARCH.md: xGMI point-to-point).

Output: one JSON object — per world, scene and schedule: every rank's ms per frame alone, its wire bytes, list entries, the
band edges; and predicted fps = 1 / (max over ranks (ms alone) + max wire time).  The driver's 8-GPU SCALE run can prove it wrong.

usage: python tools/rank_alone.py [--worlds 2,4,8] [--frames 40] [--workload cfg4] [--scenes orbit,open_sky] [--out file.json]"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, parallel, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

# The block size of the block lists is pinned for both phases: left to itself the library picks it per model from the longest walk of an
# earlier frame (gsx_frame.cpp), i.e. from statistics that arrive when the device gets to them — at other frames in an 8-thread recording
# than in a solo replay — and the per-row work the bands are balanced by depends on it: the replayed rank would plan other band edges than
# the recording and ask its transport for pieces that were never recorded.  (A real N-GPU run needs no such pin: every rank reads the same
# gathered figures whatever each rank's block size was.)
os.environ.setdefault("GSX_BLOCKS_MAX", "256")
XGMI_LINK_GBPS, XGMI_LINKS = 153.0, 7
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
D2D = 3


def dmalloc(n):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), max(int(n), 1)) == 0
    return p.value


class Hub:
    """Phase 1: the ranks' meeting point (host threads of one process)."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=300.0)
        self.pub = [None] * world
        self.log = [[] for _ in range(world)]   # per rank: [(kind, [(offset, bytes, device copy of the piece) per source])]


class RecordingTransport:
    def __init__(self, hub, rank):
        self.hub, self.rank = hub, rank

    def _meet(self, stream, item):
        hip.hipStreamSynchronize(stream)   # what this rank contributes exists
        self.hub.pub[self.rank] = item
        self.hub.barrier.wait()

    def all_to_all_v(self, d_send, so, sb, d_recv, ro, rb, stream):
        h, me = self.hub, self.rank
        self._meet(stream, (d_send, so, sb))
        pieces = []
        for p in range(h.world):
            ps, pso, psb = h.pub[p]
            assert psb[me] == rb[p], f"rank {p} has {psb[me]} bytes for rank {me}, which expects {rb[p]}"
            keep = dmalloc(rb[p])
            if rb[p]:
                hip.hipMemcpyAsync(d_recv + ro[p], ps + pso[me], rb[p], D2D, stream)
                hip.hipMemcpyAsync(keep, ps + pso[me], rb[p], D2D, stream)
            pieces.append((ro[p], rb[p], keep))
        hip.hipStreamSynchronize(stream)
        h.log[me].append(("a2a", pieces))
        h.barrier.wait()   # every reader is done with the peers' send buffers
        return 0

    def gather_v(self, d_send, n, d_recv, ro, rb, root, stream):
        h, me = self.hub, self.rank
        self._meet(stream, (d_send, n))
        pieces = []
        if root < 0 or root == me:
            for p in range(h.world):
                ps, pn = h.pub[p]
                assert pn == rb[p]
                keep = dmalloc(rb[p])
                if rb[p]:
                    if ps != d_recv + ro[p]:
                        hip.hipMemcpyAsync(d_recv + ro[p], ps, rb[p], D2D, stream)
                    hip.hipMemcpyAsync(keep, ps, rb[p], D2D, stream)
                pieces.append((ro[p], rb[p], keep))
            hip.hipStreamSynchronize(stream)
        h.log[me].append(("gather", pieces))
        h.barrier.wait()
        return 0


_RT = None


def replay_lib():
    """tools/libreplay_transport.so (tools/replay_transport.cpp, built here on first use): the replay transport as native code"""
    global _RT
    if _RT is None:
        here = os.path.dirname(os.path.abspath(__file__))
        so, src = os.path.join(here, "libreplay_transport.so"), os.path.join(here, "replay_transport.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            import subprocess
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", so])
        _RT = C.CDLL(so)
        _RT.rt_create.restype = C.c_void_p
        _RT.rt_create.argtypes = [C.c_uint32, C.c_uint32]
        _RT.rt_destroy.argtypes = [C.c_void_p]
        _RT.rt_add_call.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _RT.rt_rewind.argtypes = [C.c_void_p]
        _RT.rt_wire.restype = C.c_uint64
        _RT.rt_wire.argtypes = [C.c_void_p]
        _RT.rt_position.restype = C.c_uint64
        _RT.rt_position.argtypes = [C.c_void_p]
        _RT.rt_failed.argtypes = [C.c_void_p]
    return _RT


class NativeReplay:
    """Phase 2, native: the recorded pieces of ONE rank handed to tools/replay_transport.cpp; the library calls its two C functions
    directly (gsx_viewer_comm_init_custom_v) — no Python inside the timed rank."""

    def __init__(self, world, rank, log):
        self.lib = replay_lib()
        self.ctx = self.lib.rt_create(world, rank)
        for kind, pieces in log:
            n = len(pieces)
            off = (C.c_uint64 * max(n, 1))(*[p[0] for p in pieces])
            nb = (C.c_uint64 * max(n, 1))(*[p[1] for p in pieces])
            keep = (C.c_uint64 * max(n, 1))(*[p[2] or 0 for p in pieces])
            self.lib.rt_add_call(self.ctx, 0 if kind == "a2a" else 1, n, off, nb, keep)

    def attach(self, v, world, rank):
        self.lib.rt_rewind(self.ctx)
        v.comm_init_custom_v_native(world, rank, self.lib.rt_all_to_all_v, self.lib.rt_gather_v, self.ctx)

    @property
    def wire(self):
        return int(self.lib.rt_wire(self.ctx))

    def close(self):
        self.lib.rt_destroy(self.ctx)


class ReplayTransport:
    """Phase 2 in Python (--python-replay: the A/B of the native transport): the recorded pieces of ONE rank, served in the order the
    library asks for them."""

    def __init__(self, world, rank, log):
        self.world, self.rank, self.log, self.at = world, rank, log, 0
        self.wire = 0

    def _next(self, kind):
        k, pieces = self.log[self.at]
        assert k == kind, f"replay out of step: call {self.at} was recorded as {k}, the library now asks for {kind}"
        self.at += 1
        return pieces

    def all_to_all_v(self, d_send, so, sb, d_recv, ro, rb, stream):
        for p, (off, n, keep) in enumerate(self._next("a2a")):
            assert off == ro[p] and n == rb[p], "replay: the rank sizes its slots differently from the recording"
            if p == self.rank:
                if n:
                    hip.hipMemcpyAsync(d_recv + off, d_send + so[p], n, D2D, stream)   # its own slot: a device copy, as over RCCL
            elif n:
                hip.hipMemcpyAsync(d_recv + off, keep, n, D2D, stream)
        self.wire += sum(b for p, b in enumerate(sb) if p != self.rank)
        return 0

    def gather_v(self, d_send, n, d_recv, ro, rb, root, stream):
        pieces = self._next("gather")
        if root < 0 or root == self.rank:
            for p, (off, m, keep) in enumerate(pieces):
                assert off == ro[p] and m == rb[p]
                if p == self.rank:
                    if m and d_send != d_recv + off:
                        hip.hipMemcpyAsync(d_recv + off, d_send, m, D2D, stream)
                elif m:
                    hip.hipMemcpyAsync(d_recv + off, keep, m, D2D, stream)
        self.wire += (self.world - 1) * n if root < 0 else (0 if root == self.rank else n)
        return 0


BALANCE = True   # --balance 0: equal bands (the A/B of the balanced ones)
ORBIT, KEYS_OF = [], []
SHARDS = {}   # (world, rank) -> the shard's Gaussians (generated once per world)


CFG5_TRS = {"a": camera.ModelTransform(pos=np.array([0.0, 0.0, 2.5], np.float32)),
            "b": camera.ModelTransform(pos=np.array([2.0, 0.2, -1.0], np.float32), rot=np.array([0, 35, 0], np.float32)),
            "c": camera.ModelTransform(pos=np.array([-2.5, -0.1, -0.5], np.float32), scale=np.array([0.9, 0.9, 0.9], np.float32)),
            "d": camera.ModelTransform(pos=np.array([0.3, -0.2, 0.5], np.float32), rot=np.array([20, -35, 50], np.float32),
                                       scale=np.array([1.2, 0.9, 1.1], np.float32))}   # tools/bench_cfg5.py's scene (BASELINE.json configs[4])
WORKLOAD = "cfg4"
PYTHON_REPLAY = False
PASS_REPLAY = True
REPLAYS = 2
MARGIN, RADIUS = 0.25, 3


def make_viewer(cfg, rank, world, open_sky, lanes=1):
    """-> (viewer, frame(i, speculate, sharded=True), models): the rank's index shard(s) resident, uniforms set by frame()."""
    n, sh, w, h, seed = cfg
    v = MultiModelViewer()
    v.set_render_options(frames_in_flight=lanes)
    orbit = ORBIT
    if WORKLOAD == "cfg5":   # 4 layered models x n / 4 Gaussians, each index-sharded over the ranks; TRS per model, `0 - 1` mask on one
        nm = n // 4
        s0, c = parallel.shard_range(nm, rank, world)
        for i, k in enumerate(CFG5_TRS):
            if (world, rank, k) not in SHARDS:
                SHARDS[(world, rank, k)] = scene.synthetic_gaussians(nm, seed + i, sh, s0, c)
            v.add_model(k, c)
            v.models[k].gaussian_buffers.gaussians_buffer.update_range(0, SHARDS[(world, rank, k)])
            v.update_model_transform(k, CFG5_TRS[k].pos, CFG5_TRS[k].quat(), CFG5_TRS[k].scale)
        shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([3.0, 3.0, 3.0], np.float32)),
                  MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([1.5, 1.5, 1.5], np.float32))]
        MaskEvaluator(v).evaluate(MaskOp.parse("0 - 1"), "a", shapes)
        keys_of = KEYS_OF
        shard_max = [(nm + world - 1) // world] * 4
        models = list(CFG5_TRS)

        def frame(i, speculate, sharded=True):
            v.update_camera(orbit[i % 240], (w, h))
            if sharded:
                v.shard_render_frame_keys(keys_of[i % 240], shard_max, speculate=bool(speculate), margin=MARGIN, radius=RADIUS)
            else:
                v.render_frame(keys_of[i % 240])
    else:
        s0, c = parallel.shard_range(n, rank, world)
        if (world, rank) not in SHARDS:
            SHARDS[(world, rank)] = scene.synthetic_gaussians(n, seed, sh, s0, c)
        v.add_model("m", c)
        v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, SHARDS[(world, rank)])
        if open_sky:   # bench.py's robustness scene: the Gaussians with y <= 0.5 (the screen above the horizon stays open)
            sky = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, -4.5, 0.0], np.float32), scale=np.array([10.0, 5.0, 10.0], np.float32))]
            MaskEvaluator(v).evaluate(MaskOp.parse("0"), "m", sky)
        shard_max = (n + world - 1) // world
        models = ["m"]

        def frame(i, speculate, sharded=True):
            v.update_camera(orbit[i % 240], (w, h))
            if sharded:
                v.shard_render_frame("m", shard_max, speculate=bool(speculate), margin=MARGIN, radius=RADIUS)
            else:
                v.render_frame(["m"])
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
    v.shard_set_gather_root(0)
    if not BALANCE:
        v.shard_set_balance(False)   # the A/B: equal bands of tile rows, as until round 3
    return v, frame, models


def band_checksum(v, world, rank):
    e = v.shard_get_band_edges(world)
    fb = v.download_framebuffer()
    lo, hi = 16 * int(e[rank]), min(16 * int(e[rank + 1]), fb.shape[0])
    return int(np.frombuffer(fb[lo:hi].tobytes(), np.uint32).astype(np.uint64).sum() & 0xFFFFFFFFFFFF), e.tolist()


def run(cfg, world, frames, open_sky, speculate, orbit, lanes):
    n, sh, w, h, seed = cfg
    hub = Hub(world)
    out = [None] * world
    errors = []
    warm = max(1, min(10, frames // 4))
    # frames that are LOOKED at (a readback completes the frames in flight, and redoes a frame whose slots overflowed): the same ones in
    # the recording and in the replay, or the two would issue different collectives — the untimed ones (one frame in flight) and the last
    looked_at = lambda i: (lanes == 1 and i < warm) or i == frames - 1  # noqa: E731

    def record(rank):
        try:
            v, frame, models = make_viewer(cfg, rank, world, open_sky, lanes)
            v.comm_init_custom_v(world, rank, RecordingTransport(hub, rank).all_to_all_v, RecordingTransport(hub, rank).gather_v)
            sums = []
            per_frame = []
            for i in range(frames):
                if i == warm:
                    v.poll()   # (the replay starts its clock here: that completes the frames in flight — the same calls in both phases)
                frame(i, speculate)
                sums.append(band_checksum(v, world, rank) if looked_at(i) else None)
                if rank == 0:
                    stf = v.shard_stats()   # (host-side bookkeeping of the frames retired so far: it lags the loop by a frame or two)
                    per_frame.append(dict(slot_records_max=stf["last_slot_records"], repair_slot_records=stf["last_repair_slot_records"], redo=stf["redo_frames"],
                                          repair=stf["repair_frames"], wire_MB=round(stf["wire_bytes"] / 1e6, 2),
                                          work_busiest_over_mean=stf["last_work_permille"] / 1000.0))
            st = v.shard_stats()
            entries = sum(v.frame_stats(k)["n_tile_entries"] for k in models)
            v.close()
            out[rank] = dict(sums=sums, stats=st, entries=entries, per_frame=per_frame)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            hub.barrier.abort()

    th = [threading.Thread(target=record, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errors:
        raise [e for e in errors if not isinstance(e, threading.BrokenBarrierError)][0] if any(
            not isinstance(e, threading.BrokenBarrierError) for e in errors) else errors[0]

    def attach(v, rank):
        if PYTHON_REPLAY:
            tr = ReplayTransport(world, rank, hub.log[rank])
            v.comm_init_custom_v(world, rank, tr.all_to_all_v, tr.gather_v)
        else:
            tr = NativeReplay(world, rank, hub.log[rank])
            tr.attach(v, world, rank)
        return tr

    ranks = []
    for rank in range(world):   # phase 2: one rank at a time, nothing else on the GPU
        # the timed replay, REPLAYS times on a fresh viewer each (the recording is one fixed sequence of frames): the rank's time is the
        # fastest of them — one replay is 30-50 frames of 0.2-1.5 ms, and a host hiccup or a tuner probe phase that falls into the window
        # moved single cells of the table by 20-40 % from run to run
        elapsed, same, all_ms = None, True, []
        for rep in range(REPLAYS):
            v, frame, models = make_viewer(cfg, rank, world, open_sky, lanes)
            tr = attach(v, rank)
            t_mark = wire_mark = None
            for i in range(frames):
                if i == warm:
                    v.poll()
                    t_mark, wire_mark = time.perf_counter(), tr.wire
                frame(i, speculate)
                if looked_at(i) and i < frames - 1:   # (checking costs a readback: only untimed frames are checked, the last one after the clock stops)
                    same = same and band_checksum(v, world, rank) == out[rank]["sums"][i]
            v.poll()
            el = time.perf_counter() - t_mark
            all_ms.append(round(1e3 * el / (frames - warm), 4))
            elapsed = el if elapsed is None else min(elapsed, el)
            wire_timed = tr.wire - wire_mark
            same = same and band_checksum(v, world, rank) == out[rank]["sums"][frames - 1]
            st = v.shard_stats()
            v.close()
            if rep + 1 < REPLAYS and hasattr(tr, "close"):
                tr.close()
        timed = frames - warm
        launches = None
        # once more with every pass bracketed by events (costs a few microseconds of stream gap per bracket): where the rank's time goes;
        # and the kernel launches the library asks for per frame (GSX_LAUNCH counter)
        passes, tr2 = {}, None
        if PASS_REPLAY:
            from wgpu_3dgs_viewer_app_amd import viewer as viewer_mod
            v, frame, models = make_viewer(cfg, rank, world, open_sky, lanes)
            tr2 = attach(v, rank)
            l0 = 0
            for i in range(frames):
                if i == warm:
                    v.poll()
                    v.set_pass_timing(True)
                    v.get_pass_timing()
                    l0 = viewer_mod.launch_count()
                frame(i, speculate)
            launches = round((viewer_mod.launch_count() - l0) / timed, 1)
            v.poll()
            passes = {k: round(1e3 * t["ms"] / timed, 1) for k, t in v.get_pass_timing().items() if t["ms"] > 0}
        ranks.append(dict(rank=rank, ms_per_frame_alone=round(1e3 * elapsed / timed, 4), ms_per_frame_every_replay=all_ms, pass_us_per_frame=passes, launches_per_frame=launches,
                          wire_bytes_per_frame=int(wire_timed / timed), frames_equal_to_the_recording=bool(same),
                          list_entries_last_frame=int(out[rank]["entries"]),
                          repair_frames=round(st["repair_frames"] / max(st["frames"], 1), 3), redo_frames=round(st["redo_frames"] / max(st["frames"], 1), 3),
                          verdict_wait_us_per_frame=round(st["verdict_wait_ns"] / 1e3 / max(st["frames"], 1), 1),
                          band_rows_last_frame=[out[rank]["sums"][-1][1][rank], out[rank]["sums"][-1][1][rank + 1]]))
        if PASS_REPLAY:
            v.close()
        for t in (tr, tr2):
            if t is not None and hasattr(t, "close"):
                t.close()
    for lg in hub.log:
        for _, pieces in lg:
            for _, _, keep in pieces:
                hip.hipFree(keep)
    slowest = max(r["ms_per_frame_alone"] for r in ranks)
    wire_ms = max(r["wire_bytes_per_frame"] for r in ranks) / (XGMI_LINKS * XGMI_LINK_GBPS * 1e9) * 1e3
    ent = [r["list_entries_last_frame"] for r in ranks]
    return dict(world=world, scene="open_sky" if open_sky else "orbit", speculate=int(speculate), frames_in_flight=lanes, frames_timed=frames - warm, ranks=ranks,
                slowest_rank_ms=round(slowest, 4), fastest_rank_ms=round(min(r["ms_per_frame_alone"] for r in ranks), 4),
                wire_ms_at_7x153GBps=round(wire_ms, 4), predicted_fps=round(1e3 / (slowest + wire_ms), 1),
                list_entries_max_over_mean=round(max(ent) * world / max(sum(ent), 1), 3),
                band_edges_last_frame=out[0]["sums"][-1][1], rank0_frame_by_frame_cumulative=out[0]["per_frame"], all_frames_equal_to_the_recording=all(r["frames_equal_to_the_recording"] for r in ranks))


def single_gpu(cfg, frames, open_sky, speculate, orbit, lanes):
    """the same scene on the single-GPU path, same process: what N = 1 means in the table"""
    v, frame, _ = make_viewer(cfg, 0, 1, open_sky)
    v.set_render_options(speculative=int(speculate), frames_in_flight=lanes)
    for i in range(10):
        frame(i, speculate, sharded=False)
    v.poll()
    t0 = time.perf_counter()
    for i in range(10, 10 + frames):
        frame(i, speculate, sharded=False)
    v.poll()
    el = time.perf_counter() - t0
    v.close()
    return round(frames / el, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--scenes", default="orbit,open_sky")
    ap.add_argument("--unspeculated-frames", type=int, default=14)
    ap.add_argument("--lanes", default="1,2", help="frames in flight of the replayed rank (and of the recording)")
    ap.add_argument("--out", default="")
    ap.add_argument("--balance", type=int, default=1, help="0: equal bands of tile rows (gsx_shard_set_balance(0)) — the A/B of the balanced bands")
    ap.add_argument("--speculate", default="1,0", help="which schedules to run")
    ap.add_argument("--margin", type=float, default=0.25, help="gsx_shard_render_frame's margin (the speculated schedule)")
    ap.add_argument("--radius", type=int, default=3)
    ap.add_argument("--replays", type=int, default=2, help="timed replays per rank (the fastest counts)")
    ap.add_argument("--no-pass-replay", action="store_true", help="skip the second replay with the passes bracketed by events (for a kernel trace whose tail is the timed replay)")
    ap.add_argument("--python-replay", action="store_true", help="serve the replay from Python callbacks (rounds 4's transport: the A/B of the native one)")
    a = ap.parse_args()
    global BALANCE, WORKLOAD, PYTHON_REPLAY, ORBIT, KEYS_OF, PASS_REPLAY, MARGIN, RADIUS, REPLAYS
    BALANCE = bool(a.balance)
    WORKLOAD = a.workload
    PYTHON_REPLAY = a.python_replay
    PASS_REPLAY = not a.no_pass_replay
    MARGIN, RADIUS = a.margin, a.radius
    REPLAYS = max(1, a.replays)
    specs = [int(x) for x in a.speculate.split(",")]
    cfg = scene.CONFIGS[a.workload]
    n, sh, w, h, seed = cfg
    orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
    ORBIT = orbit
    KEYS_OF = [parallel.model_render_keys(camera.orbit_pose(k).pos, CFG5_TRS) for k in range(240)]
    if a.workload == "cfg5":
        a.scenes = "orbit"
    res = dict(tool="tools/rank_alone.py", workload=a.workload, gaussians=n, size=[w, h],
               method="every rank of an N-rank gsx_shard_render_frame run replayed ALONE on the GPU against the pieces it received in the "
                      "N-rank run (threads, one GPU), the fastest of --replays timed replays per rank; predicted fps = 1 / (slowest rank's ms alone + busiest rank's wire bytes / (7 x 153 GB/s)); "
                      + ("the replay transport's callbacks are Python (--python-replay)" if a.python_replay else
                         "the replay transport is native code (tools/replay_transport.cpp): no Python inside the timed rank"),
               balanced_bands=bool(a.balance), margin=a.margin, radius=a.radius, single_gpu_fps={}, runs=[])
    lanes_list = [int(x) for x in a.lanes.split(",")]
    for sc in a.scenes.split(","):
        for spec in specs:
            for lanes in lanes_list:
                res["single_gpu_fps"][f"{sc} speculate={spec} frames_in_flight={lanes}"] = single_gpu(cfg, max(a.frames, 120), sc == "open_sky", spec, orbit, lanes)
    SHARDS.clear()
    for world in [int(x) for x in a.worlds.split(",")]:
        SHARDS.clear()
        for sc in a.scenes.split(","):
            for spec, lanes in [(s, l) for s in specs for l in lanes_list]:
                # (speculate = 0: every visible record travels and a slot holds a whole shard — 60 MB x 8 x 8 per frame at world 8 on
                #  cfg4: the recording keeps fewer frames)
                frames = a.frames if spec else min(a.frames, a.unspeculated_frames)
                r = run(cfg, world, frames, sc == "open_sky", spec, orbit, lanes)
                res["runs"].append(r)
                print(f"world {world} {sc} speculate={spec} lanes={lanes}: predicted {r['predicted_fps']} fps (slowest rank {r['slowest_rank_ms']} ms, fastest "
                      f"{r['fastest_rank_ms']}, wire {r['wire_ms_at_7x153GBps']} ms, entries max/mean {r['list_entries_max_over_mean']}, equal "
                      f"{r['all_frames_equal_to_the_recording']})", file=sys.stderr, flush=True)
    txt = json.dumps(res)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
