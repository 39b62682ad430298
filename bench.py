#!/usr/bin/env python3
"""bench.py — frames/sec of the 3DGS render path on N MI355X (one process per GPU).

A "step" is one frame: the whole hot path (projection -> depth sort -> tile binning -> tile sort ->
composite [-> exchange/merge when N > 1]) over the resident synthetic scene at the next pose of the
benchmark orbit.  Inputs are resident in HBM before the timed region starts.
Prints ONE JSON line on rank 0.  The CPU oracle is used here only for the `cpu_baseline` leg.

    python bench.py                       # N=1, cfg4 scene (10 M Gaussians, SH-3, 1920x1080)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps K --warmup W

What one N=1 run measures, all in the same process on the same resident scene (nothing is read from profiles/):
  value                 fps of max(K, 240) frames — at least one whole orbit, whatever --steps is — in the default schedule
                        (progressive slabs + temporal occlusion speculation); value_short_window: the K-step window that was asked for.
                        Every other timed loop of the line is at least one orbit long as well ("steps_timed").
  value_unspeculated    fps with speculative = 0 (what every first frame / incoherent pose costs)
  value_synchronised    fps when the host waits for every frame: gsx_render_frame + gsx_sync (SURVEY 8d's definition of
                        the metric), and value_reference_protocol: the app's own sequence with its two blocking waits per frame —
                        preprocess + sort, poll, render, poll (src/tab/scene.rs:856-873, 613-614)
  steady_state          the headline loop again over >= 240 frames (a whole orbit), whatever --steps is
  passes                every pass of both schedules bracketed with HIP events in a loop of its own: microseconds per frame,
                        algorithmic bytes (BASELINE.md 4 / SURVEY 8d formulas, stated per pass), GB/s and fraction of 8 TB/s
  roofline              k_project<3,0,0> — the projection pass SURVEY 8d prices (SH colour + cov2d + cull + depth key),
                        HIP events around that kernel alone over the unspeculated timed loop, N*pod + N_vis*40 bytes
  roofline_speculated   k_project_geom — the geometry-only projection of the speculated loop, priced on what IT has to move
  roofline*.traffic     HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes) that this
                        script runs on short child invocations of itself before it touches the GPU (null if rocprofv3 is
                        missing or fails; --no-pmc skips them)
  frame_check           the last timed speculated frame, re-rendered with speculative = 0 and progressive = 0, must be
                        bit-identical; overflow_slabs must be 0 (no frame took the slow pair-free path)
  robustness            how much of `value` is the coherent orbit over an occluding scene: the same K-frame loops (a) with the
                        240 poses in a seeded random order — no temporal coherence — and (b) on the scene with an open sky (a
                        mask box keeps only the Gaussians below y = 0.5: the upper part of the screen never saturates), each
                        speculated and unspeculated, with the fraction of frames that needed the repair round
  summary               the figures above side by side, the last key before cpu_baseline (a truncated log tail keeps it)
  cpu_baseline          oracle/gsx_oracle.c, one frame of the WHOLE scene on all host cores
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "frames/sec @1920x1080, N-Gaussian SH3 scene, 1/2/4/8 MI355X; %HBM roofline"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--min-steps", type=int, default=240, help="every timed loop runs at least this many frames (a whole orbit) whatever "
                    "--steps says; `steps` on the line stays what was asked for, `steps_timed` says what was timed (tests pass 0)")
    ap.add_argument("--workload", default="cfg4", help="cfg2 (1 M) | cfg3 (5.8 M) | cfg4 (10 M, headline)")
    ap.add_argument("--gaussians", type=int, default=0, help="override the Gaussian count (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg5", action="store_true", help="N=1: skip the compact BASELINE configs[4] leg (4 x 6 M Gaussians at 3840x2160, ~10 s)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="Gaussians in the CPU baseline (0 = the whole scene)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC child passes (roofline.traffic = null)")
    ap.add_argument("--no-robustness", action="store_true", help="skip the speculation-robustness legs (random pose order, open sky)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # internal: the run rocprofv3 wraps
    ap.add_argument("--force-dist", action="store_true", help="run the sharded exchange path even at N=1")
    ap.add_argument("--shard-mode", default="index", help="N > 1: index = splat-index shards + speculative record exchange by screen "
                    "band (north_star's partition; the headline) | screen = whole scene on every GPU, rank g renders band g of tile "
                    "rows, one all-gather | frames = whole scene on every GPU, rank g renders every N-th frame (replicated, "
                    "frame-parallel; RGBA8 frames all-gathered)")
    ap.add_argument("--pose-stride", type=int, default=1, help="N = 1 experiment: render every S-th pose of the orbit — what one rank of "
                    "the frame-parallel mode at S GPUs sees (its speculation looks S poses back)")
    ap.add_argument("--pose-order", default="orbit", help="orbit | random (the same 240 poses in a seeded random order: no temporal "
                    "coherence, the speculation's worst case)")
    ap.add_argument("--gather", default="rgba8", help="N > 1, screen mode: rgba8 = every rank resolves its band (the app's blit to its "
                    "Rgba8Unorm surface) and 4 bytes a pixel are all-gathered on a second stream, under the next frame | float = the "
                    "(rgb, T) bands, 16 bytes a pixel, in stream order")
    ap.add_argument("--pass-timing", default="project", help="project (the roofline kernel only; default) | all (every pass, adds "
                    "a few microseconds of stream gap per pass boundary)")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="gsx_render_options.frames_in_flight of the headline loop at N=1 (the one-in-flight rate is reported beside it)")
    ap.add_argument("--dist-frames-in-flight", type=int, default=2, help="frames_in_flight of gsx_shard_render_frame (N > 1 / --force-dist): "
                    "every lane has its own communicator and stream, the verdict of a frame is read one call later")
    ap.add_argument("--no-scene-legs", action="store_true", help="robustness: skip the legs that upload another scene (translucent, surfaces)")
    ap.add_argument("--no-extra-legs", action="store_true", help="N=1: skip the synchronised / steady-state / per-pass legs")
    ap.add_argument("--unspeculated-in-flight", action="store_true",
                    help="N=1: also time the unspeculated loop with --frames-in-flight lanes (off by default: its contended k_project "
                         "launches would blur the kernel-trace average the roofline is checked against)")
    ap.add_argument("--render-options", default="", help="gsx_render_options overrides, e.g. speculative=0,min_slab=1000000 (experiments)")
    ap.add_argument("--host-profile", action="store_true", help="print host wall time per exchange-protocol section (adds syncs; debug)")
    ap.add_argument("--pod", default="single/single", help="pod storage sh/cov3d: single|half|norm8|none / single|half "
                    "(reference default is norm8/half; the headline metric is quoted on the f32 pod)")
    a = ap.parse_args()
    a.frames_in_flight_given = a.frames_in_flight is not None   # (an explicit value is taken as it is: no probe)
    if a.frames_in_flight is None:
        a.frames_in_flight = 2
    return a


def cpu_baseline(cfg, n_sample, pose=0, keep=False):
    """Oracle (reference algorithm shape: cull -> global radix sort -> back-to-front splat-major raster) on the
    host cores: one frame of the same scene (the whole scene unless --cpu-sample bounds it).  Timed: gsxo_project +
    gsxo_depth_sort + gsxo_rasterize — exactly what gsxo_render_model runs, called one by one so that `keep` can hand the depth keys,
    tile rectangles, depth order and frame to the oracle check (bench.py never routes a product result through them)."""
    import oracle
    from wgpu_3dgs_viewer_app_amd import camera, scene

    n, sh, w, h, seed = cfg
    n_sample = min(n, n_sample) if n_sample else n
    g = scene.synthetic_gaussians(n, seed, sh, 0, n_sample)
    pos, color, shc, cov = oracle.convert(g)
    del g
    cam = camera.orbit_pose(pose)
    f = oracle.frame_setup(cam.view(), cam.projection(w / h), w, h)
    fb = oracle.new_framebuffer(f)
    oracle.render_model(f, pos[:1000], color[:1000], shc[:1000], cov[:1000], fb)  # page in / thread start
    fb = oracle.new_framebuffer(f)
    t0 = time.perf_counter()
    pr = oracle.project(f, pos, color, shc, cov)
    idx, nvis = oracle.depth_sort(pr["key"])
    oracle.rasterize(f, pr, idx, nvis, fb)
    dt = time.perf_counter() - t0
    what = "the whole scene" if n_sample == n else f"the first {n_sample} of {n} Gaussians of the same scene"
    res = dict(value=round(1.0 / dt, 4), unit="frames/s", cores=oracle.num_threads(), kind="port",
               sample=f"1 frame (orbit pose {pose}) of {what} ({n_sample} Gaussians) at {w}x{h}, N_vis={nvis}, {dt:.2f} s; "
                      "oracle/gsx_oracle.c (cull -> LSD radix sort -> back-to-front splat-major 'over'), OpenMP")
    return (res, dict(key=pr["key"], rect=pr["rect"], order=idx[:nvis], n_visible=nvis, fb=fb) if n_sample == n else None) if keep else res


def oracle_check_of(gpu, ref, pose):
    """The HIP path's frame of the pose the CPU baseline rendered, against that very frame: integer stages bit-exact,
    the frame within north_star's 1e-3 per-channel L-infinity — from the plainest schedule AND from a speculated frame of the default one."""
    out = dict(pose=pose, tolerance=1e-3,
               n_visible_equal=bool(gpu["n_visible"] == ref["n_visible"]), n_visible=int(ref["n_visible"]),
               keys_equal=bool(np.array_equal(gpu["key"], ref["key"])), rects_equal=bool(np.array_equal(gpu["rect"], ref["rect"])),
               depth_order_equal=bool(np.array_equal(gpu["order"], ref["order"])),
               linf_plain=float(np.abs(gpu["fb_plain"] - ref["fb"]).max()), linf_speculated=float(np.abs(gpu["fb_spec"] - ref["fb"]).max()),
               speculated_frame_was_speculated=bool(gpu["speculated"]), speculated_equals_plain=bool(np.array_equal(gpu["fb_plain"], gpu["fb_spec"])),
               what="gsx_preprocess + gsx_sort + gsx_render (speculative = 0, progressive = 0) and the default schedule's frame of the same pose "
                    "(arrived at along the orbit, frames in flight as in `value`) against oracle/gsx_oracle.c's frame of the whole scene — the frame "
                    "cpu_baseline timed: depth keys + cull set, tile rectangles and depth order bit-exact, both frames within 1e-3 per channel")
    out["linf"] = max(out["linf_plain"], out["linf_speculated"])
    out["ok"] = bool(out["n_visible_equal"] and out["keys_equal"] and out["rects_equal"] and out["depth_order_equal"] and out["linf"] <= out["tolerance"])
    return out


def cfg5_leg(steps=60, warmup=12):
    """BASELINE.json configs[4] on this GPU, compact (tools/bench_cfg5.py is the long form): 4 models x 6 M Gaussians
    SH-3, each with its own TRS, a `0 - 1` mask (box minus ellipsoid, src/app.rs:1660-1783 grammar) on one, a stored rect selection with
    an HSV edit, 3840x2160, models layered far -> near by camera distance every frame (scene.rs:533-558)."""
    from wgpu_3dgs_viewer_app_amd import camera, parallel, query, scene
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind
    from wgpu_3dgs_viewer_app_amd import viewer as viewer_mod
    from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer

    t_all = time.perf_counter()
    n_total, sh, w, h, seed = scene.CONFIGS["cfg5"]
    n = n_total // 4
    tr = {"a": camera.ModelTransform(pos=np.array([0.0, 0.0, 2.5], np.float32)),
          "b": camera.ModelTransform(pos=np.array([2.0, 0.2, -1.0], np.float32), rot=np.array([0, 35, 0], np.float32)),
          "c": camera.ModelTransform(pos=np.array([-2.5, -0.1, -0.5], np.float32), scale=np.array([0.9, 0.9, 0.9], np.float32)),
          "d": camera.ModelTransform(pos=np.array([0.3, -0.2, 0.5], np.float32), rot=np.array([20, -35, 50], np.float32),
                                     scale=np.array([1.2, 0.9, 1.1], np.float32))}
    v = MultiModelViewer()
    try:
        for i, k in enumerate(tr):
            g = scene.synthetic_gaussians(n, seed + i, sh)
            v.add_model(k, n)
            v.models[k].gaussian_buffers.gaussians_buffer.update_range(0, g)
            v.update_model_transform(k, tr[k].pos, tr[k].quat(), tr[k].scale)
            del g
        shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([3.0, 3.0, 3.0], np.float32)),
                  MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([1.5, 1.5, 1.5], np.float32))]
        MaskEvaluator(v).evaluate(MaskOp.parse("0 - 1"), "a", shapes)
        orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
        keys_of = [parallel.model_render_keys(camera.orbit_pose(k).pos, tr) for k in range(240)]

        def frame(i):
            v.update_camera(orbit[i % 240], (w, h))
            v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
            v.render_frame(keys_of[i % 240])

        # the stored selection: a rectangle on the frame of pose 0, Set; then the HSV edit of what it selected (GaussianEditPod per Gaussian)
        v.update_query(query.QueryPod.rect((1200.0, 600.0), (2600.0, 1500.0), query.QuerySelectionOp.Set))
        frame(0)
        for k in keys_of[0]:
            v.postprocessor.postprocess(k)
        v.update_query(query.QueryPod.none())
        v.update_selection_edit_with_pod(query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.5, 1.0, 1.2), 0.1, 0.2, 1.0, 0.9))

        spilled = [0]   # depth slabs of the TIMED frames that did not fit the pair buffers (complete frames, slow path)

        def overflow_now():
            return int(sum(v.frame_stats(k)["overflow_slabs"] for k in tr))

        def loop(**opts):
            v.set_render_options(**opts)
            for i in range(warmup):
                frame(i)
            v.poll()
            ov0 = overflow_now()
            l0 = viewer_mod.launch_count()
            t0 = time.perf_counter()
            for i in range(warmup, warmup + steps):
                frame(i)
            v.poll()
            dt = time.perf_counter() - t0
            launched = viewer_mod.launch_count() - l0
            spilled[0] += overflow_now() - ov0
            return round(steps / dt, 1), round(launched / steps, 1)

        fps1, launches1 = loop()
        fps2, _ = loop(frames_in_flight=2)
        fb_two = v.download_framebuffer().copy()   # the last frame of the two-lane loop ...
        fpsu, launchesu = loop(speculative=0)
        v.set_render_options(speculative=0, progressive=0)
        frame(warmup + steps - 1)                  # ... and its pose through the plainest schedule
        equal = bool(np.array_equal(v.download_framebuffer(), fb_two))
        st = {k: v.frame_stats(k) for k in tr}
        res = dict(workload=f"cfg5: 4 x {n} Gaussians SH-3, {w}x{h}, TRS per model, mask '0 - 1' on one, stored rect selection + HSV edit; "
                            "gsx_render_frame(keys far -> near), one GPU",
                   fps_one_frame_in_flight=fps1, fps_two_frames_in_flight=fps2, fps_unspeculated=fpsu, steps=steps,
                   launches_per_frame=dict(one_frame_in_flight=launches1, unspeculated=launchesu),
                   frame_check=dict(equal_to_unspeculated_single_pass=equal, pose=(warmup + steps - 1) % 240),
                   n_visible=int(sum(s["n_visible"] for s in st.values())), overflow_slabs=int(spilled[0]),
                   resident_bytes=int(viewer_mod.device_bytes()))
    finally:
        v.close()
    res["leg_seconds"] = round(time.perf_counter() - t_all, 1)
    if not res["frame_check"]["equal_to_unspeculated_single_pass"]:
        raise SystemExit(f"bench.py: cfg5's two-lane frame differs from the speculative=0, progressive=0 frame: {res}")
    return res


# ------------------------------------------------------------------------------------------------
# HBM traffic of the projection kernels from PMC counters, measured by THIS run: rocprofv3 wraps two short child
# invocations of this script (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots")
# before the parent touches the GPU.  gfx950 correction per the guide: FETCH_SIZE tallies a wide coalesced streaming read
# at half its bytes -> doubled; WRITE_SIZE taken as reported.
# ------------------------------------------------------------------------------------------------
def predicted_for(world, workload):
    """What tools/rank_alone.py predicted for this world size from a ONE-GPU box (every rank of an N-rank frame replayed alone on the GPU
    against the pieces it received; profiles/r06_rank_alone.json, committed): the number this line's `value` can prove wrong."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_rank_alone.json")
    try:
        with open(path) as f:
            table = json.loads(f.read().strip().splitlines()[-1])
    except (OSError, ValueError):
        return None
    if table.get("workload") != workload:
        return None
    rows = {}
    for r in table["runs"]:
        if r["world"] == world and r["scene"] == "orbit":
            rows[f"speculate={r['speculate']} frames_in_flight={r['frames_in_flight']}"] = dict(
                fps=r["predicted_fps"], slowest_rank_ms_alone=r["slowest_rank_ms"], fastest_rank_ms_alone=r["fastest_rank_ms"],
                wire_ms_at_7x153GBps=r["wire_ms_at_7x153GBps"])
    return dict(source="profiles/r06_rank_alone.json (tools/rank_alone.py on a one-GPU box; read from that file, not measured in this run)",
                method=table.get("method"), single_gpu_fps_in_that_run=table.get("single_gpu_fps"), by_schedule=rows) if rows else None


def pmc_traffic(args):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="gsx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--pod", args.pod,
                   "--steps", "4", "--warmup", "3"]
            if args.gaussians:
                cmd += ["--gaussians", str(args.gaussians)]
            p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if p.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {p.returncode}: {p.stderr.decode(errors='replace')[-200:]}"
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {counter}: no counter_collection.csv"
            per = {}
            for r in csv.DictReader(open(files[0])):
                if r.get("Counter_Name") != counter:
                    continue
                k = r["Kernel_Name"]
                if "k_project" not in k:
                    continue
                variant = "speculated" if "k_project_geom" in k else "full"
                per.setdefault(variant, []).append(float(r["Counter_Value"]))
            for variant, vals in per.items():
                out.setdefault(variant, {})[counter] = (1024.0 * sum(vals) / len(vals), len(vals))  # KiB -> bytes per dispatch
    except Exception as e:  # noqa: BLE001 — the bench line must not die on a profiler problem
        return None, f"PMC pass failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for variant, c in out.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            res[variant] = dict(hbm_bytes_per_launch=2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0], fetch_size_bytes_raw=c["FETCH_SIZE"][0],
                                write_size_bytes=c["WRITE_SIZE"][0], dispatches=c["FETCH_SIZE"][1])
    return (res or None), ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on child runs of this script, same scene; "
                           "bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: wide coalesced reads are tallied at half their bytes)")


def main():
    args = parse_args()
    # stdout carries ONE JSON line and nothing else: libraries that print banners on fd 1 (RCCL's version block) go to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # Every timed loop of this file runs over at least one whole orbit (240 poses): a 20-step window misses the speculation tuner's
    # probe frames and the expensive quarter of the orbit, and read 13-16 % above the same run's whole-orbit rate in rounds 3 and 4.
    # "steps" on the line stays what the caller asked for; "steps_timed" is what every loop timed.
    steps_requested = args.steps
    if not args.pmc_child:
        args.steps = max(args.steps, args.min_steps)  # args.steps = max(args.steps, 240) by default

    # GSX_BENCH_ONE_DEVICE=1 (test aid, tests/test_gpu_bench_ranks.py): every rank uses device 0 and carries a host identity of its
    # own, so RCCL accepts N ranks on ONE GPU and connects them over its socket transport on `lo`.  The N > 1 code of this file and
    # of libgsx then runs for real, between processes; the rate it prints measures nothing and the line says so.
    one_device = world > 1 and os.environ.get("GSX_BENCH_ONE_DEVICE") == "1"
    if one_device:
        os.environ.update({"NCCL_HOSTID": f"gsx-bench-rank-{rank}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "NCCL_NET": "Socket"})
        local_rank = 0

    # PMC child passes first: nothing in this process has initialised the GPU yet
    traffic, traffic_note = None, "skipped"
    single = world == 1 and not args.force_dist
    if single and not args.no_pmc and not args.pmc_child and not args.render_options:
        traffic, traffic_note = pmc_traffic(args)

    import torch
    import torch.distributed as dist

    from wgpu_3dgs_viewer_app_amd import camera, parallel, scene
    from wgpu_3dgs_viewer_app_amd import viewer as viewer_mod

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libgsx has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n, sh, w, h, seed = scene.CONFIGS[args.workload]
    if args.gaussians:
        n = args.gaussians
    cfg = (n, sh, w, h, seed)

    # --- resident scene: each rank generates and uploads only its index shard ---
    start, count = parallel.shard_range(n, rank, world) if (use_dist and args.shard_mode == "index") else (0, n)
    t0 = time.perf_counter()
    g = scene.synthetic_gaussians(n, seed, sh, start, count)
    t_gen = time.perf_counter() - t0
    sh_kind = {"single": 0, "half": 1, "norm8": 2, "none": 3}[args.pod.split("/")[0]]
    cov_kind = {"single": 0, "half": 1}[args.pod.split("/")[1]]
    afr = use_dist and args.shard_mode == "frames"   # frame-parallel: a step is still ONE frame of the orbit; N ranks render N per round
    gather = "rgba8" if afr else (args.gather if (use_dist and args.shard_mode == "screen") else "float")
    renderer = parallel.ShardedViewer(device=local_rank, world=world, rank=rank, use_dist=use_dist, sh=sh_kind, cov3d=cov_kind,
                                      mode=args.shard_mode if use_dist else "index", gather=gather, overlap_gather=gather == "rgba8",
                                      comm="lib" if (use_dist and args.shard_mode == "index") else None)  # index mode: collectives inside libgsx (RCCL)
    viewer = renderer.stages.viewer
    overrides = {k: float(x) if "." in x else int(x) for k, x in (kv.split("=") for kv in args.render_options.split(","))} if args.render_options else {}
    if overrides:
        viewer.set_render_options(**overrides)
    t0 = time.perf_counter()
    renderer.load_shard(g, start, n)
    renderer.poll()
    t_up = time.perf_counter() - t0
    upload_gbs = g.nbytes / t_up / 1e9
    del g

    # the orbit's cameras (inputs of the path) are prepared before the clock starts: look_at / perspective in numpy cost
    # ~30 us a pose, which an un-synchronised frame loop hides but a host that waits for the device (host_verify) does not
    orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
    pose_of = list(range(240))
    if args.pose_order == "random":
        pose_of = [int(x) for x in np.random.default_rng(2024).permutation(240)]

    def frame(i):
        renderer.render_frame(orbit[pose_of[(i * args.pose_stride) % 240]], (w, h))

    if afr:
        # round j of the orbit = frames j * world .. j * world + world - 1; this rank renders frame j * world + rank
        one_frame = frame

        def frame(j, limit=None):  # noqa: F811 — limit: frames of the orbit that exist in this loop (the last round may be partial)
            if limit is None or j * world + rank < limit:
                one_frame(j * world + rank)
            else:
                renderer.skip_frame()

    per = world if afr else 1     # frames of the orbit per loop iteration
    rounds = lambda k: (k + per - 1) // per  # noqa: E731

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(first_round, steps=None):
        """W untimed warm-up steps, then EXACTLY `steps` (default K = --steps) steps between barrier + synchronize fences; max over
        ranks.  Returns (seconds, project-pass timing, index of the last frame rendered)."""
        steps = args.steps if steps is None else steps
        for i in range(rounds(args.warmup)):
            frame(first_round + i)
        renderer.poll()
        renderer.set_pass_timing(True, None if args.pass_timing == "all" else ["project", "project_geom"])
        renderer.get_pass_timing()  # reset accumulators
        first = first_round + rounds(args.warmup)
        fence()
        launches0 = viewer_mod.launch_count()
        t0 = time.perf_counter()
        if afr:   # exactly args.steps frames of the orbit, dealt round-robin; ranks without a frame in the last round only gather
            for i in range(rounds(steps)):
                frame(first + i, limit=first * world + steps)
        else:
            for i in range(steps):
                frame(first + i)
        renderer.poll()
        fence()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        timing = renderer.get_pass_timing()
        renderer.set_pass_timing(False)
        launches["per_frame"] = round((viewer_mod.launch_count() - launches0) / max(steps, 1), 2)   # kernel launches the library asked for
        return elapsed, timing, first + rounds(steps) - 1

    launches = {}
    closed = False

    def accounting(first_round, frames=32):
        """Untimed pass over the same poses: per-frame device counts (reading them costs a sync per frame)."""
        acct = []
        for i in range(min(rounds(args.steps), frames)):
            frame(first_round + i)
            st = renderer.last_stats()
            acct.append((st["n_gaussians"], st["n_visible"], st.get("n_sorted", st["n_visible"]), 1 if st.get("speculated") else 0,
                         1 if st.get("n_repair_tiles", 0) > 0 else 0, st["n_tile_entries"], st.get("overflow_slabs", 0)))
        return np.asarray(acct, np.float64)

    # working buffers (records, sort and tile-pair buffers: sized by the scene) are allocated by the first frame a model is
    # rendered in; that belongs to loading the scene, not to a step
    frame(0)
    renderer.poll()

    if args.pmc_child:
        # the run rocprofv3 wraps for the PMC passes: a few speculated frames, then a few unspeculated ones; no output
        for i in range(args.warmup + args.steps):
            frame(i)
        renderer.poll()
        viewer.set_render_options(**dict(overrides, speculative=0, slab_shading=0))   # (every Gaussian projected in full: k_project<3,0,0>)
        for i in range(args.warmup + args.steps):
            frame(i)
        renderer.poll()
        renderer.close()
        return

    if args.host_profile and use_dist:
        renderer.profile = {}
    # N = 1: the headline loop keeps --frames-in-flight frames in flight (lanes inside libgsx: own stream and per-frame buffers,
    # shared scene); the same loop with one frame in flight follows, and the kernel rooflines are measured there, uncontended
    # (N > 1 / --force-dist: gsx_shard_render_frame accepts frames_in_flight too — the verdict of frame k is then read after
    # frame k + 1 is enqueued — but with every collective on one in-order stream frame k + 1's exchange queues behind frame k's
    # band gather: 1230 -> 1015 fps at world 1.  --dist-frames-in-flight L runs it anyway.)
    lib_index = use_dist and args.shard_mode == "index"
    if lib_index:
        viewer.shard_set_gather_root(0)   # north_star: "into one framebuffer" — rank 0's; the other ranks send their band and receive none
    lanes = args.frames_in_flight if (single and not overrides) else (args.dist_frames_in_flight if lib_index else 1)

    def set_opts(**kw):
        viewer.set_render_options(**dict(overrides, **kw))

    lanes_probe = None
    if lanes > 1 and single and not args.frames_in_flight_given:
        # Untimed probe, like the library's own speculation tuner: frames in flight pay when the device runs two hardware queues side
        # by side.  One box of the pool (round 4) ran the two-lane loop at 0.6 frames/s and the one-lane loop at 1750 in the same
        # process — whatever held its second queue, a run that lands on such a box should report what the path does with one
        # frame in flight and say so, not that.  48 frames each way; the default stays when it is not clearly slower.
        def probe(l):
            set_opts(frames_in_flight=l)
            for i in range(16):
                frame(i)
            renderer.poll()
            t0 = time.perf_counter()
            for i in range(16, 64):
                frame(i)
            renderer.poll()
            return 48.0 / (time.perf_counter() - t0)
        lanes_probe = {"frames_in_flight_1": round(probe(1), 1), f"frames_in_flight_{lanes}": round(probe(lanes), 1)}
        if lanes_probe[f"frames_in_flight_{lanes}"] < 0.85 * lanes_probe["frames_in_flight_1"]:
            lanes_probe["decision"] = f"this device does not overlap two streams ({lanes} frames in flight slower than one): the headline loop runs with ONE frame in flight"
            lanes = 1
        set_opts(frames_in_flight=lanes)
    elif lanes > 1:
        set_opts(frames_in_flight=lanes)
    if lib_index:
        viewer.shard_stats(reset=True)
    # `value` is timed over a whole orbit (>= 240 frames) whatever --steps is: a K = 20 window misses the speculation tuner's probe
    # frames and read 13-16 % above the same run's whole-orbit rate in rounds 3 and 4.  The K-step window is
    # kept beside it as value_short_window.
    # (args.steps is the timed length from here on — main() raised it to a whole orbit; steps_requested is what the caller passed)
    elapsed_short = None
    if steps_requested != args.steps:
        elapsed_short, _, _ = timed_loop(0, steps_requested)
        if lib_index:
            viewer.shard_stats(reset=True)
    elapsed, timing, last_idx = timed_loop(0)
    launches["headline"] = launches["per_frame"]
    shard_stats_timed = viewer.shard_stats(reset=True) if lib_index else None
    # what the library holds on the device for this scene and this schedule, right after the headline loop (every DevBuf of the process:
    # the model's planes, its shade records, sort / bin / list buffers of every lane, framebuffers)
    resident_bytes = viewer_mod.device_bytes()
    if args.host_profile and use_dist and rank == 0:
        print("host ms/frame by section:", {k: round(1e3 * x / args.steps, 4) for k, x in renderer.profile.items()}, file=sys.stderr)
    renderer.profile = None

    # ---- the last timed frame is still in the framebuffer: keep it for the frame check ----
    frame_check = None
    fb_last = None
    if single:
        renderer.poll()
        fb_last = renderer.framebuffer().copy()
    if lib_index and world > 1:
        # N > 1: rank 0 holds the last timed frame (the bands were gathered to it); it renders that pose once more from the WHOLE
        # scene with the plainest single-GPU schedule and the two must be equal bit for bit.  The other ranks wait at the barrier.
        renderer.poll()
        if rank == 0:
            fb_dist = renderer.framebuffer().copy()
            plain = parallel.ShardedViewer(device=local_rank, world=1, rank=0, use_dist=False, sh=sh_kind, cov3d=cov_kind)
            plain.stages.viewer.set_render_options(speculative=0, progressive=0)
            plain.load_shard(scene.synthetic_gaussians(n, seed, sh, 0, n), 0, n)
            plain.render_frame(orbit[pose_of[(last_idx * args.pose_stride) % 240]], (w, h))
            plain.poll()
            fb_plain = plain.framebuffer()
            equal = bool(np.array_equal(fb_dist, fb_plain))
            frame_check = dict(pose=pose_of[(last_idx * args.pose_stride) % 240], equal_to_single_gpu_single_pass_frame=equal,
                               max_abs_diff=float(np.abs(fb_dist - fb_plain).max()),
                               checksum=int(np.frombuffer(fb_dist.tobytes(), np.uint32).astype(np.uint64).sum() & 0xFFFFFFFFFFFF),
                               note=f"rank 0's gathered frame of the last timed step ({world} index shards) against the whole scene rendered "
                                    "on rank 0 alone with speculative = 0, progressive = 0")
            plain.close()
            if not equal:
                raise SystemExit(f"bench.py: the sharded frame differs from the single-GPU frame: {frame_check}")
        dist.barrier()
    acct = accounting(rounds(args.warmup))
    # N > 1 (index shards): the same K steps with the exchange unfiltered and every shard projected and shaded in full
    # (gsx_shard_render_frame(speculate = 0)) — the schedule whose dominant pass shards 1 / N; read it against the N = 1 line's
    # value_unspeculated
    elapsed_dist_u = None
    shard_stats_unspec = None
    if lib_index and world > 1 and not args.no_extra_legs:
        renderer.speculate = False
        viewer.shard_stats(reset=True)
        elapsed_dist_u, _, _ = timed_loop(0)
        shard_stats_unspec = viewer.shard_stats(reset=True)
        renderer.speculate = True
        for i in range(rounds(args.warmup)):   # (the limits of the frames that follow come from speculated frames again)
            frame(i)
        renderer.poll()
    elapsed_1 = elapsed_ul = None
    if lanes > 1 and (single or world == 1):   # (N > 1: one timed loop, as the contract says)
        set_opts()
        elapsed_1, timing, _ = timed_loop(0)
        launches["speculated_one_frame_in_flight"] = launches["per_frame"]
        acct = accounting(rounds(args.warmup))

    # ---- N = 1: the unspeculated loop of the same run (value_unspeculated + the SURVEY 8d projection roofline) ----
    elapsed_u = timing_u = acct_u = None
    elapsed_f = timing_f = acct_f = None
    if single and "speculative" not in overrides:
        viewer.set_render_options(**dict(overrides, speculative=0))
        elapsed_u, timing_u, _ = timed_loop(0)
        launches["unspeculated"] = launches["per_frame"]
        acct_u = accounting(rounds(args.warmup))
        # ... and the same loop with EVERY Gaussian projected in full (slab_shading = 0): the loop that runs SURVEY 8d's projection pass,
        # k_project<3,0,0>, once per frame — the kernel `roofline` prices
        viewer.set_render_options(**dict(overrides, speculative=0, slab_shading=0))
        elapsed_f, timing_f, _ = timed_loop(0)
        launches["unspeculated_full_projection"] = launches["per_frame"]
        acct_f = accounting(rounds(args.warmup))
        viewer.set_render_options(**dict(overrides, speculative=0))
        if lanes > 1 and args.unspeculated_in_flight:
            set_opts(speculative=0, frames_in_flight=lanes)
            elapsed_ul, _, _ = timed_loop(0)
        # frame check: the pose of the last timed speculated frame through the plainest schedule (no slabs, no speculation)
        viewer.set_render_options(**dict(overrides, speculative=0, progressive=0))
        frame(last_idx)
        renderer.poll()
        fb_plain = renderer.framebuffer()
        equal = bool(np.array_equal(fb_last, fb_plain))
        frame_check = dict(pose=pose_of[(last_idx * args.pose_stride) % 240], equal_to_unspeculated_single_pass=equal,
                           max_abs_diff=float(np.abs(fb_last - fb_plain).max()),
                           checksum=int(np.frombuffer(fb_last.tobytes(), np.uint32).astype(np.uint64).sum() & 0xFFFFFFFFFFFF))
        viewer.set_render_options(**overrides) if overrides else viewer.set_render_options()
        if not equal:
            raise SystemExit(f"bench.py: the last timed frame differs from the speculative=0, progressive=0 frame: {frame_check}")
    # ---- N = 1: what a host that WAITS gets, a whole-orbit figure, and every pass against its bytes ----
    extra = None
    if single and not overrides and not args.no_extra_legs and not args.pose_stride > 1 and args.pose_order == "orbit":
        extra = {}
        key = renderer.KEY

        def host_sync_loop(protocol):
            """K frames, the host waiting for each: 'frame' = gsx_render_frame + gsx_sync; 'reference' = the app's sequence
            preprocess + sort, poll, render, poll (scene.rs:856-873, 613-614)."""
            st = renderer.stages

            def one(i):
                st.set_uniforms(key, orbit[pose_of[i % 240]], (w, h))
                if protocol == "frame":
                    viewer.render_frame([key])
                else:
                    viewer.preprocessor.preprocess(key)
                    viewer.radix_sorter.sort(key)
                    viewer.poll()
                    viewer.renderer.render([key])
                viewer.poll()

            for i in range(args.warmup):
                one(i)
            fence()
            t0 = time.perf_counter()
            for i in range(args.warmup, args.warmup + args.steps):
                one(i)
            fence()
            return time.perf_counter() - t0

        set_opts()
        extra["sync_frame"] = host_sync_loop("frame")
        extra["sync_reference"] = host_sync_loop("reference")
        set_opts(speculative=0)
        extra["sync_frame_unspeculated"] = host_sync_loop("frame")
        # the same loops with every launch submitted on its own (gsx_debug_set_launch_graphs(0)): what the cached, patched HIP graphs
        # of csrc/gsx_launch.h are worth, same process, same resident scene
        def host_us_per_call():
            """host time inside gsx_render_frame, free-running loop, one frame in flight"""
            st = renderer.stages
            spent = 0.0
            for i in range(args.warmup + 60):
                st.set_uniforms(key, orbit[pose_of[i % 240]], (w, h))
                t_a = time.perf_counter()
                viewer.render_frame([key])
                if i >= args.warmup:
                    spent += time.perf_counter() - t_a
            viewer.poll()
            return round(1e6 * spent / 60.0, 1)

        set_opts()
        viewer_mod.set_launch_graphs(True)
        viewer.launch_stats(reset=True)
        el_g, _, _ = timed_loop(0)
        graph_stats = viewer.launch_stats(reset=True)
        host_g = host_us_per_call()
        sync_g = host_sync_loop("frame")
        sync_ref_g = host_sync_loop("reference")
        set_opts(speculative=0)
        el_ug, _, _ = timed_loop(0)
        viewer_mod.set_launch_graphs(False)   # (the default)
        el_ud, _, _ = timed_loop(0)
        set_opts()
        el_d, _, _ = timed_loop(0)
        host_d = host_us_per_call()
        extra["launch_graphs"] = dict(
            one_frame_in_flight=dict(graphs=round(args.steps / el_g, 1), direct=round(args.steps / el_d, 1)),
            synchronised=dict(graphs=round(args.steps / sync_g, 1), direct=round(args.steps / extra["sync_frame"], 1)),
            reference_protocol=dict(graphs=round(args.steps / sync_ref_g, 1), direct=round(args.steps / extra["sync_reference"], 1)),
            unspeculated=dict(graphs=round(args.steps / el_ug, 1), direct=round(args.steps / el_ud, 1)),
            host_us_inside_gsx_render_frame=dict(graphs=host_g, direct=host_d),
            stats_of_the_graph_loop={k: v for k, v in graph_stats.items()},
            note="fps, one frame in flight, same process and scene; 'direct' (the default): every launch submitted on its own; 'graphs' "
                 "(gsx_debug_set_launch_graphs(1) / GSX_GRAPH=1): while its stream is busy gsx_render_frame records its launches and submits "
                 "them as cached HIP graphs whose nodes are patched to the frame's arguments — that saves HOST time "
                 "(host_us_inside_gsx_render_frame) and nothing on the device: a real kernel boundary costs the same inside a graph as on a "
                 "stream (tools/bench_launch.hip's 3.3 -> 1.75 us is the command processor's rate for EMPTY kernels), and a frame that is one "
                 "graph launch starts ~10 us later than one whose first kernel is already queued; an entry point that finds its stream "
                 "idle (the synchronised loops) submits launch by launch in either mode.  Same kernels, same arguments, same order: "
                 "frames are bit-identical (tests/test_gpu_graph.py)")
        # the headline schedule over a whole orbit (>= 240 frames): the tuner's probes and every part of the path are in it
        steady_frames = max(240, args.steps)
        set_opts(frames_in_flight=lanes)
        for i in range(args.warmup):
            frame(i)
        renderer.poll()
        fence()
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + steady_frames):
            frame(i)
        renderer.poll()
        fence()
        extra["steady"] = (steady_frames, time.perf_counter() - t0)
        # every pass bracketed (a loop of its own: each bracket costs a few microseconds of stream gap), one frame in flight
        saved_timing, args.pass_timing = args.pass_timing, "all"
        set_opts()
        el_p, tm_p, _ = timed_loop(0)
        ac_p = accounting(rounds(args.warmup))
        set_opts(speculative=0)
        el_pu, tm_pu, _ = timed_loop(0)
        ac_pu = accounting(rounds(args.warmup))
        set_opts(speculative=0, slab_shading=0)
        el_pf, tm_pf, _ = timed_loop(0)
        ac_pf = accounting(rounds(args.warmup))
        args.pass_timing = saved_timing
        extra["passes_raw"] = dict(speculated=(el_p, tm_p, ac_p), unspeculated=(el_pu, tm_pu, ac_pu), full_projection=(el_pf, tm_pf, ac_pf))
        set_opts()

    # ---- N = 1: the GPU side of the oracle check — pose 0 of the whole scene, the frame cpu_baseline renders on the host cores ----
    gpu_pose0 = None
    if single and not overrides and not args.no_cpu_baseline and not args.cpu_sample and args.pose_order == "orbit" and args.pose_stride == 1:
        key = renderer.KEY
        set_opts(speculative=0, progressive=0)
        renderer.stages.set_uniforms(key, orbit[0], (w, h))
        viewer.preprocessor.preprocess(key)
        viewer.radix_sorter.sort(key)
        viewer.poll()
        gp = viewer.download_projection(key)
        gpu_pose0 = dict(key=gp["key"], rect=gp["rect"], n_visible=gp["n_visible"], order=viewer.download_sorted(key).copy())
        del gp
        viewer.renderer.render([key])
        gpu_pose0["fb_plain"] = viewer.download_framebuffer().copy()
        set_opts(frames_in_flight=lanes)          # the headline schedule, arriving at pose 0 along the orbit
        for i in (232, 233, 234, 235, 236, 237, 238, 239, 240):
            frame(i)
        renderer.poll()
        gpu_pose0["fb_spec"] = renderer.framebuffer().copy()
        gpu_pose0["speculated"] = bool(renderer.last_stats().get("speculated"))
        set_opts()

    # ---- N = 1: speculation robustness legs (same process, same resident scene) ----
    robustness = None
    if single and not overrides and not args.no_robustness and not args.pose_stride > 1 and args.pose_order == "orbit":
        from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

        def leg():
            res = {}
            for name, spec in (("speculated", 1), ("unspeculated", 0)):
                viewer.set_render_options(speculative=spec)
                el, _, _ = timed_loop(0)
                ac = accounting(rounds(args.warmup))
                res[name] = dict(fps=round(args.steps / el, 1), ms_per_step=round(1e3 * el / args.steps, 4),
                                 speculated_frames=round(float(ac[:, 3].mean()), 3), frames_with_repair_round=round(float(ac[:, 4].mean()), 3),
                                 n_visible=int(ac[:, 1].mean()), n_depth_sorted=int(ac[:, 2].mean()), tile_entries=int(ac[:, 5].mean()),
                                 overflow_slabs=int(ac[:, 6].max()))
            viewer.set_render_options()
            res["speculated_over_unspeculated"] = round(res["speculated"]["fps"] / res["unspeculated"]["fps"], 3)
            return res

        robustness = {}
        saved = pose_of[:]
        pose_of[:] = [int(x) for x in np.random.default_rng(2024).permutation(240)]
        robustness["random_pose_order"] = leg()
        pose_of[:] = saved
        ev = MaskEvaluator(viewer)
        sky = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, -4.5, 0.0], np.float32), scale=np.array([10.0, 5.0, 10.0], np.float32))]
        ev.evaluate(MaskOp.parse("0"), renderer.KEY, sky)
        robustness["open_sky"] = leg()
        robustness["open_sky"]["scene"] = "mask box keeps the Gaussians with y <= 0.5 (gsx_mask_evaluate): the screen above the horizon stays open"
        ev.evaluate(None, renderer.KEY)   # MaskOpTree::Reset

        def scene_leg(variant, what):
            """the same loops on ANOTHER resident scene of the same size (scene.VARIANTS), uploaded in place of the benchmark scene; a long
            warm-up: the viewer needs ~50 frames to time both schedules and settle on the faster one (SpecTuner, gsx_frame.cpp)"""
            g2 = scene.synthetic_gaussians(n, seed, sh, 0, n, variant=variant)
            viewer.models[renderer.KEY].gaussian_buffers.gaussians_buffer.update_range(0, g2)   # (gsx_model_upload_range: in place)
            renderer.poll()
            del g2
            saved_w, args.warmup = args.warmup, max(args.warmup, 100)
            res = leg()
            args.warmup = saved_w
            res["scene"] = what
            res["list_entries_per_visible_gaussian"] = round(res["unspeculated"]["tile_entries"] / max(res["unspeculated"]["n_visible"], 1), 3)
            return res

        if not args.no_scene_legs:
            robustness["translucent"] = scene_leg("translucent", "the benchmark scene with opacity logits N(-5.5, 1.5) (median opacity 0.004): next to no tile ever "
                                                  "saturates, there is nothing to speculate on — the viewer must find that out and stop (speculated_over_unspeculated ~ 1)")
            robustness["surfaces"] = scene_leg("surfaces", "a captured-scene stand-in (the INRIA garden PLY is not in the image): Gaussians on six planes and four "
                                               "spheres, flattened along the normal, log-scales N(-3, 1.2) clamped to [-7, 1] — a heavy tail, the widest splats "
                                               "cover hundreds of tiles — opacity logits N(2, 1.5)")
        robustness["note"] = ("speculated frames are bit-identical to unspeculated ones whatever the poses (tests/test_gpu_speculation.py); "
                              "the viewer pauses speculation by itself when it keeps repairing without admitting less (gsx_frame.cpp)")
    overflow_slabs = int(max(acct[:, 6].max(), acct_u[:, 6].max() if acct_u is not None else 0))
    if overflow_slabs:
        raise SystemExit(f"bench.py: {overflow_slabs} depth slabs overflowed the tile-pair buffers during the run (slow pair-free path)")

    # gather per-rank numbers for the roofline of the projection pass (dominant HBM stream of the path)
    local = torch.tensor([acct[:, 0].mean(), acct[:, 1].mean(), acct[:, 2].mean(), acct[:, 3].mean(), acct[:, 4].mean(),
                          acct[:, 5].mean(), timing["project"]["ms"] * 1e3, timing["project"]["launches"]], dtype=torch.float64, device="cuda")
    if use_dist:
        allr = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(allr, local)
    else:
        allr = [local]
    allr = torch.stack(allr).cpu().numpy()

    # ---- N > 1 (index shards): what every rank did, so that a scaling run can be read — the library's own bookkeeping of the
    #      timed loop (gsx_shard_get_stats: host side, never synchronises) and a second, short loop with every pass bracketed ----
    per_rank = None
    if lib_index:
        frames_stats = max(shard_stats_timed["frames"], 1)
        renderer.set_pass_timing(True, None)
        renderer.get_pass_timing()
        n_prof = min(args.steps, 40)
        for i in range(n_prof):
            frame(i)
        renderer.poll()
        tm_all = renderer.get_pass_timing()
        renderer.set_pass_timing(False)
        edges_now = [int(x) for x in viewer.shard_get_band_edges(world)]
        mine = torch.tensor([shard_stats_timed["wire_bytes"] / frames_stats, shard_stats_timed["repair_frames"] / frames_stats,
                             shard_stats_timed["redo_frames"] / frames_stats, shard_stats_timed["verdict_wait_ns"] / 1e3 / frames_stats,
                             shard_stats_timed["exchange_rounds"] / frames_stats, shard_stats_timed["last_slot_records"],
                             count] + [tm_all[name]["ms"] * 1e3 / n_prof for name in ("project", "project_geom", "depth_sort", "bin", "tile_sort", "composite")]
                            + [acct[:, 5].mean(), edges_now[rank + 1] - edges_now[rank],
                               (shard_stats_unspec["wire_bytes"] / max(shard_stats_unspec["frames"], 1)) if shard_stats_unspec else 0.0,
                               shard_stats_timed["redo_fallbacks"], shard_stats_timed["last_repair_slot_records"]],
                            dtype=torch.float64, device="cuda")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        g_all = torch.stack(gathered).cpu().numpy()
        per_rank = dict(
            note="per rank, in rank order; wire bytes = what the rank put on the links (fixed-size slots to every peer, feedback and band "
                 "gathers); pass times from a second loop of %d frames with every pass bracketed by events" % n_prof,
            shard_gaussians=[int(x) for x in g_all[:, 6]], wire_bytes_per_frame=[int(x) for x in g_all[:, 0]],
            frames_with_repair_round=[round(float(x), 3) for x in g_all[:, 1]], frames_redone_with_whole_shard_slots=[round(float(x), 3) for x in g_all[:, 2]],
            verdict_wait_us_per_frame=[round(float(x), 1) for x in g_all[:, 3]], exchange_rounds_per_frame=[round(float(x), 3) for x in g_all[:, 4]],
            slot_records_last_frame=[int(x) for x in g_all[:, 5]],
            pass_us_per_frame={name: [round(float(x), 1) for x in g_all[:, 7 + k]]
                               for k, name in enumerate(("project", "project_geom", "depth_sort", "bin", "tile_sort", "composite"))},
            list_entries_per_frame=[int(x) for x in g_all[:, 13]],
            list_entries_max_over_mean=round(float(g_all[:, 13].max() * world / max(g_all[:, 13].sum(), 1.0)), 3),
            band_rows=[int(x) for x in g_all[:, 14]], band_edges=edges_now,
            work_busiest_rank_over_mean=round(shard_stats_timed["last_work_permille"] / 1000.0, 3),
            bands="tile rows [band_edges[g], band_edges[g + 1]) belong to rank g; cut by the previous frame's per-row work (list entries the "
                  "compositor's tiles walked, all-gathered with the saturation map): work_busiest_rank_over_mean is that measure for the last "
                  "timed frame, list_entries_max_over_mean the ranks' binned entries",
            wire_bytes_per_frame_unspeculated=[int(x) for x in g_all[:, 15]] if shard_stats_unspec else None,
            redone_frames_redone_again_with_whole_shard_slots=[int(x) for x in g_all[:, 16]], repair_slot_records_last_frame=[int(x) for x in g_all[:, 17]],
            protocol=("a single model's frame is enqueued whole and nothing in it waits for the device; its verdict is read when the frame is "
                      "retired (frames in flight: after the next frame is enqueued; verdict_wait_us_per_frame is what the host waited then) and the "
                      "repair round, where the verdict asks for one, is exchanged exactly sized at that point (host-decided; layered frames go out "
                      "model by model the same way); a frame whose slots overflowed is redone at its retirement, before its lane is used again and "
                      "before anybody can read it (frames_redone_with_whole_shard_slots)"
                      + ("; GSX_SHARD_LAYER_PIPELINE=0 in force: repair rounds always enqueued, decided on the device"
                         if os.environ.get("GSX_SHARD_LAYER_PIPELINE") == "0" else "")))

    comm_info = viewer.comm_info() if lib_index else None
    if rank == 0:
        fps = args.steps / elapsed
        sh_bytes = {0: 180, 1: 96, 2: 48, 3: 0}[sh_kind if sh > 0 else 3]
        cov_bytes = {0: 24, 1: 12}[cov_kind]
        pod_bytes = 16 + sh_bytes + cov_bytes
        shade_rec_bytes = {0: 256, 1: 192 if cov_kind == 0 else 128, 2: 128, 3: 0}[sh_kind if sh > 0 else 3]   # (csrc aos_layout: whole 64-byte lines)

        def roofline_of(ac, tm, label_full):
            """Projection kernel of one timed loop.  ALGORITHMIC bytes per launch (means over the accounting frames):
            full kernel (every visible Gaussian shaded in the kernel): SURVEY 8d, N*pod + N_vis*40;
            geometry-only kernel of speculated frames: N*(16 pos+rgba8 + 4 key + 4 packed tile rectangle + 1/8 ballot) + N_vis*cov —
            the SH planes and the mean / conic / colour records are k_shade's, for the few admitted Gaussians."""
            n_loc, nvis_loc, spec_frac = ac[:, 0].mean(), ac[:, 1].mean(), ac[:, 3].mean()
            lazy = spec_frac > 0.5
            if lazy:
                rect8 = (w + 15) // 16 <= 255 and (h + 15) // 16 <= 255   # packed 4-byte rectangles (Records::rect8) up to 255 x 255 tiles
                b = n_loc * 24.125 + nvis_loc * cov_bytes if rect8 else n_loc * 20.125 + nvis_loc * (16 + cov_bytes)
                kernel = (f"k_project_geom<{cov_kind},1> (projection of a speculated frame: cov2d + cull + depth key + admission; SH colour "
                          "is evaluated by k_shade for the admitted Gaussians only)")
                definition = "geometry-only projection: N*24.125 + N_vis*cov" if rect8 else "geometry-only projection: N*20.125 + N_vis*(16+cov)"
            else:
                b = n_loc * pod_bytes + nvis_loc * 40
                kernel = f"k_project<{sh},{sh_kind},{cov_kind}> ({label_full})"
                definition = f"SURVEY 8d: N*{pod_bytes} + N_vis*40"
            tp = tm["project_geom" if lazy else "project"]   # (a speculated loop holds a few full projections too: the tuner's probes)
            us = tp["ms"] * 1e3 / max(tp["launches"], 1)
            ach = b / (us * 1e-6) / 1e9 if us > 0 else 0.0
            tr = (traffic or {}).get("speculated" if lazy else "full")
            return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                    "algorithmic_bytes_per_launch": int(b), "avg_launch_us": round(us, 2), "launches_timed": int(tp["launches"]),
                    "bytes_definition": definition, "n": int(n_loc), "n_visible": int(nvis_loc),
                    "traffic_source": traffic_note if tr else (traffic_note if traffic is None else "no dispatch of this kernel in the PMC passes"),
                    "traffic_detail": tr}

        n_loc, nvis_loc, nsort_loc, spec_frac, repair_frac, entries = allr[0][:6]
        passes = {}
        for name, tv in timing.items():
            if args.pass_timing == "all" or name in ("project", "project_geom"):
                passes[name] = round(tv["ms"] / max(rounds(args.steps), 1), 4)
        if not use_dist:
            sharding, scaling = "one GPU", "strong"
        elif afr:
            sharding = (f"REPLICATED, frame-parallel x{world}: scene resident on each GPU, rank g renders frame j*{world}+g of the orbit "
                        "(one-frame latency as on one GPU), RGBA8 frames all-gathered on a second stream")
            scaling = "weak"   # every GPU holds and renders the whole scene: throughput scales, one frame does not
        elif args.shard_mode == "index":
            sharding = (f"splat-index shards x{world}: gsx_shard_render_frame — projection of the resident shard, speculative exchange of "
                        "fixed record slots by tile-row band (RCCL point-to-point inside libgsx, counts on the device), verification + "
                        "repair round, the finished bands gathered into rank 0's framebuffer (gsx_shard_set_gather_root); no host round trip inside a frame")
            scaling = "strong"
        else:
            sharding = (f"scene resident on each of {world} GPUs, rank g renders band g of tile rows, band all-gather "
                        + ("of RGBA8 pixels (resolved per band) on a second stream" if gather == "rgba8" else "of (rgb, T) float4 pixels"))
            scaling = "strong"
        out = {
            "metric": METRIC,
            "value": round(fps, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps_requested,
            "steps_timed": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: synthetic {n} Gaussians SH-deg-{sh}, {w}x{h}, orbit r=6 h=1.5 240 poses"
                            + (" in seeded random order" if args.pose_order == "random" else "") + f", seed {seed}",
                "gaussians": n, "width": w, "height": h, "sh_degree": sh, "pod": args.pod, "pod_bytes": pod_bytes,
                "sharding": sharding, "schedule": "progressive depth slabs + temporal occlusion speculation (gsx_render_options defaults)"
                if not overrides else f"overrides: {args.render_options}",
                "frames_in_flight": lanes,
                "n_visible_rank0": int(nvis_loc), "n_depth_sorted_rank0": int(nsort_loc), "tile_entries_rank0": int(entries),
                "speculated_frames": round(float(spec_frac), 3), "frames_with_repair_round": round(float(repair_frac), 3),
                "pass_ms_per_frame_rank0": passes,
                "upload_GBps_pcie_inclusive": round(upload_gbs, 2), "scene_gen_s": round(t_gen, 1),
            },
            "overflow_slabs": overflow_slabs,
        }
        if elapsed_short is not None:
            out["value_short_window"] = round(steps_requested / elapsed_short, 3)
            out["value_note"] = (f"value / ms_per_step: {args.steps} frames (a whole orbit) between the fences — every timed loop of this line is at "
                                 f"least one orbit long; value_short_window: the {steps_requested}-step window that was asked for, same loop, timed first")
        if elapsed_1 is not None:
            out["value_one_frame_in_flight"] = round(args.steps / elapsed_1, 3)
            out["ms_per_step_one_frame_in_flight"] = round(1e3 * elapsed_1 / args.steps, 4)
            out["frames_in_flight_note"] = (
                f"value: gsx_render_options.frames_in_flight = {lanes} — consecutive frames go round-robin to {lanes} lanes inside libgsx (own "
                "stream, records, sort / tile buffers, framebuffer and speculation windows; the scene is shared), so the device overlaps one "
                "frame's latency-bound tail with the next frame's bandwidth-bound projection; every frame is bit-identical to the one-lane "
                "frame (tests/test_gpu_inflight.py, and frame_check below is taken from this loop).  value_one_frame_in_flight, "
                "value_unspeculated, both rooflines and the robustness legs run with ONE frame in flight (kernel times uncontended).  "
                f"Throughput, not latency: with {lanes} frames in flight a frame spends about {lanes} x ms_per_step on the device from its first "
                "kernel to its last (ms_per_step_one_frame_in_flight is the latency of a frame that has the device to itself).")
            out["frame_latency_ms_estimate"] = round(lanes * 1e3 * elapsed / args.steps, 4)
            if lanes_probe is not None:
                out["frames_in_flight_probe"] = dict(lanes_probe, note="untimed, before the timed region: 48 frames with one frame in flight and 48 "
                                                     "with the default; the headline loop keeps the default unless it is below 0.85 x the one-lane rate")
        if elapsed_ul is not None:
            out["value_unspeculated_in_flight"] = round(args.steps / elapsed_ul, 3)
        if timing_u is not None:
            out["value_unspeculated"] = round(args.steps / elapsed_u, 3)
            out["ms_per_step_unspeculated"] = round(1e3 * elapsed_u / args.steps, 4)
            out["value_unspeculated_full_projection"] = round(args.steps / elapsed_f, 3)
            out["unspeculated_note"] = ("value_unspeculated: speculative = 0 — no windows from an earlier frame: what every first frame and every incoherent "
                                        "pose costs; such a frame projects geometry only, depth-sorts every visible Gaussian and shades, depth slab by depth "
                                        "slab, exactly the records some block of tiles still takes (gsx_render_options.slab_shading, default 1).  "
                                        "value_unspeculated_full_projection: the same loop with slab_shading = 0 — every Gaussian projected in full by "
                                        "k_project<3,0,0> (SH colour + cov2d + cull + depth key for all N: SURVEY 8d's projection pass, what the reference's "
                                        "K1 + K3 vertex stage computes): the loop `roofline` is measured on.  Same pixels in all three schedules (frame_check).")
            out["frame_check"] = frame_check
            out["roofline"] = roofline_of(acct_f, timing_f, "projection pass: SH colour + cov2d + cull + depth key; measured on the "
                                          "unspeculated, fully projecting timed loop of this run (slab_shading = 0)")
            out["roofline_speculated"] = roofline_of(acct, timing, "projection pass")
            out["config"]["tile_entries_unspeculated"] = int(acct_u[:, 5].mean())
        else:
            out["roofline"] = roofline_of(acct, timing, "projection pass: SH colour + cov2d + cull + depth key")
        if out.get("roofline") is not None:
            out["roofline"]["belongs_to"] = ("value_unspeculated_full_projection: that loop runs this kernel once per frame; the headline loop "
                                             "(`value`) and value_unspeculated project with roofline_speculated's kernel and shade only the Gaussians "
                                             "that are admitted / that some block of tiles takes (k_shade)"
                                             if timing_u is not None else "value")
        if extra is not None:
            out["value_synchronised"] = round(args.steps / extra["sync_frame"], 3)
            out["ms_per_step_synchronised"] = round(1e3 * extra["sync_frame"] / args.steps, 4)
            out["value_reference_protocol"] = round(args.steps / extra["sync_reference"], 3)
            out["value_synchronised_unspeculated"] = round(args.steps / extra["sync_frame_unspeculated"], 3)
            out["synchronised_note"] = ("the host waits for every frame, as the app does: value_synchronised = gsx_render_frame + gsx_sync per frame "
                                        "(SURVEY 8d's definition of the metric); value_reference_protocol = the app's own sequence with its TWO "
                                        "blocking waits per frame — gsx_preprocess + gsx_sort, gsx_sync, gsx_render, gsx_sync "
                                        "(src/tab/scene.rs:856-873, 613-614); default schedule, one frame in flight; `value` is the same frames "
                                        "enqueued back to back without a host wait")
            sf, st_el = extra["steady"]
            out["steady_state"] = dict(frames=sf, value=round(sf / st_el, 3), ms_per_step=round(1e3 * st_el / sf, 4),
                                       note=f"the headline loop (frames_in_flight = {lanes}) over a whole orbit, timed like `value`")

            def passes_of(el, tm, ac, speculated):
                """every pass against its algorithmic bytes; means over the accounting frames of that loop"""
                n_a, nvis, nsort, _, _, d_ent = [float(ac[:, k].mean()) for k in range(6)]
                tiles = ((w + 15) // 16) * ((h + 15) // 16)
                p_tile = max(1, -(-max(1, (tiles - 1).bit_length()) // 8))
                lazy = (speculated and float(ac[:, 3].mean()) > 0.5) or (tm.get("project_geom", {}).get("ms", 0) > tm.get("project", {}).get("ms", 0))
                rows = {}
                proj_name = "project_geom" if lazy else "project"
                proj_bytes = (n_a * 24.125 + nvis * cov_bytes) if lazy else (n_a * pod_bytes + nvis * 40)
                defs = {
                    proj_name: (proj_bytes, "geometry-only projection: N*24.125 + N_vis*cov" if lazy else f"SURVEY 8d projection: N*{pod_bytes} + N_vis*40"),
                    "depth_sort": (nsort * 68, "SURVEY 8d K2: N_sorted*68 (four 8-bit passes of 8-byte pairs, read + write, + 4 B): the records that "
                                               "enter the depth sort — all visible ones unspeculated, the admitted ones speculated; the pass also holds "
                                               "the admission compaction and, on speculated frames, the repair round's sort (shading is timed apart: `shade`)"),
                    "shade": ((nsort * (shade_rec_bytes + 48)) if (lazy and speculated) else 0,
                              f"conic / colour records for the Gaussians the frame admitted: N_sorted*({shade_rec_bytes} + 48) — the {shade_rec_bytes}-byte shade record in, "
                              "48 bytes of conic / colour out (k_shade_quads + the frame's colour ops; a gather: a 64-byte sector per 16 bytes asked for). "
                              "Unspeculated frames with slab shading: the records some block of each depth slab takes (their number is not on the line: time only)"),
                    "bin": (nsort * 44 + d_ent * 12, "BASELINE 4 binning: N_sorted*44 + D*12, D = list entries actually binned (block entries on "
                                                     "progressive frames)"),
                    "tile_sort": (d_ent * 24 + d_ent * 8, "BASELINE 4 sort: D*24*p + D*8 with p = 1 (block lists: one 8-bit pass over "
                                                                                   "the block ids; per-tile lists would need p = %d)" % p_tile),
                    "composite": (d_ent * 40 + w * h * 16, "BASELINE 4 composite: D*40 + W*H*16 (VALU / LDS bound: reported against HBM bytes as 8d asks)"),
                }
                for name, (b, definition) in defs.items():
                    t = tm.get(name)
                    if not t or t["ms"] <= 0:
                        continue
                    us = t["ms"] * 1e3 / args.steps
                    if not b:
                        rows[name] = dict(us_per_frame=round(us, 1), bytes_definition=definition)
                        continue
                    gbs = b / (us * 1e-6) / 1e9
                    rows[name] = dict(us_per_frame=round(us, 1), algorithmic_bytes=int(b), GBps=round(gbs, 1), frac_of_8TBps=round(gbs / HBM_PEAK_GBS, 4),
                                      bytes_definition=definition)
                rows["frame_ms_with_every_pass_bracketed"] = round(1e3 * el / args.steps, 4)
                rows["n_visible"], rows["n_depth_sorted"], rows["list_entries"] = int(nvis), int(nsort), int(d_ent)
                return rows

            pr = extra["passes_raw"]
            out["passes"] = dict(speculated=passes_of(*pr["speculated"], True), unspeculated=passes_of(*pr["unspeculated"], False),
                                 full_projection=passes_of(*pr["full_projection"], False),
                                 note="one frame in flight, every pass bracketed with a pair of HIP events on the viewer's stream (each bracket costs a "
                                      "few microseconds of stream gap: the frame is slower than value_one_frame_in_flight)")
        if elapsed_dist_u is not None:
            out["value_unspeculated"] = round(args.steps / elapsed_dist_u, 3)
            out["value_unspeculated_note"] = ("the same K steps through gsx_shard_render_frame(speculate = 0): every shard projected and shaded in "
                                              "full, the exchange unfiltered (whole-shard slots); compare with value_unspeculated of the N = 1 line")
        if per_rank is not None:
            out["per_rank"] = per_rank
        if lib_index and world > 1:
            out["predicted"] = predicted_for(world, args.workload)
        if frame_check is not None and "frame_check" not in out:
            out["frame_check"] = frame_check
        if one_device:
            out["one_device_emulation"] = (f"GSX_BENCH_ONE_DEVICE=1: the {world} ranks are processes sharing ONE GPU, RCCL over sockets on lo — the N > 1 "
                                           "code ran for real, `value` measures nothing")
        if launches:
            out["launches_per_frame"] = {k: v for k, v in launches.items() if k != "per_frame"}
        out["resident_bytes"] = {"device_bytes_after_headline_loop": int(resident_bytes), "per_gaussian_of_this_rank": round(resident_bytes / max(count, 1), 1),
                                 "gaussians_of_this_rank": int(count),
                                 "what": "gsx_debug_device_bytes(): every device buffer the library holds in this process (planes, shade records, sort / bin / "
                                         "list buffers of every lane, framebuffers); the planes alone are 236 B per Gaussian for the f32 pod"}
        if extra and "launch_graphs" in extra:
            out["launch_graphs"] = extra["launch_graphs"]
        if robustness is not None:
            out["robustness"] = robustness
        # the figures a reader wants side by side, as the LAST keys of the line (a log tail that truncates the line keeps them)
        summary = {"value": out["value"], "frames_in_flight": lanes, "steps_timed": args.steps}
        for k_out, k_sum in (("value_short_window", "value_short_window"), ("value_one_frame_in_flight", "value_one_frame_in_flight"),
                             ("value_synchronised", "value_synchronised"), ("value_reference_protocol", "value_reference_protocol"),
                             ("value_unspeculated", "value_unspeculated"), ("value_unspeculated_full_projection", "value_unspeculated_full_projection"),
                             ("value_synchronised_unspeculated", "value_synchronised_unspeculated")):
            if k_out in out:
                summary[k_sum] = out[k_out]
        if "steady_state" in out:
            summary["steady_state_value"] = out["steady_state"]["value"]
        if out.get("roofline"):
            summary["roofline_frac"] = out["roofline"]["frac"]
            summary["roofline_kernel_us"] = out["roofline"]["avg_launch_us"]
            summary["roofline_belongs_to"] = "value_unspeculated_full_projection" if timing_u is not None else "value"
        if out.get("roofline_speculated"):
            summary["roofline_speculated_frac"] = out["roofline_speculated"]["frac"]
            summary["roofline_speculated_kernel_us"] = out["roofline_speculated"]["avg_launch_us"]
        if "passes" in out:
            summary["passes_us"] = {sch: {k: v["us_per_frame"] for k, v in out["passes"][sch].items() if isinstance(v, dict)}
                                    for sch in ("speculated", "unspeculated", "full_projection")}
        if launches:
            summary["launches_per_frame"] = out["launches_per_frame"]
        if robustness is not None:
            summary["robustness_fps"] = {k: [v["speculated"]["fps"], v["unspeculated"]["fps"]] for k, v in robustness.items() if isinstance(v, dict)}
        if per_rank is not None:
            summary["per_rank_wire_bytes_per_frame"] = per_rank["wire_bytes_per_frame"]
        if lib_index:
            # what RCCL itself says about the communicator the frames ran over (ncclCommCount / ncclCommUserRank / ncclGetVersion)
            out["rccl"] = comm_info
        # BASELINE configs[4] on this GPU (the main scene's buffers are released first) and the oracle check of the headline config
        renderer.close()
        closed = True
        cpu_res = None
        if single and not overrides and not args.no_cfg5 and not args.no_extra_legs:
            out["cfg5"] = cfg5_leg()
            summary["cfg5_fps"] = [out["cfg5"]["fps_one_frame_in_flight"], out["cfg5"]["fps_two_frames_in_flight"], out["cfg5"]["fps_unspeculated"]]
        if world == 1 and not args.no_cpu_baseline:
            cpu_res, oracle_frame = cpu_baseline(cfg, args.cpu_sample, keep=True)
            if gpu_pose0 is not None and oracle_frame is not None:
                out["oracle_check"] = oracle_check_of(gpu_pose0, oracle_frame, 0)
                summary["oracle_check"] = {k: out["oracle_check"][k] for k in ("ok", "linf_plain", "linf_speculated", "keys_equal", "rects_equal",
                                                                                "depth_order_equal", "n_visible_equal")}
        out["summary"] = summary
        if cpu_res is not None:
            out["cpu_baseline"] = cpu_res
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if "oracle_check" in out and not out["oracle_check"]["ok"]:
            raise SystemExit(f"bench.py: the HIP frame of pose 0 differs from the oracle's: {out['oracle_check']}")

    if not closed:
        renderer.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
