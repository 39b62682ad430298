#!/usr/bin/env python3
"""bench.py — frames/sec of the 3DGS render path on N MI355X (one process per GPU).

A "step" is one frame: the whole hot path (projection -> depth sort -> tile binning -> tile sort ->
composite [-> exchange/merge when N > 1]) over the resident synthetic scene at the next pose of the
benchmark orbit.  Inputs are resident in HBM before the timed region starts.
Prints ONE JSON line on rank 0.  The CPU oracle is used here only for the `cpu_baseline` leg.

    python bench.py                       # N=1, cfg4 scene (10 M Gaussians, SH-3, 1920x1080)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg4", help="cfg2 (1 M) | cfg3 (5.8 M) | cfg4 (10 M, headline)")
    ap.add_argument("--gaussians", type=int, default=0, help="override the Gaussian count (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="Gaussians in the CPU baseline sample")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded exchange path even at N=1")
    ap.add_argument("--shard-mode", default="frames", help="N > 1: frames = whole scene on every GPU, rank g renders every N-th frame "
                    "(frame-parallel; RGBA8 frames all-gathered) | screen = whole scene on every GPU, rank g renders band g of "
                    "tile rows, one all-gather | index = splat-index shards + speculative record exchange")
    ap.add_argument("--pose-stride", type=int, default=1, help="N = 1 experiment: render every S-th pose of the orbit — what one rank of "
                    "the frame-parallel mode at S GPUs sees (its speculation looks S poses back)")
    ap.add_argument("--gather", default="rgba8", help="N > 1, screen mode: rgba8 = every rank resolves its band (the app's blit to its "
                    "Rgba8Unorm surface) and 4 bytes a pixel are all-gathered on a second stream, under the next frame | float = the "
                    "(rgb, T) bands, 16 bytes a pixel, in stream order")
    ap.add_argument("--pass-timing", default="project", help="project (the roofline kernel only; default) | all (every pass, adds "
                    "a few microseconds of stream gap per pass boundary)")
    ap.add_argument("--render-options", default="", help="gsx_render_options overrides, e.g. speculative=0,min_slab=1000000 (experiments)")
    ap.add_argument("--host-profile", action="store_true", help="print host wall time per exchange-protocol section (adds syncs; debug)")
    ap.add_argument("--pod", default="single/single", help="pod storage sh/cov3d: single|half|norm8|none / single|half "
                    "(reference default is norm8/half; the headline metric is quoted on the f32 pod)")
    return ap.parse_args()


def cpu_baseline(cfg, n_sample, pose=0):
    """Oracle (reference algorithm shape: cull -> global radix sort -> back-to-front splat-major raster) on the
    host cores, one frame of a bounded sample of the same scene."""
    import oracle
    from wgpu_3dgs_viewer_app_amd import camera, scene

    n, sh, w, h, seed = cfg
    n_sample = min(n, n_sample)
    g = scene.synthetic_gaussians(n, seed, sh, 0, n_sample)
    pos, color, shc, cov = oracle.convert(g)
    cam = camera.orbit_pose(pose)
    f = oracle.frame_setup(cam.view(), cam.projection(w / h), w, h)
    fb = oracle.new_framebuffer(f)
    oracle.render_model(f, pos[:1000], color[:1000], shc[:1000], cov[:1000], fb)  # page in / thread start
    fb = oracle.new_framebuffer(f)
    t0 = time.perf_counter()
    nvis = oracle.render_model(f, pos, color, shc, cov, fb)
    dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 4), unit="frames/s", cores=oracle.num_threads(), kind="port",
                sample=f"1 frame (orbit pose {pose}) of the first {n_sample} of {n} Gaussians of the same scene at {w}x{h}, "
                       f"N_vis={nvis}, {dt:.2f} s; oracle/gsx_oracle.c, OpenMP")


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from wgpu_3dgs_viewer_app_amd import camera, parallel, scene

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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
    start, count = parallel.shard_range(n, rank, world) if args.shard_mode == "index" else (0, n)
    t0 = time.perf_counter()
    g = scene.synthetic_gaussians(n, seed, sh, start, count)
    t_gen = time.perf_counter() - t0
    sh_kind = {"single": 0, "half": 1, "norm8": 2, "none": 3}[args.pod.split("/")[0]]
    cov_kind = {"single": 0, "half": 1}[args.pod.split("/")[1]]
    afr = use_dist and args.shard_mode == "frames"   # frame-parallel: a step is still ONE frame of the orbit; N ranks render N per round
    gather = "rgba8" if afr else (args.gather if (use_dist and args.shard_mode == "screen") else "float")
    renderer = parallel.ShardedViewer(device=local_rank, world=world, rank=rank, use_dist=use_dist, sh=sh_kind, cov3d=cov_kind,
                                      mode=args.shard_mode if use_dist else "index", gather=gather, overlap_gather=gather == "rgba8")
    if args.render_options:
        renderer.stages.viewer.set_render_options(**{k: float(x) if "." in x else int(x) for k, x in
                                                     (kv.split("=") for kv in args.render_options.split(","))})
    t0 = time.perf_counter()
    renderer.load_shard(g, start, n)
    renderer.poll()
    t_up = time.perf_counter() - t0
    upload_gbs = g.nbytes / t_up / 1e9
    del g

    # the orbit's cameras (inputs of the path) are prepared before the clock starts: look_at / perspective in numpy cost
    # ~30 us a pose, which an un-synchronised frame loop hides but a host that waits for the device (host_verify) does not
    orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]

    def frame(i):
        renderer.render_frame(orbit[(i * args.pose_stride) % 240], (w, h))

    if afr:
        # round j of the orbit = frames j * world .. j * world + world - 1; this rank renders frame j * world + rank
        one_frame = frame

        def frame(j, limit=None):  # noqa: F811 — limit: frames of the orbit that exist in this loop (the last round may be partial)
            if limit is None or j * world + rank < limit:
                one_frame(j * world + rank)
            else:
                renderer.skip_frame()

    # working buffers (records, sort and tile-pair buffers: sized by the scene) are allocated by the first frame a model is
    # rendered in; that belongs to loading the scene, not to a step
    frame(0)
    renderer.poll()
    per = world if afr else 1     # frames of the orbit per loop iteration
    rounds = lambda k: (k + per - 1) // per  # noqa: E731
    for i in range(rounds(args.warmup)):
        frame(i)
    renderer.poll()
    renderer.set_pass_timing(True, None if args.pass_timing == "all" else ["project"])
    renderer.get_pass_timing()  # reset accumulators
    if args.host_profile and use_dist:
        renderer.profile = {}

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    if afr:   # exactly args.steps frames of the orbit, dealt round-robin; ranks without a frame in the last round only gather
        first = rounds(args.warmup)
        for i in range(rounds(args.steps)):
            frame(first + i, limit=first * world + args.steps)
    else:
        for i in range(args.steps):
            frame(args.warmup + i)
    renderer.poll()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    timing = renderer.get_pass_timing()
    if args.host_profile and use_dist and rank == 0:
        print("host ms/frame by section:", {k: round(1e3 * x / args.steps, 4) for k, x in renderer.profile.items()}, file=sys.stderr)
    renderer.set_pass_timing(False)

    # Untimed accounting pass over the same poses: per-frame counts for the projection kernel's algorithmic bytes (the
    # statistics live on the device; reading them costs a sync per frame, which the timed loop must not pay).
    renderer.profile = None
    acct = []
    for i in range(min(rounds(args.steps), 64)):
        frame(rounds(args.warmup) + i)
        st = renderer.last_stats()
        acct.append((st["n_gaussians"], st["n_visible"], st.get("n_sorted", st["n_visible"]), 1 if st.get("speculated") else 0,
                     st.get("n_repair_tiles", 0), st["n_tile_entries"]))
    acct = np.asarray(acct, np.float64)
    stats = renderer.last_stats()

    # gather per-rank numbers for the roofline of the projection pass (dominant HBM stream of the path)
    local = torch.tensor([acct[:, 0].mean(), acct[:, 1].mean(), acct[:, 2].mean(), acct[:, 3].mean(), float((acct[:, 4] > 0).mean()),
                          acct[:, 5].mean(), timing["project"]["ms"] * 1e3, timing["project"]["launches"]], dtype=torch.float64, device="cuda")
    if use_dist:
        allr = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(allr, local)
    else:
        allr = [local]
    allr = torch.stack(allr).cpu().numpy()

    if rank == 0:
        fps = args.steps / elapsed
        # Projection kernel, ALGORITHMIC bytes per launch (means over the accounting frames of rank 0's shard):
        #   every frame      reads N*16 (pos + rgba8) + N_vis*cov, writes N*4 (depth key) + N_vis*16 (mean, tile rect) + N/8 (ballots)
        #   shaded Gaussians read the SH planes and write conic/opacity + colour/depth: + N_shaded*(sh + 32)
        # An unspeculated frame shades every visible Gaussian in this kernel: that is SURVEY 8d's N*pod + N_vis*40 up to
        # bookkeeping.  A speculated frame's projection pass is geometry only — the few admitted Gaussians are shaded by
        # k_shade afterwards — so the kernel moves fewer bytes BY DESIGN and is priced on what it has to move.
        n_loc, nvis_loc, nsort_loc, spec_frac, repair_frac, entries = allr[0][:6]
        sh_bytes = {0: 180, 1: 96, 2: 48, 3: 0}[sh_kind if sh > 0 else 3]
        cov_bytes = {0: 24, 1: 12}[cov_kind]
        pod_bytes = 16 + sh_bytes + cov_bytes
        n_shaded = (1.0 - spec_frac) * nvis_loc  # a speculated frame's projection pass is geometry only (k_shade does the rest)
        proj_bytes = n_loc * 16 + nvis_loc * cov_bytes + n_loc * 4 + nvis_loc * 16 + n_loc / 8 + n_shaded * (sh_bytes + 32)
        survey_bytes = n_loc * pod_bytes + nvis_loc * 40
        proj_us = allr[0][6] / max(allr[0][7], 1)
        achieved = proj_bytes / (proj_us * 1e-6) / 1e9 if proj_us > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        variant = "lazy" if spec_frac > 0.5 else "full"
        if os.path.exists(pmc) and args.pod == "single/single" and not args.render_options:
            try:
                rec = json.load(open(pmc)).get(f"{args.workload}:{world}:k_project:{variant}")
                traffic = rec["hbm_bytes_per_launch"] if rec else None
            except Exception:
                traffic = None
        passes = {}
        for name, tv in timing.items():
            if args.pass_timing == "all" or name == "project":
                passes[name] = round(tv["ms"] / max(rounds(args.steps), 1), 4)
        out = {
            "metric": "frames/sec @1920x1080, N-Gaussian SH3 scene, 1/2/4/8 MI355X; %HBM roofline",
            "value": round(fps, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: synthetic {n} Gaussians SH-deg-{sh}, {w}x{h}, orbit r=6 h=1.5 240 poses, seed {seed}",
                "gaussians": n, "width": w, "height": h, "sh_degree": sh, "pod": args.pod, "pod_bytes": pod_bytes,
                "sharding": ("one GPU" if not use_dist else
                             f"frame-parallel x{world}: scene resident on each GPU, rank g renders frame j*{world}+g of the orbit "
                             "(one-frame latency as on one GPU), RGBA8 frames all-gathered on a second stream" if afr else
                             (f"splat-index shards x{world}, speculative record exchange by tile-row band + band all-gather"
                              if args.shard_mode == "index" else
                              f"scene resident on each of {world} GPUs, rank g renders band g of tile rows, band all-gather "
                              + ("of RGBA8 pixels (resolved per band) on a second stream" if gather == "rgba8" else "of (rgb, T) float4 pixels"))),
                "n_visible_rank0": int(nvis_loc), "n_depth_sorted_rank0": int(nsort_loc), "tile_entries_rank0": int(entries),
                "speculated_frames": round(float(spec_frac), 3), "frames_with_repair_round": round(float(repair_frac), 3),
                "pass_ms_per_frame_rank0": passes,
                "upload_GBps_pcie_inclusive": round(upload_gbs, 2), "scene_gen_s": round(t_gen, 1),
            },
            "roofline": {
                "kernel": (f"k_project_geom<{cov_kind},1> (projection pass of a speculated frame: cov2d + cull + depth key + admission; "
                           "SH colour is evaluated by k_shade for the admitted Gaussians only)" if variant == "lazy" else
                           f"k_project<{sh},{sh_kind},{cov_kind}> (projection pass: SH colour + cov2d + cull + depth key)"),
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(proj_bytes), "avg_launch_us": round(proj_us, 2),
                "bytes_definition": ("geometry-only projection of a speculated frame: N*20.125 + N_vis*(16+cov)" if variant == "lazy" else
                                     "N*20.125 + N_vis*(16+cov+sh+32) (= SURVEY 8d up to bookkeeping)"),
                "survey_8d_bytes_per_launch": int(survey_bytes),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample)
        print(json.dumps(out), flush=True)

    renderer.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
