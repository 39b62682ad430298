"""The bench line carries every field of the bench contract and its numbers are consistent with each other.  CPU test.
Source: the newest DRIVER-written BENCH_r*.json at the repo root when one exists (its "tail" is the line bench.py printed on the
driver's box), and the builder-kept copy of the same command under profiles/ (profiles/r02_bench.json) — both are checked."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lines():
    out = []
    drv = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")), key=lambda p: int(re.findall(r"r(\d+)", os.path.basename(p))[0]))
    if drv:
        d = json.load(open(drv[-1]))
        line = d.get("parsed") or (json.loads(d["tail"]) if d.get("tail", "").strip().startswith("{") else None)
        if line:
            out.append((os.path.basename(drv[-1]), line))
    kept = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    if kept:
        rows = [ln for ln in open(kept[-1]).read().splitlines() if ln.strip()]
        assert len(rows) == 1, "bench.py prints ONE JSON line"
        out.append((os.path.basename(kept[-1]), json.loads(rows[0])))
    return out


@pytest.mark.parametrize("name,d", _lines(), ids=[n for n, _ in _lines()])
def test_bench_line_contract(name, d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "frames/sec" in d["metric"]
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "cfg4" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-2
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / measured launch time
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-2
    # PMC traffic (bytes actually moved) is within 25 % of the algorithmic bytes
    assert r["traffic"] is None or 0.9 * r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 1.25 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    if "value_unspeculated" in d:  # round 2 onwards: one run carries both loops, the SURVEY-8d kernel and the frame check
        assert "k_project<3,0,0>" in r["kernel"] and "SURVEY 8d" in r["bytes_definition"]
        assert r["algorithmic_bytes_per_launch"] == int(r["n"]) * 220 + int(r["n_visible"]) * 40 or abs(
            r["algorithmic_bytes_per_launch"] - (r["n"] * 220 + r["n_visible"] * 40)) < 1e6
        assert d["frame_check"]["equal_to_unspeculated_single_pass"] is True and d["overflow_slabs"] == 0
        assert "whole scene" in c["sample"]
        assert 0 < d["value_unspeculated"] < d["value"]
        assert "k_project_geom" in d["roofline_speculated"]["kernel"]
    if "passes" in d:  # round 3 onwards: what a host that waits gets, a whole-orbit figure, every pass against its bytes
        assert 0 < d["value_synchronised_unspeculated"] < d["value_synchronised"] <= 1.1 * d["value_one_frame_in_flight"]
        assert 0 < d["value_reference_protocol"] <= 1.1 * d["value_synchronised"]
        assert abs(d["value_synchronised"] - 1e3 / d["ms_per_step_synchronised"]) / d["value_synchronised"] < 1e-2
        assert d["steady_state"]["frames"] >= 240 and 0.7 * d["value"] < d["steady_state"]["value"] < 1.3 * d["value"]
        assert "value_unspeculated" in d["roofline"]["belongs_to"]
        full = "full_projection" if "full_projection" in d["passes"] else "unspeculated"   # (round 6: unspeculated frames shade slab by slab)
        schedules = [("speculated", "project_geom"), (full, "project")] + ([("unspeculated", "project_geom")] if full != "unspeculated" else [])
        for schedule, proj in schedules:
            ps = d["passes"][schedule]
            for name in (proj, "depth_sort", "bin", "tile_sort", "composite"):
                row = ps[name]
                assert row["us_per_frame"] > 0 and row["algorithmic_bytes"] > 0 and "bytes_definition" in row
                assert abs(row["GBps"] - row["algorithmic_bytes"] / (row["us_per_frame"] * 1e-6) / 1e9) / row["GBps"] < 2e-2
                assert abs(row["frac_of_8TBps"] - row["GBps"] / 8000.0) < 1e-3
            # the bracketed passes account for the frame they were measured in (event brackets leave gaps: not more than the frame)
            total = sum(ps[n]["us_per_frame"] for n in (proj, "depth_sort", "bin", "tile_sort", "composite"))
            assert 0.7 * ps["frame_ms_with_every_pass_bracketed"] * 1e3 < total <= 1.05 * ps["frame_ms_with_every_pass_bracketed"] * 1e3
        # the projection pass of the `passes` loop and the roofline kernel of the unspeculated timed loop are the same kernel
        assert abs(d["passes"][full]["project"]["us_per_frame"] - d["roofline"]["avg_launch_us"]) / d["roofline"]["avg_launch_us"] < 0.15
        if full != "unspeculated":
            assert 0 < d["value_unspeculated_full_projection"] < d["value_unspeculated"] < d["value"]
            assert "value_unspeculated_full_projection" in d["roofline"]["belongs_to"]


    if "summary" in d:  # round 5 onwards: `value` is a whole-orbit rate and the side-by-side figures are the last keys of the line
        assert d["steps_timed"] >= 240 and d["steps_timed"] >= d["steps"]
        keys = list(d)
        assert keys[-2:] == ["summary", "cpu_baseline"] or keys[-1] == "summary"
        sm = d["summary"]
        assert sm["value"] == d["value"] and sm["steps_timed"] == d["steps_timed"]
        for k in ("value_one_frame_in_flight", "value_synchronised", "value_reference_protocol", "value_unspeculated"):
            if k in d:
                assert sm[k] == d[k], k
        if "steady_state" in d:
            assert sm["steady_state_value"] == d["steady_state"]["value"]
            assert 0.85 * d["value"] < d["steady_state"]["value"] < 1.15 * d["value"]  # two whole-orbit samples of one loop
        assert sm["roofline_frac"] == d["roofline"]["frac"]
        # what the library holds on the device for the scene: at least the planes, at most a few KB per Gaussian
        rb = d["resident_bytes"]
        assert 200 * rb["gaussians_of_this_rank"] < rb["device_bytes_after_headline_loop"] < 4096 * rb["gaussians_of_this_rank"]


@pytest.mark.parametrize("name,d", _lines(), ids=[n for n, _ in _lines()])
def test_bench_line_round6_keys(name, d):
    """Round 6 onwards: the HIP frame of the pose the CPU baseline rendered is compared with that frame on
    the line itself — integer stages bit-exact, both schedules within 1e-3 — and BASELINE configs[4] has a compact leg."""
    if "oracle_check" not in d and "cfg5" not in d:
        pytest.skip("a line from before round 6")
    oc = d["oracle_check"]
    assert oc["ok"] is True and oc["keys_equal"] and oc["rects_equal"] and oc["depth_order_equal"] and oc["n_visible_equal"]
    assert 0.0 <= oc["linf_plain"] <= 1e-3 and 0.0 <= oc["linf_speculated"] <= 1e-3 and oc["speculated_frame_was_speculated"]
    assert oc["speculated_equals_plain"] is True
    assert str(oc["n_visible"]) in d["cpu_baseline"]["sample"], "the checked frame is the one cpu_baseline timed"
    assert d["summary"]["oracle_check"]["ok"] is True
    c5 = d["cfg5"]
    assert "4 x 6000000" in c5["workload"] and "3840x2160" in c5["workload"] and "0 - 1" in c5["workload"]
    assert c5["frame_check"]["equal_to_unspeculated_single_pass"] is True and c5["overflow_slabs"] == 0
    assert 0 < c5["fps_unspeculated"] < c5["fps_one_frame_in_flight"] and c5["fps_two_frames_in_flight"] > 0
    assert d["summary"]["cfg5_fps"] == [c5["fps_one_frame_in_flight"], c5["fps_two_frames_in_flight"], c5["fps_unspeculated"]]


def test_bench_source_carries_the_rccl_field_on_sharded_lines():
    """every N > 1 line says how many ranks RCCL itself counted (gsx_viewer_comm_info: ncclCommCount), so a SCALE record can be checked"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'out["rccl"] = comm_info' in src and "viewer.comm_info()" in src
    hdr = open(os.path.join(ROOT, "include", "gsx.h")).read()
    assert "gsx_viewer_comm_info" in hdr and "ncclCommCount" in hdr


def test_bench_source_prints_the_summary_last():
    """bench.py assigns out["summary"] directly before out["cpu_baseline"], the last key of the line (CPU check of the source)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    i, j = src.index('out["summary"] = summary'), src.index('out["cpu_baseline"] = cpu_res')
    assert i < j and 'out["' not in src[i + len('out["summary"] = summary'):j].replace('out["cpu_baseline"]', "")
    assert src.index("os.write(real_stdout") > j
    assert "args.steps = max(args.steps, args.min_steps)" in src and '"--min-steps", type=int, default=240' in src


def test_profiled_kernel_time_agrees_with_bench():
    """profiles/: the rocprofv3 average of the roofline kernel agrees with the HIP-event time in the kept bench line (+-10 %)."""
    kept = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    stats = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg4_kernel_stats.csv")))
    if not kept or not stats:
        pytest.skip("no kept bench line / kernel stats")
    d = json.loads(open(kept[-1]).read().strip())
    kernel = "k_project<3, 0, 0>" if "value_unspeculated" in d else "k_project_geom"
    us = d["roofline"]["avg_launch_us"]
    rows = [ln for ln in open(stats[-1]) if kernel in ln]
    assert rows, "the kernel-trace summary holds the projection kernel"
    avg_ns = float(rows[0].rsplit('",', 1)[1].split(",")[2])
    assert abs(avg_ns / 1e3 - us) / us < 0.10, (avg_ns / 1e3, us)
    # and the line the PROFILED run printed itself (same process as the trace: no run-to-run variance in between) within 5 %
    own = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg4_kernel_stats_bench_line.json")))
    if own:
        o = json.loads(open(own[-1]).read().strip())
        us_own = o["roofline"]["avg_launch_us"]
        assert abs(avg_ns / 1e3 - us_own) / us_own < 0.05, (avg_ns / 1e3, us_own)
