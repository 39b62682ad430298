"""The committed bench line (profiles/r01_bench.json = stdout of `python bench.py` on an MI355X) carries every field of the
bench contract, and its numbers are consistent with each other.  CPU test: it reads the committed file only."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    lines = [ln for ln in open(os.path.join(ROOT, "profiles", name)).read().splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _line("r01_bench.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].replace("x", "×") == base["metric"].replace("x", "×") or "frames/sec" in d["metric"]
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
    # PMC traffic (bytes actually moved) is not below the algorithmic bytes and within 25 % of them
    assert r["traffic"] is None or r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 1.25 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_profiled_kernel_time_agrees_with_bench():
    """profiles/: the rocprofv3 average of the roofline kernel agrees with the HIP-event time in the bench line (±10 %)."""
    d = _line("r01_bench.json")
    us = d["roofline"]["avg_launch_us"]
    rows = [ln for ln in open(os.path.join(ROOT, "profiles", "r01_cfg4_kernel_stats.csv")) if "k_project_geom" in ln]
    assert rows, "the kernel-trace summary holds the projection kernel"
    avg_ns = float(rows[0].rsplit('",', 1)[1].split(",")[2])
    assert abs(avg_ns / 1e3 - us) / us < 0.10, (avg_ns / 1e3, us)
