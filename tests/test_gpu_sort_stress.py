"""GPU: the radix sort alone, at sizes and key distributions the frame tests only reach by accident — depth keys (one or
two exponent bytes: whole waves share a digit in the last pass), tile keys, element counts around the persistent grid's
multiples — against std::stable_sort, with a time limit: a pass that hangs is a failure, not a stuck test run.
Regression: the exit flag of k_radix_onesweep shared an LDS word with the ticket; a late lane re-entered the loop alone
and hung the pass (seen at 6 M / 8.4 M / 12 M depth keys once the ranking phase got faster)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "bench_sort")

CASES = [("6000000", "32", "depth"), ("8400000", "32", "depth"), ("12000000", "32", "depth"), ("300000", "32", "depth"),
         ("8400000", "32", "dup"), ("300000", "32", "dup"), ("5300000", "13"), ("4096", "32"), ("4097", "13"), ("3145728", "32"), ("3145729", "32", "depth"), ("1", "32")]


@pytest.mark.parametrize("ranks", ["lane_ordered", "match"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(c))
def test_sort_matches_stable_sort(case, ranks):
    assert os.path.exists(EXE), "tools/bench_sort missing: run __graft_entry__.build()"
    env = dict(os.environ)
    env.pop("GSX_RADIX_MATCH_RANKS", None)
    if ranks == "match":
        env["GSX_RADIX_MATCH_RANKS"] = "1"
    p = subprocess.run([EXE, *case], capture_output=True, text=True, timeout=60, env=env)
    assert p.returncode == 0, p.stderr[-400:]
    assert "\nmismatches vs std::stable_sort: 0" in p.stdout, p.stdout[-600:]
    want = "lane-ordered on this device: " + ("0" if ranks == "match" else "1")
    assert want in p.stdout, p.stdout[:200]
    # the bucket sort (histogram -> MSD partition -> bucket sorts) on the same pairs: without a key range, with the range of the sort
    # before, with a device-side count below the launch bound, empty, and from a key array — all equal to std::stable_sort
    assert "bucket sort mismatches in all: 0" in p.stdout, p.stdout[-1200:]


@pytest.mark.parametrize("ranks", ["lane_ordered", "match"])
@pytest.mark.parametrize("case", [("300000", "32", "depth"), ("300000", "32", "dup"), ("40000", "32", "depth"), ("4097", "13"), ("70000", "32"), ("1", "32")],
                         ids=lambda c: "-".join(c))
def test_bucket_sort_large_bucket_path(case, ranks):
    """GSX_BUCKET_CAP=48: every bucket of more than 48 pairs is sorted through global memory, tile by tile, by its one workgroup — the path
    a bucket that does not fit the LDS takes (8.4 M-pair sorts reach it by themselves in the test above).  Same order."""
    env = dict(os.environ, GSX_BUCKET_CAP="48")
    env.pop("GSX_RADIX_MATCH_RANKS", None)
    if ranks == "match":
        env["GSX_RADIX_MATCH_RANKS"] = "1"
    p = subprocess.run([EXE, *case], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0, p.stderr[-400:]
    assert "bucket sort mismatches in all: 0" in p.stdout, p.stdout[-1200:]
