"""GPU: the other side of every A/B switch libgsx still reads must keep rendering the same frames.

Round 6 replaced three pieces of the frame by default — the depth sort of small sets (histogram -> MSD partition -> per-bucket LDS
sorts, GSX_BUCKET_SORT), the binning kernels (one fused count / look-back / emit launch, GSX_BIN_FUSED) and the projection of
unspeculated frames (geometry only + per-slab shading, GSX_SLAB_SHADING = the default of gsx_render_options.slab_shading) — and kept
the round-5 paths behind their switches for the same-box A/Bs under profiles/r06_ab_*.txt.  A switch that stays is a path that must
stay right: the float64 fixtures, the overflow / spill suite and the slab suite run again with each switch on its other side.
(DESIGN.md "Environment switches" lists every GSX_* variable; tests/test_oracle_cpu.py checks that list against the sources.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("switch", ["GSX_BUCKET_SORT=0", "GSX_BIN_FUSED=0", "GSX_SLAB_SHADING=0", "GSX_BUCKET_SORT=0 GSX_BIN_FUSED=0 GSX_SLAB_SHADING=0"])
def test_round5_paths_behind_their_switches(switch):
    if os.environ.get("GSX_SWITCH_RERUN"):
        pytest.skip("already inside the rerun")
    env = dict(os.environ, GSX_SWITCH_RERUN="1")
    for kv in switch.split():
        k, val = kv.split("=")
        env[k] = val
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_golden.py", "tests/test_gpu_overflow.py",
                        "tests/test_gpu_speculation.py", "-k", "not long_run"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, f"{switch}: " + p.stdout[-1500:] + p.stderr[-500:]
