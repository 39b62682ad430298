"""GPU: `bench.py --gpus N` exactly as the driver launches it (torch.distributed.run, one process per rank, RCCL inside libgsx
and under torch.distributed) — with N processes sharing the box's ONE GPU (GSX_BENCH_ONE_DEVICE=1: every rank takes device 0
and a host identity of its own, RCCL connects them over sockets on `lo`).  What is checked is that the N > 1 path of the
bench and of the library runs to its JSON line between real processes and that the line is consistent; the rate on it measures
nothing (the line says so).  If RCCL will not connect the ranks here the run fails at init and the test is skipped."""
import json
import os
import re
import subprocess
import warnings
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, extra, port, timeout=420):
    env = dict(os.environ)
    env.update({"GSX_BENCH_ONE_DEVICE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONPATH": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "12", "--warmup", "4",
           "--min-steps", "0", "--gaussians", "400000"] + extra
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, 9)   # the launcher's own process group (start_new_session): exactly what was started here
        out, err = p.communicate()
        pytest.fail(f"bench.py --gpus {world} {extra} did not finish in {timeout} s\n{err[-3000:]}")   # a hang is a finding: never retried
    if p.returncode != 0 and ("ncclInvalidUsage" in err or "Duplicate GPU" in err or "NCCL error" in err and "init" in err.lower()):
        pytest.skip("RCCL would not connect several ranks on one GPU here:\n" + err[-1500:])
    if p.returncode != 0 and RENDEZVOUS.search(err) and not RANK_ERROR.search(err):
        raise RendezvousError(err[-3000:])
    assert p.returncode == 0, err[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 prints ONE JSON line, got {len(lines)}:\n{out[-2000:]}"
    return json.loads(lines[0])


class RendezvousError(Exception):
    """the launcher's own rendezvous did not come up (port taken, store connection refused): the one failure that is retried"""


# positively identified launcher / rendezvous failures; anything a rank reports about its frames or the library is not one
RENDEZVOUS = re.compile(r"EADDRINUSE|Address already in use|RendezvousConnectionError|RendezvousTimeoutError|TCPStore|DistNetworkError|"
                        r"Connection refused|connection refused|failed to bind|could not bind")
RANK_ERROR = re.compile(r"GsxError|GSX_ERR|differs from|frame_check|Traceback[^\n]*\n(?:.*\n)*?.*bench\.py")


CASES = [(2, []), (2, ["--dist-frames-in-flight", "1"]), (3, []), (2, ["--shard-mode", "frames"]), (2, ["--shard-mode", "screen"])]


@pytest.mark.parametrize("case", range(len(CASES)), ids=[f"world{w}" + "".join(x.replace("--", "-") for x in e) for w, e in CASES])
def test_bench_with_ranks_as_processes(case):
    world, extra = CASES[case]
    port = 29531 + 3 * case + (os.getpid() % 200)   # a rendezvous port of its own per case and per pytest process
    try:
        d = _launch(world, extra, port)
    except RendezvousError as e:   # the port was taken / the store refused: once more on another port.  A timeout, a rank that exits
        # non-zero or reports an error is NOT retried (it would hide an intermittent hang or race in the collectives).
        warnings.warn(f"torch.distributed.run did not rendezvous on port {port}: one more attempt on {port + 1000}\n{str(e)[-800:]}")
        d = _launch(world, extra, port + 1000)
    assert d["n_gpus"] == world and d["steps"] == 12 and d["warmup"] == 4 and d["value"] > 0
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-2
    assert "one_device_emulation" in d and d["overflow_slabs"] == 0
    assert d["scaling"] == ("weak" if "frames" in extra else "strong")
    if "--shard-mode" not in extra:
        pr = d["per_rank"]
        assert len(pr["shard_gaussians"]) == world and sum(pr["shard_gaussians"]) == 400000
        assert all(b > 0 for b in pr["wire_bytes_per_frame"]), "every rank put records on the links"
        assert all(r >= 1.0 for r in pr["exchange_rounds_per_frame"])
        assert d["frame_check"]["equal_to_single_gpu_single_pass_frame"] is True, d["frame_check"]
