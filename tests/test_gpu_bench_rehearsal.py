"""bench.py's N > 1 control flow on the one-GPU box: both ranks on GPU 0, gathers over gloo (BENCH_REHEARSE=1; RCCL
refuses two ranks on one device).  Launched the way the driver launches it -- torch.distributed.run, one process per
rank -- from child processes that have not touched the GPU before.  Checks the JSON line, and that the two-rank frame
traces the same rays as the one-rank frame (the frame does not depend on the partition)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workload", "c3", "--width", "480", "--roofline-frames", "1"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _run(n, extra=()):
    env = dict(os.environ, BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    bench = os.path.join(ROOT, "bench.py")
    if n == 1:
        cmd = [sys.executable, bench, "--gpus", "1"] + COMMON + list(extra)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), bench, "--gpus", str(n)] + COMMON + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return _line(r.stdout)


def test_bench_two_ranks_pipelined_and_not():
    one = _run(1)
    assert one["n_gpus"] == 1 and one["steps"] == 2 and one["warmup"] == 1
    for extra in ((), ("--no-pipeline",)):
        two = _run(2, extra)
        assert two["n_gpus"] == 2 and two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "strong"
        assert two["metric"] == "Mrays/s" and two["unit"] == "Mrays/s" and two["value"] > 0 and two["ms_per_step"] > 0
        assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
        assert "cpu_baseline" not in two
        r = two["roofline"]
        # the dominant kernel's own line (time measured here; executed FLOP only where a committed counter pass of the
        # workload exists, which this reduced frame has not) and the two whole-frame figures by name
        assert r["bound"] in ("fp64_valu", "mfma_f16") and r["kernel_ms"] > 0 and r["second_kernel"]["kernel_ms"] > 0
        assert r["frac_algorithmic_equiv"] > 0 and r["frac_in_contract"] > 0
        assert r["in_contract"]["same_output_bits"] is True and r["in_contract"]["pairs_evaluated_frac"] >= 0.999
        assert two["grid_walk"]["same_output_bits"] is True and two["grid_walk"]["frame_sha256_16"] == two["config"]["frame_sha256_16"]
        assert two["config"]["frame_sha256_16"] == one["config"]["frame_sha256_16"]
        assert ("overlaps the next frame" in two["config"]["step"]) == (extra == ())
