"""The collective calls bench.py and raytracer_amd/dist.py make at N > 1, issued on a one-rank RCCL group (the GPU box has
one GPU): asynchronous gather into views of one tensor, all_gather_into_tensor of uint8, the int64 sum and float64 max
all-reduces, barrier.  With one rank every collective is a copy, so the values can be checked; what the test pins is
that RCCL on this image accepts these calls, dtypes and the async form.  Runs in a child process (a process group in the
pytest process would outlive the test)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from raytracer_amd.dist import FrameGatherer, sum_over_ranks
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
H, W = 36, 64
g = FrameGatherer(H, W, 1, 8, 0, 1, dev, dtype=torch.int32)     # for its buffer shapes
# FrameGatherer.gather_begin / gather_end at world > 1: async gather into views of one tensor, then wait
local = torch.arange(g.pad_rows * W, dtype=torch.int32, device=dev).reshape(g.pad_rows, W)
recv_all = torch.zeros((1, g.pad_rows, W), dtype=torch.int32, device=dev)
work = dist.gather(local, [recv_all[0]], dst=0, async_op=True)
other = local * 2                       # work on the current stream while the gather is in flight
work.wait()
torch.cuda.synchronize()
assert torch.equal(recv_all[0], local) and torch.equal(other, local * 2)
bits = torch.full((g.pad_rows, W), 5, dtype=torch.uint8, device=dev)
bits_all = torch.zeros((g.pad_rows, W), dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(bits_all, bits)
assert torch.equal(bits_all, bits)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
# the call bench.py makes; world 2 so that it does reduce (on the one-rank group: the value itself)
assert sum_over_ranks(123456789012, dev, 2) == 123456789012
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl calls ok")
"""


def test_rccl_accepts_the_collectives_of_the_multi_rank_path(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500), HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0 and "rccl calls ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
