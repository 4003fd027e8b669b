"""RCCL on the GPU box: the calls bench.py's N > 1 path makes (process group with a device id, the asynchronous
sum of a frame's accumulator while the next frame renders, barrier, max of the ranks' times), in a process group
of ONE rank -- RCCL refuses two ranks on one device, and the box has one.  What it shows: the library's own streams,
torch's stream and RCCL's stream hand the accumulators to each other in the order bench.py relies on."""
import os
import subprocess
import sys

import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import conftest
from orion_amd import api, dist as odist
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sc = api.Scene(conftest.load_model("cornell"), device=0)
cam = api.camera_vectors([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0, 1.5)
W, H, SPP = 192, 128, 8
def render(accum, b, e, tile):
    sc.render(cam, W, H, b, e, 2, 8, seed=4, tile=tile, accum=accum)
want = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
render(want, 0, SPP, None)
plan = odist.shard_plan(SPP, W, H, 0, 1, (3, 2))
accs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
pending = [None, None]
for frame in range(4):                      # bench.py's step(): two accumulators in turn
    k = frame % 2
    if pending[k] is not None:
        pending[k].wait(); pending[k] = None
    accs[k].zero_()
    odist.render_plan(render, plan, accs[k])
    pending[k] = dist.all_reduce(accs[k], op=dist.ReduceOp.SUM, async_op=True)     # (reduce_async returns None for one rank)
for w in pending:
    w.wait()
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
for a in accs:
    assert torch.equal(a[..., 3], want[..., 3])
    assert torch.allclose(a, want, rtol=1e-5, atol=1e-6)      # (tiles add their batches in another order than the whole frame)
assert odist.reduce_async(accs[0]) is None
dist.destroy_process_group()
sc.close()
print("rccl ok")
"""


def test_bench_exchange_through_rccl_with_one_rank(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=conftest.ROOT, port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=150, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
