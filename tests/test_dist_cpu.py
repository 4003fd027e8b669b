"""The N > 1 path on CPU: world_size-2 gloo processes shard a frame by sample range (and by
tile x sample chunk), reduce, and must reproduce the single-process frame.  The per-rank renderer
here is the oracle (this is a test), the sharding/reduction code is the product's."""
import os
import subprocess
import sys

import numpy as np
import pytest

import conftest
from orion_amd import dist as odist


def test_sample_ranges_partition():
    for total in (0, 1, 7, 8, 2048):
        for world in (1, 2, 3, 8):
            r = [odist.sample_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        odist.sample_range(8, 2, 2)


def test_tile_grid_and_plan_cover_exactly_once():
    W, H, spp = 37, 23, 10
    for world in (1, 2, 4, 8):
        for tiles in (None, (3, 2), (8, 1)):
            cover = np.zeros((H, W), np.int64)
            for rank in range(world):
                for rect, b, e in odist.shard_plan(spp, W, H, rank, world, tiles):
                    x0, y0, x1, y1 = rect if rect else (0, 0, W, H)
                    cover[y0:y1, x0:x1] += e - b
            assert (cover == spp).all(), (world, tiles)


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import conftest
from oracle import bindings as ob
from orion_amd import dist as odist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
sc = ob.OracleScene(conftest.load_model("cornell"))
cam = ob.camera_vectors([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0, 1.5)
W, H, SPP = 48, 32, 6
def render(accum, b, e, tile):
    a = accum.numpy()
    sc.render(cam, W, H, b, e, 2, 8, seed=4, tile=tile, accum=a, threads=2)
for name, tiles in (("samples", None), ("tiles", (2, 2))):
    acc = torch.zeros((H, W, 4), dtype=torch.float32)
    odist.render_sharded(render, SPP, W, H, acc, tiles=tiles)
    if dist.get_rank() == 0:
        np.save(os.path.join({out!r}, name + ".npy"), acc.numpy())
# bench.py's frame step: the rank's cells of the tile x sample-chunk grid, then the sum started asynchronously while the next
# frame renders into a second accumulator
plan = odist.shard_plan(SPP, W, H, dist.get_rank(), 2, (3, 2))
accs = [torch.zeros((H, W, 4), dtype=torch.float32) for _ in range(2)]
works = []
for k in range(2):
    odist.render_plan(render, plan, accs[k])
    works.append(odist.reduce_async(accs[k]))
for w in works:
    w.wait()
assert torch.equal(accs[0], accs[1])
if dist.get_rank() == 0:
    np.save(os.path.join({out!r}, "async_tiles.npy"), accs[0].numpy())
dist.barrier()
dist.destroy_process_group()
"""


def test_two_process_gloo_matches_single_process(tmp_path):
    from oracle import bindings as ob
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=conftest.ROOT, port=port, out=str(tmp_path)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    sc = ob.OracleScene(conftest.load_model("cornell"))
    cam = ob.camera_vectors([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0, 1.5)
    full = sc.render(cam, 48, 32, 0, 6, 2, 8, seed=4, threads=2)
    for name in ("samples", "tiles", "async_tiles"):
        got = np.load(tmp_path / f"{name}.npy")
        assert np.array_equal(got[:, :, 3], full[:, :, 3])
        # same samples, different summation order across ranks
        np.testing.assert_allclose(got[:, :, :3], full[:, :, :3], rtol=2e-6, atol=1e-6)
