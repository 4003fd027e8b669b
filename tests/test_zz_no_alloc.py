"""The entry point that allocates nothing (orion_hip_closest_hit_noalloc) against orion_hip_closest_hit,
through the C ABI: two calls in a row on one scene, the second with a smaller batch (the scratch it uses
lives inside the library and keeps the first call's contents), on a large scene (wide walk: work words and
chunk list in that scratch) and on a small one (no scratch needed)."""
import numpy as np
import pytest

import conftest
import raygen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["nanosuit", "cornell"])
def test_no_alloc_entry_point_matches_closest_hit(name):
    import torch
    assert torch.cuda.is_available()
    from orion_amd import api
    model = conftest.load_model(name)
    sc = api.Scene(model, device=0)
    try:
        for n, seed in ((1 << 20) + 77, 41), (50_001, 42):          # the second call is smaller, both ragged
            o, d = raygen.box_rays(model, n, seed=seed)
            o[: n // 3] = o[0]                                        # a coherent stretch: camera-like chunks and mixed ones
            d[: n // 3] = d[0] + (np.random.default_rng(seed).random((n // 3, 3)).astype(np.float32) - 0.5) * 0.1 * np.abs(d[0]).max()
            to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
            want = sc.closest_hit(to, td)
            got = torch.full((n, 5), 0x55555555, dtype=torch.int32, device="cuda")
            sc.closest_hit_noalloc(to, td, got)
            assert torch.equal(got, want), f"{name}: n = {n}"
            assert float((want[:, 0] != -1).float().mean()) > 0.05
        # limits: host arrays and oversized batches are refused with a message
        with pytest.raises(api.OrionError, match="device memory"):
            api._check(api.lib().orion_hip_closest_hit_noalloc(sc.h, o.ctypes.data, d.ctypes.data, 16, got.data_ptr()))
        with pytest.raises(api.OrionError, match="at most"):
            api._check(api.lib().orion_hip_closest_hit_noalloc(sc.h, to.data_ptr(), td.data_ptr(), (1 << 24) + 1, got.data_ptr()))
    finally:
        sc.close()


def test_no_alloc_entry_point_allocates_nothing():
    """The device's free memory does not move across calls, a scene's first-ever query on the wide walk included
    (orion_hip_closest_hit would allocate its work buffer there)."""
    import torch
    from orion_amd import api
    model = conftest.load_model("nanosuit")
    sc = api.Scene(model, device=0)
    try:
        n = 1 << 18
        o, d = raygen.box_rays(model, n, seed=43)
        to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        got = torch.zeros((n, 5), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for k in range(4):
            sc.closest_hit_noalloc(to[: n >> k], td[: n >> k], got[: n >> k])
            assert torch.cuda.mem_get_info()[0] == free0, f"call {k}: the device's free memory moved"
        want = sc.closest_hit(to[: n >> 3], td[: n >> 3])   # (this one may allocate)
        assert torch.equal(got[: n >> 3], want)
    finally:
        sc.close()


def test_overlapping_no_alloc_calls_are_refused_not_raced():
    """Two scenes on one GPU share that GPU's scratch: calls from two threads either return the right answer or fail
    with the "in use" error, never a wrong result."""
    import threading
    import torch
    from orion_amd import api
    model = conftest.load_model("nanosuit")
    scenes = [api.Scene(model, device=0) for _ in range(2)]
    try:
        n = 1 << 18
        o, d = raygen.box_rays(model, n, seed=44)
        to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        want = scenes[0].closest_hit(to, td).clone()
        outcome = [[0, 0, 0], [0, 0, 0]]   # per thread: right, refused, wrong

        def work(k):
            got = torch.zeros((n, 5), dtype=torch.int32, device="cuda")
            for _ in range(24):
                try:
                    scenes[k].closest_hit_noalloc(to, td, got)
                except api.OrionError as e:
                    outcome[k][1 if "in use" in str(e) else 2] += 1
                    continue
                outcome[k][0 if torch.equal(got, want) else 2] += 1

        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        assert outcome[0][2] == 0 and outcome[1][2] == 0, outcome
        assert outcome[0][0] + outcome[1][0] > 0, outcome
    finally:
        for sc in scenes:
            sc.close()
