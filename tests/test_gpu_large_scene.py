"""A scene fifty times the largest asset of the reference: a soup of 2^20 triangles in one mesh (half a million
BVH nodes, 218 k nodes of the 4-wide tree, a walk stack the builder bounds at 28 entries where the kernels
hold 16).  The ray queries must return what the oracle's restatement of the reference's walk
(orion/avx/sbvh.cpp:36-83, orion/model.hpp:52-62) returns, ids and bits, in every walk that applies."""
import numpy as np
import pytest

import raygen

pytestmark = pytest.mark.gpu

N_TRI = 1 << 20
N_RAYS = 1 << 16


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def big():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from orion_amd import api, host
    from oracle import bindings as ob
    api.lib()
    v, i = raygen.triangle_soup(N_TRI, edge=0.02 * (19058 / N_TRI) ** (1 / 3), seed=2)
    model = host.model_from_arrays(v, i)
    sc = api.Scene(model, device=0)
    yield api, model, sc, ob.OracleScene(model)
    sc.close()


def batches(model):
    o1, d1 = raygen.box_rays(model, N_RAYS, seed=3)                       # across the scene's box
    p, _ = raygen.box_rays(model, N_RAYS, seed=4)
    o2, d2 = raygen.rays_to_surface(p, model, seed=5)                     # towards points on triangles
    rng = np.random.default_rng(6)
    tris = model.meshes[0].vertices[model.meshes[0].indices.reshape(-1, 3), :3]
    pick = rng.integers(0, len(tris), N_RAYS)
    o3 = (tris[pick].mean(1) + 1e-3 * rng.standard_normal((N_RAYS, 3))).astype(np.float32)   # from just off a triangle
    d3 = rng.standard_normal((N_RAYS, 3)).astype(np.float32)
    return {"box": (o1, d1), "to surface": (o2, d2), "from surface": (o3, d3)}


def test_million_triangle_scene_has_a_wide_tree_and_matches_the_oracle(big):
    api, model, sc, oracle = big
    assert model.n_triangles == N_TRI
    for bname, (o, d) in batches(model).items():
        oid, otuv = oracle.closest_hit(o, d, threads=8)
        assert (oid[:, 0] != 0xFFFFFFFF).mean() > 0.3, bname
        for mode in (0, 4, 1):       # auto (coherent chunks + wide walk), the wide walk alone, the unculled walk
            sc.set_option("traversal", mode)
            h = sc.closest_hit(o, d)
            ids = np.stack([h["mesh"], h["tri"]], 1)
            tuv = np.stack([h["t"], h["u"], h["v"]], 1)
            assert np.array_equal(ids, oid), f"{bname} traversal {mode}: ids"
            assert np.array_equal(bits(tuv), bits(otuv)), f"{bname} traversal {mode}: t,u,v bits"
            occ = sc.any_hit(o, d)
            assert np.array_equal(np.asarray(occ).astype(bool), oid[:, 0] != 0xFFFFFFFF), f"{bname} traversal {mode}: any-hit"
    sc.set_option("traversal", 0)
