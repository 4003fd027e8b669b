"""The 4-wide tree of the ordered, distance-culled walk (orion_amd/csrc/wide_bvh.h) and a CPU model of
that walk (oracle/wide_model.cpp) against the oracle's restatement of the reference's unordered,
unculled walk (orion/avx/sbvh.cpp:36-83, orion/model.hpp:52-62).  No GPU: the tree builder is host code."""
import os

import numpy as np
import pytest

import conftest
import raygen
from oracle import bindings as ob

BATCHES = ["primary", "bounce", "shadow", "box", "edge", "grazing"]
SLACK = 2.0 ** -7    # kWideSlack
PLAIN, GUARDED = 1, 2  # the model's culls: without the guard (option wide_guard = 0) and as the product runs them


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["cornell", "view_test", "hindu", "nanosuit", "soup"])
def test_wide_model_matches_the_golden_vectors(name):
    """Every golden batch (reference-generated ids and t, u, v bits): the rays the wide walk takes must
    come out bit for bit; the others (zero, denormal or huge direction components, far origins) are the
    ones it hands to the unculled walk."""
    model = conftest.load_model(name)
    wm = ob.WideModel(model)
    assert wm.h, wm.why_not
    g = np.load(os.path.join(conftest.GOLDEN, f"rays_{name}.npz"))
    for b in BATCHES:
        o, d = g[f"{b}_org"], g[f"{b}_dir"]
        ids, tuv, st = wm.closest_hit(o, d, cull=GUARDED, threads=4, kappa=SLACK)
        taken = ids[:, 0] != ob.WideModel.NOT_TAKEN
        if b == "edge":
            assert taken.mean() < 0.9
        elif b == "grazing":
            assert taken.mean() > 0.5     # (some start beyond the origin bound, a few scene sizes back)
        else:
            assert taken.all(), f"{name}/{b}"
        # (hindu/edge holds rays laid exactly into the plane of a flat grid of quads: the reference's own t for the
        # triangles it "hits" edge-on is rounding noise, tens of box sizes off -- the walk's guard is what covers those)
        assert np.array_equal(ids[taken], g[f"{b}_ids"][taken]), f"{name}/{b}: ids"
        assert np.array_equal(bits(tuv[taken]), bits(g[f"{b}_tuv"][taken])), f"{name}/{b}: t,u,v bits"


@pytest.mark.parametrize("name", ["nanosuit", "soup"])
def test_wide_model_against_the_oracle_on_surface_rays(name):
    """Rays from surface points to surface points graze what they leave and what they reach: 2^20 of them,
    and the work the walk saves against the reference's."""
    model = conftest.load_model(name)
    wm = ob.WideModel(model); sc = ob.OracleScene(model)
    rng = np.random.default_rng(7)
    n = 1 << 20
    tris = np.concatenate([m.vertices[m.indices.reshape(-1, 3), :3] for m in model.meshes if m.n_triangles])
    pick = rng.integers(0, len(tris), n); a = rng.random(n); b = rng.random(n)
    flip = a + b > 1; a[flip] = 1 - a[flip]; b[flip] = 1 - b[flip]
    tp = tris[pick]
    p = tp[:, 0] + a[:, None] * (tp[:, 1] - tp[:, 0]) + b[:, None] * (tp[:, 2] - tp[:, 0])
    nr = np.cross(tp[:, 1] - tp[:, 0], tp[:, 2] - tp[:, 0]); ln = np.linalg.norm(nr, axis=1, keepdims=True)
    o = (p + 1e-3 * nr / np.where(ln > 0, ln, 1)).astype(np.float32)
    o, d = raygen.rays_to_surface(o, model, seed=8)
    oid, otuv, counters = sc.closest_hit(o, d, threads=8, count=True)
    ids, tuv, st = wm.closest_hit(o, d, cull=GUARDED, threads=8, kappa=SLACK)
    assert np.array_equal(ids, oid) and np.array_equal(bits(tuv), bits(otuv))
    assert st["max_stack"] <= wm.stack_need
    # the work the walk saves: with the guard a scene of large, bumpy triangles (nanosuit) keeps about half of the reference's
    # tests on these rays, a scene of small ones (the soup: the guard never fires) an eighth; without the guard a quarter at most
    assert st["tris"] < 0.75 * counters[1] and st["nodes"] < 0.5 * counters[0]
    ids, tuv, st = wm.closest_hit(o, d, cull=PLAIN, threads=8, kappa=SLACK)
    assert np.array_equal(ids, oid) and np.array_equal(bits(tuv), bits(otuv))
    assert st["tris"] < 0.25 * counters[1] and st["nodes"] < 0.25 * counters[0]


def test_wide_tree_structure():
    """nanosuit: the tree stands for every triangle exactly once, leaves stay small, and the builder's bound on
    the walk's stack holds."""
    model = conftest.load_model("nanosuit")
    wm = ob.WideModel(model)
    assert wm.h and wm.n_nodes > 1000 and 0 < wm.stack_need <= 32
    assert wm.pad == pytest.approx(2.0 ** -12 * max(abs(float(x)) for m in model.meshes for x in (m.vertices[:, :3].min(), m.vertices[:, :3].max())), rel=1e-6)
    # numbering: the virtual nodes above the meshes' roots first, children behind their parents, nothing below a mesh's
    # own node points back up into the virtual ones
    words, _ = wm.nodes()
    refs = words[:, [3, 7, 14, 15]]
    inner = (refs != 0x7FFFFFFF) & (refs < 0x80000000)
    assert len(model.meshes) > 4 and 0 < wm.n_virtual < wm.n_nodes
    assert (refs[inner] < wm.n_nodes).all()
    assert (refs > np.arange(wm.n_nodes, dtype=np.uint32)[:, None])[inner].all()
    assert (refs[wm.n_virtual:][inner[wm.n_virtual:]] >= wm.n_virtual).all()
    reached = np.zeros(wm.n_nodes, bool); reached[0] = True; reached[refs[inner]] = True
    assert reached.all() and inner.sum() == wm.n_nodes - 1


def many_meshes_model():
    """300 small soups and 40 quads (meshes whose tree is one leaf) scattered over a 10 x 10 x 10 box."""
    from orion_amd import host
    rng = np.random.default_rng(3)
    v, i = raygen.triangle_soup(64, edge=0.2, seed=5)
    parts = []
    for _ in range(300):
        vv = v.copy(); vv[:, :3] = vv[:, :3] * 0.3 + rng.uniform(0, 10, 3)
        parts.append(host.model_from_arrays(vv.astype(np.float32), i))
    for _ in range(40):
        q = np.zeros((4, 8), np.float32); c = rng.uniform(0, 10, 3)
        q[:, 0] = c[0] + np.array([0, .3, .3, 0]); q[:, 1] = c[1]; q[:, 2] = c[2] + np.array([0, 0, .3, .3]); q[:, 4] = 1
        parts.append(host.model_from_arrays(q, np.array([0, 1, 2, 0, 2, 3], np.uint32)))
    return host.merge_models(parts)


def quad_plane_rays(model, n, seed):
    """Rays in the planes of the model's leaf-root quads (y = const), at slopes of 10^-7.5 .. 10^-5."""
    rng = np.random.default_rng(seed)
    ys = np.array([m.vertices[0, 1] for m in model.meshes if m.n_triangles == 2], np.float64)
    lo, hi = raygen.scene_bounds(model)
    p = np.stack([rng.uniform(lo[0], hi[0], n), ys[rng.integers(0, len(ys), n)], rng.uniform(lo[2], hi[2], n)], 1)
    ang = rng.random(n) * 2 * np.pi
    d = np.stack([np.cos(ang), 10.0 ** rng.uniform(-7.5, -5, n) * rng.choice([-1, 1], n), np.sin(ang)], 1)
    o = (p - d * rng.uniform(0.1, 3, n)[:, None]).astype(np.float32)
    return o, (d * (10.0 ** rng.uniform(-1, 1, n))[:, None]).astype(np.float32)


def test_many_meshes_cost_no_more_with_the_guard():
    """The nodes above the meshes' roots are entered for the rays that meet them (or that a leaf-root mesh forces in),
    not for every ray: with 340 meshes the guarded walk stays near the unguarded one, and both return the oracle's bits."""
    model = many_meshes_model()
    wm = ob.WideModel(model); sc = ob.OracleScene(model)
    assert wm.h and wm.n_virtual > 100
    n = 1 << 15
    o, d = raygen.box_rays(model, n, seed=9)
    o2, d2 = quad_plane_rays(model, n, seed=10)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    oid, otuv = sc.closest_hit(o, d, threads=8)
    assert (oid[n:, 0] != 0xFFFFFFFF).sum() > 0
    per_ray = {}
    for mode in (PLAIN, GUARDED):
        ids, tuv, st = wm.closest_hit(o, d, cull=mode, threads=8, kappa=SLACK)
        assert (ids[:, 0] != ob.WideModel.NOT_TAKEN).all()
        assert np.array_equal(ids, oid) and np.array_equal(bits(tuv), bits(otuv)), mode
        # (the cost on the rays across the box; the rays in the quads' planes have all forty quads forced, rightly)
        per_ray[mode] = wm.closest_hit(o[:n], d[:n], cull=mode, threads=8, kappa=SLACK)[2]["nodes"] / n
    assert per_ray[GUARDED] < 1.25 * per_ray[PLAIN], per_ray


def test_scenes_the_wide_tree_refuses():
    """Trees whose child boxes are not inside their parents' cannot be collapsed (the visit rule would not
    survive): such scenes keep the unculled walk."""
    import copy
    model = copy.deepcopy(conftest.load_model("hindu"))
    nodes = model.meshes[0].nodes.copy()
    inner = np.where(nodes["is_leaf"] == 0)[0]
    nodes["hi"][inner[3]] -= 0.9 * (nodes["hi"][inner[3]] - nodes["lo"][inner[3]])    # a box that no longer holds its children
    model.meshes[0].nodes = nodes
    wm = ob.WideModel(model)
    assert not wm.h and "inside" in wm.why_not
