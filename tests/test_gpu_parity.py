"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden vectors.
Integer outputs (mesh, triangle, occlusion) must be identical; t, u, v must be BIT-identical --
stricter than the 1e-5 relative bar -- because the kernels use the reference's own operation order.
"""
import os

import numpy as np
import pytest

import conftest
import raygen

pytestmark = pytest.mark.gpu

SCENES = ["cornell", "view_test", "hindu", "nanosuit", "soup"]
BATCHES = ["primary", "bounce", "shadow", "box", "edge", "grazing"]
MODES = [(0, "auto"), (1, "lane"), (2, "wave"), (4, "wide")]


def set_walk(api, sc, mode):
    """Selects a walk; scenes too small to have a wide tree skip the tests of that walk."""
    try:
        sc.set_option("traversal", mode)
    except api.OrionError as e:
        if mode == 4 and "no wide tree" in str(e):
            pytest.skip(str(e))
        raise


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from orion_amd import api
    api.lib()   # raises if the HIP library is missing -- there is no fallback
    return api


_scenes = {}


def scene_for(api, name):
    if name not in _scenes:
        _scenes[name] = api.Scene(conftest.load_model(name), device=0)
    return _scenes[name]


def oracle_for(name):
    from oracle import bindings as ob
    return ob.OracleScene(conftest.load_model(name))


def hits_to_arrays(h):
    return np.stack([h["mesh"], h["tri"]], 1), np.stack([h["t"], h["u"], h["v"]], 1)


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("mode,tag", MODES)
def test_closest_hit_golden(gpu, name, mode, tag):
    batches = BATCHES   # every walk on every scene: big scenes take the wave-synchronous walk for their first bounce
    g = np.load(os.path.join(conftest.GOLDEN, f"rays_{name}.npz"))
    sc = scene_for(gpu, name)
    set_walk(gpu, sc, mode)
    for b in batches:
        ids, tuv = hits_to_arrays(sc.closest_hit(g[f"{b}_org"], g[f"{b}_dir"]))
        assert np.array_equal(ids, g[f"{b}_ids"]), f"{name}/{b}/{tag}: ids"
        assert np.array_equal(bits(tuv), bits(g[f"{b}_tuv"])), f"{name}/{b}/{tag}: t,u,v bits"
    sc.set_option("traversal", 0)


@pytest.mark.parametrize("name", ["nanosuit", "soup", "hindu"])
def test_rays_along_triangle_planes(gpu, name):
    """The grazing batch (reference-generated): rays along triangle planes, where the reference's triangle test accepts
    rounding noise -- a determinant a few rounding errors above the 1e-6 it rejects, t, u, v that are ratios of residue, "hits"
    up to a dozen box sizes away from the triangle.  Behind them the rays on which a distance-culled walk without a guard
    was seen to differ from the reference's walk.  Every walk must return the reference's bits, the default one included;
    with the guard switched off (option "wide_guard" = 0) the wide walk is known to miss some of them, which is what
    shows that the batch tests something."""
    g = np.load(os.path.join(conftest.GOLDEN, f"rays_{name}.npz"))
    o, d = g["grazing_org"], g["grazing_dir"]
    sc = scene_for(gpu, name)
    for mode in (0, 1, 4):
        sc.set_option("traversal", mode)
        ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
        assert np.array_equal(ids, g["grazing_ids"]), f"{name}/traversal {mode}: ids"
        assert np.array_equal(bits(tuv), bits(g["grazing_tuv"])), f"{name}/traversal {mode}: t,u,v bits"
        occ = sc.any_hit(o, d)
        assert np.array_equal(occ.astype(bool), g["grazing_ids"][:, 0] != gpu.INVALID), f"{name}/traversal {mode}: any hit"
    sc.set_option("traversal", 4); sc.set_option("wide_guard", 0)
    ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
    differ = int(((ids != g["grazing_ids"]).any(1) | (bits(tuv) != bits(g["grazing_tuv"])).any(1)).sum())
    sc.set_option("wide_guard", 1); sc.set_option("traversal", 0)
    if name != "soup":   # (the soup's triangles are too small for an ill-conditioned test to pass the reference's 1e-6)
        assert differ > 0, "the unguarded walk no longer differs on the logged rays: the batch has lost its teeth"
    else:
        assert differ == 0


def test_leaf_root_mesh_on_the_face_of_a_large_scene(gpu):
    """A mesh whose tree is one leaf has no box in the reference's walk (orion/avx/sbvh.cpp:50-59): its triangles are
    tested whatever the ray.  Large scenes test one box around all the roots before the wave-synchronous walk of coherent
    rays; a quad lying on a face of that box must still be found by rays that touch the box in one point of the quad's
    edge only (the slab test's strict tmax > tmin rejects those)."""
    from orion_amd import host
    from oracle import bindings as ob
    soup = conftest.load_model("soup")
    lo, hi = raygen.scene_bounds(soup)
    c = (lo + hi) / 2
    q = np.zeros((4, 8), np.float32)
    q[:, 0] = hi[0]
    q[:, 1] = [lo[1], hi[1], hi[1], lo[1]]
    q[:, 2] = [lo[2], lo[2], hi[2], hi[2]]
    q[:, 3] = 1.0
    quad = host.model_from_arrays(q, np.array([0, 1, 2, 0, 2, 3], np.uint32))
    model = host.merge_models([soup, quad])
    assert model.meshes[1].nodes["is_leaf"][0]
    sc = gpu.Scene(model, device=0)
    orc = ob.OracleScene(model)
    n = 4096
    z = (c[2] + np.linspace(-0.2, 0.2, n) * (hi[2] - lo[2])).astype(np.float32)
    tgt = np.stack([np.full(n, hi[0], np.float32), np.full(n, hi[1], np.float32), z], 1)     # points of the quad's edge y = hi
    org = np.repeat(np.array([[hi[0] + 1.0, hi[1] - 1.0, c[2]]], np.float32), n, 0)          # one origin: a coherent bundle
    d = (tgt - org).astype(np.float32)
    d[d[:, 2] == 0, 2] = 1e-6
    oid, otuv = orc.closest_hit(org, d, threads=4)
    assert (oid[:, 0] == 1).mean() > 0.5      # the reference finds the quad on most of them (the edge's u or v comes out as 0 or just inside)
    for mode in (0, 1, 4):
        sc.set_option("traversal", mode)
        ids, tuv = hits_to_arrays(sc.closest_hit(org, d))
        assert np.array_equal(ids, oid), f"traversal {mode}: ids"
        assert np.array_equal(bits(tuv), bits(otuv)), f"traversal {mode}: t,u,v bits"
        assert np.array_equal(sc.any_hit(org, d).astype(bool), oid[:, 0] != gpu.INVALID), f"traversal {mode}: any hit"
    sc.close()


def test_rays_along_a_leaf_root_mesh_among_many_meshes(gpu):
    """Nine meshes: the one whose tree is a single leaf (a quad) hangs below the wide tree's nodes above the meshes' roots,
    nodes whose boxes the reference does not have.  Rays in the quad's plane: the reference tests the quad's triangles
    for every one of them and accepts rounding residue on a few (orion/geometry.hpp:96-124); the wide walk must return
    the same bits, in every traversal."""
    from orion_amd import host
    from oracle import bindings as ob
    v, i = raygen.triangle_soup(4000, edge=0.03, seed=5)
    q = np.zeros((4, 8), np.float32)
    q[:, 0] = [0.2, 0.8, 0.8, 0.2]; q[:, 1] = 1.3; q[:, 2] = [0.2, 0.2, 0.8, 0.8]; q[:, 4] = 1.0
    parts = [host.model_from_arrays(v, i), host.model_from_arrays(q, np.array([0, 1, 2, 0, 2, 3], np.uint32))]
    for dx, dz in (3.0, 0.0), (0.0, 3.0), (6.0, 0.0), (0.0, 6.0), (6.0, 6.0), (-3.0, 0.0), (0.0, -3.0):
        vv = v.copy(); vv[:, 0] += dx; vv[:, 2] += dz
        parts.append(host.model_from_arrays(vv, i))
    model = host.merge_models(parts)
    assert model.meshes[1].nodes["is_leaf"][0]
    rng = np.random.default_rng(1)
    n = 1 << 19
    p = np.stack([rng.uniform(-9, 12, n), np.full(n, 1.3), rng.uniform(-9, 12, n)], 1)
    ang = rng.random(n) * 2 * np.pi
    slope = 10.0 ** rng.uniform(-7.5, -5, n) * rng.choice([-1, 1], n)
    d = np.stack([np.cos(ang), slope, np.sin(ang)], 1)
    o = (p - d * rng.uniform(0.1, 3, n)[:, None]).astype(np.float32)
    d = (d * (10.0 ** rng.uniform(-1, 1, n))[:, None]).astype(np.float32)
    oid, otuv = ob.OracleScene(model).closest_hit(o, d, threads=8)
    assert (oid[:, 0] == 1).sum() > 20          # the quad is "hit" by some of them
    sc = gpu.Scene(model, device=0)
    for mode in (0, 1, 4):
        sc.set_option("traversal", mode)
        ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
        assert np.array_equal(ids, oid), f"traversal {mode}: ids"
        assert np.array_equal(bits(tuv), bits(otuv)), f"traversal {mode}: t,u,v bits"
    sc.close()


def test_many_meshes_with_leaf_root_quads(gpu):
    """340 meshes, 40 of them quads whose tree is one leaf: rays across the scene and rays in the quads' planes return
    the oracle's bits in every traversal (the wide walk's nodes above the meshes' roots: csrc/wide_walk.h)."""
    import test_wide_tree as tw
    from oracle import bindings as ob
    model = tw.many_meshes_model()
    n = 1 << 17
    o, d = raygen.box_rays(model, n, seed=9)
    o2, d2 = tw.quad_plane_rays(model, n, seed=10)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    oid, otuv = ob.OracleScene(model).closest_hit(o, d, threads=8)
    assert (oid[n:, 0] != gpu.INVALID).sum() > 0
    sc = gpu.Scene(model, device=0)
    for mode in (0, 1, 4):
        sc.set_option("traversal", mode)
        ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
        assert np.array_equal(ids, oid), f"traversal {mode}: ids"
        assert np.array_equal(bits(tuv), bits(otuv)), f"traversal {mode}: t,u,v bits"
        assert np.array_equal(sc.any_hit(o, d).astype(bool), oid[:, 0] != gpu.INVALID), f"traversal {mode}: any hit"
    sc.close()


@pytest.mark.parametrize("kind", ["random", "boxy"])
def test_random_scenes_agree_in_every_walk(gpu, kind):
    """A dozen scenes from the fuzzer (tools/ab/fuzz_scenes.py: many meshes, leaf-root quads, degenerate and duplicated
    triangles; or axis-aligned boxes on a lattice with coincident faces): the oracle, the unculled walk, the default walk
    and the wide walk return the same bits on rays across the scene, along triangle planes and along the axes."""
    import sys
    sys.path.insert(0, os.path.join(conftest.ROOT, "tools", "ab"))
    import fuzz_scenes as fz
    from oracle import bindings as ob
    n = 1 << 14
    for seed in range(12):
        model = fz.boxy_model(seed) if kind == "boxy" else fz.random_model(seed)
        o, d = raygen.box_rays(model, n, seed=seed)
        o2, d2 = raygen.grazing_rays(model, n, seed=seed + 1)
        o3, d3 = fz.lattice_rays(model, n, seed=seed + 5)
        o, d = np.concatenate([o, o2, o3]), np.concatenate([d, d2, d3])
        oid, otuv = ob.OracleScene(model).closest_hit(o, d, threads=8)
        sc = gpu.Scene(model, device=0)
        for mode in (1, 0, 4):
            try:
                sc.set_option("traversal", mode)
            except gpu.OrionError:
                assert mode == 4
                continue
            ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
            assert np.array_equal(ids, oid), f"{kind} scene {seed}, traversal {mode}: ids"
            assert np.array_equal(bits(tuv), bits(otuv)), f"{kind} scene {seed}, traversal {mode}: t,u,v bits"
        sc.close()


@pytest.mark.parametrize("name", SCENES)
def test_any_hit_golden(gpu, name):
    g = np.load(os.path.join(conftest.GOLDEN, f"rays_{name}.npz"))
    sc = scene_for(gpu, name)
    for mode, tag in MODES:
        try:
            sc.set_option("traversal", mode)
        except gpu.OrionError:
            assert mode == 4    # a small scene: no wide tree
            continue
        for b in BATCHES:
            occ = sc.any_hit(g[f"{b}_org"], g[f"{b}_dir"])
            assert np.array_equal(occ.astype(bool), g[f"{b}_ids"][:, 0] != gpu.INVALID), f"{name}/{b}/{tag}"
    sc.set_option("traversal", 0)


@pytest.mark.parametrize("name,n", [("cornell", 1 << 20), ("soup", 1 << 18), ("nanosuit", 1 << 18)])
def test_closest_hit_vs_oracle_large(gpu, name, n):
    model = conftest.load_model(name)
    sc = scene_for(gpu, name)
    orc = oracle_for(name)
    o, d = raygen.box_rays(model, n, seed=21)
    ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
    oid, otuv = orc.closest_hit(o, d, threads=8)
    assert np.array_equal(ids, oid)
    assert np.array_equal(bits(tuv), bits(otuv))
    occ = sc.any_hit(o, d)
    assert np.array_equal(occ.astype(bool), oid[:, 0] != gpu.INVALID)


def test_device_pointers_and_empty_batches(gpu):
    import torch
    model = conftest.load_model("cornell")
    sc = scene_for(gpu, "cornell")
    o, d = raygen.box_rays(model, 5000, seed=3)   # ragged: not a multiple of the block size
    want = sc.closest_hit(o, d)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    got = sc.closest_hit(to, td).cpu().numpy().view(gpu.HIT_DTYPE).reshape(-1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    occ = sc.any_hit(to, td).cpu().numpy()
    assert np.array_equal(occ, sc.any_hit(o, d))
    assert len(sc.closest_hit(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))) == 0
    one = sc.closest_hit(o[:1], d[:1])
    assert np.array_equal(one.view(np.uint32), want[:1].view(np.uint32))


def test_binding_refuses_arrays_the_kernels_would_overrun(gpu):
    """The library takes pointers on trust; the Python binding checks element type, contiguity, shape and
    device before it hands them over."""
    import torch
    sc = scene_for(gpu, "cornell")
    o = torch.zeros((100, 3), device="cuda"); d = torch.ones((100, 3), device="cuda")
    for bad_o, bad_d, out in ((o.double(), d, None), (o, d[::2], None), (o, d.cpu(), None), (o.t(), d, None),
                              (o, d, torch.zeros((50, 5), dtype=torch.int32, device="cuda")),
                              (o, d, torch.zeros((100, 5), dtype=torch.float32, device="cuda"))):
        with pytest.raises(gpu.OrionError, match="expected a C-contiguous"):
            sc.closest_hit(bad_o, bad_d, out)
    with pytest.raises(gpu.OrionError, match="expected a C-contiguous"):
        sc.any_hit(o, d, torch.zeros(99, dtype=torch.uint8, device="cuda"))
    cam = gpu.camera_vectors(*CORNELL_CAM[:3], CORNELL_CAM[3], 1.0)
    for acc in (np.zeros((16, 16, 4), np.float64), np.zeros((16, 16, 3), np.float32), np.zeros((32, 16, 4), np.float32)[::2],
                torch.zeros((8, 16, 4), device="cuda")):
        with pytest.raises(gpu.OrionError, match="render: accum"):
            sc.render(cam, 16, 16, 0, 1, 1, 1, accum=acc)
    assert sc.render(cam, 16, 16, 0, 1, 1, 1, accum=torch.zeros((16, 16, 4), device="cuda"))[:, :, 3].min() == 1


def test_scene_validation(gpu):
    import copy
    model = copy.deepcopy(conftest.load_model("cornell"))
    model.meshes[5].tri_order = model.meshes[5].tri_order.copy()
    model.meshes[5].tri_order[0] = model.meshes[5].tri_order[1]
    with pytest.raises(gpu.OrionError, match="permutation"):
        gpu.Scene(model, device=0)
    model = copy.deepcopy(conftest.load_model("cornell"))
    model.meshes[5].nodes = model.meshes[5].nodes.copy()
    model.meshes[5].nodes["child"][0] = [0, 0]
    with pytest.raises(gpu.OrionError, match="tree"):
        gpu.Scene(model, device=0)
    # a scene with no triangles at all renders black and misses everything
    from orion_amd import host
    empty = host.Model([], [host.Material()])
    sc = gpu.Scene(empty, device=0)
    h = sc.closest_hit(np.zeros((3, 3), np.float32), np.ones((3, 3), np.float32))
    assert (h["mesh"] == gpu.INVALID).all() and np.isinf(h["t"]).all()
    sc.close()


def psnr(a, b):
    peak = float(np.abs(b).max())
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 999.0 if mse == 0 else 10 * np.log10(peak * peak / mse)


CORNELL_CAM = ([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0)


@pytest.mark.parametrize("mode,tag", MODES)
def test_render_brdf_matches_oracle(gpu, mode, tag):
    """Cornell box, BRDF path tracer (no point lights): same counter-RNG streams on both sides, so the
    images agree far inside the 40 dB bar; only libm-level sinf/cosf differences remain."""
    from oracle import bindings as ob
    sc = scene_for(gpu, "cornell")
    set_walk(gpu, sc, mode)
    orc = oracle_for("cornell")
    W, H, spp, L, depth = 96, 64, 8, 2, 12
    eye, look, up, yv = CORNELL_CAM
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    assert np.array_equal(bits(cam), bits(ob.camera_vectors(eye, look, up, yv, W / H)))
    img = sc.render(cam, W, H, 0, spp, L, depth, seed=5)
    ref = orc.render(cam, W, H, 0, spp, L, depth, seed=5, accum_mode=1)
    assert np.array_equal(img[:, :, 3], ref[:, :, 3]) and (img[:, :, 3] == spp).all()
    assert np.isfinite(img).all()
    p = psnr(img[:, :, :3], ref[:, :, :3])
    assert p > 60.0, f"PSNR {p:.1f} dB"
    rel = np.abs(img[:, :, :3] - ref[:, :, :3]) / (np.abs(ref[:, :, :3]) + 1e-3)
    assert np.quantile(rel, 0.99) < 1e-4
    # the reference's own nesting of the radiance sum differs from the iterative order by rounding only
    rec = orc.render(cam, W, H, 0, spp, L, depth, seed=5, accum_mode=0)
    assert psnr(img[:, :, :3], rec[:, :, :3]) > 60.0
    sc.set_option("traversal", 0)


@pytest.mark.parametrize("mode,tag", MODES)
def test_render_point_light_matches_oracle(gpu, mode, tag):
    """configs[0]: Cornell box with one point light, Phong + mirror recursion to depth 8."""
    sc = scene_for(gpu, "cornell")
    set_walk(gpu, sc, mode)
    orc = oracle_for("cornell")
    W = H = 64
    eye, look, up, yv = CORNELL_CAM
    cam = gpu.camera_vectors(eye, look, up, yv, 1.0)
    lights = np.array([[0, 1.5, 1.0, 100 / 255, 100 / 255, 100 / 255, 32]], np.float32)
    img = sc.render(cam, W, H, 0, 4, 1, 8, seed=2, lights=lights)
    ref = orc.render(cam, W, H, 0, 4, 1, 8, seed=2, lights=lights, accum_mode=1)
    assert psnr(img[:, :, :3], ref[:, :, :3]) > 60.0
    sc.set_option("traversal", 0)


def test_render_textured_scene_matches_oracle(gpu):
    """Texture fetch + Phong on a textured two-mesh scene (u8 image as diffuse and specular map)."""
    import copy
    from oracle import bindings as ob
    from orion_amd import host
    model = copy.deepcopy(conftest.load_model("view_test"))
    rng = np.random.default_rng(4)
    # give the quads distinct uvs so that the image texture is visible
    for m in model.meshes:
        m.vertices = m.vertices.copy()
        m.vertices[:, 6:8] = (m.vertices[:, 0:2] * 1.7 - 0.3).astype(np.float32)
    tex = host.Texture(rng.integers(0, 256, (16, 32, 3)).astype(np.uint8))
    tex4 = host.Texture(rng.integers(0, 256, (8, 8, 4)).astype(np.uint8))
    model.materials[model.meshes[0].material].tex_diffuse = tex
    model.materials[model.meshes[0].material].tex_specular = tex
    model.materials[model.meshes[1].material].tex_diffuse = tex4
    sc = gpu.Scene(model, device=0)
    orc = ob.OracleScene(model)
    r = host.parse_rtc(os.path.join(conftest.ASSETS, "view_test.rtc"))
    cam = gpu.camera_vectors(r.view_point, r.look_at, r.vector_up, r.y_view, 1.5)
    img = sc.render(cam, 96, 64, 0, 4, 1, 3, seed=9, lights=r.lights)
    ref = orc.render(cam, 96, 64, 0, 4, 1, 3, seed=9, lights=r.lights, accum_mode=1)
    assert img[:, :, :3].max() > 0
    assert psnr(img[:, :, :3], ref[:, :, :3]) > 60.0
    sc.close()


@pytest.mark.gpu
def test_render_textured_brdf_scene_matches_oracle(gpu):
    """The emissive-light path tracer on a Cornell box with image maps on two of its meshes (the shade kernel's general
    form: colours through the texture table, light samples whose term is surely zero still left out), and the same
    box with a light whose vertex normals differ (no sample is left out, the light's normal is interpolated at the hit)."""
    import copy
    from oracle import bindings as ob
    from orion_amd import host
    eye, look, up, yv = CORNELL_CAM
    W, H = 96, 54
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    rng = np.random.default_rng(12)
    for bend_light in (False, True):
        model = copy.deepcopy(conftest.load_model("cornell"))
        for k in (0, 2):   # floor and back wall
            m = model.meshes[k]
            m.vertices = m.vertices.copy()
            m.vertices[:, 6:8] = (m.vertices[:, 0:2] * 0.9 + 0.2).astype(np.float32)
            model.materials[m.material].tex_diffuse = host.Texture(rng.integers(40, 256, (32, 16, 3)).astype(np.uint8))
        if bend_light:
            light = [m for m in model.meshes if np.any(model.materials[m.material].emissive != 0)][0]
            light.vertices = light.vertices.copy()
            light.vertices[0, 3:6] = np.array([0.3, -0.9, 0.1], np.float32)
        sc = gpu.Scene(model, device=0)
        orc = ob.OracleScene(model)
        img = sc.render(cam, W, H, 0, 6, 2, 5, seed=13)
        ref = orc.render(cam, W, H, 0, 6, 2, 5, seed=13, accum_mode=1)
        assert img[:, :, :3].max() > 0
        assert psnr(img[:, :, :3], ref[:, :, :3]) > 60.0, bend_light
        sc.close()


def test_render_shards_compose_exactly(gpu):
    """Sample ranges and tiles add up to the full frame bit for bit (the multi-GPU decomposition),
    whatever the batch size; device accumulators behave like host ones."""
    import torch
    sc = scene_for(gpu, "cornell")
    W, H = 80, 48
    eye, look, up, yv = CORNELL_CAM
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    full = sc.render(cam, W, H, 0, 6, 2, 10, seed=7)
    part = sc.render(cam, W, H, 0, 2, 2, 10, seed=7)
    part = sc.render(cam, W, H, 2, 6, 2, 10, seed=7, accum=part)
    assert np.array_equal(bits(part), bits(full))
    tiles = np.zeros((H, W, 4), np.float32)
    for t in [(0, 0, 33, 20), (33, 0, W, 20), (0, 20, W, H)]:
        tiles = sc.render(cam, W, H, 0, 6, 2, 10, seed=7, tile=t, accum=tiles)
    assert np.array_equal(bits(tiles), bits(full))
    sc.set_option("batch_paths", 5000)   # forces several batches per call
    small = sc.render(cam, W, H, 0, 6, 2, 10, seed=7)
    sc.set_option("batch_paths", 16 << 20)
    assert np.array_equal(bits(small), bits(full))
    dev = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    sc.render(cam, W, H, 0, 6, 2, 10, seed=7, accum=dev)
    assert np.array_equal(bits(dev.cpu().numpy()), bits(full))


def test_full_size_properties(gpu):
    """At the benchmark's size (1920x1080) the oracle is too slow to compare against, so check what must
    hold regardless: finite radiance, every pixel sampled, determinism, and that the primary-visibility
    mask of the render equals the closest-hit query on the same camera rays."""
    sc = scene_for(gpu, "cornell")
    W, H = 1920, 1080
    eye, look, up, yv = CORNELL_CAM
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    a = sc.render(cam, W, H, 0, 2, 2, 100, seed=1)
    b = sc.render(cam, W, H, 0, 2, 2, 100, seed=1)
    assert np.array_equal(bits(a), bits(b))
    assert np.isfinite(a).all() and (a[:, :, 3] == 2).all() and (a[:, :, :3] >= 0).all()
    # pixels whose unjittered centre ray and both neighbours miss must be black
    c = cam
    j, i = np.meshgrid(np.arange(W), np.arange(H))
    x = (2 * (j.astype(np.float32) / np.float32(W)) - 1 + np.float32(1.0 / W)).astype(np.float32)
    y = (2 * (i.astype(np.float32) / np.float32(H)) - 1 + np.float32(1.0 / H)).astype(np.float32)
    d = (c[3:6][None, None] + x[..., None] * c[6:9][None, None] + (-y)[..., None] * c[9:12][None, None]).astype(np.float32)
    o = np.broadcast_to(c[0:3], d.shape).astype(np.float32)
    hit = (sc.closest_hit(o.reshape(-1, 3), d.reshape(-1, 3))["mesh"] != gpu.INVALID).reshape(H, W)
    near = hit.copy()
    near[1:] |= hit[:-1]; near[:-1] |= hit[1:]; near[:, 1:] |= hit[:, :-1]; near[:, :-1] |= hit[:, 1:]
    assert (a[:, :, :3][~near] == 0).all()
    assert a[:, :, :3][hit].mean() > 0


def test_render_argument_validation(gpu):
    sc = scene_for(gpu, "cornell")
    eye, look, up, yv = CORNELL_CAM
    cam = gpu.camera_vectors(eye, look, up, yv, 1.0)
    with pytest.raises(gpu.OrionError, match="light_samples"):
        sc.render(cam, 8, 8, 0, 1, 0, 2)
    with pytest.raises(gpu.OrionError, match="spp_end"):
        sc.render(cam, 8, 8, 2, 1, 1, 2)
    with pytest.raises(gpu.OrionError, match="tile"):
        sc.render(cam, 8, 8, 0, 1, 1, 2, tile=(0, 0, 9, 8))
    # any number of point lights, as in the reference (orion/raytracer.cpp:196 loops over a vector)
    many = sc.render(cam, 8, 8, 0, 1, 1, 2, lights=np.tile(np.float32([[0, 1.5, 1.0, 0.5, 0.5, 0.5, 0.25]]), (40, 1)))
    one = sc.render(cam, 8, 8, 0, 1, 1, 2, lights=np.float32([[0, 1.5, 1.0, 0.5, 0.5, 0.5, 10.0]]))
    assert one[:, :, :3].max() > 0 and np.allclose(many, one, rtol=1e-4, atol=1e-6)
    with pytest.raises(gpu.OrionError, match="empty image"):
        sc.render(cam, 0, 8, 0, 1, 1, 2, accum=np.zeros((8, 0, 4), np.float32))
    # zero samples is a no-op, not an error
    acc = sc.render(cam, 8, 8, 3, 3, 1, 2)
    assert (acc == 0).all()
    with pytest.raises(gpu.OrionError, match="unknown option"):
        sc.set_option("no_such_option", 1)


def test_render_without_any_light(gpu):
    """No point lights and no emissive mesh: the BRDF branch still bounces, everything stays black
    (orion/raytracer.cpp:131-194 with an empty emissiveSurfaces())."""
    from oracle import bindings as ob
    model = conftest.load_model("view_test")
    sc = gpu.Scene(model, device=0)
    cam = gpu.camera_vectors([0, 0, 2], [0, 0, 0], [0, 1, 0], 1.0, 1.0)
    img = sc.render(cam, 32, 32, 0, 2, 1, 4, seed=1)
    ref = ob.OracleScene(model).render(cam, 32, 32, 0, 2, 1, 4, seed=1)
    assert (img[:, :, :3] == 0).all() and (ref[:, :, :3] == 0).all() and (img[:, :, 3] == 2).all()
    sc.close()


def test_many_emissive_meshes_and_odd_light_samples(gpu):
    """Two emissive meshes, three light samples each: the per-mesh, per-sample accumulation order of
    orion/raytracer.cpp:133-159."""
    import copy
    from oracle import bindings as ob
    model = copy.deepcopy(conftest.load_model("cornell"))
    model.materials[model.meshes[5].material].emissive = np.array([0.5, 0.2, 0.1], np.float32)   # the short box glows too
    sc = gpu.Scene(model, device=0)
    eye, look, up, yv = CORNELL_CAM
    cam = gpu.camera_vectors(eye, look, up, yv, 1.0)
    img = sc.render(cam, 48, 48, 0, 4, 3, 6, seed=3)
    ref = ob.OracleScene(model).render(cam, 48, 48, 0, 4, 3, 6, seed=3, accum_mode=1)
    assert psnr(img[:, :, :3], ref[:, :, :3]) > 60.0
    sc.close()


@pytest.mark.parametrize("name", ["nanosuit", "soup"])
def test_ray_queries_pick_their_walk_per_wave(gpu, name):
    """On large scenes a wave of rays that leave one point in nearly one direction takes the
    wave-synchronous walk, any other wave the per-lane one; a batch that mixes both kinds (and
    waves that straddle them) must match the oracle bit for bit, any-hit answers included."""
    model = conftest.load_model(name)
    sc = scene_for(gpu, name)
    orc = oracle_for(name)
    n = 1 << 17
    o, d = raygen.box_rays(model, n, seed=91)
    rng = np.random.default_rng(92)
    for start in range(0, n, 4096):            # every other 2048-ray stretch: a fan from one point, ragged ends
        a, b = start + 37, start + 2048 + 11
        o[a:b] = o[a]
        d[a:b] = d[a] + (rng.random((b - a, 3)).astype(np.float32) - 0.5) * 0.15 * np.abs(d[a]).max()
    sc.set_option("traversal", 0)
    ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
    oid, otuv = orc.closest_hit(o, d, threads=8)
    assert np.array_equal(ids, oid)
    assert np.array_equal(bits(tuv), bits(otuv))
    assert np.array_equal(sc.any_hit(o, d).astype(bool), oid[:, 0] != gpu.INVALID)


@pytest.mark.parametrize("name", ["nanosuit", "soup", "hindu"])
def test_wide_walk_returns_the_unculled_walks_bits(gpu, name):
    """The ordered, distance-culled walk of the 4-wide tree against the unculled per-lane walk on the
    GPU, 2^22 rays per kind: rays between surface points (they graze the surfaces they leave and
    reach, which is where a culling margin would show), rays across the scene's box, a batch with
    zero / denormal / huge direction components and far origins (those lanes take the unculled walk
    inside the same launch), and the any-hit answers."""
    import torch
    model = conftest.load_model(name)
    sc = scene_for(gpu, name)
    rng = np.random.default_rng(17)
    n = 1 << 22
    tris = np.concatenate([m.vertices[m.indices.reshape(-1, 3), :3] for m in model.meshes if m.n_triangles])
    pick = rng.integers(0, len(tris), n)
    a = rng.random(n).astype(np.float32); b = rng.random(n).astype(np.float32)
    flip = a + b > 1; a[flip] = 1 - a[flip]; b[flip] = 1 - b[flip]
    tp = tris[pick]
    p = tp[:, 0] + a[:, None] * (tp[:, 1] - tp[:, 0]) + b[:, None] * (tp[:, 2] - tp[:, 0])
    nr = np.cross(tp[:, 1] - tp[:, 0], tp[:, 2] - tp[:, 0]); ln = np.linalg.norm(nr, axis=1, keepdims=True)
    o1 = (p + 1e-3 * nr / np.where(ln > 0, ln, 1)).astype(np.float32)
    o1, d1 = raygen.rays_to_surface(o1, model, seed=18)
    o2, d2 = raygen.box_rays(model, n, seed=19)
    o3, d3 = o2.copy(), d2.copy()
    d3[::5, 0] = 0.0; d3[1::7, 1] = np.float32(1e-42); d3[2::11, 2] = np.float32(3e38); d3[3::13] *= np.float32(1e-30)
    o3[4::17] *= np.float32(1e4); o3[5::19, 1] = np.float32(np.inf); d3[6::23, 0] = np.float32(np.nan)
    for kind, (o, d) in {"surface": (o1, d1), "box": (o2, d2), "untame": (o3, d3)}.items():
        to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        out = {}
        for mode in (1, 4, 0):
            sc.set_option("traversal", mode)
            out[mode] = (sc.closest_hit(to, td).clone(), sc.any_hit(to, td).clone())
        sc.set_option("traversal", 0)
        for mode in (4, 0):
            assert torch.equal(out[1][0], out[mode][0]), f"{name}/{kind}: closest hit, traversal {mode}"
            assert torch.equal(out[1][1], out[mode][1]), f"{name}/{kind}: any hit, traversal {mode}"
        if kind != "untame":
            hit = out[1][0][:, 0] != -1
            assert float(hit.float().mean()) > 0.2 and torch.equal(out[1][1].bool(), hit)


def test_the_three_walks_agree(gpu):
    """Per-lane, wave-synchronous (hand-scheduled assembly) and one-ray-per-wave traversal: same rays
    in, the same bits out -- ray queries, including waves pushed onto the exact-select slab test by
    rays with zero direction components, and a rendered image."""
    model = conftest.load_model("cornell")
    sc = gpu.Scene(model, device=0)
    rng = np.random.default_rng(77)
    n = 1 << 18
    o = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32) + np.float32([0, 1, 0])
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[::97, 0] = 0.0            # rays that push their wave onto the exact-select slab test
    d[5::131, 1] = 0.0
    o[::1013] = np.float32([0.0, 1.0, 8.0])
    eye, look, up, yv = CORNELL_CAM
    cam = gpu.camera_vectors(eye, look, up, yv, 1.5)
    out = {}
    for mode in (1, 2, 3):
        sc.set_option("traversal", mode)
        h = sc.closest_hit(o, d)
        img = sc.render(cam, 96, 64, 0, 6, 2, 20, seed=3)
        out[mode] = (h.copy(), img.copy())
    sc.close()
    a = out[1]
    assert (a[0]["mesh"] != gpu.INVALID).sum() > n // 2
    for mode in (2, 3):
        b = out[mode]
        assert np.array_equal(a[0]["mesh"], b[0]["mesh"]) and np.array_equal(a[0]["tri"], b[0]["tri"]), mode
        for f in ("t", "u", "v"):
            assert np.array_equal(bits(a[0][f]), bits(b[0][f])), (mode, f)
        assert np.array_equal(bits(a[1]), bits(b[1])), mode


@pytest.mark.parametrize("name", ["cornell", "view_test"])
def test_one_ray_per_wave_golden(gpu, name):
    """The walk the tail kernel uses (all 64 lanes on one ray: one box or triangle each, the serial
    walk's acceptance order restated as a reduction) against the same golden vectors, edge cases and
    NaN rays included, and against the oracle on a large batch."""
    g = np.load(os.path.join(conftest.GOLDEN, f"rays_{name}.npz"))
    sc = scene_for(gpu, name)
    sc.set_option("traversal", 3)
    for b in BATCHES:
        ids, tuv = hits_to_arrays(sc.closest_hit(g[f"{b}_org"], g[f"{b}_dir"]))
        assert np.array_equal(ids, g[f"{b}_ids"]), f"{name}/{b}: ids"
        assert np.array_equal(bits(tuv), bits(g[f"{b}_tuv"])), f"{name}/{b}: t,u,v bits"
        occ = sc.any_hit(g[f"{b}_org"], g[f"{b}_dir"])
        assert np.array_equal(occ.astype(bool), g[f"{b}_ids"][:, 0] != gpu.INVALID), f"{name}/{b}: any hit"
    model = conftest.load_model(name)
    o, d = raygen.box_rays(model, 1 << 18, seed=22)
    ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
    oid, otuv = oracle_for(name).closest_hit(o, d, threads=8)
    assert np.array_equal(ids, oid) and np.array_equal(bits(tuv), bits(otuv))
    sc.set_option("traversal", 0)


def test_one_ray_per_wave_needs_a_small_scene(gpu):
    sc = scene_for(gpu, "nanosuit")
    with pytest.raises(RuntimeError, match="too large"):
        sc.set_option("traversal", 3)


@pytest.mark.parametrize("point", [False, True])
def test_tail_kernel_changes_nothing(gpu, point):
    """Handing the last paths of a batch to the one-path-per-wave kernel is a scheduling decision:
    whatever the threshold, the image keeps its bits."""
    sc = gpu.Scene(conftest.load_model("cornell"), device=0)
    eye, look, up, yv = CORNELL_CAM
    W, H = 160, 96
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    lights = np.float32([[0.0, 1.6, 0.5, 1.0, 0.9, 0.8, 4.0]]) if point else None
    imgs = []
    for tail in (0, 1 << 9, 1 << 13, 1 << 24):   # never, late, early, from the third bounce on
        sc.set_option("tail_paths", tail)
        sc.profile(True)
        imgs.append(sc.render(cam, W, H, 0, 8, 2, 30, seed=9, lights=lights).copy())
        st = sc.profile_read()
        sc.profile(False)
        assert (st["tail"]["launches"] > 0) == (tail > 0), (tail, st["tail"])
    sc.close()
    assert np.isfinite(imgs[0]).all() and imgs[0][:, :, :3].max() > 0
    for im in imgs[1:]:
        assert np.array_equal(bits(im), bits(imgs[0]))


def _tie_scene(n_meshes, tris_per_mesh, seed):
    """Meshes of random triangles in which every third triangle is an exact copy of an earlier one
    (of the same mesh or of a previous mesh) and quads share their diagonal: rays at them produce
    exact ties in t, which the reference resolves by visit order."""
    from orion_amd import host
    rng = np.random.default_rng(seed)
    pool, models = [], []
    for _ in range(n_meshes):
        tri = []
        for k in range(tris_per_mesh):
            if pool and k % 3 == 2:
                tri.append(pool[rng.integers(len(pool))].copy())
            elif k % 3 == 1 and tri:
                a, b, c = tri[-1]
                tri.append(np.stack([a, c, a + (c - b)]))     # shares the edge a-c with the previous one
            else:
                ctr = rng.uniform(-1, 1, 3)
                tri.append((ctr + rng.normal(scale=0.6, size=(3, 3))).astype(np.float32))
        tri = np.asarray(tri, np.float32)
        pool.extend(tri)
        v = np.zeros((len(tri) * 3, 8), np.float32)
        v[:, :3] = tri.reshape(-1, 3)
        v[:, 3:6] = [0, 1, 0]
        models.append(host.model_from_arrays(v, np.arange(len(v), dtype=np.uint32)))
    return host.merge_models(models), np.asarray(pool, np.float32)


@pytest.mark.parametrize("n_meshes,per_mesh,modes", [(4, 9, (1, 2, 3)), (6, 90, (1, 2, 4))])
def test_exact_ties_resolve_like_the_reference(gpu, n_meshes, per_mesh, modes):
    from oracle import bindings as ob
    model, pool = _tie_scene(n_meshes, per_mesh, seed=31 + per_mesh)
    rng = np.random.default_rng(5)
    n = 1 << 16
    # aim at points inside (and on the edges and vertices of) the triangles
    t = pool[rng.integers(len(pool), size=n)]
    w = rng.dirichlet([1, 1, 1], size=n).astype(np.float32)
    w[::7] = [0.5, 0.5, 0.0]; w[::11] = [1.0, 0.0, 0.0]; w[::13] = [0.0, 0.5, 0.5]
    target = (t * w[:, :, None]).sum(1).astype(np.float32)
    o = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = (target - o).astype(np.float32)
    oid, otuv = ob.OracleScene(model).closest_hit(o, d, threads=8)
    assert (oid[:, 0] != gpu.INVALID).mean() > 0.9
    sc = gpu.Scene(model, device=0)
    for mode in modes:
        sc.set_option("traversal", mode)
        ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
        assert np.array_equal(ids, oid), f"mode {mode}: ids"
        assert np.array_equal(bits(tuv), bits(otuv)), f"mode {mode}: t,u,v"
    sc.close()


def test_quick_math_claims_hold_for_every_float(gpu):
    """The short reciprocal and square-root forms are used where they return the bits of the
    division / sqrtf.  That claim is about 2^32 arguments and is checked on all of them, with the
    kernels' own device functions (tools/micro/quick_math_claims.hip, built by build())."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orion_amd", "quick_math_claims")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "recip_quick mismatches: 0  sqrt_quick mismatches: 0  guard errors: 0" in r.stdout


def _edge_bound(model):
    """The direction bound the scene upload derives: 2^122 / E^2 (E = largest edge component), at most 2^126."""
    E = 0.0
    for m in model.meshes:
        v = m.vertices[m.indices.reshape(-1, 3), 0:3].astype(np.float32)
        E = max(E, float(np.abs(v[:, 1] - v[:, 0]).max()), float(np.abs(v[:, 2] - v[:, 0]).max()))
    return min(2.0 ** 122 / (E * E), 2.0 ** 126)


@pytest.mark.parametrize("name,modes", [("cornell", (1, 2, 3)), ("soup", (1, 4))])
def test_quick_reciprocals_return_the_divisions_bits(gpu, name, modes):
    """1/det and 1/dir are computed as reciprocal estimate + one Newton step wherever that returns
    the bits of the IEEE division (normal divisors below 2^126), and by division elsewhere; which
    one a wave takes depends on the magnitude of its rays' directions.  Scaled rays on both sides
    of every switch-over must keep matching the oracle (which divides) bit for bit."""
    model = conftest.load_model(name)
    sc = scene_for(gpu, name)
    orc = oracle_for(name)
    n = 1 << 15
    o, d0 = raygen.box_rays(model, n, seed=33)
    dmax = float(np.abs(d0).max())
    bound = _edge_bound(model)
    scales = [2.0 ** -20, 1.0, 2.0 ** 60, 0.5 * bound / dmax, 0.97 * bound / dmax, 1.03 * bound / dmax, 2.0 ** 125 / dmax, 2.0 ** 127 / dmax]
    for k, sfac in enumerate(scales):
        d = (d0.astype(np.float64) * sfac).astype(np.float32)
        if k % 2:                       # some waves with components outside the normal range
            d[::257, 0] = 0.0
            d[3::509, 1] = np.float32(1e-41)
            d[7::1021, 2] = np.float32(2.0 ** -126)
        assert np.isfinite(d).all()
        oid, otuv = orc.closest_hit(o, d, threads=8)
        if sfac in (1.0, 2.0 ** 60):    # (tiny directions fall under the |det| < 1e-6 rejection)
            assert (oid[:, 0] != gpu.INVALID).mean() > 0.3
        for mode in modes:
            sc.set_option("traversal", mode)
            ids, tuv = hits_to_arrays(sc.closest_hit(o, d))
            assert np.array_equal(ids, oid), f"{name} mode {mode} scale {sfac:g}: ids"
            assert np.array_equal(bits(tuv), bits(otuv)), f"{name} mode {mode} scale {sfac:g}: t,u,v"
            occ = sc.any_hit(o, d)
            assert np.array_equal(occ.astype(bool), oid[:, 0] != gpu.INVALID), f"{name} mode {mode} scale {sfac:g}: any hit"
    sc.set_option("traversal", 0)


def test_a_failed_render_leaves_the_handle_usable(gpu, monkeypatch):
    """A render call that fails half way (the test hook makes the n-th bounce launch fail; an allocation for the
    next batch failing ends the same way) has launches of one or both lanes in flight.  The call must wait for
    them, report, and the next call on the same handle must give the bits of a fresh one.  The hook can only be
    armed in a process that asks for it, and for one call."""
    sc = gpu.Scene(conftest.load_model("cornell"), device=0)
    monkeypatch.delenv("ORION_HIP_TEST_HOOKS", raising=False)
    with pytest.raises(gpu.OrionError, match="unknown option"):
        sc.set_option("fail_at_bounce", 1)
    monkeypatch.setenv("ORION_HIP_TEST_HOOKS", "1")
    eye, look, up, yv = CORNELL_CAM
    W, H, spp = 256, 144, 12
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    ref = sc.render(cam, W, H, 0, spp, 2, 30, seed=21).copy()
    sc.set_option("batch_paths", 3 * W * H)              # four batches, alternating between the two lanes
    sc.set_option("overlap_below", 1 << 30)
    sc.set_option("tail_paths", 0)                       # every bounce is a launch of its own: more places to fail
    for n in (1, 2, 3, 5, 9, 14, 23):                    # first launch, the other lane's head, mid-batch, later batches
        sc.set_option("fail_at_bounce", n)
        with pytest.raises(gpu.OrionError, match="injected failure"):
            sc.render(cam, W, H, 0, spp, 2, 30, seed=21)
        img = sc.render(cam, W, H, 0, spp, 2, 30, seed=21)
        assert np.array_equal(bits(ref), bits(img)), n
    sc.close()


def test_a_render_call_survives_queue_allocations_that_fail(gpu, monkeypatch):
    """The queues of a render call are sized from the memory the device has free when the call starts; if taking them fails all
    the same (somebody else allocated in between), the call halves its samples per batch and tries again.  The test hook
    makes the first attempts fail; the image must keep its bits (it does not depend on how a call is cut into batches)."""
    sc = gpu.Scene(conftest.load_model("cornell"), device=0)
    eye, look, up, yv = CORNELL_CAM
    W, H, spp = 256, 144, 16
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    ref = sc.render(cam, W, H, 0, spp, 2, 30, seed=22).copy()
    monkeypatch.setenv("ORION_HIP_TEST_HOOKS", "1")
    for n in (1, 2, 3):
        sc.set_option("fail_queue_allocs", n)
        img = sc.render(cam, W, H, 0, spp, 2, 30, seed=22)
        assert np.array_equal(bits(ref), bits(img)), n
    sc.set_option("fail_queue_allocs", 64)          # more failures than halvings: the call reports the failure ...
    with pytest.raises(gpu.OrionError, match="hipMalloc"):
        sc.render(cam, W, H, 0, spp, 2, 30, seed=22)
    img = sc.render(cam, W, H, 0, spp, 2, 30, seed=22)   # ... and the handle stays usable
    assert np.array_equal(bits(ref), bits(img))
    sc.close()


def test_batching_and_repetition_change_nothing(gpu):
    """How a render call is cut into batches (option batch_paths), and running it twice, must not
    move a bit: samples are independent and are summed per pixel in ascending sample order."""
    sc = gpu.Scene(conftest.load_model("cornell"), device=0)
    eye, look, up, yv = CORNELL_CAM
    W, H, spp = 128, 72, 12
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    ref = sc.render(cam, W, H, 0, spp, 2, 40, seed=11).copy()
    again = sc.render(cam, W, H, 0, spp, 2, 40, seed=11)
    assert np.array_equal(bits(ref), bits(again))
    for batch in (W * H, 3 * W * H, 5 * W * H + 17):     # 12, 4 and 3 (uneven) batches
        sc.set_option("batch_paths", batch)
        img = sc.render(cam, W, H, 0, spp, 2, 40, seed=11)
        assert np.array_equal(bits(ref), bits(img)), batch
    # and in two calls that continue one accumulator
    sc.set_option("batch_paths", 64 << 20)
    acc = sc.render(cam, W, H, 0, 5, 2, 40, seed=11)
    acc = sc.render(cam, W, H, 5, spp, 2, 40, seed=11, accum=acc)
    assert np.array_equal(bits(ref), bits(acc))
    sc.close()


@pytest.mark.gpu
def test_queue_partitions_change_nothing(gpu):
    """A batch's queues come in partitions with counters of their own (option queue_parts); a path's arithmetic does
    not depend on where its records lie, so the image must keep its bits for any number of partitions: frames of
    many tiles per partition, of fewer first-bounce blocks than partitions (some partitions stay empty), several
    batches on two lanes, the one-path-per-wave tail on and off, BRDF and point lights."""
    sc = gpu.Scene(conftest.load_model("cornell"), device=0)
    eye, look, up, yv = CORNELL_CAM
    lights = np.array([[0, 1.5, 1.0, 100 / 255, 100 / 255, 100 / 255, 32]], np.float32)
    for W, H, spp in ((640, 360, 16), (24, 16, 5), (7, 3, 2)):
        cam = gpu.camera_vectors(eye, look, up, yv, W / H)
        sc.set_option("queue_parts", 1); sc.set_option("batch_paths", 256 << 20); sc.set_option("tail_paths", 65536)
        ref = sc.render(cam, W, H, 0, spp, 2, 60, seed=7).copy()
        pref = sc.render(cam, W, H, 0, spp, 1, 6, seed=7, lights=lights).copy()
        assert ref[:, :, :3].max() > 0
        for parts in (2, 16):
            sc.set_option("queue_parts", parts)
            for batch, tail in ((256 << 20, 65536), (max(1, spp // 3) * W * H, 0)):
                sc.set_option("batch_paths", batch); sc.set_option("tail_paths", tail)
                assert np.array_equal(bits(ref), bits(sc.render(cam, W, H, 0, spp, 2, 60, seed=7))), (W, parts, batch)
                assert np.array_equal(bits(pref), bits(sc.render(cam, W, H, 0, spp, 1, 6, seed=7, lights=lights))), (W, parts, batch)
    with pytest.raises(gpu.OrionError):
        sc.set_option("queue_parts", 3)
    sc.close()


@pytest.mark.gpu
def test_two_lanes_change_nothing(gpu):
    """Consecutive batches alternate between two streams and overlap (option overlap_below: the next
    batch starts once the current one has that few rays left).  Whether and when they overlap must
    not move a bit, for a host accumulator and for a device one."""
    import torch
    sc = gpu.Scene(conftest.load_model("cornell"), device=0)
    eye, look, up, yv = CORNELL_CAM
    W, H, spp = 640, 360, 20
    cam = gpu.camera_vectors(eye, look, up, yv, W / H)
    sc.set_option("overlap_below", 0)                      # one lane, one batch
    ref = sc.render(cam, W, H, 0, spp, 2, 60, seed=5).copy()
    sc.set_option("batch_paths", 4 * W * H)                # five batches of 0.9 Mi paths
    for below in (0, 1, 100_000, 600_000, 1 << 30):
        sc.set_option("overlap_below", below)
        img = sc.render(cam, W, H, 0, spp, 2, 60, seed=5)
        assert np.array_equal(bits(ref), bits(img)), below
        dev = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        sc.render(cam, W, H, 0, spp, 2, 60, seed=5, accum=dev)
        torch.cuda.synchronize()
        assert np.array_equal(bits(ref), bits(dev.cpu().numpy())), below
    # uneven last batch, and point lights (any-hit shadow rays) through the same loop
    sc.set_option("batch_paths", 3 * W * H)
    img = sc.render(cam, W, H, 0, spp, 2, 60, seed=5)
    assert np.array_equal(bits(ref), bits(img))
    lights = np.array([[0, 1.5, 1.0, 100 / 255, 100 / 255, 100 / 255, 32]], np.float32)
    sc.set_option("overlap_below", 0); sc.set_option("batch_paths", 64 << 20)
    pref = sc.render(cam, W, H, 0, 8, 1, 6, seed=5, lights=lights).copy()
    sc.set_option("overlap_below", 1 << 30); sc.set_option("batch_paths", 2 * W * H)
    pimg = sc.render(cam, W, H, 0, 8, 1, 6, seed=5, lights=lights)
    assert np.array_equal(bits(pref), bits(pimg))
    sc.close()


def test_a_small_queue_budget_changes_nothing(gpu):
    """A render call sizes its batches to the memory it may take (option queue_budget_mib; by default three
    quarters of the device's free memory): with room for a few samples per batch only, the image keeps its bits."""
    import torch
    sc = scene_for(gpu, "cornell")
    cam = gpu.camera_vectors([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0, 16 / 9)
    W, H = 480, 270
    ref = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    sc.render(cam, W, H, 0, 48, 2, 12, seed=3, accum=ref)
    for mib in (64, 40):      # 144 B per path and lane: 64 MiB hold one batch of 3 samples per lane, 40 MiB one of 2
        sc.set_option("queue_budget_mib", mib)
        acc = torch.zeros_like(ref)
        sc.render(cam, W, H, 0, 48, 2, 12, seed=3, accum=acc)
        assert torch.equal(acc, ref), mib
    sc.set_option("queue_budget_mib", 0)
