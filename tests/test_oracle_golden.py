"""The CPU restatement (oracle/) against the golden vectors the real reference produced
(tests/golden/*.npz, written by tests/golden/make_golden.py from oracle/_ref).  CPU only."""
import os

import numpy as np
import pytest

import conftest
from oracle import bindings as ob

SCENES = ["cornell", "view_test", "hindu", "nanosuit", "soup"]
BATCHES = ["primary", "bounce", "shadow", "box", "edge", "grazing"]


def golden(name):
    return np.load(os.path.join(conftest.GOLDEN, f"rays_{name}.npz"))


@pytest.fixture(scope="module")
def comp():
    return np.load(os.path.join(conftest.GOLDEN, "components.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("strategy,tag", [(0, "median"), (1, "middle"), (2, "sah")])
def test_bvh_build_matches_reference(name, strategy, tag):
    g = golden(name)
    model = conftest.load_model(name)
    sc = ob.OracleScene(model, strategy=strategy)
    for mi in range(len(model.meshes)):
        boxes, meta, perm = sc.tree(mi)
        assert np.array_equal(perm, g[f"tree_{tag}_{mi}_perm"]), f"triangle permutation of mesh {mi}"
        assert np.array_equal(meta, g[f"tree_{tag}_{mi}_meta"]), f"tree shape of mesh {mi}"
        assert np.array_equal(bits(boxes), bits(g[f"tree_{tag}_{mi}_boxes"])), f"node boxes of mesh {mi}"
    if strategy == 0:
        areas = np.array([sc.area(mi) for mi in range(len(model.meshes))], np.float32)
        assert np.array_equal(bits(areas), bits(g["areas"]))


@pytest.mark.parametrize("name", SCENES)
def test_closest_hit_bit_exact(name):
    g = golden(name)
    sc = ob.OracleScene(conftest.load_model(name))
    for b in BATCHES:
        with np.errstate(all="ignore"):
            ids, tuv = sc.closest_hit(g[f"{b}_org"], g[f"{b}_dir"])
        assert not g[f"{b}_ill"].any()
        assert np.array_equal(ids, g[f"{b}_ids"]), f"{name}/{b}: mesh/triangle ids"
        assert np.array_equal(bits(tuv), bits(g[f"{b}_tuv"])), f"{name}/{b}: t,u,v bits"


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("strategy,tag", [(1, "middle"), (2, "sah")])
def test_closest_hit_other_strategies(name, strategy, tag):
    g = golden(name)
    sc = ob.OracleScene(conftest.load_model(name), strategy=strategy)
    ids, tuv = sc.closest_hit(g[f"{tag}_org"], g[f"{tag}_dir"])
    assert np.array_equal(ids, g[f"{tag}_ids"])
    assert np.array_equal(bits(tuv), bits(g[f"{tag}_tuv"]))


def test_any_hit_is_closest_hit_hit(models):
    g = golden("cornell")
    sc = ob.OracleScene(models("cornell"))
    for b in BATCHES:
        occ = sc.any_hit(g[f"{b}_org"], g[f"{b}_dir"])
        assert np.array_equal(occ.astype(bool), g[f"{b}_ids"][:, 0] != ob.INVALID)


def test_xoroshiro_streams(comp):
    for seed in (0, 42, 0x1234567890ABCDEF):
        for jumps in (0, 1, 3):
            assert np.array_equal(ob.xoroshiro_u64(seed, 64, jumps), comp[f"xoro_u64_{seed}_{jumps}"])
            assert np.array_equal(bits(ob.xoroshiro_floats(seed, 64, jumps)), bits(comp[f"xoro_f32_{seed}_{jumps}"]))


def test_light_surface_sampling(comp, models):
    model = models("cornell")
    sc = ob.OracleScene(model)
    light = len(model.meshes) - 1
    for key, mesh, seed in (("light_points", light, 42), ("box_points", 5, 7)):
        got = sc.random_points(mesh, seed, 256)
        # same triangle picks and bias exactly; the point itself may differ by FMA contraction in the reference build
        assert np.array_equal(bits(got[:, 3]), bits(comp[key][:, 3]))
        np.testing.assert_allclose(got[:, :3], comp[key][:, :3], rtol=0, atol=2e-7)


def test_normalize(comp):
    # the reference's rsqrtps is a hardware approximation (relative error <= 1.5 * 2^-12)
    exact = ob.normalize(comp["normalize_in"], rsqrt_mode=0)
    np.testing.assert_allclose(exact, comp["normalize_out"], rtol=4e-4, atol=1e-7)


def test_material_models(comp, models):
    model = models("cornell")
    sc = ob.OracleScene(model)
    light = len(model.meshes) - 1
    rows = comp["shade_rows"]
    for mi in (0, 4, 5, light):
        for fn, key in ((sc.color_brdf, "brdf"), (sc.color_phong, "phong")):
            want = comp[f"{key}_{mi}"]
            got = fn(mi, rows, rsqrt_mode=0)
            ok = np.isfinite(want).all(1)
            scale = np.abs(want[ok]).max() if ok.any() else 1.0
            # exact 1/sqrt vs rsqrtps: up to 3 normalisations, a product of two cosines and pow(x,10)
            np.testing.assert_allclose(got[ok], want[ok], rtol=2e-2, atol=1e-3 * scale)


@pytest.mark.skipif(not hasattr(np, "float32"), reason="")
def test_material_models_with_hardware_rsqrt(comp, models):
    """With the reference's rsqrt emulated (same CPU family) the two agree to rounding."""
    import platform
    if platform.machine() not in ("x86_64", "AMD64"):
        pytest.skip("rsqrtss emulation is x86 only")
    model = models("cornell")
    sc = ob.OracleScene(model)
    a = ob.normalize(comp["normalize_in"], rsqrt_mode=1)
    if not np.array_equal(bits(a), bits(comp["normalize_out"])):
        pytest.skip("this CPU's rsqrtss differs from the one the fixtures were made on")
    rows = comp["shade_rows"]
    for mi in (0, 5):
        want = comp[f"brdf_{mi}"]
        got = sc.color_brdf(mi, rows, rsqrt_mode=1)
        ok = np.isfinite(want).all(1)
        np.testing.assert_allclose(got[ok], want[ok], rtol=2e-5, atol=1e-6)
        want = comp[f"phong_{mi}"]
        got = sc.color_phong(mi, rows, rsqrt_mode=1)
        ok = np.isfinite(want).all(1)
        np.testing.assert_allclose(got[ok], want[ok], rtol=2e-4, atol=1e-5)


def test_reflect_and_attributes(comp, models):
    np.testing.assert_allclose(ob.reflect(comp["reflect_in"]), comp["reflect_out"], rtol=0, atol=5e-7)
    sc = ob.OracleScene(models("cornell"))
    got = sc.attributes(5, comp["attr_tri"], comp["attr_u"], comp["attr_v"])
    np.testing.assert_allclose(got, comp["attr_out"], rtol=0, atol=2e-7)


def test_texture_fetch(comp, models):
    from orion_amd import host
    model = models("cornell")
    import copy
    m2 = copy.deepcopy(model)
    m2.materials[m2.meshes[0].material].tex_diffuse = host.Texture(comp["tex_rgb_u8"])
    sc = ob.OracleScene(m2)
    got = sc.material_fetch(0, 0, comp["tex_uv"])
    assert np.array_equal(bits(got), bits(comp["tex_out"]))
    sc0 = ob.OracleScene(model)
    assert np.array_equal(bits(sc0.material_fetch(0, 0, comp["tex_uv"])), bits(comp["solid_out"]))


def test_camera_vectors(comp):
    for key in ("cornell", "view_test", "hindu", "nanosuit"):
        f = comp[f"rtc_{key}_floats"]
        got = ob.camera_vectors(f[1:4], f[4:7], f[7:10], float(f[10]), float(f[0]))
        np.testing.assert_allclose(got, comp[f"cam_{key}"], rtol=4e-4, atol=1e-6)


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    assert list(ob.philox(0, 0, 0, 0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    k = 0xFFFFFFFF | (0xFFFFFFFF << 32)
    assert list(ob.philox(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, k)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    k = 0xA4093822 | (0x299F31D0 << 32)
    assert list(ob.philox(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, k)) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    assert 0.0 < ob.u01(0) < ob.u01(0xFFFFFFFF) < 1.0


def test_render_modes_agree(models):
    """Recursive (reference nesting) and iterative (GPU order) accumulation differ by rounding only,
    and the xoroshiro and counter generators give statistically equal images."""
    model = models("cornell")
    sc = ob.OracleScene(model)
    cam = ob.camera_vectors([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0, 1.0)
    a = sc.render(cam, 24, 24, 0, 4, 2, 6, seed=3, accum_mode=0)
    b = sc.render(cam, 24, 24, 0, 4, 2, 6, seed=3, accum_mode=1)
    assert np.array_equal(a[:, :, 3], b[:, :, 3]) and (a[:, :, 3] == 4).all()
    np.testing.assert_allclose(a[:, :, :3], b[:, :, :3], rtol=1e-5, atol=1e-6)
    c = sc.render(cam, 24, 24, 0, 64, 2, 6, seed=3, rng_kind=0, accum_mode=0)
    d = sc.render(cam, 24, 24, 0, 64, 2, 6, seed=3, rng_kind=1, accum_mode=0)
    assert abs(c[:, :, :3].mean() - d[:, :, :3].mean()) < 0.05 * d[:, :, :3].mean()
    # shards add: samples [0,2) + [2,4) == [0,4), and tiles compose
    e = sc.render(cam, 24, 24, 0, 2, 2, 6, seed=3)
    e = sc.render(cam, 24, 24, 2, 4, 2, 6, seed=3, accum=e)
    assert np.array_equal(bits(e), bits(b))
    f = sc.render(cam, 24, 24, 0, 4, 2, 6, seed=3, tile=(0, 0, 24, 10))
    f = sc.render(cam, 24, 24, 0, 4, 2, 6, seed=3, tile=(0, 10, 24, 24), accum=f)
    assert np.array_equal(bits(f), bits(b))


def test_bounce_loop_against_the_references_published_render(models):
    """The reference's README render (1920x1080, 2048 spp, depth 100, 2 light samples), box-filtered
    8x8 to 240x135 (tests/golden/cornell_readme_ds8.npy), against the oracle rendering 240x135 directly:
    the per-sample jitter spans the whole (larger) pixel, so the expectation is the same box filter.
    The PNG's normalisation constant (its brightest full-resolution pixel) is fitted."""
    want = np.load(os.path.join(conftest.GOLDEN, "cornell_readme_ds8.npy")).astype(np.float64)
    from orion_amd import host
    r = host.parse_rtc(os.path.join(conftest.ASSETS, "cornell-box.rtc"))
    W, H, SPP = 240, 135, 384
    cam = ob.camera_vectors(r.view_point, r.look_at, r.vector_up, r.y_view, W / H)
    sc = ob.OracleScene(models("cornell"))
    for kind in (1, 0):   # counter RNG (threads) and the reference's xoroshiro stream (serial, fewer samples)
        spp = SPP if kind == 1 else 48
        acc = sc.render(cam, W, H, 0, spp, 2, 100, seed=7, rng_kind=kind, accum_mode=0, threads=8)
        got = acc[:, :, :3].astype(np.float64) / spp
        dark = want < 0.98   # saturated pixels of the PNG carry no information
        a = float((got[dark] * want[dark]).sum() / (got[dark] ** 2).sum())
        err = np.minimum(1.0, a * got)[dark] - want[dark]
        p = 10 * np.log10(1.0 / np.mean(err ** 2))
        assert p > (38.0 if kind == 1 else 33.0), f"rng {kind}: PSNR {p:.1f} dB"
        assert abs(err.mean()) < 3e-3
