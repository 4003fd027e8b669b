"""Host layer (rtc parser, OBJ import, SBVH build) and the C-ABI library's exports.  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest

import conftest
from orion_amd import host

RTC = {"cornell": "cornell-box.rtc", "view_test": "view_test.rtc", "hindu": "hindu.rtc", "nanosuit": "nanosuit.rtc"}


@pytest.fixture(scope="module")
def comp():
    return np.load(os.path.join(conftest.GOLDEN, "components.npz"))


@pytest.mark.parametrize("key", sorted(RTC))
def test_parse_rtc_matches_reference(key, comp):
    r = host.parse_rtc(os.path.join(conftest.ASSETS, RTC[key]))
    ints, floats = comp[f"rtc_{key}_ints"], comp[f"rtc_{key}_floats"]
    assert [r.recursion_level, r.xres, r.yres, len(r.lights)] == list(ints)
    got = np.concatenate([[r.aspect_ratio], r.view_point, r.look_at, r.vector_up, [r.y_view]]).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), floats.view(np.uint32))
    assert np.array_equal(r.lights.view(np.uint32), comp[f"rtc_{key}_lights"].view(np.uint32))
    assert r.obj_file == bytes(comp[f"rtc_{key}_obj"]).decode()


def test_parse_rtc_errors_and_roundtrip(tmp_path, capfd):
    r = host.parse_rtc(str(tmp_path / "missing.rtc"))
    assert "Error opening file" in capfd.readouterr().out
    assert r.xres == 0 and r.obj_file == "" and len(r.lights) == 0
    p = tmp_path / "a.rtc"
    p.write_text("# c\n\n  m.obj \nt.png\n3\n 64 32\n1 2 3\n0 0 0\n0 1 0\n0.5\nX 1 2 3\nl 1 2 3 255 0 51 7\n")
    r = host.parse_rtc(str(p))
    assert "Invalid character 'X'" in capfd.readouterr().out
    assert (r.obj_file, r.recursion_level, r.xres, r.yres, r.aspect_ratio, r.y_view) == ("m.obj", 3, 64, 32, 2.0, 0.5)
    assert r.lights.shape == (1, 7) and r.lights[0, 3] == 1.0 and r.lights[0, 4] == 0.0 and r.lights[0, 6] == 7.0
    q = tmp_path / "b.rtc"
    host.write_rtc(str(q), r)
    r2 = host.parse_rtc(str(q))
    assert r2.obj_file == r.obj_file and r2.xres == 64 and np.allclose(r2.lights, r.lights, atol=1 / 255)
    assert np.array_equal(r2.view_point, r.view_point)


def test_obj_import_shapes():
    m = conftest.load_model("cornell")
    assert [x.n_triangles for x in m.meshes] == [2, 2, 2, 2, 2, 12, 12, 2]
    assert [m.materials[x.material].name for x in m.meshes] == ["floor", "ceiling", "backWall", "rightWall", "leftWall", "shortBox", "tallBox", "light"]
    light = m.materials[m.meshes[-1].material]
    assert np.allclose(light.emissive, [4, 3.5, 3]) and np.allclose(light.diffuse, 0)
    n = m.meshes[0].vertices[:, 3:6]
    assert np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-6) and np.allclose(np.abs(n[:, 1]), 1, atol=1e-6)
    assert conftest.load_model("nanosuit").n_triangles == 19058
    assert conftest.load_model("hindu").n_triangles == 200
    h = conftest.load_model("hindu")
    assert h.materials[0].map_diffuse.endswith("basecolor_.jpg")
    uv = h.meshes[0].vertices[:, 6:8]
    assert uv.min() >= -1e-6 and uv.max() <= 1 + 1e-6
    vt = conftest.load_model("view_test")
    assert [x.n_triangles for x in vt.meshes] == [2, 2] and np.allclose(vt.meshes[0].vertices[:, 3:6], [0, 0, 1])


@pytest.mark.parametrize("name", ["cornell", "view_test", "hindu", "nanosuit", "soup"])
@pytest.mark.parametrize("strategy,tag", [(host.MEDIAN, "median"), (host.MIDDLE, "middle"), (host.SAH, "sah")])
def test_host_bvh_matches_reference(name, strategy, tag):
    g = np.load(os.path.join(conftest.GOLDEN, f"rays_{name}.npz"))
    if name == "soup":
        import raygen
        v, i = raygen.triangle_soup()
        model = host.model_from_arrays(v, i, strategy)
    else:
        model = host.load_model(conftest.scene_path(name), strategy)
    for mi, m in enumerate(model.meshes):
        assert np.array_equal(m.tri_order, g[f"tree_{tag}_{mi}_perm"])
        # walk the index-linked tree in preorder and compare with the reference's dump
        order, stack = [], [0]
        while stack:
            k = stack.pop()
            order.append(k)
            if not m.nodes[k]["is_leaf"]:
                stack.append(int(m.nodes[k]["child"][1])); stack.append(int(m.nodes[k]["child"][0]))
        nd = m.nodes[order]
        meta = np.stack([nd["is_leaf"], nd["begin"], nd["end"]], 1)
        assert np.array_equal(meta, g[f"tree_{tag}_{mi}_meta"])
        boxes = np.concatenate([nd["lo"], nd["hi"]], 1)
        assert np.array_equal(boxes.view(np.uint32), g[f"tree_{tag}_{mi}_boxes"].view(np.uint32))


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/orion_hip.h declares."""
    from orion_amd import api
    header = open(os.path.join(conftest.ROOT, "include", "orion_hip.h")).read()
    declared = set(re.findall(r"\b(orion_hip_\w+)\s*\(", header))
    assert len(declared) >= 10
    L = ctypes.CDLL(api.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} is declared but not exported"
    L.orion_hip_last_error.restype = ctypes.c_char_p
    assert L.orion_hip_last_error() == b""
    # argument validation works without touching a device
    assert L.orion_hip_closest_hit(None, None, None, 1, None) != 0
    assert b"scene is null" in L.orion_hip_last_error()


def test_png_reader_matches_pillow(tmp_path):
    """The C++ host layer's PNG reader against Pillow on every kind of PNG a material map can be:
    grey, grey+alpha, RGB, RGBA, palette (with and without transparency), 1/2/4-bit, 16-bit, every
    row filter (Pillow picks them adaptively on noisy data), odd widths."""
    from PIL import Image
    rng = np.random.default_rng(3)
    w, h = 37, 23
    noise = lambda c: rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    smooth = (np.add.outer(np.arange(h), np.arange(w)) * 3 % 256).astype(np.uint8)
    cases = {
        "L": Image.fromarray(noise(1)[:, :, 0], "L"),
        "L_smooth": Image.fromarray(smooth, "L"),
        "LA": Image.fromarray(noise(2), "LA"),
        "RGB": Image.fromarray(noise(3), "RGB"),
        "RGB_smooth": Image.fromarray(np.stack([smooth, smooth[::-1], smooth.T[:h, :w] if smooth.T.shape == smooth.shape else smooth], 2), "RGB"),
        "RGBA": Image.fromarray(noise(4), "RGBA"),
        "P": Image.fromarray(noise(3), "RGB").quantize(200),
        "P16": Image.fromarray(noise(3), "RGB").quantize(16),     # 4-bit indices
        "P2": Image.fromarray(noise(3), "RGB").quantize(2),       # 1-bit indices
        "1": Image.fromarray(noise(1)[:, :, 0] > 127),            # 1-bit grey
        "I16": Image.fromarray((rng.integers(0, 65536, (h, w))).astype(np.uint16)),
    }
    for name, im in cases.items():
        path = str(tmp_path / f"{name}.png")
        extra = {}
        if name == "P":
            extra["transparency"] = bytes(rng.integers(0, 256, 200, dtype=np.uint8))
        im.save(path, **extra)
        got = host.load_png(path).data
        back = Image.open(path)
        if back.mode == "P":
            want = np.asarray(back.convert("RGBA" if "transparency" in back.info else "RGB"))
        elif back.mode == "1":
            want = np.asarray(back, np.uint8)[:, :, None] * 255
        elif back.mode.startswith("I"):
            want = (np.asarray(back).astype(np.uint32) >> 8).astype(np.uint8)[:, :, None]   # stb keeps the high byte
        else:
            want = np.asarray(back)
            if want.ndim == 2:
                want = want[:, :, None]
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert np.array_equal(got, want), name
    with pytest.raises(RuntimeError):
        host.load_png(str(tmp_path / "missing.png"))
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not a png at all, but long enough to get past the size check........")
    with pytest.raises(RuntimeError, match="not a PNG"):
        host.load_png(str(bad))


def test_image_readers_match_the_references_decoder(tmp_path):
    """PNG and baseline-JPEG readers of the host layer against texels the reference's own decoder
    (its vendored stb_image, run through oracle/_ref when the fixtures were made) returned for the
    same files: identical bytes, JPEG included (same IDCT, upsampling and colour arithmetic)."""
    g = np.load(os.path.join(conftest.GOLDEN, "images_stb.npz"))
    names = sorted(k[:-5] for k in g.files if k.endswith("_file"))
    assert len(names) >= 30 and any(n.startswith("jpg_restart") for n in names) and "jpg_view_test" in names
    for n in names:
        kind = n[:3]
        path = str(tmp_path / f"x.{kind}")
        with open(path, "wb") as f:
            f.write(g[n + "_file"].tobytes())
        got = (host.load_png if kind == "png" else host.load_jpeg)(path).data
        want = g[n + "_texels"]
        assert got.shape == want.shape, (n, got.shape, want.shape)
        assert np.array_equal(got, want), n


def test_nanosuit_colour_maps_decode_to_the_references_texels():
    """The five 1024x1024 RGBA maps and the 128x128 glass map of the nanosuit (the reference's own data
    files) through the host layer's PNG reader: the texels the reference's decoder returned for them
    (fingerprints made by tests/golden/make_nanosuit_map_golden.py through oracle/_ref)."""
    import hashlib
    import json
    want = json.load(open(os.path.join(conftest.GOLDEN, "nanosuit_maps_stb.json")))
    assert len(want) == 6
    for name, w in want.items():
        t = host.load_png(os.path.join(conftest.ASSETS, "nanosuit2", name)).data
        assert t.shape == (w["height"], w["width"], w["channels"]), name
        assert hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest() == w["sha256"], name


def test_jpeg_reader_refuses_what_it_cannot_decode(tmp_path):
    from PIL import Image
    im = Image.fromarray(np.random.default_rng(1).integers(0, 256, (16, 16, 3), dtype=np.uint8), "RGB")
    p = str(tmp_path / "prog.jpg")
    im.save(p, "JPEG", progressive=True)
    with pytest.raises(RuntimeError, match="progressive"):
        host.load_jpeg(p)
    t = host.load_image(p)          # the general loader falls back to Pillow for it
    assert t.data.shape == (16, 16, 3)
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"\xff\xd8\xff\xda\x00\x08\x01\x01\x00\x00\x3f\x00garbage")
    with pytest.raises(RuntimeError):
        host.load_jpeg(str(bad))
