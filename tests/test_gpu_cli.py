"""The C++ host mirror of RayTracer::traceRTC (orion_rt) end to end: .rtc in, image file out,
same pixels as the Python binding renders through the same C ABI."""
import os
import subprocess

import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu

CLI = os.path.join(conftest.ROOT, "orion_amd", "orion_rt")


def read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = (int(x) for x in f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)


def test_cli_renders_derived_rtc(tmp_path):
    from orion_amd import api, host
    assert os.path.exists(CLI), "orion_rt is built by __graft_entry__.build()"
    # derived scene file (configs[0] shape): Cornell box, point light, 64x48, depth 4
    r = host.parse_rtc(os.path.join(conftest.ASSETS, "cornell-box.rtc"))
    r.obj_file = os.path.join(conftest.ASSETS, "cornell-box", "cornell-box.obj")
    r.xres, r.yres, r.aspect_ratio, r.recursion_level = 64, 48, 64 / 48, 4
    r.lights = np.array([[0, 1.5, 1.0, 100 / 255, 100 / 255, 100 / 255, 32]], np.float32)
    rtc_path = str(tmp_path / "cornell_point.rtc")
    host.write_rtc(rtc_path, r)
    # obj path is absolute; the reference joins it to the rtc directory, absolute paths survive the join here too
    r2 = host.parse_rtc(rtc_path)
    assert r2.xres == 64 and len(r2.lights) == 1
    out = str(tmp_path / "out.ppm")
    env = dict(os.environ)
    # write_rtc stores the obj path as given; make it relative to the rtc directory as the CLI expects
    r.obj_file = os.path.relpath(r.obj_file, str(tmp_path))
    host.write_rtc(rtc_path, r)
    p = subprocess.run([CLI, rtc_path, "-o", out, "-p", "4", "-l", "1", "--seed", "3"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Triangles:                  36" in p.stdout and "Number of emissive meshes: 1" in p.stdout
    img = read_ppm(out)
    fr = api.load_frame(rtc_path, load_textures=False)
    sc = api.Scene(fr.model, device=0)
    acc = sc.render(fr.camera12, 64, 48, 0, 4, 1, 4, seed=3, lights=fr.rtc.lights)
    sc.close()
    want = (np.minimum(1.0, acc[:, :, :3] * np.float32(0.25)) * 255).astype(np.uint8)
    assert img.shape == want.shape
    assert (np.abs(img.astype(int) - want.astype(int)) <= 1).all() and (img == want).mean() > 0.99
    # png and hdr writers produce files; an unknown extension falls back to .hdr with a message
    for name in ("a.png", "a.hdr", "a.xyz"):
        q = subprocess.run([CLI, rtc_path, "-o", str(tmp_path / name), "-p", "1"], capture_output=True, text=True, timeout=120)
        assert q.returncode == 0 and os.path.getsize(tmp_path / name) > 100
    assert "Extension not recognized" in q.stdout
    from PIL import Image
    png = np.asarray(Image.open(tmp_path / "a.png"))
    assert png.shape == (48, 64, 4) and png[:, :, 3].min() == 255 and png[:, :, :3].max() == 255


def test_cli_decodes_png_textures(tmp_path):
    """view_test scene with its map as an RGBA PNG: the C++ tool decodes it itself and renders the
    same pixels as the Python binding with the Pillow-decoded image."""
    from PIL import Image
    from orion_amd import api, host
    rng = np.random.default_rng(4)
    tex = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    Image.fromarray(tex, "RGBA").save(tmp_path / "map.png")
    obj = open(os.path.join(conftest.ASSETS, "view_test.obj")).read()
    mtl = open(os.path.join(conftest.ASSETS, "view_test.mtl")).read()
    import re
    mtl = re.sub(r"(?m)^\s*map_\S+.*$", "", mtl)
    mtl, n = re.subn(r"(?m)^(\s*Kd .*)$", r"\1\n\tmap_Kd map.png", mtl)
    assert n >= 1
    (tmp_path / "view_test.obj").write_text(obj)
    (tmp_path / "view_test.mtl").write_text(mtl)
    r = host.parse_rtc(os.path.join(conftest.ASSETS, "view_test.rtc"))
    r.obj_file = "view_test.obj"
    r.xres, r.yres, r.aspect_ratio, r.recursion_level = 80, 60, 80 / 60, 1
    rtc_path = str(tmp_path / "vt.rtc")
    host.write_rtc(rtc_path, r)
    out = str(tmp_path / "vt.ppm")
    p = subprocess.run([CLI, rtc_path, "-o", out, "-p", "4", "-l", "1", "--seed", "2"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "solid colour" not in p.stderr, p.stderr
    img = read_ppm(out)
    fr = api.load_frame(rtc_path, load_textures=True)
    assert any(m.tex_diffuse is not None and m.tex_diffuse.data.shape == (64, 64, 4) for m in fr.model.materials)
    sc = api.Scene(fr.model, device=0)
    acc = sc.render(fr.camera12, 80, 60, 0, 4, 1, 1, seed=2, lights=fr.rtc.lights)
    sc.close()
    want = (np.minimum(1.0, acc[:, :, :3] * np.float32(0.25)) * 255).astype(np.uint8)
    assert (np.abs(img.astype(int) - want.astype(int)) <= 1).all() and (img == want).mean() > 0.99
    assert img.std() > 5   # the noise texture is visible


def test_cli_reports_failure(tmp_path):
    """A render the device cannot do (a device that does not exist) ends with a non-zero exit status and without
    an image file -- a zeroed image would pass for a result.  Any number of lights renders, as in the reference."""
    from orion_amd import host
    r = host.parse_rtc(os.path.join(conftest.ASSETS, "cornell-box.rtc"))
    r.obj_file = os.path.relpath(os.path.join(conftest.ASSETS, "cornell-box", "cornell-box.obj"), str(tmp_path))
    r.xres, r.yres, r.aspect_ratio, r.recursion_level = 32, 24, 32 / 24, 1
    r.lights = np.array([[0, 1.5, 1.0, 0.4, 0.4, 0.4, 32]], np.float32)
    rtc_path = str(tmp_path / "ok.rtc")
    host.write_rtc(rtc_path, r)
    out = tmp_path / "no_device.png"
    p = subprocess.run([CLI, rtc_path, "-o", str(out), "-p", "1", "--device", "99"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "ERROR::ORION_HIP::" in p.stdout and not out.exists(), p.stdout
    p = subprocess.run([CLI, rtc_path, "-o", str(tmp_path / "ok.ppm"), "-p", "2", "--seed", "5"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and (tmp_path / "ok.ppm").exists()
    one = read_ppm(str(tmp_path / "ok.ppm")).astype(np.float64)
    # 70 copies of the light at a seventieth of its intensity: the same picture up to rounding
    r.lights = np.tile(r.lights, (70, 1)); r.lights[:, 6] /= 70
    many = str(tmp_path / "many.rtc")
    host.write_rtc(many, r)
    assert len(host.parse_rtc(many).lights) == 70
    p = subprocess.run([CLI, many, "-o", str(tmp_path / "many.ppm"), "-p", "2", "--seed", "5"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout
    assert np.abs(read_ppm(str(tmp_path / "many.ppm")).astype(np.float64) - one).max() <= 2


def test_cli_png_is_the_references_normalisation_of_the_oracles_render(tmp_path):
    """configs[1] at reduced size through the tool, PNG out.  The file must hold what the reference's savePNG
    (orion/raytracer.cpp:255-280) makes of the radiance image -- every channel divided by the image maximum,
    clamped to 1, times 255, truncated -- where the radiance image is the ORACLE's render of the same frame,
    not this library's own."""
    from PIL import Image
    from oracle import bindings as ob
    from orion_amd import api, host
    r = host.parse_rtc(os.path.join(conftest.ROOT, "configs", "c1_cornell_brdf_fhd.rtc"))
    r.obj_file = os.path.relpath(os.path.join(conftest.ASSETS, "cornell-box", "cornell-box.obj"), str(tmp_path))
    W, H, SPP = 160, 90, 16
    r.xres, r.yres, r.aspect_ratio, r.recursion_level = W, H, W / H, 12
    rtc_path = str(tmp_path / "c1_small.rtc")
    host.write_rtc(rtc_path, r)
    out = str(tmp_path / "c1.png")
    p = subprocess.run([CLI, rtc_path, "-o", out, "-p", str(SPP), "-l", "2", "--seed", "7"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    png = np.asarray(Image.open(out))
    assert png.shape == (H, W, 4) and (png[:, :, 3] == 255).all()
    fr = api.load_frame(rtc_path, load_textures=False)
    acc = ob.OracleScene(fr.model).render(fr.camera12, W, H, 0, SPP, 2, 12, seed=7, accum_mode=1)
    img = acc[:, :, :3] * np.float32(1.0 / SPP)
    inv_scale = np.float32(1.0) / img.max()
    want = np.floor(np.minimum(np.float32(1.0), img * inv_scale) * 255).astype(np.int32)
    diff = np.abs(png[:, :, :3].astype(np.int32) - want)
    # sinf / cosf / powf differ in the last bits between the device and glibc: a channel may land on the other
    # side of a truncation, never further
    assert diff.max() <= 1 and (diff == 0).mean() > 0.98, (int(diff.max()), float((diff == 0).mean()))
    assert png[:, :, :3].max() == 255
