"""Writes the golden vectors under tests/golden/ by running the REAL reference classes
(oracle/_ref/liborion_ref.so, built by `make -C oracle ref` from /root/reference).

Development container only: the GPU box has neither /root/reference nor a need to run
this.  The fixtures are data (inputs + the reference's outputs); rerun after changing a
ray generator in tests/raygen.py:

    python tests/golden/make_golden.py [--rays-only]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import conftest  # noqa: E402
import raygen  # noqa: E402
from oracle import bindings as ob  # noqa: E402
from orion_amd import host  # noqa: E402

N_RAYS = 4096
RTC = {"cornell": "cornell-box.rtc", "view_test": "view_test.rtc", "hindu": "hindu.rtc", "nanosuit": "nanosuit.rtc"}


def scene_camera(name, model):
    if name in RTC:
        r = host.parse_rtc(os.path.join(conftest.ASSETS, RTC[name]))
        return ob.camera_vectors(r.view_point, r.look_at, r.vector_up, r.y_view, r.aspect_ratio), r.xres, r.yres
    eye, look, up, yv = raygen.default_camera(model)
    return ob.camera_vectors(eye, look, up, yv, 1.0), 512, 512


def ray_batches(name, model, ref):
    cam, W, H = scene_camera(name, model)
    o1, d1 = raygen.primary_rays(cam, W, H, N_RAYS, seed=1)
    ids1, tuv1 = ref.closest_hit(o1, d1)
    # geometric normals of the hit triangles for the bounce batch
    nrm = np.zeros((len(o1), 3), np.float32)
    for mi, m in enumerate(model.meshes):
        sel = np.where(ids1[:, 0] == mi)[0]
        if len(sel):
            nrm[sel] = ref.attributes(mi, ids1[sel, 1], tuv1[sel, 1], tuv1[sel, 2])[:, :3]
    o2, d2 = raygen.bounce_rays(o1, d1, ids1, tuv1, nrm, seed=2)
    if len(o2) == 0:
        o2, d2 = raygen.box_rays(model, N_RAYS, seed=2)
    o3, d3 = raygen.rays_to_surface(o2, model, seed=3)
    o4, d4 = raygen.box_rays(model, N_RAYS, seed=4)
    o5, d5 = raygen.edge_case_rays(model)
    # rays along triangle planes, and behind them the rays on which a distance-culled walk WITHOUT a guard was seen to
    # differ from the reference's walk on the GPU (tools/ab/stress_wide.py with WIDE_GUARD=0; nanosuit: all 349 of 2^30,
    # hindu: 2048 of 170 144)
    o6, d6 = raygen.grazing_rays(model, 2 * N_RAYS, seed=6)
    logged = np.load(os.path.join(HERE, "grazing_logged.npz"))
    if f"{name}_org" in logged:
        o6 = np.concatenate([o6, logged[f"{name}_org"]]); d6 = np.concatenate([d6, logged[f"{name}_dir"]])
    return {"primary": (o1, d1), "bounce": (o2, d2), "shadow": (o3, d3), "box": (o4, d4), "edge": (o5, d5), "grazing": (o6, d6)}


def main():
    assert ob.have_ref(), "build oracle/_ref first: make -C oracle ref"
    have_native = ob.have_ref(ob.REF_NATIVE_LIB)
    for name in ["cornell", "view_test", "hindu", "nanosuit", "soup"]:
        model = conftest.load_model(name)
        out = {}
        for strat, tag in ((0, "median"), (1, "middle"), (2, "sah")):
            ref = ob.RefScene(model, strategy=strat)
            for mi in range(len(model.meshes)):
                boxes, meta, perm = ref.tree(mi)
                out[f"tree_{tag}_{mi}_boxes"] = boxes
                out[f"tree_{tag}_{mi}_meta"] = meta
                out[f"tree_{tag}_{mi}_perm"] = perm
            if strat == 0:
                out["areas"] = np.array([ref.area(mi) for mi in range(len(model.meshes))], np.float32)
                batches = ray_batches(name, model, ref)
                refn = ob.RefScene(model, strategy=0, lib_path=ob.REF_NATIVE_LIB) if have_native else None
                for b, (o, d) in batches.items():
                    with np.errstate(all="ignore"):
                        ids, tuv = ref.closest_hit(o, d)
                    out[f"{b}_org"], out[f"{b}_dir"], out[f"{b}_ids"], out[f"{b}_tuv"] = o, d, ids, tuv
                    if refn is not None:
                        idn, tuvn = refn.closest_hit(o, d)
                        # rays on which two builds of the reference itself disagree are ill-conditioned
                        out[f"{b}_ill"] = ~((idn == ids).all(1) & (tuvn.view(np.uint32) == tuv.view(np.uint32)).all(1))
                    print(f"{name:10s} {b:8s} rays {len(o):6d} hit {100 * (ids[:, 0] != ob.INVALID).mean():5.1f}%"
                          + (f" ill {int(out[f'{b}_ill'].sum())}" if refn is not None else ""))
            else:
                o, d = raygen.box_rays(model, 2048, seed=10 + strat)
                ids, tuv = ref.closest_hit(o, d)
                out[f"{tag}_org"], out[f"{tag}_dir"], out[f"{tag}_ids"], out[f"{tag}_tuv"] = o, d, ids, tuv
        np.savez_compressed(os.path.join(HERE, f"rays_{name}.npz"), **out)

    if "--rays-only" in sys.argv:   # (the component vectors hold rsqrt results of the machine that made them)
        return
    # ---- component vectors --------------------------------------------------------------
    comp = {}
    rng = np.random.default_rng(5)
    for seed in (0, 42, 0x1234567890ABCDEF):
        for jumps in (0, 1, 3):
            comp[f"xoro_u64_{seed}_{jumps}"] = ob.ref_xoroshiro_u64(seed, 64, jumps)
            comp[f"xoro_f32_{seed}_{jumps}"] = ob.ref_xoroshiro_floats(seed, 64, jumps)
    model = conftest.load_model("cornell")
    ref = ob.RefScene(model)
    light = len(model.meshes) - 1
    comp["light_points"] = ref.random_points(light, 42, 256)
    comp["box_points"] = ref.random_points(5, 7, 256)
    v = (rng.random((512, 3)) * 4 - 2).astype(np.float32)
    comp["normalize_in"] = v
    comp["normalize_out"] = ob.ref_normalize(v)
    rows = (rng.random((512, 18)) * 2 - 1).astype(np.float32)
    rows[:, 12] = rng.random(512) * 30   # brdf: intensity
    rows[:, 15] = rng.random(512) * 30   # phong: intensity
    comp["shade_rows"] = rows
    for mi in (0, 4, 5, light):
        comp[f"brdf_{mi}"] = ref.color_brdf(mi, rows)
        comp[f"phong_{mi}"] = ref.color_phong(mi, rows)
    rr = (rng.random((256, 6)) * 2 - 1).astype(np.float32)
    comp["reflect_in"] = rr
    comp["reflect_out"] = ob.ref_reflect(rr)
    tri = rng.integers(0, 12, 256).astype(np.uint32)
    uu = rng.random(256).astype(np.float32); vv = (rng.random(256) * (1 - uu)).astype(np.float32)
    comp["attr_tri"], comp["attr_u"], comp["attr_v"] = tri, uu, vv
    comp["attr_out"] = ref.attributes(5, tri, uu, vv)
    # texture fetch through a real image file: an 8x4 binary PPM decoded by the reference's own loader
    tex = (rng.integers(0, 256, (4, 8, 3))).astype(np.uint8)
    ppm = os.path.join(HERE, "tex8x4.ppm")
    with open(ppm, "wb") as f:
        f.write(b"P6\n8 4\n255\n" + tex.tobytes())
    reft = ob.RefScene(model, texture_files={(0, 1): ppm})
    uv = (rng.random((512, 2)) * 3 - 0.5).astype(np.float32)
    uv[:, 1] = np.abs(uv[:, 1])   # negative rows are undefined in the reference (out-of-bounds read)
    comp["tex_uv"] = uv
    comp["tex_rgb_u8"] = tex
    comp["tex_out"] = reft.material_fetch(0, 0, uv)
    comp["solid_out"] = ref.material_fetch(0, 0, uv)
    for key, f in RTC.items():
        ints, floats, lights, obj = ob.ref_parse_rtc(os.path.join(conftest.ASSETS, f))
        comp[f"rtc_{key}_ints"], comp[f"rtc_{key}_floats"], comp[f"rtc_{key}_lights"] = ints, floats, lights
        comp[f"rtc_{key}_obj"] = np.frombuffer(obj.encode(), np.uint8)
        comp[f"cam_{key}"] = ob.ref_camera_vectors(floats[1:4], floats[4:7], floats[7:10], float(floats[10]), float(floats[0]))
    np.savez_compressed(os.path.join(HERE, "components.npz"), **comp)
    print("written", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
