#!/usr/bin/env python3
"""Random scenes through the renderer, on the GPU: the fuzzer's scenes (tools/ab/fuzz_scenes.py) with mirror-like
materials and a point light, or with two emissive meshes and no light (the BRDF path), rendered at 96 x 64, 8 spp, depth 4
with every traversal the scene offers.  The walks return the same hits bit for bit, so the accumulators must be
bit-identical; a few scenes are also compared with the oracle's render (PSNR).
Usage: python tools/ab/fuzz_render.py [first seed = 0] [scenes = 60]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "ab"))
import raygen  # noqa: E402
import fuzz_scenes as fz  # noqa: E402


def dress(model, seed, brdf):
    rng = np.random.default_rng(seed + 77)
    for k, mat in enumerate(model.materials):
        mat.diffuse = rng.uniform(0.2, 0.9, 3).astype(np.float32)
        mat.specular = rng.uniform(0.0, 0.8, 3).astype(np.float32)
        mat.ambient = rng.uniform(0.0, 0.1, 3).astype(np.float32)
        mat.shininess = float(rng.uniform(1, 40))
        if brdf and k < 2:
            mat.emissive = rng.uniform(2.0, 8.0, 3).astype(np.float32)
    return model


def main():
    from orion_amd import api
    from oracle import bindings as ob
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    W, H, SPP, DEPTH = 96, 64, 8, 4
    bad = 0
    for seed in range(first, first + count):
        brdf = seed % 2 == 1
        model = dress(fz.boxy_model(seed) if seed % 4 >= 2 else fz.random_model(seed), seed, brdf)
        eye, look, up, yv = raygen.default_camera(model)
        cam = api.camera_vectors(eye, look, up, yv, W / H)
        lo, hi = raygen.scene_bounds(model)
        lights = None if brdf else np.array([[*(hi + 0.5 * (hi - lo)), 1.0, 0.9, 0.8, float(((hi - lo) ** 2).sum()) * 4]], np.float32)
        sc = api.Scene(model, device=0)
        imgs = {}
        for mode in (1, 0, 4, 2):
            try:
                sc.set_option("traversal", mode)
            except api.OrionError:
                continue
            imgs[mode] = sc.render(cam, W, H, 0, SPP, 2, DEPTH, seed=seed + 1, lights=lights).copy()
        line = f"seed {seed}: {'brdf ' if brdf else 'point'} meshes {len(model.meshes)} triangles {model.n_triangles} modes {sorted(imgs)} lit pixels {int((imgs[1][..., :3].sum(-1) > 0).sum())}"
        for mode, img in imgs.items():
            if not np.array_equal(img.view(np.uint32), imgs[1].view(np.uint32)):
                bad += 1
                line += f" | mode {mode}: {int((img != imgs[1]).any(-1).sum())} PIXELS DIFFER"
        if seed % 8 < 2:      # the oracle's render of the same frame
            want = ob.OracleScene(model).render(cam, W, H, 0, SPP, 2, DEPTH, seed=seed + 1, lights=lights, threads=8)
            a, b = imgs[1][..., :3] / SPP, want[..., :3] / SPP
            peak = max(float(b.max()), 1e-6)
            mse = float(((a - b) ** 2).mean())
            psnr = 10 * np.log10(peak * peak / mse) if mse > 0 else float("inf")
            line += f" | against the oracle: PSNR {psnr:.1f} dB"
            if psnr < 50:
                bad += 1
                line += " LOW"
        print(line, flush=True)
        sc.close()
    print("scenes with a difference:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
