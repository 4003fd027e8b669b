#!/usr/bin/env python3
"""Texture path at scale: the nanosuit frame (point-light shading: ambient, diffuse and specular
fetches per vertex) with every material's diffuse/specular map replaced by a synthetic RGBA8 image
of a given edge length.  One render per invocation so that a counter pass sees one size:

    python tools/texture_study.py <edge> [spp]        # prints one JSON line
    tools/texture_study.sh                            # all sizes, time + L2 hit rate, on a GPU box

The nearest-texel fetch touches 4 bytes per channel lookup, so what matters is how many distinct
64-byte lines the hit points of a frame spread over: a 4096^2 RGBA8 map is 64 MiB per material.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    edge = int(sys.argv[1])
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import torch
    from orion_amd import api, host
    model = host.load_model(os.path.join(ROOT, "tests", "golden", "assets", "nanosuit2", "nanosuit2.obj"))
    rng = np.random.default_rng(1)
    n_tex = 0
    for m in model.materials:
        if edge > 0 and m.map_diffuse:
            img = rng.integers(0, 256, (edge, edge, 4), dtype=np.uint8)
            m.tex_diffuse = host.Texture(img)
            m.tex_specular = m.tex_diffuse       # the reference installs the diffuse image as the specular map
            n_tex += 1
    W, H = 1920, 1080
    cam = api.camera_vectors([0, 7.5, 17], [0, 7.5, 0], [0, 1, 0], 1.0, W / H)
    lights = np.float32([[-3, 10, 10, 200, 200, 200, 32]])
    sc = api.Scene(model, device=0)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    sc.render(cam, W, H, 0, 1, 1, 1, seed=1, lights=lights, accum=accum)   # warm-up
    torch.cuda.synchronize()
    sc.profile(True)
    t0 = time.perf_counter()
    sc.render(cam, W, H, 1, 1 + spp, 1, 1, seed=1, lights=lights, accum=accum)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = sc.profile_read()
    print(json.dumps({"edge": edge, "textures": n_tex, "texel_MiB": n_tex * edge * edge * 4 / 2**20, "spp": spp,
                      "ms": 1e3 * dt, "Msamples_per_s": W * H * spp / dt / 1e6,
                      "shade_ms": st["shade"]["ms"], "vertices": st["shade"]["items"]}), flush=True)
    sc.close()


if __name__ == "__main__":
    main()
