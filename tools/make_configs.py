#!/usr/bin/env python3
"""Writes the five derived scene files of BASELINE.json's `configs` under configs/ with the
product's .rtc writer (same grammar the reference's parse_rtc reads).  Resolution, depth and
lights live in the .rtc (orion/rtc_parser.cpp:88-127); samples and light samples are command-line
parameters of the renderer and are recorded in configs/README.md."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orion_amd import host  # noqa: E402

ASSETS = os.path.join(ROOT, "tests", "golden", "assets")
OUT = os.path.join(ROOT, "configs")

# name, source rtc, xres, yres, depth, lights (None = keep the file's), spp, light samples, note
CONFIGS = [
    ("c0_cornell_point_256", "cornell-box.rtc", 256, 256, 8, [[0, 1.5, 1.0, 100, 100, 100, 32]], 16, 1,
     "point-light ray tracer (the commented-out light of the asset moved inside the box)"),
    ("c1_cornell_brdf_fhd", "cornell-box.rtc", 1920, 1080, 100, [], 2048, 2, "README render: BRDF path tracer"),
    ("c2_view_test_fhd", "view_test.rtc", 1920, 1080, 1, None, 256, 1, "two-material quad scene, Phong + one mirror bounce"),
    ("c3_nanosuit_fhd", "nanosuit.rtc", 1920, 1080, 0, None, 512, 1, "19 058 triangles, six meshes, five PNG colour maps; depth 0 as the asset has it: primary and shadow rays"),
    ("c3b_nanosuit_bounce_fhd", "nanosuit.rtc", 1920, 1080, 4, None, 512, 1,
     "the same frame with four mirror bounces weighted by the specular texel (the map the reference installs there is the DIFFUSE "
     "image, orion/model.cpp:186-187): incoherent secondary rays through the deep trees"),
    ("c4_hindu_4k", "hindu.rtc", 3840, 2160, 0, None, 1024, 1, "tile + sample sharded over 8 GPUs"),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    lines = ["# Derived scene files for BASELINE.json `configs`", "",
             "Written by `tools/make_configs.py`; render with `orion_amd/orion_rt <file> -p SPP -l LIGHT_SAMPLES`.", "",
             "| file | resolution | depth | lights | -p | -l | note |", "|---|---|---|---|---|---|---|"]
    for name, src, w, h, depth, lights, spp, ls, note in CONFIGS:
        r = host.parse_rtc(os.path.join(ASSETS, src))
        r.obj_file = os.path.relpath(os.path.join(ASSETS, r.obj_file), OUT)
        r.xres, r.yres, r.aspect_ratio, r.recursion_level = w, h, w / h, depth
        if lights is not None:
            l = np.array(lights, np.float32).reshape(-1, 7)
            l[:, 3:6] /= 255.0
            r.lights = l
        host.write_rtc(os.path.join(OUT, name + ".rtc"), r)
        lines.append(f"| {name}.rtc | {w}x{h} | {depth} | {len(r.lights)} | {spp} | {ls} | {note} |")
    open(os.path.join(OUT, "README.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
