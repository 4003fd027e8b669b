#!/usr/bin/env python3
"""Do two builds of the kernel library render the same bits on every configured scene?  Renders each config of
configs/ at a reduced size (textures on, BRDF and point-light frames, the depth the config asks for and a deeper
one) with each library in a child process and compares the accumulators bit for bit.
Usage: python tools/ab/same_configs.py build_ab/a.so build_ab/b.so"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from orion_amd import api
out = {}
for name in sorted(f[:-4] for f in os.listdir(os.path.join(%r, "configs")) if f.endswith(".rtc")):
    fr = api.load_frame(os.path.join(%r, "configs", name + ".rtc"), load_textures=True)
    W, H = 480, 270
    sc = api.Scene(fr.model, device=0)
    ls = 2 if "cornell" in name else 1
    for depth in sorted({fr.rtc.recursion_level, 6}):
        out[f"{name}_d{depth}"] = sc.render(fr.camera12, W, H, 0, 12, ls, depth, seed=11, lights=fr.rtc.lights)
        if len(fr.rtc.lights):   # ... and the same scene lit by its emissive meshes, if it has any
            out[f"{name}_d{depth}_brdf"] = sc.render(fr.camera12, W, H, 0, 12, 2, depth, seed=11)
    sc.close()
np.savez(sys.argv[1], **out)
'''


def main():
    outs = []
    for k, lib in enumerate(sys.argv[1:3]):
        out = os.path.join(ROOT, "gpurun_out", f"same_configs_{k}.npz")
        env = dict(os.environ, ORION_HIP_LIB=os.path.abspath(lib))
        subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT, ROOT), out], check=True, env=env, timeout=600)
        outs.append(np.load(out))
    bad = 0
    for key in outs[0].files:
        a, b = outs[0][key], outs[1][key]
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        bad += not same
        print(f"{key:40s} identical bits: {same}   max |difference|: {np.abs(a - b).max():g}   mean radiance: {a[:, :, :3].mean():.6f}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
