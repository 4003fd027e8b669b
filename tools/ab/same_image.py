#!/usr/bin/env python3
"""Do two builds of the kernel library render the same bits?  Renders a Cornell frame (BRDF and
point-light) with each library in a child process and compares the accumulators bit for bit.
Usage: python tools/ab/same_image.py build_ab/a.so build_ab/b.so"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import bench
from orion_amd import api, host
model = host.load_model(%r)
eye, look, up, yv = bench.CAMERA
W, H = 640, 360
cam = api.camera_vectors(eye, look, up, yv, W / H)
sc = api.Scene(model, device=0)
a = sc.render(cam, W, H, 0, 16, 2, 60, seed=3)
lights = np.array([[0, 1.5, 1.0, 100 / 255, 100 / 255, 100 / 255, 32]], np.float32)
b = sc.render(cam, W, H, 0, 8, 1, 8, seed=3, lights=lights)
np.save(sys.argv[1], np.stack([a, b]))
'''


def main():
    outs = []
    for k, lib in enumerate(sys.argv[1:3]):
        out = os.path.join(ROOT, "gpurun_out", f"same_image_{k}.npy")
        env = dict(os.environ, ORION_HIP_LIB=os.path.abspath(lib))
        code = CHILD % (ROOT, os.path.join(ROOT, "tests", "golden", "assets", "cornell-box", "cornell-box.obj"))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=300)
        outs.append(np.load(out))
    same = np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    diff = np.abs(outs[0] - outs[1]).max()
    print(f"identical bits: {same}   max |difference|: {diff:g}   mean radiance: {outs[0][0][:, :, :3].mean():.6f}")
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
