#!/usr/bin/env python3
"""Frames of several path batches: what batch size (option batch_paths) and the threshold of the two-lane overlap
(overlap_below) are worth.  Every setting renders once untimed (the queues are reallocated when the batch size
changes) and then twice timed.  Usage: large_frames.py [config ...]   SPP=512, SETTINGS="256:32,512:32" (environment)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orion_amd import api  # noqa: E402

SETTINGS = [(64 << 20, 4 << 20), (128 << 20, 8 << 20), (128 << 20, 16 << 20), (256 << 20, 16 << 20), (256 << 20, 32 << 20), (256 << 20, 64 << 20)]
if os.environ.get("SETTINGS"):      # "batch Mi:overlap Mi,..."
    SETTINGS = [tuple(int(v) << 20 for v in x.split(":")) for x in os.environ["SETTINGS"].split(",")]
spps = [int(x) for x in os.environ.get("SPP", "512").split(",")]
for name in sys.argv[1:] or ["c3_nanosuit_fhd", "c3b_nanosuit_bounce_fhd"]:
    ls = 2 if "cornell" in name else 1
    fr = api.load_frame(os.path.join(ROOT, "configs", name + ".rtc"), load_textures=True)
    W, H = fr.rtc.xres, fr.rtc.yres
    sc = api.Scene(fr.model, device=0)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    for bp, ov in SETTINGS:
        sc.set_option("batch_paths", bp); sc.set_option("overlap_below", ov)
        for spp in spps:
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sc.render(fr.camera12, W, H, 0, spp, ls, fr.rtc.recursion_level, seed=1, lights=fr.rtc.lights, accum=acc)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                if rep:
                    best = min(best, dt)
            print(f"{name} spp {spp} batch_paths {bp >> 20} Mi overlap_below {ov >> 20} Mi: {best * 1e3:.1f} ms  {W * H * spp / best / 1e9:.2f} G samples/s", flush=True)
    sc.close()
