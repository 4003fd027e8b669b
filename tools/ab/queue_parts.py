#!/usr/bin/env python3
"""What queue partitions (option queue_parts) are worth on the configured frames other than the headline's: 256-spp
frames with 1 and 16 partitions, bounce rays drawn by resident waves (draw_rays 1: one partition whatever the option
says) and four per thread (draw_rays 0).  Usage: python tools/ab/queue_parts.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orion_amd import api  # noqa: E402

for name in ("c3b_nanosuit_bounce_fhd", "c4_hindu_4k", "c2_view_test_fhd"):
    fr = api.load_frame(os.path.join(ROOT, "configs", name + ".rtc"), load_textures=True)
    W, H = fr.rtc.xres, fr.rtc.yres
    sc = api.Scene(fr.model, device=0)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    for dr in (1, 0):
        try: sc.set_option("draw_rays", dr)
        except Exception: pass
        for parts in (1, 16, 1, 16):
            sc.set_option("queue_parts", parts)
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sc.render(fr.camera12, W, H, 0, 256, 1, fr.rtc.recursion_level, seed=1, lights=fr.rtc.lights, accum=acc)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                if rep: best = min(best, dt)
            print(name, "draw_rays", dr, "parts", parts, f"{best*1e3:.1f} ms", flush=True)
    sc.close()
