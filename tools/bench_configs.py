#!/usr/bin/env python3
"""Throughput of the five BASELINE.json configs at their full resolution on one GPU (a bounded
number of samples per pixel, projected to the config's spp).  Prints a markdown table."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = [("c0_cornell_point_256", 16, 1, 16), ("c1_cornell_brdf_fhd", 2048, 2, 64), ("c2_view_test_fhd", 256, 1, 64),
           ("c3_nanosuit_fhd", 512, 1, 32), ("c3b_nanosuit_bounce_fhd", 512, 1, 32), ("c4_hindu_4k", 1024, 1, 16)]


def main():
    import torch
    from orion_amd import api
    print("| config | resolution | depth | traversal | spp timed | Msamples/s | Mrays/s | projected wall (config spp) |")
    print("|---|---|---|---|---|---|---|---|")
    only = sys.argv[1:]
    for name, spp_full, ls, spp in CONFIGS:
        if only and name not in only:
            continue
        fr = api.load_frame(os.path.join(ROOT, "configs", name + ".rtc"), load_textures=True)   # (hindu's JPEG is not in the tree: solid colour)
        W, H = fr.rtc.xres, fr.rtc.yres
        sc = api.Scene(fr.model, device=0)
        forced = int(os.environ.get("TRAVERSAL", "0"))     # A/B runs: 1 per-lane, 2 wave-synchronous
        if forced:
            sc.set_option("traversal", forced)
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        # (untimed, with the same sample count: the queues are sized for the call's samples per batch)
        sc.render(fr.camera12, W, H, 0, spp, ls, fr.rtc.recursion_level, seed=1, lights=fr.rtc.lights, accum=acc)
        sc.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.render(fr.camera12, W, H, 0, spp, ls, fr.rtc.recursion_level, seed=1, lights=fr.rtc.lights, accum=acc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = sc.profile_read()
        rays = st["closest_hit"]["items"] + st["shadow"]["items"]
        n_ops = sum(len(m.nodes) for m in fr.model.meshes)
        mode = {1: "per-lane", 2: "wave-synchronous", 4: "wide tree"}.get(forced) or ("wave-synchronous" if n_ops <= 384 else "wide tree, primary rays wave-synchronous")
        print(f"| {name} | {W}x{H} | {fr.rtc.recursion_level} | {mode} | {spp} | {W * H * spp / dt / 1e6:.1f} | {rays / dt / 1e6:.1f} | "
              f"{W * H * spp_full / (W * H * spp / dt):.2f} s |", flush=True)
        sc.close()


if __name__ == "__main__":
    main()
