#!/usr/bin/env python3
"""Per-kernel HIP-event time of one pass of a configured frame, one batch at a time (no two-lane overlap), for A/B
builds (ORION_HIP_LIB).  Usage: frame_kernels.py [config ...]   SPP=64, BVH=median|middle|sah, TRAVERSAL=<option> (environment)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orion_amd import api, host  # noqa: E402

strategy = {"median": host.MEDIAN, "middle": host.MIDDLE, "sah": host.SAH}[os.environ.get("BVH", "median")]

spp = int(os.environ.get("SPP", "64"))
for name in sys.argv[1:] or ["c3_nanosuit_fhd", "c3b_nanosuit_bounce_fhd"]:
    ls = 2 if "cornell" in name else 1
    fr = api.load_frame(os.path.join(ROOT, "configs", name + ".rtc"), load_textures=True, strategy=strategy)
    W, H = fr.rtc.xres, fr.rtc.yres
    sc = api.Scene(fr.model, device=0)
    sc.set_option("overlap_below", 0)
    if os.environ.get("TRAVERSAL"):
        sc.set_option("traversal", int(os.environ["TRAVERSAL"]))
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    args = dict(seed=1, lights=fr.rtc.lights, accum=acc)
    sc.render(fr.camera12, W, H, 0, spp, ls, fr.rtc.recursion_level, **args)
    best = None
    for rep in range(3):
        sc.profile(True)
        sc.render(fr.camera12, W, H, 0, spp, ls, fr.rtc.recursion_level, **args)
        st = sc.profile_read(); sc.profile(False)
        if best is None or sum(v["ms"] for v in st.values()) < sum(v["ms"] for v in best.values()):
            best = st
    print(name, f"{spp} spp |", " | ".join(f"{k} {v['ms']:.2f} ms {v['launches']} launches {v['items'] / 1e6:.1f} M items" for k, v in best.items() if v["launches"]),
          "| total %.2f ms" % sum(v["ms"] for v in best.values()), flush=True)
    sc.close()
