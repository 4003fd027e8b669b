#!/usr/bin/env python3
"""Per-kernel HIP-event times of the nanosuit and hindu config frames (one render call each)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from orion_amd import api  # noqa: E402

for name, ls, spp in (("c3_nanosuit_fhd",1,32),("c3b_nanosuit_bounce_fhd",1,32),("c4_hindu_4k",1,16)):
    fr = api.load_frame(os.path.join(ROOT, "configs", name + ".rtc"), load_textures=True)
    W, H = fr.rtc.xres, fr.rtc.yres
    sc = api.Scene(fr.model, device=0)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    sc.render(fr.camera12, W, H, 0, 4, ls, fr.rtc.recursion_level, seed=1, lights=fr.rtc.lights, accum=acc)
    sc.profile(True)
    t0=time.perf_counter()
    sc.render(fr.camera12, W, H, 0, spp, ls, fr.rtc.recursion_level, seed=1, lights=fr.rtc.lights, accum=acc)
    dt=time.perf_counter()-t0
    st = sc.profile_read()
    print(name, round(dt*1e3,2), 'ms', {k:(round(v['ms'],2), v['launches'], v['items']) for k,v in st.items() if v['launches']})
    sc.close()
