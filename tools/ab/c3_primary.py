#!/usr/bin/env python3
"""One 64-spp batch of the nanosuit frame without bounces (the first-bounce kernel is what a counter pass is after)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orion_amd import api
fr = api.load_frame(os.path.join(ROOT, "configs", "c3_nanosuit_fhd.rtc"), load_textures=False)
sc = api.Scene(fr.model, device=0)
acc = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
for _ in range(3):
    sc.render(fr.camera12, 1920, 1080, 0, 64, 1, 0, seed=1, lights=fr.rtc.lights, accum=acc)
torch.cuda.synchronize()
print("done")
