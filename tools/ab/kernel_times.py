#!/usr/bin/env python3
"""Per-kernel HIP-event time of one 128-spp pass of the headline frame (one batch at a time), for A/B builds
(ORION_HIP_LIB).  Images of experiment builds are not checked."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orion_amd import api, host  # noqa: E402

model = host.load_model(os.path.join(ROOT, "tests", "golden", "assets", "cornell-box", "cornell-box.obj"))
cam = api.camera_vectors([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0, 1920 / 1080)
sc = api.Scene(model, device=0)
sc.set_option("overlap_below", 0)
acc = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
sc.render(cam, 1920, 1080, 0, 128, 2, 100, seed=1, accum=acc)
best = None
for rep in range(3):
    sc.profile(True)
    sc.render(cam, 1920, 1080, 0, 128, 2, 100, seed=1, accum=acc)
    st = sc.profile_read(); sc.profile(False)
    t = {k: v["ms"] for k, v in st.items() if v["launches"]}
    if best is None or sum(t.values()) < sum(best.values()):
        best = t
print(" ".join(f"{k} {v:.2f} ms" for k, v in best.items()), "| total %.2f ms" % sum(best.values()))
