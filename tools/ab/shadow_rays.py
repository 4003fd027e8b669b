#!/usr/bin/env python3
"""What the point-light shadow rays of the nanosuit frame cost as a ray-query batch: the primary hits of the frame in
pixel order (a few jittered samples per pixel), one ray from each towards the light, through orion_hip_any_hit with
each traversal option.  The shade kernel traces the same rays one per lane from start to end."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from orion_amd import api  # noqa: E402

fr = api.load_frame(os.path.join(ROOT, "configs", "c3_nanosuit_fhd.rtc"), load_textures=False)
W, H = 1920, 1080
sc = api.Scene(fr.model, device=0)
c = np.asarray(fr.camera12, np.float32); eye, front, right, up = c[0:3], c[3:6], c[6:9], c[9:12]
jj, ii = np.meshgrid(np.arange(W), np.arange(H))
os_, ds_ = [], []
rng = np.random.default_rng(1)
for k in range(int(os.environ.get("SPP", "8"))):
    jx, jy = rng.random(2)
    x = (2 * (jj.astype(np.float32) / W) - 1 + jx * 2 / W).astype(np.float32).ravel(); y = (2 * (ii.astype(np.float32) / H) - 1 + jy * 2 / H).astype(np.float32).ravel()
    d = (front[None, :] + x[:, None] * right[None, :] + (-y)[:, None] * up[None, :]).astype(np.float32)
    os_.append(np.repeat(eye[None, :], len(d), 0).astype(np.float32)); ds_.append(d)
o = np.concatenate(os_); d = np.concatenate(ds_)
h = sc.closest_hit(o, d)
hit = h["mesh"] != api.INVALID
nrm = np.zeros((len(o), 3), np.float32)
for mi, m in enumerate(fr.model.meshes):
    sel = np.where(h["mesh"] == mi)[0]
    if len(sel):
        nrm[sel] = m.vertices[m.indices[3 * h["tri"][sel]], 3:6]
nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
p = o + d * h["t"][:, None]
light = np.asarray(fr.rtc.lights[0][:3], np.float32)
so = (p + 1e-3 * nrm)[hit].astype(np.float32); sd = (light[None, :] - p[hit]).astype(np.float32)
print("shadow rays:", len(so), "of", len(o), "primary rays")
to, td = torch.from_numpy(so).cuda(), torch.from_numpy(sd).cuda()
ref = None
for mode in (1, 4, 0):
    sc.set_option("traversal", mode)
    occ = sc.any_hit(to, td); sc.profile(True)
    for _ in range(5):
        occ = sc.any_hit(to, td)
    st = sc.profile_read(); sc.profile(False)
    k = [v for v in st.values() if v["launches"]]
    ms = sum(v["ms"] for v in k) / 5
    if ref is None:
        ref = occ.clone()
    print(f"traversal {mode}: {ms:.3f} ms per batch, {len(so) / ms / 1e3:.0f} Mrays/s, occluded {float(occ.float().mean()):.3f}, same as unculled {bool(torch.equal(occ, ref))}")
