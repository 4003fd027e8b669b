#!/usr/bin/env python3
"""Lane occupancy of the ray-drawing wide walk (k_query_wide) on the ray-bench batches, from a build of the
library with -DORION_WIDE_STATS=1 (ORION_HIP_LIB points at it).  Usage: wide_stats.py [scene ...]   WIDE_GUARD=0: without the walk's guard"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import conftest, raygen
    from orion_amd import api
    L = api.lib()
    L.orion_hip_debug_wide_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.orion_hip_debug_wide_times.argtypes = [ctypes.c_void_p]
    n = 1 << int(os.environ.get("RAYS_LOG2", "22"))
    for name in sys.argv[1:] or ["nanosuit", "soup"]:
        model = conftest.load_model(name)
        sc = api.Scene(model, device=0)
        sc.set_option("traversal", 4)
        if os.environ.get("WIDE_GUARD"):
            sc.set_option("wide_guard", int(os.environ["WIDE_GUARD"]))
        eye, look, up, yv = raygen.default_camera(model)
        cam = api.camera_vectors(eye, look, up, yv, 16 / 9)
        o1, d1 = raygen.primary_rays(cam, 1920, 1080, n, seed=1)
        h = sc.closest_hit(o1, d1)
        nrm = np.zeros((n, 3), np.float32)
        for mi, m in enumerate(model.meshes):
            sel = np.where(h["mesh"] == mi)[0]
            if len(sel):
                nrm[sel] = m.vertices[m.indices[3 * h["tri"][sel]], 3:6]
        o2, d2 = raygen.bounce_rays(o1, d1, np.stack([h["mesh"], h["tri"]], 1), np.stack([h["t"], h["u"], h["v"]], 1), nrm, seed=2)
        rep = int(np.ceil(n / len(o2)))
        o2 = np.tile(o2, (rep, 1))[:n]; d2 = np.tile(d2, (rep, 1))[:n]
        perm = np.random.default_rng(5).permutation(n)
        o2, d2 = o2[perm], d2[perm]
        o3, d3 = raygen.rays_to_surface(o2, model, seed=3)
        for bname, (o, d) in {"R1 primary": (o1, d1), "R2 bounce": (o2, d2), "R3 shadow": (o3, d3)}.items():
            to, td = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
            out = torch.empty((n, 5), dtype=torch.int32, device="cuda")
            sc.closest_hit(to, td, out)
            st = (ctypes.c_ulonglong * 16)()
            L.orion_hip_debug_wide_stats(st, 1)
            sc.closest_hit(to, td, out)
            L.orion_hip_debug_wide_stats(st, 1)
            s = [int(x) for x in st]
            outer, nsteps, nlanes, lsteps, llanes, trips, ltris, refills, handed, shares, slow, waves = s[:12]
            print(f"{name} {bname}: rays {n}  waves {waves}  per ray: node visits {nlanes / n:.2f}  leaf visits {llanes / n:.2f}  triangles {ltris / n:.2f}")
            print(f"   node steps {nsteps} with {nlanes / max(1, nsteps):.1f} lanes; leaf steps {lsteps} with {llanes / max(1, lsteps):.1f} lanes, "
                  f"{trips / max(1, lsteps):.2f} triangle turns each ({ltris / max(1, trips):.1f} lanes per turn); outer turns {outer}; "
                  f"refills {refills} handing out {handed / max(1, refills):.1f} rays; share rounds {shares}; unculled {slow}")
            tm = np.zeros((8192, 4), np.uint64)
            L.orion_hip_debug_wide_times(tm.ctypes.data_as(ctypes.c_void_p))
            tm = tm[:waves].astype(np.int64)
            t0 = tm[:, 0].min()
            us = lambda a: (a - t0) / 100.0
            q = lambda a: " ".join(f"{x:.0f}" for x in np.percentile(us(a), [0, 10, 50, 90, 99, 100]))
            print(f"   per wave, microseconds after the first wave's start (min p10 p50 p90 p99 max): start {q(tm[:, 0])} | pool empty {q(tm[:, 1])} | end {q(tm[:, 2])}; rays per wave min/mean/max {tm[:, 3].min()}/{tm[:, 3].mean():.0f}/{tm[:, 3].max()}")
            print(f"   wave-level per ray: node steps {nsteps / n:.3f}  triangle turns {trips / n:.3f}  outer turns {outer / n:.3f}  refills {refills / n:.3f}")
        sc.close()


if __name__ == "__main__":
    main()
