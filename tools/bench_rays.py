#!/usr/bin/env python3
"""Closest-hit throughput (Mrays/s) of the traversal kernels on device-resident ray batches,
timed with the library's HIP events.  Usage: python tools/bench_rays.py [scene ...]
Environment: RAYS_LOG2 (batch size, default 22), RAYS_ONLY (name of the one batch to run),
RAYS_MODES (comma-separated traversal options to run, default "1,4,0,2"),
WIDE_GUARD (0: the wide walk without its guard, option "wide_guard")."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import conftest  # noqa: E402
import raygen  # noqa: E402


def main():
    import torch
    from orion_amd import api
    scenes = sys.argv[1:] or ["cornell", "hindu", "nanosuit", "soup"]
    n = 1 << int(os.environ.get("RAYS_LOG2", "22"))
    for name in scenes:
        model = conftest.load_model(name)
        sc = api.Scene(model, device=0)
        if os.environ.get("WIDE_GUARD"):
            sc.set_option("wide_guard", int(os.environ["WIDE_GUARD"]))
        eye, look, up, yv = raygen.default_camera(model) if name != "cornell" else ([-0.10141, 1, 3.00498], [-0.10141, 1, 2.00498], [0, 1, 0], 1.0)
        cam = api.camera_vectors(eye, look, up, yv, 16 / 9)
        o1, d1 = raygen.primary_rays(cam, 1920, 1080, n, seed=1)
        # sort primary rays by pixel so that waves are coherent, as the generate kernel's are
        batches = {"R1 primary": (o1, d1)}
        h = sc.closest_hit(o1, d1)
        hit = h["mesh"] != api.INVALID
        # bounce rays: geometric normal from the oracle-free route: use interpolated normals of the hit triangle's first vertex
        nrm = np.zeros((n, 3), np.float32)
        for mi, m in enumerate(model.meshes):
            sel = np.where(h["mesh"] == mi)[0]
            if len(sel):
                nrm[sel] = m.vertices[m.indices[3 * h["tri"][sel]], 3:6]
        ids = np.stack([h["mesh"], h["tri"]], 1)
        tuv = np.stack([h["t"], h["u"], h["v"]], 1)
        o2, d2 = raygen.bounce_rays(o1, d1, ids, tuv, nrm, seed=2)
        if len(o2) > 1000:
            rep = int(np.ceil(n / len(o2)))
            o2 = np.tile(o2, (rep, 1))[:n]; d2 = np.tile(d2, (rep, 1))[:n]
            perm = np.random.default_rng(5).permutation(n)
            batches["R2 bounce"] = (o2[perm], d2[perm])
            o3, d3 = raygen.rays_to_surface(o2[perm], model, seed=3)
            batches["R3 shadow"] = (o3, d3)
            if name == "cornell":
                # shadow rays as the path tracer casts them: towards the light mesh only
                import copy
                lm = copy.copy(model); lm.meshes = [model.meshes[-1]]
                o4, d4 = raygen.rays_to_surface(o2[perm], lm, seed=4)
                batches["R4 to-light"] = (o4, d4)
                # the same rays grouped by the surface they start on (mesh, triangle of the previous hit)
                key = (ids[hit][:, 0].astype(np.int64) * 1000 + ids[hit][:, 1])
                key = np.tile(key, rep)[:n][perm]
                order = np.argsort(key, kind="stable")
                batches["R4 binned"] = (o4[order], d4[order])
                batches["R2 binned"] = (o2[perm][order], d2[perm][order])
                # grouped inside tiles of 1024 consecutive rays only, as k_extend does it: by triangle,
                # and by triangle + quadrant of the barycentrics (a finer spatial key)
                uvq = tuv[hit][:, 1:3]
                uvq = np.tile(uvq, (rep, 1))[:n][perm]
                quad = (uvq[:, 0] > 0.25).astype(np.int64) * 2 + (uvq[:, 1] > 0.25).astype(np.int64)
                tile = np.arange(n) // 1024
                for nm, k2 in (("R4 tile tri", key), ("R4 tile tri+q", key * 4 + quad)):
                    o_t = np.lexsort((k2, tile))
                    batches[nm] = (o4[o_t], d4[o_t])
                # ... and with larger tiles
                for size in (65536, 1024, 16384, 4096, 2048):
                    o_t = np.lexsort((key, np.arange(n) // size))
                    batches[f"R4 tile {size}"] = (o4[o_t], d4[o_t])
                # ... and inside each surface by direction octant
                dd = d2[perm]
                octant = (dd[:, 0] > 0).astype(np.int64) | ((dd[:, 1] > 0).astype(np.int64) << 1) | ((dd[:, 2] > 0).astype(np.int64) << 2)
                order8 = np.argsort(key * 8 + octant, kind="stable")
                batches["R2 bin+oct"] = (o2[perm][order8], d2[perm][order8])
        o5, d5 = raygen.box_rays(model, n, seed=6)
        batches["R5 box"] = (o5, d5)
        if os.environ.get("RAYS_SORT"):                       # what ordering the rays by origin cell and direction octant is worth
            lo, hi = raygen.scene_bounds(model)
            bits = int(os.environ["RAYS_SORT"])
            for bname in list(batches):
                o, d = batches[bname]
                cell = np.clip(((o - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
                octant = (d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2)
                key = (((cell[:, 0] << bits | cell[:, 1]) << bits | cell[:, 2]) << 3) | octant
                order = np.argsort(key, kind="stable")
                batches[bname] = (o[order], d[order])
        exact = {}                                            # results of the unculled per-lane walk, batch by batch
        wanted = [int(x) for x in os.environ.get("RAYS_MODES", "1,4,0,2").split(",")]
        for mode, tag in ((1, "lane"), (4, "wide"), (0, "auto"), (2, "wave")):
            if mode not in wanted or (tag == "wave" and model.n_triangles > 2000):
                continue
            try:
                sc.set_option("traversal", mode)
            except api.OrionError as e:
                print(f"{name:9s} {tag:5s} not available: {e}", flush=True)
                continue
            only = os.environ.get("RAYS_ONLY")                # run just this batch (counter passes)
            for bname, (o, d) in batches.items():
                if only and bname != only:
                    continue
                to, td = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
                out = torch.empty((len(o), 5), dtype=torch.int32, device="cuda")
                sc.closest_hit(to, td, out)
                sc.profile(True)
                for _ in range(5):
                    sc.closest_hit(to, td, out)
                st = sc.profile_read()["closest_hit"]
                sc.profile(False)
                same = ""
                if tag == "lane":
                    exact[bname] = out.clone()
                elif bname in exact:
                    same = "  same bits as lane: %s" % bool(torch.equal(out, exact[bname]))
                print(f"{name:9s} {tag:5s} {bname:11s} rays {len(o):8d}  {st['items'] / st['ms'] / 1e3:9.1f} Mrays/s  ({st['ms'] / st['launches']:.3f} ms/launch){same}", flush=True)
        sc.close()


if __name__ == "__main__":
    main()
