#!/usr/bin/env python3
"""Where does the wave-synchronous walk stop paying?  Closest-hit throughput of the three walks on
tessellated spheres of growing size (one connected mesh over a ground quad), for rays in pixel
order (coherent), cosine bounce rays and rays towards surface points.
Usage: python tools/ab/mode_threshold.py [triangle counts ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import raygen  # noqa: E402


def sphere(n_tri):
    """UV sphere of about n_tri triangles on a ground quad; vertices [*, 8], indices."""
    rings = max(2, int(round(np.sqrt(n_tri / 4)))); segs = 2 * rings
    th = np.linspace(0, np.pi, rings + 1); ph = np.linspace(0, 2 * np.pi, segs + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    pos = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    v = np.zeros((len(pos) + 4, 8), np.float32)
    v[:len(pos), 0:3] = pos * 0.8 + [0, 1, 0]; v[:len(pos), 3:6] = pos
    idx = []
    for i in range(rings):
        for j in range(segs):
            a = i * (segs + 1) + j; b = a + segs + 1
            idx += [a, b, a + 1, a + 1, b, b + 1]
    g = len(pos)
    v[g:g + 4, 0:3] = [[-3, 0, -3], [3, 0, -3], [3, 0, 3], [-3, 0, 3]]; v[g:g + 4, 3:6] = [0, 1, 0]
    idx += [g, g + 1, g + 2, g, g + 2, g + 3]
    return v, np.array(idx, np.uint32)


def main():
    import torch
    from orion_amd import api, host
    sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048, 8192]
    W, H = 1024, 1024
    n = W * H
    for n_tri in sizes:
        v, ix = sphere(n_tri)
        model = host.model_from_arrays(v, ix)
        sc = api.Scene(model, device=0)
        cam = api.camera_vectors([0, 1.5, 4], [0, 1, 0], [0, 1, 0], 1.0, 1.0)
        c = np.asarray(cam, np.float32); eye, front, right, up = c[0:3], c[3:6], c[6:9], c[9:12]
        jj, ii = np.meshgrid(np.arange(W), np.arange(H))
        # pixel order in 8x8 blocks, as neighbouring lanes see neighbouring pixels in a frame
        order = np.lexsort((jj.ravel() % 8, ii.ravel() % 8, jj.ravel() // 8, ii.ravel() // 8))
        x = (2.0 * jj.ravel()[order] / W - 1).astype(np.float32); y = -(2.0 * ii.ravel()[order] / H - 1).astype(np.float32)
        d1 = (front + x[:, None] * right + y[:, None] * up).astype(np.float32)
        o1 = np.broadcast_to(eye, d1.shape).astype(np.float32).copy()
        h = sc.closest_hit(o1, d1)
        hit = h["mesh"] != api.INVALID
        nrm = np.zeros((n, 3), np.float32)
        m = model.meshes[0]
        nrm[hit] = m.vertices[m.indices[3 * h["tri"][hit]], 3:6]
        o2, d2 = raygen.bounce_rays(o1, d1, np.stack([h["mesh"], h["tri"]], 1), np.stack([h["t"], h["u"], h["v"]], 1), nrm, seed=2)
        o3, d3 = raygen.rays_to_surface(o2, model, seed=3)
        batches = {"pixel order": (o1, d1), "bounce": (o2, d2), "to surface": (o3, d3)}
        nodes = sum(len(x.nodes) for x in model.meshes)
        line = f"{model.n_triangles:6d} triangles {nodes:6d} nodes |"
        for bname, (o, d) in batches.items():
            to, td = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
            out = torch.empty((len(o), 5), dtype=torch.int32, device="cuda")
            rates = []
            for mode in (1, 2, 4):
                try:
                    sc.set_option("traversal", mode)
                except api.OrionError:
                    rates.append(0.0)
                    continue
                sc.closest_hit(to, td, out)
                sc.profile(True)
                for _ in range(3):
                    sc.closest_hit(to, td, out)
                st = sc.profile_read()["closest_hit"]
                sc.profile(False)
                rates.append(st["items"] / st["ms"] / 1e6)
            line += f" {bname}: lane {rates[0]:6.2f} wave {rates[1]:6.2f} wide {rates[2]:6.2f} G/s |"
        print(line, flush=True)
        sc.close()


if __name__ == "__main__":
    main()
