#!/usr/bin/env python3
"""Random scenes against the walks, on the GPU: per seed a scene of 1 to 40 meshes -- quads and single triangles (meshes
whose tree is one leaf), soups of 8 to 3000 triangles at scales from 1e-3 to 1e2, with zero-area triangles, needles and
duplicated triangles (equal t) mixed in -- and three ray batches (across the box, along triangle planes, surface to
surface plus the predicates' edge cases).  The unculled per-lane walk (option traversal 1, itself checked against the
oracle by the tests) is the reference; the default walk and the wide walk must return its bits.
Usage: python tools/ab/fuzz_scenes.py [first seed = 0] [scenes = 100] [log2 rays per batch = 17]
Environment: FUZZ_KIND=boxy for scenes of axis-aligned boxes on a lattice (coincident faces, shared edges, far origins)
with rays almost along the axes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raygen  # noqa: E402


def random_model(seed):
    from orion_amd import host
    rng = np.random.default_rng(seed)
    scale = 10.0 ** rng.uniform(-3, 2)
    parts = []
    for _ in range(int(rng.integers(1, 41))):
        kind = rng.random()
        centre = rng.uniform(0, 10, 3) * scale
        if kind < 0.25:       # a quad or a single triangle: a tree of one leaf
            a = rng.normal(size=3); b = rng.normal(size=3)
            a *= scale * 10.0 ** rng.uniform(-2, 0.5) / np.linalg.norm(a); b *= scale * 10.0 ** rng.uniform(-2, 0.5) / np.linalg.norm(b)
            q = np.zeros((4, 8), np.float32)
            q[:, :3] = centre + np.array([0 * a, a, a + b, b])
            q[:, 4] = 1.0
            idx = [0, 1, 2, 0, 2, 3] if rng.random() < 0.7 else [0, 1, 2]
            parts.append(host.model_from_arrays(q, np.array(idx, np.uint32)))
            continue
        n = int(2 ** rng.uniform(3, 11.5))
        v, i = raygen.triangle_soup(n, edge=10.0 ** rng.uniform(-2.5, -0.3), seed=int(rng.integers(1 << 30)))
        v[:, :3] = v[:, :3] * (scale * 10.0 ** rng.uniform(-1, 0.7)) + centre
        v = v.astype(np.float32)
        t = v.reshape(-1, 3, 8)
        k = max(1, n // 50)
        t[rng.integers(0, n, k), 2, :3] = t[rng.integers(0, n, k), 1, :3]                        # zero-area and odd triangles
        pick = rng.integers(0, n, k)
        t[pick, 2, :3] = t[pick, 0, :3] + (t[pick, 1, :3] - t[pick, 0, :3]) * np.float32(1 + 1e-4)   # needles
        src = rng.integers(0, n, k); dst = rng.integers(0, n, k)
        t[dst] = t[src]                                                                           # duplicates: equal t
        parts.append(host.model_from_arrays(t.reshape(-1, 8), i))
    return host.merge_models(parts)


def boxy_model(seed):
    """Axis-aligned boxes and slabs on a lattice, some sharing faces and edges, far from the origin at times: coplanar
    triangles, exactly equal hit distances, and whole families of triangles whose planes a ray can run along."""
    from orion_amd import host
    rng = np.random.default_rng(seed)
    off = rng.choice([0.0, 100.0, -3000.0]) + rng.uniform(0, 1, 3)
    step = 2.0 ** rng.integers(-3, 2)
    parts = []
    for _ in range(int(rng.integers(2, 30))):
        lo = off + rng.integers(0, 12, 3) * step
        size = rng.integers(0, 4, 3) * step            # (a zero size gives a flat slab: coincident faces)
        hi = lo + size
        c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                      [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
        v = np.zeros((8, 8), np.float32); v[:, :3] = c; v[:, 4] = 1.0
        idx = np.array([0, 1, 2, 0, 2, 3, 4, 6, 5, 4, 7, 6, 0, 4, 5, 0, 5, 1, 1, 5, 6, 1, 6, 2, 2, 6, 7, 2, 7, 3, 3, 7, 4, 3, 4, 0], np.uint32)
        reps = int(rng.integers(1, 4))                  # the same box as one mesh of 12, 24 or 36 triangles
        parts.append(host.model_from_arrays(v, np.tile(idx, reps)))
    if rng.random() < 0.5:                              # and something big enough for a wide tree
        sv, si = raygen.triangle_soup(int(2 ** rng.uniform(7, 11)), edge=0.05, seed=seed)
        sv[:, :3] = sv[:, :3] * (12 * step) + off
        parts.append(host.model_from_arrays(sv.astype(np.float32), si))
    return host.merge_models(parts)


def lattice_rays(model, n, seed):
    """Rays almost along the axes and the lattice planes: slopes of 10^-8 .. 10^-3 in the other two components."""
    rng = np.random.default_rng(seed)
    lo, hi = raygen.scene_bounds(model)
    p = lo + rng.random((n, 3)) * (hi - lo)
    snap = rng.random((n, 3)) < 0.5                     # half of the coordinates on a vertex coordinate of the scene
    allv = np.concatenate([m.vertices[:, :3] for m in model.meshes])
    p = np.where(snap, allv[rng.integers(0, len(allv), n)], p)
    axis = rng.integers(0, 3, n)
    d = 10.0 ** rng.uniform(-8, -3, (n, 3)) * rng.choice([-1.0, 1.0], (n, 3))
    d[np.arange(n), axis] = rng.choice([-1.0, 1.0], n)
    o = p - d * ((hi - lo).max() * rng.uniform(0.0, 2.0, n))[:, None]
    return o.astype(np.float32), (d * (10.0 ** rng.uniform(-1, 1, n))[:, None]).astype(np.float32)


def main():
    import torch
    from orion_amd import api
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 17)
    bad_total = 0
    for seed in range(first, first + count):
        boxy = os.environ.get("FUZZ_KIND", "") == "boxy"
        model = boxy_model(seed) if boxy else random_model(seed)
        sc = api.Scene(model, device=0)
        batches = {"box": raygen.box_rays(model, n, seed=seed), "grazing": raygen.grazing_rays(model, n, seed=seed + 1)}
        if boxy:
            batches["lattice"] = lattice_rays(model, n, seed=seed + 5)
        o3, d3 = raygen.box_rays(model, n, seed=seed + 2)
        sc.set_option("traversal", 1)
        h = sc.closest_hit(o3, d3)
        ok = h["mesh"] != api.INVALID
        p = (o3 + d3 * h["t"][:, None])[ok]
        if len(p) > 16:
            p = np.tile(p, (int(np.ceil(n / len(p))), 1))[:n].astype(np.float32)
            batches["surface"] = raygen.rays_to_surface(p, model, seed=seed + 3)
        eo, ed = raygen.edge_case_rays(model, seed=seed + 4)
        batches["edges"] = (eo, ed)
        modes = [0]
        try:
            sc.set_option("traversal", 4); modes.append(4)
        except api.OrionError:
            pass
        line = f"seed {seed}: meshes {len(model.meshes)} triangles {model.n_triangles} modes {modes}"
        for bname, (o, d) in batches.items():
            to, td = torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda()
            sc.set_option("traversal", 1)
            want = sc.closest_hit(to, td).clone()
            occ = sc.any_hit(to, td).clone()
            hits = int((want[:, 0] != -1).sum())
            for mode in modes:
                sc.set_option("traversal", mode)
                got = sc.closest_hit(to, td)
                differ = int((got != want).any(1).sum()) + int((sc.any_hit(to, td) != occ).sum())
                if differ:
                    bad_total += differ
                    k = int(torch.nonzero((got != want).any(1))[0]) if (got != want).any() else -1
                    line += f" | {bname} mode {mode}: {differ} DIFFER (first ray {k}: o {o[k].tolist()} d {d[k].tolist()})"
            line += f" | {bname} {hits}"
        print(line, flush=True)
        sc.close()
    print("rays that differ in total:", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
