#!/usr/bin/env python3
"""The ordered, distance-culled walk of the 4-wide tree against the unculled per-lane walk on the GPU, on
as many rays as one cares to wait for: rays from surface points to surface points (they graze the surfaces they
leave and reach), rays from surface points in random directions, and rays across the scene's box, generated on
the device.  Prints the number of rays whose five output words differ.
Usage: python tools/ab/stress_wide.py [log2 of rays per scene and kind = 28] [scene ...]
Environment: STRESS_KINDS (kinds to run, separated by ';'), WIDE_GUARD (0: the wide walk without its guard),
STRESS_DUMP (file under gpurun_out/ that receives the rays that differ, per scene and kind, as an .npz)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import conftest  # noqa: E402


def main():
    import torch
    from orion_amd import api
    total_log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    scenes = sys.argv[2:] or ["nanosuit", "soup", "hindu"]
    n = 1 << 22
    rounds = max(1, (1 << total_log2) // n)
    g = torch.Generator(device="cuda"); g.manual_seed(12345)
    grand = 0
    dump = {}
    for name in scenes:
        model = conftest.load_model(name)
        sc = api.Scene(model, device=0)
        if os.environ.get("WIDE_GUARD"):
            sc.set_option("wide_guard", int(os.environ["WIDE_GUARD"]))
        tris = torch.from_numpy(np.concatenate([m.vertices[m.indices.reshape(-1, 3), :3] for m in model.meshes if m.n_triangles])).cuda()
        lo = tris.reshape(-1, 3).min(0).values; hi = tris.reshape(-1, 3).max(0).values

        def surface_points(k):
            pick = torch.randint(0, len(tris), (k,), device="cuda", generator=g)
            a = torch.rand(k, device="cuda", generator=g); b = torch.rand(k, device="cuda", generator=g)
            flip = a + b > 1
            a = torch.where(flip, 1 - a, a); b = torch.where(flip, 1 - b, b)
            tp = tris[pick]
            p = tp[:, 0] + a[:, None] * (tp[:, 1] - tp[:, 0]) + b[:, None] * (tp[:, 2] - tp[:, 0])
            nr = torch.linalg.cross(tp[:, 1] - tp[:, 0], tp[:, 2] - tp[:, 0])
            nr = nr / nr.norm(dim=1, keepdim=True).clamp_min(1e-30)
            return p, nr

        kinds = {"surface to surface": 0, "surface, random direction": 0, "across the box": 0, "grazing": 0}
        if os.environ.get("STRESS_KINDS"):
            kinds = {k: 0 for k in os.environ["STRESS_KINDS"].split(";")}   # (one kind has a comma in its name)
        t0 = time.time()
        for r in range(rounds):
            for kind in kinds:
                if kind == "across the box":
                    c = (lo + hi) / 2; ext = (hi - lo).max()
                    o = c + (torch.rand((n, 3), device="cuda", generator=g) - 0.5) * ext * 2.5
                    d = lo + torch.rand((n, 3), device="cuda", generator=g) * (hi - lo) - o
                elif kind == "grazing":
                    # rays that run along a triangle's plane at a slope between 1e-8 and 1e-2, through a point of the
                    # triangle or just beside it, from a point up to a few scene sizes back: where a triangle test's
                    # t, u, v are worst conditioned
                    pick = torch.randint(0, len(tris), (n,), device="cuda", generator=g)
                    tp = tris[pick]
                    e1 = tp[:, 1] - tp[:, 0]; e2 = tp[:, 2] - tp[:, 0]
                    nr = torch.linalg.cross(e1, e2); nr = nr / nr.norm(dim=1, keepdim=True).clamp_min(1e-30)
                    a = torch.rand(n, device="cuda", generator=g) * 1.4 - 0.2; b = torch.rand(n, device="cuda", generator=g) * 1.4 - 0.2
                    p = tp[:, 0] + a[:, None] * e1 + b[:, None] * e2
                    ang = torch.rand(n, device="cuda", generator=g) * 6.2831853
                    u1 = e1 / e1.norm(dim=1, keepdim=True).clamp_min(1e-30); u2 = torch.linalg.cross(nr, u1)
                    inplane = torch.cos(ang)[:, None] * u1 + torch.sin(ang)[:, None] * u2
                    slope = 10.0 ** (-8.0 + 6.0 * torch.rand(n, device="cuda", generator=g)) * (torch.randint(0, 2, (n,), device="cuda", generator=g) * 2 - 1)
                    d = inplane + slope[:, None] * nr
                    back = (hi - lo).max() * 10.0 ** (-2.0 + 2.5 * torch.rand(n, device="cuda", generator=g))
                    o = p - back[:, None] * d
                    d = d * 10.0 ** (-1.0 + 2.0 * torch.rand(n, device="cuda", generator=g))[:, None]
                else:
                    p, nr = surface_points(n)
                    o = p + 1e-3 * nr
                    if kind == "surface to surface":
                        d = surface_points(n)[0] - o
                    else:
                        d = torch.randn((n, 3), device="cuda", generator=g)
                o = o.float().contiguous(); d = d.float().contiguous()
                sc.set_option("traversal", 1)
                want = sc.closest_hit(o, d).clone()
                sc.set_option("traversal", 4)
                got = sc.closest_hit(o, d)
                bad = int((want != got).any(1).sum())
                kinds[kind] += bad
                if bad and os.environ.get("STRESS_DUMP"):
                    sel = (want != got).any(1).nonzero()[:, 0]
                    key = f"{name}/{kind}"
                    dump.setdefault(key + "/org", []).append(o[sel].cpu().numpy()); dump.setdefault(key + "/dir", []).append(d[sel].cpu().numpy())
                if bad:
                    i = int((want != got).any(1).nonzero()[0])
                    print(f"  {name} / {kind}: ray o={o[i].tolist()} d={d[i].tolist()} unculled {want[i].tolist()} wide {got[i].tolist()}", flush=True)
            if (r + 1) % 16 == 0:
                print(f"{name}: {(r + 1) * n:>11d} rays per kind, differing so far {kinds}  ({time.time() - t0:.0f} s)", flush=True)
        print(f"{name}: {rounds * n} rays per kind, rays that differ: {kinds}", flush=True)
        grand += sum(kinds.values())
        sc.close()
    if dump:
        path = os.path.join(ROOT, "gpurun_out", os.environ["STRESS_DUMP"])
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez_compressed(path, **{k: np.concatenate(v) for k, v in dump.items()})
        print("rays that differ written to", path)
    print("rays that differ in total:", grand)
    sys.exit(1 if grand else 0)


if __name__ == "__main__":
    main()
