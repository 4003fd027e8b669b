#!/usr/bin/env python3
"""CPU model of the wide walk (oracle/wide_model.cpp): node / leaf / triangle visits per ray with the plain culls (mode 1)
and with the guarded culls (mode 2), and the number of rays on which each differs from the oracle's restatement of the
reference's walk.  Usage: python tools/ab/guard_model.py [scene ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import raygen  # noqa: E402
from oracle import bindings as ob  # noqa: E402

SLACK = float(os.environ.get("SLACK_LOG2") and 2.0 ** float(os.environ["SLACK_LOG2"]) or 2.0 ** -7)
MODES = [int(x) for x in os.environ.get('MODES', '1,2,4').split(',')]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def main():
    scenes = sys.argv[1:] or ["nanosuit", "soup", "hindu"]
    n = 1 << 16
    for name in scenes:
        model = conftest.load_model(name)
        wm = ob.WideModel(model); sc = ob.OracleScene(model)
        g = np.load(os.path.join(conftest.GOLDEN, f"rays_{name}.npz"))
        rng = np.random.default_rng(5)
        tris = np.concatenate([m.vertices[m.indices.reshape(-1, 3), :3] for m in model.meshes if m.n_triangles])
        pick = rng.integers(0, len(tris), n); a = rng.random(n); b = rng.random(n)
        flip = a + b > 1; a[flip] = 1 - a[flip]; b[flip] = 1 - b[flip]
        tp = tris[pick]
        p = tp[:, 0] + a[:, None] * (tp[:, 1] - tp[:, 0]) + b[:, None] * (tp[:, 2] - tp[:, 0])
        nr = np.cross(tp[:, 1] - tp[:, 0], tp[:, 2] - tp[:, 0]); ln = np.linalg.norm(nr, axis=1, keepdims=True)
        nr = nr / np.where(ln > 0, ln, 1)
        o3 = (p + 1e-3 * nr).astype(np.float32)
        batches = {}
        # bounce: cosine directions about the normal (either side)
        s = np.sqrt(rng.random(n)); c = np.sqrt(1 - s * s); psi = rng.random(n) * 2 * np.pi
        tang = np.cross(nr, np.array([0, 1, 0.0])); bad = (tang * tang).sum(1) == 0
        tang[bad] = np.cross(nr[bad], np.array([0, 0, 1.0])); bit = np.cross(nr, tang)
        d2 = (s * np.cos(psi))[:, None] * tang + (s * np.sin(psi))[:, None] * bit + c[:, None] * nr
        batches["bounce"] = (o3, d2.astype(np.float32))
        batches["to surface"] = raygen.rays_to_surface(o3, model, seed=8)
        batches["across box"] = raygen.box_rays(model, n, seed=9)
        batches["grazing"] = raygen.grazing_rays(model, n, seed=10)
        for bname, (o, d) in batches.items():
            oid, otuv = sc.closest_hit(o, d, threads=8)
            line = f"{name:9s} {bname:11s}"
            for mode in MODES:
                ids, tuv, st = wm.closest_hit(o, d, cull=mode, threads=8, kappa=SLACK)
                taken = ids[:, 0] != ob.WideModel.NOT_TAKEN
                bad = ((ids != oid).any(1) | (bits(tuv) != bits(otuv)).any(1)) & taken
                k = taken.sum()
                line += f" | mode {mode}: nodes {st['nodes'] / k:6.2f} leaves {st['leaves'] / k:6.2f} tris {st['tris'] / k:6.2f} differ {int(bad.sum()):5d}"
                if mode >= 2:
                    line += f" danger visits {st['danger_visits'] / k:5.2f}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
