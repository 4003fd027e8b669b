#!/usr/bin/env python3
"""One-off stress: many more rays than the test suite holds, every walk, against the oracle.
Usage: python tools/ab/stress_parity.py [rays_log2] [seeds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import conftest  # noqa: E402
import raygen  # noqa: E402


def main():
    from oracle import bindings as ob
    from orion_amd import api
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 21)
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    bad = 0
    for name in ("cornell", "hindu", "nanosuit", "soup"):
        model = conftest.load_model(name)
        sc = api.Scene(model, device=0)
        orc = ob.OracleScene(model)
        modes = (0, 1, 2, 3) if name == "cornell" else (0, 1, 2, 4)
        for seed in range(seeds):
            o, d = raygen.box_rays(model, n, seed=100 + seed)
            if seed % 2:   # coherent half: one origin, a narrow fan
                o[: n // 2] = o[0]
                d[: n // 2] = d[0] + (np.random.default_rng(seed).random((n // 2, 3)).astype(np.float32) - 0.5) * 0.2
            t0 = time.time()
            oid, otuv = orc.closest_hit(o, d, threads=16)
            t1 = time.time()
            for mode in modes:
                sc.set_option("traversal", mode)
                h = sc.closest_hit(o, d)
                ids = np.stack([h["mesh"], h["tri"]], 1); tuv = np.stack([h["t"], h["u"], h["v"]], 1)
                ok = np.array_equal(ids, oid) and np.array_equal(tuv.view(np.uint32), otuv.view(np.uint32))
                occ = sc.any_hit(o, d)
                ok = ok and np.array_equal(occ.astype(bool), oid[:, 0] != api.INVALID)
                bad += 0 if ok else 1
                print(f"{name:9s} seed {seed} mode {mode}: {'ok' if ok else 'MISMATCH'}  (oracle {t1 - t0:.1f} s, hits {float((oid[:, 0] != api.INVALID).mean()):.2f})", flush=True)
        sc.close()
    print("mismatching runs:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
