#!/usr/bin/env python3
"""Does the chip gain from two render loops in flight at once?  Two scene handles (each with its
own stream, queues and counters) render half of the samples each from two host threads; compared
with one handle rendering all of them.  Usage: python tools/ab/two_scenes.py [spp]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    import torch
    from orion_amd import api, host
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    model = host.load_model(os.path.join(ROOT, "tests", "golden", "assets", "cornell-box", "cornell-box.obj"))
    eye, look, up, yv = bench.CAMERA
    cam = api.camera_vectors(eye, look, up, yv, bench.W / bench.H)
    nmax = 4
    scenes = [api.Scene(model, device=0) for _ in range(nmax)]
    accums = [torch.zeros((bench.H, bench.W, 4), dtype=torch.float32, device="cuda") for _ in range(nmax)]

    def render(i, lo, hi, delay=0.0):
        if delay:
            t_end = time.perf_counter() + delay
            while time.perf_counter() < t_end:
                pass
        scenes[i].render(cam, bench.W, bench.H, lo, hi, bench.LIGHT_SAMPLES, bench.MAX_DEPTH, seed=1, accum=accums[i])

    def many(n):
        def fn():
            ts = [threading.Thread(target=render, args=(i, i * spp // n, (i + 1) * spp // n)) for i in range(n)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        return fn

    def staggered(ms):
        def fn():
            ts = [threading.Thread(target=render, args=(i, i * spp // 2, (i + 1) * spp // 2, i * ms * 1e-3)) for i in range(2)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        return fn

    cases = [("1 handle", many(1))] + [(f"2 handles, second {ms} ms late", staggered(ms)) for ms in (0, 1, 2, 3, 4.5, 6, 7.5)] + [("1 handle", many(1))]
    for name, fn in cases:
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name:32s} {spp} spp: {dt * 1e3:8.2f} ms  {bench.W * bench.H * spp / dt / 1e9:.2f} G samples/s", flush=True)
    for s in scenes:
        s.close()


if __name__ == "__main__":
    main()
