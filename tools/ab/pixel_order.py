#!/usr/bin/env python3
"""Primary rays of a frame in row order (a wave = 64 pixels of one row) against 8x8 pixel blocks:
closest-hit throughput of the wave-synchronous walk.  Usage: python tools/ab/pixel_order.py [config ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from orion_amd import api
    for name in sys.argv[1:] or ["c3_nanosuit_fhd", "c1_cornell_brdf_fhd", "c4_hindu_4k"]:
        fr = api.load_frame(os.path.join(ROOT, "configs", name + ".rtc"), load_textures=False)
        W, H = 1920, 1080
        c = np.asarray(api.camera_vectors(fr.rtc.view_point, fr.rtc.look_at, fr.rtc.vector_up, fr.rtc.y_view, W / H), np.float32)
        eye, front, right, up = c[0:3], c[3:6], c[6:9], c[9:12]
        jj, ii = np.meshgrid(np.arange(W), np.arange(H))
        jj, ii = jj.ravel(), ii.ravel()
        sc = api.Scene(fr.model, device=0)
        sc.set_option("traversal", int(os.environ.get("TRAVERSAL", "2")))
        line = f"{name:22s}"
        for tag, order in (("rows", np.arange(W * H)), ("8x8 blocks", np.lexsort((jj % 8, ii % 8, jj // 8, ii // 8))),
                           ("16x4 blocks", np.lexsort((jj % 16, ii % 4, jj // 16, ii // 4)))):
            x = (2.0 * jj[order] / W - 1).astype(np.float32); y = -(2.0 * ii[order] / H - 1).astype(np.float32)
            d = (front + x[:, None] * right + y[:, None] * up).astype(np.float32)
            o = np.broadcast_to(eye, d.shape).astype(np.float32).copy()
            to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
            out = torch.empty((len(o), 5), dtype=torch.int32, device="cuda")
            sc.closest_hit(to, td, out)
            sc.profile(True)
            for _ in range(3):
                sc.closest_hit(to, td, out)
            st = sc.profile_read()["closest_hit"]
            sc.profile(False)
            line += f" | {tag}: {st['items'] / st['ms'] / 1e6:6.2f} G rays/s"
        print(line, flush=True)
        sc.close()


if __name__ == "__main__":
    main()
