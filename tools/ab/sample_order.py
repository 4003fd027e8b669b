#!/usr/bin/env python3
"""Camera rays of a frame, 64 jittered samples per pixel of the frame's central window: lanes of a wave as 64 pixels of
a row in one sample (the order the render loop generates them in) against 64 samples of one pixel.  Closest-hit
throughput of the wave-synchronous walk (traversal 2) on device-resident rays.
Usage: python tools/ab/sample_order.py [config ...]"""
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
        W, H, NS = 1920, 1080, 64
        c = np.asarray(api.camera_vectors(fr.rtc.view_point, fr.rtc.look_at, fr.rtc.vector_up, fr.rtc.y_view, W / H), np.float32)
        eye, front, right, up = c[0:3], c[3:6], c[6:9], c[9:12]
        jj, ii = np.meshgrid(np.arange(W // 4, 3 * W // 4), np.arange(H // 4, 3 * H // 4))
        jj, ii = jj.ravel(), ii.ravel()
        P = len(jj)
        rng = np.random.default_rng(1)
        jit = rng.random((NS, 2)).astype(np.float32) * np.array([2.0 / W, 2.0 / H], np.float32)
        x0 = (2.0 * jj / W - 1).astype(np.float32); y0 = (2.0 * ii / H - 1).astype(np.float32)
        # [sample][pixel]
        x = x0[None, :] + jit[:, 0:1]; y = -(y0[None, :] + jit[:, 1:2])
        d = (front[None, None, :] + x[:, :, None] * right[None, None, :] + y[:, :, None] * up[None, None, :]).astype(np.float32)
        sc = api.Scene(fr.model, device=0)
        sc.set_option("traversal", int(os.environ.get("TRAVERSAL", "2")))
        line = f"{name:22s} {P * NS / 1e6:.1f} M rays"
        for tag, dd in (("64 pixels of a row", d.reshape(-1, 3)), ("64 samples of a pixel", np.ascontiguousarray(d.transpose(1, 0, 2)).reshape(-1, 3))):
            td = torch.from_numpy(np.ascontiguousarray(dd)).cuda()
            to = torch.from_numpy(np.broadcast_to(eye, dd.shape).astype(np.float32).copy()).cuda()
            out = torch.empty((len(dd), 5), dtype=torch.int32, device="cuda")
            sc.closest_hit(to, td, out)
            sc.profile(True)
            for _ in range(3):
                sc.closest_hit(to, td, out)
            st = sc.profile_read()["closest_hit"]
            sc.profile(False)
            hits = int((out[:, 0] != -1).sum())
            line += f" | {tag}: {st['items'] / st['ms'] / 1e6:6.2f} G rays/s ({hits / len(dd):.2f} hit)"
            del to, td, out
        print(line, flush=True)
        sc.close()


if __name__ == "__main__":
    main()
