#!/usr/bin/env python3
"""One Cornell FHD render with a given number of light samples, for counter passes that split the
shade kernel's instruction count into its per-vertex and per-light-sample parts.
Usage: shade_cost.py <light_samples> [spp]; prints vertices shaded and rays."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    L = int(sys.argv[1]); spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import torch
    import bench
    from orion_amd import api, host
    model = host.load_model(os.path.join(ROOT, "tests", "golden", "assets", "cornell-box", "cornell-box.obj"))
    eye, look, up, yv = bench.CAMERA
    cam = api.camera_vectors(eye, look, up, yv, bench.W / bench.H)
    sc = api.Scene(model, device=0)
    sc.set_option("tail_paths", 0)
    accum = torch.zeros((bench.H, bench.W, 4), dtype=torch.float32, device="cuda")
    sc.profile(True)
    sc.render(cam, bench.W, bench.H, 0, spp, max(L, 1), bench.MAX_DEPTH, seed=1, accum=accum)
    st = sc.profile_read()
    print(json.dumps({"L": L, "vertices": st["shade"]["items"], "extend_rays": st["closest_hit"]["items"], "shade_ms": st["shade"]["ms"]}))
    sc.close()


if __name__ == "__main__":
    main()
