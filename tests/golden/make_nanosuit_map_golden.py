#!/usr/bin/env python3
"""Fingerprints of the texels the reference's own decoder (its vendored stb_image, through oracle/_ref)
returns for the nanosuit colour maps shipped under tests/golden/assets/nanosuit2/ (data files of the
reference: assets/nanosuit2/*_dif.png).  Run in the development container:

    python tests/golden/make_nanosuit_map_golden.py      # writes tests/golden/nanosuit_maps_stb.json
"""
import ctypes as C
import glob
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import bindings as ob  # noqa: E402


def main():
    L = ob.ref_lib()
    L.ref_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    out = {}
    for path in sorted(glob.glob(os.path.join(HERE, "assets", "nanosuit2", "*.png"))):
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        buf = np.zeros(2048 * 2048 * 4, np.uint8)
        assert L.ref_load_image(path.encode(), C.byref(w), C.byref(h), C.byref(c), buf.ctypes.data, buf.size) == 0, path
        n = w.value * h.value * c.value
        out[os.path.basename(path)] = {"width": w.value, "height": h.value, "channels": c.value,
                                       "sha256": hashlib.sha256(buf[:n].tobytes()).hexdigest()}
    json.dump(out, open(os.path.join(HERE, "nanosuit_maps_stb.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
