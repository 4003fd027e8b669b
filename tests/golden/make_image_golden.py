#!/usr/bin/env python3
"""Golden vectors for the host layer's image readers: image FILES (bytes) and the texels the
reference's own decoder (its vendored stb_image, through oracle/_ref) returns for them.
Run in the development container (needs /root/reference for `make -C oracle ref`):

    python tests/golden/make_image_golden.py      # writes tests/golden/images_stb.npz

The files are synthetic (Pillow-encoded noise and gradients: every chroma subsampling, greyscale,
restart intervals, optimised Huffman tables, odd and tiny sizes; every PNG kind) plus the
reference's small view_test.jpg."""
import ctypes as C
import io
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import bindings as ob  # noqa: E402


def ref_decode(data: bytes, suffix: str) -> np.ndarray:
    L = ob.ref_lib()
    L.ref_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    path = f"/tmp/_golden_image{suffix}"
    open(path, "wb").write(data)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    buf = np.zeros(2048 * 2048 * 4, np.uint8)
    assert L.ref_load_image(path.encode(), C.byref(w), C.byref(h), C.byref(c), buf.ctypes.data, buf.size) == 0, suffix
    return buf[: w.value * h.value * c.value].reshape(h.value, w.value, c.value).copy()


def main():
    rng = np.random.default_rng(11)
    out = {}

    def picture(w, h, c):
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([(x * 255 // max(1, w - 1)), (y * 255 // max(1, h - 1)), ((x + y) * 7 % 256)], 2)[:, :, :c]
        noise = rng.integers(-40, 40, (h, w, c))
        return np.clip(base + noise, 0, 255).astype(np.uint8)

    def jpeg(name, w, h, c=3, **kw):
        im = Image.fromarray(picture(w, h, c) if c == 3 else picture(w, h, 1)[:, :, 0], "RGB" if c == 3 else "L")
        b = io.BytesIO()
        im.save(b, "JPEG", **kw)
        out[f"jpg_{name}_file"] = np.frombuffer(b.getvalue(), np.uint8)
        out[f"jpg_{name}_texels"] = ref_decode(b.getvalue(), ".jpg")

    for sub, tag in ((0, "444"), (1, "422"), (2, "420")):
        jpeg(f"{tag}_64x48", 64, 48, subsampling=sub, quality=85)
        jpeg(f"{tag}_37x23", 37, 23, subsampling=sub, quality=60)
        jpeg(f"{tag}_1x1", 1, 1, subsampling=sub)
        jpeg(f"{tag}_2x9", 2, 9, subsampling=sub, quality=95)
        jpeg(f"{tag}_17x1", 17, 1, subsampling=sub)
    jpeg("grey_50x31", 50, 31, c=1, quality=75)
    jpeg("grey_8x8", 8, 8, c=1)
    jpeg("opt_420_40x40", 40, 40, subsampling=2, optimize=True, quality=30)
    jpeg("q100_444_33x33", 33, 33, subsampling=0, quality=100)
    jpeg("q5_420_64x64", 64, 64, subsampling=2, quality=5)
    for rows in (1, 2):
        try:
            jpeg(f"restart_rows{rows}_420_48x40", 48, 40, subsampling=2, restart_marker_rows=rows)
            jpeg(f"restart_blocks{rows + 2}_444_48x40", 48, 40, subsampling=0, restart_marker_blocks=rows + 2)
        except TypeError:
            pass
    ref_jpg = "/root/reference/assets/view_test.jpg"
    if os.path.exists(ref_jpg):
        data = open(ref_jpg, "rb").read()
        out["jpg_view_test_file"] = np.frombuffer(data, np.uint8)
        out["jpg_view_test_texels"] = ref_decode(data, ".jpg")

    def png(name, im, **kw):
        b = io.BytesIO()
        im.save(b, "PNG", **kw)
        out[f"png_{name}_file"] = np.frombuffer(b.getvalue(), np.uint8)
        out[f"png_{name}_texels"] = ref_decode(b.getvalue(), ".png")

    w, h = 37, 23
    png("L", Image.fromarray(picture(w, h, 1)[:, :, 0], "L"))
    png("LA", Image.fromarray(np.concatenate([picture(w, h, 1), picture(w, h, 1)], 2), "LA"))
    png("RGB", Image.fromarray(picture(w, h, 3), "RGB"))
    png("RGBA", Image.fromarray(np.concatenate([picture(w, h, 3), picture(w, h, 1)], 2), "RGBA"))
    png("P200", Image.fromarray(picture(w, h, 3), "RGB").quantize(200))
    png("P200_trns", Image.fromarray(picture(w, h, 3), "RGB").quantize(200), transparency=bytes(rng.integers(0, 256, 200, dtype=np.uint8)))
    png("P16", Image.fromarray(picture(w, h, 3), "RGB").quantize(16))
    png("P4", Image.fromarray(picture(w, h, 3), "RGB").quantize(4))
    png("P2", Image.fromarray(picture(w, h, 3), "RGB").quantize(2))
    png("bit1", Image.fromarray(picture(w, h, 1)[:, :, 0] > 127))
    png("I16", Image.fromarray(rng.integers(0, 65536, (h, w)).astype(np.uint16)))
    png("L_key", Image.fromarray(picture(w, h, 1)[:, :, 0] // 32 * 32, "L"), transparency=64)
    png("RGB_key", Image.fromarray(picture(w, h, 3) // 64 * 64, "RGB"), transparency=(64, 64, 128))
    np.savez_compressed(os.path.join(HERE, "images_stb.npz"), **out)
    print(len(out) // 2, "images,", os.path.getsize(os.path.join(HERE, "images_stb.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
