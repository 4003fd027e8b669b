"""Reduces the reference's own published render (images/cornell-box_fhd2.png: Cornell box,
1920x1080, 2048 spp, depth 100, 2 light samples -- README.md:38-46) to a small fixture:
8x8 box-filtered, float32 [135,240,3] in [0,1].  Development container only.

    python tests/golden/make_readme_fixture.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/images/cornell-box_fhd2.png"


def main():
    im = np.asarray(Image.open(SRC).convert("RGB"), np.float32) / 255.0
    assert im.shape == (1080, 1920, 3)
    ds = im.reshape(135, 8, 240, 8, 3).mean(axis=(1, 3)).astype(np.float32)
    np.save(os.path.join(HERE, "cornell_readme_ds8.npy"), ds)
    print("written", ds.shape, ds.min(), ds.max())


if __name__ == "__main__":
    main()
