import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ASSETS = os.path.join(ROOT, "tests", "golden", "assets")
GOLDEN = os.path.join(ROOT, "tests", "golden")

SCENES = {
    "cornell": "cornell-box/cornell-box.obj",
    "view_test": "view_test.obj",
    "hindu": "hindu/hindu.obj",
    "nanosuit": "nanosuit2/nanosuit2.obj",
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name: str) -> str:
    return os.path.join(ASSETS, SCENES[name])


_models = {}


def load_model(name: str):
    """host.Model of a named test scene ('soup' is the synthetic triangle soup; 'nanosuit@sah' and 'nanosuit@middle'
    build the trees with the reference's other split strategies)."""
    from orion_amd import host
    if name not in _models:
        if "@" in name:
            base, how = name.split("@")
            _models[name] = host.load_model(scene_path(base), strategy={"median": host.MEDIAN, "middle": host.MIDDLE, "sah": host.SAH}[how])
        elif name == "soup":
            import raygen
            v, i = raygen.triangle_soup()
            _models[name] = host.model_from_arrays(v, i)
        elif name.startswith("soup") and name[4:].isdigit():      # "soup20": 2^20 triangles at the same density of coverage
            import raygen
            n = 1 << int(name[4:])
            v, i = raygen.triangle_soup(n, edge=0.02 * (19058 / n) ** (1 / 3), seed=2)
            _models[name] = host.model_from_arrays(v, i)
        else:
            _models[name] = host.load_model(scene_path(name))
    return _models[name]


@pytest.fixture(scope="session")
def models():
    return load_model
