import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU time; only runs when RT_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("RT_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="set RT_SLOW=1 to run")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def path_scene():
    from raytracing_amd.scene import FlatScene
    return FlatScene.load_npz(os.path.join(GOLDEN, "cornellBox_path.npz"))


@pytest.fixture(scope="session")
def whitted_scene():
    from raytracing_amd.scene import FlatScene
    return FlatScene.load_npz(os.path.join(GOLDEN, "cornellBox_whitted.npz"))


@pytest.fixture(scope="session")
def sphere_scene():
    """scenes/cornellBox-PathTracing/cornell.json: the cornellBox meshes plus a mirror sphere and an emissive sphere."""
    from raytracing_amd.scene import FlatScene
    return FlatScene.load_npz(os.path.join(GOLDEN, "cornell_spheres.npz"))


@pytest.fixture(scope="session")
def reference_png():
    from raytracing_amd.pngio import read_png
    return read_png(os.path.join(GOLDEN, "reference_cornellBox_path_800x800_512spp.png"))[..., :3]


@pytest.fixture(scope="session")
def gpu_ctx():
    from raytracing_amd.device import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
