import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="session")
def scenes(golden_dir):
    from pathtracer_amd.scene import Scene

    return {
        "cornell": Scene.load(os.path.join(golden_dir, "cornell.ptxscene")),
        "prims": Scene.load(os.path.join(golden_dir, "prims_scene.npz")),
        "scifi": Scene.load(os.path.join(golden_dir, "scifi_scene.npz")),
        "bubbletea": Scene.load(os.path.join(golden_dir, "bubbletea_scene.npz")),
    }


@pytest.fixture(scope="session")
def oracles(scenes):
    from harness import Oracle

    return {k: Oracle(v) for k, v in scenes.items()}


@pytest.fixture(scope="session")
def gpu(scenes):
    """One Device per scene, uploaded once."""
    import pathtracer_amd as pt

    devs = {}
    for k, sc in scenes.items():
        d = pt.Device(0)
        d.scene_upload(sc)
        devs[k] = d
    yield devs
    for d in devs.values():
        d.close()
