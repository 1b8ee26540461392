import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    """Directory with small stand-in OBJ files (low subdivision levels keep CPU tests fast)."""
    from rust_raytracer_amd import meshes
    d = tmp_path_factory.mktemp("models")
    return meshes.ensure_model_dir(str(d), levels={"bunny.obj": 4, "santa.obj": 4, "sphere_high.obj": 3,
                                                   "sphere_low.obj": 1})


@pytest.fixture(scope="session")
def scenes_dir():
    return os.path.join(ROOT, "scenes")
