import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_addoption(parser):
    parser.addoption("--rt-lib", default="", help="another build of librt_amd.so to test (e.g. a guarded variant, "
                                                   "raytracer_amd/variants/guard.so built with -DRT_GUARD=1)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    lib = config.getoption("--rt-lib")
    if lib:      # read by raytracer_amd when it loads the library; the host binary keeps its own link
        os.environ["RT_AMD_LIB"] = os.path.abspath(lib)


@pytest.fixture(scope="session")
def golden_scenes():
    import numpy as np
    return np.load(os.path.join(HERE, "golden", "ref_scenes.npz"))


@pytest.fixture(scope="session")
def golden_stock():
    import numpy as np
    return np.load(os.path.join(HERE, "golden", "stock_800x600_d3.npz"))
