import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_scenes():
    import numpy as np
    return np.load(os.path.join(HERE, "golden", "ref_scenes.npz"))


@pytest.fixture(scope="session")
def golden_stock():
    import numpy as np
    return np.load(os.path.join(HERE, "golden", "stock_800x600_d3.npz"))
