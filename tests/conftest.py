import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


H2O = "O 0.0 0.0 0.1174; H -0.757 0.0 -0.4696; H 0.757 0.0 -0.4696"
H2_BOHR = "H -0.757 4. -0.4696; H 0.757 4. -0.4696"


def benzene_atoms():
    import numpy as np
    rc, rh = 1.39, 1.39 + 1.09
    out = []
    for k in range(6):
        t = np.pi / 3 * k
        out.append(("C", (rc * np.cos(t), rc * np.sin(t), 0.0)))
        out.append(("H", (rh * np.cos(t), rh * np.sin(t), 0.0)))
    return out
