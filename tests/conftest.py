import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: GPU gates beyond the driver-run suite's time budget (run with -m slow on the round's final GPU pass)")


# Order of the GPU suite under ``-x``: the cheap per-class oracle gates first, the full-size BASELINE configurations (minutes each,
# whole SCF runs) last, so that one failing expensive test cannot hide the parity evidence of every other row.
_FILE_ORDER = ["test_jk_gpu", "test_jk_pair_gpu", "test_int1e_gpu", "test_dft_gpu", "test_boundary_gpu", "test_grad_gpu",
               "test_jk_fullsize_gpu", "test_dft_fullsize_gpu", "test_configs_gpu"]
_HEAVY = ("112_atoms", "config3", "config4", "config5", "two_ranks")


def pytest_collection_modifyitems(config, items):
    def key(item):
        mod = item.module.__name__.rsplit(".", 1)[-1]
        rank = _FILE_ORDER.index(mod) if mod in _FILE_ORDER else -1          # CPU files keep their place in front
        return (any(h in item.name for h in _HEAVY) and rank >= 0, rank)
    items.sort(key=key)                                                     # (stable: the order inside a file is kept)


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


# One basis written twice: with general contractions (several coefficient columns per primitive set, 4-5 primitives per
# contraction: the patterns of 6-31G / cc-pVDZ / cc-pVTZ in the reference's tests/test_basis_sets_jk.py) and as the equivalent
# segmented shells.  Same AO set in the same order, so every matrix must agree between the two definitions.
GENERAL_BASIS = {"O": [[0, [12.0, 0.10, -0.03], [2.5, 0.45, -0.12], [0.7, 0.50, 0.30], [0.25, 0.10, 0.60], [0.09, 0.0, 0.35]],
                       [1, [3.0, 0.3, 0.1], [0.8, 0.5, 0.2], [0.2, 0.4, 0.9]], [2, [0.9, 1.0]]],
                 "H": [[0, [5.0, 0.15, 0.0], [1.0, 0.6, 0.1], [0.2, 0.4, 1.0]], [1, [0.7, 1.0]]]}
SEGMENTED_BASIS = {"O": [[0, [12.0, 0.10], [2.5, 0.45], [0.7, 0.50], [0.25, 0.10]],
                         [0, [12.0, -0.03], [2.5, -0.12], [0.7, 0.30], [0.25, 0.60], [0.09, 0.35]],
                         [1, [3.0, 0.3], [0.8, 0.5], [0.2, 0.4]], [1, [3.0, 0.1], [0.8, 0.2], [0.2, 0.9]], [2, [0.9, 1.0]]],
                   "H": [[0, [5.0, 0.15], [1.0, 0.6], [0.2, 0.4]], [0, [1.0, 0.1], [0.2, 1.0]], [1, [0.7, 1.0]]]}
