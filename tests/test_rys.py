"""Rys tables vs the committed mpmath golden vectors (tests/golden/rys_mpmath.json)."""
import json
import os

import numpy as np

from oracle import jk as O

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rys_tables_match_mpmath():
    with open(os.path.join(HERE, "golden", "rys_mpmath.json")) as f:
        vecs = json.load(f)
    worst = 0.0
    for v in vecs:
        r, w = O.rys_roots(v["n"], v["x"])
        rr, ww = np.array(v["roots"]), np.array(v["weights"])
        worst = max(worst, np.abs(r - rr).max() / rr.max(), np.abs(w - ww).max() / ww.max())
        np.testing.assert_allclose(r, rr, rtol=2e-13, atol=0)
        np.testing.assert_allclose(w, ww, rtol=0, atol=2e-13 * ww.max())
    assert worst < 1e-13


def test_weights_sum_to_boys_f0():
    from scipy.special import erf
    for n in range(1, 10):
        for x in (0.3, 4.0, 17.0, 44.0, 90.0):
            _, w = O.rys_roots(n, x)
            f0 = 0.5 * np.sqrt(np.pi / x) * erf(np.sqrt(x))
            assert abs(w.sum() - f0) < 2e-14


def test_long_range_scaling():
    # erf-attenuated kernel: roots scale by w^2/(w^2+theta), weights by its square root (rys_roots.cu:42-47)
    n, x, theta, omega = 3, 7.0, 1.3, 0.4
    tf = omega ** 2 / (omega ** 2 + theta)
    r0, w0 = O.rys_roots(n, x * tf, theta)
    r1, w1 = O.rys_roots(n, x, theta, omega)
    np.testing.assert_allclose(r1, r0 * tf, rtol=1e-14)
    np.testing.assert_allclose(w1, w0 * np.sqrt(tf), rtol=1e-14)
