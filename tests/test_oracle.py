"""Pins the CPU oracle: reference-produced vectors, analytic value, independent MD engine."""
import numpy as np
import pytest

from oracle import jk as O
from oracle import md_eri as M


def _row(l, center, prims, ao_loc=0):
    r = np.zeros(12)
    r[:3] = center
    r[3] = ao_loc
    for p, (c, e) in enumerate(prims):
        r[4 + 2 * p], r[5 + 2 * p] = c, e
    r[10], r[11] = len(prims), l
    return r


def test_reference_ssss(kats):
    k = kats["ssss"]
    b = np.array([_row(0, (0, 0, 0), [(1.0, k["alpha"])])])
    vj, vk = O.jk_raw(b, np.ones((1, 1)), [[0, 0, 0, 0]])
    assert abs(vj[0, 0, 0] - k["vj_raw"]) < 1e-13
    assert abs(vk[0, 0, 0] - k["vk_raw"]) < 1e-13
    # epilogue: 4 vj_raw = 2 vk_raw = 2 pi^2.5 / (p q sqrt(p+q)), p = q = 1
    assert abs(4 * vj[0, 0, 0] - 2 * np.pi ** 2.5 / np.sqrt(2.0)) < 1e-13


def test_reference_psps(kats):
    k = kats["psps"]
    b = np.array([_row(0, k["shell0"]["center"], [(1.0, k["shell0"]["alpha"])], 0),
                  _row(1, k["shell1"]["center"], [(1.0, k["shell1"]["alpha"])], 1)])
    dm = (0.1 * (1 + np.arange(16) % 5)).reshape(4, 4)
    vj, vk = O.jk_raw(b, dm, [k["quartet"]])
    np.testing.assert_allclose(vj.ravel()[1:4], k["vj_raw_1_3"], atol=2e-13)
    assert vj.ravel()[0] == 0 and np.all(vj.ravel()[4:] == 0)
    np.testing.assert_allclose(vk.ravel(), k["vk_raw"], atol=2e-13)


CASES = [((0, 0, 0, 0), (3, 2, 1, 3)), ((1, 0, 1, 1), (2, 1, 3, 1)), ((2, 1, 1, 0), (1, 2, 1, 1)),
         ((2, 2, 2, 2), (1, 1, 1, 1)), ((3, 2, 1, 0), (1, 1, 2, 1)), ((3, 3, 0, 0), (1, 1, 1, 1)),
         ((4, 0, 4, 0), (1, 1, 1, 1)), ((4, 3, 2, 1), (1, 1, 1, 1)), ((3, 1, 3, 2), (1, 1, 1, 1)),
         ((4, 4, 1, 0), (1, 1, 1, 1))]


@pytest.mark.parametrize("ls,nps", CASES)
@pytest.mark.parametrize("omega", [0.0, 0.3])
def test_rys_oracle_vs_mcmurchie_davidson(ls, nps, omega):
    rng = np.random.default_rng(sum(ls) * 7 + sum(nps))
    rows = np.array([_row(l, rng.uniform(-1.2, 1.2, 3), [(rng.uniform(0.3, 1.5), rng.uniform(0.2, 3.0)) for _ in range(n)])
                     for l, n in zip(ls, nps)])
    a = O.eri_block(rows, 0, 1, 2, 3, omega)
    b = M.eri_block(rows, 0, 1, 2, 3, omega)
    assert np.abs(a - b).max() <= 2e-14 * max(np.abs(b).max(), 1e-300) + 1e-18


def test_noncanonical_quartets_are_dropped():
    b = np.array([_row(0, (0, 0, 0), [(1.0, 0.5)], 0), _row(0, (0, 0, 1), [(1.0, 0.7)], 1)])
    dm = np.ones((2, 2))
    for q in ([0, 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, 1]):     # i<j, k>i, l>k  (1q1t.cu:92-94)
        vj, vk = O.jk_raw(b, dm, [q])
        assert not vj.any() and not vk.any()


def test_schwarz_matches_diagonal_blocks():
    rng = np.random.default_rng(5)
    rows = np.array([_row(l, rng.uniform(-1, 1, 3), [(1.0, rng.uniform(0.3, 2.0))]) for l in (0, 1, 2)])
    q = O.schwarz(rows)
    for i in range(3):
        for j in range(3):
            blk = M.eri_block(rows, i, j, i, j)
            d = np.array([[abs(blk[a, b, a, b]) for b in range(blk.shape[1])] for a in range(blk.shape[0])])
            assert abs(q[i, j] - np.sqrt(d.max())) < 1e-12
