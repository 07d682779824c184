"""Cartesian -> real-spherical transformation coefficients in PySCF/libcint conventions.

Conventions reproduced (they are what ``mol.cart2sph_coeff`` and the reference's
``jqc/backend/common/cart2sph.cu`` tables encode; cf. ``jqc/backend/tests/test_cart2sph.py:61-109``
which checks against ``gto.mole.cart2sph(l, normalized="sp")``):

  * Cartesian components of a shell are ordered lx descending, then ly descending
    (xx, xy, xz, yy, yz, zz); this is also the order used inside every kernel
    (``/root/reference/jqc/backend/util.py:21-36``).
  * l = 0, 1: identity (p functions stay in x, y, z order); the factors sqrt(1/4pi), sqrt(3/4pi)
    are folded into the contraction coefficients instead (``jqc/pyscf/basis.py:549-553``).
  * l >= 2: rows are r^l Y_lm (real, m = -l..l, no Condon-Shortley phase), i.e. the radial part
    carries the whole normalisation.

The coefficients are derived here from the closed form of the real solid harmonics
(Helgaker, Jorgensen, Olsen, "Molecular Electronic-Structure Theory", eq. 6.4.47); they are not
copied from the reference's tables.
"""
from functools import lru_cache
from math import comb, factorial, pi, sqrt

import numpy as np


def cart_powers(l):
    """[(lx, ly, lz)] in libcint order."""
    return [(lx, ly, l - lx - ly) for lx in range(l, -1, -1) for ly in range(l - lx, -1, -1)]


def ncart(l):
    return (l + 1) * (l + 2) // 2


@lru_cache(maxsize=None)
def cart2sph_l(l):
    """Matrix C[ncart, nsph] such that  phi_sph = phi_cart @ C."""
    nc = ncart(l)
    if l == 0:
        return np.ones((1, 1))
    if l == 1:
        return np.eye(3)
    powers = {p: i for i, p in enumerate(cart_powers(l))}
    out = np.zeros((nc, 2 * l + 1))
    for m in range(-l, l + 1):
        am = abs(m)
        nlm = sqrt(2.0 * factorial(l + am) * factorial(l - am) / (2.0 if m == 0 else 1.0)) / (2 ** am * factorial(l))
        two_vm = 0 if m >= 0 else 1  # 2*v_m
        for t in range((l - am) // 2 + 1):
            for u in range(t + 1):
                # v runs over v_m, v_m+1, ... <= floor(|m|/2 - v_m) + v_m
                nv = int((am / 2.0 - two_vm / 2.0) // 1)
                for iv in range(nv + 1):
                    two_v = 2 * iv + two_vm
                    sign = (-1) ** (t + iv)
                    c = sign * 0.25 ** t * comb(l, t) * comb(l - t, am + t) * comb(t, u) * comb(am, two_v)
                    lx = 2 * t + am - 2 * u - two_v
                    ly = 2 * u + two_v
                    lz = l - 2 * t - am
                    out[powers[(lx, ly, lz)], m + l] += nlm * c
    return out * sqrt((2 * l + 1) / (4 * pi))


def fac_sp(l):
    """libcint's CINTcommon_fac_sp: extra factor carried by s and p coefficients."""
    if l == 0:
        return 0.282094791773878143
    if l == 1:
        return 0.488602511902919921
    return 1.0
