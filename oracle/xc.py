"""Closed-form exchange-correlation functionals for spin-restricted densities (NumPy).  TEST INFRASTRUCTURE ONLY.

The reference evaluates functionals through libxc (third party, ``ni.eval_xc_eff``, /root/reference/jqc/pyscf/rks.py:341);
libxc is not in this image, so the functionals whose energies the reference's own tests hold
(/root/reference/jqc/pyscf/tests/test_dft.py:75-86: "LDA,vwn5" -75.9046410402 and "PBE" -76.3800182418 for H2O /
def2-TZVPP) are restated here from their published closed forms, with libxc's constants:

  * Slater/Dirac exchange                       e_x = -3/4 (3/pi)^(1/3) rho^(1/3)
  * VWN5 correlation, paramagnetic branch       Vosko, Wilk, Nusair, Can. J. Phys. 58, 1200 (1980), eq. 4.4 with
                                                A = 0.0310907, x0 = -0.10498, b = 3.72744, c = 12.9352   (libxc LDA_C_VWN)
  * PBE exchange                                Perdew, Burke, Ernzerhof, PRL 77, 3865 (1996): F_x = 1 + kappa - kappa /
                                                (1 + mu s^2 / kappa), kappa = 0.804, mu = beta pi^2 / 3
  * PBE correlation                             e_c^PW92 + H(rs, t); beta = 0.06672455060314922, gamma = (1 - ln 2) / pi^2,
                                                PW92 in libxc's "modified" parametrisation (LDA_C_PW_MOD: A = 0.0310906908696549)

Only zeta = 0 is needed (closed shells).  The potentials are obtained by COMPLEX-STEP differentiation of rho * e_xc:
d f / d x = Im f(x + i h) / h with h = 1e-30 is exact to rounding for the analytic expressions above, so no derivative
formula is written down (and none can be wrong).  ``eval_xc_eff`` returns what PySCF's ``NumInt.eval_xc_eff`` returns for
deriv = 1: ``exc[ngrids]`` (energy per particle) and ``vxc[nvar, ngrids]`` = d(rho e_xc) / d(rho, grad rho) with nvar = 1
(LDA) or 4 (GGA; components 1..3 = 2 v_sigma grad rho).
"""
import numpy as np

_H = 1e-30


def _rs(rho):
    return (3.0 / (4.0 * np.pi * rho)) ** (1.0 / 3.0)


def slater_x(rho):
    return -0.75 * (3.0 / np.pi) ** (1.0 / 3.0) * rho ** (1.0 / 3.0)


def vwn5_c(rho):
    A, x0, b, c = 0.0310907, -0.10498, 3.72744, 12.9352
    x = np.sqrt(_rs(rho))
    X = lambda y: y * y + b * y + c
    Q = np.sqrt(4.0 * c - b * b)
    at = np.arctan(Q / (2.0 * x + b))
    return A * (np.log(x * x / X(x)) + 2.0 * b / Q * at
                - b * x0 / X(x0) * (np.log((x - x0) ** 2 / X(x)) + 2.0 * (b + 2.0 * x0) / Q * at))


def pw92_c(rho, A=0.0310906908696549):
    a1, b1, b2, b3, b4 = 0.21370, 7.5957, 3.5876, 1.6382, 0.49294
    rs = _rs(rho)
    s = np.sqrt(rs)
    return -2.0 * A * (1.0 + a1 * rs) * np.log(1.0 + 1.0 / (2.0 * A * (b1 * s + b2 * rs + b3 * rs * s + b4 * rs * rs)))


_BETA = 0.06672455060314922
_GAMMA = (1.0 - np.log(2.0)) / np.pi ** 2


def pbe_x(rho, sigma):
    kappa, mu = 0.804, _BETA * np.pi ** 2 / 3.0
    kf = (3.0 * np.pi ** 2 * rho) ** (1.0 / 3.0)
    s2 = sigma / (2.0 * kf * rho) ** 2
    return slater_x(rho) * (1.0 + kappa - kappa / (1.0 + mu * s2 / kappa))


def pbe_c(rho, sigma):
    ec = pw92_c(rho)
    kf = (3.0 * np.pi ** 2 * rho) ** (1.0 / 3.0)
    ks2 = 4.0 * kf / np.pi
    t2 = sigma / (4.0 * ks2 * rho * rho)
    Aa = _BETA / _GAMMA / (np.exp(-ec / _GAMMA) - 1.0)
    At2 = Aa * t2
    return ec + _GAMMA * np.log(1.0 + _BETA / _GAMMA * t2 * (1.0 + At2) / (1.0 + At2 + At2 * At2))


# name -> (type, e_xc(rho[, sigma]))
FUNCTIONALS = {
    "slater": ("LDA", lambda r: slater_x(r)),
    "lda,vwn5": ("LDA", lambda r: slater_x(r) + vwn5_c(r)),
    "pbe": ("GGA", lambda r, s: pbe_x(r, s) + pbe_c(r, s)),
}


def xc_type(xc_code):
    return FUNCTIONALS[xc_code.lower().replace(" ", "")][0]


def eval_xc_eff(xc_code, rho, rho_floor=1e-14):
    """(exc[ngrids], vxc[nvar, ngrids]) for rho[ngrids] (LDA) or rho[4, ngrids] (GGA); points below ``rho_floor`` give zero
    (libxc's density threshold has the same role)."""
    kind, f = FUNCTIONALS[xc_code.lower().replace(" ", "")]
    rho = np.asarray(rho, dtype=np.float64)
    r = rho if rho.ndim == 1 else rho[0]
    ok = r > rho_floor
    rr = np.where(ok, r, 1.0)
    if kind == "LDA":
        exc = np.where(ok, f(rr), 0.0)
        v = np.where(ok, np.imag((rr + 1j * _H) * f(rr + 1j * _H)) / _H, 0.0)
        return exc, v.reshape(1, -1)
    g = rho[1:4]
    sg = np.where(ok, (g * g).sum(axis=0), 0.0)
    exc = np.where(ok, f(rr, sg), 0.0)
    vrho = np.imag((rr + 1j * _H) * f(rr + 1j * _H, sg)) / _H
    vsig = np.imag(rr * f(rr, sg + 1j * _H)) / _H
    vxc = np.zeros((4, r.size))
    vxc[0] = np.where(ok, vrho, 0.0)
    vxc[1:4] = np.where(ok, 2.0 * vsig, 0.0) * g
    return exc, vxc
