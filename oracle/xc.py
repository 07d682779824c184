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

  * B88 exchange                                Becke, PRA 38, 3098 (1988), beta = 0.0042 (per spin channel, closed shell: rho_s = rho / 2)
  * LYP correlation                             Lee, Yang, Parr, PRB 37, 785 (1988) in the Laplacian-free form of Miehlich, Savin,
                                                Stoll, Preuss, CPL 157, 200 (1989), closed shell; a, b, c, d = 0.04918, 0.132,
                                                0.2533, 0.349
  * VWN-RPA correlation                         VWN eq. 4.4 with the RPA fit: A = 0.0310907, x0 = -0.409286, b = 13.0720,
                                                c = 42.7198  (libxc LDA_C_VWN_RPA)
  * B3LYP                                       0.2 HF + 0.08 Slater + 0.72 B88 + 0.81 LYP + 0.19 VWN-RPA (libxc HYB_GGA_XC_B3LYP,
                                                what PySCF's "b3lyp" is): reference energies -76.4666495594 (spherical) and
                                                -76.4672144985 (Cartesian), tests/test_dft.py:87-92,110-114
  * omega-B97                                   Chai, Head-Gordon, JCP 128, 084106 (2008): long-range HF exchange (erf, omega = 0.4) +
                                                short-range B97 exchange [LSDA exchange x attenuation F(omega / 2 k_F) x power
                                                series in u = gamma s^2 / (1 + gamma s^2)] + B97 correlation (Stoll-partitioned
                                                PW92, same-spin and opposite-spin series); libxc HYB_GGA_XC_WB97, reference energy
                                                -76.4486274326 (tests/test_dft.py:99-103)
  * omega-B97M-V                                Mardirossian, Head-Gordon, JCP 144, 214110 (2016), Table III: range-separated
                                                hybrid meta-GGA (15 % short-range, 100 % long-range exact exchange, omega = 0.3)
                                                + VV10 (b = 6.0, C = 0.01).  Same skeleton as omega-B97 above with a two-variable
                                                power series per channel, sum_ij c_ij w^i u^j, in u = gamma s^2 / (1 + gamma s^2)
                                                and the kinetic-energy variable w = (t - 1) / (t + 1), t = tau_UEG / tau
                                                (per spin channel; opposite spin: averages of s^2 and t).  libxc
                                                HYB_MGGA_XC_WB97M_V; reference energy -76.4334218842 (tests/test_dft.py:105-109) --
                                                the number that pins the tau branch of rho / V_xc AND VV10 (nr_nlc_vxc).
Only zeta = 0 is needed (closed shells).  The potentials are obtained by COMPLEX-STEP differentiation of rho * e_xc:
d f / d x = Im f(x + i h) / h with h = 1e-30 is exact to rounding for the analytic expressions above, so no derivative
formula is written down (and none can be wrong).  ``eval_xc_eff`` returns what PySCF's ``NumInt.eval_xc_eff`` returns for
deriv = 1: ``exc[ngrids]`` (energy per particle) and ``vxc[nvar, ngrids]`` = d(rho e_xc) / d(rho, grad rho) with nvar = 1
(LDA), 4 (GGA; components 1..3 = 2 v_sigma grad rho) or 5 (meta-GGA; component 4 = d / d tau, tau = 1/2 sum |grad psi|^2).
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


def vwn_rpa_c(rho):
    A, x0, b, c = 0.0310907, -0.409286, 13.0720, 42.7198
    x = np.sqrt(_rs(rho))
    X = lambda y: y * y + b * y + c
    Q = np.sqrt(4.0 * c - b * b)
    at = np.arctan(Q / (2.0 * x + b))
    return A * (np.log(x * x / X(x)) + 2.0 * b / Q * at
                - b * x0 / X(x0) * (np.log((x - x0) ** 2 / X(x)) + 2.0 * (b + 2.0 * x0) / Q * at))


def b88_x(rho, sigma):
    """Becke 88 exchange, energy per particle, closed shell (two equal spin channels)."""
    beta = 0.0042
    rs_ = 0.5 * rho                                             # spin density
    x = np.sqrt(0.25 * sigma) / rs_ ** (4.0 / 3.0)
    cx = 1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3.0)             # LSDA exchange per spin: -cx rho_s^(4/3)
    ex_spin = -rs_ ** (4.0 / 3.0) * (cx + beta * x * x / (1.0 + 6.0 * beta * x * np.arcsinh(x)))
    return 2.0 * ex_spin / rho


def lyp_c(rho, sigma):
    """LYP correlation, energy per particle, closed shell: -a / (1 + d rho^-1/3) [1 + b C_F exp(-c rho^-1/3) ...] with the
    gradient term + a b omega rho^2 sigma (3 + 7 delta) / 72 (derivation: Miehlich et al. with rho_a = rho_b)."""
    a, b, c, d = 0.04918, 0.132, 0.2533, 0.349
    cf = 0.3 * (3.0 * np.pi ** 2) ** (2.0 / 3.0)
    r13 = rho ** (-1.0 / 3.0)
    den = 1.0 + d * r13
    omega = np.exp(-c * r13) / den * rho ** (-11.0 / 3.0)
    delta = c * r13 + d * r13 / den
    e = -a * rho / den - a * b * omega * (cf * rho ** (14.0 / 3.0) - rho * rho * sigma * (1.0 / 24.0 + 7.0 * delta / 72.0))
    return e / rho


def pw92_c(rho, A=0.0310906908696549):
    a1, b1, b2, b3, b4 = 0.21370, 7.5957, 3.5876, 1.6382, 0.49294
    rs = _rs(rho)
    s = np.sqrt(rs)
    return -2.0 * A * (1.0 + a1 * rs) * np.log(1.0 + 1.0 / (2.0 * A * (b1 * s + b2 * rs + b3 * rs * s + b4 * rs * rs)))


def pw92_c_ferro(rho):
    """PW92 correlation energy per particle of the FULLY polarised gas (zeta = 1) of total density ``rho``."""
    A, a1, b1, b2, b3, b4 = 0.0155453454348274, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517
    rs = _rs(rho)
    s = np.sqrt(rs)
    return -2.0 * A * (1.0 + a1 * rs) * np.log(1.0 + 1.0 / (2.0 * A * (b1 * s + b2 * rs + b3 * rs * s + b4 * rs * rs)))


def _attenuation(a):
    """F(a) of the erfc-attenuated LSDA exchange, a = omega / (2 k_F); series beyond a = 5 (the closed form cancels there)."""
    from scipy.special import erf
    a_small = np.where(np.real(a) < 5.0, a, 1.0)
    f = 1.0 - 8.0 / 3.0 * a_small * (np.sqrt(np.pi) * erf(0.5 / a_small) - 3.0 * a_small + 4.0 * a_small ** 3
                                     + (2.0 * a_small - 4.0 * a_small ** 3) * np.exp(-0.25 / (a_small * a_small)))
    a_big = np.where(np.real(a) < 5.0, 5.0, a)
    i2 = 1.0 / (a_big * a_big)
    ser = i2 * (1.0 / 36.0 - i2 * (1.0 / 960.0 - i2 * (1.0 / 26880.0 - i2 / 829440.0)))
    return np.where(np.real(a) < 5.0, f, ser)


def wb97_xc(rho, sigma, omega=0.4):
    """Semilocal part of omega-B97, energy per particle, closed shell."""
    cx = (1.00000, 1.13116, -2.74915, 12.0900, -5.71642)
    css = (1.00000, -2.55352, 11.8926, -26.9452, 17.0927)
    cab = (1.00000, 3.99051, -17.0066, 1.07292, 8.88211)
    gx, gss, gab = 0.004, 0.2, 0.006
    rs_ = 0.5 * rho                                             # spin density
    s2 = 0.25 * sigma / rs_ ** (8.0 / 3.0)                      # s_sigma^2 = |grad rho_s|^2 / rho_s^(8/3)

    def series(c, g, x2):
        u = g * x2 / (1.0 + g * x2)
        return c[0] + u * (c[1] + u * (c[2] + u * (c[3] + u * c[4])))
    # short-range exchange, both spin channels
    kf = (6.0 * np.pi ** 2 * rs_) ** (1.0 / 3.0)
    ex_lsda = -1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3.0) * rs_ ** (4.0 / 3.0)
    ex = 2.0 * ex_lsda * _attenuation(omega / (2.0 * kf)) * series(cx, gx, s2)
    # correlation: Stoll partition of PW92
    ec_ss = rs_ * pw92_c_ferro(rs_)                            # one spin channel alone (energy density)
    ec_ab = rho * pw92_c(rho) - 2.0 * ec_ss
    ec = 2.0 * ec_ss * series(css, gss, s2) + ec_ab * series(cab, gab, s2)      # (s_av^2 = s_sigma^2 for a closed shell)
    return (ex + ec) / rho


def wb97mv_xc(rho, sigma, tau, omega=0.3):
    """Semilocal part of omega-B97M-V, energy per particle, closed shell (tau = 1/2 sum_i |grad psi_i|^2 over both spins)."""
    # (coefficient, power of w, power of u)
    cx = ((0.85, 0, 0), (1.007, 0, 1), (0.259, 1, 0))
    css = ((0.443, 0, 0), (-1.437, 0, 4), (-4.535, 1, 0), (-3.39, 2, 0), (4.278, 4, 3))
    cos_ = ((1.000, 0, 0), (1.358, 1, 0), (2.924, 2, 0), (-8.812, 2, 1), (-1.39, 6, 0), (9.142, 6, 1))
    gx, gss, gos = 0.004, 0.2, 0.006
    rs_ = 0.5 * rho                                             # spin density
    s2 = 0.25 * sigma / rs_ ** (8.0 / 3.0)                      # |grad rho_s|^2 / rho_s^(8/3)
    ts = 0.5 * tau / rs_ ** (5.0 / 3.0)                         # tau_s / rho_s^(5/3)
    kc = 0.3 * (6.0 * np.pi ** 2) ** (2.0 / 3.0)                # tau_UEG,s / rho_s^(5/3)
    w = (kc - ts) / (kc + ts)                                   # = (t - 1) / (t + 1), t = tau_UEG / tau; closed shell: w_os = w_s

    def series(c, g):
        u = g * s2 / (1.0 + g * s2)
        return sum(cf * w ** i * u ** j for cf, i, j in c)
    kf = (6.0 * np.pi ** 2 * rs_) ** (1.0 / 3.0)
    ex_lsda = -1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3.0) * rs_ ** (4.0 / 3.0)
    ex = 2.0 * ex_lsda * _attenuation(omega / (2.0 * kf)) * series(cx, gx)
    ec_ss = rs_ * pw92_c_ferro(rs_)
    ec_ab = rho * pw92_c(rho) - 2.0 * ec_ss
    ec = 2.0 * ec_ss * series(css, gss) + ec_ab * series(cos_, gos)
    return (ex + ec) / rho


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
    "b3lyp": ("GGA", lambda r, s: 0.08 * slater_x(r) + 0.72 * b88_x(r, s) + 0.81 * lyp_c(r, s) + 0.19 * vwn_rpa_c(r)),
    "wb97": ("GGA", lambda r, s: wb97_xc(r, s)),
    "wb97m-v": ("MGGA", lambda r, s, t: wb97mv_xc(r, s, t)),
}
_ALIASES = {"hyb_gga_xc_wb97": "wb97", "hyb_mgga_xc_wb97m_v": "wb97m-v", "wb97m_v": "wb97m-v", "wb97mv": "wb97m-v"}


def _key(xc_code):
    k = xc_code.lower().replace(" ", "")
    return _ALIASES.get(k, k)

# (omega, alpha, hyb) as pyscf's ni.rsh_and_hybrid_coeff returns them: long-range HF fraction alpha at range separation omega,
# short-range / global HF fraction hyb
HYBRID = {"b3lyp": (0.0, 0.0, 0.2), "wb97": (0.4, 1.0, 0.0), "wb97m-v": (0.3, 1.0, 0.15)}
# VV10 parameters ((b, C), factor) as pyscf's ni.nlc_coeff returns them
NLC = {"wb97m-v": (((6.0, 0.01), 1.0),)}


def rsh_and_hybrid_coeff(xc_code):
    return HYBRID.get(_key(xc_code), (0.0, 0.0, 0.0))


def nlc_coeff(xc_code):
    return NLC.get(_key(xc_code), ())


def xc_type(xc_code):
    return FUNCTIONALS[_key(xc_code)][0]


def eval_xc_eff(xc_code, rho, rho_floor=1e-14):
    """(exc[ngrids], vxc[nvar, ngrids]) for rho[ngrids] (LDA) or rho[4, ngrids] (GGA); points below ``rho_floor`` give zero
    (libxc's density threshold has the same role)."""
    kind, f = FUNCTIONALS[_key(xc_code)]
    rho = np.asarray(rho, dtype=np.float64)
    r = rho if rho.ndim == 1 else rho[0]
    ok = r > rho_floor
    rr = np.where(ok, r, 1.0)
    if kind == "LDA":
        exc = np.where(ok, f(rr), 0.0)
        v = np.where(ok, np.imag((rr + 1j * _H) * f(rr + 1j * _H)) / _H, 0.0)
        return exc, v.reshape(1, -1)
    g = rho[1:4]
    sg = np.where(ok, np.maximum((g * g).sum(axis=0), 1e-40), 1e-40)     # (B88 is a function of sqrt(sigma): keep the step off the branch point)
    if kind == "MGGA":
        # tau >= tau_W = sigma / (8 rho) for any N-representable density; the floor only keeps dead points finite
        tt = np.where(ok, np.maximum(rho[4], 1e-40), 1.0)
        exc = np.where(ok, f(rr, sg, tt), 0.0)
        vxc = np.zeros((5, r.size))
        vxc[0] = np.where(ok, np.imag((rr + 1j * _H) * f(rr + 1j * _H, sg, tt)) / _H, 0.0)
        vxc[1:4] = np.where(ok, 2.0 * np.imag(rr * f(rr, sg + 1j * _H, tt)) / _H, 0.0) * g
        vxc[4] = np.where(ok, np.imag(rr * f(rr, sg, tt + 1j * _H)) / _H, 0.0)
        return exc, vxc
    exc = np.where(ok, f(rr, sg), 0.0)
    vrho = np.imag((rr + 1j * _H) * f(rr + 1j * _H, sg)) / _H
    vsig = np.imag(rr * f(rr, sg + 1j * _H)) / _H
    vxc = np.zeros((4, r.size))
    vxc[0] = np.where(ok, vrho, 0.0)
    vxc[1:4] = np.where(ok, 2.0 * vsig, 0.0) * g
    return exc, vxc
