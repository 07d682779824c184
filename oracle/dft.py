"""CPU restatement of the reference's DFT grid path (NumPy).  TEST INFRASTRUCTURE ONLY.

Follows (file:line under /root/reference):
  * Cartesian GTO values and gradients, PySCF component order  jqc/backend/dft/eval_rho.cu:150-300
      phi = x^lx y^ly z^lz sum_p c_p exp(-a_p r^2);  d/dx: (lx x^(lx-1) - 2 a x^(lx+1)) ...
  * rho / grad rho / tau accumulation                           jqc/backend/dft/eval_rho.cu:300-383
      rho = sum_ab D_ab phi_a phi_b, grad rho = sum_ab D_ab (grad phi_a phi_b + phi_a grad phi_b),
      tau = 1/2 sum_ab D_ab grad phi_a . grad phi_b             (symmetric D, as the kernel assumes)
  * V_xc matrix, conventions of the reference's own tests       jqc/pyscf/tests/test_rks.py:104-107,158-162,185-192
      V_ab = sum_g [w0 phi_a phi_b + (w.grad phi_a) phi_b + phi_a (w.grad phi_b) + 1/2 w4 grad phi_a.grad phi_b]
  * VV10 double sum and its pre/post algebra                    jqc/backend/dft/vv10.cu:29-118, jqc/backend/rks.py:398-715
The dense formulas ARE the reference tests' oracle (they compare against ni.eval_ao + ni.eval_rho and
ao.dot((w ao).T)).  PINNED (tests/test_dft_known_answers.py): an RKS SCF built from this module, oracle/xc.py and
oracle/rks.py reproduces the energies the reference's own tests hold for H2O / def2-TZVPP
(jqc/pyscf/tests/test_dft.py:75-114): "LDA,vwn5" -75.9046410402 to 1e-9 Eh, "PBE" -76.3800182418 to 6e-8 Eh, "B3LYP"
-76.4666495594 (spherical) to 3e-9 Eh and -76.4672144985 (Cartesian), "HYB_GGA_XC_WB97" -76.4486274326 to 6e-8 Eh (LDA and
GGA branches: AO values, gradients, the V_xc conventions; exact exchange of a global hybrid; long-range exchange of a
range-separated hybrid, omega = 0.4).  The AO values are also checked against analytic normalisation
integrals and finite differences (tests/test_dft_oracle.py); the meta-GGA tau term and VV10 have no reference-held number
a closed-form functional could reproduce (M06 / wB97M-V need libxc) and stay pinned by those checks only.
"""
import numpy as np


def cart_powers(l):
    return [(lx, ly, l - lx - ly) for lx in range(l, -1, -1) for ly in range(l - lx, -1, -1)]


def eval_ao_cart(packed, ao_loc, coords, deriv=0):
    """AO values on the grid in the internal Cartesian order: [ncomp, nao_int, ngrids], ncomp = 1 or 4."""
    packed = np.asarray(packed, dtype=float)
    coords = np.asarray(coords, dtype=float)          # [ngrids, 3]
    nao = int(ao_loc[-1])
    ng = coords.shape[0]
    out = np.zeros((4 if deriv else 1, nao, ng))
    for n in range(packed.shape[0]):
        w = int(ao_loc[n + 1] - ao_loc[n])
        if w == 0:
            continue
        l, npr = int(packed[n, 11]), int(packed[n, 10])
        r = coords - packed[n, :3]
        r2 = (r * r).sum(axis=1)
        rad = np.zeros(ng)
        drad = np.zeros(ng)                            # d/d(r^2) of the radial part times 2 -> factor of x_i
        for p in range(npr):
            c, a = packed[n, 4 + 2 * p], packed[n, 5 + 2 * p]
            e = c * np.exp(-a * r2)
            rad += e
            drad += -2.0 * a * e
        x, y, z = r[:, 0], r[:, 1], r[:, 2]
        pw = lambda v, k: v ** k if k > 0 else np.ones(ng)
        for i, (lx, ly, lz) in enumerate(cart_powers(l)):
            ang = pw(x, lx) * pw(y, ly) * pw(z, lz)
            row = ao_loc[n] + i
            out[0, row] = ang * rad
            if deriv:
                dx = (lx * pw(x, lx - 1) if lx > 0 else 0.0) * pw(y, ly) * pw(z, lz)
                dy = pw(x, lx) * (ly * pw(y, ly - 1) if ly > 0 else 0.0) * pw(z, lz)
                dz = pw(x, lx) * pw(y, ly) * (lz * pw(z, lz - 1) if lz > 0 else 0.0)
                out[1, row] = dx * rad + ang * x * drad
                out[2, row] = dy * rad + ang * y * drad
                out[3, row] = dz * rad + ang * z * drad
    return out


def eval_ao_mol(layout, coords, deriv=0):
    """AO values in the molecule's own AO basis (sph or cart): [ncomp, nao_mol, ngrids]."""
    ao = eval_ao_cart(layout.packed, layout.ao_loc, coords, deriv)
    T = layout.transform_matrix()                      # [nao_int, nao_mol]
    return np.einsum("pi,cpg->cig", T, ao)


NDIM = {"LDA": 1, "GGA": 4, "MGGA": 5}


def eval_rho(layout, coords, dm, xctype="LDA"):
    xctype = xctype.upper()
    ao = eval_ao_mol(layout, coords, deriv=0 if xctype == "LDA" else 1)
    dm = np.asarray(dm, dtype=float)
    c0 = dm @ ao[0]                                    # [nao, ng]
    rho = np.zeros((NDIM[xctype], ao.shape[2]))
    rho[0] = np.einsum("ig,ig->g", ao[0], c0)
    if xctype != "LDA":
        for x in range(3):
            rho[1 + x] = np.einsum("ig,ig->g", ao[1 + x], c0) + np.einsum("ig,ig->g", ao[0], dm @ ao[1 + x])
        if xctype == "MGGA":
            rho[4] = 0.5 * sum(np.einsum("ig,ig->g", ao[1 + x], dm @ ao[1 + x]) for x in range(3))
    return rho


def eval_vxc(layout, coords, wv, xctype="LDA"):
    xctype = xctype.upper()
    wv = np.asarray(wv, dtype=float).reshape(NDIM[xctype], -1)
    ao = eval_ao_mol(layout, coords, deriv=0 if xctype == "LDA" else 1)
    if xctype == "LDA":
        return ao[0] @ (ao[0] * wv[0]).T
    w = wv.copy()
    w[0] *= 0.5
    aow = np.einsum("nig,ng->ig", ao[:4], w[:4])
    v = ao[0] @ aow.T
    if xctype == "MGGA":
        w4 = 0.5 * 0.5 * wv[4]                          # test_rks.py:187 (x0.5) and _tau_dot's own 1/2
        v += sum(ao[1 + x] @ (ao[1 + x] * w4).T for x in range(3))
    return v + v.T


# ------------------------------------------------------------------------------------------- VV10
def vv10_kernel(coords, vvcoords, W0, K, W0p, Kp, RpW, chunk=2048):
    """F, U, W of jqc/backend/dft/vv10.cu:86-117 (float64 throughout)."""
    n = coords.shape[0]
    F = np.zeros(n); U = np.zeros(n); W = np.zeros(n)
    for i0 in range(0, n, chunk):
        c = coords[i0:i0 + chunk]
        R2 = ((c[:, None, :] - vvcoords[None, :, :]) ** 2).sum(-1)
        gp = R2 * W0p[None] + Kp[None]
        g = R2 * W0[i0:i0 + chunk, None] + K[i0:i0 + chunk, None]
        gt = g + gp
        T = RpW[None] / (gp * (g * gt) ** 2)
        F[i0:i0 + chunk] = (T * g * gt).sum(1)
        U[i0:i0 + chunk] = (T * (g + gt)).sum(1)
        W[i0:i0 + chunk] = (T * R2 * (g + gt)).sum(1)
    return -1.5 * F, U, W


def vv10nlc(rho, coords, vvrho, vvweight, vvcoords, nlc_pars, sums=vv10_kernel):
    """exc[ngrids], vxc[2, ngrids] of jqc/backend/rks.py:542-715 (thresholding, pre/post algebra); ``sums`` = the pair-sum
    routine (NumPy above, or the C statement of the same loop, oracle/jk.py:vv10_sums)."""
    thresh = 1e-10
    rho = np.asarray(rho); vvrho = np.asarray(vvrho)
    m = rho[0] >= thresh
    mi = vvrho[0] >= thresh
    dens, g2 = rho[0][m], (rho[1:4][:, m] ** 2).sum(0)
    idens, ig2 = vvrho[0][mi], (vvrho[1:4][:, mi] ** 2).sum(0)
    Bvv, Cvv = nlc_pars
    Pi43 = 4.0 * np.pi / 3.0
    Kvv = Bvv * 1.5 * np.pi * ((9.0 * np.pi) ** (-1.0 / 6.0))
    Beta = ((3.0 / (Bvv * Bvv)) ** 0.75) / 32.0
    W0p = np.sqrt(Cvv * (ig2 / idens ** 2) ** 2 + Pi43 * idens)
    Kp = Kvv * idens ** (1.0 / 6.0)
    W0tmp = Cvv * (g2 / dens ** 2) ** 2
    W0 = np.sqrt(W0tmp + Pi43 * dens)
    K = Kvv * dens ** (1.0 / 6.0)
    dKdR = K / 6.0
    F, U, W = sums(np.asarray(coords)[m], np.asarray(vvcoords)[mi], W0, K, W0p, Kp, idens * np.asarray(vvweight)[mi])
    dW0dR = (0.5 * Pi43 * dens - 2.0 * W0tmp) / W0
    dW0dG = W0tmp * dens / (g2 * W0)
    exc = np.zeros(rho.shape[1]); vxc = np.zeros((2, rho.shape[1]))
    exc[m] = Beta + 0.5 * F
    vxc[0, m] = Beta + F + 1.5 * (U * dKdR + W * dW0dR)
    vxc[1, m] = 1.5 * W * dW0dG
    return exc, vxc
