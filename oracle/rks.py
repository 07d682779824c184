"""Oracle-side Kohn-Sham potential: the reference's nr_rks + get_veff restated on the CPU.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/jqc/pyscf/rks.py:308-364 (nr_rks: rho -> eval_xc_eff -> nelec, excsum, vxcmat; without the
incremental bookkeeping, which changes no number) and :184-262 (get_veff for a pure functional: vxc + vj, ecoul =
1/2 tr(D J), exc from the quadrature) with the dense AO formulas of oracle/dft.py, the closed-form functionals of
oracle/xc.py and the Rys J/K oracle.  AO values on the grid are cached (they depend on the geometry only).
"""
import numpy as np

from . import dft, jk, xc


class Tagged(np.ndarray):
    """ndarray with the attributes PySCF's tag_array attaches to a potential (ecoul, exc, vj, vk)."""


def make_get_veff(layout, coords, weights, xc_code, get_j, get_k=None, nlc_grid=None):
    """``get_veff(mol, dm, ...)`` of an RKS object for the functional ``xc_code``; ``get_j(dm) -> J`` in the molecule's AO
    basis; hybrids: ``get_k(dm, omega) -> K`` (omega = None: full range), combined as reference rks.py:232-250 does:
    K = hyb K_full + (alpha - hyb) K_lr(omega), V -= K / 2, E_xc -= tr(D K) / 4.  A functional with a VV10 part
    (``xc.nlc_coeff``) adds nr_nlc_vxc (reference rks.py:661-714: GGA-type rho on the NLC grid ``nlc_grid = (coords,
    weights)``, ``vv10nlc`` with the grid as its own partner set, E_nlc = sum rho w e_nlc, V_nlc from
    wv = (v_rho, 2 v_sigma grad rho) w) as get_veff does at :199-209."""
    kind = xc.xc_type(xc_code)
    ao = dft.eval_ao_mol(layout, coords, deriv=0 if kind == "LDA" else 1)       # [ncomp, nao, ngrids]
    w = np.asarray(weights, dtype=np.float64)
    stats = {}

    def nr_rks(dm):
        dm = np.asarray(dm, dtype=np.float64)
        c0 = dm @ ao[0]
        rho = np.empty((dft.NDIM[kind], ao.shape[2]))
        rho[0] = np.einsum("ig,ig->g", ao[0], c0)
        for x in range(1, min(rho.shape[0], 4)):
            rho[x] = 2.0 * np.einsum("ig,ig->g", ao[x], c0)                     # symmetric D (reference eval_rho.cu:300-383)
        if kind == "MGGA":                                                      # tau = 1/2 sum_ab D_ab grad phi_a . grad phi_b (:328-377)
            rho[4] = 0.5 * sum(np.einsum("ig,ig->g", ao[x], dm @ ao[x]) for x in range(1, 4))
        exc, vxc = xc.eval_xc_eff(xc_code, rho if kind != "LDA" else rho[0])
        den = rho[0] * w
        wv = vxc * w
        if kind == "LDA":
            vmat = ao[0] @ (ao[0] * wv[0]).T
        else:
            wv[0] *= 0.5                                                        # reference tests/test_rks.py:158-162
            v = ao[0] @ np.einsum("nig,ng->ig", ao[:4], wv[:4]).T
            if kind == "MGGA":                                                  # test_rks.py:185-192: wv[4] * 0.5, and tau's own 1/2
                v += sum(ao[x] @ (ao[x] * (0.25 * wv[4])).T for x in range(1, 4))
            vmat = v + v.T
        return float(den.sum()), float((den * exc).sum()), vmat

    nlc = xc.nlc_coeff(xc_code)
    if nlc:
        nc, nw = (coords, w) if nlc_grid is None else (np.asarray(nlc_grid[0], float), np.asarray(nlc_grid[1], float))
        ao_n = ao[:4] if nlc_grid is None else dft.eval_ao_mol(layout, nc, deriv=1)

    def nr_nlc_vxc(dm):
        c0 = dm @ ao_n[0]
        rho = np.empty((4, ao_n.shape[2]))
        rho[0] = np.einsum("ig,ig->g", ao_n[0], c0)
        for x in range(1, 4):
            rho[x] = 2.0 * np.einsum("ig,ig->g", ao_n[x], c0)
        exc, vxc = 0.0, 0.0
        for pars, fac in nlc:
            e, v = dft.vv10nlc(rho, nc, rho, nw, nc, pars, sums=jk.vv10_sums)
            exc, vxc = exc + fac * e, vxc + fac * v
        wv = np.empty((4, rho.shape[1]))
        wv[0] = 0.5 * vxc[0] * nw                                               # (x 0.5: the GGA convention above)
        wv[1:4] = 2.0 * vxc[1] * rho[1:4] * nw                                  # xc_deriv.transform_vxc(rho, vxc, "GGA", spin=0), rks.py:700
        v = ao_n[0] @ np.einsum("nig,ng->ig", ao_n, wv).T
        return float((rho[0] * nw * exc).sum()), v + v.T

    def get_veff(mol=None, dm=None, dm_last=None, vhf_last=None, hermi=1):
        dm = np.asarray(dm, dtype=np.float64)
        nelec, exc, vxc = nr_rks(dm)
        if nlc:
            enlc, vnlc = nr_nlc_vxc(dm)
            exc, vxc = exc + enlc, vxc + vnlc
            stats["enlc"] = enlc
        vj = get_j(dm)
        omega, alpha, hyb = xc.rsh_and_hybrid_coeff(xc_code)
        vk = None
        if abs(hyb) > 1e-10 or abs(alpha) > 1e-10:
            vk = hyb * get_k(dm, None) if abs(hyb) > 1e-10 else 0.0
            if abs(omega) > 1e-10:
                vk = vk + (alpha - hyb) * get_k(dm, omega)
            vxc = vxc - 0.5 * vk
            exc -= 0.25 * float(np.einsum("ij,ji->", dm, vk))
        out = (vxc + vj).view(Tagged)
        out.ecoul = 0.5 * float(np.einsum("ij,ji->", dm, vj))
        out.exc, out.vj, out.vk = exc, vj, vk
        stats["nelec"] = nelec
        return out
    get_veff.stats = stats
    return get_veff
