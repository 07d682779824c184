"""Oracle of the nuclear gradient of the two-electron energy (TEST INFRASTRUCTURE ONLY).

The reference holds NO gradient code (JoltQC leaves gradients to GPU4PySCF, /root/reference/jqc/pyscf/__init__.py:63-97), so
there is no reference kernel, test vector or known answer to restate here.  What pins this oracle instead:

* ``jk_energy`` is the two-electron energy from the pinned J/K oracle (oracle/dense.py -> oracle/jk_oracle.c, which follows
  reference jk/1q1t.cu:86-638 and is pinned by the reference's own energies and vectors);
* ``jk_energy_per_atom_fd`` differentiates it numerically (central differences, one Richardson step: error O(h^4));
* ``jk_energy_per_atom`` is the analytic gradient from the independent McMurchie-Davidson engine (oracle/md_eri.py) through
  d/dA_x [a b|c d] = 2 alpha [a+1_x b|c d] - a_x [a-1_x b|c d]; tests/test_grad_oracle.py checks it against the finite
  differences, the GPU tests check the HIP kernels against both.

Energy convention (the one GPU4PySCF's ``_jk_energy_per_atom`` differentiates): for n spin densities D^s, D = sum_s D^s,
    E2 = 1/2 j_factor tr(D J[D]) - 1/4 k_factor n sum_s tr(D^s K[D^s]).
"""
import numpy as np

from . import dense, md_eri


def _spin_list(dm):
    dm = np.asarray(dm, dtype=np.float64)
    return dm.reshape(-1, dm.shape[-1], dm.shape[-1])


def jk_energy(layout, dm, j_factor=1.0, k_factor=1.0, omega=None):
    ds = _spin_list(dm)
    n = ds.shape[0]
    dt = ds.sum(0)
    e = 0.0
    if j_factor:
        vj = dense.get_jk(layout, dt, 1, omega=omega, with_k=False)[0]
        e += 0.5 * j_factor * float(np.einsum("ij,ji->", dt, vj))
    if k_factor:
        for d in ds:
            vk = dense.get_jk(layout, d, 1, omega=omega, with_j=False)[1]
            e -= 0.25 * k_factor * n * float(np.einsum("ij,ji->", d, vk))
    return e


def jk_energy_per_atom_fd(make_layout, coords, dm, j_factor=1.0, k_factor=1.0, omega=None, h=4e-3):
    """``make_layout(coords_bohr[natm, 3]) -> BasisLayout``; densities fixed.  (4 f(h) - f(2h)) / 3 of the central differences."""
    coords = np.asarray(coords, dtype=np.float64)
    out = np.zeros_like(coords)

    def central(ia, x, step):
        c = coords.copy()
        c[ia, x] += step
        ep = jk_energy(make_layout(c), dm, j_factor, k_factor, omega)
        c[ia, x] -= 2 * step
        em = jk_energy(make_layout(c), dm, j_factor, k_factor, omega)
        return (ep - em) / (2 * step)

    for ia in range(coords.shape[0]):
        for x in range(3):
            out[ia, x] = (4.0 * central(ia, x, h) - central(ia, x, 2 * h)) / 3.0
    return out


def _shifted_rows(row, prim, dl):
    """One-primitive copy of a packed shell row with angular momentum l + dl (same centre, coefficient and exponent)."""
    r = np.zeros(12)
    r[:3] = row[:3]
    r[4], r[5] = row[4 + 2 * prim], row[5 + 2 * prim]
    r[7] = r[9] = 1.0
    r[10] = 1
    r[11] = int(row[11]) + dl
    return r


def _deriv_block(rows, q, pos, omega):
    """d/dR of the Cartesian block (ij|kl) with respect to the centre of the shell at position ``pos`` of the quartet:
    array [3, nfi, nfj, nfk, nfl]."""
    base = [np.asarray(rows[s], dtype=float) for s in q]
    l0 = int(base[pos][11])
    pw = md_eri.cart_powers(l0)
    up = {p: n for n, p in enumerate(md_eri.cart_powers(l0 + 1))}
    dn = {p: n for n, p in enumerate(md_eri.cart_powers(l0 - 1))} if l0 > 0 else {}
    shape = [len(md_eri.cart_powers(int(b[11]))) for b in base]
    out = np.zeros([3] + shape)
    for prim in range(int(base[pos][10])):
        alpha = base[pos][5 + 2 * prim]
        tmp = list(base)
        tmp[pos] = _shifted_rows(base[pos], prim, +1)
        blk_up = np.moveaxis(md_eri.eri_block(tmp, 0, 1, 2, 3, omega or 0.0), pos, 0)
        blk_dn = None
        if l0 > 0:
            tmp[pos] = _shifted_rows(base[pos], prim, -1)
            blk_dn = np.moveaxis(md_eri.eri_block(tmp, 0, 1, 2, 3, omega or 0.0), pos, 0)
        o = np.moveaxis(out, pos + 1, 1)                     # [3, nf(pos), ...]
        for n, (ax, ay, az) in enumerate(pw):
            for x, a in enumerate((ax, ay, az)):
                raised = [ax, ay, az]
                raised[x] += 1
                o[x, n] += 2.0 * alpha * blk_up[up[tuple(raised)]]
                if a > 0:
                    lowered = [ax, ay, az]
                    lowered[x] -= 1
                    o[x, n] -= a * blk_dn[dn[tuple(lowered)]]
    return out


def jk_energy_per_atom(layout, dm, j_factor=1.0, k_factor=1.0, omega=None):
    """Analytic gradient [natm, 3] over every canonical quartet of the real shells (no screening); small systems only."""
    ds = _spin_list(dm)
    n = ds.shape[0]
    T = layout.transform_matrix()
    di = np.einsum("pi,nij,qj->npq", T, ds, T)
    di = 0.5 * (di + di.transpose(0, 2, 1))
    dt = di.sum(0)
    rows, loc, atom = layout.packed, layout.ao_loc, layout.atom_of
    natm = int(atom.max()) + 1
    out = np.zeros((natm, 3))
    nb = layout.nbasis
    for (i, j, k, l) in dense.canonical_quartets(layout).astype(int):
        atoms = (atom[i], atom[j], atom[k], atom[l])
        if len(set(atoms)) == 1:
            continue
        fac = 1.0
        if i == j:
            fac *= 0.5
        if k == l:
            fac *= 0.5
        if i * nb + j == k * nb + l:
            fac *= 0.5
        si, sj, sk, sl = (slice(loc[s], loc[s + 1]) for s in (i, j, k, l))
        P = 4.0 * j_factor * np.einsum("ab,cd->abcd", dt[si, sj], dt[sk, sl])
        for d in di:
            P -= k_factor * n * (np.einsum("ac,bd->abcd", d[si, sk], d[sj, sl]) + np.einsum("ad,bc->abcd", d[si, sl], d[sj, sk]))
        P *= fac
        tot = np.zeros(3)
        for pos in range(3):
            g = np.einsum("xabcd,abcd->x", _deriv_block(rows, (i, j, k, l), pos, omega), P)
            out[atoms[pos]] += g
            tot += g
        out[atoms[3]] -= tot
    return out


# ------------------------------------------------------------------------------------------------ exchange-correlation part
def xc_linear_energy(layout, grid_coords, dm, wv, xctype="LDA"):
    """L = sum_g sum_c wv[c, g] rho_c(g): the first-order change of E_xc for the potential wv = weights x vxc; its nuclear
    derivative at fixed D and fixed grid is the XC gradient without grid response (what GPU4PySCF's `get_vxc` gradient gives
    with ``grid_response = False``)."""
    from . import dft
    rho = dft.eval_rho(layout, grid_coords, dm, xctype)
    wv = np.asarray(wv, dtype=float).reshape(rho.shape[0], -1)
    return float((rho * wv).sum())


def xc_energy_per_atom_fd(make_layout, coords, grid_coords, dm, wv, xctype="LDA", h=2e-3):
    """Central differences + one Richardson step of ``xc_linear_energy`` with respect to the nuclear positions."""
    coords = np.asarray(coords, dtype=np.float64)
    out = np.zeros_like(coords)

    def central(ia, x, step):
        c = coords.copy()
        c[ia, x] += step
        ep = xc_linear_energy(make_layout(c), grid_coords, dm, wv, xctype)
        c[ia, x] -= 2 * step
        em = xc_linear_energy(make_layout(c), grid_coords, dm, wv, xctype)
        return (ep - em) / (2 * step)

    for ia in range(coords.shape[0]):
        for x in range(3):
            out[ia, x] = (4.0 * central(ia, x, h) - central(ia, x, 2 * h)) / 3.0
    return out
