"""Nuclear gradient of the two-electron energy on the GPU (csrc/kernels/jk_grad.hip through the C ABI).

The reference has no gradient kernels (it defers to GPU4PySCF, /root/reference/jqc/pyscf/tests/test_geom_opt.py:250-354), so
the checks are: (1) the analytic oracle of oracle/grad.py (McMurchie-Davidson, pinned by finite differences of the pinned J/K
oracle in tests/test_grad_oracle.py) on small systems, closed shell / two spin densities / long range / scaled J and K,
spherical and Cartesian; (2) size-independent: the directional derivative of the GPU's own two-electron energy (the parity-green
get_jk) along random displacements, for every angular class up to g (benzene with the artificial s..g basis), for
benzene / def2-TZVPP and for the 112-atom stand-in with def2-SVP; translational invariance everywhere.
"""
import os
import sys

import numpy as np
import pytest

from conftest import H2O, benzene_atoms

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASIS = {"O": [[0, [11.0, 0.3], [2.1, 0.5], [0.6, 0.4]], [1, [1.9, 0.6], [0.45, 0.5]], [2, [0.9, 1.0]]],
         "H": [[0, [3.4, 0.2], [0.6, 0.8]], [1, [0.8, 1.0]]]}


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


@pytest.mark.parametrize("cart", [False, True])
def test_gradient_kernels_against_the_analytic_oracle(cart):
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import grad
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import grad as G
    mol = mole.Mole(atom="O 0 0.05 -0.1; H 0.3 1.1 1.45; H -1.2 0.4 -0.9", basis=BASIS, unit="B", cart=cart)
    lay = BasisLayout.from_mol(mol, alignment=1)
    fn = grad.generate_jk_energy_per_atom(lay, cutoff=1e-16)
    rng = np.random.default_rng(5)
    n = mol.nao
    d = rng.random((n, n)) - 0.3
    a, b = rng.random((n, n)) - 0.4, rng.random((n, n)) - 0.6
    cases = [(d + d.T, 1.0, 1.0, None), (d + d.T, 1.0, 0.0, None), (d + d.T, 0.0, 1.0, 0.3),
             (np.stack([a + a.T, b + b.T]), 0.7, 0.35, None), (np.stack([a + a.T, b + b.T]), 0.0, 0.2, 0.4)]
    for dm, jf, kf, om in cases:
        ref = G.jk_energy_per_atom(lay, dm, jf, kf, om)
        out = fn(mol, dm, j_factor=jf, k_factor=kf, omega=om)
        assert isinstance(out, np.ndarray) and out.shape == (3, 3)
        scale = np.abs(ref).max()
        assert np.abs(out - ref).max() < 1e-10 * scale, (jf, kf, om, out, ref)
        assert np.abs(out.sum(0)).max() < 1e-10 * scale


def _directional_check(mol_of, coords, basis_layout_of, dm, ndir, h, tol, cutoff=1e-14, **kw):
    """Gradient kernel . v  vs  Richardson central difference of the GPU two-electron energy along v."""
    import torch
    from joltqc_amd.pyscf import grad
    from joltqc_amd.pyscf import jk as jkmod

    def energy(c):
        mol = mol_of(c)
        lay = basis_layout_of(mol)
        vj, vk = jkmod.generate_jk_kernel(lay, cutoff, cutoff)(mol, dm, hermi=1, **kw)
        return float((dm * (0.5 * vj - 0.25 * vk)).sum())

    mol0 = mol_of(coords)
    lay0 = basis_layout_of(mol0)
    fn = grad.generate_jk_energy_per_atom(lay0, cutoff=cutoff)
    g = _np(fn(mol0, dm, **kw))
    scale = np.abs(g).max()
    assert np.abs(g.sum(0)).max() < 1e-9 * scale
    rng = np.random.default_rng(11)
    worst = 0.0
    for _ in range(ndir):
        v = rng.normal(size=coords.shape)
        v /= np.linalg.norm(v)
        d1 = (energy(coords + h * v) - energy(coords - h * v)) / (2 * h)
        d2 = (energy(coords + 2 * h * v) - energy(coords - 2 * h * v)) / (4 * h)
        fd = (4 * d1 - d2) / 3
        ana = float((g * v).sum())
        worst = max(worst, abs(fd - ana) / max(abs(fd), scale * 1e-2))
        assert abs(fd - ana) < tol * max(abs(fd), scale * 1e-2), (fd, ana)
    return fn, worst


def test_every_class_up_to_g_directional_derivative():
    """benzene with the artificial s..g basis of the reference's autotuner (generate_fragment.py:97-114): all 140 classes."""
    import torch
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
              [2, [0.8, 1.0]], [3, [0.9, 1.0]], [4, [1.0, 1.0]]]
    atoms = benzene_atoms()
    sym = [a[0] for a in atoms]
    coords = np.array([a[1] for a in atoms]) / 0.52917721092
    mol_of = lambda c: mole.Mole(atom=[(s, tuple(x)) for s, x in zip(sym, c)], basis={"C": shells, "H": shells}, unit="B")
    lay_of = lambda m: BasisLayout.from_mol(m, alignment=tile_width)
    n = mol_of(coords).nao
    rng = np.random.default_rng(3)
    c = rng.random((n, 21)) - 0.5
    dm = torch.from_numpy(c @ c.T / 21).cuda()
    fn, _ = _directional_check(mol_of, coords, lay_of, dm, 2, 4e-3, 1e-5)
    assert fn.stats["launches"] >= 140 and fn.quartet_count() > 1e6


def test_benzene_tzvpp_long_range_exchange_directional_derivative():
    import torch
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    atoms = benzene_atoms()
    sym = [a[0] for a in atoms]
    coords = np.array([a[1] for a in atoms]) / 0.52917721092
    mol_of = lambda c: mole.Mole(atom=[(s, tuple(x)) for s, x in zip(sym, c)], basis="def2-tzvpp", unit="B")
    lay_of = lambda m: BasisLayout.from_mol(m, alignment=tile_width)
    n = mol_of(coords).nao
    rng = np.random.default_rng(4)
    c = rng.random((n, 21)) - 0.5
    dm = torch.from_numpy(c @ c.T / 21).cuda()
    _directional_check(mol_of, coords, lay_of, dm, 2, 4e-3, 1e-5)
    _directional_check(mol_of, coords, lay_of, dm, 1, 4e-3, 1e-5, omega=0.3)


def test_112_atoms_svp_directional_derivative_and_default_cutoff():
    """Taxol-size stand-in / def2-SVP: gradient at the default screening threshold vs the tight one, one random direction."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import grad
    from joltqc_amd.pyscf.basis import BasisLayout
    xyz = mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", "0112-elongated-nitrogenous.xyz"))
    rows = [r.split() for r in xyz.splitlines() if r.strip()]
    sym = [r[0] for r in rows]
    coords = np.array([[float(x) for x in r[1:4]] for r in rows]) / 0.52917721092
    mol_of = lambda c: mole.Mole(atom=[(s, tuple(x)) for s, x in zip(sym, c)], basis="def2-svp", unit="B")
    lay_of = lambda m: BasisLayout.from_mol(m, alignment=tile_width)
    mol0 = mol_of(coords)
    np.random.seed(9)
    nocc = mol0.nelectron // 2
    c = np.random.rand(mol0.nao, nocc) - 0.5
    dm = torch.from_numpy(c @ c.T / nocc).cuda()
    fn, _ = _directional_check(mol_of, coords, lay_of, dm, 1, 4e-3, 1e-5, cutoff=1e-13)
    g13 = _np(fn(mol0, dm))
    g10 = _np(grad.generate_jk_energy_per_atom(lay_of(mol0), cutoff=1e-10)(mol0, dm))
    assert np.abs(g13 - g10).max() < 1e-6 * np.abs(g13).max()


@pytest.mark.parametrize("kind", ["lane", "cooperative"])
def test_per_atom_lds_tables_and_their_overflow_paths(kind):
    """The gradient kernels accumulate into a per-atom table in LDS (1 024 atoms in the one-quartet-per-lane form, 256 in the
    cooperative form); atoms beyond the table go to global memory directly.  Lattices with MORE atoms than the tables hold -- 1 100 He
    atoms with one s function (class (ss|ss): lane form), 300 with s, p, d functions (the d classes: cooperative form) -- and the same
    lattice with its atoms in REVERSED order, so that every atom changes sides of the table boundary: the gradient must come out
    permuted (1e-11 of the largest component), sum to zero over the atoms, and differ from zero."""
    import torch
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import grad
    from joltqc_amd.pyscf.basis import BasisLayout
    rng = np.random.default_rng(17)
    if kind == "lane":
        n1, basis = 11, {"He": [[0, [0.9, 1.0]]]}                                   # 11 x 10 x 10 = 1 100 atoms
        xyz = np.array([(a, b, c) for a in range(n1) for b in range(10) for c in range(10)], dtype=float) * 3.2
    else:
        basis = {"He": [[0, [1.1, 1.0]], [1, [0.9, 1.0]], [2, [1.0, 1.0]]]}
        xyz = np.array([(a, b, c) for a in range(10) for b in range(6) for c in range(5)], dtype=float) * 3.6   # 300 atoms
    xyz = xyz + rng.normal(size=xyz.shape) * 0.15
    natm = len(xyz)
    out = []
    for order in (np.arange(natm), np.arange(natm)[::-1]):
        mol = mole.Mole(atom=[("He", tuple(xyz[a])) for a in order], basis=basis, unit="B")
        lay = BasisLayout.from_mol(mol, alignment=1)
        nao = mol.nao
        per = nao // natm
        # a density that is the same function of the ATOMS in both orders: block (a, b) depends on the atoms' lattice identities
        blocks = np.random.default_rng(3).random((per, per)) - 0.4
        w = np.exp(-0.08 * np.linalg.norm(xyz[order][:, None] - xyz[order][None], axis=2) ** 2)
        dm = np.kron(w, blocks + blocks.T)
        fn = grad.generate_jk_energy_per_atom(lay, cutoff=1e-12)
        g = _np(fn(mol, torch.from_numpy(dm).cuda()))
        assert g.shape == (natm, 3) and np.isfinite(g).all()
        back = np.empty_like(g)
        back[order] = g                                                             # gradient per LATTICE atom
        out.append(back)
    scale = np.abs(out[0]).max()
    assert scale > 1e-3
    assert np.abs(out[0].sum(axis=0)).max() < 1e-10 * scale * natm
    assert np.abs(out[0] - out[1]).max() < 1e-11 * scale, np.abs(out[0] - out[1]).max() / scale


def test_rhf_forces_through_apply_match_finite_differences_of_the_scf_energy():
    """End to end on H2O / def2-SVP: apply() installs ``_jqc_jk_energy_per_atom``; with the one-electron, overlap and nuclear
    terms differentiated numerically at FIXED density (cheap CPU integrals) the force along a random displacement equals the
    finite difference of the converged SCF energy (Hellmann-Feynman + Pulay terms consistent with the SCF that produced D)."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dense
    from standin_scf import RHF
    sym = ["O", "H", "H"]
    coords = np.array([[0.0, 0.0, 0.1174], [-0.757, 0.0, -0.4696], [0.857, 0.1, -0.4696]]) / 0.52917721092
    mol_of = lambda c: mole.Mole(atom=[(s, tuple(x)) for s, x in zip(sym, c)], basis="def2-svp", unit="B")

    def int1e(mol):
        S, T, V = dense.int1e_mol(BasisLayout.from_mol(mol), mol)
        return T + V, S

    def scf(c):
        mf = jp.apply(RHF(mol_of(c), int1e=int1e))
        mf.conv_tol = 1e-12
        e = mf.kernel()
        assert mf.converged
        return e, mf

    e0, mf = scf(coords)
    mol = mf.mol
    D = np.asarray(mf.make_rdm1())
    nocc = mol.nelectron // 2
    C, eps = np.asarray(mf.mo_coeff), np.asarray(mf.mo_energy)
    W = 2.0 * (C[:, :nocc] * eps[:nocc]) @ C[:, :nocc].T
    ejk = mf._jqc_jk_energy_per_atom(mol, D)
    assert isinstance(ejk, np.ndarray) and ejk.shape == (3, 3)
    rng = np.random.default_rng(2)
    v = rng.normal(size=coords.shape)
    v /= np.linalg.norm(v)

    def one_electron(c):
        m = mol_of(c)
        h, S = int1e(m)
        return float(np.einsum("ij,ji->", D, h)) - float(np.einsum("ij,ji->", W, S)) + m.energy_nuc()

    def richardson(f, h):
        d1 = (f(coords + h * v) - f(coords - h * v)) / (2 * h)
        d2 = (f(coords + 2 * h * v) - f(coords - 2 * h * v)) / (4 * h)
        return (4 * d1 - d2) / 3

    force = richardson(one_electron, 2e-3) + float((ejk * v).sum())
    fd = richardson(lambda c: scf(c)[0], 4e-3)
    assert abs(force - fd) < 2e-7, (force, fd)


@pytest.mark.parametrize("xctype,cart", [("LDA", False), ("GGA", False), ("GGA", True), ("MGGA", False), ("MGGA", True)])
def test_xc_gradient_kernels_against_finite_differences_of_the_oracle(xctype, cart):
    """dE_xc/dR at fixed D and fixed grid (LDA, GGA, meta-GGA): the MFMA kernels vs finite differences of the CPU oracle's
    linearised functional sum_g wv . (rho, grad rho, tau) (oracle/dft.py eval_rho), s..f shells, points near and far from the
    nuclei."""
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import rks
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import grad as G
    basis = {"O": BASIS["O"] + [[3, [1.1, 1.0]]], "H": BASIS["H"] + [[2, [0.7, 1.0]]]}
    sym = ["O", "H", "H"]
    coords = np.array([[0.0, 0.05, -0.1], [0.3, 1.1, 1.45], [-1.2, 0.4, -0.9]])
    mol_of = lambda c: mole.Mole(atom=[(s, tuple(x)) for s, x in zip(sym, c)], basis=basis, unit="B", cart=cart)
    lay_of = lambda c: BasisLayout.from_mol(mol_of(c), alignment=1)
    mol = mol_of(coords)
    rng = np.random.default_rng(8)
    ng = 1024
    pts = coords[rng.integers(0, 3, ng)] + rng.normal(0, 0.9, (ng, 3))
    pts = pts[np.lexsort(pts.T)]
    ndim = {"LDA": 1, "GGA": 4, "MGGA": 5}[xctype]
    wv = rng.normal(0, 1.0, (ndim, ng)) * 0.01
    n = mol.nao
    d = rng.random((n, n)) - 0.4
    dm = d + d.T

    class Gr:
        pass
    g = Gr(); g.coords = pts; g.weights = np.ones(ng)
    rks_fun, _, _ = rks.generate_rks_kernel(lay_of(coords), cutoff_fp64=1e-16, cutoff_fp32=1e-16)
    out = rks_fun.xcgrad_fun(mol, g, xctype, dm, wv)
    ref = G.xc_energy_per_atom_fd(lay_of, coords, pts, dm, wv, xctype)
    assert isinstance(out, np.ndarray) and out.shape == (3, 3)
    assert np.abs(out - ref).max() < 2e-8 * np.abs(ref).max(), (out, ref)


def test_xc_gradient_112_atoms_directional_derivative():
    """Taxol-size stand-in / def2-SVP on a Becke grid: XC gradient kernels (GGA) vs the directional derivative of
    sum_g wv . rho computed with the parity-green rho kernels at displaced geometries (size-independent property)."""
    import torch
    from joltqc_amd.gto import mole
    from joltqc_amd.gto.grids import Grids
    from joltqc_amd.pyscf import rks
    from joltqc_amd.pyscf.basis import BasisLayout
    xyz = mole.read_xyz(os.path.join(ROOT, "joltqc_amd/data/molecules", "0112-elongated-nitrogenous.xyz"))
    rows = [r.split() for r in xyz.splitlines() if r.strip()]
    sym = [r[0] for r in rows]
    coords = np.array([[float(x) for x in r[1:4]] for r in rows]) / 0.52917721092
    mol_of = lambda c: mole.Mole(atom=[(s, tuple(x)) for s, x in zip(sym, c)], basis="def2-svp", unit="B")
    mol0 = mol_of(coords)
    gg = Grids(mol0, 20, 6).build()
    order = rks.arg_group_grids(gg.coords)
    n = gg.coords.shape[0] // 256 * 256

    class Gr:
        pass
    g = Gr(); g.coords = gg.coords[order][:n]; g.weights = gg.weights[order][:n]
    np.random.seed(9)
    nocc = mol0.nelectron // 2
    c = np.random.rand(mol0.nao, nocc) - 0.5
    dm = torch.from_numpy(c @ c.T / nocc).cuda()
    rng = np.random.default_rng(1)
    wv = torch.from_numpy(rng.normal(0, 1.0, (4, n)) * g.weights).cuda()

    def lin(cc):
        m = mol_of(cc)
        _, rho_k, _ = rks.generate_rks_kernel(BasisLayout.from_mol(m, alignment=1), cutoff_fp64=1e-15, cutoff_fp32=1e-15)
        return float((rho_k(m, g, "GGA", dm) * wv).sum())

    rks_fun, _, _ = rks.generate_rks_kernel(BasisLayout.from_mol(mol0, alignment=1), cutoff_fp64=1e-15, cutoff_fp32=1e-15)
    gx = _np(rks_fun.xcgrad_fun(mol0, g, "GGA", dm, wv))
    assert gx.shape == (len(sym), 3)
    v = rng.normal(size=coords.shape)
    v /= np.linalg.norm(v)
    h = 2e-3
    d1 = (lin(coords + h * v) - lin(coords - h * v)) / (2 * h)
    d2 = (lin(coords + 2 * h * v) - lin(coords - 2 * h * v)) / (4 * h)
    fd = (4 * d1 - d2) / 3
    ana = float((gx * v).sum())
    assert abs(fd - ana) < 1e-6 * max(abs(fd), np.abs(gx).max() * 1e-2), (fd, ana)


def test_rks_forces_lda_match_finite_differences_of_the_scf_energy():
    """RKS (Slater exchange stand-in for libxc) through apply() on H2O / def2-SVP with a grid FIXED in space: J gradient
    (k_factor 0) + XC gradient kernels + numerically differentiated one-electron / overlap / nuclear terms equal the finite
    difference of the converged SCF energy along a random displacement."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dense
    from standin_scf import RKS, Grids as G, SlaterNumInt
    sym = ["O", "H", "H"]
    coords = np.array([[0.0, 0.0, 0.1174], [-0.757, 0.0, -0.4696], [0.857, 0.1, -0.4696]]) / 0.52917721092
    mol_of = lambda c: mole.Mole(atom=[(s, tuple(x)) for s, x in zip(sym, c)], basis="def2-svp", unit="B")
    rng = np.random.default_rng(7)
    pts = coords[rng.integers(0, 3, 8192)] + rng.normal(0, 1.0, (8192, 3))
    pts = pts[np.lexsort(pts.T)]
    weights = np.full(8192, 0.003)

    def int1e(mol):
        S, T, V = dense.int1e_mol(BasisLayout.from_mol(mol), mol)
        return T + V, S

    cfg = jp.get_default_config()
    cfg["dft"] = {"cutoff_fp32": 1e-14, "cutoff_fp64": 1e-14}

    def scf(c):
        m = mol_of(c)
        h, S = int1e(m)
        mf = jp.apply(RKS(m, h, S, G(pts, weights)), cfg)
        mf.conv_tol = 1e-12
        e = mf.kernel()
        assert mf.converged
        return e, mf

    e0, mf = scf(coords)
    mol = mf.mol
    D = np.asarray(mf.make_rdm1())
    nocc = mol.nelectron // 2
    C, eps = np.asarray(mf.mo_coeff), np.asarray(mf.mo_energy)
    W = 2.0 * (C[:, :nocc] * eps[:nocc]) @ C[:, :nocc].T
    ej = mf._jqc_jk_energy_per_atom(mol, D, j_factor=1.0, k_factor=0.0)
    # potential of the converged density: vrho of Slater exchange, weighted
    rho = _np(mf._numint.get_rho(mol, D, mf.grids))
    rho = np.maximum(rho.reshape(-1)[:8192], 0)
    wv = (4.0 / 3.0 * SlaterNumInt.CX * rho ** (1.0 / 3.0) * weights)[None]
    exc1 = mf._jqc_xc_energy_per_atom(mol, mf.grids, "LDA", D, wv)
    v = rng.normal(size=coords.shape)
    v /= np.linalg.norm(v)

    def one_electron(c):
        m = mol_of(c)
        h, S = int1e(m)
        return float(np.einsum("ij,ji->", D, h)) - float(np.einsum("ij,ji->", W, S)) + m.energy_nuc()

    def richardson(f, h):
        d1 = (f(coords + h * v) - f(coords - h * v)) / (2 * h)
        d2 = (f(coords + 2 * h * v) - f(coords - 2 * h * v)) / (4 * h)
        return (4 * d1 - d2) / 3

    force = richardson(one_electron, 2e-3) + float(((np.asarray(ej) + np.asarray(exc1)) * v).sum())
    fd = richardson(lambda c: scf(c)[0], 4e-3)
    assert abs(force - fd) < 5e-7, (force, fd)
