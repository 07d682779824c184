"""GPU parity of the DFT grid path (rho / vxc / VV10) against the CPU oracle, modelled on the reference's
jqc/pyscf/tests/test_rks.py (H2, inline s/p/d/f basis, seed 9, tolerances 1e-7)."""
import numpy as np
import pytest

from conftest import H2O, H2_BOHR

pytestmark = pytest.mark.gpu

BASIS = {"H": [[0, [34.0613410, 0.60251978e-2], [5.1235746, 0.45021094e-1], [1.1646626, 0.20189726]],
               [0, [0.32723041, 1.0]], [0, [0.10307241, 1.0]], [1, [1.407, 1.0]], [1, [0.388, 1.0]],
               [2, [1.057, 1.0]], [3, [1.057, 1.0]]]}          # reference tests/test_rks.py:37-56


class Grids:
    """Synthetic quadrature: random points around the atoms with random positive weights (the kernels are
    agnostic to how a grid was generated; Becke/Lebedev generation is third-party in the reference)."""

    def __init__(self, mol, n, seed=1):
        rng = np.random.default_rng(seed)
        at = mol.atom_coords()
        self.coords = at[rng.integers(0, len(at), n)] + rng.normal(0, 1.3, (n, 3))
        order = np.lexsort(self.coords.T)
        self.coords = np.ascontiguousarray(self.coords[order])
        self.weights = rng.random(n) * 0.05


def _setup(atom, basis, cart, n, unit="angstrom"):
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import rks
    from joltqc_amd.pyscf.basis import BasisLayout
    mol = mole.Mole(atom=atom, basis=basis, cart=cart, unit=unit)
    lay = BasisLayout.from_mol(mol, alignment=1)
    grids = Grids(mol, n)
    _, rho_k, vxc_k = rks.generate_rks_kernel(lay)
    return mol, lay, grids, rho_k, vxc_k


@pytest.mark.parametrize("xctype,ndim", [("LDA", 1), ("GGA", 4), ("MGGA", 5)])
@pytest.mark.parametrize("cart", [True, False])
def test_rho_and_vxc_h2(xctype, ndim, cart):
    from oracle import dft
    mol, lay, grids, rho_k, vxc_k = _setup(H2_BOHR, BASIS, cart, 1000, unit="B")   # 1000: exercises grid padding
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao)
    dm = dm @ dm.T
    rho = rho_k(mol, grids, xctype, dm).cpu().numpy()
    ref = dft.eval_rho(lay, grids.coords, dm, xctype)
    assert rho.shape == (ndim, 1000)
    assert np.abs(rho - ref).max() < 1e-7 * max(1.0, np.abs(ref).max())
    wv = np.random.rand(ndim, 1000)
    v = vxc_k(mol, grids, xctype, wv).cpu().numpy()
    vref = dft.eval_vxc(lay, grids.coords, wv, xctype)
    assert np.abs(v - vref).max() < 1e-7 * max(1.0, np.abs(vref).max())
    assert np.abs(v - v.T).max() < 1e-9 * max(1.0, np.abs(v).max())


def test_rho_vxc_water_svp_gga_sparse_blocks():
    from oracle import dft
    mol, lay, grids, rho_k, vxc_k = _setup(H2O, "def2-svp", False, 4096)
    grids.coords[:256] += 30.0                 # one far-away block: almost no significant shells
    np.random.seed(3)
    dm = np.random.rand(mol.nao, mol.nao)
    dm = dm + dm.T
    rho = rho_k(mol, grids, "GGA", dm).cpu().numpy()
    ref = dft.eval_rho(lay, grids.coords, dm, "GGA")
    assert np.abs(rho - ref).max() < 1e-8 * np.abs(ref).max()
    assert rho_k.stats["nrow_h"].min() < rho_k.stats["nrow_h"].max()
    wv = np.random.rand(4, 4096)
    v = vxc_k(mol, grids, "GGA", wv).cpu().numpy()
    vref = dft.eval_vxc(lay, grids.coords, wv, "GGA")
    assert np.abs(v - vref).max() < 1e-8 * np.abs(vref).max()


def test_fp32_window_and_pair_cutoff():
    """Reference precision windows (jqc/pyscf/rks.py:446-493, eval_rho.cu:93-106): AO pairs whose estimate lies in
    [cutoff_fp32, cutoff_fp64) go through the FP32 MFMA.  With the default DFT cutoffs of apply() (1e-13 / 1e-6) the result
    must stay within the reference's 1e-7 of the all-FP64 oracle, and must not be bit-identical to the all-FP64 run (the
    FP32 path really ran)."""
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import rks
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dft
    mol = mole.Mole(atom=benzene_like(), basis="def2-svp")          # two rings 14 Bohr apart: many weak AO pairs per block
    lay = BasisLayout.from_mol(mol, alignment=1)
    grids = Grids(mol, 8192)
    _, rho64, vxc64 = rks.generate_rks_kernel(lay, cutoff_fp64=1e-13, cutoff_fp32=1e-13)
    _, rho_m, vxc_m = rks.generate_rks_kernel(lay, cutoff_fp64=1e-6, cutoff_fp32=1e-13)
    np.random.seed(4)
    c = np.random.rand(mol.nao, mol.nelectron // 2) - 0.5
    dm = 2 * c @ c.T
    wv = np.random.rand(4, 8192)
    ref_r = dft.eval_rho(lay, grids.coords, dm, "GGA")
    ref_v = dft.eval_vxc(lay, grids.coords, wv, "GGA")
    for fn, arg, ref in ((rho64, dm, ref_r), (rho_m, dm, ref_r), (vxc64, wv, ref_v), (vxc_m, wv, ref_v)):
        got = fn(mol, grids, "GGA", arg).cpu().numpy()
        assert np.abs(got - ref).max() < 1e-7 * max(1.0, np.abs(ref).max())
    d_r = np.abs(rho_m(mol, grids, "GGA", dm).cpu().numpy() - rho64(mol, grids, "GGA", dm).cpu().numpy()).max()
    d_v = np.abs(vxc_m(mol, grids, "GGA", wv).cpu().numpy() - vxc64(mol, grids, "GGA", wv).cpu().numpy()).max()
    assert 0.0 < d_r < 1e-7 * np.abs(ref_r).max() and 0.0 < d_v < 1e-7 * np.abs(ref_v).max()


def benzene_like():
    from conftest import benzene_atoms
    ring = benzene_atoms()
    return ring + [(sym, (x + 7.4, y + 0.5, z + 1.0)) for sym, (x, y, z) in ring]


def test_vv10_kernel_and_driver():
    from joltqc_amd.pyscf import rks
    from oracle import dft
    mol, lay, grids, rho_k, _ = _setup(H2O, "def2-svp", False, 1500)
    np.random.seed(5)
    c = np.random.rand(mol.nao, 5) - 0.5
    dm = 2 * c @ c.T
    rho = rho_k(mol, grids, "GGA", dm)
    rho_h = rho.cpu().numpy()
    pars = (6.0, 0.01)                          # wB97M-V: b = 6.0, C = 0.01
    e_ref, v_ref = dft.vv10nlc(rho_h, grids.coords, rho_h, grids.weights, grids.coords, pars)
    e64, v64 = rks.vv10nlc(rho, grids.coords, rho, grids.weights, grids.coords, pars, dtype=np.float64)
    assert np.abs(e64.cpu().numpy() - e_ref).max() < 1e-10 * max(1.0, np.abs(e_ref).max())
    assert np.abs(v64.cpu().numpy() - v_ref).max() < 1e-9 * max(1.0, np.abs(v_ref).max())
    e32, v32 = rks.vv10nlc(rho, grids.coords, rho, grids.weights, grids.coords, pars)     # reference default: fp32 inner loop
    assert np.abs(e32.cpu().numpy() - e_ref).max() < 2e-5 * max(1.0, np.abs(e_ref).max())
    assert np.abs(v32.cpu().numpy() - v_ref).max() < 2e-4 * max(1.0, np.abs(v_ref).max())


def test_vv10_pair_sums_kernel_modes():
    """``vv10_sums`` / ``jqc_vv10`` directly on synthetic points, uneven outer / inner sizes (inner loop split over several
    workgroups, partial blocks): the packed-FP32 kernel with the reference's denominator test (mode 1), without it (mode 3,
    chosen by ``vv10_sums`` when every K, Kp >= 1e-3) and the FP64 kernel against a NumPy restatement of dft/vv10.cu:86-117;
    points with K = Kp = 0 on top of each other (denominator 0) must contribute nothing, as in the reference."""
    import torch
    from joltqc_amd.backend import lib as L
    from joltqc_amd.pyscf import rks
    dev = L.require_gpu()
    rng = np.random.default_rng(11)
    no, ni = 3 * 256, 7 * 256

    def make(kmin):
        outer = np.vstack([rng.random((3, no)) * 6, rng.random((1, no)) + 0.5, rng.random((1, no)) + kmin])
        inner = np.vstack([rng.random((3, ni)) * 6, rng.random((1, ni)) + 0.5, rng.random((1, ni)) + kmin, rng.random((1, ni)) * 1e-2])
        return outer, inner

    def ref(outer, inner):
        d = outer[:3, :, None] - inner[:3, None, :]
        R2 = (d ** 2).sum(0)
        g = outer[3][:, None] * R2 + outer[4][:, None]
        gp = inner[3][None, :] * R2 + inner[4][None, :]
        gt = g + gp
        den = gp * (g * gt) ** 2
        T = np.where(den > 1e-30, inner[5][None, :] / np.where(den > 1e-30, den, 1.0), 0.0)
        return np.stack([-1.5 * (T * g * gt).sum(1), (T * (g + gt)).sum(1), (T * R2 * (g + gt)).sum(1)])

    outer, inner = make(0.5)
    want = ref(outer, inner)
    o_d, i_d = torch.from_numpy(outer).to(dev), torch.from_numpy(inner).to(dev)
    got64 = rks.vv10_sums(o_d, i_d, fp32=False).cpu().numpy()
    assert np.abs(got64 - want).max() < 1e-11 * np.abs(want).max()
    got32 = rks.vv10_sums(o_d, i_d, fp32=True).cpu().numpy()            # K >= 0.5: mode 3
    # the inner loop of the packed kernel is split over several workgroups here (7 inner blocks): the shares are added in a fixed
    # order (scratch + vv10_reduce_shares, no atomics), so a second run gives the same bits
    assert np.array_equal(got32, rks.vv10_sums(o_d, i_d, fp32=True).cpu().numpy())
    assert np.abs(got32 - want).max() < 2e-5 * np.abs(want).max()
    lib = L.lib()
    out = torch.empty((3, no), dtype=torch.float64, device=dev)
    L.check(lib.jqc_vv10(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), i_d[:3].contiguous().data_ptr(),
                         o_d[:3].contiguous().data_ptr(), i_d[3].data_ptr(), o_d[3].contiguous().data_ptr(),
                         o_d[4].contiguous().data_ptr(), i_d[4].data_ptr(), i_d[5].data_ptr(), ni, no, 1, L.stream_ptr()))
    assert np.abs(out.cpu().numpy() - got32).max() < 1e-12 * np.abs(want).max()      # the test changes no value when it cannot fail
    # coinciding points with K = Kp = 0: denominator exactly 0 -> the pair is skipped (mode 1 is chosen: K < 1e-3)
    outer[4, :5] = 0.0
    inner[4, :5] = 0.0
    inner[:3, :5] = outer[:3, :5]
    want = ref(outer, inner)
    got = rks.vv10_sums(torch.from_numpy(outer).to(dev), torch.from_numpy(inner).to(dev), fp32=True).cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() < 2e-5 * np.abs(want).max()


def test_incremental_nr_rks_with_slater_exchange():
    """nr_rks plumbing (incremental rho / V_xc caches) with an analytic LDA functional standing in for libxc."""
    import torch
    from joltqc_amd.pyscf import rks
    from oracle import dft
    mol, lay, grids, _, _ = _setup(H2O, "def2-svp", False, 2048)
    nr_rks = rks.generate_nr_rks(lay)
    cx = -0.75 * (3.0 / np.pi) ** (1.0 / 3.0)

    from standin_scf import SlaterNumInt
    ni = SlaterNumInt()                       # NumPy-only, like libxc's NumInt: a device array handed to it raises
    ni._jqc_numpy_boundary = True             # what apply() marks on a CPU object's NumInt; an unmarked one is served device arrays
                                              # (the reference's generate_* closures stay device-side)
    np.random.seed(1)
    c = np.random.rand(mol.nao, 5) - 0.5
    dm1 = 2 * c @ c.T
    dm2 = dm1 + 0.01 * (np.random.rand(mol.nao, mol.nao) - 0.5)
    dm2 = 0.5 * (dm2 + dm2.T)
    for dm in (dm1, dm2):                      # second call goes through the incremental path
        n, e, v = nr_rks(ni, mol, grids, "slater", dm)
        rho = np.maximum(dft.eval_rho(lay, grids.coords, dm, "LDA")[0], 0)
        n_ref = (rho * grids.weights).sum()
        e_ref = (cx * rho ** (4.0 / 3.0) * grids.weights).sum()
        v_ref = dft.eval_vxc(lay, grids.coords, 4.0 / 3.0 * cx * rho ** (1.0 / 3.0) * grids.weights, "LDA")
        assert abs(n - n_ref) < 1e-8 * abs(n_ref) and abs(e - e_ref) < 1e-8 * abs(e_ref)
        v = v.cpu().numpy() if hasattr(v, "cpu") else v
        assert np.abs(v - v_ref).max() < 1e-8 * np.abs(v_ref).max()


def test_rks_scf_through_apply_matches_cpu_oracle_scf():
    """BASELINE config 3 in miniature: apply() on an RKS object (Slater exchange stand-in for libxc), full SCF on
    the GPU path vs the same SCF driven by the CPU oracle (dense rho / V_xc / J) on the same grid."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    from joltqc_amd.pyscf.rks import tag_array
    from standin_scf import RKS, Grids as G, SlaterNumInt
    from oracle import dense, dft
    mol = mole.Mole(atom=H2O, basis="def2-svp")
    lay = BasisLayout.from_mol(mol)
    S, T, V = dense.int1e_mol(lay, mol)
    rng = np.random.default_rng(7)
    at = mol.atom_coords()
    coords = at[rng.integers(0, 3, 6000)] + rng.normal(0, 1.0, (6000, 3))
    coords = coords[np.lexsort(coords.T)]
    weights = np.full(6000, 0.004)
    # all-FP64 grid path for the 1e-8 comparison; the default DFT config (cutoff_fp64 = 1e-6, reference
    # __init__.py:100-118) sends the weak AO pairs through the FP32 MFMA and is checked to the reference's own 1e-6 below
    cfg = jp.get_default_config()
    cfg["dft"] = {"cutoff_fp32": 1e-13, "cutoff_fp64": 1e-13}
    mf = jp.apply(RKS(mol, T + V, S, G(coords, weights)), cfg)
    assert mf._joltqc_applied and mf.get_veff.__func__.__name__ == "get_veff"
    e_gpu = mf.kernel()
    e_default = jp.apply(RKS(mol, T + V, S, G(coords, weights))).kernel()

    ref = RKS(mol, T + V, S, G(coords, weights))
    q = dense.canonical_quartets(lay)
    cx = SlaterNumInt.CX

    def veff_cpu(self, mol_=None, dm=None, dm_last=0, vhf_last=0, hermi=1):
        rho = np.maximum(dft.eval_rho(lay, coords, dm, "LDA")[0], 0)
        exc = float((cx * rho ** (4.0 / 3.0) * weights).sum())
        vxc = dft.eval_vxc(lay, coords, 4.0 / 3.0 * cx * rho ** (1.0 / 3.0) * weights, "LDA")
        vj, _ = dense.get_jk(lay, dm, 1, with_k=False, quartets=q)
        return tag_array(vxc + vj, ecoul=0.5 * float(np.einsum("ij,ji->", dm, vj)), exc=exc, vj=vj, vk=None)
    from types import MethodType
    ref.get_veff = MethodType(veff_cpu, ref)
    e_cpu = ref.kernel()
    assert mf.converged and ref.converged
    assert abs(e_gpu - e_cpu) < 1e-8, e_gpu - e_cpu
    assert abs(e_default - e_cpu) < 1e-6, e_default - e_cpu


@pytest.mark.parametrize("xc_code,e_ref", [("lda,vwn5", -75.9046410402), ("pbe", -76.3800182418), ("b3lyp", -76.4666495594),
                                           ("wb97", -76.4486274326)])
def test_reference_dft_energies_through_apply(kats, xc_code, e_ref):
    """The reference's own known answers for the grid path (jqc/pyscf/tests/test_dft.py:75-103: H2O / def2-TZVPP, tolerance
    1e-5): RKS through ``apply()`` -- device J (pair backend) / J and K (tiled kernels; omega-B97: the long-range K-only
    builds), rho_fun / vxc_fun on the MFMA kernels with the default precision windows, the closed-form functional standing in
    for libxc, Becke grid of gto/grids.py."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.gto.grids import Grids
    from joltqc_amd.pyscf.basis import BasisLayout
    from standin_scf import RKS, ClosedFormNumInt
    from oracle import dense
    mol = mole.Mole(atom=kats["h2o_def2tzvpp"]["atom"], basis="def2-tzvpp")
    S, T, V = dense.int1e_mol(BasisLayout.from_mol(mol), mol)
    mf = jp.apply(RKS(mol, T + V, S, Grids(mol, 90, 24), xc=xc_code, numint=ClosedFormNumInt()))
    e = mf.kernel()
    assert mf.converged
    assert abs(e - e_ref) < 1e-5, e - e_ref          # the reference's bar (default mixed-precision windows included)
    cfg = jp.get_default_config()
    cfg["dft"] = {"cutoff_fp32": 1e-13, "cutoff_fp64": 1e-13}
    mf64 = jp.apply(RKS(mol, T + V, S, Grids(mol, 90, 24), xc=xc_code, numint=ClosedFormNumInt()), cfg)
    e64 = mf64.kernel()
    assert abs(e64 - e_ref) < 1e-6, e64 - e_ref      # all-FP64 grid path: the CPU oracle's bar (tests/test_dft_known_answers.py)


def test_reference_wb97mv_energy_through_apply_pins_tau_and_vv10(kats):
    """jqc/pyscf/tests/test_dft.py:105-109 ("HYB_MGGA_XC_WB97M_V", -76.4334218842, tolerance 1e-5 there; BASELINE config 4's
    functional) through ``apply()``: the meta-GGA instantiations of rho_mfma_kernel / vxc_mfma_kernel (tau branch of reference
    eval_rho.cu:328-377 / eval_vxc.cu:373-383), ``nr_nlc_vxc`` on its own NLC grid (pyscf/rks.py:670-712) with the packed-FP32
    vv10_kernel (vv10.cu:89-107), and the range-separated get_veff with a short-range AND a long-range exact-exchange fraction
    (full-range K scaled by 0.15 + long-range K-only build scaled by 0.85, omega = 0.3).  The functional is the closed form of
    oracle/xc.py standing in for libxc; the CPU oracle reproduces the same number to 4e-8 (tests/test_dft_known_answers.py)."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.gto.grids import Grids
    from joltqc_amd.pyscf.basis import BasisLayout
    from standin_scf import RKS, ClosedFormNumInt
    from oracle import dense
    mol = mole.Mole(atom=kats["h2o_def2tzvpp"]["atom"], basis="def2-tzvpp")
    S, T, V = dense.int1e_mol(BasisLayout.from_mol(mol), mol)
    e_ref = -76.4334218842

    def run(cfg):
        mf = RKS(mol, T + V, S, Grids(mol, 90, 24), xc="HYB_MGGA_XC_WB97M_V", numint=ClosedFormNumInt(), nlcgrids=Grids(mol, 50, 14))
        assert mf.do_nlc()
        mf = jp.apply(mf, cfg)
        patched, n_nlc = mf._numint.nr_nlc_vxc, [0]

        def counted(*a, **k):
            n_nlc[0] += 1
            return patched(*a, **k)
        mf._numint.nr_nlc_vxc = counted
        e = mf.kernel()
        assert mf.converged and n_nlc[0] >= mf.cycles, "nr_nlc_vxc of apply() was not on the path"
        return e, mf
    e, mf = run(None)                                 # default windows (FP32 MFMA band, FP32 VV10 inner loop)
    assert abs(e - e_ref) < 1e-5, e - e_ref
    cfg = jp.get_default_config()
    cfg["dft"] = {"cutoff_fp32": 1e-13, "cutoff_fp64": 1e-13}
    e64, _ = run(cfg)
    assert abs(e64 - e_ref) < 1e-6, e64 - e_ref


def test_build_grids_through_apply_on_a_generated_becke_grid():
    """A17 (reference rks.py:100-177): apply() replaces ``grids.build``; on first use the object's own generator runs (here the
    Becke generator of joltqc_amd/gto/grids.py standing in for PySCF's), the result is sorted into 1-Bohr boxes and padded to a
    multiple of 256 with zero-weight points; the SCF on that grid integrates the density to N_e and a rebuilt grid restarts the
    incremental caches."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.gto.grids import Grids
    from joltqc_amd.pyscf.basis import BasisLayout
    from standin_scf import RKS
    from oracle import dense
    mol = mole.Mole(atom=H2O, basis="def2-svp")
    S, T, V = dense.int1e_mol(BasisLayout.from_mol(mol), mol)
    g = Grids(mol, 45, 14)
    assert g.coords is None
    mf = jp.apply(RKS(mol, T + V, S, g))
    assert hasattr(g, "_jqc_original_build")
    e = mf.kernel()
    assert mf.converged and g._jqc_generation == 1
    n = g.coords.shape[0]
    assert n % 256 == 0 and n > 4000
    raw = Grids(mol, 45, 14).build()
    npad = n - raw.coords.shape[0]
    assert 0 <= npad < 256 and (g.weights[n - npad:] == 0).all() and abs(g.weights.sum() - raw.weights.sum()) < 1e-10 * raw.weights.sum()
    # box-sorted: the box key of the reference's arg_group_grids is non-decreasing along the points (sorting again is the
    # identity), which the generator's atom-by-atom order is not
    from joltqc_amd.pyscf import rks
    assert (rks.arg_group_grids(g.coords[: n - npad]) == np.arange(n - npad)).all()
    assert not (rks.arg_group_grids(raw.coords) == np.arange(n - npad)).all()
    D = np.asarray(mf.make_rdm1())
    rho = mf._numint.get_rho(mol, D, g)
    rho = rho.cpu().numpy() if hasattr(rho, "cpu") else np.asarray(rho)
    assert abs(float((rho[:n] * g.weights).sum()) - mol.nelectron) < 2e-3
    g.build()                                        # a rebuilt grid bumps the generation; the next SCF restarts its increments
    assert g._jqc_generation == 2
    assert abs(mf.kernel() - e) < 1e-9


@pytest.mark.parametrize("cart", [True, False])
def test_rho_and_vxc_general_contraction_basis(cart):
    """Reference tests/test_basis_sets_dft.py (6-31G ... cc-pVTZ: general contractions): rho / vxc (GGA) of a generally contracted
    basis on the GPU against the oracle evaluated on the segmented spelling of the same functions."""
    from conftest import GENERAL_BASIS, SEGMENTED_BASIS
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dft
    mol, lay, grids, rho_k, vxc_k = _setup(H2O, GENERAL_BASIS, cart, 1536)
    lay2 = BasisLayout.from_mol(mole.Mole(atom=H2O, basis=SEGMENTED_BASIS, cart=cart), alignment=1)
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao)
    dm = dm @ dm.T
    rho = rho_k(mol, grids, "GGA", dm).cpu().numpy()
    ref = dft.eval_rho(lay2, grids.coords, dm, "GGA")
    assert np.abs(rho - ref).max() < 1e-9 * np.abs(ref).max()
    wv = np.random.rand(4, 1536)
    v = vxc_k(mol, grids, "GGA", wv).cpu().numpy()
    vref = dft.eval_vxc(lay2, grids.coords, wv, "GGA")
    assert np.abs(v - vref).max() < 1e-9 * np.abs(vref).max()
