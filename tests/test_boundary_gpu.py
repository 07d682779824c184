"""The drop-in boundary on a plain-PySCF-like object (tests/standin_scf.py: NumPy only, TypeError on CUDA tensors):
NumPy in / NumPy out, ``reset`` / ``as_scanner`` after a geometry change (reference jqc/pyscf/tests/test_geom_opt.py:55-354),
range-separated-hybrid and VV10 branches of the RKS ``get_veff`` (reference jqc/pyscf/rks.py:184-260, :661-714).
When PySCF itself is importable (not in this image) the same is repeated on real ``scf.RHF`` / ``dft.RKS`` objects against
the reference's hard-coded energies (jqc/pyscf/tests/test_scf.py:70,77)."""
import os

import numpy as np
import pytest

from conftest import H2O

pytestmark = pytest.mark.gpu
H2O_STRETCHED = "O 0.0 0.0 0.1174; H -0.857 0.0 -0.4696; H 0.757 0.1 -0.4696"


def _int1e(mol):
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dense
    S, T, V = dense.int1e_mol(BasisLayout.from_mol(mol), mol)
    return T + V, S


def _oracle_rhf_energy(mol):
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dense
    from standin_scf import RHF
    lay = BasisLayout.from_mol(mol)
    q = dense.canonical_quartets(lay)
    mf = RHF(mol, int1e=_int1e)
    mf.get_jk = lambda m, dm, hermi=1, **kw: dense.get_jk(lay, dm, hermi, quartets=q)
    e = mf.kernel()
    assert mf.converged
    return e


def test_numpy_in_numpy_out_on_a_cpu_object():
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from standin_scf import RHF
    mol = mole.Mole(atom=H2O, basis="def2-svp")
    mf = jp.apply(RHF(mol, int1e=_int1e))
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao)
    dm = dm @ dm.T
    vj, vk = mf.get_jk(mol, dm, hermi=1)
    assert isinstance(vj, np.ndarray) and isinstance(vk, np.ndarray) and vj.shape == dm.shape
    assert isinstance(mf.get_j(mol, dm, hermi=1), np.ndarray) and isinstance(mf.get_k(mol, dm, hermi=1), np.ndarray)
    vhf = mf.get_veff(mol, dm)
    assert isinstance(vhf, np.ndarray) and np.abs(vhf - (vj - 0.5 * vk)).max() < 1e-10
    vhf2 = mf.get_veff(mol, dm * 1.01, dm_last=dm, vhf_last=vhf)             # incremental branch, NumPy history
    assert isinstance(vhf2, np.ndarray) and np.abs(vhf2 - 1.01 * vhf).max() < 1e-9 * np.abs(vhf).max()


def test_get_j_goes_through_the_pair_backend_by_default():
    """apply() routes J-only calls to the pair-based Coulomb kernels (config jk.pair_j, single GPU, all-FP64): same numbers as
    the J of get_jk, NumPy at the boundary, the first call cross-checked against the tiled J kernels; pair_j = False keeps the
    tiled kernels."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from standin_scf import RHF
    mol = mole.Mole(atom=H2O, basis="def2-tzvpp")
    mf = jp.apply(RHF(mol, int1e=_int1e))
    np.random.seed(3)
    dm = np.random.rand(mol.nao, mol.nao) - 0.5
    dm = dm + dm.T
    vj = mf.get_j(mol, dm, hermi=1)
    st = mf._jqc_pair_jk.stats
    assert isinstance(vj, np.ndarray) and st["pair_launches"] > 0 and st["pair_classes"] > 0
    ref = mf.get_jk(mol, dm, hermi=1)[0]
    assert np.abs(vj - ref).max() < 1e-11 * np.abs(ref).max()
    vj3 = mf.get_j(mol, np.stack([dm, 2 * dm, dm @ dm]), hermi=0)          # several densities, non-symmetric one included
    ref3 = mf.get_jk(mol, np.stack([dm, 2 * dm, dm @ dm]), hermi=0)[0]
    assert vj3.shape == (3,) + dm.shape and np.abs(vj3 - ref3).max() < 1e-11 * np.abs(ref3).max()
    cfg = jp.get_default_config()
    cfg["jk"]["pair_j"] = False
    mf2 = jp.apply(RHF(mol, int1e=_int1e), cfg)
    assert not hasattr(mf2, "_jqc_pair_jk")
    assert np.abs(mf2.get_j(mol, dm, hermi=1) - ref).max() < 1e-11 * np.abs(ref).max()


def test_reset_and_scanner_follow_a_geometry_change():
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from standin_scf import RHF
    mol1 = mole.Mole(atom=H2O, basis="def2-svp")
    mol2 = mole.Mole(atom=H2O_STRETCHED, basis="def2-svp")
    e1_ref, e2_ref = _oracle_rhf_energy(mol1), _oracle_rhf_energy(mol2)
    assert abs(e1_ref - e2_ref) > 1e-4

    mf = jp.apply(RHF(mol1, int1e=_int1e))
    assert abs(mf.kernel() - e1_ref) < 1e-8
    lay1 = mf.get_jk.layout
    out = mf.reset(mol2)                       # reference create_reset_function: original reset, then apply again
    assert out is mf and mf._joltqc_applied and mf.mol is mol2
    assert mf.get_jk.layout is not lay1
    assert np.allclose(np.unique(mf.get_jk.layout.packed[:, :3], axis=0), np.unique(mol2.atom_coords(), axis=0))
    assert abs(mf.kernel() - e2_ref) < 1e-8
    mf.reset()                                 # reset without a molecule keeps the geometry and stays patched
    assert mf._joltqc_applied and abs(mf.kernel() - e2_ref) < 1e-8

    scanner = jp.apply(RHF(mol1, int1e=_int1e)).as_scanner()
    assert scanner._joltqc_applied
    assert abs(scanner(mol1) - e1_ref) < 1e-8
    assert abs(scanner(mol2) - e2_ref) < 1e-8          # the scanner's reset re-applied the kernels for the new geometry
    assert abs(scanner(mol1) - e1_ref) < 1e-8


def test_rks_reset_reapplies_once_per_reset_without_recursion():
    """Reference tests/test_geom_opt.py::test_reset_rks_object / test_reset_without_recursion: an RKS object is re-patched by
    every reset (new layouts, grid caches restarted), the reset wrapper is installed once however often it runs, and the energy
    after reset(mol2) equals a freshly applied object's."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from standin_scf import RKS, Grids as G
    mol1 = mole.Mole(atom=H2O, basis="def2-svp")
    mol2 = mole.Mole(atom=H2O_STRETCHED, basis="def2-svp")
    rng = np.random.default_rng(7)
    at = mol1.atom_coords()
    coords = at[rng.integers(0, 3, 4096)] + rng.normal(0, 1.0, (4096, 3))
    coords = coords[np.lexsort(coords.T)]
    w = np.full(4096, 0.005)

    def fresh(mol):
        h, S = _int1e(mol)
        return jp.apply(RKS(mol, h, S, G(coords, w), int1e=_int1e))

    ks = fresh(mol1)
    e1 = ks.kernel()
    original = ks._jqc_original_reset
    for _ in range(4):                                   # repeated resets: one wrapper, no growing call chain
        assert ks.reset(mol2) is ks and ks._joltqc_applied and ks._jqc_original_reset is original
    e2 = ks.kernel()
    assert abs(e2 - fresh(mol2).kernel()) < 1e-9 and abs(e1 - e2) > 1e-5
    ks.reset(mol1)
    assert abs(ks.kernel() - e1) < 1e-9


class _RSHNumInt:
    """libxc stand-in for a range-separated hybrid with VV10 (the wB97M-V branch structure): Slater exchange as the
    semilocal part, omega = 0.3, alpha = 1.0, hyb = 0.2, one NLC term (b, C) = (6.0, 0.01)."""
    from standin_scf import SlaterNumInt as _S

    class libxc:
        is_hybrid_xc = staticmethod(lambda xc: True)
        is_nlc = staticmethod(lambda xc: True)

    def _xc_type(self, xc_code):
        return "LDA"

    def eval_xc_eff(self, xc_code, rho, deriv=1, xctype="LDA"):
        return self._S().eval_xc_eff(xc_code, rho, deriv, xctype)

    def rsh_and_hybrid_coeff(self, xc_code, spin=0):
        return 0.3, 1.0, 0.2

    def nlc_coeff(self, xc_code):
        return (((6.0, 0.01), 1.0),)


def test_rks_get_veff_range_separated_hybrid_with_vv10():
    """One ``get_veff`` of an RKS object whose functional is a range-separated hybrid with VV10: semilocal XC + J
    - 1/2 [hyb K + (alpha - hyb) K_lr(omega)] + V_nlc, every piece against the CPU oracle."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    from standin_scf import RKS, Grids as G, SlaterNumInt
    from oracle import dense, dft
    mol = mole.Mole(atom=H2O, basis="def2-svp")
    lay = BasisLayout.from_mol(mol)
    h, S = _int1e(mol)
    rng = np.random.default_rng(7)
    at = mol.atom_coords()
    coords = at[rng.integers(0, 3, 3000)] + rng.normal(0, 1.0, (3000, 3))
    coords = coords[np.lexsort(coords.T)]
    weights = np.full(3000, 0.008)
    ks = RKS(mol, h, S, G(coords, weights), xc="rsh-vv10-standin", numint=_RSHNumInt())
    ks.do_nlc = lambda: True
    ks = jp.apply(ks)
    np.random.seed(2)
    c = np.random.rand(mol.nao, 5) - 0.5
    dm = 2 * c @ c.T
    veff = ks.get_veff(mol, dm)
    assert isinstance(veff, np.ndarray) and isinstance(veff.vj, np.ndarray) and isinstance(veff.vk, np.ndarray)

    cx = SlaterNumInt.CX
    rho = np.maximum(dft.eval_rho(lay, coords, dm, "LDA")[0], 0)
    vxc = dft.eval_vxc(lay, coords, 4.0 / 3.0 * cx * rho ** (1.0 / 3.0) * weights, "LDA")
    exc = float((cx * rho ** (4.0 / 3.0) * weights).sum())
    q = dense.canonical_quartets(lay)
    vj, vk = dense.get_jk(lay, dm, 1, quartets=q)
    _, vk_lr = dense.get_jk(lay, dm, 1, quartets=q, omega=0.3, with_j=False)
    rho4 = dft.eval_rho(lay, coords, dm, "GGA")
    e_nlc, v_nlc = dft.vv10nlc(rho4, coords, rho4, weights, coords, (6.0, 0.01))
    wv = np.empty((4, 3000))
    wv[0] = v_nlc[0] * weights
    wv[1:4] = 2.0 * v_nlc[1] * rho4[1:4] * weights
    vnlc = dft.eval_vxc(lay, coords, wv, "GGA")
    k_tot = 0.2 * vk + (1.0 - 0.2) * vk_lr
    ref = vxc + vnlc + vj - 0.5 * k_tot
    scale = np.abs(ref).max()
    assert np.abs(veff.vj - vj).max() < 1e-9 * scale
    assert np.abs(veff.vk - k_tot).max() < 1e-9 * scale
    # VV10 default inner loop is fp32 (reference vv10.cu): 2e-4 relative on the potential, as in test_vv10_kernel_and_driver
    assert np.abs(np.asarray(veff) - ref).max() < 2e-4 * scale
    e_ref = exc + float((rho4[0] * weights * e_nlc).sum()) - 0.25 * float(np.einsum("ij,ji->", dm, k_tot))
    assert abs(float(veff.exc) - e_ref) < 2e-5 * abs(e_ref)
    assert abs(float(veff.ecoul) - 0.5 * float(np.einsum("ij,ji->", dm, vj))) < 1e-9 * abs(float(veff.ecoul))


def test_real_pyscf_objects_when_pyscf_is_importable(kats):
    """On a box with PySCF: apply() on pyscf.scf.RHF / pyscf.dft.RKS reproduces the reference's own numbers
    (jqc/pyscf/tests/test_scf.py:70,77); skipped where PySCF is absent (this image)."""
    pyscf = pytest.importorskip("pyscf")
    import joltqc_amd.pyscf as jp
    k = kats["h2o_def2tzvpp"]
    mol = pyscf.M(atom=k["atom"], basis="def2-tzvpp", verbose=0)
    mf = jp.apply(mol.RHF())
    e = mf.kernel()
    assert abs(e - k["e_rhf_sph"]) < 1e-8
    ks = jp.apply(mol.RKS(xc="b3lyp"))
    ks.grids.level = 3
    e_ks = ks.kernel()
    ref = mol.RKS(xc="b3lyp")
    ref.grids.level = 3
    assert abs(e_ks - ref.kernel()) < 1e-7
    scanner = jp.apply(mol.RHF()).as_scanner()
    mol2 = pyscf.M(atom=H2O_STRETCHED, basis="def2-tzvpp", verbose=0)
    assert abs(scanner(mol2) - mol2.RHF().kernel()) < 1e-8


def _parallel_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    # two ranks on ONE GPU: RCCL refuses duplicate devices, gloo stages the device tensors through the host; everything
    # above the collectives (announce + broadcast, sharded plans and block ranges, one all-reduce per call) is the same
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import parallel as par
    from standin_scf import RHF, RKS, Grids as G
    mol = mole.Mole(atom=H2O, basis="def2-svp")
    cfg = dict(jp.get_default_config(), parallel=True)
    out = {}
    mf = jp.apply(RHF(mol, int1e=_int1e), cfg)
    rng = np.random.default_rng(7)
    at = mol.atom_coords()
    coords = at[rng.integers(0, 3, 6000)] + rng.normal(0, 1.0, (6000, 3))
    coords = coords[np.lexsort(coords.T)]
    h, S = _int1e(mol)
    ks = jp.apply(RKS(mol, h, S, G(coords, np.full(6000, 0.004))), cfg)
    if rank == 0:
        out["e_rhf"] = mf.kernel()
        out["n_rhf"] = mf.get_jk.quartet_counts()[0]
        out["dm"] = np.asarray(mf.make_rdm1())
        out["ejk"] = mf._jqc_jk_energy_per_atom(mol, out["dm"])                       # forces: the quartet queue is shared too
        out["n_grad"] = mf._jqc_jk_energy_per_atom.quartet_count()
        par.stop()
        out["e_rks"] = ks.kernel()
        out["blocks"] = ks._numint.nr_rks.__func__.gcache.ngrids_pad // 256
        par.stop()
        q.put((0, out))
    else:
        n1 = par.serve(mf)
        out["n_rhf"] = mf._jqc_parallel[par.OP_JK].quartet_counts()[0]
        out["n_grad"] = mf._jqc_parallel[par.OP_GRADJK].quartet_count()
        n2 = par.serve(ks)
        out["calls"] = (n1, n2)
        out["range"] = ks._jqc_parallel[par.OP_RHO][0][0].stats.get("block_range")
        q.put((rank, out))
    dist.destroy_process_group()


def test_two_ranks_through_apply_rank0_owns_the_object():
    """SURVEY 8e process model: rank 0 runs the SCF on the patched object, rank 1 mirrors every call from
    ``parallel.serve``; the quartet work (J/K) and the grid blocks (rho / V_xc) are shared, one all-reduce per call."""
    import socket
    import torch.multiprocessing as mp
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from standin_scf import RHF, RKS, Grids as G
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_parallel_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    mol = mole.Mole(atom=H2O, basis="def2-svp")
    mf = jp.apply(RHF(mol, int1e=_int1e))
    e_rhf = mf.kernel()
    n_all = mf.get_jk.quartet_counts()[0]
    rng = np.random.default_rng(7)
    at = mol.atom_coords()
    coords = at[rng.integers(0, 3, 6000)] + rng.normal(0, 1.0, (6000, 3))
    coords = coords[np.lexsort(coords.T)]
    h, S = _int1e(mol)
    e_rks = jp.apply(RKS(mol, h, S, G(coords, np.full(6000, 0.004)))).kernel()
    assert abs(res[0]["e_rhf"] - e_rhf) < 1e-9 and abs(res[0]["e_rks"] - e_rks) < 1e-9
    assert res[0]["n_rhf"] + res[1]["n_rhf"] == n_all and min(res[0]["n_rhf"], res[1]["n_rhf"]) > 0
    assert res[1]["calls"][0] > 5 and res[1]["calls"][1] > 10            # every SCF iteration was mirrored
    ejk = mf._jqc_jk_energy_per_atom(mol, res[0]["dm"])
    assert np.abs(res[0]["ejk"] - ejk).max() < 1e-9 * np.abs(ejk).max()
    assert res[0]["n_grad"] + res[1]["n_grad"] == mf._jqc_jk_energy_per_atom.quartet_count() and min(res[0]["n_grad"], res[1]["n_grad"]) > 0
    b0, b1 = res[1]["range"]
    assert 0 < b0 < b1 == res[0]["blocks"]                                  # rank 1 took the upper range of the grid blocks


def _rccl_worker(port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))       # backend "nccl" IS RCCL on ROCm
    from joltqc_amd.pyscf import parallel as par
    assert par.world() == (0, 1) and par._device().type == "cuda"
    # the collectives of the multi-GPU path on device buffers of its own sizes: header + density broadcast (parallel.py), the ONE
    # all-reduce of the raw [vj; vk] of config 5 (2 x 4 400^2 doubles = 310 MB, pyscf/jk.py)
    h = par._bcast_header([par.OP_JK, 2, 1, 4400, 1, 1, 1, 0.0])
    d = par._bcast_matrix(torch.ones((64, 64), dtype=torch.float64), (64, 64))
    fock = torch.full((2, 4400, 4400), 0.5, dtype=torch.float64, device="cuda")
    dist.all_reduce(fock)
    torch.cuda.synchronize()
    q.put((h[:4], float(d.sum()), float(fock.sum()), d.is_cuda))
    dist.destroy_process_group()


def test_rccl_initialises_and_runs_the_collectives_of_the_multi_gpu_path():
    """One GPU per box here, so the N > 1 path runs on gloo in this suite; what CAN be exercised on hardware is RCCL itself:
    ``init_process_group("nccl")`` (world size 1), the header / matrix broadcasts of pyscf/parallel.py on device tensors and the
    all-reduce of a Fock-sized buffer -- the calls ``bench.py --gpus N`` and ``apply(mf, {"parallel": True})`` make."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    head, dsum, fsum, on_dev = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0
    assert head == [1.0, 2.0, 1.0, 4400.0] and dsum == 64.0 * 64.0 and on_dev
    assert abs(fsum - 0.5 * 2 * 4400 * 4400) < 1e-3
