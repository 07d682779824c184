"""GPU parity tests of the J/K path (through the C-ABI library) against the CPU oracle.
Modelled on the reference's jqc/pyscf/tests/test_jk.py and test_scf.py."""
import numpy as np
import pytest

from conftest import H2O, H2_BOHR, benzene_atoms

pytestmark = pytest.mark.gpu


def _setup(atom, basis, cart=False, unit="angstrom", cut64=1e-13, cut32=1e-13):
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    mol = mole.Mole(atom=atom, basis=basis, cart=cart, unit=unit)
    from joltqc_amd.constants import tile_width
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    return mol, lay, jkmod.generate_jk_kernel(lay, cutoff_fp64=cut64, cutoff_fp32=cut32)


def _dm(nao, seed=9, n=None):
    np.random.seed(seed)
    if n is None:
        d = np.random.rand(nao, nao)
        return d @ d.T
    d = np.random.rand(n, nao, nao)
    return np.einsum("nij,nkj->nik", d, d)


def _np(x):
    return x.detach().cpu().numpy()


def test_native_library_is_loaded():
    from joltqc_amd.backend import lib as L
    L.lib()
    with open("/proc/self/maps") as f:
        assert "libjqc_hip.so" in f.read()


@pytest.mark.parametrize("cart", [True, False])
def test_jk_h2_tzvpp(cart):
    # reference test_jk.py:62-103 (H2, def2-TZVPP, Bohr, seed 9, tol 1e-7); bar here 1e-9
    from oracle import dense
    mol, lay, get_jk = _setup(H2_BOHR, "def2-tzvpp", cart, unit="B")
    dm = _dm(mol.nao)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    assert vj.shape == dm.shape and vk.shape == dm.shape
    assert np.abs(_np(vj) - rj).max() < 1e-9
    assert np.abs(_np(vk) - rk).max() < 1e-9


def test_jk_multiple_dms_and_j_only_k_only():
    from oracle import dense
    mol, lay, get_jk = _setup(H2O, "def2-svp")
    dm = _dm(mol.nao, n=3)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    assert vj.shape == dm.shape
    assert np.abs(_np(vj) - rj).max() < 1e-9 and np.abs(_np(vk) - rk).max() < 1e-9
    vj1, vk0 = get_jk(mol, dm[0], hermi=1, with_k=False)
    vj0, vk1 = get_jk(mol, dm[0], hermi=1, with_j=False)
    assert vk0 == 0 and vj0 == 0
    assert np.abs(_np(vj1) - rj[0]).max() < 1e-9 and np.abs(_np(vk1) - rk[0]).max() < 1e-9


def test_jk_hermi0_and_long_range():
    from oracle import dense
    mol, lay, get_jk = _setup(H2O, "def2-svp")
    np.random.seed(3)
    dm = np.random.rand(mol.nao, mol.nao)          # non-symmetric
    vj, vk = get_jk(mol, dm, hermi=0)
    rj, rk = dense.get_jk(lay, dm, hermi=0)
    assert np.abs(_np(vj) - rj).max() < 1e-9 and np.abs(_np(vk) - rk).max() < 1e-9
    dms = _dm(mol.nao)
    # a SYMMETRIC matrix passed with the signature's default hermi = 0 takes the one-matrix path (same result; the stacked
    # [D, D^T] call of a non-symmetric matrix costs 1.49x a hermi = 1 call)
    vj0, vk0 = get_jk(mol, dms)                     # (hermi defaults to 0)
    rj, rk = dense.get_jk(lay, dms, hermi=0)
    assert np.abs(_np(vj0) - rj).max() < 1e-9 and np.abs(_np(vk0) - rk).max() < 1e-9
    vj1, vk1 = get_jk(mol, dms, hermi=1)
    assert float((vj0 - vj1).abs().max()) < 1e-12 and float((vk0 - vk1).abs().max()) < 1e-12
    for omega in (0.3, 0.5):                        # reference test_jk.py:171-216
        vj, vk = get_jk(mol, dms, hermi=1, omega=omega)
        rj, rk = dense.get_jk(lay, dms, hermi=1, omega=omega)
        assert np.abs(_np(vj) - rj).max() < 1e-9 and np.abs(_np(vk) - rk).max() < 1e-9
    with pytest.raises(AssertionError):
        get_jk(mol, dms, hermi=1, omega=-0.1)


def test_jk_fp32_and_mixed_precision():
    # reference test_jk.py:105-121 (pure fp32 < 1e-3) and :218-248 (mixed 1e-13/1e-7 < 1e-7)
    from oracle import dense
    mol, lay, _ = _setup(H2O, "def2-svp")
    dm = _dm(mol.nao)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    _, _, jk32 = _setup(H2O, "def2-svp", cut64=1e100, cut32=1e-13)
    vj, vk = jk32(mol, dm, hermi=1)
    n64, n32, _ = jk32.quartet_counts()
    assert n64 == 0 and n32 > 0
    assert np.abs(_np(vj) - rj).max() < 1e-3 and np.abs(_np(vk) - rk).max() < 1e-3
    _, _, jkmix = _setup(H2O, "def2-svp", cut64=1e-7, cut32=1e-13)
    vj, vk = jkmix(mol, dm, hermi=1)
    assert np.abs(_np(vj) - rj).max() < 1e-7 and np.abs(_np(vk) - rk).max() < 1e-7


def test_mixed_precision_with_the_fp32_window_split_off(monkeypatch):
    """The reference's mixed mode proper (jk.py:330-420): estimate in (cutoff_fp32, cutoff_fp64] -> fp32 kernel, above -> fp64
    kernel.  The gfx950 scheme table leaves the window with the fp64 kernel (measured faster); JQC_FP32_WINDOW=1 forces the
    split for every class, which this test keeps correct."""
    from oracle import dense
    monkeypatch.setenv("JQC_FP32_WINDOW", "1")
    # (a water molecule has no quartet below 1e-7: the window is widened to 1e-4 so that both kernels get work)
    mol, lay, jkmix = _setup(H2O, "def2-svp", cut64=1e-4, cut32=1e-13)
    dm = _dm(mol.nao)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    vj, vk = jkmix(mol, dm, hermi=1)
    n64, n32, _ = jkmix.quartet_counts()
    assert n64 > 0 and n32 > 0
    assert n64 + n32 <= len(dense.canonical_quartets(lay))
    assert np.abs(_np(vj) - rj).max() < 1e-7 and np.abs(_np(vk) - rk).max() < 1e-7


def test_mixed_precision_split_by_tile_pairs(monkeypatch):
    """Mixed precision as this build applies it (pyscf/jk.py build_tile_plan): the tile pairs whose bound is at or below cutoff_fp64
    go to the FP32 kernel of their class, everything else to the FP64 kernel, no tile pair to both.  Forced on for every class on two
    benzene rings 14 Bohr apart (many weak shell pairs): both parts get work, every quartet is evaluated exactly once, and the
    result holds the reference's bar for mixed precision (1e-7, jqc/pyscf/tests/test_jk.py:218-248)."""
    from oracle import dense
    from conftest import benzene_atoms
    ring = benzene_atoms()
    atoms = ring + [(sym, (x + 7.4, y + 0.5, z + 1.0)) for sym, (x, y, z) in ring]
    monkeypatch.setenv("JQC_FP32_TILE_SPLIT", "1")
    mol, lay, jkmix = _setup(atoms, "def2-svp", cut64=1e-7, cut32=1e-13)
    np.random.seed(4)
    c = np.random.rand(mol.nao, mol.nelectron // 2) - 0.5
    dm = 2 * c @ c.T / mol.nao
    vj, vk = jkmix(mol, dm, hermi=1)
    n64, n32, _ = jkmix.quartet_counts()
    monkeypatch.setenv("JQC_FP32_TILE_SPLIT", "0")
    monkeypatch.setenv("JQC_FP32_WINDOW", "0")            # (nor the per-quartet split a few classes use by default)
    _, _, jk64 = _setup(atoms, "def2-svp", cut64=1e-7, cut32=1e-13)
    rj, rk = jk64(mol, dm, hermi=1)                       # the FP64 kernels on both windows (checked against the oracle elsewhere)
    m64, m32, _ = jk64.quartet_counts()
    assert n32 > 0 and n64 > 0 and m32 == 0 and n64 + n32 == m64, (n64, n32, m64, m32)
    assert float((vj - rj).abs().max()) < 1e-7 and float((vk - rk).abs().max()) < 1e-7
    assert float((vj - rj).abs().max()) > 0.0             # (the FP32 kernels really ran)
    oj, ok = dense.get_jk(lay, dm, hermi=1, cutoff=1e-15)
    assert np.abs(_np(vj) - oj).max() < 1e-7 and np.abs(_np(vk) - ok).max() < 1e-7


def test_jk_screening_far_apart_atoms():
    # reference test_jk.py:250-276: 100 Bohr apart -> inter-atomic quartets are screened out
    from oracle import dense
    mol, lay, get_jk = _setup("H 0 0 0; H 0 0 100", "def2-tzvpp", unit="B")
    dm = _dm(mol.nao)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    n64, _, _ = get_jk.quartet_counts()
    assert n64 < len(dense.canonical_quartets(lay))
    assert np.abs(_np(vj) - rj).max() < 1e-9 and np.abs(_np(vk) - rk).max() < 1e-9


def test_g_functions_class():
    # LMAX = 4 path (reference test_scf.py:90-108 uses def2-QZVPP H2); inline basis with one g shell per atom
    from oracle import dense
    basis = {"H": [[0, [1.2, 1.0]], [4, [1.1, 1.0]]]}
    mol, lay, get_jk = _setup("H 0 0 0; H 0 0 1.4", basis, unit="B")
    dm = _dm(mol.nao)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    assert np.abs(_np(vj) - rj).max() < 1e-9 and np.abs(_np(vk) - rk).max() < 1e-9


def test_rhf_energy_h2o_tzvpp_through_apply(kats):
    """BASELINE config 1 on the GPU path: apply(mf) on the stand-in RHF object."""
    import joltqc_amd.pyscf as jp
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    from standin_scf import RHF
    from oracle import dense
    k = kats["h2o_def2tzvpp"]
    mol = mole.Mole(atom=k["atom"], basis="def2-tzvpp")
    S, T, V = dense.int1e_mol(BasisLayout.from_mol(mol), mol)
    mf = jp.apply(RHF(mol, T + V, S))
    assert mf._joltqc_applied
    e = mf.kernel()
    assert mf.converged
    assert abs(e - k["e_rhf_sph"]) < 1e-8, e - k["e_rhf_sph"]


def test_benzene_svp_parity_full_size():
    from oracle import dense
    mol, lay, get_jk = _setup(benzene_atoms(), "def2-svp")
    dm = _dm(mol.nao)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    scale = max(np.abs(rj).max(), np.abs(rk).max())
    assert np.abs(_np(vj) - rj).max() < 1e-11 * scale
    assert np.abs(_np(vk) - rk).max() < 1e-11 * scale


def test_ket_chunks_and_workgroup_splits(monkeypatch):
    """Launch geometry must not change the result: long ket chunks per workgroup (bra-resident loop, J_ij kept in LDS
    across ket pairs) and candidates of one tile pair dealt to several workgroups."""
    from joltqc_amd.pyscf import jk as jkmod
    from oracle import dense
    mol, lay, _ = _setup(H2O, "def2-tzvpp")
    dm = _dm(mol.nao)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    scale = max(np.abs(rj).max(), np.abs(rk).max())
    for target, kmax, split_below, nsplit in ((1, 7, 0, 1), (1 << 30, 1, 1 << 30, 8), (4, 3, 1 << 30, 4)):
        monkeypatch.setattr(jkmod, "TARGET_WGS", target)
        monkeypatch.setattr(jkmod, "KCHUNK_MAX", kmax)
        monkeypatch.setattr(jkmod, "SPLIT_BELOW_WGS", split_below)
        monkeypatch.setattr(jkmod, "NSPLIT_MAX", nsplit)
        get_jk = jkmod.generate_jk_kernel(lay, cutoff_fp64=1e-13, cutoff_fp32=1e-13)
        vj, vk = get_jk(mol, dm, hermi=1)
        assert np.abs(_np(vj) - rj).max() < 1e-11 * scale, (target, kmax, nsplit)
        assert np.abs(_np(vk) - rk).max() < 1e-11 * scale, (target, kmax, nsplit)
        n64, _, _ = get_jk.quartet_counts()
        assert n64 == len(dense.canonical_quartets(lay))


ORED, PAROOT, NDM2 = 1 << 18, 1 << 19, 1 << 20        # include/jqc_hip.h (round 3: owner reduction, per-root phase A, two DMs)
RSPLIT = lambda code: code << 22                      # (round 4: Rys roots in code + 1 groups through phase A / phase B)
QUAD = 1 << 24                                        # (round 5: one quartet per quad of lanes; classes with a p shell, <= 4 roots)
HB, HEJ = 1 << 29, (lambda code: code << 25)          # (round 6: h form -- bra HRR in phase A, lane = (i component, j group of <= 1/2/3/6))
KW = 1 << 30                                          # (round 6: the two k chunks of a class on different waves of a 512-thread workgroup)


@pytest.mark.parametrize("variant", [0x21, 0x21 | 0x100, 0x21 | 0x400, 0x21 | 0x100 | 0x400, 0x22, 0x32, 0x21 | 0x800,
                                     0x21 | 0x100 | 0x800, 0x1022, 0x2022, 0x3022,
                                     0x921 | ORED, 0x921 | ORED | PAROOT, 0xd21 | ORED | PAROOT, 0x521 | ORED, 0x421 | ORED | PAROOT,
                                     0x30521 | ORED, 0x21 | ORED, 0x10d21 | ORED | PAROOT,
                                     0xd21 | ORED | RSPLIT(1), 0xd21 | ORED | RSPLIT(2), 0x521 | ORED | RSPLIT(1), 0x121 | ORED | RSPLIT(1),
                                     0xd21 | ORED | PAROOT | RSPLIT(1),
                                     0x1022 | QUAD, 0x1122 | QUAD, 0x0132 | QUAD, 0x1032 | QUAD,
                                     0x1122 | QUAD | (1 << 25), 0x1122 | QUAD | (2 << 25) | (2 << 27),      # (+ chunks: 2 over i, 3 over k)
                                     0x521 | ORED | HB, 0x521 | ORED | HB | HEJ(2), 0x121 | ORED | HB | HEJ(1) | RSPLIT(1) | PAROOT,
                                     0x923 | ORED | KW])      # (k chunks on wave groups: the bit is kept where a class has two chunks)
def test_every_kernel_variant_of_the_scheme_table(monkeypatch, variant):
    """The gfx950 scheme table picks one of these variants per class (algorithm | waves per SIMD | Rys table through
    L2 | single TRR buffer | wave-local steps | j in registers | 2, 4, 8 ket pairs per iteration | owner reduction | per-root
    phase A | integral-chunk caps | root groups | one quartet per quad of lanes | h form; include/jqc_hip.h JQC_VARIANT_*): each must give the same J and K.
    Role of the reference's 1q1t == 1qnt cross-check (jqc/backend/data/generate_fragment.py:278-309)."""
    from joltqc_amd.backend import jk as router
    from oracle import dense
    monkeypatch.setenv("JQC_JK_ALGO", "v%d" % variant)
    router.gen_jk_kernel.cache_clear()
    basis = {"O": [[0, [11.0, 0.3], [2.1, 0.5], [0.5, 0.4]], [0, [0.3, 1.0]], [1, [3.4, 0.4], [0.7, 0.7]], [1, [0.2, 1.0]],
                   [2, [1.2, 1.0]], [3, [1.4, 1.0]]],
             "H": [[0, [5.0, 0.3], [0.8, 0.8]], [0, [0.2, 1.0]], [1, [0.8, 1.0]], [2, [1.0, 1.0]]]}
    mol, lay, get_jk = _setup(H2O, basis)
    dm = _dm(mol.nao)
    vj, vk = get_jk(mol, dm, hermi=1)
    rj, rk = dense.get_jk(lay, dm, hermi=1)
    scale = max(np.abs(rj).max(), np.abs(rk).max())
    router.gen_jk_kernel.cache_clear()
    assert np.abs(_np(vj) - rj).max() < 1e-11 * scale
    assert np.abs(_np(vk) - rk).max() < 1e-11 * scale


def test_pair_prefactor_table_matches_closed_form():
    """jqc_pair_table: {c_a c_b exp(-a b/(a+b) R^2), 1/(a+b), a+b} per primitive pair (reference 1q1t.cu:146-171)."""
    from joltqc_amd.pyscf import jk as jkmod
    mol, lay, _ = _setup(H2O, "def2-svp")
    tt = jkmod._TileTables(lay, 0.0)
    tab = _np(tt.pair_tab).reshape(-1, 9, 3)
    off = _np(tt.pp_off).view(np.uint32)
    P = lay.packed
    for t in (0, len(tt.sh_host) // 2, len(tt.sh_host) - 1):
        ish0, jsh0 = int(tt.sh_host[t]) >> 16, int(tt.sh_host[t]) & 0xffff
        wi, wj = int(tt.wij_host[t]) >> 16, int(tt.wij_host[t]) & 0xffff
        for a in range(wi):
            for b in range(wj):
                s1, s2 = P[ish0 + a], P[jsh0 + b]
                r2 = float(((s1[:3] - s2[:3]) ** 2).sum())
                for p1 in range(int(s1[10])):
                    for p2 in range(int(s2[10])):
                        a1, a2 = s1[5 + 2 * p1], s2[5 + 2 * p2]
                        ref = (s1[4 + 2 * p1] * s2[4 + 2 * p2] * np.exp(-a1 * a2 / (a1 + a2) * r2), 1 / (a1 + a2), a1 + a2)
                        got = tab[int(off[t]) + a * wj + b, p1 * 3 + p2]
                        assert np.allclose(got, ref, rtol=1e-13, atol=0)


def _shard_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    # two ranks on ONE GPU: RCCL refuses duplicate devices, gloo stages the CUDA tensor through the host; the code path
    # above the collective (sharded plan, partial Fock matrices, one all_reduce, epilogue on every rank) is the same
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mol, lay, _ = _setup(benzene_atoms(), "def2-svp")
    from joltqc_amd.pyscf import jk as jkmod
    get_jk = jkmod.generate_jk_kernel(lay, cutoff_fp64=1e-13, cutoff_fp32=1e-13, shard=(rank, world))
    vj, vk = get_jk(mol, _dm(mol.nao), hermi=1)
    n64, _, per = get_jk.quartet_counts()
    q.put((rank, _np(vj), _np(vk), n64, len(per)))
    dist.destroy_process_group()


def test_two_ranks_share_the_quartets_and_allreduce_the_fock_matrix():
    """Row (e): quartet work sharded over ranks (cost-aware class/strip split) + ONE all-reduce of the raw J/K."""
    import socket
    import torch.multiprocessing as mp
    from oracle import dense
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    mol, lay, get_jk = _setup(benzene_atoms(), "def2-svp")
    rj, rk = dense.get_jk(lay, _dm(mol.nao), hermi=1)
    scale = max(np.abs(rj).max(), np.abs(rk).max())
    for rank, vj, vk, n64, ncls in res:
        assert np.abs(vj - rj).max() < 1e-11 * scale and np.abs(vk - rk).max() < 1e-11 * scale
    get_jk(mol, _dm(mol.nao), hermi=1)
    n_all, _, _ = get_jk.quartet_counts()
    assert res[0][3] + res[1][3] == n_all          # every dispatched quartet on exactly one rank
    assert min(res[0][3], res[1][3]) > 0.25 * n_all


_CLASS_ORACLE = {}


@pytest.mark.parametrize("mode", ["jk", "j", "k", "lr", "fp32", "k_lr", "fp32_lr", "jk_main", "k_lr_main", "jk_2dm", "jk_2dm_main",
                                  "jk_3dm", "fused", "fused_main", "fused32", "fused_lr"])
def test_every_angular_class_against_the_oracle(mode, monkeypatch):
    """All 140 angular classes s..g, CLASS BY CLASS: the kernel the scheme table selects for the class vs the CPU oracle
    restricted to the quartets of that class (three atoms, artificial s/p/d/f/g basis, the reference autotuner's kind of
    test system, jqc/backend/data/generate_fragment.py:97-114) -- for each of the five builds of a class kernel: J+K,
    J only, K only, long-range (omega = 0.3) and fp32, plus the long-range K-only build the RKS ``get_veff`` asks for
    with range-separated hybrids and the long-range fp32 build.  Launches of this size take the small-launch scheme
    table ("fp64_small"); the ``*_main`` modes force the main table (tuned on 112 atoms) with long ket chunks.
    A miscompiled or racy class kernel shows up here even when the common molecules never reach it ((gg|fp) did).
    ``fused*``: the mixed-precision builds of the lane-per-quartet classes (JQC_VARIANT_MIXED: FP64 phase + packed-FP32 phase, two
    quartets per lane, in ONE launch; reference: an fp32 and an fp64 launch per class, jqc/pyscf/jk.py:293-328) forced on --
    ``fused``: windows 1e-13 / 30, both phases populated; ``fused32``: every quartet through the packed-FP32 phase."""
    import os
    from joltqc_amd.pyscf import jk as jkmod
    from oracle import dense
    shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
              [2, [0.8, 1.0]], [3, [0.9, 1.0]], [4, [1.0, 1.0]]]
    mol, lay, _ = _setup("C 0 0 0; C 0 0.3 2.4; H 1.5 0.2 0.9", {"C": shells, "H": shells}, unit="B")
    dm = _dm(mol.nao)
    allq = dense.canonical_quartets(lay)
    qa = np.asarray(lay.angs)[allq.astype(int)]
    main = mode.endswith("_main")
    if main:
        monkeypatch.setattr(jkmod, "TARGET_WGS", 1)        # no launch counts as small; ket chunks up to KCHUNK_MAX
        mode = mode[:-5]
    two_dm = mode in ("jk_2dm", "jk_3dm")  # several density matrices in one call: the NDM = 2 builds (a pair contracted against ONE
    if two_dm:                         # evaluation of the integrals, reference jk/1q1t.cu:423-638); three = a pair + the odd tail
        dm = np.stack([dm, _dm(mol.nao)[::-1, ::-1].copy() * 0.5 + 0.1 * np.eye(mol.nao)] +
                      ([_dm(mol.nao).T[::-1].copy() * 0.25 + 0.2 * np.eye(mol.nao)] if mode == "jk_3dm" else []))
        dm = 0.5 * (dm + dm.transpose(0, 2, 1))
        mode = "jk"
    with_j, with_k = mode not in ("k", "k_lr"), mode != "j"
    omega = 0.3 if mode.endswith("lr") else None
    cut64, tol = (1e100, 2e-5) if mode.startswith("fp32") else (1e-13, 1e-11)      # fp32: every quartet through the fp32 kernels
    fused = mode.startswith("fused")
    if fused:
        monkeypatch.setenv("JQC_MIXED_FUSED", "1")
        monkeypatch.setenv("JQC_FP32_TILE_SPLIT", "0")        # (the default mixed modes -- FP32 kernels on the low-bound tile pairs
        monkeypatch.setenv("JQC_FP32_WINDOW", "0")            #  or on the per-quartet window -- would take classes away from the fused builds)
        # (three atoms 2-3 Bohr apart and a density of O(10) elements: the estimates Q_ij Q_kl |D| of this system lie around 1e0 - 1e2)
        cut64, tol = (1e20, 2e-5) if mode == "fused32" else (30.0, 2e-5)
    bad, nclass, nfused, nboth = [], 0, 0, 0
    # (three matrices: s..d in the driver-run suite -- the pair builds of every class run in jk_2dm, the odd-tail logic is the same for f, g;
    #  the full s..g three-matrix gate is tests/test_slow_gates.py, run with -m slow on the round's final GPU pass)
    lmax = 2 if dm.ndim == 3 and dm.shape[0] == 3 and os.environ.get("JQC_TEST_FULL_3DM") != "1" else 4
    get_jk = jkmod.generate_jk_kernel(lay, cutoff_fp64=cut64, cutoff_fp32=1e-13)
    try:
        for li in range(lmax + 1):
            for lj in range(li + 1):
                for lk in range(li + 1):
                    for ll in range(lk + 1):
                        sel = (qa == np.array([li, lj, lk, ll])).all(1)
                        if not sel.any():
                            continue
                        nclass += 1
                        key = "%d%d%d%d" % (li, lj, lk, ll)
                        # (the oracle's J and K of a class are evaluated once per (density, omega) and shared by the modes that
                        #  differ only in what the DEVICE does with them: 16 modes, 4 oracle passes -- the suite's time budget)
                        ck = (key, omega, dm.shape)
                        if ck not in _CLASS_ORACLE:
                            _CLASS_ORACLE[ck] = dense.get_jk(lay, dm, hermi=1, quartets=allq[sel], omega=omega)
                        rj, rk = _CLASS_ORACLE[ck]
                        os.environ["JQC_ONLY_CLASS"] = key          # (read by get_jk at call time: one closure, one plan for all classes)
                        vj, vk = get_jk(mol, dm, hermi=1, with_j=with_j, with_k=with_k, omega=omega)
                        sc = max(np.abs(rj).max() if with_j else 0.0, np.abs(rk).max() if with_k else 0.0)
                        err = max(np.abs(_np(vj) - rj).max() if with_j else 0.0, np.abs(_np(vk) - rk).max() if with_k else 0.0) / sc
                        n64, n32, _ = get_jk.quartet_counts()
                        nfused += n32 > 0
                        nboth += n32 > 0 and n64 > 0
                        if not err < tol or n64 + n32 != int(sel.sum()):
                            bad.append((key, err, n64 + n32, int(sel.sum())))
                        if two_dm:          # every further matrix must not be a copy of the first one's result
                            for m in range(1, dm.shape[0]):
                                e2 = np.abs(_np(vk)[m] - rk[m]).max() / max(np.abs(rk[m]).max(), 1e-300)
                                if not e2 < tol:
                                    bad.append((key, "dm%d" % (m + 1), e2))
    finally:
        os.environ.pop("JQC_ONLY_CLASS", None)
    assert nclass == (140 if lmax == 4 else 25) and not bad, bad
    if fused:                   # the packed-FP32 phase did run in the lane-per-quartet classes (46 of the 140 in the main table)
        # (measured: 21 classes, both phases in 9; before nine classes moved to the quad form, which has no fused build: 29-30 and 12)
        assert nfused >= 18 and (mode != "fused" or nboth >= 7), (nfused, nboth)


@pytest.mark.parametrize("cart", [False, True])
def test_general_contraction_basis_against_the_oracle(cart):
    """Reference tests/test_basis_sets_jk.py (6-31G, cc-pVDZ, cc-pVTZ: general contractions, 4-8 primitives): a generally
    contracted basis through the whole path -- decontraction, splitting into <= 3-primitive shells, sorting, padding, the tiled
    kernels, the epilogue -- against the CPU oracle on the SEGMENTED spelling of the same functions (another shell table)."""
    from conftest import GENERAL_BASIS, SEGMENTED_BASIS
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import dense
    m1 = mole.Mole(atom=H2O, basis=GENERAL_BASIS, cart=cart)
    m2 = mole.Mole(atom=H2O, basis=SEGMENTED_BASIS, cart=cart)
    np.random.seed(9)
    dm = np.random.rand(3, m1.nao, m1.nao)
    dm[0] = dm[0] + dm[0].T
    get_jk = jkmod.generate_jk_kernel(BasisLayout.from_mol(m1, alignment=tile_width), cutoff_fp64=1e-13, cutoff_fp32=1e-13)
    vj, vk = get_jk(m1, dm, hermi=0)
    rj, rk = dense.get_jk(BasisLayout.from_mol(m2), dm, hermi=0)
    scale = max(np.abs(rj).max(), np.abs(rk).max())
    assert np.abs(_np(vj) - rj).max() < 1e-11 * scale and np.abs(_np(vk) - rk).max() < 1e-11 * scale


def test_shell_block_max_against_numpy():
    """A5, reference backend/tests/test_linalg_helper.py::test_max_block_pooling: max |D| per shell block over all density
    matrices, irregular shell widths incl. zero-width (padding) shells, float32 out."""
    import torch
    from joltqc_amd.backend import lib as L
    dev = L.require_gpu()
    rng = np.random.default_rng(42)
    loc = np.array([0, 1, 1, 4, 10, 10, 25, 31, 40], dtype=np.int32)         # widths 1, 0, 3, 6, 0, 15, 6, 9
    nb, nao = len(loc) - 1, int(loc[-1])
    for n_dm in (1, 3):
        d = rng.normal(size=(n_dm, nao, nao)) * rng.choice([1e-8, 1.0, 1e3], size=(n_dm, nao, nao))
        ref = np.zeros((nb, nb), dtype=np.float32)
        for i in range(nb):
            for j in range(nb):
                blk = np.abs(d[:, loc[i]:loc[i + 1], loc[j]:loc[j + 1]])
                ref[i, j] = blk.max() if blk.size else 0.0
        dt = torch.from_numpy(d).to(dev).contiguous()
        out = torch.full((nb, nb), -1.0, dtype=torch.float32, device=dev)
        L.check(L.lib().jqc_shell_block_max(dt.data_ptr(), n_dm, nao, torch.from_numpy(loc).to(dev).data_ptr(), nb, out.data_ptr(),
                                          L.stream_ptr()))
        got = out.cpu().numpy()
        assert np.allclose(got, ref, rtol=1e-6, atol=0), np.abs(got - ref).max()


@pytest.mark.parametrize("cart", [False, True])
def test_dm_transforms_of_stacked_matrices(cart):
    """A4, reference test_basis_layout.py::test_3d_array_handling / test_dm_from_mol_dimensions / test_dm_to_mol_dimensions:
    ``dm_from_mol`` / ``dm_to_mol`` on the device, single and stacked matrices, against T D T^T / T^T V T in NumPy."""
    from conftest import GENERAL_BASIS
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf.basis import BasisLayout
    m = mole.Mole(atom=H2O, basis=GENERAL_BASIS, cart=cart)
    lay = BasisLayout.from_mol(m, alignment=4)
    T = lay.transform_matrix()
    rng = np.random.default_rng(2)
    d3 = rng.random((3, m.nao, m.nao))
    out3 = _np(lay.dm_from_mol(d3))
    assert out3.shape == (3, lay.nao, lay.nao) and np.abs(out3 - np.einsum("pi,nij,qj->npq", T, d3, T)).max() < 1e-12
    assert np.abs(_np(lay.dm_from_mol(d3[1])) - out3[1]).max() < 1e-13
    v3 = rng.random((3, lay.nao, lay.nao))
    back3 = _np(lay.dm_to_mol(v3))
    assert back3.shape == (3, m.nao, m.nao) and np.abs(back3 - np.einsum("pi,npq,qj->nij", T, v3, T)).max() < 1e-12
    assert np.abs(_np(lay.dm_to_mol(v3[2])) - back3[2]).max() < 1e-13
