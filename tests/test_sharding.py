"""Multi-GPU path on CPU: the quartet work is dealt to ranks by `build_tile_plan(shard=(rank, world))`; the raw Fock
contributions are summed with ONE all-reduce.  Here two gloo processes evaluate their shares with the CPU oracle."""
import os
import socket

import numpy as np
import pytest

from conftest import H2O


def _layout_and_tables():
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import jk as O
    mol = mole.Mole(atom=H2O, basis="def2-svp")
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    q = np.log(O.schwarz(lay.packed) + 1e-300).astype(np.float32)
    q[lay.pad_id, :] = -100
    q[:, lay.pad_id] = -100
    tt = jkmod._TileTables(lay, 0.0, q_host=q)
    return mol, lay, q, tt


def _plan_quartets(lay, tt, plans, q, log_cut, log_dm):
    """Canonical quartets covered by a plan, applying the kernel's in-tile predicate (jk_tile.hip screening)."""
    from joltqc_amd.constants import tile_width
    nb = lay.nbasis
    out = set()
    for ang, (tab, *_rest) in plans.items():
        tw = [tile_width(l) for l in ang]
        for ij0, nij, kl0, nkl in tab[:, :4]:
            for pij in tt.sh_host[ij0:ij0 + nij]:
                for pkl in tt.sh_host[kl0:kl0 + nkl]:
                    i0, j0, k0, l0 = int(pij) >> 16, int(pij) & 0xffff, int(pkl) >> 16, int(pkl) & 0xffff
                    for a in range(tw[0]):
                        for b in range(tw[1]):
                            for c in range(tw[2]):
                                for d in range(tw[3]):
                                    i, j, k, l = i0 + a, j0 + b, k0 + c, l0 + d
                                    if i >= j and k >= l and i * nb + j >= k * nb + l and q[i, j] + q[k, l] + log_dm > log_cut:
                                        assert (i, j, k, l) not in out
                                        out.add((i, j, k, l))
    return out


def test_shards_partition_the_plan():
    from joltqc_amd.pyscf import jk as jkmod
    from oracle import dense
    mol, lay, q, tt = _layout_and_tables()
    log_cut, log_dm = float(np.log(1e-13)), 0.0
    full = jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True)
    rows_full = sum(p[0].shape[0] for p in full.values())
    parts = [jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True, shard=(r, 3)) for r in range(3)]
    assert sum(sum(p[0].shape[0] for p in part.values()) for part in parts) == rows_full
    qs = [_plan_quartets(lay, tt, part, q, log_cut, log_dm) for part in parts]
    assert not (qs[0] & qs[1]) and not (qs[0] & qs[2]) and not (qs[1] & qs[2])
    allq = qs[0] | qs[1] | qs[2]
    assert allq == _plan_quartets(lay, tt, full, q, log_cut, log_dm)
    # with these cutoffs nothing of H2O/def2-SVP is screened: the union is every canonical quartet
    assert allq == {tuple(int(x) for x in t) for t in dense.canonical_quartets(lay)}


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    from joltqc_amd.pyscf import jk as jkmod
    from oracle import jk as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mol, lay, q, tt = _layout_and_tables()
    log_cut = float(np.log(1e-13))
    plan = jkmod.build_tile_plan(lay, tt, log_cut, 0.0, lambda a: True, shard=(rank, world))
    mine = np.array(sorted(_plan_quartets(lay, tt, plan, q, log_cut, 0.0)), dtype=np.uint16).reshape(-1, 4)
    rng = np.random.default_rng(9)
    dm = rng.random((lay.nao, lay.nao))
    dm = dm @ dm.T
    vj, vk = O.jk_raw(lay.packed, dm, mine)
    fock = torch.from_numpy(np.stack([vj[0], vk[0]]))
    dist.all_reduce(fock)                       # the one collective of the path
    if rank == 0:
        ret["fock"] = fock.numpy()
        ret["dm"] = dm
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_rank():
    import torch.multiprocessing as mp
    from oracle import dense
    from oracle import jk as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context("spawn").Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    mol, lay, q, tt = _layout_and_tables()
    vj, vk = O.jk_raw(lay.packed, ret["dm"], dense.canonical_quartets(lay))
    assert np.abs(ret["fock"][0] - vj[0]).max() < 1e-11
    assert np.abs(ret["fock"][1] - vk[0]).max() < 1e-11


def test_launches_stay_below_the_work_item_limit(monkeypatch):
    """A launch of more than 2^24 - 1 workgroups of 256 is silently truncated by the hardware queue (32-bit work-item count):
    build_tile_plan lengthens the ket chunks until every class fits, whatever KCHUNK_MAX says (here with a toy limit); the
    task rectangles (bra range x ket range) -- i.e. the quartets covered -- do not change, only how many workgroups share them."""
    import math
    from conftest import benzene_atoms
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import jk as O
    mol = mole.Mole(atom=benzene_atoms(), basis="def2-svp")
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    q = np.log(O.schwarz(lay.packed) + 1e-300).astype(np.float32)
    q[lay.pad_id, :] = -100
    q[:, lay.pad_id] = -100
    tt = jkmod._TileTables(lay, 0.0, q_host=q)
    log_cut, log_dm = math.log(1e-13), 2.0
    monkeypatch.setattr(jkmod, "KCHUNK_MAX", 1)
    monkeypatch.setattr(jkmod, "NSPLIT_MAX", 1)
    free = jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True)
    monkeypatch.setattr(jkmod, "MAX_WGS_PER_LAUNCH", 40)
    capped = jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True)
    assert max(p[1] for p in free.values()) > 400                     # the toy limit bites
    for ang, (tab, nblk, _, index) in capped.items():
        f = free[ang][0]
        assert np.array_equal(tab[:, :4], f[:, :4])                  # same rectangles
        nbra = int(tab[:, 1].sum())
        assert nblk <= 40 + nbra, (ang, nblk)                        # ceil() per task row: at most one extra chunk per bra pair
        kch = tab[:, 7] & 0xffff
        assert (kch == kch[0]).all() and kch[0] >= math.ceil(int((f[:, 1] * f[:, 3]).sum()) / 40)
        assert np.array_equal(tab[:, 4], -(-tab[:, 3] // kch)) and index[-1] == tab.shape[0] - 1


@pytest.mark.parametrize("name,basis", [("0112-elongated-nitrogenous", "def2-tzvpp"), ("0425-globular-nitrogenous", "def2-svp")])
def test_predicted_load_is_balanced_for_2_4_8_ranks(name, basis):
    """BASELINE configs 4 / 5 sizes: the cost-aware deal of task rows (``_shard_assign``: measured ns per quartet of every
    class x tile-pair products, heaviest row to the least loaded rank) leaves the predicted load of the slowest rank within
    10 % of the mean for 2, 4 and 8 ranks, every row goes to exactly one rank, and every class has a weight (g classes included)."""
    import math
    from joltqc_amd.backend import jk as router
    from joltqc_amd.constants import tile_width
    from joltqc_amd.gto import mole
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    from oracle import jk as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mol = mole.Mole(atom=mole.read_xyz(os.path.join(root, "joltqc_amd/data/molecules", name + ".xyz")), basis=basis)
    lay = BasisLayout.from_mol(mol, alignment=tile_width)
    q = np.log(O.schwarz(lay.packed) + 1e-300).astype(np.float32)
    q[lay.pad_id, :] = -100
    q[:, lay.pad_id] = -100
    tt = jkmod._TileTables(lay, 0.0, q_host=q)
    log_cut, log_dm = math.log(1e-13), 0.0
    full = jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True)
    nrows = sum(p[0].shape[0] for p in full.values())
    for world in (2, 4, 8):
        parts = [jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True, shard=(r, world)) for r in range(world)]
        load = jkmod.build_tile_plan.last_predicted_load
        assert len(load) == world and max(load) < 1.10 * (sum(load) / world), (world, load)
        assert sum(sum(p[0].shape[0] for p in part.values()) for part in parts) == nrows
        # tile-pair products (~ candidate quartets, NOT cost: a rank with the f-heavy classes gets fewer) per rank: no rank starves
        work = [sum(int((p[0][:, 1].astype(np.int64) * p[0][:, 3]).sum()) for p in part.values()) for part in parts]
        assert min(work) > 0, (world, work)
    cost = router.class_cost_table()
    for a in full:
        assert router.class_key(a) in cost, a
    for l4 in [(4, 0, 0, 0), (4, 4, 4, 4), (4, 3, 2, 1)]:
        assert router.class_key(l4) in cost, l4


def test_precision_split_by_tile_pairs_partitions_the_plan():
    """Mixed precision by tile pairs (``build_tile_plan(log_cut64=...)``): the FP64 part and the FP32 part together cover exactly the
    quartets of the unsplit plan, no quartet twice, and every quartet of the FP32 part has an estimate q_ij + q_kl + log max|D| at or
    below cutoff_fp64 -- the reference's FP32 window (jqc/backend/jk/screen_jk_tasks.cu:241-261), so no quartet the reference would
    evaluate in FP64 is evaluated in FP32."""
    from joltqc_amd.pyscf import jk as jkmod
    mol, lay, q, tt = _layout_and_tables()
    log_cut, log_dm = float(np.log(1e-13)), 0.0
    full = jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True)
    both = 0
    for cut64 in (3.0, 0.3, 1e-2, 1e-7):
        log_cut64 = float(np.log(cut64))
        p64, p32 = jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True, log_cut64=log_cut64, split=lambda a: True)
        q64 = _plan_quartets(lay, tt, p64, q, log_cut, log_dm)
        q32 = _plan_quartets(lay, tt, p32, q, log_cut, log_dm)
        assert not (q64 & q32)
        assert (q64 | q32) == _plan_quartets(lay, tt, full, q, log_cut, log_dm)
        assert all(q[i, j] + q[k, l] + log_dm <= log_cut64 + 1e-5 for i, j, k, l in q32)
        both += bool(q32) and bool(q64)
    assert both >= 1                                # (H2O / def2-SVP: the bounds lie around 1e-1 .. 1e1)
    # classes the predicate leaves out stay whole
    p64, p32 = jkmod.build_tile_plan(lay, tt, log_cut, log_dm, lambda a: True, log_cut64=float(np.log(3.0)), split=lambda a: a[0] < 2)
    assert all(a[0] < 2 for a in p32)
    assert sum(pl[0].shape[0] for a, pl in p64.items() if a[0] >= 2) == sum(pl[0].shape[0] for a, pl in full.items() if a[0] >= 2)
