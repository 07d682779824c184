"""CPU tests (gloo, world size 2) of the multi-GPU driver/worker protocol (joltqc_amd/pyscf/parallel.py):
rank 0 announces every call and broadcasts its matrix, the workers mirror it, partial results meet in one all-reduce.
The compute functions are stand-ins (the kernels need a GPU; tests/test_boundary_gpu.py runs the real ones on two ranks)."""
import socket

import numpy as np


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from joltqc_amd.pyscf import parallel as par
    log = []

    def fake_jk(mol, dm, hermi, vhfopt, with_j, with_k, omega, verbose):
        part = torch.as_tensor(dm, dtype=torch.float64) * (rank + 1.0)          # this rank's share of the "work"
        buf = torch.stack([part, 2.0 * part])
        dist.all_reduce(buf)
        log.append(("jk", hermi, with_j, with_k, omega, tuple(dm.shape)))
        return (buf[0] if with_j else 0), (buf[1] if with_k else 0)

    def fake_grid(mol, grids, xctype, mat, ref=None):
        out = torch.as_tensor(mat, dtype=torch.float64).sum() * (rank + 1.0) * torch.ones(3, dtype=torch.float64)
        dist.all_reduce(out)
        log.append(("grid", xctype, grids, tuple(mat.shape), ref))
        return out

    def fake_sums(outer, inner, fp32):
        out = torch.zeros(3, outer.shape[1], dtype=torch.float64)
        n = outer.shape[1] // world
        out[:, rank * n:(rank + 1) * n] = outer[3:4, rank * n:(rank + 1) * n] * inner[5].sum()
        dist.all_reduce(out)
        return out

    def fake_grad(mol, dm, j_factor=1.0, k_factor=1.0, omega=None, hermi=1, verbose=None):
        out = torch.as_tensor(dm, dtype=torch.float64).sum() * (rank + 1.0) * j_factor * torch.ones((4, 3), dtype=torch.float64)
        dist.all_reduce(out)
        log.append(("grad", float(j_factor), float(k_factor), omega, tuple(dm.shape)))
        return out

    assert par.world() == (rank, world)
    if rank == 0:
        get_jk = par.drive_jk(fake_jk)
        get_jk.return_numpy = True
        np.random.seed(1)
        dm = np.random.rand(5, 5)
        vj, vk = get_jk(None, dm, hermi=1)
        dm3 = np.random.rand(2, 5, 5)
        vj3, vk3 = get_jk(None, dm3, hermi=0, with_j=False, omega=0.3)
        rho = par.drive_grid(fake_grid, par.OP_RHO, 0)(None, "G0", "GGA", np.ones((4, 7)))
        vx = par.drive_grid(fake_grid, par.OP_VXC, 1)(None, "G1", "MGGA", np.ones(6), 0.25)     # (an increment: magnitude of the full matrix)
        sm = par.drive_vv10(fake_sums)(torch.arange(40, dtype=torch.float64).reshape(5, 8), torch.ones(6, 4, dtype=torch.float64), True)
        gr = par.drive_grad_jk(fake_grad)(None, dm3, j_factor=0.5, k_factor=0.25, omega=0.2)
        par.stop()
        q.put((0, dict(vj=vj, vk=vk, vj3=vj3, vk3=vk3, rho=rho.numpy(), vx=vx.numpy(), sm=sm.numpy(), dm=dm, dm3=dm3, gr=gr), log))
    else:
        n = par.serve({par.OP_JK: fake_jk, par.OP_RHO: {0: (fake_grid, None, lambda: "G0")}, par.OP_VXC: {1: (fake_grid, None, "G1")},
                       par.OP_VV10: fake_sums, par.OP_GRADJK: fake_grad})
        q.put((rank, n, log))
    dist.destroy_process_group()


def test_driver_and_worker_mirror_every_call():
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, r0, log0), (_, ncalls, log1) = res
    assert ncalls == 6 and log0 == log1                      # same calls, same arguments, same order on both ranks
    assert isinstance(r0["vj"], np.ndarray) and np.allclose(r0["vj"], 3.0 * r0["dm"]) and np.allclose(r0["vk"], 6.0 * r0["dm"])
    assert r0["vj3"] == 0 and np.allclose(r0["vk3"], 6.0 * r0["dm3"]) and log1[1] == ("jk", 0, False, True, 0.3, (2, 5, 5))
    assert np.allclose(r0["rho"], 28.0 * 3.0) and np.allclose(r0["vx"], 6.0 * 3.0)
    assert log1[2] == ("grid", "GGA", "G0", (4, 7), None) and log1[3] == ("grid", "MGGA", "G1", (6,), 0.25)
    assert np.allclose(r0["sm"], np.tile(np.arange(24, 32) * 4.0, (3, 1)))
    assert isinstance(r0["gr"], np.ndarray) and np.allclose(r0["gr"], 1.5 * r0["dm3"].sum()) and log1[-1] == ("grad", 0.5, 0.25, 0.2, (2, 5, 5))


def test_split_blocks_is_a_balanced_partition():
    from joltqc_amd.pyscf.parallel import split_blocks
    rng = np.random.default_rng(0)
    cost = rng.integers(1, 400, 1000).astype(float) ** 2
    for w in (1, 2, 3, 8):
        cuts = [split_blocks(cost, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(cost)
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(w - 1))
        loads = [cost[a:b].sum() for a, b in cuts]
        assert max(loads) < 1.05 * cost.sum() / w + cost.max()
    assert split_blocks([], 0, 4) == (0, 0)
    assert [split_blocks([5.0], r, 3) for r in range(3)].count((0, 1)) == 1
