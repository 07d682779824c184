#!/usr/bin/env python3
"""Headline benchmark: J/K Fock-build wall-time and ERI quartets/s, def2-TZVPP (BASELINE.json).

A "step" is ONE get_jk call (J and K, FP64, hermi=1, one density matrix) on benzene / def2-TZVPP
(BASELINE.json configs[1]); the density matrix is synthetic (``rand; D = R R^T``, seed 9, the
reference's own test convention, jqc/pyscf/tests/test_jk.py:68-71) and resident in HBM before the
timed region.  With N > 1 ranks (torch.distributed, backend nccl = RCCL) the quartet work of the SAME
molecule is dealt round-robin to the ranks and the raw Fock contributions are summed with one
all-reduce per step (strong scaling).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      FP64-VALU roofline of the dominant class kernel (the path is FMA-bound, not HBM- or
                MFMA-bound: SURVEY.md 8d); achieved = algorithmic FLOP of that class per launch /
                HIP-event duration of its launch, measured live in the timed region.
  cpu_baseline  the CPU oracle (a scalar C port of the reference arithmetic) timed on a bounded random
                sample of the same workload's canonical quartets, 1 core.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def benzene_atoms():
    rc, rh = 1.39, 1.39 + 1.09
    out = []
    for k in range(6):
        t = np.pi / 3 * k
        out.append(("C", (rc * np.cos(t), rc * np.sin(t), 0.0)))
        out.append(("H", (rh * np.cos(t), rh * np.sin(t), 0.0)))
    return out


def load_workload(name):
    from joltqc_amd.gto import mole
    if name == "benzene":
        return mole.Mole(atom=benzene_atoms(), basis="def2-tzvpp"), "benzene C6H6 RHF/def2-TZVPP J+K"
    if name == "benzene-spdfg":
        # artificial basis with every angular momentum up to g on every atom (the reference autotuner's test system,
        # jqc/backend/data/generate_fragment.py:97-114): exercises all 140 angular classes
        shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
                  [2, [0.8, 1.0]], [3, [0.9, 1.0]], [4, [1.0, 1.0]]]
        return mole.Mole(atom=benzene_atoms(), basis={"C": shells, "H": shells}), "benzene, artificial s/p/d/f/g basis"
    path = os.path.join(ROOT, "joltqc_amd", "data", "molecules", name + ".xyz")
    return mole.Mole(atom=mole.read_xyz(path), basis="def2-tzvpp"), f"{name} def2-TZVPP J+K"


def cpu_baseline(layout, seconds=12.0):
    """Oracle (scalar C port) on a bounded random sample of canonical quartets of the same layout."""
    from oracle import jk as O
    rng = np.random.default_rng(0)
    real = np.nonzero(~layout.pad_id)[0]
    nb = layout.nbasis
    n = 8000000
    i = rng.choice(real, n); j = rng.choice(real, n); k = rng.choice(real, n); l = rng.choice(real, n)
    i, j = np.maximum(i, j), np.minimum(i, j)
    k, l = np.maximum(k, l), np.minimum(k, l)
    sw = i * nb + j < k * nb + l
    i2, j2, k2, l2 = np.where(sw, k, i), np.where(sw, l, j), np.where(sw, i, k), np.where(sw, j, l)
    q = np.stack([i2, j2, k2, l2], 1).astype(np.uint16)
    dm = rng.random((layout.nao, layout.nao))
    dm = dm + dm.T
    O.jk_raw(layout.packed, dm, q[:2000])          # warm-up / library load
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and done < n:
        m = min(200000, n - done)
        O.jk_raw(layout.packed, dm, q[done:done + m])
        done += m
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "quartets/s", "cores": 1, "kind": "port",
            "sample": f"{done} uniformly random canonical quartets of the same shell table, oracle/jk_oracle.c, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="benzene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from joltqc_amd.constants import tile_width
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    from joltqc_amd.roofline import FP64_VALU_PEAK_TFLOPS, quartet_flops

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # dry runs of the N-rank path on a one-GPU box: JQC_BENCH_BACKEND=gloo JQC_BENCH_ONE_DEVICE=1 (RCCL refuses two
    # ranks on one device); the driver's multi-GPU runs use the defaults: one rank per GPU over RCCL
    backend = os.environ.get("JQC_BENCH_BACKEND", "nccl")
    if os.environ.get("JQC_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    mol, wname = load_workload(args.workload)
    layout = BasisLayout.from_mol(mol, alignment=tile_width)
    np.random.seed(9)
    dm = np.random.rand(mol.nao, mol.nao)
    dm = torch.from_numpy(dm @ dm.T).cuda()
    get_jk = jkmod.generate_jk_kernel(layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13,
                                      shard=(rank, world) if world > 1 else None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        vj, vk = get_jk(mol, dm, hermi=1)
    torch.cuda.synchronize()
    n64, n32, per = get_jk.quartet_counts()
    # dominant class = most algorithmic FLOP among this rank's classes
    flops_by_ang = {}
    for (ang, npr), (a, b) in per.items():
        flops_by_ang[ang] = flops_by_ang.get(ang, 0) + (a + b) * quartet_flops(ang, npr or (1, 1, 1, 1))
    dom = max(flops_by_ang, key=flops_by_ang.get)
    total_flops = sum(flops_by_ang.values())

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vj, vk = get_jk(mol, dm, hermi=1)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    nq = torch.tensor([float(n64 + n32), float(total_flops)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nq)
    dt = float(tmax.item())
    quartets, flops_all = float(nq[0].item()), float(nq[1].item())

    # roofline leg: the dominant class kernel bracketed by HIP events on its own stream, class kernels
    # launched back to back on ONE stream (in the timed region above they overlap on several streams,
    # which makes a single kernel's duration ill-defined); same process, same inputs, right after the loop
    get_jk.set_streams(1)
    get_jk.set_probe(dom)
    for _ in range(max(3, min(args.steps, 10))):
        get_jk(mol, dm, hermi=1)
    torch.cuda.synchronize()
    evs = get_jk.stats.get("probe_events", [])
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs])) if evs else None
    if rank == 0:
        ms = dt / args.steps * 1e3
        achieved = flops_by_ang[dom] / (kern_ms * 1e-3) / 1e12 if kern_ms else None
        out = {
            "metric": "ERI quartets/s (J/K Fock build, def2-TZVPP)", "value": quartets * args.steps / dt,
            "unit": "quartets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "jk_wall_s": ms * 1e-3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wname, "nao": mol.nao, "split_shells": int((~layout.pad_id).sum()),
                       "quartets_per_step": quartets, "model_gflop_per_step": flops_all / 1e9,
                       "whole_path_tflops": flops_all * args.steps / dt / 1e12,
                       "density": "rand(nao,nao) R R^T seed 9", "cutoff": 1e-13,
                       "parallelism": f"quartet work sharded over {world} rank(s) (cost-aware class/strip split) + 1 Fock all-reduce"},
            "roofline": {"bound": "valu_fp64", "kernel": "jk_tile*_%d%d%d%d" % dom,
                         "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VALU_PEAK_TFLOPS if achieved else None,
                         "kernel_ms": kern_ms, "kernel_gflop": flops_by_ang[dom] / 1e9, "traffic": None},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(layout)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
