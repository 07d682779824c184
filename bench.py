#!/usr/bin/env python3
"""Headline benchmark: J/K Fock-build wall-time and ERI quartets/s, def2-TZVPP (BASELINE.json).

A "step" is ONE get_jk call (J and K, FP64, hermi=1, one density matrix).  Default workload: the 112-atom CHNO
molecule ``0112-elongated-nitrogenous`` with def2-TZVPP -- the Taxol-size stand-in of the north-star target (Taxol
C47H51NO14 has 113 atoms; the reference ships this geometry under benchmarks/molecules) -- ``--workload benzene`` gives
BASELINE.json configs[1].  The density matrix is synthetic (``rand; D = R R^T``, seed 9, the reference's own test
convention, jqc/pyscf/tests/test_jk.py:68-71: dense, i.e. no help from density screening) and resident in HBM before the
timed region.

``--gpus N``: when not already launched by torchrun (WORLD_SIZE unset) this process starts N ranks itself -- child
processes, one per GPU, before anything here touches a GPU -- and rank 0 prints the line.  With N > 1 ranks
(torch.distributed, backend nccl = RCCL) the quartet work of the SAME molecule is dealt to the ranks (cost-aware
class/strip split) and the raw Fock contributions are summed with one all-reduce per step (strong scaling).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      FP64-VALU roofline (the path is FMA-bound, not HBM- or MFMA-bound: SURVEY.md 8d) of the class kernel that
                takes the most TIME (HIP events around every class launch, launches serialised on one stream for that leg);
                achieved = algorithmic FLOP of that class per launch / its event duration; ``whole_path`` = all classes'
                algorithmic FLOP / step wall-time.  ``traffic`` = HBM bytes per launch of that kernel from the committed
                PMC summary (profiles/*pmc_traffic*.json, separate rocprofv3 --pmc passes, FETCH_SIZE doubled as the
                guide prescribes) with the algorithmic bytes beside it; null when no summary covers the kernel.
  cpu_baseline  the CPU oracle (C restatement of the reference arithmetic, OpenMP over os.cpu_count() host cores) on a
                bounded sample drawn from the dispatched class histogram of this workload; PySCF's get_jk is used
                instead when it is importable on the box (it is not in this image).
  parity        max|dJ|, max|dK| of the timed workload's own result against the CPU oracle on a fixed sample of shell blocks
                (every quartet those blocks need: oracle/dense.py:sampled_blocks), outside the timed region.
  grid_path     rho / vxc (GGA) of the DFT grid path on the same molecule and basis with a Becke grid (own generator):
                grid points x AO pairs per second and the fraction of the FP64 MFMA peak (N = 1 only).
  forces        one two-electron gradient (jk_grad kernels) of the SCF-like density of the realistic_density
                leg, next to that leg's J/K build (N = 1 only; one untimed call is not made: the kernels are built ahead of time).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
DEFAULT_WORKLOAD = "0112-elongated-nitrogenous"


def benzene_atoms():
    rc, rh = 1.39, 1.39 + 1.09
    out = []
    for k in range(6):
        t = np.pi / 3 * k
        out.append(("C", (rc * np.cos(t), rc * np.sin(t), 0.0)))
        out.append(("H", (rh * np.cos(t), rh * np.sin(t), 0.0)))
    return out


def load_workload(name):
    """``name`` = benzene | benzene-spdfg | <xyz name under joltqc_amd/data/molecules>[@<basis>] (default basis def2-tzvpp; e.g.
    ``0425-globular-nitrogenous@def2-svp`` = the size and basis of BASELINE config 5)."""
    from joltqc_amd.gto import mole
    name, _, basis = name.partition("@")
    basis = basis or "def2-tzvpp"
    if name == "benzene":
        return mole.Mole(atom=benzene_atoms(), basis="def2-tzvpp"), "benzene C6H6 RHF/def2-TZVPP J+K"
    if name == "benzene-spdfg":
        # artificial basis with every angular momentum up to g on every atom (the reference autotuner's test system,
        # jqc/backend/data/generate_fragment.py:97-114): exercises all 140 angular classes
        shells = [[0, [8.0, 0.2], [1.6, 0.5], [0.4, 0.4]], [0, [0.15, 1.0]], [1, [4.0, 0.3], [0.9, 0.5], [0.25, 0.4]],
                  [2, [0.8, 1.0]], [3, [0.9, 1.0]], [4, [1.0, 1.0]]]
        return mole.Mole(atom=benzene_atoms(), basis={"C": shells, "H": shells}), "benzene, artificial s/p/d/f/g basis"
    path = os.path.join(ROOT, "joltqc_amd", "data", "molecules", name + ".xyz")
    atoms = mole.read_xyz(path)
    el = [ln.split()[0] for ln in atoms.splitlines() if ln.strip()]
    formula = "".join(f"{e}{el.count(e)}" for e in sorted(set(el), key=lambda e: (e != "C", e != "H", e)))
    return mole.Mole(atom=atoms, basis=basis), f"{name} ({formula}, {len(el)} atoms) RHF/{basis.replace('def2-', 'def2-').upper().replace('DEF2', 'def2')} J+K"


def sample_quartets(layout, per_class, n, rng):
    """``n`` canonical quartets drawn from the dispatched histogram {(ang, nprim pattern): count}: the class and the
    primitive pattern follow the histogram, the shells inside the four (l, nprim) groups are uniform."""
    groups = {}
    for s in np.nonzero(~layout.pad_id)[0]:
        groups.setdefault((int(layout.angs[s]), int(layout.nprims[s])), []).append(int(s))
    groups = {k: np.array(v) for k, v in groups.items()}
    keys = [k for k in per_class if k[1] is not None and sum(per_class[k]) > 0]
    w = np.array([float(sum(per_class[k])) for k in keys])
    counts = rng.multinomial(n, w / w.sum())
    nb = layout.nbasis
    out = []
    for (ang, npr), c in zip(keys, counts):
        if c == 0:
            continue
        i, j, k, l = (rng.choice(groups[(ang[x], npr[x])], c) for x in range(4))
        i, j = np.maximum(i, j), np.minimum(i, j)
        k, l = np.maximum(k, l), np.minimum(k, l)
        sw = i * nb + j < k * nb + l
        out.append(np.stack([np.where(sw, k, i), np.where(sw, l, j), np.where(sw, i, k), np.where(sw, j, l)], 1))
    q = np.concatenate(out).astype(np.uint16)
    return q[rng.permutation(len(q))]


def physical_cores():
    """Physical cores this process may run on, capped by the container's CPU quota (one OpenMP thread per core: the oracle's inner
    loops are FP64-FMA bound, a second hardware thread per core adds nothing but cache pressure)."""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                seen.add(f.read().strip())
        except OSError:
            seen.add(str(c))
    n = max(1, len(seen))
    # a container's CPU-time quota (cgroup v2 cpu.max / v1 cfs_quota): the GPU box shows 256 logical CPUs and grants 16 CPUs' worth
    # of time -- more threads than that only time-slice (measured: 128 threads = 9.8x one thread)
    for quota_f, period_f in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            with open(quota_f) as f:
                w = f.read().split()
            if period_f:
                with open(period_f) as f:
                    w.append(f.read().split()[0])
            if w[0] not in ("max", "-1") and float(w[1]) > 0:
                n = max(1, min(n, int(float(w[0]) / float(w[1]) + 0.5)))
            break
        except (OSError, IndexError, ValueError):
            continue
    return n


def cpu_baseline(mol, layout, per_class, seconds=12.0):
    cores = physical_cores()
    try:                                           # the reference's own CPU oracle, when the box has it
        from pyscf import gto as pgto, lib as plib, scf as pscf       # noqa: F401
        have_pyscf = True
    except ImportError:
        have_pyscf = False
    rng = np.random.default_rng(0)
    if have_pyscf and hasattr(mol, "to_pyscf"):
        pm = mol.to_pyscf()
        plib.num_threads(cores)
        dm = rng.random((pm.nao, pm.nao)); dm = dm @ dm.T
        t0 = time.perf_counter()
        pscf.hf.get_jk(pm, dm, hermi=1)
        dt = time.perf_counter() - t0
        return {"value": dt, "unit": "s per J/K build (pyscf.scf.hf.get_jk)", "cores": cores, "kind": "reference",
                "sample": "the whole workload, one call"}
    os.environ.setdefault("OMP_PROC_BIND", "spread")           # (read when libgomp starts its first team)
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle import jk as O
    dm = rng.random((layout.nao, layout.nao))
    dm = dm + dm.T
    # the sample is drawn BEFORE the clock starts; the timed region is C only (OpenMP, thread-local digestion: no atomics,
    # no shared Fock matrix -- oracle/jk_oracle.c:jqc_oracle_jk_bench)
    sample = sample_quartets(layout, per_class, 4000 * cores, rng)
    mix = {}
    for row in sample:
        key = "".join(str(int(layout.angs[s])) for s in row)
        mix[key] = mix.get(key, 0) + 1
    top = sorted(mix.items(), key=lambda kv: -kv[1])[:6]

    def rate(rows, nthreads, budget):
        O.jk_bench(layout.packed, dm, rows[: max(64, len(rows) // 10)], 1, nthreads=nthreads)      # warm-up / library load
        t0 = time.perf_counter()
        O.jk_bench(layout.packed, dm, rows, 1, nthreads=nthreads)
        one = time.perf_counter() - t0
        reps = max(1, int(budget / max(one, 1e-3)))
        t0 = time.perf_counter()
        O.jk_bench(layout.packed, dm, rows, reps, nthreads=nthreads)
        dt = time.perf_counter() - t0
        return reps * len(rows) / dt, reps, dt
    # one thread on a slice of the same sample (same class mix), then every physical core on all of it
    r1, reps1, dt1 = rate(sample[:: max(1, cores // 2)], 1, 0.25 * seconds)
    rn, reps, dt = rate(sample, cores, 0.75 * seconds)
    done = reps * len(sample)
    return {"value": rn, "unit": "quartets/s", "cores": cores, "kind": "port",
            "per_core": rn / cores, "one_thread": r1, "parallel_efficiency": rn / (r1 * cores),
            "class_mix_top": {k: round(v / len(sample), 4) for k, v in top},
            "sample": f"{len(sample)} canonical quartets drawn (before the clock starts) from this workload's dispatched "
                      f"(class, primitive pattern) histogram, {reps} passes = {done} quartets in {dt:.1f} s on {cores} OpenMP "
                      f"threads (one per physical core the container's CPU quota grants), and every {max(1, cores // 2)}th of them on ONE thread for {dt1:.1f} s "
                      f"({r1:.3e} quartets/s; parallel efficiency {rn / (r1 * cores):.2f}): ERI block + six contractions per "
                      f"quartet in oracle/jk_oracle.c (gcc -O3 -mavx2 -mfma), thread-local digestion sized by the shells' nf "
                      f"(no atomics, no Python in the timed region)"}


def internal_jk(mol, get_jk, dm):
    """One more (untimed) call that keeps J and K in the internal AO order (every rank: the call holds the path's collective)."""
    get_jk.keep_internal = True
    get_jk(mol, dm, hermi=1)
    get_jk.keep_internal = False
    return get_jk.stats.pop("vj_internal")[0], get_jk.stats.pop("vk_internal")[0]


def parity_blocks(mol, layout, internal, dm, nthreads):
    """max|dJ|, max|dK| of the timed workload's result against the CPU oracle on a fixed sample of shell blocks in the internal AO
    order (reference bar jqc/pyscf/tests/test_jk.py:83-84: 1e-7 absolute for FP64).  A block needs O(N^2) quartets
    (oracle/dense.py:sampled_blocks), so a handful finishes in seconds where a full oracle build would take hours.  Outside every
    timed region.  The sample: two J blocks of neighbouring shell pairs and two K blocks of random shell pairs (see below)."""
    from oracle import dense
    T = layout.transform_matrix()
    dm_int = T @ dm.cpu().numpy() @ T.T
    real = np.nonzero(~layout.pad_id)[0]
    by_l = {}
    for s_ in real:
        by_l.setdefault(int(layout.angs[s_]), []).append(int(s_))
    ls = sorted(by_l)
    rng = np.random.default_rng(6)
    pick = lambda l: int(rng.choice(by_l[l]))
    xyz = np.asarray(layout.packed)[:, :3]

    def near(i, l):                                            # the shell of angular momentum l closest to shell i (J_ij is of the size of
        c = np.array([s_ for s_ in by_l[l] if s_ != i] or by_l[l])     # the pair's overlap distribution: a far-apart pair would test nothing)
        return int(c[np.argmin(((xyz[c] - xyz[i]) ** 2).sum(1))])
    top, mid, low = ls[-1], ls[len(ls) // 2], ls[min(1, len(ls) - 1)]
    # two J and two K blocks (a block costs N^2/2 resp. N^2 oracle quartets, the f-containing ones ~10 us each on a core): a top-l shell
    # with its nearest neighbour one l below and a low-l neighbour pair; a top-l shell against a random s shell and a mid-l against a low-l one
    a = pick(top)
    b = near(a, ls[-2] if len(ls) > 1 else top)
    j_pairs = [(max(a, b), min(a, b))]
    a = pick(low)
    j_pairs.append(tuple(sorted((a, near(a, ls[0])), reverse=True)))
    k_pairs = [(pick(top), pick(ls[0])), (pick(mid), pick(low))]
    vj, vk = internal
    t0 = time.perf_counter()
    oj, ok = dense.sampled_blocks(layout, dm_int, j_pairs, k_pairs, nthreads=nthreads)
    dt = time.perf_counter() - t0
    loc = np.asarray(layout.ao_loc)
    blk = lambda m, i, j: m[int(loc[i]):int(loc[i + 1]), int(loc[j]):int(loc[j + 1])].cpu().numpy()
    dj = max(float(np.abs(blk(vj, i, j) - b).max()) for (i, j), b in oj.items())
    dk = max(float(np.abs(blk(vk, i, k) - b).max()) for (i, k), b in ok.items())
    sj = max(float(np.abs(b).max()) for b in oj.values())
    sk = max(float(np.abs(b).max()) for b in ok.values())
    nq = len(real) * (len(real) + 1) // 2 * len(j_pairs) + len(real) ** 2 * len(k_pairs)
    return {"max_abs_dJ": dj, "max_abs_dK": dk, "max_abs_J": sj, "max_abs_K": sk, "max_rel_dJ": dj / sj, "max_rel_dK": dk / sk,
            "blocks": len(oj) + len(ok),
            "block_classes": {"J": ["(%d%d|" % (layout.angs[i], layout.angs[j]) for i, j in j_pairs],
                              "K": ["(%d.|%d." % (layout.angs[i], layout.angs[k]) for i, k in k_pairs]},
            "oracle_quartets": int(nq), "oracle_seconds": round(dt, 2),
            "bar": "1e-7 absolute (the reference's FP64 bar, jqc/pyscf/tests/test_jk.py:83-84, on matrices of ITS size; north_star: Fock "
                   "elements within 1e-6); with this workload's D = R R^T of 2 588 functions the J elements reach 1e5, so the relative figures "
                   "are the ones to read: 1e-13 is FP64 round-off of sums of ~1e6 terms",
            "how": "shell blocks of J and K of the timed workload (internal AO order, after the epilogue) against oracle/dense.py:"
                   "sampled_blocks -- every quartet a block needs, same C oracle as the parity tests; outside the timed region"}


def committed_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the newest committed PMC summary that lists it."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            rec = json.load(open(f)).get(kernel)
        except (OSError, ValueError):
            continue
        if rec:
            return dict(rec, source=os.path.relpath(f, ROOT))
    return None


def grid_leg(mol, nsteps=5):
    """rho / vxc (GGA) throughput on a Becke grid; algorithmic flops = 256 (2 m^2 + 8 m) per block of m significant AOs."""
    import torch
    from joltqc_amd.pyscf import rks
    from joltqc_amd.pyscf.basis import BasisLayout
    from joltqc_amd.roofline import FP64_MFMA_PEAK_TFLOPS
    lay = BasisLayout.from_mol(mol, alignment=1)
    # Becke grid, 30 radial x (8 x 16) angular points per atom (joltqc_amd/gto/grids.py: grid generation is third party in the
    # reference), box-sorted and padded to 256 like rks.build_grids does
    from joltqc_amd.gto.grids import Grids
    gg = Grids(mol, 30, 8).build()
    order = rks.arg_group_grids(gg.coords)
    coords, weights = gg.coords[order], gg.weights[order]
    n = coords.shape[0] // 256 * 256
    per = coords.shape[0] // mol.natm

    class G:
        pass
    g = G(); g.coords = coords[:n]; g.weights = weights[:n]
    _, rho_k, vxc_k = rks.generate_rks_kernel(lay)
    np.random.seed(9)
    nocc = max(mol.nelectron // 2, 1)
    c = np.random.rand(mol.nao, nocc) - 0.5
    dm = torch.from_numpy(c @ c.T / nocc).cuda()
    wv = torch.rand((4, n), dtype=torch.float64, device="cuda")
    out = {"xc": "GGA", "ngrids": n, "nao": mol.nao, "grid": f"Becke, {per} points per atom (30 radial x 128 angular), box-sorted"}
    for fn, arg, label in ((rho_k, dm, "rho"), (vxc_k, wv, "vxc")):
        for _ in range(2):                                           # (the second call still pays one-off set-up: two warm-ups)
            fn(mol, g, "GGA", arg)
        torch.cuda.synchronize()
        each = []
        for _ in range(nsteps):
            t = time.perf_counter()
            fn(mol, g, "GGA", arg)
            torch.cuda.synchronize()
            each.append(time.perf_counter() - t)
        dt = float(np.median(each))
        m = rho_k.stats["nrow_h"].astype(float)                      # significant Cartesian AOs per 256-point block
        pairs = float((m * m).sum()) * 256                            # grid points x AO pairs actually contracted
        fl = 2.0 * pairs + 8.0 * 256 * float(m.sum())                # 256 (2 m^2 + 8 m) per block
        out[label] = {"ms": dt * 1e3, "points_x_ao_pairs_per_s": pairs / dt, "tflops": fl / dt / 1e12,
                      "frac_fp64_mfma_peak": fl / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS, "mean_ao_per_block": float(m.mean()),
                      "ms_each_step": [round(x * 1e3, 3) for x in each]}
    # meta-GGA (ndim 5: the tau GEMMs; BASELINE config 4's functional is one): flops = 256 (8 m^2 + 10 m) rho, 256 (8 m^2 + ...) vxc
    wv5 = torch.rand((5, n), dtype=torch.float64, device="cuda")
    mg = {}
    for fn, arg, label in ((rho_k, dm, "rho"), (vxc_k, wv5, "vxc")):
        fn(mol, g, "MGGA", arg); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(nsteps):
            fn(mol, g, "MGGA", arg)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t) / nsteps
        m = rho_k.stats["nrow_h"].astype(float)
        flm = 4.0 * 2.0 * float((m * m).sum()) * 256 + 10.0 * 256 * float(m.sum())       # four GEMMs (phi, d_x, d_y, d_z) + dots
        mg[label] = {"ms": dtm * 1e3, "tflops": flm / dtm / 1e12, "frac_fp64_mfma_peak": flm / dtm / 1e12 / FP64_MFMA_PEAK_TFLOPS}
    # (the model counts the four GEMMs of the reference's formulation; since round 4 the vxc kernel computes its three symmetric tau
    #  passes on and below the diagonal only, i.e. executes ~2.5 of them: the vxc fraction is algorithmic work / time, not MFMA busy)
    mg["note"] = "vxc: tau passes computed on and below the diagonal (about 2.5 of the 4 model GEMMs are executed)"
    out["meta_gga"] = mg
    # VV10 pair sums (reference dft/vv10.cu): 262 144 points of this grid against themselves, FP32 inner loop / FP64 accumulation,
    # 30 flop per pair against the FP32 vector peak
    try:
        from joltqc_amd.roofline import FP32_VALU_PEAK_TFLOPS
    except ImportError:
        FP32_VALU_PEAK_TFLOPS = 157.3
    nv = min(262144, n)
    xyz = torch.from_numpy(np.ascontiguousarray(coords[:nv].T)).cuda()
    rnd = torch.rand((3, nv), dtype=torch.float64, device="cuda")
    outer = torch.cat([xyz, rnd[0:1] + 0.5, rnd[1:2] + 0.5]).contiguous()
    inner = torch.cat([xyz, rnd[0:1] + 0.5, rnd[1:2] + 0.5, rnd[2:3] * 1e-3]).contiguous()
    rks.vv10_sums(outer, inner, True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        rks.vv10_sums(outer, inner, True)
    torch.cuda.synchronize()
    dtv = (time.perf_counter() - t) / 3
    out["vv10"] = {"points": nv, "ms": dtv * 1e3, "pairs_per_s": nv * nv / dtv, "tflops_30flop_per_pair": 30.0 * nv * nv / dtv / 1e12,
                   "frac_fp32_valu_peak": 30.0 * nv * nv / dtv / 1e12 / FP32_VALU_PEAK_TFLOPS}
    # the same two calls with the default DFT cutoffs of apply() (cutoff_fp64 = 1e-6: weak AO pairs through the FP32 MFMA)
    _, rho_m, vxc_m = rks.generate_rks_kernel(lay, cutoff_fp64=1e-6, cutoff_fp32=1e-13)
    for fn, arg, label in ((rho_m, dm, "rho"), (vxc_m, wv, "vxc")):
        fn(mol, g, "GGA", arg); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(nsteps):
            fn(mol, g, "GGA", arg)
        torch.cuda.synchronize()
        out[label]["ms_mixed_fp32_window"] = (time.perf_counter() - t) / nsteps * 1e3
    return out


def spawn_ranks(args):
    """Start ``--gpus`` ranks as child processes (this process has not touched a GPU) and relay rank 0's line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grid", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    small = args.workload.startswith("benzene")
    steps = args.steps if args.steps is not None else (20 if small else 5)
    warmup = args.warmup if args.warmup is not None else (3 if small else 1)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)

    import torch
    import torch.distributed as dist
    from joltqc_amd.constants import tile_width
    from joltqc_amd.pyscf import jk as jkmod
    from joltqc_amd.pyscf.basis import BasisLayout
    from joltqc_amd.roofline import (FP64_VALU_PEAK_MEASURED_BY_WAVES, FP64_VALU_PEAK_MEASURED_TFLOPS, FP64_VALU_PEAK_TFLOPS,
                                     quartet_bytes, quartet_flops)

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

    for _ in range(max(warmup, 1)):
        vj, vk = get_jk(mol, dm, hermi=1)
    torch.cuda.synchronize()
    n64, n32, per = get_jk.quartet_counts()
    flops_by_ang, bytes_by_ang, count_by_ang = {}, {}, {}
    for (ang, npr), (a, b) in per.items():
        flops_by_ang[ang] = flops_by_ang.get(ang, 0) + (a + b) * quartet_flops(ang, npr or (1, 1, 1, 1))
        bytes_by_ang[ang] = bytes_by_ang.get(ang, 0) + (a + b) * quartet_bytes(ang)
        count_by_ang[ang] = count_by_ang.get(ang, 0) + a + b
    total_flops = sum(flops_by_ang.values())

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        vj, vk = get_jk(mol, dm, hermi=1)
    barrier()
    dt = time.perf_counter() - t0
    dt_rank = dt
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    nq = torch.tensor([float(n64 + n32), float(total_flops)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nq)
    dt = float(tmax.item())
    quartets, flops_all = float(nq[0].item()), float(nq[1].item())

    # roofline leg: EVERY class kernel bracketed by HIP events on the stream it is launched on, class kernels launched
    # back to back on ONE stream (in the timed region above they overlap on several streams, which makes a single
    # kernel's duration ill-defined); same process, same inputs, right after the timed loop.  The kernel reported is
    # the one that takes the most time on this rank.
    get_jk.set_streams(1)
    get_jk.set_probe("all")
    nprobe = 3 if small else 2
    for _ in range(nprobe):
        get_jk(mol, dm, hermi=1)
    torch.cuda.synchronize()
    tm = {}
    for ang, (e0, e1) in zip(get_jk.stats.get("probe_classes", []), get_jk.stats.get("probe_events", [])):
        tm.setdefault(ang, []).append(e0.elapsed_time(e1))
    tm = {a: float(np.mean(v)) for a, v in tm.items()}
    serial_ms = sum(tm.values())
    # what a rank does NOT shard: everything of a call but the class kernels -- D into the internal order (two GEMMs), shell-block
    # maxima + their logarithm, the plan lookup and the launch loop, zeroing the Fock buffers, the epilogue (transposes, two GEMMs per
    # matrix back to the molecule's order).  Measured by a call whose class filter rejects every class; at N ranks this time stays
    # while the kernel time divides, so it bounds the scaling curve: T(N) ~ kernel_sum / N x imbalance + host_serial + all-reduce.
    get_jk.set_probe(None)
    get_jk(mol, dm, hermi=1, _classes=lambda a: False)
    torch.cuda.synchronize()
    th = time.perf_counter()
    for _ in range(3):
        get_jk(mol, dm, hermi=1, _classes=lambda a: False)
    torch.cuda.synchronize()
    host_serial_ms = (time.perf_counter() - th) / 3 * 1e3
    get_jk.set_streams(None)
    internal = None if args.no_parity else internal_jk(mol, get_jk, dm)
    per_rank = None
    if world > 1:
        # diagnosability of the scaling curve: what every rank did -- its kernels' serial time, its quartets, the one Fock
        # all-reduce (timed alone on a buffer of the call's size), its wall time for the timed steps
        buf = torch.zeros((2, layout.nao, layout.nao), dtype=torch.float64, device="cuda")
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        ta = time.perf_counter()
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - ta) / 3 * 1e3
        mine = torch.tensor([serial_ms, float(n64 + n32), ar_ms, dt_rank / steps * 1e3], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [[float(x) for x in t.tolist()] for t in allr]
        ks = [r[0] for r in rows]
        per_rank = {"kernel_ms_sum": [round(r[0], 3) for r in rows], "quartets": [r[1] for r in rows],
                    "allreduce_ms": [round(r[2], 3) for r in rows], "step_ms": [round(r[3], 3) for r in rows],
                    "kernel_imbalance_max_over_mean": max(ks) / (sum(ks) / world) if sum(ks) > 0 else None,
                    "allreduce_bytes": int(buf.numel() * 8)}
    if rank == 0:
        from joltqc_amd.backend import jk as router
        ms = dt / steps * 1e3
        dom = max(tm, key=tm.get)
        kern_ms = tm[dom]
        achieved = flops_by_ang.get(dom, 0) / (kern_ms * 1e-3) / 1e12
        sel = router.select_algo(dom, small=False)
        mode = sel & 0xf
        kname = (("jk_quad_" if sel & (1 << 24) else "jk_tile1q_") if mode == 2 else "jk_tile512_" if mode == 3 else "jk_tile_") + "%d%d%d%d" % dom
        best = max(tm, key=lambda a: flops_by_ang.get(a, 0) / tm[a])
        out = {
            "metric": "ERI quartets/s (J/K Fock build, def2-TZVPP)", "value": quartets * steps / dt,
            "unit": "quartets/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms, "jk_wall_s": ms * 1e-3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wname, "natm": mol.natm, "nao": mol.nao, "split_shells": int((~layout.pad_id).sum()),
                       "quartets_per_step": quartets, "model_gflop_per_step": flops_all / 1e9,
                       "density": "rand(nao,nao) R R^T seed 9 (dense: no density screening)", "cutoff": 1e-13,
                       "parallelism": f"quartet work sharded over {world} rank(s) (cost-aware class/strip split) + 1 Fock all-reduce"},
            "roofline": {"bound": "valu_fp64", "kernel": kname, "selected_by": "largest measured launch time (HIP events, serial streams)",
                         "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VALU_PEAK_TFLOPS,
                         "peak_measured": FP64_VALU_PEAK_MEASURED_TFLOPS, "frac_of_measured_peak": achieved / FP64_VALU_PEAK_MEASURED_TFLOPS,
                         "peak_measured_by_waves_per_simd": FP64_VALU_PEAK_MEASURED_BY_WAVES,
                         "peak_note": "peak = datasheet FP64 vector rate; peak_measured = v_fma_f64 micro-benchmark on this pool's MI355X "
                                      "(tools/micro/fp64_fma_peak.hip, profiles/r05_fp64_fma_peak_microbench.json)",
                         "kernel_ms": kern_ms, "kernel_gflop": flops_by_ang.get(dom, 0) / 1e9,
                         "kernel_quartets": count_by_ang.get(dom, 0),
                         "algorithmic_bytes": bytes_by_ang.get(dom, 0), "traffic": None,
                         "whole_path": {"achieved": flops_all * steps / dt / 1e12,
                                        "frac": flops_all * steps / dt / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                        "frac_of_measured_peak": flops_all * steps / dt / 1e12 / FP64_VALU_PEAK_MEASURED_TFLOPS,
                                        "serial_kernel_sum_ms": serial_ms, "classes": len(tm)},
                         "best_kernel": {"kernel": "%d%d%d%d" % best,
                                         "achieved": flops_by_ang.get(best, 0) / tm[best] / 1e9,
                                         "frac": flops_by_ang.get(best, 0) / tm[best] / 1e9 / FP64_VALU_PEAK_TFLOPS}},
        }
        ar_bytes = 2 * layout.nao * layout.nao * 8
        out["host_serial_ms"] = host_serial_ms          # (N > 1: includes this path's all-reduce; the model below is an N = 1 prediction)
        if world == 1:
            out["scaling_model"] = {
                "host_serial_ms": host_serial_ms, "kernel_sum_ms": serial_ms, "allreduce_bytes": ar_bytes,
                "form": "T(N) = kernel_sum_ms / N x imbalance + host_serial_ms + allreduce_ms(N)",
                # ring all-reduce over xGMI: 2 (N - 1) / N x bytes per rank over one ~153 GB/s link direction at 70 % efficiency
                "predicted_ms": {str(n): round(serial_ms / n * 1.1 + host_serial_ms + (0.0 if n == 1 else 2.0 * (n - 1) / n * ar_bytes / (0.7 * 153e9) * 1e3), 1)
                                 for n in (1, 2, 4, 8)},
                "assumptions": "imbalance 1.10 (LPT split, tests/test_sharding.py: <= 10 %), ring all-reduce at 70 % of one 153 GB/s xGMI link "
                               "direction (MI355X_MICROARCH.md); a prediction until an N > 1 node runs the bench"}
        if per_rank is not None:
            out["per_rank"] = per_rank
        tr = committed_traffic(kname) if args.workload == DEFAULT_WORKLOAD else None
        if tr:
            out["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_detail"] = tr
            nq1 = float(tr.get("quartets") or 0)
            if world > 1 and nq1 > 0:
                # the committed counters are of the one-rank launch of this kernel; a rank of an N-rank run launches it over its share
                # of the class's task rows: HBM bytes scaled by the quartets this rank's launch processed (counter runs need one rank)
                share = count_by_ang.get(dom, 0) / nq1
                out["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch") * share
                out["roofline"]["traffic_detail"] = dict(tr, scaled_by_quartet_share_of_this_rank=share)
        if internal is not None:
            try:
                out["parity"] = parity_blocks(mol, layout, internal, dm, physical_cores())
            except Exception as e:  # noqa: BLE001  (the headline must survive a failure of the extra leg)
                out["parity"] = {"error": repr(e)[:300]}
        if world == 1:
            # the dense density above disables density screening (as the reference's ones-D benchmark does); the same
            # build with a density of SCF-like decay, D = C C^T / n_occ, for orientation (not `value`)
            get_jk.set_streams(None)
            get_jk.set_probe(None)
            nocc = max(mol.nelectron // 2, 1)
            np.random.seed(9)
            c = np.random.rand(mol.nao, nocc) - 0.5
            dm2 = torch.from_numpy(c @ c.T / nocc).cuda()
            get_jk(mol, dm2, hermi=1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                get_jk(mol, dm2, hermi=1)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t1) / 2
            q2 = float(sum(get_jk.quartet_counts()[:2]))
            out["realistic_density"] = {"density": "C C^T / n_occ, C = rand(nao, n_occ) - 1/2, seed 9", "ms_per_step": dt2 * 1e3,
                                        "quartets_per_step": q2, "quartets_per_s": q2 / dt2}
            # the same build with the mixed FP32 / FP64 windows of BASELINE config 4 (quartets whose bound lies between 1e-13 and
            # 1e-7 in FP32 kernels, the rest in FP64; reference jk.py:236-248): time, share of the FP32 queue, deviation from FP64
            try:
                ref_j, ref_k = (x.clone() for x in get_jk(mol, dm2, hermi=1))
                get_mixed = jkmod.generate_jk_kernel(layout, cutoff_fp64=1e-7, cutoff_fp32=1e-13)
                vj, vk = get_mixed(mol, dm2, hermi=1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    vj, vk = get_mixed(mol, dm2, hermi=1)
                torch.cuda.synchronize()
                dt3 = (time.perf_counter() - t1) / 2
                n64, n32 = get_mixed.quartet_counts()[:2]
                out["realistic_density"]["mixed_precision"] = {
                    "cutoff_fp64": 1e-7, "cutoff_fp32": 1e-13, "ms_per_step": dt3 * 1e3, "fp32_quartet_share": n32 / max(n64 + n32, 1),
                    "max_rel_dev_j": float((vj - ref_j).abs().max() / ref_j.abs().max()),
                    "max_rel_dev_k": float((vk - ref_k).abs().max() / ref_k.abs().max())}
            except Exception as e:  # noqa: BLE001  (the headline must survive a failure of the extra leg)
                out["realistic_density"]["mixed_precision"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_grid and "realistic_density" in out:
            try:
                from joltqc_amd.pyscf import grad as gradmod
                fn = gradmod.generate_jk_energy_per_atom(layout, cutoff=1e-13)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                g2 = fn(mol, dm2)
                torch.cuda.synchronize()
                dtg = time.perf_counter() - t1
                out["forces"] = {"two_electron_gradient_ms": dtg * 1e3, "quartets": float(fn.quartet_count()),
                                 "jk_builds_of_the_same_density": dtg / out["realistic_density"]["ms_per_step"] * 1e3,
                                 "finite": bool(torch.isfinite(g2).all()), "launches": int(fn.stats["launches"])}
            except Exception as e:  # noqa: BLE001  (the headline must survive a failure of the extra leg)
                out["forces"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_grid:
            try:
                out["grid_path"] = grid_leg(mol)
            except Exception as e:  # noqa: BLE001  (the headline must survive a failure of the extra leg)
                out["grid_path"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline:          # (N > 1: on rank 0 only, after the collectives of the timed region are over)
            out["cpu_baseline"] = cpu_baseline(mol, layout, per)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
