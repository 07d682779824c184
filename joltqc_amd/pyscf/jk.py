"""J/K Fock build on MI355X behind the reference's ``get_jk`` interface.

Mirrors ``/root/reference/jqc/pyscf/jk.py``: ``generate_jk_kernel`` (:93) returns the ``get_jk``
closure (:109-380); ``generate_get_j/k/jk`` (:51-75); RHF ``generate_get_veff`` (:78-90).
Same signature, same return shapes, same screening predicate and the same epilogue (:350-370).

What is organised differently for MI355X (HIP streams instead of one blocking D2H per chunk):
  * Screening works on per-group-pair lists of shell pairs sorted by Schwarz bound; ONE screening
    launch per chunk appends the surviving quartets of EVERY class to per-class queue regions
    (device-side counters).  J/K kernels read their task count from device memory, so the host
    never waits between queue generation and the launches (the reference blocks on ``info.get()``
    per class and chunk, jk.py:280).
  * One kernel per ANGULAR class (primitive counts are run-time loop bounds), i.e. at most 140
    code objects instead of one per (l, nprim) pattern.
  * The single host read per call is ``log_max_dm`` (the reference does the same, jk.py:183).
"""
import math
import time
from typing import Dict, List, Tuple

import numpy as np

from ..backend import lib as _lib
from ..backend import jk as _router

__all__ = ["generate_jk_kernel", "generate_get_j", "generate_get_k", "generate_get_jk", "generate_get_veff",
           "make_pair_lists"]

PAIR_CUTOFF = 1e-13          # reference jk.py:48
QUEUE_DEPTH = 1 << 26        # quartets per chunk (8 B each -> 512 MiB); reference uses 2^28 (jk_tasks.py:30)
STRIPS_PER_LIST = 64         # host-side trimming granularity of the sorted pair lists


def generate_get_j(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13):
    kern = generate_jk_kernel(basis_layout, cutoff_fp64=cutoff_fp64, cutoff_fp32=cutoff_fp32)

    def get_j(*args, **kwargs):
        return kern(*args, with_j=True, with_k=False, **kwargs)[0]
    return get_j


def generate_get_k(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13):
    kern = generate_jk_kernel(basis_layout, cutoff_fp64=cutoff_fp64, cutoff_fp32=cutoff_fp32)

    def get_k(*args, **kwargs):
        return kern(*args, with_j=False, with_k=True, **kwargs)[1]
    return get_k


def generate_get_jk(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13):
    kern = generate_jk_kernel(basis_layout, cutoff_fp64=cutoff_fp64, cutoff_fp32=cutoff_fp32)

    def get_jk(*args, **kwargs):
        return kern(*args, **kwargs)
    return get_jk


def generate_get_veff():
    """RHF get_veff with incremental Fock build (reference jk.py:78-90)."""
    import torch

    def get_veff(mf, mol=None, dm=None, dm_last=None, vhf_last=None, hermi=1):
        if dm is None:
            dm = mf.make_rdm1()
        dev = _lib.require_gpu()
        as_t = lambda x: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float64, device=dev)
        incremental = dm_last is not None and not (np.isscalar(dm_last) and dm_last == 0) and getattr(mf, "direct_scf", True)
        d = as_t(dm) - as_t(dm_last) if incremental else as_t(dm)
        vj, vk = mf.get_jk(mol, d, hermi)
        vhf = vj - 0.5 * vk
        if vhf_last is not None and not (np.isscalar(vhf_last) and vhf_last == 0):
            vhf = vhf + as_t(vhf_last)
        return vhf
    return get_veff


def make_pair_lists(group_offset, q_host, pad_id):
    """Per group pair (gi >= gj): shell pairs ish >= jsh sorted by Schwarz bound, largest first.
    Plays the role of ``make_tile_pairs`` (reference jk.py:385-431) at pair granularity.
    Returns {(gi, gj): (pair_sh uint32[n] = ish<<16|jsh, pair_q float32[n])}."""
    out = {}
    ng = len(group_offset) - 1
    for gi in range(ng):
        i0, i1 = int(group_offset[gi]), int(group_offset[gi + 1])
        for gj in range(gi + 1):
            j0, j1 = int(group_offset[gj]), int(group_offset[gj + 1])
            ii, jj = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), indexing="ij")
            m = (ii >= jj) & ~pad_id[ii] & ~pad_id[jj]
            ii, jj = ii[m], jj[m]
            q = q_host[ii, jj]
            keep = q > -80.0
            ii, jj, q = ii[keep], jj[keep], q[keep]
            if ii.size == 0:
                continue
            order = np.argsort(-q, kind="stable")
            sh = ((ii[order].astype(np.uint32) << np.uint32(16)) | jj[order].astype(np.uint32)).astype(np.uint32)
            out[gi, gj] = (sh, q[order].astype(np.float32))
    return out


class _PairTables:
    """Device copies of the concatenated pair lists of one (layout, omega)."""

    def __init__(self, layout, omega):
        import torch
        dev = _lib.require_gpu()
        q_dev = layout.q_matrix(omega)
        q_host = q_dev.cpu().numpy()
        lists = make_pair_lists(layout.group_offset, q_host, layout.pad_id)
        self.offset: Dict[Tuple[int, int], int] = {}
        self.q_host: Dict[Tuple[int, int], np.ndarray] = {}
        sh_all, q_all, off = [], [], 0
        for key, (sh, q) in lists.items():
            self.offset[key] = off
            self.q_host[key] = q
            sh_all.append(sh)
            q_all.append(q)
            off += sh.size
        self.npairs = off
        if off:
            self.sh = torch.from_numpy(np.concatenate(sh_all).view(np.int32)).to(dev)
            self.q = torch.from_numpy(np.concatenate(q_all)).to(dev)
        else:
            self.sh = torch.zeros(1, dtype=torch.int32, device=dev)
            self.q = torch.zeros(1, dtype=torch.float32, device=dev)
        self.qmax = max((float(q[0]) for q in self.q_host.values()), default=-100.0)


def _class_id(ang):
    return ((ang[0] * 5 + ang[1]) * 5 + ang[2]) * 5 + ang[3]


def build_screen_plan(layout, pt: "_PairTables", log_cut: float, log_max_dm: float, queue_depth: int):
    """Host-side plan of one get_jk call: list of chunks, each with its screen tasks, the per-class
    queue regions and upper bounds.  Everything is derived from the sorted Schwarz lists; no
    device work, no synchronisation."""
    gkey = layout.group_key
    ng = layout.ngroups
    pair_cut = math.log(PAIR_CUTOFF) - log_max_dm          # reference jk.py:184-186
    nkeep = {k: int(np.searchsorted(-q, -pair_cut, side="left")) for k, q in pt.q_host.items()}
    raw = []   # (cls, ang, ij0, nij, kl0, nkl)
    for gi in range(ng):
        for gj in range(gi + 1):
            nij_all = nkeep.get((gi, gj), 0)
            if nij_all == 0:
                continue
            qij = pt.q_host[gi, gj]
            for gk in range(gi + 1):
                for gl in range(gk + 1):
                    nkl_all = nkeep.get((gk, gl), 0)
                    if nkl_all == 0:
                        continue
                    qkl = pt.q_host[gk, gl]
                    ang = (int(gkey[gi, 0]), int(gkey[gj, 0]), int(gkey[gk, 0]), int(gkey[gl, 0]))
                    strip = max(16, -(-nij_all // STRIPS_PER_LIST))
                    strip = (strip + 15) // 16 * 16
                    for s0 in range(0, nij_all, strip):
                        n_ij = min(strip, nij_all - s0)
                        # kl pairs that can still pass together with the best ij of this strip
                        thr = log_cut - log_max_dm - float(qij[s0])
                        n_kl = min(nkl_all, int(np.searchsorted(-qkl, -thr, side="left")))
                        if n_kl <= 0:
                            break
                        raw.append((_class_id(ang), ang, pt.offset[gi, gj] + s0, n_ij, pt.offset[gk, gl], n_kl))
    # split oversize tasks, then pack chunks
    tasks = []
    for cls, ang, ij0, nij, kl0, nkl in raw:
        if nij * nkl <= queue_depth:
            tasks.append((cls, ang, ij0, nij, kl0, nkl))
            continue
        rows = max(16, (queue_depth // max(nkl, 1)) // 16 * 16)
        if rows * nkl > queue_depth:
            # a single 16-row strip does not fit: split the kl range as well
            cols = max(16, (queue_depth // 16) // 16 * 16)
            for a in range(0, nij, 16):
                for b in range(0, nkl, cols):
                    tasks.append((cls, ang, ij0 + a, min(16, nij - a), kl0 + b, min(cols, nkl - b)))
        else:
            for a in range(0, nij, rows):
                tasks.append((cls, ang, ij0 + a, min(rows, nij - a), kl0, nkl))
    chunks, cur, cur_n = [], [], 0
    for t in tasks:
        ub = t[3] * t[5]
        if cur and cur_n + ub > queue_depth:
            chunks.append(cur)
            cur, cur_n = [], 0
        cur.append(t)
        cur_n += ub
    if cur:
        chunks.append(cur)
    plans = []
    for ch in chunks:
        classes: Dict[int, Tuple] = {}
        ub: Dict[int, int] = {}
        for cls, ang, ij0, nij, kl0, nkl in ch:
            classes[cls] = ang
            ub[cls] = ub.get(cls, 0) + nij * nkl
        cls_list = sorted(classes, reverse=True)               # high angular momentum first (jk.py:209)
        slot = {c: n for n, c in enumerate(cls_list)}
        region = np.zeros((len(cls_list), 2), dtype=np.int64)
        pos = 0
        for c in cls_list:
            region[slot[c]] = (pos, pos + ub[c])
            pos += ub[c]
        tab = np.zeros((len(ch), 8), dtype=np.int32)
        blk = 0
        for n, (cls, ang, ij0, nij, kl0, nkl) in enumerate(ch):
            tab[n, :6] = (ij0, nij, kl0, nkl, slot[cls], blk)
            blk += ((nij + 15) // 16) * ((nkl + 15) // 16)
        plans.append({"tasks": tab, "nblocks": blk, "region": region, "classes": [classes[c] for c in cls_list],
                      "ub": [ub[c] for c in cls_list], "total": pos})
    return plans


def generate_jk_kernel(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13):
    import torch
    log_cutoff_fp64 = float(np.float32(math.log(cutoff_fp64)))
    log_cutoff_fp32 = float(np.float32(math.log(cutoff_fp32)))
    mixed = cutoff_fp32 < cutoff_fp64
    layout = basis_layout
    nbas = layout.nbasis
    nao = layout.nao
    state = {"pairs": {}, "queue": None, "stats": {}}

    def get_jk(mol_ref=None, dm=None, hermi=0, vhfopt=None, with_j=True, with_k=True, omega=None, verbose=None):
        """Compute J, K; compatible with ``pyscf.scf.hf.get_jk`` / the reference closure (jk.py:109-118).
        ``mol_ref`` is ignored in favour of the layout captured at ``apply`` time (jk.py:123)."""
        assert with_j or with_k
        if omega is not None:
            assert omega >= 0.0, "short ranged J/K not supported"
        t_start = time.perf_counter()
        dev = _lib.require_gpu()
        _lib.ensure_rys()
        L = _lib.lib()
        stream = _lib.stream_ptr()
        om = float(omega) if omega else 0.0
        lr = om > 0.0

        dm_in = dm
        dm_t = torch.as_tensor(np.asarray(dm) if not torch.is_tensor(dm) else dm, dtype=torch.float64, device=dev)
        out_shape = tuple(dm_t.shape)
        dms = layout.dm_from_mol(dm_t.reshape(-1, layout.nao_mol, layout.nao_mol)).contiguous()

        # shell-block density bounds (max_block_pooling, linalg_helper.py:125)
        dm_cond = torch.empty((nbas, nbas), dtype=torch.float32, device=dev)
        _lib.check(L.jqc_shell_block_max(dms.data_ptr(), dms.shape[0], nao, layout.device_ao_loc().data_ptr(), nbas,
                                         dm_cond.data_ptr(), stream))
        if hermi == 0:
            dm_cond = dm_cond + dm_cond.T
        log_dm_cond = torch.log(dm_cond.double() + 1e-300).float().contiguous()
        # d_large is floored at -36.8 inside the predicate (screen_jk_tasks.cu:241), so is every bound here
        log_max_dm = max(float(log_dm_cond.max().item()), -36.8)

        if om not in state["pairs"]:
            state["pairs"][om] = _PairTables(layout, om)
        pt = state["pairs"][om]

        if hermi == 0:
            dms = torch.cat([dms, dms.transpose(1, 2)], dim=0).contiguous()   # jk.py:189-191
        n_dm = dms.shape[0]
        dms_fp32 = dms.float().contiguous() if mixed else None
        vj = torch.zeros_like(dms) if with_j else None
        vk = torch.zeros_like(dms) if with_k else None

        plans = build_screen_plan(layout, pt, log_cutoff_fp32, log_max_dm, QUEUE_DEPTH)
        qsize = max((p["total"] for p in plans), default=0)
        if state["queue"] is None or state["queue"].numel() < qsize * 4:
            state["queue"] = torch.empty(max(qsize, 1) * 4, dtype=torch.int16, device=dev)
        queue = state["queue"]
        b64 = layout.basis_data_fp64["packed"]
        b32 = layout.basis_data_fp32["packed"] if mixed else None
        n_launch = 0
        counter_bufs = []
        for p in plans:
            ncls = len(p["classes"])
            tasks_d = torch.from_numpy(p["tasks"]).to(dev)
            region_d = torch.from_numpy(p["region"]).to(dev)
            counters = torch.zeros((ncls, 2), dtype=torch.int32, device=dev)
            counter_bufs.append((counters, p))
            _lib.check(L.jqc_screen_jk_tasks(tasks_d.data_ptr(), p["tasks"].shape[0], p["nblocks"], pt.sh.data_ptr(),
                                             pt.q.data_ptr(), log_dm_cond.data_ptr(), nbas, int(with_j), int(with_k),
                                             log_cutoff_fp32, log_cutoff_fp64, log_max_dm, queue.data_ptr(),
                                             region_d.data_ptr(), counters.data_ptr(), stream))
            n_launch += 1
            for n, ang in enumerate(p["classes"]):
                beg, end = int(p["region"][n, 0]), int(p["region"][n, 1])
                h64 = _router.gen_jk_kernel(ang, do_j=with_j, do_k=with_k, rys_lr=lr, fp32=False)
                _lib.check(L.jqc_jk_launch(h64, nao, b64.data_ptr(), dms.data_ptr(),
                                           vj.data_ptr() if with_j else None, vk.data_ptr() if with_k else None, om,
                                           queue.data_ptr() + beg * 8, counters.data_ptr() + (2 * n) * 4,
                                           end - beg, 1, n_dm, stream))
                n_launch += 1
                if mixed:
                    h32 = _router.gen_jk_kernel(ang, do_j=with_j, do_k=with_k, rys_lr=lr, fp32=True)
                    _lib.check(L.jqc_jk_launch(h32, nao, b32.data_ptr(), dms_fp32.data_ptr(),
                                               vj.data_ptr() if with_j else None, vk.data_ptr() if with_k else None,
                                               om, queue.data_ptr() + (end - 1) * 8,
                                               counters.data_ptr() + (2 * n + 1) * 4, end - beg, -1, n_dm, stream))
                    n_launch += 1

        # epilogue (reference jk.py:350-370)
        if with_j:
            if hermi == 1:
                vj = vj * 2.0
            else:
                h = n_dm // 2
                vj = vj[:h] + vj[h:].transpose(1, 2)
            vj = vj + vj.transpose(1, 2)
            vj = layout.dm_to_mol(vj).reshape(out_shape)
        else:
            vj = 0
        if with_k:
            if hermi == 1:
                vk = vk + vk.transpose(1, 2)
            else:
                h = n_dm // 2
                vk = vk[:h] + vk[h:].transpose(1, 2)
            vk = layout.dm_to_mol(vk).reshape(out_shape)
        else:
            vk = 0

        st = state["stats"]
        st["launches"] = n_launch
        st["chunks"] = len(plans)
        st["counter_bufs"] = counter_bufs          # read lazily by quartet_counts()
        st["host_seconds"] = time.perf_counter() - t_start
        if isinstance(dm_in, np.ndarray) and getattr(get_jk, "return_numpy", False):
            vj = vj.cpu().numpy() if with_j else 0
            vk = vk.cpu().numpy() if with_k else 0
        return vj, vk

    def quartet_counts():
        """(n_fp64, n_fp32) quartets dispatched by the last call (synchronises)."""
        n64 = n32 = 0
        per_class = {}
        for counters, p in state["stats"].get("counter_bufs", []):
            c = counters.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            n64 += int(c[:, 0].sum())
            n32 += int(c[:, 1].sum())
            for n, ang in enumerate(p["classes"]):
                a = per_class.setdefault(tuple(ang), [0, 0])
                a[0] += int(c[n, 0])
                a[1] += int(c[n, 1])
        return n64, n32, per_class

    get_jk.quartet_counts = quartet_counts
    get_jk.stats = state["stats"]
    get_jk.layout = layout
    get_jk.return_numpy = False
    return get_jk
