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
import os
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
N_STREAMS = int(__import__('os').environ.get('JQC_STREAMS', '8'))                # class kernels are independent (atomic accumulation): spread them over HIP streams
N_STREAMS_BIG = 4            # calls whose class launches each fill the chip many times over: fewer kernels in flight (4 790 ms
                             # against 4 900 ms on the 112-atom step; benzene-size calls are faster on 8-16: 9.9-10.0 against 10.2 ms)
BIG_CALL_WGS = 1_000_000     # summed workgroups of a call from which N_STREAMS_BIG applies (unless JQC_STREAMS / set_streams say otherwise)


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
    """RHF get_veff with incremental Fock build (reference jk.py:78-90).  On a plain (CPU) PySCF object -- ``apply`` marks it
    with ``_jqc_numpy_boundary`` -- the potential goes back as a NumPy array: PySCF adds it to ``h1e`` and contracts it with
    the density in NumPy."""
    import torch
    from .rks import IncrementPolicy
    policy = IncrementPolicy()

    def get_veff(mf, mol=None, dm=None, dm_last=None, vhf_last=None, hermi=1):
        if dm is None:
            dm = mf.make_rdm1()
        dev = _lib.require_gpu()
        as_t = lambda x: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float64, device=dev)
        none = lambda x: x is None or (np.isscalar(x) and x == 0)
        incremental = not none(dm_last) and not none(vhf_last) and getattr(mf, "direct_scf", True)
        d = as_t(dm)
        if incremental:
            # increments as in the reference, starting over from the full density whenever the increment has shrunk by 1e3 since
            # the last full build, and every 12 calls (rks.IncrementPolicy: the sub-cutoff terms every increment drops pile up)
            dd = d - as_t(dm_last)
            incremental = not policy.full_build(float(dd.abs().max()))
        else:
            policy.reset()
            policy.full_build(float(d.abs().max()))
        announce = getattr(mf.get_jk, "set_increment_of", None)
        if announce and incremental:
            announce(float(d.abs().max()))
        try:
            vj, vk = mf.get_jk(mol, dd if incremental else d, hermi)
        finally:
            if announce and incremental:
                announce(None)
        vhf = as_t(vj) - 0.5 * as_t(vk)
        if incremental:
            vhf = vhf + as_t(vhf_last)
        if getattr(mf, "_jqc_numpy_boundary", False):
            return vhf.cpu().numpy()
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


def build_screen_plan(layout, pt: "_PairTables", log_cut: float, log_max_dm: float, queue_depth: int, want=None,
                      shard=None):
    """Host-side plan of one get_jk call: list of chunks, each with its screen tasks, the per-class
    queue regions and upper bounds.  Everything is derived from the sorted Schwarz lists; no
    device work, no synchronisation."""
    gkey = layout.group_key
    ng = layout.ngroups
    pair_cut = math.log(PAIR_CUTOFF) - log_max_dm          # reference jk.py:184-186
    nkeep = {k: int(np.searchsorted(-q, -pair_cut, side="left")) for k, q in pt.q_host.items()}
    raw = []   # (cls, ang, ij0, nij, kl0, nkl)
    strip_no = 0
    rank, world = shard if shard is not None else (0, 1)
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
                    if want is not None and not want(ang):
                        continue
                    strip = max(16, -(-nij_all // STRIPS_PER_LIST))
                    strip = (strip + 15) // 16 * 16
                    for s0 in range(0, nij_all, strip):
                        n_ij = min(strip, nij_all - s0)
                        # kl pairs that can still pass together with the best ij of this strip
                        thr = log_cut - log_max_dm - float(qij[s0])
                        n_kl = min(nkl_all, int(np.searchsorted(-qkl, -thr, side="left")))
                        if n_kl <= 0:
                            break
                        strip_no += 1
                        if strip_no % world != rank:
                            continue
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


class _TileTables:
    """Per group pair (gi >= gj): list of shell-tile pairs sorted by their largest Schwarz bound.
    Tile = ``tile_width(l)`` consecutive shells of one (l, nprim) group (groups are padded to that
    multiple); plays the role of ``make_tile_pairs`` (reference jk.py:385-431)."""

    def __init__(self, layout, omega, q_host=None):
        """``q_host`` (float32 [nbas, nbas] log-Schwarz matrix) makes the tables host-only: used by the CPU tests
        of the sharding logic; the product path always takes the device matrix."""
        from ..constants import tile_width
        host_only = q_host is not None
        if not host_only:
            import torch
            dev = _lib.require_gpu()
            self.q_dev = layout.q_matrix(omega)
            q_host = self.q_dev.cpu().numpy()
        goff, gkey = layout.group_offset, layout.group_key
        self.offset: Dict[Tuple[int, int], int] = {}
        self.q_host: Dict[Tuple[int, int], np.ndarray] = {}
        sh_all, q_all, w_all, off = [], [], [], 0
        for gi in range(layout.ngroups):
            wi = tile_width(int(gkey[gi, 0]))
            assert (goff[gi + 1] - goff[gi]) % wi == 0, "layout groups must be padded to the tile width"
            ti = np.arange(goff[gi], goff[gi + 1], wi)
            for gj in range(gi + 1):
                wj = tile_width(int(gkey[gj, 0]))
                tj = np.arange(goff[gj], goff[gj + 1], wj)
                blk = q_host[goff[gi]:goff[gi + 1], goff[gj]:goff[gj + 1]]
                qt = blk.reshape(len(ti), wi, len(tj), wj).max(axis=(1, 3))
                ii, jj = np.meshgrid(ti, tj, indexing="ij")
                m = qt > -80.0
                if gi == gj:
                    m &= ii >= jj
                if not m.any():
                    continue
                ii, jj, qq = ii[m], jj[m], qt[m]
                order = np.argsort(-qq, kind="stable")
                sh = ((ii[order].astype(np.uint32) << np.uint32(16)) | jj[order].astype(np.uint32)).astype(np.uint32)
                self.offset[gi, gj] = off
                self.q_host[gi, gj] = qq[order].astype(np.float32)
                sh_all.append(sh)
                q_all.append(self.q_host[gi, gj])
                w_all.append(np.full(sh.size, (wi << 16) | wj, dtype=np.uint32))
                off += sh.size
        self.sh_host = np.concatenate(sh_all) if off else np.zeros(1, dtype=np.uint32)
        self.wij_host = np.concatenate(w_all) if off else np.zeros(1, dtype=np.uint32)
        assert layout.nao < 65536, "tile-pair AO offsets are packed into 16 bits"
        aol = np.asarray(layout.ao_loc).astype(np.uint32)
        self.ao_host = ((aol[self.sh_host >> np.uint32(16)] << np.uint32(16)) | aol[self.sh_host & np.uint32(0xffff)]).astype(np.uint32)
        if host_only:
            return
        if off:
            self.sh = torch.from_numpy(self.sh_host.view(np.int32)).to(dev)
            self.q = torch.from_numpy(np.concatenate(q_all)).to(dev)
        else:
            self.sh = torch.zeros(1, dtype=torch.int32, device=dev)
            self.q = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ao = torch.from_numpy(self.ao_host.view(np.int32)).to(dev)
        # primitive-pair prefactor table of every tile pair (jqc_pair_table), fp64 and (lazily) fp32
        nshp = (self.wij_host >> np.uint32(16)).astype(np.int64) * (self.wij_host & np.uint32(0xffff)).astype(np.int64)
        pp_off = np.concatenate([[0], np.cumsum(nshp)])
        assert pp_off[-1] < 2 ** 32
        self.pp_off = torch.from_numpy(pp_off[:-1].astype(np.uint32).view(np.int32)).to(dev)
        self.pair_tab = torch.empty(max(int(pp_off[-1]), 1) * 27, dtype=torch.float64, device=dev)
        self._pair_tab32 = None
        if off:
            wij_d = torch.from_numpy(self.wij_host.view(np.int32)).to(dev)
            _lib.check(_lib.lib().jqc_pair_table(layout.basis_data_fp64["packed"].data_ptr(), self.sh.data_ptr(),
                                                 wij_d.data_ptr(), self.pp_off.data_ptr(), int(off),
                                                 self.pair_tab.data_ptr(), _lib.stream_ptr()))

    def pair_tab32(self):
        if self._pair_tab32 is None:
            self._pair_tab32 = self.pair_tab.float()
        return self._pair_tab32


def build_tile_plan(layout, tt: "_TileTables", log_cut: float, log_max_dm: float, want, shard=None, log_cut64=None, split=None):
    """Task tables of the tiled kernels for one call: {ang: (int32[ntasks, 8], nblocks, nprim patterns, coarse index)}
    with rows (ij0, nij, kl0, nkl, nchunk, blk0, counter slot, kchunk | nsplit << 16) (include/jqc_hip.h).

    ``log_cut64`` + ``split(ang)`` (mixed precision, single rank): the precision windows applied to whole TILE PAIRS.  A task row
    (strip of the Schwarz-sorted bra list x leading ket pairs) is cut at the first ket pair whose bound with the strip's STRONGEST
    bra pair is at or below cutoff_fp64: every quartet behind the cut has an estimate <= cutoff_fp64 whatever its bra pair, i.e.
    lies in the reference's FP32 window (screen_jk_tasks.cu:241-261), and goes to the FP32 kernel of the class; everything in
    front of it goes to the FP64 kernel (its FP32-eligible quartets included).  No tile pair is staged by both launches -- the
    second staging pass is what made the per-quartet split of the reference (jk.py:293-328) a loss on this chip.  Returns
    (plans of the FP64 parts, plans of the FP32 parts)."""
    from ..constants import tile_width
    gkey = layout.group_key
    ng = layout.ngroups
    pair_cut = math.log(PAIR_CUTOFF) - log_max_dm
    nkeep = {k: int(np.searchsorted(-q, -pair_cut, side="left")) for k, q in tt.q_host.items()}
    per_class: Dict[Tuple[int, int, int, int], list] = {}
    per_class32: Dict[Tuple[int, int, int, int], list] = {}
    strip_no = 0
    rank, world = shard if shard is not None else (0, 1)
    two = log_cut64 is not None and world == 1
    for gi in range(ng):
        for gj in range(gi + 1):
            nij_all = nkeep.get((gi, gj), 0)
            if nij_all == 0:
                continue
            qij = tt.q_host[gi, gj]
            for gk in range(gi + 1):
                for gl in range(gk + 1):
                    nkl_all = nkeep.get((gk, gl), 0)
                    if nkl_all == 0:
                        continue
                    ang = (int(gkey[gi, 0]), int(gkey[gj, 0]), int(gkey[gk, 0]), int(gkey[gl, 0]))
                    if not want(ang):
                        continue
                    qkl = tt.q_host[gk, gl]
                    rows = per_class.setdefault(ang, [])
                    cut_here = two and (split is None or split(ang))
                    strip = max(1, -(-nij_all // STRIPS_PER_LIST))
                    for s0 in range(0, nij_all, strip):
                        n_ij = min(strip, nij_all - s0)
                        thr = log_cut - log_max_dm - float(qij[s0])
                        n_kl = min(nkl_all, int(np.searchsorted(-qkl, -thr, side="left")))
                        if n_kl <= 0:
                            break
                        npr = (int(gkey[gi, 1]), int(gkey[gj, 1]), int(gkey[gk, 1]), int(gkey[gl, 1]))
                        n64 = n_kl
                        if cut_here:       # ket pairs [n64, n_kl): bound <= cutoff_fp64 with the strongest bra pair of the strip
                            n64 = min(n_kl, int(np.searchsorted(-qkl, -(log_cut64 - log_max_dm - float(qij[s0])), side="left")))
                            if n_kl > n64:
                                per_class32.setdefault(ang, []).append((tt.offset[gi, gj] + s0, n_ij, tt.offset[gk, gl] + n64,
                                                                        n_kl - n64, npr))
                        if n64 > 0:
                            rows.append((tt.offset[gi, gj] + s0, n_ij, tt.offset[gk, gl], n64, npr))
    if world > 1:
        owner, load = _shard_assign(per_class, world)          # (one O(rows x world) pass per plan)
        build_tile_plan.last_predicted_load = load             # ns per rank (diagnostics, tests)
        per_class = _shard_rows(per_class, rank, world, owner)
    plans = _finish_tile_plans(per_class)
    if log_cut64 is not None:
        return plans, _finish_tile_plans(per_class32)
    return plans


def _finish_tile_plans(per_class):
    """Launch geometry (ket chunks, workgroup splits, block offsets, coarse index) of every class's task rows."""
    from ..constants import tile_width
    plans = {}
    for ang, rows in per_class.items():
        if not rows:
            continue
        # ket chunk: consecutive ket tile pairs walked by ONE workgroup (bra-side staging amortised); as long as
        # the class still fills the chip (>= TARGET_WGS workgroups) the chunks grow up to KCHUNK_MAX
        nblk1 = sum(r[1] * r[3] for r in rows)
        kchunk = int(min(KCHUNK_MAX, max(1, nblk1 // TARGET_WGS)))
        # hard limit of a launch: the dispatch packet counts WORK-ITEMS in 32 bits, i.e. at most 2^24 - 1 workgroups of 256 --
        # a larger grid is silently truncated (found with KCHUNK_MAX = 1 on the 425-atom molecule: J off by 70 %).  Longer ket
        # chunks keep every class under it whatever the molecule size and the knobs.
        kchunk = max(kchunk, -(-nblk1 // MAX_WGS_PER_LAUNCH))
        # small launches (few tile pairs, e.g. the f classes of a small molecule): deal the candidates of every tile
        # pair to nsplit workgroups so that the class still spreads over the chip
        nq = 1
        for l in ang:
            nq *= tile_width(l)
        nsplit = 1
        if kchunk == 1 and nblk1 < SPLIT_BELOW_WGS:
            nsplit = int(max(1, min(NSPLIT_MAX, SPLIT_BELOW_WGS // max(nblk1, 1), nq // 16)))
        tab = np.zeros((len(rows), 8), dtype=np.int32)
        blk = 0
        align = CHUNK_ALIGN if nsplit == 1 else 1
        for n, (ij0, nij, kl0, nkl, _) in enumerate(rows):
            nchunk = -(-nkl // kchunk)
            if align > 1 and nchunk > align:
                # XCD affinity: chunk count and first block of the row are multiples of 8 (the surplus workgroups return at once),
                # so ket chunk c of EVERY bra pair runs on XCD (c mod 8) and its ket-side tables are shared through one L2
                nchunk = -(-nchunk // align) * align
                blk = -(-blk // align) * align
            tab[n] = (ij0, nij, kl0, nkl, nchunk, blk, 0, kchunk | (nsplit << 16))
            blk += nij * nchunk * nsplit
        if blk * 512 >= 2 ** 32:                 # (512: the widest workgroup of the tiled kernels)
            raise RuntimeError(f"class {ang}: {blk} workgroups exceed the 32-bit work-item count of one launch")
        # coarse index: task row of every 256th workgroup (the kernel probes forward from there)
        starts = tab[:, 5].astype(np.int64)
        index = (np.searchsorted(starts, np.arange(0, blk + 256, 256), side="right") - 1).astype(np.int32)
        plans[ang] = (tab, blk, [r[4] for r in rows], index)
    return plans


def _shard_assign(per_class, world):
    """Deterministic assignment of the task rows to ``world`` ranks: {class: [rank of every row]} and the predicted load
    (ns) of every rank.  Classes are taken in order of decreasing cost (tile-pair products x measured ns per quartet of
    the class, gfx950_scheme.json; the FLOP model where unmeasured).  A class worth more than half a rank's fair share is
    dealt ROW BY ROW (strips of the Schwarz-sorted bra list), each row -- heaviest first -- to the least loaded rank
    (longest-processing-time rule: the rows of one class differ a lot in cost, the leading strips see the longest ket
    lists, so a round-robin deal hands rank 0 the heaviest row of every group); a cheaper class goes as a whole to the
    least loaded rank, so a small molecule does not pay every class's launch on every rank."""
    from ..roofline import quartet_flops
    measured = _router.class_cost_table()          # ns per quartet of the class on gfx950, where measured
    unit = {a: measured.get(_router.class_key(a), 1.5e-4 * float(quartet_flops(a))) for a in per_class}
    cost = {a: sum(r[1] * r[3] for r in rows) * unit[a] for a, rows in per_class.items()}
    fair = sum(cost.values()) / world
    load = [0.0] * world
    owner = {}
    for a in sorted(per_class, key=lambda a: (-cost[a], a)):
        rows = per_class[a]
        if cost[a] > 0.5 * fair and len(rows) >= world:
            w = [r[1] * r[3] * unit[a] for r in rows]
            own = [0] * len(rows)
            for n in sorted(range(len(rows)), key=lambda n: (-w[n], n)):
                tgt = min(range(world), key=lambda k: (load[k], k))
                load[tgt] += w[n]
                own[n] = tgt
            owner[a] = own
        else:
            tgt = min(range(world), key=lambda k: (load[k], k))
            load[tgt] += cost[a]
            owner[a] = [tgt] * len(rows)
    return owner, load


def _shard_rows(per_class, rank, world, owner=None):
    """This rank's share of the task rows (every rank evaluates the same assignment, ``_shard_assign``)."""
    if owner is None:
        owner, _ = _shard_assign(per_class, world)
    mine = {}
    for a, rows in per_class.items():
        keep = [r for r, o in zip(rows, owner[a]) if o == rank]
        if keep:
            mine[a] = keep
    return mine


_FIRST_USE_OK = set()       # kernel builds outside the verified manifest that passed their first-use cross-check (per process)

NDM2 = __import__('os').environ.get('JQC_NDM2', '1') != '0'      # 0: one density matrix per integral evaluation (A/B, diagnostics)
CHUNK_ALIGN = int(__import__('os').environ.get('JQC_CHUNK_ALIGN', '1'))   # 8: ket chunk <-> XCD affinity (needs kernels with the surplus-workgroup guard)
KCHUNK_MAX = int(__import__('os').environ.get('JQC_KCHUNK_MAX', '16'))
SPLIT_BELOW_WGS = int(__import__('os').environ.get('JQC_SPLIT_BELOW', '1024'))
NSPLIT_MAX = int(__import__('os').environ.get('JQC_NSPLIT_MAX', '8'))
TARGET_WGS = int(__import__('os').environ.get('JQC_TARGET_WGS', '4096'))
MAX_WGS_PER_LAUNCH = 4_000_000       # < 2^32 / 512 work-items per launch (see build_tile_plan)


def generate_jk_kernel(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, shard=None):
    """``shard=(rank, world_size)``: this process evaluates its share of the quartet work and the raw
    Fock contributions are summed over ranks with ONE all-reduce (RCCL) before the epilogue."""
    import torch
    log_cutoff_fp64 = float(np.float32(math.log(cutoff_fp64)))
    log_cutoff_fp32 = float(np.float32(math.log(cutoff_fp32)))
    mixed = cutoff_fp32 < cutoff_fp64
    fp32_only = cutoff_fp64 >= 1e30          # an explicit all-fp32 request (reference tests use 1e100) is always honoured
    layout = basis_layout
    nbas = layout.nbasis
    nao = layout.nao
    # the kernels index nbas x nbas tables and order shell pairs by ish * nbas + jsh in 32-bit integers
    assert nbas <= 46340, "more than 46340 (padded) shells are not supported by the J/K kernels"
    from ..constants import tile_width
    # the tiled kernels need every (l, nprim) group padded to its tile width
    tiled_layout = all((layout.group_offset[g + 1] - layout.group_offset[g]) % tile_width(int(layout.group_key[g, 0])) == 0
                       for g in range(layout.ngroups))
    state = {"pairs": {}, "tiles": {}, "queue": None, "stats": {}, "probe": None, "plan_cache": {}, "streams": None,
             "nstreams": N_STREAMS, "auto_streams": "JQC_STREAMS" not in __import__('os').environ}

    def get_jk(mol_ref=None, dm=None, hermi=0, vhfopt=None, with_j=True, with_k=True, omega=None, verbose=None,
               _classes=None):
        """Compute J, K; compatible with ``pyscf.scf.hf.get_jk`` / the reference closure (jk.py:109-118).
        ``mol_ref`` is ignored in favour of the layout captured at ``apply`` time (jk.py:123).
        ``_classes`` (internal): predicate on the angular class (li, lj, lk, ll); classes it rejects are left out
        (the pair-based backend evaluates those itself, joltqc_amd/pyscf/jk_pair.py)."""
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
        if hermi == 0:
            # hermi = 0 is the DEFAULT of this signature (reference jk.py:109) and of pyscf.scf.hf.get_jk: most callers that leave it
            # pass symmetric matrices.  Those take the hermi = 1 path -- one matrix instead of the stacked [D, D^T] (1.49x a
            # hermi = 1 call) -- with the same result: for D = D^T the reference's epilogue vj[:n] + vj[n:]^T is 2 vj[:n]
            asym = torch.stack([(dms - dms.transpose(1, 2)).abs().max(), dms.abs().max()]).tolist()
            if asym[0] <= 1e-14 * asym[1]:
                hermi = 1

        # shell-block density bounds (max_block_pooling, linalg_helper.py:125)
        dm_cond = torch.empty((nbas, nbas), dtype=torch.float32, device=dev)
        _lib.check(L.jqc_shell_block_max(dms.data_ptr(), dms.shape[0], nao, layout.device_ao_loc().data_ptr(), nbas,
                                         dm_cond.data_ptr(), stream))
        if hermi == 0:
            dm_cond = dm_cond + dm_cond.T
        log_dm_cond = torch.log(dm_cond.double() + 1e-300).float().contiguous()
        # d_large is floored at -36.8 inside the predicate (screen_jk_tasks.cu:241), so is every bound here
        log_max_dm = max(float(log_dm_cond.max().item()), -36.8)

        if hermi == 0:
            dms = torch.cat([dms, dms.transpose(1, 2)], dim=0).contiguous()   # jk.py:189-191
        n_dm = dms.shape[0]
        dms_fp32 = dms.float().contiguous() if mixed else None
        fock = torch.zeros((int(with_j) + int(with_k),) + tuple(dms.shape), dtype=torch.float64, device=dev)
        vj = fock[0] if with_j else None
        vk = fock[-1] if with_k else None
        vj_p = vj.data_ptr() if with_j else None
        vk_p = vk.data_ptr() if with_k else None
        b64 = layout.basis_data_fp64["packed"]
        b32 = layout.basis_data_fp32["packed"] if mixed else None
        is_tile = lambda ang: (_router.select_algo(ang) & 0xf) in (_lib.ALGO_TILE, _lib.ALGO_TILE1Q, _lib.ALGO_TILE512)
        n_launch = 0
        counter_bufs = []
        tile_counts = None
        INF = 3.0e38

        def run_queue(want, vj_q, vk_q, cut32, cut64, use_fp32, shard_q, record):
            """Screening launch + one launch per class of the queue-driven one-quartet-per-lane kernels (jk_1q1t.hip) for the
            classes ``want`` selects, accumulating into vj_q / vk_q."""
            nonlocal n_launch
            if om not in state["pairs"]:
                state["pairs"][om] = _PairTables(layout, om)
            pt = state["pairs"][om]
            plans = build_screen_plan(layout, pt, cut32, log_max_dm, QUEUE_DEPTH, want, shard_q)
            qsize = max((p["total"] for p in plans), default=0)
            if plans and (state["queue"] is None or state["queue"].numel() < qsize * 4):
                state["queue"] = torch.empty(max(qsize, 1) * 4, dtype=torch.int16, device=dev)
            queue = state["queue"]
            for p in plans:
                ncls = len(p["classes"])
                tasks_d = torch.from_numpy(p["tasks"]).to(dev)
                region_d = torch.from_numpy(p["region"]).to(dev)
                counters = torch.zeros((ncls, 2), dtype=torch.int32, device=dev)
                if record:
                    counter_bufs.append((counters, p))
                _lib.check(L.jqc_screen_jk_tasks(tasks_d.data_ptr(), p["tasks"].shape[0], p["nblocks"], pt.sh.data_ptr(),
                                                 pt.q.data_ptr(), log_dm_cond.data_ptr(), nbas, int(with_j), int(with_k),
                                                 cut32, cut64, log_max_dm, queue.data_ptr(),
                                                 region_d.data_ptr(), counters.data_ptr(), stream))
                n_launch += 1
                for n, ang in enumerate(p["classes"]):
                    beg, end = int(p["region"][n, 0]), int(p["region"][n, 1])
                    h64 = _router.gen_jk_kernel(ang, do_j=with_j, do_k=with_k, rys_lr=lr, fp32=False, algo=_lib.ALGO_1Q1T)
                    _lib.check(L.jqc_jk_launch(h64, nao, b64.data_ptr(), dms.data_ptr(), vj_q, vk_q, om,
                                               queue.data_ptr() + beg * 8, counters.data_ptr() + (2 * n) * 4,
                                               end - beg, 1, n_dm, stream))
                    n_launch += 1
                    if use_fp32:
                        h32 = _router.gen_jk_kernel(ang, do_j=with_j, do_k=with_k, rys_lr=lr, fp32=True,
                                                    algo=_lib.ALGO_1Q1T)
                        _lib.check(L.jqc_jk_launch(h32, nao, b32.data_ptr(), dms_fp32.data_ptr(), vj_q, vk_q,
                                                   om, queue.data_ptr() + (end - 1) * 8,
                                                   counters.data_ptr() + (2 * n + 1) * 4, end - beg, -1, n_dm, stream))
                        n_launch += 1
            return len(plans)

        def tile_launch(h, fp32_kernel, vj_x, vk_x, lo, hi, cnt, sp_, tab_p, ntab, nblk, idx_p, cnt32=None):
            tt = state["tiles"][om]
            bas, dmat = (b32, dms_fp32) if fp32_kernel else (b64, dms)
            ptab = tt.pair_tab32() if fp32_kernel else tt.pair_tab
            _lib.check(L.jqc_jk_tile_launch(h, nao, bas.data_ptr(), dmat.data_ptr(), vj_x, vk_x, om, tab_p, ntab, nblk,
                                            tt.sh.data_ptr(), tt.q.data_ptr(), tt.q_dev.data_ptr(), log_dm_cond.data_ptr(),
                                            nbas, lo, hi, log_max_dm, n_dm, cnt, idx_p, tt.ao.data_ptr(),
                                            tt.pp_off.data_ptr(), ptab.data_ptr(), cnt32, sp_))

        def first_use_check(ang, algo_req, fp32_kernel, h, bucket, fused=False):
            """A tile-kernel build that is not in the verified manifest (joltqc_amd/data/verified_kernels.json: another
            variant, edited sources, another compiler) is run once against the independent one-quartet-per-lane kernel on
            this call's own inputs (whole class, unsharded); a build that disagrees raises instead of contributing to J/K."""
            built = _router.resolved_algo(ang, with_j, with_k, lr, fp32_kernel, algo_req)
            key = _router.kernel_key(ang, with_j, with_k, lr, fp32_kernel, built)
            if key in _FIRST_USE_OK or _router.is_verified(ang, with_j, with_k, lr, fp32_kernel, built):
                return
            tab, nblk, _, index = build_tile_plan(layout, state["tiles"][om], log_cutoff_fp32, bucket,
                                                  lambda a: tuple(a) == tuple(ang), None)[tuple(ang)]
            tab_d, index_d = torch.from_numpy(tab).to(dev), torch.from_numpy(index).to(dev)
            scratch = torch.zeros((2,) + tuple(fock.shape), dtype=torch.float64, device=dev)
            pj = lambda m: scratch[m, 0].data_ptr() if with_j else None
            pk = lambda m: scratch[m, -1].data_ptr() if with_k else None
            run_queue(lambda a: tuple(a) == tuple(ang), pj(1), pk(1), log_cutoff_fp32, log_cutoff_fp32, False, None, False)
            ref = float(scratch[1].abs().max().item())
            # a fused (JQC_VARIANT_MIXED) build is checked twice: every quartet through its FP64 phase (window (cut, cut] empty),
            # then every quartet through its packed-FP32 phase (window (cut, inf))
            for hi, loose in (((log_cutoff_fp32, False), (INF, True)) if fused else ((INF, fp32_kernel),)):
                scratch[0].zero_()
                tile_launch(h, fp32_kernel, pj(0), pk(0), log_cutoff_fp32, hi, None, stream, tab_d.data_ptr(), tab.shape[0], nblk,
                            index_d.data_ptr())
                err = float((scratch[0] - scratch[1]).abs().max().item())
                tol = (2e-4 if loose else 1e-9) * max(ref, 1e-300)
                if not err <= tol:
                    raise RuntimeError(f"J/K kernel build {key} disagrees with the one-quartet-per-lane reference kernel on its "
                                       f"first use (max |diff| {err:.3e}, largest element {ref:.3e}): the build is rejected")
            _FIRST_USE_OK.add(key)

        # ---------------- tiled kernels: no queue, one launch per angular class, classes spread over streams
        if tiled_layout:
            if om not in state["tiles"]:
                state["tiles"][om] = _TileTables(layout, om)
            tt = state["tiles"][om]
            # the plan only depends on the density through log_max_dm: bucket it (upwards = looser, safe)
            bucket = math.ceil(log_max_dm * 2.0) / 2.0
            # mixed precision by TILE PAIRS (build_tile_plan): the classes of the measured table (gfx950_scheme.json
            # "fp32_tile_split") hand the tile pairs whose bound is at or below cutoff_fp64 to their FP32 kernel
            # (J+K calls only: the mode the table was measured in and whose FP32 builds are compiled ahead of time and gated)
            tsplit = (lambda a: _router.fp32_tile_split(a)) if (mixed and not fp32_only and shard is None and with_j and with_k) else None
            # an INCREMENT of a density matrix (get_veff sets ``increment_of`` = largest element of the full matrix) is cut where a
            # build of the full matrix would be cut: FP32 rounding relative to the increment's own size would otherwise spread
            # over ever more tile pairs as the increments shrink (the grid path's lesson)
            ref = state.get("increment_of")
            ref_bucket = max(bucket, math.ceil(math.log(ref) * 2.0) / 2.0) if ref else bucket
            pkey = (om, bucket, shard, tsplit is not None and (os.environ.get("JQC_FP32_TILE_SPLIT"), ref_bucket))        # (with_j / with_k do not
                                                                                                                          #  enter the plan otherwise)
            if pkey not in state["plan_cache"]:
                if tsplit is not None:
                    tplans, tplans32 = build_tile_plan(layout, tt, log_cutoff_fp32, bucket, is_tile, shard,
                                                       log_cut64=log_cutoff_fp64 - (ref_bucket - bucket), split=tsplit)
                else:
                    tplans, tplans32 = build_tile_plan(layout, tt, log_cutoff_fp32, bucket, is_tile, shard), {}
                entry = None
                if tplans or tplans32:
                    from ..roofline import quartet_flops
                    plan_of = lambda it: (tplans32 if it[1] else tplans)[it[0]]
                    items = [(a, False) for a in tplans] + [(a, True) for a in tplans32]      # (class, FP32 part?)
                    cost = {it: sum(int(r[1]) * int(r[3]) for r in plan_of(it)[0]) * quartet_flops(it[0]) for it in items}
                    order = sorted(items, key=lambda it: -cost[it])              # longest first over the streams
                    tabs = np.concatenate([plan_of(it)[0] for it in order])
                    tabs[:, 6] = np.arange(tabs.shape[0])                       # counter slot of every task row
                    index = np.concatenate([plan_of(it)[3] for it in order])
                    ioff, pos = {}, 0
                    for it in order:
                        ioff[it] = pos
                        pos += plan_of(it)[3].size
                    entry = {"order": order, "tabs_d": torch.from_numpy(tabs).to(dev), "nrows": tabs.shape[0],
                             "index_d": torch.from_numpy(index).to(dev), "index_off": ioff,
                             "plans": tplans, "plans32": tplans32,
                             "row_meta": [(it[0], npr) for it in order for npr in plan_of(it)[2]]}
                if len(state["plan_cache"]) > 64:
                    state["plan_cache"].clear()
                state["plan_cache"][pkey] = entry
            entry = state["plan_cache"][pkey]
            if entry is not None:
                order, tabs_d, tplans, tplans32 = entry["order"], entry["tabs_d"], entry["plans"], entry["plans32"]
                # 32 leading slots: diagnostic cycle stamps of -DSTAMPS=1 kernel builds (tools/stamps_profile.py)
                counts_buf = torch.zeros(32 + 2 * entry["nrows"], dtype=torch.int64, device=dev)
                tile_counts = counts_buf[32:].view(2, entry["nrows"])
                state["stats"]["stamps"] = counts_buf[:32]
                nst = state["nstreams"]
                if state["auto_streams"] and sum(int((tplans32 if f32 else tplans)[a][1]) for a, f32 in order) > BIG_CALL_WGS:
                    nst = min(nst, N_STREAMS_BIG)
                if state["streams"] is None or len(state["streams"]) != nst:
                    state["streams"] = [torch.cuda.Stream(device=dev) for _ in range(nst)]
                side = state["streams"]
                if mixed:
                    tt.pair_tab32()            # created on the current stream BEFORE the side streams fork from it
                cur = torch.cuda.current_stream()
                ev = torch.cuda.Event()
                ev.record(cur)
                for st_ in side:
                    st_.wait_event(ev)
                row = 0
                only = __import__('os').environ.get("JQC_ONLY_CLASS")
                for n, (ang, part32) in enumerate(order):
                    tab, nblk, _, _ = (tplans32 if part32 else tplans)[ang]
                    if (only and "%d%d%d%d" % tuple(ang) not in only.split(",")) or (_classes is not None and not _classes(ang)):
                        row += tab.shape[0]
                        continue
                    idx_p = entry["index_d"].data_ptr() + entry["index_off"][(ang, part32)] * 4
                    sid = side[n % len(side)]
                    sp = sid.cuda_stream
                    if part32:
                        # tile pairs whose every quartet lies in the FP32 window: the FP32 kernel of the class, nobody else
                        algo32 = _router.select_algo(ang, True)
                        if n_dm > 1 and NDM2 and _router.supports_ndm2(ang, algo32):
                            algo32 |= _router.VARIANT_NDM2
                        h32 = _router.gen_jk_kernel(ang, do_j=with_j, do_k=with_k, rys_lr=lr, fp32=True, algo=algo32)
                        first_use_check(ang, algo32, True, h32, bucket)
                        probing = state["probe"] is not None and (state["probe"] == "all" or tuple(state["probe"]) == tuple(ang))
                        if probing:
                            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            ev0.record(sid)
                        tile_launch(h32, True, vj_p, vk_p, log_cutoff_fp32, INF, tile_counts[1].data_ptr(), sp,
                                    tabs_d.data_ptr() + row * 32, tab.shape[0], nblk, idx_p)
                        n_launch += 1
                        if probing:
                            ev1.record(sid)
                            state["stats"].setdefault("probe_events", []).append((ev0, ev1))
                            state["stats"].setdefault("probe_classes", []).append(tuple(ang))
                        row += tab.shape[0]
                        continue
                    tile_split_class = tsplit is not None and tsplit(ang)
                    algo64 = _router.select_algo(ang, small=nblk < TARGET_WGS)
                    if n_dm > 1 and NDM2 and _router.supports_ndm2(ang, algo64):
                        # every pair of density matrices is contracted against ONE evaluation of the integrals
                        # (reference jk/1q1t.cu:423-638); the kernel walks n_dm in pairs
                        algo64 |= _router.VARIANT_NDM2
                    h64 = _router.gen_jk_kernel(ang, do_j=with_j, do_k=with_k, rys_lr=lr, fp32=False, algo=algo64)
                    geo = (tabs_d.data_ptr() + row * 32, tab.shape[0], nblk, idx_p)
                    first_use_check(ang, algo64, False, h64, bucket)
                    probing = state["probe"] is not None and (state["probe"] == "all" or tuple(state["probe"]) == tuple(ang))
                    if probing:
                        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ev0.record(sid)
                    # mixed precision: quartets with estimate in (cutoff_fp32, cutoff_fp64] go to the fp32 kernel of the class
                    # -- where that kernel is measured faster per quartet than the fp64 one (gfx950_scheme.json "fp32_pays");
                    # elsewhere the fp64 kernel takes both windows in ONE launch (more accurate and, on this chip, faster:
                    # two launches stage and screen every tile pair twice)
                    split = mixed and not tile_split_class and (fp32_only or _router.fp32_pays(ang))
                    # ... or BOTH windows in one launch of the fused build (JQC_VARIANT_MIXED: FP64 phase + packed-FP32 phase, two
                    # quartets per lane, behind one staging / screening / flush of every tile pair)
                    fused = mixed and not split and not tile_split_class and n_dm == 1 and _router.mixed_fused(ang, algo64)
                    if fused:
                        amx = _router.mixed_variant(ang, algo64)
                        hmx = _router.gen_jk_kernel(ang, do_j=with_j, do_k=with_k, rys_lr=lr, fp32=False, algo=amx)
                        fused = bool(_router.resolved_algo(ang, with_j, with_k, lr, False, amx) & _router.VARIANT_MIXED)
                    if fused:
                        first_use_check(ang, amx, False, hmx, bucket, fused=True)
                        tile_launch(hmx, False, vj_p, vk_p, log_cutoff_fp32, log_cutoff_fp64, tile_counts[0].data_ptr(), sp, *geo,
                                    cnt32=tile_counts[1].data_ptr())
                    else:
                        tile_launch(h64, False, vj_p, vk_p, log_cutoff_fp64 if split else log_cutoff_fp32, INF,
                                    tile_counts[0].data_ptr(), sp, *geo)
                    n_launch += 1
                    if split:
                        algo32 = _router.select_algo(ang, True)
                        if n_dm > 1 and NDM2 and _router.supports_ndm2(ang, algo32):
                            algo32 |= _router.VARIANT_NDM2
                        h32 = _router.gen_jk_kernel(ang, do_j=with_j, do_k=with_k, rys_lr=lr, fp32=True, algo=algo32)

                        first_use_check(ang, algo32, True, h32, bucket)
                        tile_launch(h32, True, vj_p, vk_p, log_cutoff_fp32, log_cutoff_fp64, tile_counts[1].data_ptr(), sp, *geo)
                        n_launch += 1
                    if probing:
                        ev1.record(sid)
                        state["stats"].setdefault("probe_events", []).append((ev0, ev1))
                        state["stats"].setdefault("probe_classes", []).append(tuple(ang))
                    row += tab.shape[0]
                for st_ in side:
                    e2 = torch.cuda.Event()
                    e2.record(st_)
                    cur.wait_event(e2)
                state["stats"]["tile_rows"] = entry["row_meta"]

        # ---------------- queue path (one quartet per lane) for the classes routed to it
        want_q = (lambda ang: not is_tile(ang) and (_classes is None or _classes(ang))) if tiled_layout else _classes
        need_queue = (not tiled_layout) or any(
            not is_tile((int(a), int(b), int(c), int(d)))
            for a in set(layout.angs) for b in set(layout.angs) for c in set(layout.angs) for d in set(layout.angs)
            if a >= b and a >= c and c >= d)
        if need_queue:
            run_queue(want_q, vj_p, vk_p, log_cutoff_fp32, log_cutoff_fp64, mixed, shard, True)

        if shard is not None and shard[1] > 1 and not getattr(get_jk, "local_only", False):
            # (`local_only`: diagnostic -- this rank's partial Fock matrices without the collective, so that one rank can time
            #  its own share of the kernels while the others wait: tests/test_configs_gpu.py, bench.py per_rank)
            import torch.distributed as dist
            dist.all_reduce(fock)                 # the one collective of the path: sum of raw J/K over ranks
            vj = fock[0] if with_j else None
            vk = fock[-1] if with_k else None

        # epilogue (reference jk.py:350-370)
        if with_j:
            if hermi == 1:
                vj = vj * 2.0
            else:
                h = n_dm // 2
                vj = vj[:h] + vj[h:].transpose(1, 2)
            vj = vj + vj.transpose(1, 2)
            if getattr(get_jk, "keep_internal", False):      # (bench.py's parity figure: shell blocks in the internal AO order)
                state["stats"]["vj_internal"] = vj
            vj = layout.dm_to_mol(vj).reshape(out_shape)
        else:
            vj = 0
        if with_k:
            if hermi == 1:
                vk = vk + vk.transpose(1, 2)
            else:
                h = n_dm // 2
                vk = vk[:h] + vk[h:].transpose(1, 2)
            if getattr(get_jk, "keep_internal", False):
                state["stats"]["vk_internal"] = vk
            vk = layout.dm_to_mol(vk).reshape(out_shape)
        else:
            vk = 0

        st = state["stats"]
        st["launches"] = n_launch
        st["chunks"] = len(counter_bufs)
        st["counter_bufs"] = counter_bufs          # read lazily by quartet_counts()
        st["tile_counts"] = tile_counts
        st["host_seconds"] = time.perf_counter() - t_start
        if isinstance(dm_in, np.ndarray) and getattr(get_jk, "return_numpy", False):
            vj = vj.cpu().numpy() if with_j else 0
            vk = vk.cpu().numpy() if with_k else 0
        return vj, vk

    def quartet_counts():
        """(n_fp64, n_fp32, {(ang, nprim): [n64, n32]}) quartets dispatched by the last call (synchronises)."""
        n64 = n32 = 0
        per_class = {}
        tc = state["stats"].get("tile_counts")
        if tc is not None:
            c = tc.cpu().numpy()
            n64 += int(c[0].sum())
            n32 += int(c[1].sum())
            for n, key in enumerate(state["stats"]["tile_rows"]):
                a = per_class.setdefault(key, [0, 0])
                a[0] += int(c[0, n])
                a[1] += int(c[1, n])
        for counters, p in state["stats"].get("counter_bufs", []):
            c = counters.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            n64 += int(c[:, 0].sum())
            n32 += int(c[:, 1].sum())
            for n, ang in enumerate(p["classes"]):
                a = per_class.setdefault((tuple(ang), None), [0, 0])
                a[0] += int(c[n, 0])
                a[1] += int(c[n, 1])
        return n64, n32, per_class

    def set_probe(ang):
        """Bracket the launches of angular class ``ang`` with HIP events (bench.py roofline leg)."""
        state["probe"] = ang
        state["stats"]["probe_events"] = []
        state["stats"]["probe_classes"] = []

    def set_streams(n):
        """Number of HIP streams the class kernels are spread over (1 = serial launches; None = the default policy: N_STREAMS,
        N_STREAMS_BIG for calls of more than BIG_CALL_WGS workgroups)."""
        if n is None:                                # back to the default policy
            state["nstreams"], state["auto_streams"] = N_STREAMS, "JQC_STREAMS" not in __import__('os').environ
            return
        state["nstreams"] = max(1, int(n))
        state["auto_streams"] = False

    def set_increment_of(dmax):
        """The next calls evaluate increments of a density matrix whose largest element is ``dmax`` (None: full matrices again)."""
        state["increment_of"] = float(dmax) if dmax else None

    get_jk.set_increment_of = set_increment_of
    get_jk.set_streams = set_streams
    get_jk.set_probe = set_probe
    get_jk.quartet_counts = quartet_counts
    get_jk.stats = state["stats"]
    get_jk.layout = layout
    get_jk.return_numpy = False
    return get_jk
