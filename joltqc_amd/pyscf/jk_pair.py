"""Pair-based J/K backend (second algorithm).

Mirrors ``/root/reference/jqc/pyscf/jk_pair.py``: ``generate_jk_kernel`` (:85) returns a ``get_jk`` closure with the
signature of the quartet-based one; ``generate_get_j / generate_get_k / generate_get_jk`` (:49-82).  It is an opt-in
alternative to ``joltqc_amd.pyscf.jk`` (``apply`` installs the tiled kernels, as the reference's ``apply`` installs its
quartet kernels), not a replacement.

What runs where on MI355X:

* **J** -- ``pair_vj`` kernels (joltqc_amd/csrc/kernels/pair_vj.hip; reference jk/pair_vj.cu): a lane owns one bra shell
  pair, the 64 lanes of a wave walk the Schwarz-sorted ket pair list together, ``J_ij`` stays in registers and is written
  once.  The density enters through per-ket-pair coefficients ``E`` built once per call (``jqc_pair_ket_density``), so the
  walk reads neither D nor runs the ket horizontal recurrence.  Classes whose kernel would spill registers (large f
  classes) stay on the tiled J kernels -- same raw-J convention, so the two sums simply add.
* **K** -- the reference's ``pair_vk`` fixes an (i, k) shell pair per block and sums over (j, l) with a block reduction
  (jk/pair_vk.cu:82-522): it harvests ONE of the four K blocks a quartet feeds, i.e. evaluates every integral four times
  compared with the 8-fold symmetric quartet loop.  On this chip the LDS-tile K kernels are cheaper than 4x the
  integrals, so K goes through the tiled K-only kernels (``jk.generate_jk_kernel(..., with_j=False)``); ``pair_wide_vk``
  is accepted for interface parity and unused.
"""
import math
from typing import Dict, Tuple

import numpy as np

from ..backend import lib as _lib
from . import jk as _jk

__all__ = ["generate_get_j", "generate_get_k", "generate_get_jk", "generate_jk_kernel"]

PAIR_CUTOFF = 1e-13      # reference jk_pair.py:44
PAIR_WIDE_VJ = 256       # bra pairs per workgroup (reference :45)
PAIR_WIDE_VK = 64        # reference :46 (unused here, see module docstring)
TARGET_WGS = 2048        # a launch is split over the ket list until it has about this many workgroups
PAIR_MAX_L = 3           # highest angular momentum with ahead-of-time pair kernels (__graft_entry__._compile_pair)


_PAIR_CHECKED = set()      # (source tag, long-range, pair classes) whose first-use cross-check passed in this process

def generate_get_j(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, pair_wide_vk=PAIR_WIDE_VK):
    kern = generate_jk_kernel(basis_layout, cutoff_fp64=cutoff_fp64, cutoff_fp32=cutoff_fp32, pair_wide_vk=pair_wide_vk)

    def get_j(*args, **kwargs):
        return kern(*args, with_j=True, with_k=False, **kwargs)[0]
    return get_j


def generate_get_k(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, pair_wide_vk=PAIR_WIDE_VK):
    kern = generate_jk_kernel(basis_layout, cutoff_fp64=cutoff_fp64, cutoff_fp32=cutoff_fp32, pair_wide_vk=pair_wide_vk)

    def get_k(*args, **kwargs):
        return kern(*args, with_j=False, with_k=True, **kwargs)[1]
    return get_k


def generate_get_jk(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, pair_wide_vk=PAIR_WIDE_VK):
    kern = generate_jk_kernel(basis_layout, cutoff_fp64=cutoff_fp64, cutoff_fp32=cutoff_fp32, pair_wide_vk=pair_wide_vk)

    def get_jk(*args, **kwargs):
        return kern(*args, **kwargs)
    return get_jk


def pair_kernel(la, lb, lc, ld, lr=False, compile_only=False):
    """Handle of the pair_vj kernel of bra class (la lb) and ket class (lc ld), or None when that class does not fit the
    register file (it then stays on the tiled kernels)."""
    rc = _lib.lib().jqc_gen_pair_vj_kernel(int(la), int(lb), int(lc), int(ld), int(bool(lr)), int(bool(compile_only)))
    if rc == -4:
        return None
    return _lib.check(rc)


class _ClassPairs:
    """Per angular pair class (la >= lb): the Schwarz-sorted shell-pair lists of its (l, nprim) group pairs, concatenated;
    one segment per group pair.  Role of ``make_pairs_symmetric`` (reference backend/jk_pair.py:142-193)."""

    def __init__(self, layout, omega):
        import torch
        dev = _lib.require_gpu()
        self.q_dev = layout.q_matrix(omega)
        lists = _jk.make_pair_lists(layout.group_offset, self.q_dev.cpu().numpy(), layout.pad_id)
        gkey = layout.group_key
        by_class: Dict[Tuple[int, int], list] = {}
        for (gi, gj), (sh, q) in lists.items():
            by_class.setdefault((int(gkey[gi, 0]), int(gkey[gj, 0])), []).append((gi, gj, sh, q))
        self.cls = {}
        L = _lib.lib()
        for ab, segs in by_class.items():
            sh = np.concatenate([s[2] for s in segs])
            q = np.concatenate([s[3] for s in segs])
            start = np.concatenate([[0], np.cumsum([len(s[2]) for s in segs])]).astype(np.int64)
            seg = np.stack([start[:-1], np.diff(start)], 1).astype(np.int32)
            n = int(sh.size)
            sh_d = torch.from_numpy(sh.view(np.int32)).to(dev)
            tab = torch.empty(n * 27, dtype=torch.float64, device=dev)
            ones = torch.full((n,), (1 << 16) | 1, dtype=torch.int32, device=dev)
            off = torch.arange(n, dtype=torch.int32, device=dev)
            _lib.check(L.jqc_pair_table(layout.basis_data_fp64["packed"].data_ptr(), sh_d.data_ptr(), ones.data_ptr(),
                                        off.data_ptr(), n, tab.data_ptr(), _lib.stream_ptr()))
            self.cls[ab] = {"n": n, "sh": sh_d, "q": torch.from_numpy(q).to(dev), "q_host": q, "tab": tab,
                            "seg": torch.from_numpy(seg).to(dev), "seg_host": seg,
                            "nprim": [(int(gkey[s[0], 1]), int(gkey[s[1], 1])) for s in segs]}


def generate_jk_kernel(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, pair_wide_vk=PAIR_WIDE_VK, tile_jk=None,
                       max_l=PAIR_MAX_L):
    """Pair-based ``get_jk`` for ``basis_layout`` (a tile-aligned layout, as for ``jk.generate_jk_kernel``).  FP64 only:
    ``cutoff_fp32`` is the screening threshold, ``cutoff_fp64`` is accepted for interface parity.  ``tile_jk``: an existing
    tiled ``get_jk`` of the same layout and cutoffs to share (K and the J classes without a pair kernel go through it);
    ``max_l``: classes with a higher angular momentum stay on the tiled kernels (their pair kernels are not in the
    ahead-of-time set of ``__graft_entry__.build`` and take minutes to generate)."""
    import os
    import torch
    layout = basis_layout
    nao, nbas = layout.nao, layout.nbasis
    log_cutoff = float(np.float32(math.log(min(cutoff_fp32, cutoff_fp64))))
    if tile_jk is None:
        tile_jk = _jk.generate_jk_kernel(layout, cutoff_fp64=cutoff_fp64, cutoff_fp32=cutoff_fp32)
    state = {"pairs": {}, "stats": {}, "checked": set()}

    def supported(ab, cd, lr):
        """Both directions of the canonical class (ab|cd) have a spill-free pair kernel."""
        if max(ab + cd) > max_l:
            return False
        return pair_kernel(*ab, *cd, lr=lr) is not None and pair_kernel(*cd, *ab, lr=lr) is not None

    def get_jk(mol_ref=None, dm=None, hermi=0, vhfopt=None, with_j=True, with_k=True, omega=None, verbose=None):
        assert with_j or with_k
        if omega is not None:
            assert omega >= 0.0, "short ranged J/K not supported"
        dev = _lib.require_gpu()
        _lib.ensure_rys()
        L = _lib.lib()
        stream = _lib.stream_ptr()
        om = float(omega) if omega else 0.0
        lr = om > 0.0
        dm_in = dm
        dm_t = torch.as_tensor(np.asarray(dm) if not torch.is_tensor(dm) else dm, dtype=torch.float64, device=dev)
        out_shape = tuple(dm_t.shape)
        vk = 0
        if with_k:
            vk = tile_jk(mol_ref, dm_t, hermi, vhfopt, False, True, omega, verbose)[1]
        vj = 0
        if with_j:
            if om not in state["pairs"]:
                state["pairs"][om] = _ClassPairs(layout, om)
            cp = state["pairs"][om]
            classes = sorted(cp.cls)
            on_pairs = {(ab, cd) for ab in classes for cd in classes if ab >= cd and supported(ab, cd, lr)}
            # J only sees the symmetric part of D (reference jk.py:179-191 stacks [D, D^T]; here D_s = (D + D^T) / 2)
            dms = layout.dm_from_mol(dm_t.reshape(-1, layout.nao_mol, layout.nao_mol))
            dms = (0.5 * (dms + dms.transpose(1, 2))).contiguous()
            raw = torch.zeros_like(dms)
            counter = torch.zeros(1, dtype=torch.int64, device=dev)
            basis = layout.basis_data_fp64["packed"]
            n_launch = 0
            for idm in range(dms.shape[0]):
                D = dms[idm]
                log_max_dm = max(float(torch.log(D.abs().max() + 1e-300).item()), -36.8)
                ket = {}
                for cd in classes:
                    c = cp.cls[cd]
                    nt = L.jqc_pair_ntrip(cd[0], cd[1])
                    E = torch.empty(c["n"] * nt, dtype=torch.float64, device=dev)
                    ld = torch.empty(c["n"], dtype=torch.float32, device=dev)
                    _lib.check(L.jqc_pair_ket_density(basis.data_ptr(), D.data_ptr(), nao, c["sh"].data_ptr(), c["n"], cd[0],
                                                      cd[1], E.data_ptr(), ld.data_ptr(), stream))
                    ket[cd] = (E, ld)
                pair_cut = math.log(PAIR_CUTOFF) - log_max_dm
                for ab in classes:
                    a = cp.cls[ab]
                    for cd in classes:
                        if (max(ab, cd), min(ab, cd)) not in on_pairs:
                            continue
                        c = cp.cls[cd]
                        h = pair_kernel(*ab, *cd, lr=lr)
                        E, ld = ket[cd]
                        qk_best = float(max(c["q_host"][s0] for s0, _ in c["seg_host"]))
                        for (s0, sn), (npi, npj) in zip(a["seg_host"], a["nprim"]):
                            # bra pairs that can pass with the best ket pair at all (the list is sorted by bound)
                            qa = a["q_host"][s0:s0 + sn]
                            nb = int(np.searchsorted(-qa, -(log_cutoff - log_max_dm - qk_best), side="left"))
                            nb = min(nb, int(np.searchsorted(-qa, -pair_cut, side="left")))
                            if nb <= 0:
                                continue
                            nblk = (nb + PAIR_WIDE_VJ - 1) // PAIR_WIDE_VJ
                            nsplit = int(max(1, min(64, TARGET_WGS // nblk, c["n"] // 8 + 1)))
                            _lib.check(L.jqc_pair_vj_launch(
                                h, nao, basis.data_ptr(), E.data_ptr(), raw[idm].data_ptr(), om,
                                a["sh"].data_ptr() + int(s0) * 4, nb, a["q"].data_ptr() + int(s0) * 4,
                                a["tab"].data_ptr() + int(s0) * 27 * 8, c["sh"].data_ptr(), c["q"].data_ptr(),
                                ld.data_ptr(), c["tab"].data_ptr(), c["seg"].data_ptr(), int(c["seg_host"].shape[0]),
                                log_cutoff, log_max_dm, npi, npj, nsplit, counter.data_ptr(), stream))
                            n_launch += 1
            vjr = raw * 2.0                                         # epilogue of the raw convention (reference jk.py:350-370)
            vjr = vjr + vjr.transpose(1, 2)
            vj = layout.dm_to_mol(vjr).reshape(out_shape)
            # canonical classes without a pair kernel: tiled J kernels on exactly those classes
            angs = sorted(set(int(x) for x in layout.angs))
            all_canon = [(a, b, c, d) for a in angs for b in angs for c in angs for d in angs if a >= b and a >= c and c >= d]
            on_tiles = [q for q in all_canon if ((q[0], q[1]), (q[2], q[3])) not in on_pairs
                        and ((q[2], q[3]), (q[0], q[1])) not in on_pairs]
            if on_tiles:
                keep = set(on_tiles)
                # (the tiled path sees the full D; for hermi = 0 it stacks [D, D^T] itself)
                vj = vj + tile_jk(mol_ref, dm_t, hermi, vhfopt, True, False, omega, verbose, _classes=lambda q: tuple(q) in keep)[0]
            state["stats"].update(pair_launches=n_launch, pair_counter=counter, pair_classes=len(on_pairs), tile_classes=len(on_tiles))
            # first J of this closure (per range-separation mode): cross-check the pair kernels against the tiled J kernels
            # on this call's own density, as jk.first_use_check does for tile builds outside the verified manifest
            ckey = (_lib.lib().jqc_pair_source_tag(), _lib.lib().jqc_source_tag(), bool(lr), frozenset(on_pairs))
            if ckey not in _PAIR_CHECKED and os.environ.get("JQC_TRUST_KERNELS") != "1":
                ref = tile_jk(mol_ref, dm_t, hermi, vhfopt, True, False, omega, verbose)[0]
                scale = float(ref.abs().max().item())
                err = float((vj - ref).abs().max().item())
                if not err <= 1e-9 * max(scale, 1e-300):
                    raise RuntimeError(f"pair-based J disagrees with the tiled J kernels on its first use (max |diff| {err:.3e}, "
                                       f"largest element {scale:.3e}): the pair kernels are rejected")
            _PAIR_CHECKED.add(ckey)       # per process, like jk._FIRST_USE_OK: a scan / optimisation re-applies per geometry
        if isinstance(dm_in, np.ndarray) and getattr(get_jk, "return_numpy", False):
            vj = vj.cpu().numpy() if with_j else 0
            vk = vk.cpu().numpy() if with_k else 0
        return vj, vk

    get_jk.stats = state["stats"]
    get_jk.layout = layout
    get_jk.return_numpy = False
    return get_jk
