"""BasisLayout: the shell table every kernel consumes.

Mirrors the interface of the reference's ``jqc.pyscf.basis`` (``/root/reference/jqc/pyscf/basis.py``):
``BasisLayout.from_mol`` (:374), ``split_basis`` (:678), ``sort_group_basis`` (:483), packed rows
``[x, y, z, ao_loc | c0, e0, c1, e1, c2, e2 | -, -]`` with stride 12 (:280-371), ``q_matrix`` (:218),
``dm_from_mol`` / ``dm_to_mol`` (:419/:452), ``compute_q_matrix`` (:840).

MI355X-first differences (results are unchanged, see tests/test_basis_layout.py):
  * the molecule is read through ``_atm/_bas/_env`` only, so a PySCF ``Mole`` or the stand-alone
    ``joltqc_amd.gto.Mole`` both work; no PySCF import;
  * unused packed slots 10/11 carry ``nprim`` and ``l`` so one angular-class kernel serves every
    primitive pattern (the reference recompiles per pattern);
  * Cartesian<->spherical + split-shell scatter is ONE dense transformation matrix applied with two
    device GEMMs (``D_int = T D_mol T^T``, ``V_mol = T^T V_int T``) instead of per-l atomic scatter
    kernels (reference ``jqc/backend/cart2sph.py``);
  * Schwarz bounds are evaluated on the GPU by this build's own diagonal-ERI kernel
    (``jqc_schwarz``), not by libcvhf.
"""
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import os

import numpy as np

from ..constants import BASIS_STRIDE, LMAX, NPRIM_MAX, SLOT_ANG, SLOT_NPRIM
from ..gto.c2s import cart2sph_l, fac_sp, ncart

# libcint slots (identical in PySCF)
ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF, PTR_EXP, PTR_COEFF = 0, 1, 2, 3, 5, 6
PTR_COORD = 1

__all__ = ["BasisLayout", "split_basis", "sort_group_basis", "compute_q_matrix"]


@dataclass
class SplitShell:
    atom: int
    l: int
    exps: np.ndarray
    coefs: np.ndarray          # libcint-normalised, WITHOUT the s/p factor
    parent: int                # index of the decontracted parent function (one per contraction)
    coord: np.ndarray


def split_basis(mol):
    """Decontract (nctr > 1 -> separate shells) and split contractions with more than
    ``NPRIM_MAX`` primitives into pieces that share the parent's AO range
    (reference ``basis.py:678-837``).  Returns (list[SplitShell], parent_of_split)."""
    bas, env, atm = np.asarray(mol._bas), np.asarray(mol._env), np.asarray(mol._atm)
    shells = []
    parent = 0
    for ib in range(bas.shape[0]):
        ia, l, nprim, nctr = (int(bas[ib, s]) for s in (ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF))
        pe, pc = int(bas[ib, PTR_EXP]), int(bas[ib, PTR_COEFF])
        exps = env[pe:pe + nprim]
        coefs = env[pc:pc + nprim * nctr].reshape(nctr, nprim)
        pcoord = int(atm[ia, PTR_COORD])
        coord = env[pcoord:pcoord + 3].copy()
        for ic in range(nctr):
            for p0 in range(0, nprim, NPRIM_MAX):
                p1 = min(p0 + NPRIM_MAX, nprim)
                shells.append(SplitShell(ia, l, exps[p0:p1].copy(), coefs[ic, p0:p1].copy(), parent, coord))
            parent += 1
    return shells, np.array([s.parent for s in shells], dtype=np.int32)


# How the shells of an (l, nprim) group are ordered, i.e. which shells share a tile (a run of `alignment` shells).  A listed (bra
# tile pair, ket tile pair) keeps the more of its candidate quartets the more alike the Schwarz bounds of a tile pair's shell
# pairs are, i.e. the more alike the shells of a tile are in POSITION and in EXTENT.  Tiles are clusters in the four coordinates
# (x, y, z, EXP_WEIGHT * ln(most diffuse exponent)).  Measured on 112 atoms / def2-TZVPP (profiles/r02_tile_clustering_*), J+K step:
#   "morton"   shells along a Morton curve (until round 2)                                              6 218 ms
#   "cluster"  tile = next free shell on the curve + its nearest free neighbours (position only)        5 526 ms
#   "cluster2" the same with seeds taken from the outside in (free shell farthest from the centroid)   5 356 ms
#   "cluster2" + exponent coordinate, EXP_WEIGHT = 2 / 4 / 5 / 6 / 10 Bohr per e-fold   5 210 / 5 012 / 4 962 / 4 982 / 5 030 ms
#   "cluster3" cluster2 + pairwise-swap refinement of the tiles, EXP_WEIGHT = 5                        4 917 ms   <- default
#   "cluster_exp" hard classes of similar exponents first, clusters inside (larger tiles)               5 640-5 943 ms
# Results do not depend on the order (tests/test_basis_layout.py, every GPU parity test).
SPATIAL_MODE = os.environ.get("JQC_SPATIAL_SORT", "cluster3")
EXP_CLASS = float(os.environ.get("JQC_EXP_CLASS", "2.5"))
EXP_WEIGHT = float(os.environ.get("JQC_EXP_WEIGHT", "5"))


def _cluster_tiles(idxs, coords, width):
    free = np.ones(len(idxs), dtype=bool)
    out = []
    if SPATIAL_MODE.startswith(("cluster2", "cluster3")):                 # seeds from the outside in: the free shell farthest from the centroid of the rest
        while free.any():
            cand = np.nonzero(free)[0]
            c0 = coords[cand].mean(0)
            seed = int(cand[np.argmax(((coords[cand] - c0) ** 2).sum(1))])
            free[seed] = False
            cand = np.nonzero(free)[0]
            take = [seed]
            if cand.size:
                d = ((coords[cand] - coords[seed]) ** 2).sum(1)
                take += [int(c) for c in cand[np.argsort(d, kind="stable")[:width - 1]]]
            free[take] = False
            out += [idxs[t] for t in take]
        if SPATIAL_MODE.startswith("cluster3"):
            out = _refine_tiles(out, {i: c for i, c in zip(idxs, coords)}, width)
        return out
    for seed in range(len(idxs)):
        if not free[seed]:
            continue
        free[seed] = False
        cand = np.nonzero(free)[0]
        take = [seed]
        if cand.size:
            d = ((coords[cand] - coords[seed]) ** 2).sum(1)
            take += [int(c) for c in cand[np.argsort(d, kind="stable")[:width - 1]]]
        free[take] = False
        out += [idxs[t] for t in take]
    return out


def _refine_tiles(order, coord_of, width, sweeps=6, near=6):
    """Local search on a tiling (runs of ``width`` entries of ``order``): swap two shells of neighbouring tiles whenever that
    lowers the summed squared distance of the shells to their tile centroids.  With sum_i |p_i - c|^2 = sum |p_i|^2 - |S|^2 / w
    (S = sum of the tile's points) a swap a_i <-> b_j changes the total by (|S_A|^2 + |S_B|^2 - |S_A - d|^2 - |S_B + d|^2) / w,
    d = a_i - b_j: all w x w candidate swaps of a tile pair at once."""
    n = len(order) // width
    if n < 2:
        return order
    P = np.array([coord_of[i] for i in order[:n * width]]).reshape(n, width, -1)
    ids = np.array(order[:n * width]).reshape(n, width)
    rest = list(order[n * width:])
    for _ in range(sweeps):
        cen = P.mean(1)
        changed = False
        for a in range(n):
            for b in np.argsort(((cen - cen[a]) ** 2).sum(1))[1:near + 1]:
                sa, sb = P[a].sum(0), P[b].sum(0)
                d = P[a][:, None, :] - P[b][None, :, :]                       # [ia, ib, dim]
                gain = ((sa ** 2).sum() + (sb ** 2).sum() - ((sa - d) ** 2).sum(2) - ((sb + d) ** 2).sum(2)) / width
                k = int(np.argmin(gain))
                if gain.flat[k] < -1e-12:
                    ia, ib = divmod(k, width)
                    P[a, ia], P[b, ib] = P[b, ib].copy(), P[a, ia].copy()
                    ids[a, ia], ids[b, ib] = ids[b, ib], ids[a, ia]
                    changed = True
        if not changed:
            break
    return [int(i) for i in ids.reshape(-1)] + rest


def sort_group_basis(shells, alignment=1, spatial_sort=True):
    """Group split shells by (l, nprim), l ascending then nprim descending, pad every group to a
    multiple of ``alignment`` with zero-width duplicates of its first shell
    (reference ``basis.py:483-675``).  Within a group the reference keeps molecule order; this build
    arranges the shells so that every tile (run of ``alignment`` shells) is spatially compact
    (``SPATIAL_MODE`` above); results do not depend on the order."""
    groups: Dict[Tuple[int, int], list] = {}
    for idx, s in enumerate(shells):
        groups.setdefault((s.l, len(s.exps)), []).append(idx)
    keys = sorted(groups.keys(), key=lambda k: (k[0], -k[1]))
    order, pad, gkey, goff = [], [], [], [0]
    if spatial_sort and shells:
        allc = np.array([s.coord for s in shells])
        lo = allc.min(axis=0)
        span = max(float((allc.max(axis=0) - lo).max()), 1e-9)
    for k in keys:
        idxs = groups[k]
        if spatial_sort:
            def morton(i):
                q = np.minimum(((shells[i].coord - lo) / span * 1023).astype(np.int64), 1023)
                code = 0
                for b in range(10):
                    for d in range(3):
                        code |= ((int(q[d]) >> b) & 1) << (3 * b + d)
                return (code, i)
            idxs = sorted(idxs, key=morton)
        align = alignment(k[0]) if callable(alignment) else int(alignment)
        if spatial_sort and SPATIAL_MODE.startswith("cluster") and align > 1 and len(idxs) > align:
            if SPATIAL_MODE.endswith("_exp"):
                # shells of similar extent together: classes of the most diffuse exponent (factor EXP_CLASS apart), most diffuse
                # class first, spatial clusters inside each class
                cls = {}
                for i in idxs:
                    cls.setdefault(int(np.floor(np.log(float(np.min(shells[i].exps))) / np.log(EXP_CLASS))), []).append(i)
                idxs = []
                for c in sorted(cls):
                    sub = cls[c]
                    idxs += _cluster_tiles(sub, np.array([shells[i].coord for i in sub]), align) if len(sub) > align else sub
            else:
                xyz = np.array([shells[i].coord for i in idxs])
                if EXP_WEIGHT > 0:      # fourth coordinate: log of the most diffuse exponent (Bohr per e-fold), see the table above
                    xyz = np.hstack([xyz, EXP_WEIGHT * np.log([[float(np.min(shells[i].exps))] for i in idxs])])
                idxs = _cluster_tiles(idxs, xyz, align)
        npad = (-len(idxs)) % align
        order += idxs + [idxs[0]] * npad
        pad += [False] * len(idxs) + [True] * npad
        gkey.append(k)
        goff.append(len(order))
    return (np.array(order, dtype=np.int32), np.array(pad, dtype=bool), np.array(gkey, dtype=np.int32).reshape(-1, 2),
            np.array(goff, dtype=np.int32))


@dataclass
class BasisLayout:
    packed: np.ndarray            # float64 [nbas, 12]
    angs: np.ndarray              # int32 [nbas]
    nprims: np.ndarray            # int32 [nbas]
    to_split_map: np.ndarray      # int32 [nbas]  internal -> split index
    pad_id: np.ndarray            # bool  [nbas]
    group_key: np.ndarray         # int32 [ngroups, 2] = (l, nprim)
    group_offset: np.ndarray      # int32 [ngroups+1]
    ao_loc: np.ndarray            # int32 [nbas+1]  internal Cartesian AO offsets (pads have width 0)
    mol_ao_loc: np.ndarray        # int32 [nbas]    offset of the parent contraction in the molecule's AO order
    nao_mol: int
    cart: bool
    atom_of: np.ndarray           # int32 [nbas]
    _mol: Optional[object] = None
    _split_to_decontracted: Optional[np.ndarray] = None
    _cache: dict = field(default_factory=dict)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_mol(cls, mol, alignment=1, dtype=np.float64, spatial_sort: bool = True) -> "BasisLayout":
        shells, parent = split_basis(mol)
        assert all(s.l <= LMAX for s in shells), f"angular momentum above {LMAX} is not supported"
        order, pad, gkey, goff = sort_group_basis(shells, alignment, spatial_sort)
        nbas = len(order)
        cart = bool(mol.cart)
        # AO offsets of every decontracted parent function in the molecule's own AO order
        bas = np.asarray(mol._bas)
        dims = []
        for ib in range(bas.shape[0]):
            l, nctr = int(bas[ib, ANG_OF]), int(bas[ib, NCTR_OF])
            dims += [ncart(l) if cart else 2 * l + 1] * nctr
        parent_loc = np.concatenate([[0], np.cumsum(dims)]).astype(np.int32)
        packed = np.zeros((nbas, BASIS_STRIDE))
        angs = np.zeros(nbas, dtype=np.int32)
        nprims = np.zeros(nbas, dtype=np.int32)
        mol_ao_loc = np.zeros(nbas, dtype=np.int32)
        atom_of = np.zeros(nbas, dtype=np.int32)
        ao_loc = np.zeros(nbas + 1, dtype=np.int32)
        for n, (si, is_pad) in enumerate(zip(order, pad)):
            s = shells[si]
            np_ = len(s.exps)
            packed[n, :3] = s.coord
            packed[n, 3] = ao_loc[n]
            packed[n, 4:4 + 2 * np_:2] = s.coefs * fac_sp(s.l)   # reference basis.py:549-553
            packed[n, 5:5 + 2 * np_:2] = s.exps
            packed[n, 5 + 2 * np_:10:2] = 1.0                     # harmless exponent for unused slots (c = 0)
            packed[n, SLOT_NPRIM] = np_
            packed[n, SLOT_ANG] = s.l
            angs[n], nprims[n], atom_of[n] = s.l, np_, s.atom
            mol_ao_loc[n] = parent_loc[s.parent]
            ao_loc[n + 1] = ao_loc[n] + (0 if is_pad else ncart(s.l))
        return cls(packed=packed, angs=angs, nprims=nprims, to_split_map=order, pad_id=pad, group_key=gkey,
                   group_offset=goff, ao_loc=ao_loc, mol_ao_loc=mol_ao_loc, nao_mol=int(parent_loc[-1]), cart=cart,
                   atom_of=atom_of, _mol=mol, _split_to_decontracted=parent)

    # ------------------------------------------------------------------ compat accessors
    @property
    def nbasis(self) -> int:
        return int(self.packed.shape[0])

    @property
    def ngroups(self) -> int:
        return int(self.group_key.shape[0])

    @property
    def nao(self) -> int:
        return int(self.ao_loc[-1])

    @property
    def group_info(self):
        return self.group_key, self.group_offset

    @property
    def bas_info(self):
        return self.packed[:, 4:10], self.packed[:, :4], self.angs, self.nprims

    @property
    def ce(self):
        return self.packed[:, 4:10]

    @property
    def coords(self):
        return self.packed[:, :4]

    @property
    def angs_no_pad(self):
        return self.angs[~self.pad_id]

    @property
    def ao_loc_no_pad(self):
        return np.concatenate([self.ao_loc[:-1][~self.pad_id], self.ao_loc[-1:]]).astype(np.int32)

    # ------------------------------------------------------------------ transformation matrix
    def transform_matrix(self) -> np.ndarray:
        """T[nao_int, nao_mol]: internal Cartesian AO (row) expressed in the molecule's AOs."""
        if "T" not in self._cache:
            T = np.zeros((self.nao, self.nao_mol))
            for n in range(self.nbasis):
                if self.pad_id[n]:
                    continue
                l = int(self.angs[n])
                c = np.eye(ncart(l)) if self.cart else cart2sph_l(l)
                r0, c0 = self.ao_loc[n], self.mol_ao_loc[n]
                T[r0:r0 + c.shape[0], c0:c0 + c.shape[1]] = c
            self._cache["T"] = T
        return self._cache["T"]

    # ------------------------------------------------------------------ device side
    def _dev(self):
        from ..backend import lib as _lib
        return _lib.require_gpu()

    def device_T(self):
        if "T_dev" not in self._cache:
            import torch
            self._cache["T_dev"] = torch.from_numpy(self.transform_matrix()).to(self._dev())
        return self._cache["T_dev"]

    @property
    def basis_data_fp64(self) -> dict:
        if "b64" not in self._cache:
            import torch
            t = torch.from_numpy(np.ascontiguousarray(self.packed)).to(self._dev())
            self._cache["b64"] = {"packed": t, "coords": t[:, :4], "ce": t[:, 4:10]}
        return self._cache["b64"]

    @property
    def basis_data_fp32(self) -> dict:
        if "b32" not in self._cache:
            t = self.basis_data_fp64["packed"].float().contiguous()
            self._cache["b32"] = {"packed": t, "coords": t[:, :4], "ce": t[:, 4:10]}
        return self._cache["b32"]

    def device_ao_loc(self):
        if "ao_loc_dev" not in self._cache:
            import torch
            self._cache["ao_loc_dev"] = torch.from_numpy(self.ao_loc.astype(np.int32)).to(self._dev())
        return self._cache["ao_loc_dev"]

    def dm_from_mol(self, mat):
        """Molecule AO order (sph or cart) -> internal sorted/split Cartesian order.
        ``D_int = T D T^T`` (reference dm_from_mol + sph2cart, basis.py:419-450)."""
        import torch
        T = self.device_T()
        m = torch.as_tensor(mat, dtype=torch.float64, device=T.device)
        return T @ m @ T.T

    def dm_to_mol(self, mat):
        """Internal order -> molecule AO order, ACCUMULATING split shells into their parent block.
        ``V = T^T V_int T`` (reference dm_to_mol + cart2sph, basis.py:452-480)."""
        import torch
        T = self.device_T()
        m = torch.as_tensor(mat, dtype=torch.float64, device=T.device)
        return T.T @ m @ T

    # ------------------------------------------------------------------ Schwarz
    def q_matrix(self, omega=0.0):
        """float32 [nbas, nbas] device tensor of log(Q_ij + 1e-300), pads = -100
        (reference BasisLayout.q_matrix, basis.py:218-243)."""
        key = ("q", float(omega or 0.0))
        if key not in self._cache:
            self._cache[key] = compute_q_matrix(self, omega)
        return self._cache[key]


def compute_q_matrix(layout: BasisLayout, omega=0.0):
    """Schwarz bounds of all split-shell pairs on the GPU.
    Replaces ``CVHFnr_int2e_q_cond`` (reference basis.py:840-867): Q_ij = sqrt(max_ab |(ab|ab)|)."""
    import torch
    from ..backend import lib as _lib
    dev = _lib.require_gpu()
    _lib.ensure_rys()
    L = _lib.lib()
    nbas = layout.nbasis
    basis = layout.basis_data_fp64["packed"]
    q = torch.zeros((nbas, nbas), dtype=torch.float64, device=dev)
    goff, gkey = layout.group_offset, layout.group_key
    for gi in range(layout.ngroups):
        for gj in range(gi + 1):
            i0, i1, j0, j1 = int(goff[gi]), int(goff[gi + 1]), int(goff[gj]), int(goff[gj + 1])
            ii, jj = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), indexing="ij")
            m = ii >= jj
            ii, jj = ii[m], jj[m]
            if ii.size == 0:
                continue
            pairs = torch.from_numpy(((ii.astype(np.int64) << 16) | jj).astype(np.uint32).view(np.int32)).to(dev)
            out = torch.empty(ii.size, dtype=torch.float64, device=dev)
            _lib.check(L.jqc_schwarz(int(gkey[gi, 0]), int(gkey[gj, 0]), basis.data_ptr(), pairs.data_ptr(),
                                     int(ii.size), float(omega or 0.0), out.data_ptr(), _lib.stream_ptr()))
            it = torch.from_numpy(ii).to(dev)
            jt = torch.from_numpy(jj).to(dev)
            q[it, jt] = out
            q[jt, it] = out
    logq = torch.log(q + 1e-300).float()
    pad = torch.from_numpy(layout.pad_id).to(dev)
    logq[pad, :] = -100.0
    logq[:, pad] = -100.0
    return logq
