"""Oracle-side get_jk: the reference's full pipeline restated on the CPU (TEST INFRASTRUCTURE ONLY).

D_mol -> internal Cartesian order -> raw J/K over EVERY canonical quartet (no screening) with
oracle/jk_oracle.c -> epilogue of reference jqc/pyscf/jk.py:350-370 -> back to the molecule's AOs.
Only layout *data* (packed rows, ao_loc, transformation matrix) is taken from BasisLayout.
"""
import numpy as np

from . import jk as O
from . import md_eri


def canonical_quartets(layout):
    real = np.nonzero(~layout.pad_id)[0]
    nb = layout.nbasis
    out = []
    for i in real:
        for j in real[real <= i]:
            for k in real[real <= i]:
                for l in real[real <= k]:
                    if i * nb + j >= k * nb + l:
                        out.append((i, j, k, l))
    return np.array(out, dtype=np.uint16).reshape(-1, 4)


def get_jk(layout, dm, hermi=1, omega=None, with_j=True, with_k=True, quartets=None, cutoff=None, dense_loop=False):
    """``dense_loop``/``cutoff``: generate the canonical quartets inside the C loop (OpenMP over the host cores), optionally
    dropping those whose Schwarz estimate is below ``cutoff`` -- for cases with 1e7..1e8 quartets (benzene/def2-TZVPP)."""
    dm = np.asarray(dm, dtype=np.float64)
    shape = dm.shape
    T = layout.transform_matrix()
    dms = dm.reshape(-1, layout.nao_mol, layout.nao_mol)
    dms = np.einsum("pi,nij,qj->npq", T, dms, T)
    if hermi == 0:
        dms = np.concatenate([dms, dms.transpose(0, 2, 1)])
    if quartets is None and (dense_loop or cutoff is not None):
        vj, vk, _ = O.jk_raw_dense(layout.packed, dms, layout.pad_id, omega or 0.0, with_j, with_k, cutoff)
    else:
        q = canonical_quartets(layout) if quartets is None else quartets
        vj, vk = O.jk_raw(layout.packed, dms, q, omega or 0.0, with_j, with_k)
    n = dms.shape[0]
    res = []
    if with_j:
        vj = vj * 2.0 if hermi == 1 else vj[: n // 2] + vj[n // 2:].transpose(0, 2, 1)
        vj = vj + vj.transpose(0, 2, 1)
        res.append(np.einsum("pi,npq,qj->nij", T, vj, T).reshape(shape))
    else:
        res.append(0)
    if with_k:
        vk = vk + vk.transpose(0, 2, 1) if hermi == 1 else vk[: n // 2] + vk[n // 2:].transpose(0, 2, 1)
        res.append(np.einsum("pi,npq,qj->nij", T, vk, T).reshape(shape))
    else:
        res.append(0)
    return tuple(res)


def int1e_mol(layout, mol):
    """S, T, V in the molecule's AO basis from the independent MD engine."""
    S, Tk, V = md_eri.int1e(layout.packed, layout.ao_loc, mol.atom_coords(), mol.atom_charges())
    Tm = layout.transform_matrix()
    f = lambda M: Tm.T @ M @ Tm
    return f(S), f(Tk), f(V)


def _canon(i, j, k, l, nb):
    """Canonical form (i >= j, k >= l, (ij) >= (kl)) of the quartets (i j | k l), element-wise on integer arrays."""
    i, j = np.maximum(i, j), np.minimum(i, j)
    k, l = np.maximum(k, l), np.minimum(k, l)
    sw = i * nb + j < k * nb + l
    return np.stack([np.where(sw, k, i), np.where(sw, l, j), np.where(sw, i, k), np.where(sw, j, l)], 1)


def _unique_rows(q):
    """Distinct rows of an integer [n, 4] array of shell indices < 65536 (one 64-bit key per row: far faster than np.unique(axis=0))."""
    q = q.astype(np.uint64)
    key = (q[:, 0] << np.uint64(48)) | (q[:, 1] << np.uint64(32)) | (q[:, 2] << np.uint64(16)) | q[:, 3]
    key = np.unique(key)
    return np.stack([(key >> np.uint64(48)) & np.uint64(0xffff), (key >> np.uint64(32)) & np.uint64(0xffff),
                     (key >> np.uint64(16)) & np.uint64(0xffff), key & np.uint64(0xffff)], 1).astype(np.uint16)


def sampled_blocks(layout, dm_int, j_pairs=(), k_pairs=(), omega=None, nthreads=1):
    """Shell blocks of J and K in the INTERNAL (sorted, split, Cartesian) AO order, each from its own complete quartet list --
    O(N^2) quartets per block instead of the O(N^4) of a full build: the checker of bench.py's parity figure at sizes where
    ``get_jk`` above cannot finish (reference bar jqc/pyscf/tests/test_jk.py:83-84).

    ``J[i, j] = sum_kl (ij|kl) D_kl`` needs every canonical quartet that holds the pair (i j); ``K[i, k] = sum_jl (ij|kl) D_jl``
    every canonical quartet with i on one side and k on the other.  Each list goes through the same raw digestion + epilogue as
    a full build (jk_oracle.c, reference jk.py:350-370); only the requested block of the result is complete, and only it is
    returned.  ``dm_int``: symmetric density in the internal order; returns ({(i, j): block}, {(i, k): block})."""
    dm_int = np.ascontiguousarray(dm_int, dtype=np.float64)
    real = np.nonzero(~layout.pad_id)[0].astype(np.int64)
    nb = layout.nbasis
    loc = np.asarray(layout.ao_loc)

    def rng(s):
        return slice(int(loc[s]), int(loc[s + 1]))

    outj, outk = {}, {}
    # ONE digestion per kind: the union of the blocks' quartet lists, every canonical quartet once (a quartet of one list that also
    # touches another requested block belongs to that block's own list as well, so the union leaves every requested block complete)
    if len(j_pairs):
        kk, ll = np.meshgrid(real, real, indexing="ij")
        m = kk >= ll
        q = _unique_rows(np.concatenate([_canon(np.full(m.sum(), i), np.full(m.sum(), j), kk[m], ll[m], nb) for (i, j) in j_pairs]))
        vj, _ = O.jk_raw(layout.packed, dm_int, q.astype(np.uint16), omega or 0.0, True, False, nthreads=nthreads)
        v = vj[0] * 2.0
        v = v + v.T
        for (i, j) in j_pairs:
            outj[(i, j)] = v[rng(i), rng(j)].copy()
    if len(k_pairs):
        jj, ll = np.meshgrid(real, real, indexing="ij")
        q = _unique_rows(np.concatenate([_canon(np.full(jj.size, i), jj.ravel(), np.full(jj.size, k), ll.ravel(), nb) for (i, k) in k_pairs]))
        _, vk = O.jk_raw(layout.packed, dm_int, q.astype(np.uint16), omega or 0.0, False, True, nthreads=nthreads)
        v = vk[0] + vk[0].T
        for (i, k) in k_pairs:
            outk[(i, k)] = v[rng(i), rng(k)].copy()
    return outj, outk
