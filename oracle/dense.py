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
