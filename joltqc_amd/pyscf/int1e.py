"""One-electron integrals on the device (second half).

The reference leaves ``mf.get_hcore()`` / ``mf.get_ovlp()`` to PySCF (libcint on the CPU); here overlap, kinetic energy and
nuclear attraction come from ``jqc_int1e`` (joltqc_amd/csrc/jqc_hip.cpp: Obara-Saika + Rys quadrature with the ERI tables),
evaluated on the split shells of the ``BasisLayout`` and folded back to the molecule's AOs with the same dense transform
as J/K (``V_mol = T^T V_int T``), so contracted shells cut into <= 3-primitive pieces add up correctly.

    S, T, V = int1e(layout, mol)                         # device tensors [nao_mol, nao_mol]
    apply(mf, {**get_default_config(), "int1e": True})   # mf.get_hcore / mf.get_ovlp from the device (NumPy out on a CPU object;
                                                         #  a molecule with ECPs gets the device ECP matrix added to h_core)
"""
import numpy as np

from ..backend import lib as _lib

__all__ = ["int1e", "generate_get_hcore", "generate_get_ovlp"]


def int1e(layout, mol=None):
    """(S, T, V) in the molecule's AO basis (device tensors).  ``mol`` defaults to the molecule the layout was built from;
    only its nuclear coordinates (Bohr) and charges are read."""
    import torch
    dev = _lib.require_gpu()
    _lib.ensure_rys()
    mol = mol if mol is not None else layout._mol
    real = np.nonzero(~layout.pad_id)[0]
    ii, jj = np.meshgrid(real, real, indexing="ij")
    m = ii >= jj
    pairs = ((ii[m].astype(np.int64) << 16) | jj[m]).astype(np.uint32)
    pairs_d = torch.from_numpy(pairs.view(np.int32)).to(dev)
    atoms = np.concatenate([np.asarray(mol.atom_coords(), dtype=np.float64),
                            np.asarray(mol.atom_charges(), dtype=np.float64)[:, None]], axis=1)
    atoms_d = torch.from_numpy(np.ascontiguousarray(atoms)).to(dev)
    nao = layout.nao
    out = torch.zeros((3, nao, nao), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().jqc_int1e(layout.basis_data_fp64["packed"].data_ptr(), layout.device_ao_loc().data_ptr(),
                                    pairs_d.data_ptr(), int(pairs.size), atoms_d.data_ptr(), int(atoms.shape[0]), nao,
                                    out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _lib.stream_ptr()))
    S, T, V = (layout.dm_to_mol(out[k]) for k in range(3))
    return S, T, V


def generate_get_hcore(layout, numpy_out=True):
    """``mf.get_hcore(mol=None)``: T + V_nuc, plus the scalar ECP matrix of a molecule that carries effective core potentials --
    what ``pyscf.scf.hf.get_hcore`` adds through ``mol.intor("ECPscalar")``; here from ``backend.ecp.get_ecp`` (the nuclear
    charges in ``mol._atm`` are already lowered by the core electrons, as PySCF lowers them)."""
    def get_hcore(mol=None):
        m = mol if mol is not None else layout._mol
        # T, S and the ECP matrix are evaluated on the LAYOUT's shells and ECP centres: a different molecule (a scanner's
        # displaced geometry) needs its own layout -- `apply()` builds one per geometry -- and is refused here rather than mixed
        if m is not layout._mol and not (np.allclose(np.asarray(m.atom_coords()), np.asarray(layout._mol.atom_coords()), atol=1e-12)
                                         and np.array_equal(np.asarray(getattr(m, "_ecpbas", ())), np.asarray(getattr(layout._mol, "_ecpbas", ())))):
            raise ValueError("get_hcore(mol): the molecule differs from the one this layout was built from; call apply() / reset() "
                             "for the new geometry")
        _, T, V = int1e(layout, m)
        h = T + V
        if len(getattr(m, "_ecpbas", ())) > 0:
            from ..backend import ecp as _ecp
            h = h + _ecp.get_ecp(layout)
        return h.cpu().numpy() if numpy_out else h
    return get_hcore


def generate_get_ovlp(layout, numpy_out=True):
    def get_ovlp(mol=None):
        S = int1e(layout, mol)[0]
        return S.cpu().numpy() if numpy_out else S
    return get_ovlp
