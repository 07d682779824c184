"""Nuclear gradient of the two-electron (J/K) energy on the device -- SURVEY.md section 8(f) row 3, the step after the SCF path.

JoltQC has no gradient kernels: its ``apply`` only re-patches the mean-field object that a gradient scanner resets
(``/root/reference/jqc/pyscf/__init__.py:63-97``) and the derivative integrals come from GPU4PySCF's CUDA code
(``tests/test_geom_opt.py:250-354``), which does not exist on ROCm.  This module supplies the expensive part of an RHF / RKS /
UHF gradient in the form GPU4PySCF's gradient drivers consume it (``gpu4pyscf.grad.rhf._jk_energy_per_atom``):

    ejk[atom, x] = d/dR_atom,x [ 1/2 j_factor tr(D J[D]) - 1/4 k_factor n_dm sum_s tr(D^s K[D^s]) ]   at fixed densities

for one (closed shell, total density) or two (alpha, beta) density matrices, full-range or long-range (``omega``) Coulomb
operator.  ``rhf_grad_elec`` assembles the rest of an RHF gradient for objects that offer PySCF's derivative one-electron
integrals; the quartets are screened by the same queue kernel as the one-quartet-per-lane J/K path (``jqc_screen_jk_tasks``),
one launch of ``jk_grad_<class>`` (csrc/kernels/jk_grad.hip) per angular class, and shard over ranks like the J/K build
(one all-reduce of natm x 3 doubles).
"""
import math

import numpy as np

from ..backend import lib as _lib
from . import jk as _jk

__all__ = ["generate_jk_energy_per_atom", "rhf_grad_elec"]

NREP = 64             # replicas of the per-atom accumulator (same-address f64 atomics of different workgroups)


def generate_jk_energy_per_atom(basis_layout, cutoff=1e-13, shard=None):
    """``jk_energy_per_atom(mol, dm, j_factor=1.0, k_factor=1.0, omega=None, hermi=1) -> [natm, 3]`` for ``basis_layout`` (any
    alignment).  ``dm``: ``[nao, nao]`` (closed shell, total density) or ``[2, nao, nao]`` (alpha, beta) in the molecule's AO
    basis, symmetric; NumPy in -> NumPy out, device tensor in -> device tensor out."""
    import torch
    layout = basis_layout
    nao, nbas = layout.nao, layout.nbasis
    natm = int(layout.atom_of.max()) + 1 if layout._mol is None else int(layout._mol.natm)
    log_cutoff = float(np.float32(math.log(cutoff)))
    state = {"pairs": {}, "queue": None, "stats": {}, "atom": None}

    def jk_energy_per_atom(mol=None, dm=None, j_factor=1.0, k_factor=1.0, omega=None, hermi=1, verbose=None, _classes=None):
        # (_classes: predicate on the angular class (li, lj, lk, ll) -- per-class timing tools only)
        assert hermi == 1, "the gradient kernels take symmetric densities"
        if omega is not None:
            assert omega >= 0.0, "short ranged J/K not supported"
        dev = _lib.require_gpu()
        _lib.ensure_rys()
        L = _lib.lib()
        stream = _lib.stream_ptr()
        om = float(omega) if omega else 0.0
        lr = om > 0.0
        dm_t = torch.as_tensor(np.asarray(dm) if not torch.is_tensor(dm) else dm, dtype=torch.float64, device=dev)
        dms = layout.dm_from_mol(dm_t.reshape(-1, layout.nao_mol, layout.nao_mol))
        dms = (0.5 * (dms + dms.transpose(1, 2))).contiguous()
        n_dm = int(dms.shape[0])
        assert n_dm in (1, 2), "one (closed shell) or two (alpha, beta) density matrices"
        if state["atom"] is None:
            state["atom"] = torch.from_numpy(np.ascontiguousarray(layout.atom_of, dtype=np.int32)).to(dev)
        # density bounds for the screening predicate (as in get_jk: shell-block max of |D|)
        dm_cond = torch.empty((nbas, nbas), dtype=torch.float32, device=dev)
        _lib.check(L.jqc_shell_block_max(dms.data_ptr(), n_dm, nao, layout.device_ao_loc().data_ptr(), nbas,
                                         dm_cond.data_ptr(), stream))
        log_dm_cond = torch.log(dm_cond.double() + 1e-300).float().contiguous()
        log_max_dm = max(float(log_dm_cond.max().item()), -36.8)
        # the energy is quadratic in D: the largest contribution of a quartet carries two density factors
        log_max_dm2 = log_max_dm + max(log_max_dm, 0.0)
        if om not in state["pairs"]:
            state["pairs"][om] = _jk._PairTables(layout, om)
        pt = state["pairs"][om]
        plans = _jk.build_screen_plan(layout, pt, log_cutoff, log_max_dm2, _jk.QUEUE_DEPTH, _classes, shard)
        qsize = max((p["total"] for p in plans), default=0)
        if plans and (state["queue"] is None or state["queue"].numel() < qsize * 4):
            state["queue"] = torch.empty(max(qsize, 1) * 4, dtype=torch.int16, device=dev)
        queue = state["queue"]
        grad = torch.zeros((NREP, natm, 3), dtype=torch.float64, device=dev)
        b64 = layout.basis_data_fp64["packed"]
        n_launch = 0
        counters_all = []
        for p in plans:
            ncls = len(p["classes"])
            tasks_d = torch.from_numpy(p["tasks"]).to(dev)
            region_d = torch.from_numpy(p["region"]).to(dev)
            counters = torch.zeros((ncls, 2), dtype=torch.int32, device=dev)
            counters_all.append(counters)
            _lib.check(L.jqc_screen_jk_tasks(tasks_d.data_ptr(), p["tasks"].shape[0], p["nblocks"], pt.sh.data_ptr(),
                                             pt.q.data_ptr(), log_dm_cond.data_ptr(), nbas, 1, 1, log_cutoff - max(log_max_dm, 0.0),
                                             log_cutoff - max(log_max_dm, 0.0), log_max_dm, queue.data_ptr(),
                                             region_d.data_ptr(), counters.data_ptr(), stream))
            for n, ang in enumerate(p["classes"]):
                beg, end = int(p["region"][n, 0]), int(p["region"][n, 1])
                h = _lib.check(L.jqc_gen_jk_grad_kernel(*[int(x) for x in ang], int(lr), 0))
                _lib.check(L.jqc_jk_grad_launch(h, nao, b64.data_ptr(), dms.data_ptr(), n_dm, grad.data_ptr(),
                                                state["atom"].data_ptr(), natm, NREP, float(j_factor), float(k_factor), om,
                                                queue.data_ptr() + beg * 8, counters.data_ptr() + (2 * n) * 4,
                                                end - beg, 1, stream))
                n_launch += 1
        out = grad.sum(dim=0)
        if shard is not None and shard[1] > 1:
            import torch.distributed as dist
            dist.all_reduce(out)
        state["stats"].update(launches=n_launch, counters=counters_all)
        if isinstance(dm, np.ndarray):
            return out.cpu().numpy()
        return out

    def quartet_count():
        return int(sum(int(c[:, 0].sum().item()) for c in state["stats"].get("counters", [])))

    jk_energy_per_atom.stats = state["stats"]
    jk_energy_per_atom.quartet_count = quartet_count
    jk_energy_per_atom.layout = layout
    return jk_energy_per_atom


def rhf_grad_elec(mf, jk_energy_per_atom, dm=None, hyb=1.0, mo_energy=None, mo_coeff=None, mo_occ=None, mol=None, atmlst=None):
    """Electronic RHF gradient [natm, 3] of a converged mean-field object whose molecule offers PySCF's derivative one-electron
    integrals (``mol.intor('int1e_ipovlp')``, ``int1e_ipkin``, ``int1e_ipnuc``, ``int1e_iprinv`` with ``with_rinv_at_nucleus``):
    the one-electron and overlap terms as in ``pyscf.grad.rhf.grad_elec``, the two-electron term from the device kernels.
    ``mo_energy / mo_coeff / mo_occ`` default to the object's own, ``mol`` to ``mf.mol``; ``atmlst`` selects rows."""
    mol = mf.mol if mol is None else mol
    mo_energy = mf.mo_energy if mo_energy is None else mo_energy
    mo_coeff = mf.mo_coeff if mo_coeff is None else mo_coeff
    mo_occ = mf.mo_occ if mo_occ is None else mo_occ
    mo_e, mo_c, occ = (np.asarray(x.cpu() if hasattr(x, "cpu") else x) for x in (mo_energy, mo_coeff, mo_occ))
    if dm is None:
        dm0 = (mo_c[:, occ > 0] * occ[occ > 0]) @ mo_c[:, occ > 0].T
    else:
        dm0 = np.asarray(dm.cpu() if hasattr(dm, "cpu") else dm)
    dme0 = (mo_c[:, occ > 0] * (mo_e[occ > 0] * occ[occ > 0])) @ mo_c[:, occ > 0].T
    s1 = -mol.intor("int1e_ipovlp", comp=3)
    h1 = -(mol.intor("int1e_ipkin", comp=3) + mol.intor("int1e_ipnuc", comp=3))
    aoslices = mol.aoslice_by_atom()
    de = np.zeros((mol.natm, 3))
    for ia in range(mol.natm):
        p0, p1 = aoslices[ia, 2], aoslices[ia, 3]
        with mol.with_rinv_at_nucleus(ia):
            vrinv = -mol.atom_charge(ia) * mol.intor("int1e_iprinv", comp=3)
        hc = vrinv.copy()
        hc[:, p0:p1] += h1[:, p0:p1]
        hc = hc + hc.transpose(0, 2, 1)
        de[ia] = np.einsum("xij,ij->x", hc, dm0) - 2.0 * np.einsum("xij,ij->x", s1[:, p0:p1], dme0[p0:p1])
    ejk = jk_energy_per_atom(mol, dm0, j_factor=1.0, k_factor=hyb)
    de = de + np.asarray(ejk.cpu() if hasattr(ejk, "cpu") else ejk)
    return de if atmlst is None else de[list(atmlst)]


def patch_gradients(g):
    """Route ``grad_elec`` of a PySCF ``Gradients`` object (``mf.nuc_grad_method()``) through the device kernels.

    The override lives on a SUBCLASS of the object's class, not in its ``__dict__``: PySCF's ``g.as_scanner()`` (geomeTRIC /
    berny optimisers) copies ``g.__dict__`` into an instance of a class derived from ``g.__class__`` and replaces ``g.base``
    by ``base.as_scanner()``, so an instance attribute bound to the old object would keep answering for the old geometry.
    ``grad_elec`` therefore reads everything from ``self``: ``self.base`` (the mean field the scanner has just converged,
    with the ``_jqc_jk_energy_per_atom`` closure its own ``apply`` / ``reset`` installed) and ``self.mol``.  Molecules
    with ECPs keep PySCF's own ``grad_elec`` (the ECP gradient terms are not built here)."""
    base_cls = g.__class__
    if getattr(base_cls, "_jqc_patched", False):
        return g

    class JQCGradients(base_cls):
        _jqc_patched = True

        def grad_elec(self, mo_energy=None, mo_coeff=None, mo_occ=None, atmlst=None):
            mf = self.base
            fn = getattr(mf, "_jqc_jk_energy_per_atom", None)
            mol = getattr(self, "mol", None) or mf.mol
            if fn is None or getattr(mol, "has_ecp", lambda: False)():
                return base_cls.grad_elec(self, mo_energy, mo_coeff, mo_occ, atmlst)
            return rhf_grad_elec(mf, fn, mo_energy=mo_energy, mo_coeff=mo_coeff, mo_occ=mo_occ, mol=mol, atmlst=atmlst)

    JQCGradients.__name__ = base_cls.__name__
    g.__class__ = JQCGradients
    return g
