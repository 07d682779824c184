"""``joltqc_amd.pyscf.apply(mf)`` -- the drop-in boundary.

Same contract as ``jqc.pyscf.apply`` (/root/reference/jqc/pyscf/__init__.py:121-254): patches, in
place, ``get_jk / get_j / get_k / get_veff`` (and for RKS ``_numint.get_rho / nr_rks / nr_nlc_vxc``,
``grids.build``) of an RHF/RKS-like mean-field object with MI355X kernels, wraps ``reset`` and
``as_scanner`` so geometry changes re-apply, and returns the object.

Difference forced by the platform: GPU4PySCF does not exist on ROCm, so ``obj.to_gpu()`` is only
attempted when the object offers it AND gpu4pyscf is importable; otherwise the CPU PySCF object
(or any object with ``.mol`` and ``istype``) is patched directly, NumPy in / NumPy out at the
boundary with device buffers kept inside the closures ("Callers").
"""
from functools import wraps
from types import MethodType
from typing import Any, Dict, Optional

from . import parallel  # noqa: F401  (multi-GPU driver/worker protocol; imports nothing heavy)

__all__ = ["apply", "reset", "get_default_config", "parallel"]


def get_default_config() -> Dict[str, Any]:
    """Default cutoffs (reference __init__.py:100-118)."""
    return {
        # cutoffs None -> obj.direct_scf_tol.  pair_j: J-only calls (``get_j``: every SCF iteration of a pure functional) go
        # through the pair-based Coulomb kernels (jk_pair.py; 2.3x faster than the tiled J kernels at 112 atoms / def2-TZVPP)
        # when the build is single-GPU and all-FP64; K and J+K always use the tiled kernels
        "jk": {"cutoff_fp32": None, "cutoff_fp64": None, "pair_j": True},
        "dft": {"cutoff_fp32": 1e-13, "cutoff_fp64": 1e-6},
        # True: share the work over the ranks of the initialised torch.distributed group (joltqc_amd/pyscf/parallel.py);
        # not in the reference, which drives one device
        "parallel": False,
        # True: mf.get_hcore / mf.get_ovlp from the device kernels (joltqc_amd/pyscf/int1e.py) instead of the object's own
        # (PySCF: libcint on the CPU); a molecule with ECPs gets the device ECP matrix added to h_core (backend/ecp.py)
        "int1e": False,
        # True: RHF objects with ``nuc_grad_method`` (PySCF) get gradient objects whose ``grad_elec`` takes the two-electron term
        # from the device kernels (joltqc_amd/pyscf/grad.py).  ``obj._jqc_jk_energy_per_atom`` is installed
        # in any case.  Not in the reference, which leaves gradients to GPU4PySCF
        "grad": False,
    }


def create_reset_function(original_reset, config):
    @wraps(original_reset)
    def reset(self, mol=None):
        mf = original_reset(self, mol)
        return apply(mf, config)
    return reset


def create_scanner_wrapper(original_as_scanner, config):
    @wraps(original_as_scanner)
    def as_scanner(self, **kwargs):
        scanner = original_as_scanner(self, **kwargs)
        scanner._joltqc_applied = True
        if hasattr(scanner, "reset"):
            original_scanner_reset = scanner.reset.__func__
            scanner.reset = MethodType(create_reset_function(original_scanner_reset, config), scanner)
        return scanner
    return as_scanner


def reset(obj, mol=None):
    """Module-level convenience named in the reference's ``__all__`` (never defined there)."""
    return obj.reset(mol)


def _is(obj, name):
    return hasattr(obj, "istype") and obj.istype(name)


def apply(obj, config: Optional[Dict[str, Any]] = None):
    """Patch ``obj`` in place with the MI355X J/K (and DFT grid) kernels and return it."""
    is_device_obj = "gpu4pyscf" in obj.__class__.__module__
    if not is_device_obj and hasattr(obj, "to_gpu"):
        try:
            import gpu4pyscf  # noqa: F401  (absent on ROCm images)
            obj = obj.to_gpu()
            is_device_obj = True
        except ImportError:
            pass

    if config is None:
        config = get_default_config()
    jk_cutoff_fp32 = config.get("jk", {}).get("cutoff_fp32")
    jk_cutoff_fp64 = config.get("jk", {}).get("cutoff_fp64")
    dft_cutoff_fp32 = config.get("dft", {}).get("cutoff_fp32")
    dft_cutoff_fp64 = config.get("dft", {}).get("cutoff_fp64")
    if jk_cutoff_fp32 is None:
        jk_cutoff_fp32 = getattr(obj, "direct_scf_tol", 1e-12)
    if jk_cutoff_fp64 is None:
        jk_cutoff_fp64 = getattr(obj, "direct_scf_tol", 1e-12)
    if dft_cutoff_fp32 is None:
        dft_cutoff_fp32 = 1e-13
    if dft_cutoff_fp64 is None:
        dft_cutoff_fp64 = 1e-6

    if hasattr(obj, "istype") and not obj.istype("RHF"):
        return obj

    from . import jk as _jk
    from .basis import BasisLayout
    from ..constants import tile_width

    # the reference builds two layouts (alignment 1 for DFT, TILE=4 for JK, __init__.py:188-189); here the
    # JK layout pads every (l, nprim) group to the tile width of the tiled kernels (4/4/4/2/1 for s..g)
    basis_layout_jk = BasisLayout.from_mol(obj.mol, alignment=tile_width)
    obj._jqc_basis_layout = basis_layout_jk
    numpy_boundary = not is_device_obj
    obj._jqc_numpy_boundary = numpy_boundary
    shard = None
    if config.get("parallel"):
        from . import parallel as _par
        rank, nranks = _par.world()
        if nranks > 1:
            shard = (rank, nranks)
            obj._jqc_parallel = {}


    if hasattr(obj, "istype") and not obj.istype("DFRHF") and not obj.istype("DFRKS"):
        if hasattr(obj, "get_jk"):
            get_jk = _jk.generate_jk_kernel(basis_layout_jk, cutoff_fp32=jk_cutoff_fp32, cutoff_fp64=jk_cutoff_fp64,
                                            shard=shard)
            if shard is not None:
                # every rank evaluates its share of the quartets; rank 0 (the one that runs the SCF) announces each
                # call and broadcasts D, the others mirror it from parallel.serve(obj)
                obj._jqc_parallel[_par.OP_JK] = get_jk
                if shard[0] == 0:
                    get_jk = _par.drive_jk(get_jk)
            get_jk.return_numpy = numpy_boundary
            obj.get_jk = get_jk
            if hasattr(obj, "get_j"):
                obj.get_j = lambda *a, **k: get_jk(*a, with_j=True, with_k=False, **k)[0]
                if config.get("jk", {}).get("pair_j", True) and shard is None and jk_cutoff_fp32 == jk_cutoff_fp64:
                    from . import jk_pair as _jk_pair
                    pair_jk = _jk_pair.generate_jk_kernel(basis_layout_jk, cutoff_fp32=jk_cutoff_fp32,
                                                          cutoff_fp64=jk_cutoff_fp64, tile_jk=get_jk)
                    pair_jk.return_numpy = numpy_boundary
                    obj._jqc_pair_jk = pair_jk
                    obj.get_j = lambda *a, **k: pair_jk(*a, with_j=True, with_k=False, **k)[0]
            if hasattr(obj, "get_k"):
                obj.get_k = lambda *a, **k: get_jk(*a, with_j=False, with_k=True, **k)[1]
        if _is(obj, "RHF") and not _is(obj, "RKS"):
            obj.get_veff = MethodType(_jk.generate_get_veff(), obj)
        # gradient of the two-electron energy at fixed density (the closure does no device work until it is called)
        from . import grad as _grad
        jk_grad = _grad.generate_jk_energy_per_atom(basis_layout_jk, cutoff=min(jk_cutoff_fp32, jk_cutoff_fp64), shard=shard)
        if shard is not None:
            obj._jqc_parallel[_par.OP_GRADJK] = jk_grad            # the quartet queue is dealt to the ranks like the J/K build
            if shard[0] == 0:
                jk_grad = _par.drive_grad_jk(jk_grad)
        obj._jqc_jk_energy_per_atom = jk_grad
        if config.get("grad") and _is(obj, "RHF") and not _is(obj, "RKS") and hasattr(obj, "nuc_grad_method") \
                and not hasattr(obj, "_jqc_original_nuc_grad_method"):
            original_ngm = obj.nuc_grad_method
            obj._jqc_original_nuc_grad_method = original_ngm
            obj.nuc_grad_method = lambda *a, **k: _grad.patch_gradients(original_ngm(*a, **k))

    if config.get("int1e"):
        from . import int1e as _int1e
        lay1 = BasisLayout.from_mol(obj.mol, alignment=1)
        # (bound like the originals: mf.get_hcore(mol=None), mf.get_ovlp(mol=None))
        hc, ov = _int1e.generate_get_hcore(lay1, numpy_boundary), _int1e.generate_get_ovlp(lay1, numpy_boundary)
        obj.get_hcore = lambda mol=None: hc(None)
        obj.get_ovlp = lambda mol=None: ov(None)

    if _is(obj, "RKS"):                      # after the J/K closures: the RKS get_veff calls them
        from . import rks as _rks
        _rks.patch(obj, BasisLayout.from_mol(obj.mol, alignment=1), dft_cutoff_fp32, dft_cutoff_fp64, numpy_boundary, shard)

    obj._joltqc_applied = True
    if not hasattr(obj, "_jqc_original_reset") and hasattr(obj, "reset"):
        original_reset = obj.reset.__func__
        obj._jqc_original_reset = original_reset
        obj.reset = MethodType(create_reset_function(original_reset, config), obj)
    if hasattr(obj, "as_scanner") and hasattr(obj.as_scanner, "__func__"):
        obj.as_scanner = MethodType(create_scanner_wrapper(obj.as_scanner.__func__, config), obj)
    return obj
