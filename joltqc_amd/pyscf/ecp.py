"""ECP patching interface (role of ``/root/reference/jqc/pyscf/ecp.py:27-118``): same four functions, same dictionary keys.

``apply_ecp(mol)`` returns ``{"get_ecp": closure, "ecp_kernel": launcher info, "precision": "fp64", "original_methods": {}}``
(empty dict for a molecule without ECP), ``patch_ecp_integrals(mol)`` installs ``mol.get_ecp`` in place and records
``mol._jqc_ecp_info``, ``restore_ecp_methods(mol)`` undoes it.  The integrals come from ``joltqc_amd.backend.ecp.get_ecp``
(device kernel ``ecp_scalar_kernel``); like the reference's module this one patches the scalar potential matrix only -- the derivative
integrals are called directly: ``joltqc_amd.backend.ecp.get_ecp_ip`` / ``get_ecp_ipip`` (reference ``jqc/backend/ecp.py:953-1340``).
"""
from typing import Any, Dict

from ..backend import ecp as _ecp


def apply_ecp(mol, precision: str = "fp64", cutoff_fp32: float = 1e-8, cutoff_fp64: float = 1e-12) -> Dict[str, Any]:
    if getattr(mol, "_ecpbas", None) is None or len(mol._ecpbas) == 0:
        return {}
    if precision != "fp64":
        raise ValueError("Only double precision ('fp64') is supported for ECP kernels")

    def jit_get_ecp():
        return _ecp.get_ecp(mol, precision)
    return {"get_ecp": jit_get_ecp, "ecp_kernel": {"kernel": "ecp_scalar_kernel", "nr": _ecp.NR_DEFAULT}, "precision": precision,
            "original_methods": {}}


def patch_ecp_integrals(mol, **kwargs) -> None:
    patches = apply_ecp(mol, **kwargs)
    if not patches:
        return
    original = {}
    for name, fn in patches.items():
        if name not in ("original_methods", "ecp_kernel", "precision"):
            if hasattr(mol, name):
                original[name] = getattr(mol, name)
            setattr(mol, name, fn)
    mol._jqc_ecp_info = {"precision": patches["precision"], "kernel_type": "ecp_scalar", "original_methods": original}


def restore_ecp_methods(mol) -> None:
    info = getattr(mol, "_jqc_ecp_info", None)
    if info is None:
        return
    for name in ("get_ecp",):
        if name in info["original_methods"]:
            setattr(mol, name, info["original_methods"][name])
        elif name in mol.__dict__:
            delattr(mol, name)
    delattr(mol, "_jqc_ecp_info")
