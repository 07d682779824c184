"""``jqc.pyscf`` -> ``joltqc_amd.pyscf`` (same ``__all__`` as /root/reference/jqc/pyscf/__init__.py:20)."""
from joltqc_amd.pyscf import apply, get_default_config, parallel, reset  # noqa: F401

__all__ = ["apply", "reset", "get_default_config"]
