"""Import-name alias: ``import jqc.pyscf`` resolves to ``joltqc_amd.pyscf`` so that a script written for the reference
(``import jqc.pyscf; mf = jqc.pyscf.apply(mf)``, /root/reference/jqc/pyscf/__init__.py:20,121) runs unchanged."""
