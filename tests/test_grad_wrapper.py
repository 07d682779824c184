"""The ``nuc_grad_method`` wrapper of ``apply(mf, {"grad": True})`` (joltqc_amd/pyscf/grad.py:patch_gradients) against a
stand-in for ``pyscf.grad.rhf.Gradients`` that behaves like PySCF where it matters: ``as_scanner()`` copies ``__dict__`` into
an instance of a derived class and swaps ``base`` for a scanner of the mean field.  Checked on the CPU (no kernels run: the
numeric pieces are recorded, not evaluated): the scanner's gradient is computed from ITS base / mol (the new geometry), the
orbital arguments are honoured, ``atmlst`` selects rows, ECP molecules keep the original method."""
import copy

import numpy as np

from joltqc_amd.pyscf import grad as G


class Mol:
    def __init__(self, tag, ecp=False, natm=3):
        self.tag, self.natm, self._ecp = tag, natm, ecp

    def has_ecp(self):
        return self._ecp


class MF:
    def __init__(self, mol):
        self.mol = mol
        self.mo_energy, self.mo_coeff, self.mo_occ = "e_" + mol.tag, "c_" + mol.tag, "o_" + mol.tag
        self._jqc_jk_energy_per_atom = ("fn", mol.tag)

    def as_scanner(self):
        return copy.copy(self)


class Gradients:
    """PySCF's shape: base, mol, grad_elec, as_scanner (lib.GradScanner pattern)."""

    def __init__(self, mf):
        self.base, self.mol = mf, mf.mol

    def grad_elec(self, mo_energy=None, mo_coeff=None, mo_occ=None, atmlst=None):
        return "original"

    def as_scanner(self):
        cls = self.__class__

        class Scanner(cls):
            def __init__(self, g):
                self.__dict__.update(g.__dict__)
                self.base = g.base.as_scanner()

            def __call__(self, mol):
                self.mol = mol
                self.base = MF(mol)                 # "runs the SCF" on the new geometry
                return self.grad_elec()
        return Scanner(self)


def test_scanner_gradient_follows_the_new_geometry(monkeypatch):
    calls = []

    def fake(mf, fn, dm=None, hyb=1.0, mo_energy=None, mo_coeff=None, mo_occ=None, mol=None, atmlst=None):
        calls.append((mf.mol.tag, fn, mo_energy, mol.tag, atmlst))
        de = np.arange(9.0).reshape(3, 3)
        return de if atmlst is None else de[list(atmlst)]
    monkeypatch.setattr(G, "rhf_grad_elec", fake)
    g = G.patch_gradients(Gradients(MF(Mol("A"))))
    assert g.grad_elec().shape == (3, 3) and calls[-1][0] == "A" and calls[-1][1] == ("fn", "A")
    sc = g.as_scanner()
    sc(Mol("B"))
    assert calls[-1][0] == "B" and calls[-1][1] == ("fn", "B") and calls[-1][3] == "B"      # not the captured object "A"
    assert g.grad_elec(mo_energy="custom")[0, 0] == 0 and calls[-1][2] == "custom"
    assert g.grad_elec(atmlst=[2, 0]).tolist() == [[6.0, 7.0, 8.0], [0.0, 1.0, 2.0]]
    assert G.patch_gradients(g) is g and type(g).__mro__.count(type(g)) == 1                # patched once


def test_ecp_molecules_keep_the_original_gradient(monkeypatch):
    monkeypatch.setattr(G, "rhf_grad_elec", lambda *a, **k: (_ for _ in ()).throw(AssertionError("device path used")))
    g = G.patch_gradients(Gradients(MF(Mol("E", ecp=True))))
    assert g.grad_elec() == "original"
    mf = MF(Mol("N"))
    del mf._jqc_jk_energy_per_atom
    assert G.patch_gradients(Gradients(mf)).grad_elec() == "original"
