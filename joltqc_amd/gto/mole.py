"""A small stand-alone molecule/basis container with PySCF's (libcint's) memory conventions.

The drop-in boundary ``joltqc_amd.pyscf.apply(mf)`` consumes ``mf.mol`` through the same
attributes the reference reads from a ``pyscf.gto.Mole`` (``_atm``, ``_bas``, ``_env``, ``cart``,
``nao``, ``nbas``, ``natm``; see ``/root/reference/jqc/pyscf/basis.py:501-603, :709-834``).
PySCF is not installed in the build or GPU images, so this class provides those attributes with
identical slot layout and identical coefficient normalisation; a real PySCF ``Mole`` can be passed
to ``apply`` unchanged wherever PySCF exists.
"""
import math
import re

import numpy as np

from . import basis_data

# libcint slots
CHARGE_OF, PTR_COORD, NUC_MOD_OF, PTR_ZETA, ATM_SLOTS = 0, 1, 2, 3, 6
ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF, KAPPA_OF, PTR_EXP, PTR_COEFF, BAS_SLOTS = 0, 1, 2, 3, 4, 5, 6, 8
PTR_ENV_START = 20
PTR_RANGE_OMEGA = 8
BOHR = 0.52917721092  # Angstrom, the value PySCF uses (CODATA 2010)

ELEMENTS = ["X", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si",
            "P", "S", "Cl", "Ar", "K", "Ca", "Sc", "Ti", "V", "Cr", "Mn", "Fe", "Co", "Ni", "Cu",
            "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr"]
_CHARGE = {s.upper(): z for z, s in enumerate(ELEMENTS)}


def gaussian_int(n, alpha):
    """int_0^inf x^n exp(-alpha x^2) dx"""
    n1 = (n + 1) * 0.5
    return math.gamma(n1) / (2.0 * alpha ** n1)


def gto_norm(l, expnt):
    """Radial normalisation 1/sqrt(int r^(2l+2) exp(-2 a r^2) dr)."""
    return 1.0 / math.sqrt(gaussian_int(l * 2 + 2, 2 * expnt))


def _normalize_contracted(l, es, cs):
    """PySCF's contracted-AO normalisation (radial part normalised to one)."""
    es = np.asarray(es, dtype=float)
    ee = es[:, None] + es[None, :]
    n1 = (l * 2 + 2 + 1) * 0.5
    ee = math.gamma(n1) / (2.0 * ee ** n1)
    s1 = 1.0 / np.sqrt(np.einsum("pi,pq,qi->i", cs, ee, cs))
    return cs * s1


def parse_atom(atom, unit="angstrom"):
    if isinstance(atom, str):
        rows = [r for r in re.split(r"[;\n]", atom) if r.strip()]
        out = []
        for r in rows:
            t = r.replace(",", " ").split()
            out.append((t[0], tuple(float(x) for x in t[1:4])))
    else:
        out = [(a[0], tuple(float(x) for x in (a[1] if len(a) == 2 else a[1:4]))) for a in atom]
    scale = 1.0 if unit.lower().startswith(("b", "au")) else 1.0 / BOHR
    return [(re.sub(r"[^A-Za-z]", "", s).capitalize(), tuple(c * scale for c in xyz)) for s, xyz in out]


def read_xyz(path):
    with open(path) as f:
        lines = f.read().strip().splitlines()
    n = int(lines[0].split()[0])
    return "\n".join(lines[2:2 + n])


class Mole:
    """Minimal ``pyscf.gto.Mole`` look-alike (attributes only; integrals live elsewhere)."""

    def __init__(self, atom=None, basis="sto-3g", unit="angstrom", cart=False, charge=0, spin=0,
                 verbose=0, ecp=None, **_ignored):
        self.atom = atom
        self.basis = basis
        self.ecp = ecp                    # {symbol: NWChem-format text}, as pyscf.gto.M(ecp=...) takes it (gto/ecp.py)
        self.unit = unit
        self.cart = bool(cart)
        self.charge = charge
        self.spin = spin
        self.verbose = verbose
        self.omega = 0.0
        self._built = False
        if atom is not None:
            self.build()

    # ------------------------------------------------------------------ build
    def build(self):
        atoms = parse_atom(self.atom, self.unit)
        self._atom = atoms
        env = [0.0] * PTR_ENV_START
        atm = []
        for sym, xyz in atoms:
            atm.append([_CHARGE[sym.upper()], len(env), 1, len(env) + 3, 0, 0])
            env.extend(xyz)
            env.append(0.0)
        bas = []
        cache = {}
        for ia, (sym, _) in enumerate(atoms):
            if isinstance(self.basis, str):
                shells = basis_data.load(self.basis, sym)
            else:
                b = self.basis.get(sym, self.basis.get(sym.upper()))
                shells = basis_data.load(b, sym) if isinstance(b, str) else b
            if sym not in cache:
                entries = []
                for sh in shells:
                    l = int(sh[0])
                    rows = np.array(sh[1:], dtype=float)
                    es = rows[:, 0]
                    cs = rows[:, 1:]
                    cs = cs * np.array([gto_norm(l, e) for e in es])[:, None]
                    cs = _normalize_contracted(l, es, cs)
                    ptr_e = len(env)
                    env.extend(es.tolist())
                    ptr_c = len(env)
                    env.extend(cs.T.reshape(-1).tolist())
                    entries.append((l, len(es), cs.shape[1], ptr_e, ptr_c))
                cache[sym] = entries
            for l, nprim, nctr, ptr_e, ptr_c in cache[sym]:
                bas.append([ia, l, nprim, nctr, 0, ptr_e, ptr_c, 0])
        self._atm = np.array(atm, dtype=np.int32).reshape(-1, ATM_SLOTS)
        self._bas = np.array(bas, dtype=np.int32).reshape(-1, BAS_SLOTS)
        self._env = np.array(env, dtype=np.float64)
        self._ecpbas = np.zeros((0, 8), dtype=np.int32)
        if self.ecp:
            from . import ecp as _ecp
            _ecp.attach(self, self.ecp)
        self._built = True
        return self

    # ------------------------------------------------------------- properties
    @property
    def natm(self):
        return int(self._atm.shape[0])

    @property
    def nbas(self):
        return int(self._bas.shape[0])

    def _dims(self):
        l = self._bas[:, ANG_OF]
        d = (l + 1) * (l + 2) // 2 if self.cart else 2 * l + 1
        return d * self._bas[:, NCTR_OF]

    @property
    def nao(self):
        return int(self._dims().sum())

    def nao_nr(self):
        return self.nao

    @property
    def ao_loc(self):
        return np.concatenate([[0], np.cumsum(self._dims())]).astype(np.int32)

    def ao_loc_nr(self):
        return self.ao_loc

    def atom_coords(self):
        p = self._atm[:, PTR_COORD]
        return np.stack([self._env[q:q + 3] for q in p])

    def atom_charges(self):
        return self._atm[:, CHARGE_OF].copy()

    def atom_symbol(self, i):
        return self._atom[i][0]

    def has_ecp(self):
        """pyscf.gto.Mole.has_ecp: does any atom carry an effective core potential?"""
        return len(getattr(self, "_ecpbas", ())) > 0

    @property
    def nelectron(self):
        return int(self.atom_charges().sum()) - self.charge

    def energy_nuc(self):
        z = self.atom_charges().astype(float)
        r = self.atom_coords()
        d = np.linalg.norm(r[:, None] - r[None], axis=-1)
        iu = np.triu_indices(self.natm, 1)
        return float((z[:, None] * z[None])[iu].__truediv__(d[iu]).sum())

    def bas_angular(self, i):
        return int(self._bas[i, ANG_OF])

    def bas_nprim(self, i):
        return int(self._bas[i, NPRIM_OF])

    def bas_nctr(self, i):
        return int(self._bas[i, NCTR_OF])

    def bas_exp(self, i):
        p = self._bas[i, PTR_EXP]
        return self._env[p:p + self._bas[i, NPRIM_OF]].copy()

    def bas_ctr_coeff(self, i):
        """libcint-normalised contraction coefficients, shape (nprim, nctr)."""
        p = self._bas[i, PTR_COEFF]
        n, c = self._bas[i, NPRIM_OF], self._bas[i, NCTR_OF]
        return self._env[p:p + n * c].reshape(c, n).T.copy()

    def bas_coord(self, i):
        p = self._atm[self._bas[i, ATOM_OF], PTR_COORD]
        return self._env[p:p + 3].copy()

    def set_geom_(self, coords_bohr):
        coords_bohr = np.asarray(coords_bohr, dtype=float)
        for ia in range(self.natm):
            p = self._atm[ia, PTR_COORD]
            self._env[p:p + 3] = coords_bohr[ia]
        self._atom = [(s, tuple(c)) for (s, _), c in zip(self._atom, coords_bohr)]
        return self

    def copy(self):
        import copy
        m = copy.copy(self)
        m._atm = self._atm.copy()
        m._bas = self._bas.copy()
        m._env = self._env.copy()
        return m


def M(**kw):
    return Mole(**kw)
