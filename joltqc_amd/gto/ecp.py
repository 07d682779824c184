"""Effective-core-potential input in PySCF's memory conventions (``mol._ecpbas`` + ``mol._env``).

The reference's ECP path reads a ``pyscf.gto.Mole``: ``mol._ecpbas[:, (ATOM_OF, ANG_OF, NPRIM_OF, RADI_POWER, SO_TYPE_OF,
PTR_EXP, PTR_COEFF, 0)]`` with ``ANG_OF = -1`` for the local ("ul") channel (``/root/reference/jqc/backend/ecp.py:1290-1343``
sorts those rows, ``:1371-1503`` consumes them).  PySCF is not in this image, so ``parse_ecp`` reads the same NWChem-style text
PySCF's ``gto.basis.parse_ecp`` reads (the reference's tests carry their ECPs inline in that format,
``jqc/pyscf/tests/test_ecp_small.py:47-68``) and ``attach`` lays the rows out exactly as ``Mole.build`` does, so that a real
PySCF molecule and the stand-in of ``gto/mole.py`` look the same to ``joltqc_amd.backend.ecp.get_ecp``.

A potential is   U(r) = sum_k c_k r^(n_k - 2) exp(-zeta_k r^2)   per channel; the semi-local channels l = 0, 1, ... act through
the projectors |lm><lm| around the ECP centre, the local channel on everything.
"""
import numpy as np

ATOM_OF, ANG_OF, NPRIM_OF, RADI_POWER, SO_TYPE_OF, PTR_EXP, PTR_COEFF, ECPBAS_SLOTS = 0, 1, 2, 3, 4, 5, 6, 8
_ANG = {"UL": -1, "S": 0, "P": 1, "D": 2, "F": 3, "G": 4, "H": 5}


def parse_ecp(text):
    """``"Na nelec 10\\nNa ul\\n2 1.0 0.5\\nNa S\\n2 13.65 732.27 ..."`` -> ``(nelec, [(l, [(power, zeta, coef), ...]), ...])``
    (l = -1: local channel).  Same grammar as ``pyscf.gto.basis.parse_ecp`` for scalar potentials."""
    nelec, chans, cur = 0, [], None
    for raw in text.strip().splitlines():
        t = raw.split("#")[0].split()
        if not t:
            continue
        if len(t) >= 3 and t[1].lower() == "nelec":
            nelec = int(t[2])
        elif len(t) == 2 and t[1].upper() in _ANG:
            cur = []
            chans.append((_ANG[t[1].upper()], cur))
        else:
            assert cur is not None and len(t) == 3, f"cannot read ECP line {raw!r}"
            cur.append((int(t[0]), float(t[1]), float(t[2])))
    return nelec, chans


def attach(mol, ecp):
    """Build ``mol._ecpbas`` (+ exponents / coefficients appended to ``mol._env``) from ``ecp = {symbol: text or parsed}`` and
    lower the nuclear charges by the core electrons, as ``pyscf.gto.Mole.build`` does."""
    env = list(mol._env)
    rows = []
    parsed = {k.capitalize(): (parse_ecp(v) if isinstance(v, str) else v) for k, v in ecp.items()}
    for ia in range(mol.natm):
        sym = mol.atom_symbol(ia)
        if sym not in parsed:
            continue
        nelec, chans = parsed[sym]
        mol._atm[ia, 0] -= nelec
        for l, terms in chans:
            for power in sorted({t[0] for t in terms}):
                sel = [t for t in terms if t[0] == power]
                pe = len(env)
                env.extend(t[1] for t in sel)
                pc = len(env)
                env.extend(t[2] for t in sel)
                rows.append([ia, l, len(sel), power, 0, pe, pc, 0])
    mol._env = np.asarray(env, dtype=np.float64)
    mol._ecpbas = np.asarray(rows, dtype=np.int32).reshape(-1, ECPBAS_SLOTS)
    mol._ecp = parsed
    return mol


def channels(mol):
    """``{atom index: [(l, power, zeta[], coef[]), ...]}`` from ``mol._ecpbas`` / ``mol._env`` (any object with PySCF's layout)."""
    out = {}
    eb = np.asarray(getattr(mol, "_ecpbas", np.zeros((0, ECPBAS_SLOTS), dtype=np.int32)))
    env = np.asarray(mol._env)
    for r in eb:
        # scalar potential only: spin-orbit rows (SO_TYPE_OF != 0, e.g. crenbl / crenbs potentials) are dropped, as the reference
        # does before it sorts the rows (/root/reference/jqc/backend/ecp.py:1313-1315) and as libcint's ECPscalar does
        if int(r[SO_TYPE_OF]) != 0:
            continue
        n = int(r[NPRIM_OF])
        out.setdefault(int(r[ATOM_OF]), []).append((int(r[ANG_OF]), int(r[RADI_POWER]),
                                                    env[r[PTR_EXP]:r[PTR_EXP] + n].copy(), env[r[PTR_COEFF]:r[PTR_COEFF] + n].copy()))
    return out
