"""TEST SCAFFOLDING: minimal RHF / RKS drivers with the attribute surface ``joltqc_amd.pyscf.apply`` patches.

Stand-in for ``pyscf.scf.RHF`` / ``pyscf.dft.RKS`` on images without PySCF (this one, and the GPU box).  It behaves
like a plain CPU PySCF object and is deliberately STRICT about it: everything it receives from the patched closures
goes through ``numpy.asarray`` exactly where PySCF does its NumPy arithmetic (``h1e + vhf``, ``energy_elec``, libxc's
``eval_xc_eff``), which raises ``TypeError`` on a CUDA tensor -- so a closure that leaks a device array across the
boundary fails here the way it would under PySCF.  It owns the SCF loop and DIIS, calls ``self.get_veff`` /
``self.get_jk`` where PySCF does, and computes no integrals itself: ``hcore`` and ``ovlp`` come from the caller (or
from ``int1e``, a callback mol -> (hcore, ovlp), which ``reset(mol)`` uses after a geometry change).
"""
import copy

import numpy as np


_BIG = 600       # matrices from this size on: the driver's OWN dense linear algebra (products, eigh) runs through torch on the GPU


def _on_device(mats):
    if max(max(m.shape) for m in mats) < _BIG:
        return None
    try:
        import torch
        return torch.device("cuda") if torch.cuda.is_available() else None
    except ImportError:
        return None


def _mm(*mats):
    """Product of dense NumPy matrices -> NumPy.  A 2 588-AO SCF spends more time in NumPy's products and eigh on the host than in the
    J/K builds under test; where a GPU is present the stand-in's own linear algebra (NOT the closures under test, which still get and
    return what PySCF would hand them) goes through torch in FP64.  Same arithmetic, NumPy in, NumPy out."""
    dev = _on_device(mats)
    if dev is None:
        out = mats[0]
        for m in mats[1:]:
            out = out @ m
        return out
    import torch
    out = torch.from_numpy(np.ascontiguousarray(mats[0])).to(dev)
    for m in mats[1:]:
        out = out @ torch.from_numpy(np.ascontiguousarray(m)).to(dev)
    return out.cpu().numpy()


def _eigh(A):
    dev = _on_device((A,))
    if dev is None:
        return np.linalg.eigh(A)
    import torch
    e, c = torch.linalg.eigh(torch.from_numpy(np.ascontiguousarray(A)).to(dev))
    return e.cpu().numpy(), c.cpu().numpy()


def _strict(x):
    """What PySCF's NumPy code does with a potential: ``numpy.asarray`` (TypeError on a CUDA tensor)."""
    if hasattr(x, "is_cuda") and x.is_cuda:
        raise TypeError("a CUDA tensor crossed the NumPy boundary of a CPU PySCF-like object")
    return np.asarray(x)


def atomic_density_guess(mol, cycles=40):
    """Initial density matrix for large molecules (the role of PySCF's ``init_guess='minao'`` / ``'atom'``): superposition of
    spherically averaged free-atom densities, block-diagonal in ``mol``'s AO order.  One small restricted SCF per element with
    fractional aufbau occupations (a degenerate level shares its electrons evenly, so the density stays spherical), run on the CPU
    oracle (``oracle/dense.py``; 5-36 AOs per atom).  The core-Hamiltonian guess of ``kernel`` does not converge for a
    100-atom molecule (``tools/scf_probe.py``: the energy swings by thousands of hartree for 50 cycles)."""
    from oracle import dense
    from joltqc_amd.gto import mole as gmole
    from joltqc_amd.pyscf.basis import BasisLayout
    atom_dm = {}
    for ia in range(mol.natm):
        sym = mol.atom_symbol(ia)
        if sym in atom_dm:
            continue
        atom = gmole.Mole(atom=[(sym, (0.0, 0.0, 0.0))], basis=mol.basis, cart=mol.cart, unit="B")
        lay = BasisLayout.from_mol(atom, alignment=1)
        S, T, V = dense.int1e_mol(lay, atom)
        h = T + V
        s, U = np.linalg.eigh(S)
        X = U[:, s > 1e-10] / np.sqrt(s[s > 1e-10])
        q = dense.canonical_quartets(lay)
        nelec = int(atom.atom_charges()[0])

        def density(F):
            e, c = np.linalg.eigh(X.T @ F @ X)
            c = X @ c
            occ = np.zeros(len(e))
            left, i = float(nelec), 0
            while left > 1e-12 and i < len(e):
                j = i
                while j + 1 < len(e) and e[j + 1] - e[i] < 1e-5:
                    j += 1
                g = j - i + 1
                fill = min(left, 2.0 * g)
                occ[i:j + 1] = fill / g
                left -= fill
                i = j + 1
            return (c * occ) @ c.T

        dm = density(h)
        for _ in range(cycles):
            vj, vk = dense.get_jk(lay, dm, 1, quartets=q)
            dm = 0.5 * dm + 0.5 * density(h + np.asarray(vj) - 0.5 * np.asarray(vk))
        atom_dm[sym] = dm
    nao = mol.nao
    dm = np.zeros((nao, nao))
    loc = np.asarray(mol.ao_loc_nr())
    atom_of = np.asarray(mol._bas[:, 0])
    for ia in range(mol.natm):
        sh = np.nonzero(atom_of == ia)[0]
        a0, a1 = int(loc[sh[0]]), int(loc[sh[-1] + 1])
        dm[a0:a1, a0:a1] = atom_dm[mol.atom_symbol(ia)]
    return dm


class RHF:
    def __init__(self, mol, hcore=None, ovlp=None, int1e=None):
        self.mol = mol
        self._int1e = int1e
        if hcore is None and int1e is not None:
            hcore, ovlp = int1e(mol)
        self._hcore = hcore
        self._ovlp = ovlp
        self.direct_scf = True
        self.direct_scf_tol = 1e-13
        self.conv_tol = 1e-9              # PySCF's defaults (scf/hf.py): |dE| < conv_tol and |orbital gradient| < conv_tol_grad
        self.conv_tol_grad = None         # None -> sqrt(conv_tol)
        self.max_cycle = 60
        self.diis_space = 8
        self.verbose = 0
        self.mo_coeff = None
        self.mo_energy = None
        self.e_tot = None
        self.converged = False
        self.cycles = 0

    # --- the PySCF surface apply() looks at -------------------------------------------------
    def istype(self, name):
        return name in ("RHF", "SCF")

    def get_hcore(self, mol=None):
        return self._hcore

    def get_ovlp(self, mol=None):
        return self._ovlp

    def get_jk(self, mol=None, dm=None, hermi=1, **kw):
        raise RuntimeError("no J/K engine attached: call joltqc_amd.pyscf.apply(mf) (or attach an oracle in tests)")

    def get_j(self, mol=None, dm=None, hermi=1, **kw):
        return self.get_jk(mol, dm, hermi, with_k=False, **kw)[0]

    def get_k(self, mol=None, dm=None, hermi=1, **kw):
        return self.get_jk(mol, dm, hermi, with_j=False, **kw)[1]

    def get_veff(self, mol=None, dm=None, dm_last=None, vhf_last=None, hermi=1):
        vj, vk = self.get_jk(mol, dm, hermi)
        return vj - 0.5 * vk

    def reset(self, mol=None):
        """pyscf.scf.hf.SCF.reset: new molecule, cached integrals dropped."""
        if mol is not None:
            self.mol = mol
            if self._int1e is not None:
                self._hcore, self._ovlp = self._int1e(mol)
        self.mo_coeff = self.mo_energy = self.e_tot = None
        self.converged = False
        return self

    def as_scanner(self):
        """pyscf.scf.hf.as_scanner: an object that, called with a molecule, resets itself to it and returns the energy."""
        base = self.__class__

        class Scanner(base):
            def __init__(self, mf):
                self.__dict__.update(copy.copy(mf.__dict__))

            def __call__(self, mol, **kw):
                self.reset(mol)
                return self.kernel()
        return Scanner(self)

    def make_rdm1(self, mo_coeff=None, nocc=None):
        c = self.mo_coeff if mo_coeff is None else mo_coeff
        nocc = self.mol.nelectron // 2 if nocc is None else nocc
        return 2.0 * c[:, :nocc] @ c[:, :nocc].T

    # --- SCF loop -----------------------------------------------------------------------------
    _np = staticmethod(_strict)

    def _converged(self, e_tot, e_last, F, c, nocc):
        """PySCF's test (scf/hf.py kernel): |E - E_last| < conv_tol and the norm of the orbital gradient g = 2 C_occ^T F C_vir of
        the orbitals the density was made of, divided by sqrt(its size), below conv_tol_grad (default sqrt(conv_tol))."""
        if c is None:
            return False
        g = 2.0 * _mm(c[:, :nocc].T, F, c[:, nocc:])
        self.norm_gorb = float(np.linalg.norm(g)) / np.sqrt(g.size)
        tol_g = np.sqrt(self.conv_tol) if self.conv_tol_grad is None else self.conv_tol_grad
        return abs(e_tot - e_last) < self.conv_tol and self.norm_gorb < tol_g

    def kernel(self, dm0=None):
        S, h = np.asarray(self._ovlp), np.asarray(self._hcore)
        s, U = _eigh(S)
        X = U[:, s > 1e-10] / np.sqrt(s[s > 1e-10])
        nocc = self.mol.nelectron // 2
        enuc = self.mol.energy_nuc()

        def solve(F):
            e, c = _eigh(_mm(X.T, F, X))
            return e, _mm(X, c)

        c_cur = None                            # orbitals the current density was made of (none for a guess density)
        if dm0 is None:
            _, c_cur = solve(h)
            dm = 2.0 * _mm(c_cur[:, :nocc], c_cur[:, :nocc].T)
        else:
            dm = np.asarray(dm0)
        dm_last, vhf_last = None, None
        errs, focks = [], []
        e_last = 0.0
        for it in range(self.max_cycle):
            vhf = self._np(self.get_veff(self.mol, dm, dm_last=dm_last, vhf_last=vhf_last, hermi=1))
            dm_last, vhf_last = dm, vhf
            F = h + vhf
            e_tot = 0.5 * float(np.einsum("ij,ji->", dm, h + F)) + enuc
            done = self._converged(e_tot, e_last, F, c_cur, nocc)
            fds = _mm(F, dm, S)
            err = _mm(X.T, fds - fds.T, X)                      # F D S - S D F (F, D, S symmetric)
            focks.append(F)
            errs.append(err)
            focks, errs = focks[-self.diis_space:], errs[-self.diis_space:]
            if len(errs) > 1:
                n = len(errs)
                B = -np.ones((n + 1, n + 1))
                B[n, n] = 0
                for a in range(n):
                    for b in range(n):
                        B[a, b] = float(np.vdot(errs[a], errs[b]))
                rhs = np.zeros(n + 1)
                rhs[n] = -1
                try:
                    w = np.linalg.solve(B, rhs)[:n]
                    F = sum(wi * Fi for wi, Fi in zip(w, focks))
                except np.linalg.LinAlgError:
                    pass
            self.cycles = it + 1
            if done:
                self.converged = True
                e_last = e_tot
                break
            self.mo_energy, self.mo_coeff = solve(F)
            c_cur = self.mo_coeff
            dm = 2.0 * _mm(c_cur[:, :nocc], c_cur[:, :nocc].T)
            e_last = e_tot
        # final energy with the converged density
        vhf = self._np(self.get_veff(self.mol, dm, dm_last=dm_last, vhf_last=vhf_last, hermi=1))
        self.e_tot = 0.5 * float(np.einsum("ij,ji->", dm, 2 * h + vhf)) + enuc
        return self.e_tot


class _LibXCStub:
    """The two libxc predicates get_veff consults; the stand-in functional is a pure (non-hybrid, non-NLC) one."""

    @staticmethod
    def is_hybrid_xc(xc):
        return False

    @staticmethod
    def is_nlc(xc):
        return False


class SlaterNumInt:
    """Stand-in for ``pyscf.dft.numint.NumInt`` restricted to Slater (Dirac) exchange, so the RKS boundary can be
    exercised without libxc:  e_x = -3/4 (3/pi)^(1/3) rho^(1/3),  v_x = 4/3 e_x."""
    libxc = _LibXCStub()
    CX = -0.75 * (3.0 / np.pi) ** (1.0 / 3.0)

    def _xc_type(self, xc_code):
        return "LDA"

    def eval_xc_eff(self, xc_code, rho, deriv=1, xctype="LDA"):
        rho = _strict(rho)                 # libxc's NumInt calls numpy.asarray on the density
        r = np.maximum(rho[0] if rho.ndim == 2 else rho, 0)
        e = self.CX * r ** (1.0 / 3.0)
        return e.reshape(-1, 1), (4.0 / 3.0 * e).reshape(1, -1)

    def rsh_and_hybrid_coeff(self, xc_code, spin=0):
        return 0.0, 0.0, 0.0

    def nlc_coeff(self, xc_code):
        return ()


class _ClosedFormLibXC:
    @staticmethod
    def is_hybrid_xc(xc_code):
        from oracle import xc
        o, a, h = xc.rsh_and_hybrid_coeff(xc_code)
        return abs(a) > 1e-10 or abs(h) > 1e-10

    @staticmethod
    def is_nlc(xc_code):
        from oracle import xc
        return bool(xc.nlc_coeff(xc_code))


class ClosedFormNumInt:
    """Stand-in for ``pyscf.dft.numint.NumInt`` + libxc for the functionals whose energies the reference's tests hold and that
    have closed forms ("lda,vwn5", "pbe", "b3lyp", "wb97", "wb97m-v": jqc/pyscf/tests/test_dft.py:75-114): oracle/xc.py behind
    the ``eval_xc_eff`` signature (NumPy in, NumPy out, like a plain CPU PySCF NumInt); ``nlc_coeff`` carries the VV10
    parameters of a functional that has a non-local part."""
    libxc = _ClosedFormLibXC()

    def _xc_type(self, xc_code):
        from oracle import xc
        return xc.xc_type(xc_code)

    def eval_xc_eff(self, xc_code, rho, deriv=1, xctype="LDA"):
        from oracle import xc
        rho = _strict(rho)
        return xc.eval_xc_eff(xc_code, rho[0] if (rho.ndim == 2 and xc.xc_type(xc_code) == "LDA") else rho)

    def rsh_and_hybrid_coeff(self, xc_code, spin=0):
        from oracle import xc
        return xc.rsh_and_hybrid_coeff(xc_code)

    def nlc_coeff(self, xc_code):
        from oracle import xc
        return xc.nlc_coeff(xc_code)


class Grids:
    def __init__(self, coords, weights):
        self.coords, self.weights = coords, weights

    def build(self, mol=None, with_non0tab=False, sort_grids=True, **kw):
        return self


class RKS(RHF):
    """Minimal restricted Kohn-Sham driver with the attribute surface ``apply`` patches on an RKS object
    (``_numint``, ``grids``, ``xc``, ``get_j/get_k/get_jk``, ``get_veff`` returning a tagged potential)."""

    def __init__(self, mol, hcore, ovlp, grids, xc="slater", numint=None, int1e=None, nlcgrids=None):
        super().__init__(mol, hcore, ovlp, int1e)
        self.grids = grids
        self.nlcgrids = grids if nlcgrids is None else nlcgrids
        self.xc = xc
        self.nlc = ""
        self._numint = numint or SlaterNumInt()
        self._eri = None

    def istype(self, name):
        return name in ("RKS", "RHF", "SCF", "KohnShamDFT")

    def do_nlc(self):
        """pyscf.dft.rks.KohnShamDFT.do_nlc: does the functional (or ``self.nlc``) carry a VV10 part?"""
        return bool(self._numint.nlc_coeff(self.xc)) if hasattr(self._numint, "nlc_coeff") else False

    def get_j(self, mol=None, dm=None, hermi=1, **kw):
        return self.get_jk(mol, dm, hermi, with_k=False, **kw)[0]

    def get_k(self, mol=None, dm=None, hermi=1, **kw):
        return self.get_jk(mol, dm, hermi, with_j=False, **kw)[1]

    def kernel(self, dm0=None):
        S, h = np.asarray(self._ovlp), np.asarray(self._hcore)
        s, U = _eigh(S)
        X = U[:, s > 1e-10] / np.sqrt(s[s > 1e-10])
        nocc = self.mol.nelectron // 2
        enuc = self.mol.energy_nuc()
        _, c = _eigh(_mm(X.T, h, X))
        c = _mm(X, c)
        dm = 2.0 * _mm(c[:, :nocc], c[:, :nocc].T) if dm0 is None else np.asarray(dm0)
        c_cur = c if dm0 is None else None
        dm_last, v_last, e_last = 0, 0, 0.0
        errs, focks = [], []
        for it in range(self.max_cycle):
            veff = self.get_veff(self.mol, dm, dm_last=dm_last, vhf_last=v_last, hermi=1)
            dm_last, v_last = dm, veff
            F = h + self._np(veff)
            _strict(veff.vj)
            e_tot = float(np.einsum("ij,ji->", dm, h)) + float(veff.ecoul) + float(veff.exc) + enuc
            self.cycles = it + 1
            if self._converged(e_tot, e_last, F, c_cur, nocc):
                self.converged = True
                break
            fds = _mm(F, dm, S)
            err = _mm(X.T, fds - fds.T, X)                      # F D S - S D F (F, D, S symmetric)
            focks.append(F); errs.append(err)
            focks, errs = focks[-self.diis_space:], errs[-self.diis_space:]
            if len(errs) > 1:
                n = len(errs)
                B = -np.ones((n + 1, n + 1)); B[n, n] = 0
                for a in range(n):
                    for b in range(n):
                        B[a, b] = float(np.vdot(errs[a], errs[b]))
                rhs = np.zeros(n + 1); rhs[n] = -1
                try:
                    w = np.linalg.solve(B, rhs)[:n]
                    F = sum(wi * Fi for wi, Fi in zip(w, focks))
                except np.linalg.LinAlgError:
                    pass
            e, cc = _eigh(_mm(X.T, F, X))
            self.mo_energy, self.mo_coeff = e, _mm(X, cc)
            c_cur = self.mo_coeff
            dm = 2.0 * _mm(c_cur[:, :nocc], c_cur[:, :nocc].T)
            e_last = e_tot
        self.e_tot = e_tot
        return e_tot
