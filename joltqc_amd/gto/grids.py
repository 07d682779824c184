"""Atom-centred Becke quadrature grids (stand-in for ``pyscf.dft.gen_grid`` on images without PySCF).

Grid GENERATION is third-party in the reference (``jqc/pyscf/rks.py:128-133`` calls GPU4PySCF's ``gen_atomic_grids`` /
``get_partition``) and stays outside the hot path here as well: this module exists so that the grid path can be exercised,
benchmarked and checked for ``int rho = N_e`` at full size on a box that has no PySCF.  It is a plain textbook construction:

  * radial: Gauss-Chebyshev points of the second kind mapped with Becke's r = R (1 + x) / (1 - x), R = the Bragg-Slater radius
    halved for everything but hydrogen (Becke, J. Chem. Phys. 88, 2547 (1988));
  * angular: Gauss-Legendre in cos(theta) x uniform in phi (exact for spherical harmonics up to degree 2 n_theta - 1);
  * partition: Becke's fuzzy cells, three iterations of p(mu) = 3/2 mu - 1/2 mu^3, without atomic-size adjustment.

``Grids(mol, nrad, ntheta).build()`` gives ``coords [n, 3]`` (Bohr) and ``weights [n]`` like PySCF's ``Grids``; points
with negligible weight are dropped.  Runs on the GPU through torch when one is there (the partition is O(N_atoms^2 n)).
"""
import numpy as np

# Bragg-Slater radii (Angstrom) of the elements the bundled basis sets cover
_BRAGG = {1: 0.35, 2: 1.40, 3: 1.45, 4: 1.05, 5: 0.85, 6: 0.70, 7: 0.65, 8: 0.60, 9: 0.50, 10: 1.50,
          15: 1.00, 16: 1.00, 17: 1.00}
_ANG2BOHR = 1.8897261246257702


def _radial(n, R):
    i = np.arange(1, n + 1)
    x = np.cos(i * np.pi / (n + 1))
    w = np.pi / (n + 1) * np.sin(i * np.pi / (n + 1)) ** 2            # Gauss-Chebyshev (2nd kind) weights of sqrt(1-x^2)
    r = R * (1 + x) / (1 - x)
    dr = 2 * R / (1 - x) ** 2
    return r, w / np.sqrt(1 - x * x) * dr * r * r


def _angular(ntheta):
    ct, wt = np.polynomial.legendre.leggauss(ntheta)
    nphi = 2 * ntheta
    phi = (np.arange(nphi) + 0.5) * 2 * np.pi / nphi
    st = np.sqrt(1 - ct * ct)
    xyz = np.stack([np.outer(st, np.cos(phi)).ravel(), np.outer(st, np.sin(phi)).ravel(), np.repeat(ct, nphi)], 1)
    return xyz, np.repeat(wt, nphi) * (2 * np.pi / nphi)


class Grids:
    def __init__(self, mol, nrad=50, ntheta=10, prune_below=1e-14):
        self.mol = mol
        self.nrad, self.ntheta, self.prune_below = nrad, ntheta, prune_below
        self.coords = None
        self.weights = None

    def build(self, mol=None, with_non0tab=False, sort_grids=True, **kwargs):
        import torch
        mol = mol if mol is not None else self.mol
        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        at = torch.as_tensor(np.asarray(mol.atom_coords(), dtype=np.float64), device=dev)
        Z = np.asarray(mol.atom_charges())
        natm = at.shape[0]
        ang, wang = _angular(self.ntheta)
        ang_t, wang_t = torch.as_tensor(ang, device=dev), torch.as_tensor(wang, device=dev)
        Rab = torch.cdist(at, at)
        out_c, out_w = [], []
        for a in range(natm):
            R = _BRAGG.get(int(Z[a]), 1.0) * _ANG2BOHR * (1.0 if Z[a] == 1 else 0.5)
            r, wr = _radial(self.nrad, R)
            r_t, wr_t = torch.as_tensor(r, device=dev), torch.as_tensor(wr, device=dev)
            pts = at[a] + (r_t[:, None, None] * ang_t[None]).reshape(-1, 3)
            w = (wr_t[:, None] * wang_t[None]).reshape(-1)
            # Becke cell functions P_b(r) = prod_{c != b} s(mu_bc), w_a = P_a / sum_b P_b
            P = torch.empty((pts.shape[0], natm), dtype=torch.float64, device=dev)
            step = max(1, (1 << 26) // (natm * natm))                 # points per chunk: the mu tensor is [chunk, natm, natm]
            eye = torch.eye(natm, dtype=torch.bool, device=dev)
            for p0 in range(0, pts.shape[0], step):
                d = torch.cdist(pts[p0:p0 + step], at)                # [n, natm]
                mu = (d[:, :, None] - d[:, None, :]) / (Rab[None] + 1e-300)      # mu_bc
                for _ in range(3):
                    mu = 1.5 * mu - 0.5 * mu ** 3
                s = 0.5 * (1.0 - mu)
                s = torch.where(eye[None], torch.ones_like(s), s)
                P[p0:p0 + step] = s.prod(dim=2)
            w = w * P[:, a] / P.sum(dim=1)
            keep = w > self.prune_below
            out_c.append(pts[keep])
            out_w.append(w[keep])
        self.coords = torch.cat(out_c).cpu().numpy()
        self.weights = torch.cat(out_w).cpu().numpy()
        return self
