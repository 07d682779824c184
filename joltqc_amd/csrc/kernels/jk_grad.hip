// Nuclear gradient of the two-electron energy, "one quartet per lane" form.  Entry point: jk_grad_<class>.
//
// the step AFTER the SCF path.  JoltQC itself has no gradient kernels (its scanners fall back to
// GPU4PySCF's CUDA gradients, /root/reference/jqc/pyscf/__init__.py:63-97, tests/test_geom_opt.py:250-354), so there is no
// reference kernel to mirror; the mathematics is the Rys derivative scheme every Rys code uses, built on this build's own
// 1-D recurrences (jk_axis.h / reference jk/1q1t.cu:250-382 for the undifferentiated integrals):
//
//   d/dA_x [a b | c d] = 2 alpha_a [a + 1_x  b | c d] - a_x [a - 1_x  b | c d]      (primitive Gaussians, centre A of shell a)
//
// and the same for the centres of b and c; the fourth centre follows from translational invariance.  One more Rys root than
// the energy class needs: NRG = (L + 1) / 2 + 1.  The energy being differentiated is, for a set of n_dm spin densities D^s
// with total D = sum_s D^s (n_dm = 1: closed shell, D = the total density),
//
//   E2 = 1/2 j_factor sum D_ab D_cd (ab|cd)  -  1/4 k_factor n_dm sum_s sum D^s_ac D^s_bd (ab|cd)
//
// which per canonical quartet (i >= j, k >= l, ij >= kl; degeneracy factor inside the integral, as in jk_1q1t.hip) is
//   sum_abcd (ab|cd) P_abcd,   P = 4 j_factor D_ab D_cd - k_factor n_dm sum_s (D^s_ac D^s_bd + D^s_ad D^s_bc).
// The lane accumulates the nine derivative sums (three centres x three directions) over primitives, roots and components and
// adds them to the per-atom gradient (replicated `nrep` times to spread same-address atomics; the host sums the replicas).
#include "jk_common.h"
#include "jk_axis.h"

constexpr int NRG = (LIJ + LKL + 1) / 2 + 1;          // Rys roots of the derivative class
constexpr int DI = LI + 2, DJ = LJ + 2, DK = LK + 2, DL = LL + 1;
constexpr int SL = 1, SK = DL, SJ = DK * DL, SI = DJ * DK * DL;
constexpr int GSZ = DI * SI;
#ifndef GRAD_UNROLL
#define GRAD_UNROLL (NINT <= 108)
#endif
constexpr bool P_ARRAY = NINT <= 1296;               // effective density as one value per component (else from its six sub-blocks)
constexpr int NS_MAX = 2;                            // spin densities (n_dm <= 2)
#if GRAD_UNROLL
#define GUNROLL _Pragma("unroll")
#else
#define GUNROLL _Pragma("nounroll")
#endif

// rys_roots of jk_common.h for NRG roots (the tables passed in are those of NRG)
__device__ __forceinline__ void rys_roots_g(real x, real theta, real omega, const real* __restrict__ cheb,
                                            const real* __restrict__ large, real* __restrict__ rw)
{
    real tf = 1, stf = 1;
    x *= theta;
#if RYS_LR
    {
        const real w2 = omega * omega;
        tf = w2 / (w2 + theta);
        x *= tf;
        stf = sqrt(tf);
    }
#endif
    if (x >= real(5 * NRG + 35)) {
        const real isx = rsqrt(x);
        const real ix = isx * isx;
        for (int i = 0; i < NRG; i++) {
            rw[2 * i] = large[2 * i] * ix * tf;
            rw[2 * i + 1] = large[2 * i + 1] * isx * stf;
        }
        return;
    }
    const int it = (int)(x * real(0.4));
    const real u = (x - real(2.5) * it) * real(0.8) - real(1);
    const real u2 = u + u;
    const real* __restrict__ c = cheb + it * (NRG * NCOEF * 2);
    for (int i = 0; i < NRG; i++, c += NCOEF * 2) {
        real br1 = 0, br2 = 0, bw1 = 0, bw2 = 0;
#pragma unroll
        for (int k = NCOEF - 1; k >= 1; k--) {
            real t = c[2 * k] + u2 * br1 - br2; br2 = br1; br1 = t;
            t = c[2 * k + 1] + u2 * bw1 - bw2; bw2 = bw1; bw1 = t;
        }
        rw[2 * i] = (c[0] + u * br1 - br2) * tf;
        rw[2 * i + 1] = (c[1] + u * bw1 - bw2) * stf;
    }
}

// 1-D integrals of one axis and one root with every index range one higher than the class needs on i, j, k:
// out[i * SI + j * SJ + k * SK + l], i <= LI + 1, j <= LJ + 1, k <= LK + 1, l <= LL; entries with i + j > LIJ + 1 are not
// produced (no derivative raises two indices at once).
__device__ __forceinline__ void axis_integrals_g(real g0, real c0, real cp, real b10, real b01, real b00, real rij, real rkl,
                                                 real* __restrict__ out)
{
    constexpr int NA = LIJ + 2, NC = LKL + 2;
    real t[NA][NC];
    t[0][0] = g0;
    t[1][0] = c0 * g0;
    for (int a = 1; a < NA - 1; a++) t[a + 1][0] = c0 * t[a][0] + a * b10 * t[a - 1][0];
    for (int c = 0; c < NC - 1; c++)
        for (int a = 0; a < NA; a++) {
            real v = cp * t[a][c];
            if (c > 0) v += c * b01 * t[a][c - 1];
            if (a > 0) v += a * b00 * t[a - 1][c];
            t[a][c + 1] = v;
        }
    // bra transfer i -> j in place over a: after step j, t[a][.] holds (a, j) for a <= NA - 1 - j
    for (int j = 0; j < DJ; j++) {
        const int imax = (LIJ + 1 - j) < (LI + 1) ? (LIJ + 1 - j) : (LI + 1);
        for (int i = 0; i <= imax; i++) {
            real w[NC];
            for (int c = 0; c < NC; c++) w[c] = t[i][c];
            for (int l = 0; l < DL; l++) {
                for (int k = 0; k < DK; k++) out[i * SI + j * SJ + k * SK + l] = w[k];
                if (l < DL - 1)
                    for (int c = 0; c < NC - 1 - l; c++) w[c] = w[c + 1] - rkl * w[c];
            }
        }
        if (j < DJ - 1)
            for (int a = 0; a < NA - 1 - j; a++)
                for (int c = 0; c < NC; c++) t[a][c] = t[a + 1][c] - rij * t[a][c];
    }
}

// base index ranges of the class (the derivative records below): idx = i * BI + j * BJ + k * BK + l
constexpr int BK = LL + 1, BJ = (LK + 1) * BK, BI = (LJ + 1) * BJ, GSB = (LI + 1) * BI;

// q[idx] = {g, 2 a_i g(i+1) - i g(i-1), 2 a_j g(j+1) - j g(j-1), 2 a_k g(k+1) - k g(k-1)} from the extended array of one axis
__device__ __forceinline__ void derivative_records(const real* __restrict__ g, const real ai2, const real aj2, const real ak2,
                                                   real (*__restrict__ q)[4])
{
    for (int i = 0; i <= LI; i++)
        for (int j = 0; j <= LJ; j++)
            for (int k = 0; k <= LK; k++)
                for (int l = 0; l <= LL; l++) {
                    const int e = i * SI + j * SJ + k * SK + l;
                    const int b = i * BI + j * BJ + k * BK + l;
                    q[b][0] = g[e];
                    q[b][1] = ai2 * g[e + SI] - (i ? i * g[e - SI] : real(0));
                    q[b][2] = aj2 * g[e + SJ] - (j ? j * g[e - SJ] : real(0));
                    q[b][3] = ak2 * g[e + SK] - (k ? k * g[e - SK] : real(0));
                }
}

__device__ __forceinline__ void quartet_grad(const int nao, const real* __restrict__ basis, const real* __restrict__ dm,
                                             const int n_dm, double* __restrict__ grad, const int* __restrict__ shell_atom,
                                             const real jfac, const real kfac, const real omega, const ushort4 sq,
                                             const real* __restrict__ rys_cheb, const real* __restrict__ rys_large)
{
    const int ish = sq.x, jsh = sq.y, ksh = sq.z, lsh = sq.w;
    if (ksh > ish || ish < jsh || lsh > ksh) return;
    const int atom_i = shell_atom[ish], atom_j = shell_atom[jsh], atom_k = shell_atom[ksh], atom_l = shell_atom[lsh];
    if (atom_i == atom_j && atom_i == atom_k && atom_i == atom_l) return;      // translational invariance: no net force
    real fac = real(34.98683665524972497);  // 2 pi^2.5
    if (ish == jsh) fac *= real(0.5);
    if (ksh == lsh) fac *= real(0.5);
    if (ish == ksh && jsh == lsh) fac *= real(0.5);

    const real* __restrict__ bi = basis + ish * BASIS_STRIDE;
    const real* __restrict__ bj = basis + jsh * BASIS_STRIDE;
    const real* __restrict__ bk = basis + ksh * BASIS_STRIDE;
    const real* __restrict__ bl = basis + lsh * BASIS_STRIDE;
    const real rix = bi[0], riy = bi[1], riz = bi[2];
    const real rkx = bk[0], rky = bk[1], rkz = bk[2];
    const real rij[3] = {bj[0] - rix, bj[1] - riy, bj[2] - riz};
    const real rkl[3] = {bl[0] - rkx, bl[1] - rky, bl[2] - rkz};
    const real rr_ij = rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2];
    const real rr_kl = rkl[0] * rkl[0] + rkl[1] * rkl[1] + rkl[2] * rkl[2];
    const int npi = (int)bi[10], npj = (int)bj[10], npk = (int)bk[10], npl = (int)bl[10];
    const int i0 = (int)bi[3], j0 = (int)bj[3], k0 = (int)bk[3], l0 = (int)bl[3];

    // effective two-particle density of the quartet (independent of the primitives): as one value per component where that
    // array is small, otherwise from the six density sub-blocks (total density for J, one set per spin for K)
    const size_t nao2 = (size_t)nao * nao;
    real P[P_ARRAY ? NINT : 1];
    real tij[P_ARRAY ? 1 : NFI * NFJ], tkl[P_ARRAY ? 1 : NFK * NFL];
    real sik[P_ARRAY ? 1 : NS_MAX * NFI * NFK], sil[P_ARRAY ? 1 : NS_MAX * NFI * NFL];
    real sjk[P_ARRAY ? 1 : NS_MAX * NFJ * NFK], sjl[P_ARRAY ? 1 : NS_MAX * NFJ * NFL];
    const real kscale = kfac * n_dm;
    if (P_ARRAY) {
        GUNROLL
        for (int i = 0; i < NFI; i++)
        GUNROLL
        for (int j = 0; j < NFJ; j++)
        GUNROLL
        for (int k = 0; k < NFK; k++)
        GUNROLL
        for (int l = 0; l < NFL; l++) {
            real dij = 0, dkl = 0, kk = 0;
            for (int s = 0; s < n_dm; s++) {
                const real* __restrict__ D = dm + s * nao2;
                dij += D[(size_t)(i0 + i) * nao + j0 + j];
                dkl += D[(size_t)(k0 + k) * nao + l0 + l];
                kk += D[(size_t)(i0 + i) * nao + k0 + k] * D[(size_t)(j0 + j) * nao + l0 + l] +
                      D[(size_t)(i0 + i) * nao + l0 + l] * D[(size_t)(j0 + j) * nao + k0 + k];
            }
            P[((i * NFJ + j) * NFK + k) * NFL + l] = real(4) * jfac * dij * dkl - kscale * kk;
        }
    } else {
        for (int i = 0; i < NFI; i++)
            for (int j = 0; j < NFJ; j++) {
                real v = 0;
                for (int s = 0; s < n_dm; s++) v += dm[s * nao2 + (size_t)(i0 + i) * nao + j0 + j];
                tij[i * NFJ + j] = real(4) * jfac * v;
            }
        for (int k = 0; k < NFK; k++)
            for (int l = 0; l < NFL; l++) {
                real v = 0;
                for (int s = 0; s < n_dm; s++) v += dm[s * nao2 + (size_t)(k0 + k) * nao + l0 + l];
                tkl[k * NFL + l] = v;
            }
        for (int s = 0; s < n_dm; s++) {
            const real* __restrict__ D = dm + s * nao2;
            for (int i = 0; i < NFI; i++) {
                for (int k = 0; k < NFK; k++) sik[(s * NFI + i) * NFK + k] = kscale * D[(size_t)(i0 + i) * nao + k0 + k];
                for (int l = 0; l < NFL; l++) sil[(s * NFI + i) * NFL + l] = kscale * D[(size_t)(i0 + i) * nao + l0 + l];
            }
            for (int j = 0; j < NFJ; j++) {
                for (int k = 0; k < NFK; k++) sjk[(s * NFJ + j) * NFK + k] = D[(size_t)(j0 + j) * nao + k0 + k];
                for (int l = 0; l < NFL; l++) sjl[(s * NFJ + j) * NFL + l] = D[(size_t)(j0 + j) * nao + l0 + l];
            }
        }
    }

    real gA[3] = {0, 0, 0}, gB[3] = {0, 0, 0}, gC[3] = {0, 0, 0};
    for (int kp = 0; kp < npk; kp++)
    for (int lp = 0; lp < npl; lp++) {
        const real ck = bk[4 + 2 * kp], ak = bk[5 + 2 * kp];
        const real cl = bl[4 + 2 * lp], al = bl[5 + 2 * lp];
        const real akl = ak + al;
        const real inv_akl = real(1) / akl;
        const real al_akl = al * inv_akl;
        const real ckcl = ck * cl * exp(-ak * al_akl * rr_kl);
        for (int ip = 0; ip < npi; ip++)
        for (int jp = 0; jp < npj; jp++) {
            const real ci = bi[4 + 2 * ip], ai = bi[5 + 2 * ip];
            const real cj = bj[4 + 2 * jp], aj = bj[5 + 2 * jp];
            const real aij = ai + aj;
            const real inv_aij = real(1) / aij;
            const real aj_aij = aj * inv_aij;
            const real cicj = fac * ci * cj * exp(-ai * aj_aij * rr_ij);
            const real rpa[3] = {rij[0] * aj_aij, rij[1] * aj_aij, rij[2] * aj_aij};
            const real rqc[3] = {rkl[0] * al_akl, rkl[1] * al_akl, rkl[2] * al_akl};
            const real rpq[3] = {rpa[0] + rix - rqc[0] - rkx, rpa[1] + riy - rqc[1] - rky, rpa[2] + riz - rqc[2] - rkz};
            const real rr = rpq[0] * rpq[0] + rpq[1] * rpq[1] + rpq[2] * rpq[2];
            const real inv = real(1) / (aij + akl);
            const real theta = aij * akl * inv;
            const real gy0 = cicj * inv_aij * inv_akl * sqrt(inv);
            const real ai2 = ai + ai, aj2 = aj + aj, ak2 = ak + ak;
            real rw[2 * NRG];
            rys_roots_g(rr, theta, omega, rys_cheb, rys_large, rw);
            for (int ir = 0; ir < NRG; ir++) {
                const real t2 = rw[2 * ir], wt = rw[2 * ir + 1];
                const real rt_aa = t2 * inv;
                const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                const real b00 = real(0.5) * rt_aa;
                // per axis: the 1-D integrals over the class's own index ranges together with their three centre derivatives,
                // q[idx][0..3] = {g, dg/dA, dg/dB, dg/dC} (one 32-byte record per index: three wide loads per component below)
                real qx[GSB][4], qy[GSB][4], qz[GSB][4];
                {
                    real g[GSZ];
                    axis_integrals_g(ckcl, rpa[0] - rt_aij * rpq[0], rqc[0] + rt_akl * rpq[0], b10, b01, b00, rij[0], rkl[0], g);
                    derivative_records(g, ai2, aj2, ak2, qx);
                    axis_integrals_g(gy0, rpa[1] - rt_aij * rpq[1], rqc[1] + rt_akl * rpq[1], b10, b01, b00, rij[1], rkl[1], g);
                    derivative_records(g, ai2, aj2, ak2, qy);
                    axis_integrals_g(wt, rpa[2] - rt_aij * rpq[2], rqc[2] + rt_akl * rpq[2], b10, b01, b00, rij[2], rkl[2], g);
                    derivative_records(g, ai2, aj2, ak2, qz);
                }
                GUNROLL
                for (int i = 0; i < NFI; i++)
                GUNROLL
                for (int j = 0; j < NFJ; j++)
                GUNROLL
                for (int k = 0; k < NFK; k++)
                GUNROLL
                for (int l = 0; l < NFL; l++) {
                    const int bx = TI.x[i] * BI + TJ.x[j] * BJ + TK.x[k] * BK + TL.x[l];
                    const int by = TI.y[i] * BI + TJ.y[j] * BJ + TK.y[k] * BK + TL.y[l];
                    const int bz = TI.z[i] * BI + TJ.z[j] * BJ + TK.z[k] * BK + TL.z[l];
                    real p;
                    if (P_ARRAY) p = P[((i * NFJ + j) * NFK + k) * NFL + l];
                    else {
                        p = tij[i * NFJ + j] * tkl[k * NFL + l];
                        for (int s = 0; s < n_dm; s++)
                            p -= sik[(s * NFI + i) * NFK + k] * sjl[(s * NFJ + j) * NFL + l] +
                                 sil[(s * NFI + i) * NFL + l] * sjk[(s * NFJ + j) * NFK + k];
                    }
                    const real X = qx[bx][0], Y = qy[by][0], Z = qz[bz][0];
                    const real pyz = p * Y * Z, pxz = p * X * Z, pxy = p * X * Y;
                    gA[0] += pyz * qx[bx][1]; gB[0] += pyz * qx[bx][2]; gC[0] += pyz * qx[bx][3];
                    gA[1] += pxz * qy[by][1]; gB[1] += pxz * qy[by][2]; gC[1] += pxz * qy[by][3];
                    gA[2] += pxy * qz[bz][1]; gB[2] += pxy * qz[bz][2]; gC[2] += pxy * qz[bz][3];
                }
            }
        }
    }
#pragma unroll
    for (int x = 0; x < 3; x++) {
        atomic_add_f64(grad + atom_i * 3 + x, (double)gA[x]);
        atomic_add_f64(grad + atom_j * 3 + x, (double)gB[x]);
        atomic_add_f64(grad + atom_k * 3 + x, (double)gC[x]);
        atomic_add_f64(grad + atom_l * 3 + x, -(double)(gA[x] + gB[x] + gC[x]));
    }
}

#ifndef KNAME
#define KNAME jk_grad
#endif
extern "C" __global__ void __launch_bounds__(BLOCK)
KNAME(const int nao, const real* __restrict__ basis, const real* __restrict__ dm, const int n_dm, double* __restrict__ grad,
      const int* __restrict__ shell_atom, const int natm, const int nrep, const real jfac, const real kfac, const real omega,
      const ushort4* __restrict__ quartets, const unsigned* __restrict__ ntasks_ptr, const int qstride,
      const real* __restrict__ rys_cheb, const real* __restrict__ rys_large)
{
    const long ntasks = *ntasks_ptr;
    double* __restrict__ g = grad + (size_t)(blockIdx.x % nrep) * natm * 3;
    for (long task = (long)blockIdx.x * blockDim.x + threadIdx.x; task < ntasks; task += (long)gridDim.x * blockDim.x)
        quartet_grad(nao, basis, dm, n_dm, g, shell_atom, jfac, kfac, omega, quartets[task * qstride], rys_cheb, rys_large);
}
