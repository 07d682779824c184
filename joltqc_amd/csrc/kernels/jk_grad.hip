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
#define GRAD_UNROLL 1    // component loops of the one-quartet-per-lane form carry "#pragma unroll" (compile-time record offsets; the compiler
                         // gives up by itself on the largest classes); 0: "nounroll" (A/B)
#endif
#define NF_PP(l) (((l) + 1) * ((l) + 2) / 2)
#define NINT_PP (NF_PP(LI) * NF_PP(LJ) * NF_PP(LK) * NF_PP(LL))      // NINT for the preprocessor
#ifndef GRAD_WFORM
#define GRAD_WFORM (NINT_PP <= 400)   // one-quartet-per-lane form: the W form of the component loop (below) up to 400 integrals -- measured per class
                                      // on the 112-atom def2-TZVPP gradient (profiles/r05_grad_forms_per_class_112atoms_tzvpp.txt): 1.1-2.4x
                                      // faster from 27 to 360 integrals ((fp|ps) 610 -> 253 ms), slower from 540 on (its arrays spill more)
#endif
#ifndef GRAD_PMAX
#define GRAD_PMAX 1296
#endif
#ifndef GRAD_ABL
#define GRAD_ABL 0   // timing-only ablations (WRONG results; tools/grad_ab.py): 1 = no global atomics, 2 = no density gathers (P = 1),
                     // 4 = one primitive combination per quartet; cooperative form: 8 / 16 / 32 = without phase B / A2 / A1
#endif
constexpr bool P_ARRAY = NINT <= GRAD_PMAX;               // effective density as one value per component (else from its six sub-blocks)
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
#ifdef GRAD_RYS_NOUNROLL
#pragma nounroll
#endif
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

// Per-atom accumulation of the one-quartet-per-lane form: a table of the first LDS_ATOMS atoms in LDS (ds_add_f64), flushed once per
// workgroup; twelve GLOBAL atomics per quartet were 80 % of the (ps|ss) kernel's time and 54 % of (ds|ps)'s
// (profiles/r05_grad_one_lane_ablations.txt).  Atoms beyond the table go to global memory directly.
#ifndef LDS_ATOMS
#define LDS_ATOMS 1024
#endif
__device__ __forceinline__ void atom_add(double* __restrict__ table, double* __restrict__ grad, const int atom, const int x, const double v)
{
    if (atom < LDS_ATOMS) atomicAdd(&table[atom * 3 + x], v);
    else atomic_add_f64(grad + atom * 3 + x, v);
}

__device__ __forceinline__ void quartet_grad(const int nao, const real* __restrict__ basis, const real* __restrict__ dm,
                                             const int n_dm, double* __restrict__ table, double* __restrict__ grad, const int* __restrict__ shell_atom,
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
#if GRAD_ABL & 4
    const int npi = 1, npj = 1, npk = 1, npl = 1;
#else
    const int npi = (int)bi[10], npj = (int)bj[10], npk = (int)bk[10], npl = (int)bl[10];
#endif
    const int i0 = (int)bi[3], j0 = (int)bj[3], k0 = (int)bk[3], l0 = (int)bl[3];

    // effective two-particle density of the quartet (independent of the primitives): as one value per component where that
    // array is small, otherwise from the six density sub-blocks (total density for J, one set per spin for K)
    // The six density sub-blocks of the quartet are read ONCE (NFI*NFJ + NFK*NFL + n_dm (NFI + NFJ)(NFK + NFL) gathers; reading them
    // per component cost six gathers per integral: 90 % of the (fd|ps) kernel's time, profiles/r05_grad_one_lane_ablations.txt), then
    // P is formed from them: as one value per component where that array is small, otherwise on the fly in the component loop.
    const size_t nao2 = (size_t)nao * nao;
    real P[P_ARRAY ? NINT : 1];
    real tij[NFI * NFJ], tkl[NFK * NFL];
    real sik[NS_MAX][NFI * NFK], sil[NS_MAX][NFI * NFL], sjk[NS_MAX][NFJ * NFK], sjl[NS_MAX][NFJ * NFL];
    const real kscale = kfac * n_dm;
    {
        const bool two = n_dm > 1 && !(GRAD_ABL & 2);
        const real* __restrict__ D0 = dm;
        const real* __restrict__ D1 = dm + (two ? nao2 : 0);
        const real on0 = (GRAD_ABL & 2) ? real(0) : real(1);
        GUNROLL
        for (int i = 0; i < NFI; i++) {
            GUNROLL
            for (int j = 0; j < NFJ; j++) {
                const size_t a = (size_t)(i0 + i) * nao + j0 + j;
                tij[i * NFJ + j] = real(4) * jfac * (on0 * D0[a] + (two ? D1[a] : real(0)));
            }
            GUNROLL
            for (int k = 0; k < NFK; k++) {
                const size_t a = (size_t)(i0 + i) * nao + k0 + k;
                sik[0][i * NFK + k] = kscale * on0 * D0[a];
                sik[1][i * NFK + k] = two ? kscale * D1[a] : real(0);
            }
            GUNROLL
            for (int l = 0; l < NFL; l++) {
                const size_t a = (size_t)(i0 + i) * nao + l0 + l;
                sil[0][i * NFL + l] = kscale * on0 * D0[a];
                sil[1][i * NFL + l] = two ? kscale * D1[a] : real(0);
            }
        }
        GUNROLL
        for (int j = 0; j < NFJ; j++) {
            GUNROLL
            for (int k = 0; k < NFK; k++) {
                const size_t a = (size_t)(j0 + j) * nao + k0 + k;
                sjk[0][j * NFK + k] = on0 * D0[a];
                sjk[1][j * NFK + k] = two ? D1[a] : real(0);
            }
            GUNROLL
            for (int l = 0; l < NFL; l++) {
                const size_t a = (size_t)(j0 + j) * nao + l0 + l;
                sjl[0][j * NFL + l] = on0 * D0[a];
                sjl[1][j * NFL + l] = two ? D1[a] : real(0);
            }
        }
        GUNROLL
        for (int k = 0; k < NFK; k++) {
            GUNROLL
            for (int l = 0; l < NFL; l++) {
                const size_t a = (size_t)(k0 + k) * nao + l0 + l;
                tkl[k * NFL + l] = on0 * D0[a] + (two ? D1[a] : real(0));
            }
        }
    }
    if (P_ARRAY) {
        GUNROLL
        for (int i = 0; i < NFI; i++)
        GUNROLL
        for (int j = 0; j < NFJ; j++)
        GUNROLL
        for (int k = 0; k < NFK; k++)
        GUNROLL
        for (int l = 0; l < NFL; l++) {
            real p = tij[i * NFJ + j] * tkl[k * NFL + l];
#pragma unroll
            for (int s = 0; s < NS_MAX; s++)
                p -= sik[s][i * NFK + k] * sjl[s][j * NFL + l] + sil[s][i * NFL + l] * sjk[s][j * NFK + k];
#if GRAD_ABL & 2
            p = real(1);
#endif
            P[((i * NFJ + j) * NFK + k) * NFL + l] = p;
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
#if GRAD_WFORM
                // W form: the nine sums are  sum_b dX_c(b) W_x(b)  with  W_x(b) = sum over the components whose x index tuple is b of
                // P Y Z  (and the same for y, z): the component loop touches the three undifferentiated 1-D values and three
                // accumulators (6 flops, 3 + 1 loads) instead of three 32-byte records and nine sums (13 flops, 12 + 1 loads); the
                // derivative values are used once per index tuple in the epilogue.  The lane's arrays live in scratch for all but the
                // smallest classes: the traffic of the component loop is what this form cuts.
                real v[3][GSB], d[3][GSB][3], W[3][GSB];
                {
                    real g[GSZ];
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) {
                        axis_integrals_g(ax == 0 ? ckcl : ax == 1 ? gy0 : wt, rpa[ax] - rt_aij * rpq[ax], rqc[ax] + rt_akl * rpq[ax], b10, b01, b00,
                                         rij[ax], rkl[ax], g);
                        for (int i = 0; i <= LI; i++)
                            for (int j = 0; j <= LJ; j++)
                                for (int k = 0; k <= LK; k++)
                                    for (int l = 0; l <= LL; l++) {
                                        const int e = i * SI + j * SJ + k * SK + l;
                                        const int b = i * BI + j * BJ + k * BK + l;
                                        v[ax][b] = g[e];
                                        d[ax][b][0] = ai2 * g[e + SI] - (i ? i * g[e - SI] : real(0));
                                        d[ax][b][1] = aj2 * g[e + SJ] - (j ? j * g[e - SJ] : real(0));
                                        d[ax][b][2] = ak2 * g[e + SK] - (k ? k * g[e - SK] : real(0));
                                        W[ax][b] = 0;
                                    }
                    }
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
#pragma unroll
                        for (int s = 0; s < NS_MAX; s++)
                            p -= sik[s][i * NFK + k] * sjl[s][j * NFL + l] + sil[s][i * NFL + l] * sjk[s][j * NFK + k];
                    }
                    const real X = v[0][bx], Y = v[1][by], Z = v[2][bz];
                    W[0][bx] += p * (Y * Z);
                    W[1][by] += p * (X * Z);
                    W[2][bz] += p * (X * Y);
                }
#pragma unroll
                for (int ax = 0; ax < 3; ax++)
                    GUNROLL
                    for (int b = 0; b < GSB; b++) {
                        const real w = W[ax][b];
                        gA[ax] += w * d[ax][b][0]; gB[ax] += w * d[ax][b][1]; gC[ax] += w * d[ax][b][2];
                    }
#else
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
#pragma unroll
                        for (int s = 0; s < NS_MAX; s++)
                            p -= sik[s][i * NFK + k] * sjl[s][j * NFL + l] + sil[s][i * NFL + l] * sjk[s][j * NFK + k];
                    }
                    const real X = qx[bx][0], Y = qy[by][0], Z = qz[bz][0];
                    const real pyz = p * Y * Z, pxz = p * X * Z, pxy = p * X * Y;
                    gA[0] += pyz * qx[bx][1]; gB[0] += pyz * qx[bx][2]; gC[0] += pyz * qx[bx][3];
                    gA[1] += pxz * qy[by][1]; gB[1] += pxz * qy[by][2]; gC[1] += pxz * qy[by][3];
                    gA[2] += pxy * qz[bz][1]; gB[2] += pxy * qz[bz][2]; gC[2] += pxy * qz[bz][3];
                }
#endif  // GRAD_WFORM
            }
        }
    }
#if GRAD_ABL & 1
    if (gA[0] != real(1.2345e300)) return;
#endif
#pragma unroll
    for (int x = 0; x < 3; x++) {
        atom_add(table, grad, atom_i, x, (double)gA[x]);
        atom_add(table, grad, atom_j, x, (double)gB[x]);
        atom_add(table, grad, atom_k, x, (double)gC[x]);
        atom_add(table, grad, atom_l, x, -(double)(gA[x] + gB[x] + gC[x]));
    }
}

#ifndef GRAD_COOP
#define GRAD_COOP 0
#endif
#ifndef KNAME
#define KNAME jk_grad
#endif
#if GRAD_COOP
// =====================================================================================================================
// Cooperative form (round 3): a quartet is worked on by T = nf_i * nf_j lanes (one per bra Cartesian pair, as the row-lane J/K
// kernels of jk_tile.hip), G = 256 / T quartets per pass of a workgroup.  The one-quartet-per-lane form above keeps the 1-D
// arrays of a lane (3 axes x GSB x 4 doubles: 3.4 KB for (dp|dp), 31 KB for (ff|ff)) in scratch and reads twelve of them per
// integral and root, i.e. it runs at the speed of the scratch path.  Here they live in LDS, once per quartet:
//   per primitive combination:  Rys roots by NRG lanes of the quartet                                     -> sRW
//   per root:  A1  one lane per (quartet, axis, row i): that row of the extended 1-D array (axis_row_g)   -> sExt
//              A2  all lanes of the quartet: derivative records {g, dg/dA, dg/dB, dg/dC} per index tuple   -> sQ
//              B   lane (ci, cj): loop over the ket components, three 32-byte record reads per component,
//                  nine derivative sums in registers; the effective density from D_ij (register), D_kl (LDS, shared by the
//                  quartet's lanes) and the lane's own rows of D_ik, D_il, D_jk, D_jl (registers)
//   per quartet:  the nine sums of its T lanes are added in LDS, twelve global atomics per quartet instead of per lane.
// Quartets of one pass may have different primitive counts: the combination loop runs to the largest count of the pass.
constexpr int T = NFI * NFJ;
#ifndef LDS_ATOMS_COOP
#define LDS_ATOMS_COOP 256
#endif
// quartets per pass: as many as the 256 lanes hold, capped by the LDS their 1-D arrays take.  Two workgroups per CU (78 KB each,
// two waves per SIMD: one workgroup's barriers and thin phases hide behind the other's phase B) where that costs at most a
// quarter of the lanes, otherwise one workgroup with up to 150 KB (+ 6 KB each for the per-atom table sGA below: 2 x 78 / 156 of 160 KB).  (The host launcher repeats this arithmetic: jqc_hip.cpp,
// grad_quartets_per_pass.)
#ifndef GRAD_A1_MAP
#define GRAD_A1_MAP 1     // phase-A1 jobs dealt over the whole workgroup ordered by row (see A1 below); needs the quartets' parameters in LDS
#endif
constexpr int NPAR = 24;  // per-quartet parameters of the current primitive combination (sPar)
#ifndef QPAD
#define QPAD 2    // doubles between the record slots of two quartets (16 bytes: the records stay 16-byte aligned for ds_read_b128; an 8-byte pad cost 25 %)
#endif
// The records themselves are indexed i * CBI + j * CBJ + k * BK + l with CBJ one more than the dense stride where that is a multiple of 4: with the dense stride (BJ = 4 for a (pp) ket) the records of
// the bra index tuples a quartet's lanes read in one instruction start 32 LDS banks apart -- two bank groups for six addresses.
// (measured: stride 4 -> 5 and 8 -> 9 win -- (fd|pp) 585 -> 536 ms, (fp|fp) 275 -> 233 --, 6 -> 7 loses its LDS to fewer quartets per pass -- (dp|dp) 576 -> 726)
constexpr int CBJ = ((LK + 1) * BK) % 4 == 0 ? (LK + 1) * BK + 1 : (LK + 1) * BK, CBI = (LJ + 1) * CBJ, CGSB = (LI + 1) * CBI;
// ... and one double between their extended arrays where those would start 0 or 32 banks apart
constexpr int EPAD = (6 * GSZ) % 32 == 0 ? 1 : 0;
constexpr int QBYTES = (3 * (GSZ + 4 * CGSB) + QPAD + EPAD + 2 * NRG + NFK * NFL + 9 + (GRAD_A1_MAP ? NPAR : 0)) * 8;
constexpr int gcap(int budget) { return budget / QBYTES < 256 / T ? (budget / QBYTES < 1 ? 1 : budget / QBYTES) : 256 / T; }
#ifndef GRAD_TWO_WG
#define GRAD_TWO_WG 1
#endif
// (measured: the 256-register cap of two workgroups per CU pays up to a ket block of 18 components -- (fp|dp) 1 334 -> 967 ms --,
//  larger ket blocks spill under it -- (fd|dd) 322 -> 504 ms -- and keep one workgroup with 512 registers per lane)
constexpr bool TWO_WG = GRAD_TWO_WG && NFK * NFL <= 18 && 4 * gcap(72 * 1024) >= 3 * gcap(150 * 1024);
// effective density of the lane's components held in registers for the whole pass: up to 18 components under the 256-register cap
// of two workgroups per CU, up to 60 with one workgroup (512 registers)
constexpr bool P_REGS = NFK * NFL <= (TWO_WG ? 18 : 60);
constexpr int G = TWO_WG ? gcap(72 * 1024) : gcap(150 * 1024);
#ifdef EXPECT_G     // (the generator passes its own evaluation of this arithmetic, jqc_hip.cpp:grad_quartets_per_pass)
static_assert(EXPECT_G == G, "jqc_hip.cpp:grad_quartets_per_pass is out of sync with the constants of jk_grad.hip");
#endif
#if ((LK + 1) * (LK + 2) / 2) * ((LL + 1) * (LL + 2) / 2) <= 100
#define BUNROLL _Pragma("unroll")          // ket loop of phase B with compile-time record offsets
#ifndef GRAD_COOP_W
#define GRAD_COOP_W 1
#endif
#else
#define BUNROLL _Pragma("nounroll")
#undef GRAD_COOP_W
#define GRAD_COOP_W 0
#endif
static_assert(T <= 256, "a quartet must fit one workgroup");

__device__ __forceinline__ void rys_root_one_g(real x, real theta, real omega, const int r, const real* __restrict__ cheb,
                                               const real* __restrict__ large, real& root, real& weight)
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
        root = large[2 * r] * isx * isx * tf;
        weight = large[2 * r + 1] * isx * stf;
        return;
    }
    const int it = (int)(x * real(0.4));
    const real u = (x - real(2.5) * it) * real(0.8) - real(1);
    const real u2 = u + u;
    const real* __restrict__ c = cheb + (it * NRG + r) * (NCOEF * 2);
    real br1 = 0, br2 = 0, bw1 = 0, bw2 = 0;
#pragma unroll
    for (int k = NCOEF - 1; k >= 1; k--) {
        real t = c[2 * k] + u2 * br1 - br2; br2 = br1; br1 = t;
        t = c[2 * k + 1] + u2 * bw1 - bw2; bw2 = bw1; bw1 = t;
    }
    root = (c[0] + u * br1 - br2) * tf;
    weight = (c[1] + u * bw1 - bw2) * stf;
}

// The rows i = IE of axis_integrals_g's array (all j, k, l of that i): one phase-A1 job.  The transfer recurrence is redone per
// job (cheap), the bra transfer and the ket transfer of the job's own rows are what is shared out.  IE is a template argument
// (the caller switches on the job's row): every array index is a compile-time constant.  (A run-time row picked by a chain of
// selects compiled, passed the s..f classes and raised a memory-aperture violation in (gs|gg) on the MI355X.)
template <int IE>
__device__ __forceinline__ void axis_row_g(real g0, real c0, real cp, real b10, real b01, real b00, real rij, real rkl,
                                           real* __restrict__ out)
{
    constexpr int NA = LIJ + 2, NC = LKL + 2;
    real t[NA][NC];
    t[0][0] = g0;
    t[1][0] = c0 * g0;
#pragma unroll
    for (int a = 1; a < NA - 1; a++) t[a + 1][0] = c0 * t[a][0] + a * b10 * t[a - 1][0];
#pragma unroll
    for (int c = 0; c < NC - 1; c++)
#pragma unroll
        for (int a = 0; a < NA; a++) {
            real v = cp * t[a][c];
            if (c > 0) v += c * b01 * t[a][c - 1];
            if (a > 0) v += a * b00 * t[a - 1][c];
            t[a][c + 1] = v;
        }
    // (IE, j) exists for IE + j <= LIJ + 1: no derivative raises two indices at once
    constexpr int NJ = (LIJ + 1 - IE) < (DJ - 1) ? (LIJ + 2 - IE) : DJ;         // number of j values of this row
    real h[NJ][NC];
#pragma unroll
    for (int m = 0; m < NJ; m++)
#pragma unroll
        for (int c = 0; c < NC; c++) h[m][c] = t[IE + m][c];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        real w[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) w[c] = h[0][c];
#pragma unroll
        for (int l = 0; l < DL; l++) {
#pragma unroll
            for (int k = 0; k < DK; k++) out[IE * SI + j * SJ + k * SK + l] = w[k];
            if (l < DL - 1) {
#pragma unroll
                for (int c = 0; c < NC - 1 - l; c++) w[c] = w[c + 1] - rkl * w[c];
            }
        }
        if (j < NJ - 1) {
#pragma unroll
            for (int m = 0; m < NJ - 1 - j; m++)
#pragma unroll
                for (int c = 0; c < NC; c++) h[m][c] = h[m + 1][c] - rij * h[m][c];
        }
    }
}

extern "C" __global__ void __launch_bounds__(256, TWO_WG ? 2 : 1)
KNAME(const int nao, const real* __restrict__ basis, const real* __restrict__ dm, const int n_dm, double* __restrict__ grad,
      const int* __restrict__ shell_atom, const int natm, const int nrep, const real jfac, const real kfac, const real omega,
      const ushort4* __restrict__ quartets, const unsigned* __restrict__ ntasks_ptr, const int qstride,
      const real* __restrict__ rys_cheb, const real* __restrict__ rys_large)
{
    __shared__ real sExt[G][3 * GSZ + EPAD];                           // extended 1-D arrays of the current (combination, root)
    // derivative records of the same.  QPAD: 96 GSB bytes is a multiple of 256 for most classes, which put the records of all quartets of a wave on
    // the same LDS banks (bank conflicts on half of the LDS cycles, profiles/r05_pmc_grad_class_kernels_112atoms.txt): slots 16 bytes further apart
    __shared__ __attribute__((aligned(32))) real sQ[G][3 * CGSB * 4 + QPAD];
    __shared__ real sRW[G][2 * NRG];
    __shared__ real sDkl[G][NFK * NFL];
    __shared__ double sAcc[G][9];
    __shared__ real sPar[GRAD_A1_MAP ? G : 1][NPAR];
    __shared__ int sRec[((3 * GSB + T - 1) / T) * T], sRecQ[((3 * GSB + T - 1) / T) * T];
    __shared__ double sGA[LDS_ATOMS_COOP * 3];     // per-atom sums of this workgroup (atoms below LDS_ATOMS_COOP), flushed once at the end
    __shared__ int s_ncomb;
    const int tid = threadIdx.x;
    const int slot = tid / T, lt = tid - slot * T;
    const bool lane_on = slot < G;
    const int sl = lane_on ? slot : 0;                                  // (idle lanes alias slot 0 for addresses only)
    const int ci = lt / NFJ, cj = lt - ci * NFJ;
    const long ntasks = *ntasks_ptr;
    double* __restrict__ gout = grad + (size_t)(blockIdx.x % nrep) * natm * 3;
    const size_t nao2 = (size_t)nao * nao;
    const real kscale = kfac * n_dm;
    // index bases of this lane's bra component pair inside the record arrays
    const int bx0 = TI.x[ci] * CBI + TJ.x[cj] * CBJ, by0 = TI.y[ci] * CBI + TJ.y[cj] * CBJ, bz0 = TI.z[ci] * CBI + TJ.z[cj] * CBJ;
    // phase A2: the records lane lt of a quartet builds per root (n = lt + m T < 3 GSB), as offset into sExt[slot] | i << 20 | j << 23 |
    // k << 26: a table in LDS shared by the quartets (in registers it cost the 256-register builds spills: (dp|dp) 677 -> 841 ms)
    constexpr int NREC = (3 * GSB + T - 1) / T;
    for (int n = tid; n < NREC * T; n += 256) {
        const int ax = n / GSB, b = n - ax * GSB;
        const int i = b / BI, j = (b / BJ) % (LJ + 1), k = (b / BK) % (LK + 1), l = b % (LL + 1);
        sRec[n] = n < 3 * GSB ? ((ax * GSZ + i * SI + j * SJ + k * SK + l) | (i << 20) | (j << 23) | (k << 26)) : -1;
        sRecQ[n] = (ax * CGSB + i * CBI + j * CBJ + k * BK + l) * 4;          // where the record goes in sQ[slot]
    }
#ifdef GRAD_STAMPS       // cycle counts of the phases of workgroup 0 (wave 0), printed at the end: tools/grad_ab.py with -DGRAD_STAMPS=1
    unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#define GSTAMP(n) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[n] += t_ - st_t; st_t = t_; }
#else
#define GSTAMP(n)
#endif
    const int ntab = (natm < LDS_ATOMS_COOP ? natm : LDS_ATOMS_COOP) * 3;
    for (int n = tid; n < ntab; n += 256) sGA[n] = 0;      // (the first barrier of the pass loop orders this before any use)

    // (the quartet of the NEXT pass is requested at the top of this one: one of the three dependent global-load levels of the pass
    //  setup -- quartet -> shell rows -> density blocks -- is off the critical path)
    ushort4 sq_next = {0, 0, 0, 0};
    if (lane_on && (long)blockIdx.x * G + slot < ntasks) sq_next = quartets[((long)blockIdx.x * G + slot) * qstride];
    for (long base = (long)blockIdx.x * G; base < ntasks; base += (long)gridDim.x * G) {
        const long task = base + slot;
        bool on = lane_on && task < ntasks;
        const ushort4 sq = sq_next;
        {
            const long tn = task + (long)gridDim.x * G;
            if (lane_on && tn < ntasks) sq_next = quartets[tn * qstride];
        }
        const int ish = sq.x, jsh = sq.y, ksh = sq.z, lsh = sq.w;
        if (ksh > ish || ish < jsh || lsh > ksh) on = false;
        const int atom_i = shell_atom[ish], atom_j = shell_atom[jsh], atom_k = shell_atom[ksh], atom_l = shell_atom[lsh];
        if (atom_i == atom_j && atom_i == atom_k && atom_i == atom_l) on = false;   // translational invariance: no net force
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
        const int ncomb = on ? npi * npj * npk * npl : 0;

        // densities of this lane: D_ij (J part, total density), its rows of the four exchange blocks (per spin)
        real tij = 0;
        real sik[NS_MAX][NFK], sil[NS_MAX][NFL], sjk[NS_MAX][NFK], sjl[NS_MAX][NFL];
        for (int s = 0; s < NS_MAX; s++) {
            const bool have = on && s < n_dm;
            const real* __restrict__ D = dm + (have ? s : 0) * nao2;
            if (have) tij += D[(size_t)(i0 + ci) * nao + j0 + cj];
            for (int k = 0; k < NFK; k++) {
                sik[s][k] = have ? kscale * D[(size_t)(i0 + ci) * nao + k0 + k] : real(0);
                sjk[s][k] = have ? D[(size_t)(j0 + cj) * nao + k0 + k] : real(0);
            }
            for (int l = 0; l < NFL; l++) {
                sil[s][l] = have ? kscale * D[(size_t)(i0 + ci) * nao + l0 + l] : real(0);
                sjl[s][l] = have ? D[(size_t)(j0 + cj) * nao + l0 + l] : real(0);
            }
        }
        tij *= real(4) * jfac;
        GSTAMP(9)
        // D_kl of the quartet (shared by its lanes through LDS): requested together with the lane's own gathers above, stored after the
        // barrier -- one global-load level and one barrier less per pass than gathering it after the barrier
        constexpr int NKLV = (NFK * NFL + T - 1) / T;
        real dklv[NKLV];
#pragma unroll
        for (int m = 0; m < NKLV; m++) {
            const int n = lt + m * T;
            real v = 0;
            if (on && n < NFK * NFL)
                for (int s = 0; s < n_dm; s++) v += dm[s * nao2 + (size_t)(k0 + n / NFL) * nao + l0 + n % NFL];
            dklv[m] = v;
        }
        if (tid == 0) s_ncomb = 0;                     // (its readers of the previous pass are behind that pass's last barrier)
        GSTAMP(10)
        __syncthreads();
        GSTAMP(11)                               // the previous pass has left sAcc / sDkl
        if (on) {
#pragma unroll
            for (int m = 0; m < NKLV; m++)
                if (lt + m * T < NFK * NFL) sDkl[sl][lt + m * T] = dklv[m];
            if (lt == 0) atomicMax(&s_ncomb, ncomb);
        }
        if (lane_on)
            for (int n = lt; n < 9; n += T) sAcc[sl][n] = 0;
        __syncthreads();
        const int ncomb_max = s_ncomb;

        real pkl[P_REGS ? NFK * NFL : 1];
        if (P_REGS) {
#pragma unroll
            for (int k = 0; k < NFK; k++)
#pragma unroll
                for (int l = 0; l < NFL; l++) {
                    real p = tij * sDkl[sl][k * NFL + l];
                    for (int s = 0; s < NS_MAX; s++) p -= sik[s][k] * sjl[s][l] + sil[s][l] * sjk[s][k];
                    pkl[k * NFL + l] = on ? p : real(0);
                }
        }
        real gA[3] = {0, 0, 0}, gB[3] = {0, 0, 0}, gC[3] = {0, 0, 0};
        GSTAMP(0)
        for (int cmb = 0; cmb < ncomb_max; cmb++) {
            const bool act = cmb < ncomb;
            int c_ = act ? cmb : 0;
            const int jp = c_ % (act ? npj : 1); c_ /= (act ? npj : 1);
            const int ip = c_ % (act ? npi : 1); c_ /= (act ? npi : 1);
            const int lp = c_ % (act ? npl : 1);
            const int kp = c_ / (act ? npl : 1);
            const real ck = bk[4 + 2 * kp], ak = bk[5 + 2 * kp];
            const real cl = bl[4 + 2 * lp], al = bl[5 + 2 * lp];
            const real akl = ak + al;
            const real inv_akl = real(1) / akl;
            const real al_akl = al * inv_akl;
            const real ckcl = ck * cl * exp(-ak * al_akl * rr_kl);
            const real ci_ = bi[4 + 2 * ip], ai = bi[5 + 2 * ip];
            const real cj_ = bj[4 + 2 * jp], aj = bj[5 + 2 * jp];
            const real aij = ai + aj;
            const real inv_aij = real(1) / aij;
            const real aj_aij = aj * inv_aij;
            const real cicj = fac * ci_ * cj_ * exp(-ai * aj_aij * rr_ij);
            const real rpa[3] = {rij[0] * aj_aij, rij[1] * aj_aij, rij[2] * aj_aij};
            const real rqc[3] = {rkl[0] * al_akl, rkl[1] * al_akl, rkl[2] * al_akl};
            const real rpq[3] = {rpa[0] + rix - rqc[0] - rkx, rpa[1] + riy - rqc[1] - rky, rpa[2] + riz - rqc[2] - rkz};
            const real rr = rpq[0] * rpq[0] + rpq[1] * rpq[1] + rpq[2] * rpq[2];
            const real inv = real(1) / (aij + akl);
            const real theta = aij * akl * inv;
            const real gy0 = cicj * inv_aij * inv_akl * sqrt(inv);
            const real ai2 = ai + ai, aj2 = aj + aj, ak2 = ak + ak;
            if (act)
                for (int r = lt; r < NRG; r += T) {
                    real t2, wt;
                    rys_root_one_g(rr, theta, omega, r, rys_cheb, rys_large, t2, wt);
                    sRW[sl][2 * r] = t2;
                    sRW[sl][2 * r + 1] = wt;
                }
#if GRAD_A1_MAP
            if (lane_on && lt == 0) {
                real* __restrict__ par = sPar[sl];
                par[0] = act ? real(1) : real(0);
                par[1] = ckcl; par[2] = gy0;
                par[3] = rpa[0]; par[4] = rpa[1]; par[5] = rpa[2];
                par[6] = rqc[0]; par[7] = rqc[1]; par[8] = rqc[2];
                par[9] = rpq[0]; par[10] = rpq[1]; par[11] = rpq[2];
                par[12] = rij[0]; par[13] = rij[1]; par[14] = rij[2];
                par[15] = rkl[0]; par[16] = rkl[1]; par[17] = rkl[2];
                par[18] = inv; par[19] = aij; par[20] = akl; par[21] = inv_aij; par[22] = inv_akl;
            }
#endif
            __syncthreads();
            GSTAMP(1)
            for (int ir = 0; ir < NRG; ir++) {
                // ---- A1: one lane per (quartet, axis, row i of the extended array)  (GRAD_A1_ROWS=0: per (quartet, axis), A/B)
#ifndef GRAD_A1_ROWS
#define GRAD_A1_ROWS 1
#endif
#if GRAD_A1_MAP && GRAD_A1_ROWS
                // The row is a template argument behind a switch: lanes of one wave with different rows take the switch DI ways (that
                // was half of the kernel's time, profiles/r05_grad_one_lane_ablations.txt).  The G * 3 * DI jobs of the pass are
                // therefore dealt over ALL lanes of the workgroup ordered by row, a wave holding jobs of one row only, and a job reads
                // its quartet's parameters from LDS (sPar) instead of the registers of that quartet's lanes.
                constexpr int NJW = 3 * G, NJWP = (NJW + 63) / 64 * 64;      // (padded: a wave holds jobs of ONE row)
                if (!(GRAD_ABL & 32))
                for (int J = tid; J < DI * NJWP; J += 256) {
                    const int ie = J / NJWP, r_ = J - ie * NJWP, q_ = r_ / 3, ax = r_ - 3 * q_;
                    if (r_ >= NJW) continue;
                    const real* __restrict__ par = sPar[q_];
                    if (par[0] == real(0)) continue;
                    const real t2 = sRW[q_][2 * ir], wt = sRW[q_][2 * ir + 1];
                    const real inv_ = par[18], aij_ = par[19], akl_ = par[20];
                    const real rt_aa = t2 * inv_;
                    const real rt_aij = rt_aa * akl_, rt_akl = rt_aa * aij_;
                    const real b10 = real(0.5) * par[21] * (real(1) - rt_aij);
                    const real b01 = real(0.5) * par[22] * (real(1) - rt_akl);
                    const real b00 = real(0.5) * rt_aa;
                    const real g0 = ax == 0 ? par[1] : ax == 1 ? par[2] : wt;
                    const real pa = par[3 + ax], qc = par[6 + ax], pq = par[9 + ax], dij = par[12 + ax], dkl = par[15 + ax];
                    real* __restrict__ dst = &sExt[q_][ax * GSZ];
                    const real c0 = pa - rt_aij * pq, cp = qc + rt_akl * pq;
                    switch (ie) {            // DI = LI + 2 <= 6 rows
                    case 0: axis_row_g<0>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                    case 1: axis_row_g<1>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                    case 2: if (DI > 2) axis_row_g<(DI > 2 ? 2 : 0)>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                    case 3: if (DI > 3) axis_row_g<(DI > 3 ? 3 : 0)>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                    case 4: if (DI > 4) axis_row_g<(DI > 4 ? 4 : 0)>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                    default: if (DI > 5) axis_row_g<(DI > 5 ? 5 : 0)>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                    }
                }
#else
                constexpr int NJOB = GRAD_A1_ROWS ? 3 * DI : 3;
                if (act && !(GRAD_ABL & 32))
                    for (int job = lt; job < NJOB; job += T) {
                        const int ax = GRAD_A1_ROWS ? job / DI : job, ie = GRAD_A1_ROWS ? job - ax * DI : 0;
                        const real t2 = sRW[sl][2 * ir], wt = sRW[sl][2 * ir + 1];
                        const real rt_aa = t2 * inv;
                        const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                        const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                        const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                        const real b00 = real(0.5) * rt_aa;
                        const real g0 = ax == 0 ? ckcl : ax == 1 ? gy0 : wt;
                        const real pa = ax == 0 ? rpa[0] : ax == 1 ? rpa[1] : rpa[2];
                        const real qc = ax == 0 ? rqc[0] : ax == 1 ? rqc[1] : rqc[2];
                        const real pq = ax == 0 ? rpq[0] : ax == 1 ? rpq[1] : rpq[2];
                        const real dij = ax == 0 ? rij[0] : ax == 1 ? rij[1] : rij[2];
                        const real dkl = ax == 0 ? rkl[0] : ax == 1 ? rkl[1] : rkl[2];
#if GRAD_A1_ROWS
                        real* __restrict__ dst = &sExt[sl][ax * GSZ];
                        const real c0 = pa - rt_aij * pq, cp = qc + rt_akl * pq;
                        switch (ie) {            // DI = LI + 2 <= 6 rows
                        case 0: axis_row_g<0>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                        case 1: axis_row_g<1>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                        case 2: if (DI > 2) axis_row_g<(DI > 2 ? 2 : 0)>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                        case 3: if (DI > 3) axis_row_g<(DI > 3 ? 3 : 0)>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                        case 4: if (DI > 4) axis_row_g<(DI > 4 ? 4 : 0)>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                        default: if (DI > 5) axis_row_g<(DI > 5 ? 5 : 0)>(g0, c0, cp, b10, b01, b00, dij, dkl, dst); break;
                        }
#else
                        axis_integrals_g(g0, pa - rt_aij * pq, qc + rt_akl * pq, b10, b01, b00, dij, dkl, &sExt[sl][ax * GSZ]);
#endif
                    }
#endif  // GRAD_A1_MAP
                GSTAMP(2)
                __syncthreads();
                GSTAMP(3)
                // ---- A2: derivative records {g, dg/dA, dg/dB, dg/dC} of every index tuple, all lanes of the quartet
                if (act && !(GRAD_ABL & 16)) {
                    // (index tuple and extended-array offset of each of the lane's records: worked out once per kernel, sRec)
#pragma unroll
                    for (int m = 0; m < NREC; m++) {
                        const int rc = sRec[lt + m * T];
                        if (rc < 0) continue;
                        const int i = (rc >> 20) & 7, j = (rc >> 23) & 7, k = (rc >> 26) & 7;
                        const real* __restrict__ g = &sExt[sl][0] + (rc & 0xfffff);
                        const real g0 = g[0];
                        const real dA = ai2 * g[SI] - real(i) * g[i ? -SI : 0];
                        const real dB = aj2 * g[SJ] - real(j) * g[j ? -SJ : 0];
                        const real dC = ak2 * g[SK] - real(k) * g[k ? -SK : 0];
                        real* __restrict__ q = &sQ[sl][0] + sRecQ[lt + m * T];
                        q[0] = g0; q[1] = dA; q[2] = dB; q[3] = dC;
                    }
                }
                GSTAMP(4)
                __syncthreads();
                GSTAMP(5)
                // ---- B: this lane's bra component pair against every ket component
                if (act && !(GRAD_ABL & 8)) {
                    const real (*__restrict__ qx)[4] = reinterpret_cast<const real (*)[4]>(&sQ[sl][0]);
                    const real (*__restrict__ qy)[4] = qx + CGSB;
                    const real (*__restrict__ qz)[4] = qx + 2 * CGSB;
#if GRAD_COOP_W
                    // W form (as in the one-quartet-per-lane form above): per ket component three 8-byte reads and three accumulators
                    // indexed by the ket index tuple (compile-time indices: registers), then one 32-byte record per tuple and axis
                    constexpr int NKT = (LK + 1) * (LL + 1);
                    real wx[NKT], wy[NKT], wz[NKT];
#pragma unroll
                    for (int t = 0; t < NKT; t++) wx[t] = wy[t] = wz[t] = 0;
#pragma unroll
                    for (int k = 0; k < NFK; k++)
#pragma unroll
                    for (int l = 0; l < NFL; l++) {
                        const int tx = TK.x[k] * BK + TL.x[l], ty = TK.y[k] * BK + TL.y[l], tz = TK.z[k] * BK + TL.z[l];
                        real p;
                        if (P_REGS) p = pkl[k * NFL + l];
                        else {
                            p = tij * sDkl[sl][k * NFL + l];
                            for (int s = 0; s < NS_MAX; s++) p -= sik[s][k] * sjl[s][l] + sil[s][l] * sjk[s][k];
                        }
                        const real X = qx[bx0 + tx][0], Y = qy[by0 + ty][0], Z = qz[bz0 + tz][0];
                        wx[tx] += p * (Y * Z);
                        wy[ty] += p * (X * Z);
                        wz[tz] += p * (X * Y);
                    }
#pragma unroll
                    for (int t = 0; t < NKT; t++) {
                        gA[0] += wx[t] * qx[bx0 + t][1]; gB[0] += wx[t] * qx[bx0 + t][2]; gC[0] += wx[t] * qx[bx0 + t][3];
                        gA[1] += wy[t] * qy[by0 + t][1]; gB[1] += wy[t] * qy[by0 + t][2]; gC[1] += wy[t] * qy[by0 + t][3];
                        gA[2] += wz[t] * qz[bz0 + t][1]; gB[2] += wz[t] * qz[bz0 + t][2]; gC[2] += wz[t] * qz[bz0 + t][3];
                    }
#else
                    BUNROLL
                    for (int k = 0; k < NFK; k++)
                    BUNROLL
                    for (int l = 0; l < NFL; l++) {
                        const int bx = bx0 + TK.x[k] * BK + TL.x[l];
                        const int by = by0 + TK.y[k] * BK + TL.y[l];
                        const int bz = bz0 + TK.z[k] * BK + TL.z[l];
                        real p;
                        if (P_REGS) p = pkl[k * NFL + l];
                        else {
                            p = tij * sDkl[sl][k * NFL + l];
                            for (int s = 0; s < NS_MAX; s++) p -= sik[s][k] * sjl[s][l] + sil[s][l] * sjk[s][k];
                        }
                        const real X = qx[bx][0], Y = qy[by][0], Z = qz[bz][0];
                        const real pyz = p * Y * Z, pxz = p * X * Z, pxy = p * X * Y;
                        gA[0] += pyz * qx[bx][1]; gB[0] += pyz * qx[bx][2]; gC[0] += pyz * qx[bx][3];
                        gA[1] += pxz * qy[by][1]; gB[1] += pxz * qy[by][2]; gC[1] += pxz * qy[by][3];
                        gA[2] += pxy * qz[bz][1]; gB[2] += pxy * qz[bz][2]; gC[2] += pxy * qz[bz][3];
                    }
#endif  // GRAD_COOP_W
                }
                GSTAMP(6)
                // (the next A1 writes sExt only; its barrier separates this B from the next A2)
            }
        }
        // ---- sum over the quartet's lanes in LDS, then twelve global atomics per quartet
        GSTAMP(8)
        if (on) {
#pragma unroll
            for (int x = 0; x < 3; x++) {
                atomicAdd(&sAcc[sl][x], (double)gA[x]);
                atomicAdd(&sAcc[sl][3 + x], (double)gB[x]);
                atomicAdd(&sAcc[sl][6 + x], (double)gC[x]);
            }
        }
        __syncthreads();
        if (on)
            for (int n = lt; n < 12; n += T) {
                const int c = n / 3, x = n - c * 3;
                const double v = c < 3 ? sAcc[sl][c * 3 + x] : -(sAcc[sl][x] + sAcc[sl][3 + x] + sAcc[sl][6 + x]);
                const int atom = c == 0 ? atom_i : c == 1 ? atom_j : c == 2 ? atom_k : atom_l;
                if (atom < LDS_ATOMS_COOP) atomicAdd(&sGA[atom * 3 + x], v);
                else atomic_add_f64(gout + atom * 3 + x, v);
            }
    }
    __syncthreads();
    for (int n = tid; n < ntab; n += 256) {
        const double v = sGA[n];
        if (v != 0.0) atomic_add_f64(gout + n, v);
    }
#ifdef GRAD_STAMPS
    GSTAMP(7)
    if (blockIdx.x == 0 && tid == 0)
        printf("stamps (cycles of s_memtime): pass setup after its first barrier %llu  combination setup + Rys %llu  A1 %llu  barrier %llu  A2 %llu  barrier %llu  B %llu  tail %llu"
               "  (loop tail of B %llu)  previous pass's reduction + flush + this pass's quartet, rows, density gathers issued %llu  D_kl gathers %llu  first barrier %llu\n",
               st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8], st[9], st[10], st[11]);
#endif
}
#else
#ifndef GRAD_MINW
#define GRAD_MINW 1      // workgroups per CU the register allocation aims at (A/B: tools/grad_ab.py)
#endif
extern "C" __global__ void __launch_bounds__(BLOCK, GRAD_MINW)
KNAME(const int nao, const real* __restrict__ basis, const real* __restrict__ dm, const int n_dm, double* __restrict__ grad,
      const int* __restrict__ shell_atom, const int natm, const int nrep, const real jfac, const real kfac, const real omega,
      const ushort4* __restrict__ quartets, const unsigned* __restrict__ ntasks_ptr, const int qstride,
      const real* __restrict__ rys_cheb, const real* __restrict__ rys_large)
{
    __shared__ double sG[LDS_ATOMS * 3];
    // Chebyshev table of the class's NRG roots in LDS (as the J/K kernels keep theirs, jk_tile.hip sRys): every lane reads the 28
    // coefficients per root of ITS OWN x interval, 64 different cache lines per wave instruction when the table is in global memory
    constexpr int RYS_TAB_G = (2 * NRG + 14) * NRG * NCOEF * 2;
    constexpr bool RYS_LDS_G = RYS_TAB_G * (int)sizeof(real) <= 36 * 1024;          // up to six roots
    __shared__ real sRysG[RYS_LDS_G ? RYS_TAB_G : 1];
    if (RYS_LDS_G)
        for (int n = threadIdx.x; n < RYS_TAB_G; n += blockDim.x) sRysG[n] = rys_cheb[n];
    const real* cheb_tab = RYS_LDS_G ? sRysG : rys_cheb;
    const long ntasks = *ntasks_ptr;
    double* __restrict__ g = grad + (size_t)(blockIdx.x % nrep) * natm * 3;
    const int ntab = (natm < LDS_ATOMS ? natm : LDS_ATOMS) * 3;
    for (int n = threadIdx.x; n < ntab; n += blockDim.x) sG[n] = 0;
    __syncthreads();
    for (long task = (long)blockIdx.x * blockDim.x + threadIdx.x; task < ntasks; task += (long)gridDim.x * blockDim.x)
        quartet_grad(nao, basis, dm, n_dm, sG, g, shell_atom, jfac, kfac, omega, quartets[task * qstride], cheb_tab, rys_large);
    __syncthreads();
    for (int n = threadIdx.x; n < ntab; n += blockDim.x) {
        const double v = sG[n];
        if (v != 0.0) atomic_add_f64(g + n, v);
    }
}
#endif  // GRAD_COOP
