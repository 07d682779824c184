// Schwarz bounds  Q_p = sqrt(max_ab |(ab|ab)|)  for a list of shell pairs of one (li, lj) class.
// Replaces the libcvhf call of the reference (jqc/pyscf/basis.py:840-867).  Compile with -DLI= -DLJ=.
#define LK LI
#define LL LJ
#define DO_J 0
#define DO_K 0
#ifndef RYS_LR
#define RYS_LR 0
#endif
#include "jk_common.h"
#include "jk_axis.h"

extern "C" __global__ void __launch_bounds__(64)
schwarz(const double* __restrict__ basis, const unsigned* __restrict__ pairs, const int npairs, const double omega,
        double* __restrict__ out, const double* __restrict__ rys_cheb, const double* __restrict__ rys_large)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const int ish = pairs[p] >> 16, jsh = pairs[p] & 0xffff;
    const double* __restrict__ bi = basis + ish * BASIS_STRIDE;
    const double* __restrict__ bj = basis + jsh * BASIS_STRIDE;
    const double rij[3] = {bj[0] - bi[0], bj[1] - bi[1], bj[2] - bi[2]};
    const double rr_ij = rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2];
    const int npi = (int)bi[10], npj = (int)bj[10];
    double I[NFI * NFJ];
    for (int n = 0; n < NFI * NFJ; n++) I[n] = 0;
    for (int kp = 0; kp < npi; kp++)
    for (int lp = 0; lp < npj; lp++) {
        const double ck = bi[4 + 2 * kp], ak = bi[5 + 2 * kp];
        const double cl = bj[4 + 2 * lp], al = bj[5 + 2 * lp];
        const double akl = ak + al, inv_akl = 1.0 / akl, al_akl = al * inv_akl;
        const double ckcl = ck * cl * exp(-ak * al_akl * rr_ij);
        for (int ip = 0; ip < npi; ip++)
        for (int jp = 0; jp < npj; jp++) {
            const double ci = bi[4 + 2 * ip], ai = bi[5 + 2 * ip];
            const double cj = bj[4 + 2 * jp], aj = bj[5 + 2 * jp];
            const double aij = ai + aj, inv_aij = 1.0 / aij, aj_aij = aj * inv_aij;
            const double cicj = 34.98683665524972497 * ci * cj * exp(-ai * aj_aij * rr_ij);
            const double rpa[3] = {rij[0] * aj_aij, rij[1] * aj_aij, rij[2] * aj_aij};
            const double rqc[3] = {rij[0] * al_akl, rij[1] * al_akl, rij[2] * al_akl};
            const double rpq[3] = {rpa[0] - rqc[0], rpa[1] - rqc[1], rpa[2] - rqc[2]};
            const double rr = rpq[0] * rpq[0] + rpq[1] * rpq[1] + rpq[2] * rpq[2];
            const double inv = 1.0 / (aij + akl);
            const double theta = aij * akl * inv;
            const double gy0 = cicj * inv_aij * inv_akl * sqrt(inv);
            double rw[2 * NROOTS];
            rys_roots(rr, theta, omega, rys_cheb, rys_large, rw);
            for (int ir = 0; ir < NROOTS; ir++) {
                const double t2 = rw[2 * ir], wt = rw[2 * ir + 1];
                const double rt_aa = t2 * inv;
                const double rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                const double b10 = 0.5 * inv_aij * (1.0 - rt_aij);
                const double b01 = 0.5 * inv_akl * (1.0 - rt_akl);
                const double b00 = 0.5 * rt_aa;
                double gx[GSIZE], gy[GSIZE], gz[GSIZE];
                axis_integrals(ckcl, rpa[0] - rt_aij * rpq[0], rqc[0] + rt_akl * rpq[0], b10, b01, b00, rij[0], rij[0], gx);
                axis_integrals(gy0, rpa[1] - rt_aij * rpq[1], rqc[1] + rt_akl * rpq[1], b10, b01, b00, rij[1], rij[1], gy);
                axis_integrals(wt, rpa[2] - rt_aij * rpq[2], rqc[2] + rt_akl * rpq[2], b10, b01, b00, rij[2], rij[2], gz);
                for (int i = 0; i < NFI; i++)
                for (int j = 0; j < NFJ; j++) {
                    const int ax = TI.x[i] * (GS_I + GS_K) + TJ.x[j] * (GS_J + GS_L);
                    const int ay = TI.y[i] * (GS_I + GS_K) + TJ.y[j] * (GS_J + GS_L);
                    const int az = TI.z[i] * (GS_I + GS_K) + TJ.z[j] * (GS_J + GS_L);
                    I[i * NFJ + j] += gx[ax] * gy[ay] * gz[az];
                }
            }
        }
    }
    double m = 0;
    for (int n = 0; n < NFI * NFJ; n++) m = fmax(m, fabs(I[n]));
    out[p] = sqrt(m);
}
