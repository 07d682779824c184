// J/K kernel, "one quartet per lane" form.  Entry point: jk_1q1t.
// Same kernel ABI as the reference's rys_1q1t_vjk (/root/reference/jqc/backend/jk/1q1t.cu:45-52)
// extended by the run-time arguments this build needs (n_dm, Rys table pointer):
//   jk_1q1t(nao, basis[nbas*12], dm[n_dm*nao*nao], vj, vk (f64), omega, quartets (ushort4), ntasks, n_dm, rys)
// Each lane evaluates one shell quartet completely in registers/scratch and adds its six Fock
// contributions with native f64 atomics.  It is the fully general form (any l <= 4) and the
// cross-check for the tiled kernels; the tuned classes are routed elsewhere by the host table.
#include "jk_common.h"
#include "jk_axis.h"

__device__ __forceinline__ void quartet_jk(const int nao, const real* __restrict__ basis, const real* __restrict__ dm,
                                           double* __restrict__ vj, double* __restrict__ vk, const real omega,
                                           const ushort4 sq, const int n_dm, const real* __restrict__ rys_cheb,
                                           const real* __restrict__ rys_large)
{
    const int ish = sq.x, jsh = sq.y, ksh = sq.z, lsh = sq.w;
    // canonical-order filter and degeneracy factors (reference 1q1t.cu:86-94)
    if (ksh > ish || ish < jsh || lsh > ksh) return;
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

    real I[NINT];
    UNROLL
    for (int n = 0; n < NINT; n++) I[n] = 0;

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
            real rw[2 * NROOTS];
            rys_roots(rr, theta, omega, rys_cheb, rys_large, rw);
            for (int ir = 0; ir < NROOTS; ir++) {
                const real t2 = rw[2 * ir], wt = rw[2 * ir + 1];
                const real rt_aa = t2 * inv;
                const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                const real b00 = real(0.5) * rt_aa;
                real gx[GSIZE], gy[GSIZE], gz[GSIZE];
                axis_integrals(ckcl, rpa[0] - rt_aij * rpq[0], rqc[0] + rt_akl * rpq[0], b10, b01, b00, rij[0], rkl[0], gx);
                axis_integrals(gy0, rpa[1] - rt_aij * rpq[1], rqc[1] + rt_akl * rpq[1], b10, b01, b00, rij[1], rkl[1], gy);
                axis_integrals(wt, rpa[2] - rt_aij * rpq[2], rqc[2] + rt_akl * rpq[2], b10, b01, b00, rij[2], rkl[2], gz);
                UNROLL
                for (int i = 0; i < NFI; i++)
                UNROLL
                for (int j = 0; j < NFJ; j++)
                UNROLL
                for (int k = 0; k < NFK; k++)
                UNROLL
                for (int l = 0; l < NFL; l++) {
                    const int ax = TI.x[i] * GS_I + TJ.x[j] * GS_J + TK.x[k] * GS_K + TL.x[l];
                    const int ay = TI.y[i] * GS_I + TJ.y[j] * GS_J + TK.y[k] * GS_K + TL.y[l];
                    const int az = TI.z[i] * GS_I + TJ.z[j] * GS_J + TK.z[k] * GS_K + TL.z[l];
                    I[((i * NFJ + j) * NFK + k) * NFL + l] += gx[ax] * gy[ay] * gz[az];
                }
            }
        }
    }

    const int i0 = (int)bi[3], j0 = (int)bj[3], k0 = (int)bk[3], l0 = (int)bl[3];
    const size_t nao2 = (size_t)nao * nao;
    for (int idm = 0; idm < n_dm; idm++) {
        const real* __restrict__ D = dm + idm * nao2;
#if DO_J
        {
            double* __restrict__ J = vj + idm * nao2;
            // J_kl += sum_ij (ij|kl) D_ij ;  J_ij += sum_kl (ij|kl) D_kl   (reference 1q1t.cu:426-494)
            real jkl[NFK * NFL];
            UNROLL
            for (int n = 0; n < NFK * NFL; n++) jkl[n] = 0;
            real dkl[NFK * NFL];
            UNROLL
            for (int k = 0; k < NFK; k++)
            UNROLL
            for (int l = 0; l < NFL; l++) dkl[k * NFL + l] = D[(k0 + k) + (size_t)(l0 + l) * nao];
            UNROLL
            for (int i = 0; i < NFI; i++)
            UNROLL
            for (int j = 0; j < NFJ; j++) {
                const real dij = D[(i0 + i) + (size_t)(j0 + j) * nao];
                real acc = 0;
                UNROLL
                for (int n = 0; n < NFK * NFL; n++) {
                    const real v = I[(i * NFJ + j) * NFK * NFL + n];
                    acc += v * dkl[n];
                    jkl[n] += v * dij;
                }
                atomic_add_f64(J + (i0 + i) + (size_t)(j0 + j) * nao, (double)acc);
            }
            UNROLL
            for (int k = 0; k < NFK; k++)
            UNROLL
            for (int l = 0; l < NFL; l++)
                atomic_add_f64(J + (k0 + k) + (size_t)(l0 + l) * nao, (double)jkl[k * NFL + l]);
        }
#endif
#if DO_K
        {
            double* __restrict__ K = vk + idm * nao2;
            // K_ik += (ij|kl) D_jl ; K_il += (ij|kl) D_jk ; K_jk += (ij|kl) D_il ; K_jl += (ij|kl) D_ik
            // (reference 1q1t.cu:498-637)
            real kjk[NFJ * NFK], kjl[NFJ * NFL];
            UNROLL
            for (int n = 0; n < NFJ * NFK; n++) kjk[n] = 0;
            UNROLL
            for (int n = 0; n < NFJ * NFL; n++) kjl[n] = 0;
            real djk[NFJ * NFK], djl[NFJ * NFL];
            UNROLL
            for (int j = 0; j < NFJ; j++) {
                UNROLL
                for (int k = 0; k < NFK; k++) djk[j * NFK + k] = D[(size_t)(j0 + j) * nao + k0 + k];
                UNROLL
                for (int l = 0; l < NFL; l++) djl[j * NFL + l] = D[(size_t)(j0 + j) * nao + l0 + l];
            }
            UNROLL
            for (int i = 0; i < NFI; i++) {
                real kik[NFK], kil[NFL], dik[NFK], dil[NFL];
                UNROLL
                for (int k = 0; k < NFK; k++) { kik[k] = 0; dik[k] = D[(size_t)(i0 + i) * nao + k0 + k]; }
                UNROLL
                for (int l = 0; l < NFL; l++) { kil[l] = 0; dil[l] = D[(size_t)(i0 + i) * nao + l0 + l]; }
                UNROLL
                for (int j = 0; j < NFJ; j++)
                UNROLL
                for (int k = 0; k < NFK; k++)
                UNROLL
                for (int l = 0; l < NFL; l++) {
                    const real v = I[((i * NFJ + j) * NFK + k) * NFL + l];
                    kik[k] += v * djl[j * NFL + l];
                    kil[l] += v * djk[j * NFK + k];
                    kjk[j * NFK + k] += v * dil[l];
                    kjl[j * NFL + l] += v * dik[k];
                }
                UNROLL
                for (int k = 0; k < NFK; k++) atomic_add_f64(K + (size_t)(i0 + i) * nao + k0 + k, (double)kik[k]);
                UNROLL
                for (int l = 0; l < NFL; l++) atomic_add_f64(K + (size_t)(i0 + i) * nao + l0 + l, (double)kil[l]);
            }
            UNROLL
            for (int j = 0; j < NFJ; j++) {
                UNROLL
                for (int k = 0; k < NFK; k++) atomic_add_f64(K + (size_t)(j0 + j) * nao + k0 + k, (double)kjk[j * NFK + k]);
                UNROLL
                for (int l = 0; l < NFL; l++) atomic_add_f64(K + (size_t)(j0 + j) * nao + l0 + l, (double)kjl[j * NFL + l]);
            }
        }
#endif
    }
}

// The task count lives in device memory (written by the screening kernel) so that the host never
// synchronises between queue generation and the J/K launch; the grid is a bound, lanes stride.
#ifndef KNAME
#define KNAME jk_1q1t
#endif
extern "C" __global__ void __launch_bounds__(BLOCK)
KNAME(const int nao, const real* __restrict__ basis, const real* __restrict__ dm, double* __restrict__ vj,
        double* __restrict__ vk, const real omega, const ushort4* __restrict__ quartets,
        const unsigned* __restrict__ ntasks_ptr, const int qstride, const int n_dm,
        const real* __restrict__ rys_cheb, const real* __restrict__ rys_large)
{
    const long ntasks = *ntasks_ptr;
    for (long task = (long)blockIdx.x * blockDim.x + threadIdx.x; task < ntasks; task += (long)gridDim.x * blockDim.x)
        quartet_jk(nao, basis, dm, vj, vk, omega, quartets[task * qstride], n_dm, rys_cheb, rys_large);
}
