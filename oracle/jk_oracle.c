/*
 * oracle/jk_oracle.c -- CPU restatement of the reference's Rys-quadrature ERI -> J/K path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under joltqc_amd/ imports, links or calls this file; it is
 * the checker used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * What it restates (file:line under /root/reference):
 *   - per-quartet arithmetic of rys_1q1t_vjk            jqc/backend/jk/1q1t.cu:45-644
 *       symmetry factor / zeroing of non-canonical quartets          :86-94
 *       primitive loops, pair prefactors K_ab, K_cd                   :146-232
 *       seeds g_x = c_k c_l K_cd, g_y = c_i c_j K_ab /(p q sqrt(p+q)), g_z = w   :239-242
 *       transfer recurrence (TRR) in (i+j, k+l)                        :250-330
 *       horizontal recurrences into j and l                           :336-382
 *       integral assembly  I[ijkl] += gx*gy*gz                        :388-405
 *       the two J and four K contractions and where they land         :423-638
 *   - rys_roots (root = t^2, weight)                     jqc/backend/rys/rys_roots.cu:30-160
 *       x = theta |P-Q|^2, long-range scaling theta_fac = w^2/(w^2+theta)  :42-47
 *       asymptotic branch for x > 5 nroots + 35                        :66-85
 *       Chebyshev/Clenshaw branch, intervals of width 2.5              :109-159
 *     The Chebyshev tables themselves are NOT the reference's: they are regenerated from scratch
 *     with mpmath by tools/gen_rys_tables.py (same degree/interval structure, own coefficient
 *     convention) and passed in by the caller.  The reference's small-x linear branch and its
 *     nroots==1 erf closed form are not needed: the regenerated interval-0 polynomial is valid
 *     down to x = 0 and for nroots = 1 (max rel. error 2.5e-14 vs mpmath, tests/test_rys.py).
 *   - packed shell table  [x,y,z,ao_loc | c0,e0,c1,e1,c2,e2 | nprim, l]   jqc/pyscf/basis.py:280-371
 *     (slots 10/11 are unused by the reference; this build stores nprim and l there).
 *
 * Parity pins (tests/test_oracle.py): the two regression vectors the surveyor obtained by running
 * the reference's own 1q1t.cu on the host, the analytic (ss|ss) value,
 * an independent McMurchie-Davidson/Boys implementation (oracle/md_eri.py) for all l <= 4, and the
 * reference's hard-coded H2O/def2-TZVPP RHF energies (jqc/pyscf/tests/test_scf.py:70,77).
 *
 * Code structure is this build's own: run-time angular momenta, explicit 2-D then 4-D g arrays.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define STRIDE 12
#define LMAX 4
#define NROOTS_MAX 9
#define NCOEF 14
#define NF_MAX 15            /* (LMAX+1)(LMAX+2)/2 */
#define L1 (LMAX + 1)
#define LL1 (2 * LMAX + 1)

static const double PI_FAC = 34.98683665524972497; /* 2 pi^2.5 */

/* ---- Rys tables: blob layout built by joltqc_amd/backend/rys.py:pack_tables -------------
 * header doubles: [0]=nmax; for n=1..nmax: [2n-1]=offset of cheb_n, [2n]=offset of large_n
 * cheb_n : [nint(n)=2n+14][n][14][2]   large_n : [n][2]                                      */
static const double *g_rys = 0;

void jqc_oracle_set_rys(const double *blob) { g_rys = blob; }

static void rys_roots(int n, double x, double theta, double omega, double *rw)
{
    double tf = 1.0, stf = 1.0;
    x *= theta;
    if (omega > 0.0) {
        const double w2 = omega * omega;
        tf = w2 / (w2 + theta);
        x *= tf;
        stf = sqrt(tf);
    }
    const double *cheb = g_rys + (long)g_rys[2 * n - 1];
    const double *large = g_rys + (long)g_rys[2 * n];
    if (x >= 5.0 * n + 35.0) {
        const double ix = 1.0 / x, isx = sqrt(ix);
        for (int i = 0; i < n; i++) {
            rw[2 * i] = large[2 * i] * ix * tf;
            rw[2 * i + 1] = large[2 * i + 1] * isx * stf;
        }
        return;
    }
    const int it = (int)(x * 0.4);
    const double u = (x - 2.5 * it) * 0.8 - 1.0, u2 = 2.0 * u;
    const double *c = cheb + (long)it * n * NCOEF * 2;
    for (int i = 0; i < n; i++, c += NCOEF * 2) {
        double br1 = 0, br2 = 0, bw1 = 0, bw2 = 0;
        for (int k = NCOEF - 1; k >= 1; k--) {
            double t = c[2 * k] + u2 * br1 - br2; br2 = br1; br1 = t;
            t = c[2 * k + 1] + u2 * bw1 - bw2; bw2 = bw1; bw1 = t;
        }
        rw[2 * i] = (c[0] + u * br1 - br2) * tf;
        rw[2 * i + 1] = (c[1] + u * bw1 - bw2) * stf;
    }
}

void jqc_oracle_rys_roots(int n, double x, double theta, double omega, double *rw)
{
    rys_roots(n, x, theta, omega, rw);
}

static int cart_pow(int l, int (*p)[3])
{
    int n = 0;
    for (int lx = l; lx >= 0; lx--)
        for (int ly = l - lx; ly >= 0; ly--) { p[n][0] = lx; p[n][1] = ly; p[n][2] = l - lx - ly; n++; }
    return n;
}

/* Integral block of one shell quartet: out[((i*nfj+j)*nfk+k)*nfl+l], scaled by `fac`. */
static void eri_block(const double *bi, const double *bj, const double *bk, const double *bl,
                      double omega, double fac, double *out)
{
    const int li = (int)bi[11], lj = (int)bj[11], lk = (int)bk[11], ll = (int)bl[11];
    const int npi = (int)bi[10], npj = (int)bj[10], npk = (int)bk[10], npl = (int)bl[10];
    const int lij = li + lj, lkl = lk + ll, nroots = (lij + lkl) / 2 + 1;
    int pi_[NF_MAX][3], pj_[NF_MAX][3], pk_[NF_MAX][3], pl_[NF_MAX][3];
    const int nfi = cart_pow(li, pi_), nfj = cart_pow(lj, pj_), nfk = cart_pow(lk, pk_), nfl = cart_pow(ll, pl_);
    double rij[3], rkl[3];
    double rr_ij = 0, rr_kl = 0;
    for (int x = 0; x < 3; x++) {
        rij[x] = bj[x] - bi[x]; rr_ij += rij[x] * rij[x];
        rkl[x] = bl[x] - bk[x]; rr_kl += rkl[x] * rkl[x];
    }
    memset(out, 0, sizeof(double) * nfi * nfj * nfk * nfl);
    /* g2[axis][a][c], g4[axis][i][j][k][l] */
    static __thread double g2[3][LL1][LL1];
    static __thread double g4[3][L1][L1][L1][L1];
    static __thread double h[3][LL1][L1][LL1];   /* after j-HRR: [a<=lij-j][j][c] */
    double rw[2 * NROOTS_MAX];

    for (int kp = 0; kp < npk; kp++)
    for (int lp = 0; lp < npl; lp++) {
        const double ck = bk[4 + 2 * kp], ak = bk[5 + 2 * kp];
        const double cl = bl[4 + 2 * lp], al = bl[5 + 2 * lp];
        const double akl = ak + al, al_akl = al / akl;
        const double ckcl = ck * cl * exp(-ak * al_akl * rr_kl);
        for (int ip = 0; ip < npi; ip++)
        for (int jp = 0; jp < npj; jp++) {
            const double ci = bi[4 + 2 * ip], ai = bi[5 + 2 * ip];
            const double cj = bj[4 + 2 * jp], aj = bj[5 + 2 * jp];
            const double aij = ai + aj, aj_aij = aj / aij;
            const double cicj = fac * ci * cj * exp(-ai * aj_aij * rr_ij);
            double rpq[3], rpa[3], rqc[3], rr = 0;
            for (int x = 0; x < 3; x++) {
                rpa[x] = rij[x] * aj_aij;
                rqc[x] = rkl[x] * al_akl;
                rpq[x] = (rpa[x] + bi[x]) - (rqc[x] + bk[x]);
                rr += rpq[x] * rpq[x];
            }
            const double inv = 1.0 / (aij + akl);
            const double theta = aij * akl * inv;
            const double gy0 = cicj / (aij * akl) * sqrt(inv);
            rys_roots(nroots, rr, theta, omega, rw);
            for (int ir = 0; ir < nroots; ir++) {
                const double t2 = rw[2 * ir], wt = rw[2 * ir + 1];
                const double rt_aa = t2 * inv;
                const double rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                const double b10 = 0.5 / aij * (1.0 - rt_aij);
                const double b01 = 0.5 / akl * (1.0 - rt_akl);
                const double b00 = 0.5 * rt_aa;
                for (int x = 0; x < 3; x++) {
                    const double c0 = rpa[x] - rt_aij * rpq[x];
                    const double cp = rqc[x] + rt_akl * rpq[x];
                    double (*g)[LL1] = g2[x];
                    g[0][0] = (x == 0) ? ckcl : (x == 1) ? gy0 : wt;
                    for (int a = 0; a < lij; a++)
                        g[a + 1][0] = c0 * g[a][0] + (a > 0 ? a * b10 * g[a - 1][0] : 0.0);
                    for (int c = 0; c < lkl; c++)
                        for (int a = 0; a <= lij; a++) {
                            double v = cp * g[a][c];
                            if (c > 0) v += c * b01 * g[a][c - 1];
                            if (a > 0) v += a * b00 * g[a - 1][c];
                            g[a][c + 1] = v;
                        }
                    /* HRR on the bra: h[a][j][c], g(a,j+1) = g(a+1,j) + (Ri-Rj) g(a,j) */
                    for (int a = 0; a <= lij; a++)
                        for (int c = 0; c <= lkl; c++) h[x][a][0][c] = g[a][c];
                    for (int j = 0; j < lj; j++)
                        for (int a = 0; a <= lij - j - 1; a++)
                            for (int c = 0; c <= lkl; c++)
                                h[x][a][j + 1][c] = h[x][a + 1][j][c] - rij[x] * h[x][a][j][c];
                    /* HRR on the ket for every (i,j): g4[i][j][c][l] */
                    for (int i = 0; i <= li; i++)
                        for (int j = 0; j <= lj; j++) {
                            double t[LL1][L1];
                            for (int c = 0; c <= lkl; c++) t[c][0] = h[x][i][j][c];
                            for (int l = 0; l < ll; l++)
                                for (int c = 0; c <= lkl - l - 1; c++)
                                    t[c][l + 1] = t[c + 1][l] - rkl[x] * t[c][l];
                            for (int k = 0; k <= lk; k++)
                                for (int l = 0; l <= ll; l++) g4[x][i][j][k][l] = t[k][l];
                        }
                }
                double *o = out;
                for (int i = 0; i < nfi; i++)
                for (int j = 0; j < nfj; j++)
                for (int k = 0; k < nfk; k++)
                for (int l = 0; l < nfl; l++, o++)
                    *o += g4[0][pi_[i][0]][pj_[j][0]][pk_[k][0]][pl_[l][0]]
                        * g4[1][pi_[i][1]][pj_[j][1]][pk_[k][1]][pl_[l][1]]
                        * g4[2][pi_[i][2]][pj_[j][2]][pk_[k][2]][pl_[l][2]];
            }
        }
    }
}

/* Plain (ij|kl) block of shells (ish jsh | ksh lsh), no symmetry factor (PI_FAC included). */
void jqc_oracle_eri_block(const double *basis, int ish, int jsh, int ksh, int lsh, double omega, double *out)
{
    eri_block(basis + ish * STRIDE, basis + jsh * STRIDE, basis + ksh * STRIDE, basis + lsh * STRIDE,
              omega, PI_FAC, out);
}

/*
 * Raw (pre-epilogue) J/K accumulation over a quartet list, exactly as the reference kernel
 * defines it (1q1t.cu:423-638): dm, vj, vk are [n_dm][nao][nao] row-major, internal Cartesian order.
 *   vj[k + l*nao] += sum_ij (ij|kl) dm[i + j*nao]      vj[i + j*nao] += sum_kl (ij|kl) dm[k + l*nao]
 *   vk[i*nao + k] += sum_jl (ij|kl) dm[j*nao + l]      vk[i*nao + l] += sum_jk (ij|kl) dm[j*nao + k]
 *   vk[j*nao + k] += sum_il (ij|kl) dm[i*nao + l]      vk[j*nao + l] += sum_ik (ij|kl) dm[i*nao + k]
 */
/* shared != 0: vj / vk are shared by several OpenMP threads, every update is an atomic add (large matrices, where
 * one private copy per thread would not fit) */
#define ACC(p, v) do { if (shared) { _Pragma("omp atomic") (p) += (v); } else (p) += (v); } while (0)
static void jk_one_quartet(int nao, const double *basis, int n_dm, const double *dm, double *vj, double *vk, double omega,
                           int ish, int jsh, int ksh, int lsh, int do_j, int do_k, double *blk, int shared)
{
    const long nao2 = (long)nao * nao;
    double fac = PI_FAC;
    if (ish == jsh) fac *= 0.5;
    if (ksh == lsh) fac *= 0.5;
    if (ish == ksh && jsh == lsh) fac *= 0.5;
    if (ksh > ish || ish < jsh || lsh > ksh) return;   /* zeroed in the reference */
    const double *bi = basis + ish * STRIDE, *bj = basis + jsh * STRIDE;
    const double *bk = basis + ksh * STRIDE, *bl = basis + lsh * STRIDE;
    eri_block(bi, bj, bk, bl, omega, fac, blk);
    const int li = (int)bi[11], lj = (int)bj[11], lk = (int)bk[11], ll = (int)bl[11];
    const int nfi = (li + 1) * (li + 2) / 2, nfj = (lj + 1) * (lj + 2) / 2;
    const int nfk = (lk + 1) * (lk + 2) / 2, nfl = (ll + 1) * (ll + 2) / 2;
    const int i0 = (int)bi[3], j0 = (int)bj[3], k0 = (int)bk[3], l0 = (int)bl[3];
    for (int d = 0; d < n_dm; d++) {
        const double *D = dm + d * nao2;
        double *J = vj ? vj + d * nao2 : 0, *K = vk ? vk + d * nao2 : 0;
        const double *e = blk;
        for (int i = 0; i < nfi; i++)
        for (int j = 0; j < nfj; j++)
        for (int k = 0; k < nfk; k++)
        for (int l = 0; l < nfl; l++, e++) {
            const double v = *e;
            const int I = i0 + i, Jx = j0 + j, Kx = k0 + k, Lx = l0 + l;
            if (do_j) {
                ACC(J[Kx + (long)Lx * nao], v * D[I + (long)Jx * nao]);
                ACC(J[I + (long)Jx * nao], v * D[Kx + (long)Lx * nao]);
            }
            if (do_k) {
                ACC(K[(long)I * nao + Kx], v * D[(long)Jx * nao + Lx]);
                ACC(K[(long)I * nao + Lx], v * D[(long)Jx * nao + Kx]);
                ACC(K[(long)Jx * nao + Kx], v * D[(long)I * nao + Lx]);
                ACC(K[(long)Jx * nao + Lx], v * D[(long)I * nao + Kx]);
            }
        }
    }
}

/* nthreads <= 1: the plain serial loop.  Otherwise the quartet list is dealt to OpenMP threads, each with private
 * J/K accumulators that are summed at the end (host memory: nthreads * 2 * n_dm * nao^2 doubles). */
void jqc_oracle_jk_mt(int nao, const double *basis, int n_dm, const double *dm, double *vj, double *vk,
                      double omega, const uint16_t *quartets, long ntasks, int do_j, int do_k, int nthreads)
{
    const long nmat = (long)n_dm * nao * nao;
    if (nthreads <= 1) {
        double *blk = (double *)malloc(sizeof(double) * NF_MAX * NF_MAX * NF_MAX * NF_MAX);
        for (long t = 0; t < ntasks; t++)
            jk_one_quartet(nao, basis, n_dm, dm, vj, vk, omega, quartets[4 * t], quartets[4 * t + 1], quartets[4 * t + 2],
                           quartets[4 * t + 3], do_j, do_k, blk, 0);
        free(blk);
        return;
    }
    /* private accumulators cost nthreads * 2 * nmat doubles: beyond 1 GiB the threads share vj / vk through atomic adds */
    const int shared = (double)nthreads * 2.0 * (double)nmat * 8.0 > 1073741824.0;
#pragma omp parallel num_threads(nthreads)
    {
        double *blk = (double *)malloc(sizeof(double) * NF_MAX * NF_MAX * NF_MAX * NF_MAX);
        double *pj = shared ? vj : (double *)calloc(nmat, sizeof(double)), *pk = shared ? vk : (double *)calloc(nmat, sizeof(double));
#pragma omp for schedule(dynamic, 256)
        for (long t = 0; t < ntasks; t++)
            jk_one_quartet(nao, basis, n_dm, dm, pj, pk, omega, quartets[4 * t], quartets[4 * t + 1], quartets[4 * t + 2],
                           quartets[4 * t + 3], do_j, do_k, blk, shared);
        if (!shared) {
#pragma omp critical
            {
                if (vj) for (long n = 0; n < nmat; n++) vj[n] += pj[n];
                if (vk) for (long n = 0; n < nmat; n++) vk[n] += pk[n];
            }
            free(pj); free(pk);
        }
        free(blk);
    }
}

/*
 * CPU-baseline leg of bench.py (not a parity path): the same per-quartet work -- eri_block + the six contractions of
 * 1q1t.cu:423-638 with the real density matrix -- but the six Fock sub-blocks of a quartet go to thread-local stack
 * arrays and from there into a per-thread checksum, the way a CPU direct-SCF code digests into thread-private Fock
 * copies: no shared write, no atomic, no O(nao^2) private matrix per thread (256 threads x 2 x 3180^2 doubles would
 * be 41 GB).  Returns the sum of the per-thread checksums (keeps the work alive); `reps` passes over the list.
 */
double jqc_oracle_jk_bench(int nao, const double *basis, const double *dm, double omega, const uint16_t *quartets,
                           long ntasks, int reps, int nthreads)
{
    double total = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : total)
    {
        double *blk = (double *)malloc(sizeof(double) * NF_MAX * NF_MAX * NF_MAX * NF_MAX);
        double chk = 0;
        for (int rep = 0; rep < reps; rep++) {
#pragma omp for schedule(dynamic, 64) nowait
            for (long t = 0; t < ntasks; t++) {
                const int ish = quartets[4 * t], jsh = quartets[4 * t + 1], ksh = quartets[4 * t + 2], lsh = quartets[4 * t + 3];
                if (ksh > ish || ish < jsh || lsh > ksh) continue;
                double fac = PI_FAC;
                if (ish == jsh) fac *= 0.5;
                if (ksh == lsh) fac *= 0.5;
                if (ish == ksh && jsh == lsh) fac *= 0.5;
                const double *bi = basis + ish * STRIDE, *bj = basis + jsh * STRIDE;
                const double *bk = basis + ksh * STRIDE, *bl = basis + lsh * STRIDE;
                eri_block(bi, bj, bk, bl, omega, fac, blk);
                const int li = (int)bi[11], lj = (int)bj[11], lk = (int)bk[11], ll = (int)bl[11];
                const int nfi = (li + 1) * (li + 2) / 2, nfj = (lj + 1) * (lj + 2) / 2;
                const int nfk = (lk + 1) * (lk + 2) / 2, nfl = (ll + 1) * (ll + 2) / 2;
                const int i0 = (int)bi[3], j0 = (int)bj[3], k0 = (int)bk[3], l0 = (int)bl[3];
                /* the six density sub-blocks of the quartet are gathered ONCE (nf x nf each) and the six Fock sub-blocks are
                 * sized by the shells' own nf: no 15 x 15 zero-fill / checksum for an (ss|ss) quartet */
                double dij[NF_MAX * NF_MAX], dkl[NF_MAX * NF_MAX], djl[NF_MAX * NF_MAX], djk[NF_MAX * NF_MAX];
                double dil[NF_MAX * NF_MAX], dik[NF_MAX * NF_MAX];
                double jij[NF_MAX * NF_MAX], jkl[NF_MAX * NF_MAX], kik[NF_MAX * NF_MAX];
                double kil[NF_MAX * NF_MAX], kjk[NF_MAX * NF_MAX], kjl[NF_MAX * NF_MAX];
                for (int a = 0; a < nfi; a++) for (int b = 0; b < nfj; b++) { dij[a * nfj + b] = dm[(i0 + a) + (long)(j0 + b) * nao]; jij[a * nfj + b] = 0; }
                for (int a = 0; a < nfk; a++) for (int b = 0; b < nfl; b++) { dkl[a * nfl + b] = dm[(k0 + a) + (long)(l0 + b) * nao]; jkl[a * nfl + b] = 0; }
                for (int a = 0; a < nfj; a++) for (int b = 0; b < nfl; b++) { djl[a * nfl + b] = dm[(long)(j0 + a) * nao + l0 + b]; kjl[a * nfl + b] = 0; }
                for (int a = 0; a < nfj; a++) for (int b = 0; b < nfk; b++) { djk[a * nfk + b] = dm[(long)(j0 + a) * nao + k0 + b]; kjk[a * nfk + b] = 0; }
                for (int a = 0; a < nfi; a++) for (int b = 0; b < nfl; b++) { dil[a * nfl + b] = dm[(long)(i0 + a) * nao + l0 + b]; kil[a * nfl + b] = 0; }
                for (int a = 0; a < nfi; a++) for (int b = 0; b < nfk; b++) { dik[a * nfk + b] = dm[(long)(i0 + a) * nao + k0 + b]; kik[a * nfk + b] = 0; }
                const double *e = blk;
                for (int i = 0; i < nfi; i++)
                for (int j = 0; j < nfj; j++) {
                    double sij = 0;
                    const double vdij = dij[i * nfj + j];
                    for (int k = 0; k < nfk; k++) {
                        double sik = 0, sjk = 0;
                        const double vdjk = djk[j * nfk + k], vdik = dik[i * nfk + k];
                        for (int l = 0; l < nfl; l++, e++) {
                            const double v = *e;
                            jkl[k * nfl + l] += v * vdij;
                            sij += v * dkl[k * nfl + l];
                            sik += v * djl[j * nfl + l];
                            kil[i * nfl + l] += v * vdjk;
                            sjk += v * dil[i * nfl + l];
                            kjl[j * nfl + l] += v * vdik;
                        }
                        kik[i * nfk + k] += sik;
                        kjk[j * nfk + k] += sjk;
                    }
                    jij[i * nfj + j] += sij;
                }
                for (int n = 0; n < nfi * nfj; n++) chk += jij[n];
                for (int n = 0; n < nfk * nfl; n++) chk += jkl[n];
                for (int n = 0; n < nfi * nfk; n++) chk += kik[n];
                for (int n = 0; n < nfi * nfl; n++) chk += kil[n];
                for (int n = 0; n < nfj * nfk; n++) chk += kjk[n];
                for (int n = 0; n < nfj * nfl; n++) chk += kjl[n];
            }
        }
        total += chk;
        free(blk);
    }
    return total;
}

void jqc_oracle_jk(int nao, const double *basis, int n_dm, const double *dm, double *vj, double *vk,
                   double omega, const uint16_t *quartets, long ntasks, int do_j, int do_k)
{
    jqc_oracle_jk_mt(nao, basis, n_dm, dm, vj, vk, omega, quartets, ntasks, do_j, do_k, 1);
}

/*
 * Every canonical quartet (i >= j, k >= l, (ij) >= (kl)) of the shells with skip[s] == 0, generated here instead of
 * being passed as a list (benzene/def2-TZVPP has 6e7 of them).  With log_cut > -1e30 a quartet is dropped when
 * lq[i,j] + lq[k,l] + log_dmax <= log_cut (lq = natural log of the Schwarz bound): the checker's own, much looser
 * screening (tests pass 1e-18 where the product path cuts at 1e-13) keeps big cases affordable; the neglected
 * sum is bounded by count * cutoff.  Returns the number of quartets evaluated.  OpenMP over the bra pairs.
 */
long jqc_oracle_jk_dense(int nao, const double *basis, int nbas, const unsigned char *skip, int n_dm, const double *dm,
                         double *vj, double *vk, double omega, const double *lq, double log_dmax, double log_cut,
                         int do_j, int do_k, int nthreads)
{
    const long nmat = (long)n_dm * nao * nao;
    long total = 0;
    if (nthreads < 1) nthreads = 1;
    const int shared = nthreads > 1 && (double)nthreads * 2.0 * (double)nmat * 8.0 > 1073741824.0;
#pragma omp parallel num_threads(nthreads) reduction(+ : total)
    {
        double *blk = (double *)malloc(sizeof(double) * NF_MAX * NF_MAX * NF_MAX * NF_MAX);
        double *pj = shared ? vj : (double *)calloc(nmat, sizeof(double)), *pk = shared ? vk : (double *)calloc(nmat, sizeof(double));
#pragma omp for schedule(dynamic, 1)
        for (int i = nbas - 1; i >= 0; i--) {
            if (skip[i]) continue;
            for (int j = 0; j <= i; j++) {
                if (skip[j]) continue;
                const double qij = lq ? lq[(long)i * nbas + j] : 0.0;
                for (int k = 0; k <= i; k++) {
                    if (skip[k]) continue;
                    for (int l = 0; l <= k; l++) {
                        if (skip[l] || (long)i * nbas + j < (long)k * nbas + l) continue;
                        if (lq && qij + lq[(long)k * nbas + l] + log_dmax <= log_cut) continue;
                        jk_one_quartet(nao, basis, n_dm, dm, pj, pk, omega, i, j, k, l, do_j, do_k, blk, shared);
                        total++;
                    }
                }
            }
        }
        if (!shared) {
#pragma omp critical
            {
                if (vj) for (long n = 0; n < nmat; n++) vj[n] += pj[n];
                if (vk) for (long n = 0; n < nmat; n++) vk[n] += pk[n];
            }
            free(pj); free(pk);
        }
        free(blk);
    }
    return total;
}

/*
 * Schwarz bound the way PySCF's CVHFnr_int2e_q_cond defines it (third-party, restated from its
 * published behaviour; jqc/pyscf/basis.py:840-867 consumes it):
 *     Q[i,j] = sqrt( max_{a in i, b in j} |(ab|ab)| )
 * evaluated on the split shells.  out is nbas x nbas (natural log is taken by the caller).
 */
void jqc_oracle_schwarz(const double *basis, int nbas, double omega, double *out)
{
#pragma omp parallel
    {
    double *blk = (double *)malloc(sizeof(double) * NF_MAX * NF_MAX * NF_MAX * NF_MAX);
#pragma omp for schedule(dynamic, 4)
    for (int i = 0; i < nbas; i++)
        for (int j = 0; j <= i; j++) {
            const double *bi = basis + i * STRIDE, *bj = basis + j * STRIDE;
            eri_block(bi, bj, bi, bj, omega, PI_FAC, blk);
            const int li = (int)bi[11], lj = (int)bj[11];
            const int nfi = (li + 1) * (li + 2) / 2, nfj = (lj + 1) * (lj + 2) / 2;
            double m = 0;
            for (int a = 0; a < nfi; a++)
                for (int b = 0; b < nfj; b++) {
                    const double v = fabs(blk[((a * nfj + b) * nfi + a) * nfj + b]);
                    if (v > m) m = v;
                }
            out[i * nbas + j] = out[j * nbas + i] = sqrt(m);
        }
    free(blk);
    }
}

/*
 * VV10 pair sums in double precision, restating /root/reference/jqc/backend/dft/vv10.cu:86-117:
 *     g = W0_i R^2 + K_i,  g' = W0p_j R^2 + Kp_j,  gt = g + g',  T = RpW_j / (g' (g gt)^2)
 *     F_i = -1.5 sum_j T g gt,   U_i = sum_j T (g + gt),   W_i = sum_j T R^2 (g + gt)
 * (same arithmetic as oracle/dft.py:vv10_kernel, which stays the NumPy statement of it; this one lets the known-answer SCF of
 * tests/test_dft_known_answers.py finish in seconds).  coords [n][3], vvcoords [m][3].
 */
void jqc_oracle_vv10(int n, const double *coords, const double *W0, const double *K, int m, const double *vvcoords,
                     const double *W0p, const double *Kp, const double *RpW, double *F, double *U, double *W)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        const double x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2], w0 = W0[i], k = K[i];
        double f = 0, u = 0, w = 0;
        for (int j = 0; j < m; j++) {
            const double dx = x - vvcoords[3 * j], dy = y - vvcoords[3 * j + 1], dz = z - vvcoords[3 * j + 2];
            const double r2 = dx * dx + dy * dy + dz * dz;
            const double gp = r2 * W0p[j] + Kp[j], g = r2 * w0 + k, gt = g + gp;
            const double t = RpW[j] / (gp * (g * gt) * (g * gt));
            f += t * g * gt;
            u += t * (g + gt);
            w += t * r2 * (g + gt);
        }
        F[i] = -1.5 * f; U[i] = u; W[i] = w;
    }
}
