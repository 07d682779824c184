// Pair-based Coulomb kernel for gfx950 (second J algorithm).  Entry point: pair_vj.
//
// What it computes follows the reference's rys_pair_vj (/root/reference/jqc/backend/jk/pair_vj.cu:43-465): a lane owns
// ONE bra shell pair (i >= j), walks the Schwarz-sorted list of ket shell pairs (k >= l) and keeps its block
//     J_ij += sum_{k >= l} (2 - delta_kl) (ij|kl) D_kl          (D symmetric: the host symmetrises it)
// in registers; one pass of global f64 atomics per bra pair at the very end, no ij <-> kl symmetry (every ordered pair of
// pairs is evaluated: twice the integrals of the 8-fold symmetric tile kernel, but J only, no Fock tiles, no LDS atomics).
//
// How it is organised is this build's own, for CDNA4: the 64 lanes of a wave hold 64 bra pairs and walk the ket list
// TOGETHER, so everything that belongs to the ket pair -- shell rows, primitive-pair prefactors, the D_kl block -- is
// wave-uniform and is fetched with scalar loads into SGPRs (no LDS staging, no barriers in the loop: the reference
// stages 256 ket pairs per block in shared memory and synchronises twice per block, pair_vj.cu:143-219).  The contraction
// with the density is done per Rys root (J is linear in the integrals), so the integral block itself is never stored, and the
// density is folded into the KET side once per ket pair, before the walk (jqc_pair_ket_density): the ket horizontal
// recurrence g(k,l) = sum_c H(k,l;c) t(c), H(k,l;c) = C(l, c-k) (R_k - R_l)^(l-c+k), is linear and independent of the
// primitives, so   sum_kl D_kl (ij|kl) = sum_{cx,cy,cz} E[cx,cy,cz] <ij| cx cy cz>,  E = sum_kl D_kl Hx Hy Hz,
// with <ij|c> the integrals whose ket momentum still sits on centre k (c = 0..lk+ll per axis, lk <= cx+cy+cz <= lk+ll).
// A lane therefore never runs the ket recurrence nor reads D: it multiplies its bra-transferred 1-D integrals with the
// wave-uniform coefficients E (scalar loads).  Registers
// hold the J_ij block, the three 1-D integral arrays of one root and the bra-side constants (shell centre, the pair's
// primitive prefactors {c_i c_j K_ij, 1/(a_i+a_j), a_i+a_j}); LDS holds only the Rys Chebyshev table (where it fits), so
// several workgroups share a CU and hide the scalar-load latency of the ket walk.
#include "jk_common.h"
#include "jk_axis.h"

#ifndef KNAME
#define KNAME pair_vj
#endif
#ifndef RYS_LDS_MAX
#define RYS_LDS_MAX 28672
#endif
#ifndef MINW
#define MINW 2
#endif
constexpr int PB = 256;                                                  // bra pairs per workgroup (4 independent waves)
constexpr int RYS_TAB = (2 * NROOTS + 14) * NROOTS * NCOEF * 2;
constexpr bool RYS_IN_LDS = RYS_TAB * (int)sizeof(real) <= RYS_LDS_MAX;

__device__ __forceinline__ void rys_root_one(real x, real theta, real omega, const int r, const real* cheb,
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
    if (x >= real(5 * NROOTS + 35)) {
        const real isx = rsqrt(x);
        root = large[2 * r] * isx * isx * tf;
        weight = large[2 * r + 1] * isx * stf;
        return;
    }
    const int it = (int)(x * real(0.4));
    const real u = (x - real(2.5) * it) * real(0.8) - real(1);
    const real u2 = u + u;
    const real* c = cheb + (it * NROOTS + r) * (NCOEF * 2);
    real br1 = 0, br2 = 0, bw1 = 0, bw2 = 0;
#pragma unroll
    for (int k = NCOEF - 1; k >= 1; k--) {
        real t = c[2 * k] + u2 * br1 - br2; br2 = br1; br1 = t;
        t = c[2 * k + 1] + u2 * bw1 - bw2; bw2 = bw1; bw1 = t;
    }
    root = (c[0] + u * br1 - br2) * tf;
    weight = (c[1] + u * bw1 - bw2) * stf;
}

// combined ket indices (cx, cy, cz), LK <= cx + cy + cz <= LKL, in the order jqc_pair_ket_density writes E:
// total n ascending, then cx descending, then cy descending
constexpr int ntrip_of(int lo, int hi) { int n = 0; for (int t = lo; t <= hi; t++) n += (t + 1) * (t + 2) / 2; return n; }
constexpr int NTRIP = ntrip_of(LK, LKL);
struct Trip { int x[NTRIP], y[NTRIP], z[NTRIP]; };
constexpr Trip make_trips()
{
    Trip t{};
    int n = 0;
    for (int tot = LK; tot <= LKL; tot++)
        for (int cx = tot; cx >= 0; cx--)
            for (int cy = tot - cx; cy >= 0; cy--) { t.x[n] = cx; t.y[n] = cy; t.z[n] = tot - cx - cy; n++; }
    return t;
}
static constexpr Trip TR = make_trips();
// position of (cx, cy, cz) in that order
constexpr int trip_index(int cx, int cy, int cz)
{
    const int tot = cx + cy + cz;
    int n = ntrip_of(LK, tot - 1);
    for (int x = tot; x > cx; x--) n += tot - x + 1;
    return n + (tot - cx - cy);
}
constexpr int HS_C = 1, HS_J = LKL + 1, HS_I = HS_J * (LJ + 1), HSIZE = HS_I * (LI + 1);

// 1-D integrals of one axis for one root with the bra momentum distributed over (i, j) and the ket momentum left on
// centre k: out[i * HS_I + j * HS_J + c], c = 0..LKL  (TRR in (a, c), then the bra HRR; reference 1q1t.cu:250-358)
__device__ __forceinline__ void axis_bra(real g0, real c0, real cp, real b10, real b01, real b00, real rij, real* __restrict__ out)
{
    real t[LIJ + 1][LKL + 1];
    t[0][0] = g0;
    if (LIJ > 0) {
        t[1][0] = c0 * g0;
#pragma unroll
        for (int a = 1; a < LIJ; a++) t[a + 1][0] = c0 * t[a][0] + a * b10 * t[a - 1][0];
    }
#pragma unroll
    for (int c = 0; c < LKL; c++) {
#pragma unroll
        for (int a = 0; a <= LIJ; a++) {
            real v = cp * t[a][c];
            if (c > 0) v += c * b01 * t[a][c - 1];
            if (a > 0) v += a * b00 * t[a - 1][c];
            t[a][c + 1] = v;
        }
    }
#pragma unroll
    for (int j = 0; j <= LJ; j++) {
#pragma unroll
        for (int i = 0; i <= LI; i++)
#pragma unroll
            for (int c = 0; c <= LKL; c++) out[i * HS_I + j * HS_J + c] = t[i][c];
        if (j < LJ) {
#pragma unroll
            for (int a = 0; a < LIJ - j; a++)
#pragma unroll
                for (int c = 0; c <= LKL; c++) t[a][c] = t[a + 1][c] - rij * t[a][c];
        }
    }
}

// bra_pairs / ket_pairs: ish << 16 | jsh, lists sorted by Schwarz bound (descending) inside every segment; the bra list of
// a launch is ONE (l, nprim) group pair (npi, npj are launch arguments).  *_tab: 27 reals per pair, layout of jqc_pair_table.  ket_seg[2 s], [2 s + 1] = first
// entry and length of sorted segment s.  ket_ld: log of the largest |D_kl| element of the pair; ket_E: NTRIP density coefficients per ket pair
// (jqc_pair_ket_density; they carry the (2 - delta_kl) weight of the k <-> l image).
// gridDim.y workgroups share one block of bra pairs: workgroup y takes the ket entries y, y + gridDim.y, ... of a segment.
extern "C" __global__ void __launch_bounds__(PB, MINW)
KNAME(const int nao, const real* __restrict__ basis, const real* __restrict__ ket_E, double* __restrict__ vj, const real omega,
      const unsigned* __restrict__ bra_pairs, const int n_bra, const float* __restrict__ bra_q, const real* __restrict__ bra_tab,
      const unsigned* __restrict__ ket_pairs, const float* __restrict__ ket_q, const float* __restrict__ ket_ld,
      const real* __restrict__ ket_tab, const int* __restrict__ ket_seg, const int nseg, const float log_cut,
      const float log_max_dm, const int npi, const int npj, const real* __restrict__ rys_cheb, const real* __restrict__ rys_large,
      unsigned long long* __restrict__ counter)
{
    __shared__ real sRys[RYS_IN_LDS ? RYS_TAB : 1];
    const int tid = threadIdx.x;
    const int p = blockIdx.x * PB + tid;
    const bool have = p < n_bra;
    const unsigned pij = have ? bra_pairs[p] : 0u;
    const float qij = have ? bra_q[p] : -1e30f;
    const int ish = pij >> 16, jsh = pij & 0xffff;
    if (RYS_IN_LDS) {
        for (int n = tid; n < RYS_TAB; n += PB) sRys[n] = rys_cheb[n];
        __syncthreads();
    }
    const real* cheb_tab = RYS_IN_LDS ? sRys : rys_cheb;
    const real* bi = basis + ish * BASIS_STRIDE;
    const real* bj = basis + jsh * BASIS_STRIDE;
    const real rix = bi[0], riy = bi[1], riz = bi[2];
    const real rij[3] = {bj[0] - rix, bj[1] - riy, bj[2] - riz};

    // raw-J convention of the tile kernels (the epilogue doubles and adds the transpose, reference jk.py:350-370): off-diagonal
    // shell pairs carry 1/2 of the true block, diagonal ones 1/4
    const real fbra = real(34.98683665524972497) * (ish == jsh ? real(0.25) : real(0.5));
    const real* __restrict__ ptab = bra_tab + (size_t)(have ? p : 0) * 27;
    // wave-wide largest bra bound: the early exit of the ket walk
    float qmax = qij;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qmax = fmaxf(qmax, __shfl_xor(qmax, o));
    qmax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qmax)));

    double J[NFI * NFJ];
#pragma unroll
    for (int n = 0; n < NFI * NFJ; n++) J[n] = 0;
    unsigned nq = 0;

    for (int s = 0; s < nseg; s++) {
        const int k_beg = ket_seg[2 * s], k_len = ket_seg[2 * s + 1];
        for (int kk = blockIdx.y; kk < k_len; kk += gridDim.y) {
            const int kt = k_beg + kk;
            const float qkl = ket_q[kt];
            if (qmax + qkl + log_max_dm <= log_cut) break;            // sorted segment: nothing further passes for this wave
            const float est = qkl + ket_ld[kt];
            if (!__builtin_amdgcn_readfirstlane((int)__any(qij + est > log_cut))) continue;
            const unsigned pkl = ket_pairs[kt];
            const int ksh = pkl >> 16, lsh = pkl & 0xffff;
            const real* __restrict__ bk = basis + ksh * BASIS_STRIDE;   // uniform address: scalar loads
            const real* __restrict__ bl = basis + lsh * BASIS_STRIDE;
            const real rkx = bk[0], rky = bk[1], rkz = bk[2];
            const real rkl[3] = {bl[0] - rkx, bl[1] - rky, bl[2] - rkz};
            const int npk = (int)bk[10], npl = (int)bl[10];
            const real* __restrict__ pk = ket_tab + (size_t)kt * 27;
            const real* __restrict__ E = ket_E + (size_t)kt * NTRIP;       // uniform address: scalar loads
            if (qij + est > log_cut) {
                nq++;
                // bra primitive pairs outermost, one body for all of them (rolled loops): the pair's prefactors
                // {c_i c_j K_ij, 1/(a_i+a_j), a_i+a_j} come from the per-geometry table (three L1/L2 loads per lane and
                // primitive pair, amortised over the ket primitives and the roots)
#pragma clang loop unroll(disable)
                for (int ipj = 0; ipj < npi * npj; ipj++) {
                    const int ip = ipj / npj, jp = ipj - ip * npj;
                    const real* __restrict__ pb = ptab + (ip * 3 + jp) * 3;
                    const real cicj = fbra * pb[0], inv_aij = pb[1], aij = pb[2];
                    const real aj_aij = bj[5 + 2 * jp] * inv_aij;
                    const real rpa[3] = {rij[0] * aj_aij, rij[1] * aj_aij, rij[2] * aj_aij};
#pragma clang loop unroll(disable)
                    for (int kpl = 0; kpl < npk * npl; kpl++) {
                        const int kp = kpl / npl, lp = kpl - kp * npl;
                        const real ckcl = pk[(kp * 3 + lp) * 3], inv_akl = pk[(kp * 3 + lp) * 3 + 1], akl = pk[(kp * 3 + lp) * 3 + 2];
                        const real al_akl = bl[5 + 2 * lp] * inv_akl;
                        const real rqc[3] = {rkl[0] * al_akl, rkl[1] * al_akl, rkl[2] * al_akl};
                        const real rpq[3] = {rpa[0] + rix - rqc[0] - rkx, rpa[1] + riy - rqc[1] - rky, rpa[2] + riz - rqc[2] - rkz};
                        const real rr = rpq[0] * rpq[0] + rpq[1] * rpq[1] + rpq[2] * rpq[2];
                        const real sinv = fast_rsqrt(aij + akl);
                        const real inv = sinv * sinv;
                        const real theta = aij * akl * inv;
                        const real gy0 = cicj * inv_aij * inv_akl * sinv;
#pragma clang loop unroll(disable)
                        for (int ir = 0; ir < NROOTS; ir++) {
                            real t2, wt;
                            rys_root_one(rr, theta, omega, ir, cheb_tab, rys_large, t2, wt);
                            const real rt_aa = t2 * inv;
                            const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                            const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                            const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                            const real b00 = real(0.5) * rt_aa;
                            real hx[HSIZE], hy[HSIZE], hz[HSIZE];
                            axis_bra(ckcl, rpa[0] - rt_aij * rpq[0], rqc[0] + rt_akl * rpq[0], b10, b01, b00, rij[0], hx);
                            __builtin_amdgcn_sched_barrier(0);
                            axis_bra(gy0, rpa[1] - rt_aij * rpq[1], rqc[1] + rt_akl * rpq[1], b10, b01, b00, rij[1], hy);
                            __builtin_amdgcn_sched_barrier(0);
                            axis_bra(wt, rpa[2] - rt_aij * rpq[2], rqc[2] + rt_akl * rpq[2], b10, b01, b00, rij[2], hz);
                            __builtin_amdgcn_sched_barrier(0);
                            // J_ij += sum_{cx,cy} (hx hy)[i,j] * W_{cx,cy}[iz,jz],  W = sum_cz E[cx,cy,cz] hz[iz,jz,cz]:
                            // the z sum runs first, on the (LI+1)(LJ+1) z-power pairs of the bra only
#pragma unroll
                            for (int cx = 0; cx <= LKL; cx++)
#pragma unroll
                            for (int cy = 0; cy <= LKL - cx; cy++) {
                                constexpr int dummy = 0; (void)dummy;
                                const int z_lo = LK - cx - cy > 0 ? LK - cx - cy : 0, z_hi = LKL - cx - cy;
                                real wz[LI + 1][LJ + 1];
#pragma unroll
                                for (int a = 0; a <= LI; a++)
#pragma unroll
                                    for (int b = 0; b <= LJ; b++) wz[a][b] = 0;
#pragma unroll
                                for (int cz = z_lo; cz <= z_hi; cz++) {
                                    const real e = E[trip_index(cx, cy, cz)];
#pragma unroll
                                    for (int a = 0; a <= LI; a++)
#pragma unroll
                                        for (int b = 0; b <= LJ; b++) wz[a][b] += e * hz[a * HS_I + b * HS_J + cz];
                                }
#pragma unroll
                                for (int i = 0; i < NFI; i++)
#pragma unroll
                                for (int j = 0; j < NFJ; j++)
                                    J[i * NFJ + j] += (double)(hx[TI.x[i] * HS_I + TJ.x[j] * HS_J + cx] *
                                                               hy[TI.y[i] * HS_I + TJ.y[j] * HS_J + cy] * wz[TI.z[i]][TJ.z[j]]);
                            }
                        }
                    }
                }
            }
        }
    }
    if (have && nq) {
        const int i0 = (int)bi[3], j0 = (int)bj[3];
#pragma unroll
        for (int i = 0; i < NFI; i++)
#pragma unroll
            for (int j = 0; j < NFJ; j++) atomic_add_f64(vj + (size_t)(j0 + j) * nao + i0 + i, J[i * NFJ + j]);
    }
    if (counter) {
        unsigned long long t = nq;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if ((tid & 63) == 0 && t) atomicAdd(counter, t);
    }
}
