// J/K kernel, tiled form for gfx950.  Entry point: jk_tile.
//
// One 256-thread workgroup owns one (bra shell-tile pair) x (ket shell-tile pair): up to
// TSI*TSJ*TSK*TSL shell quartets whose six Fock sub-blocks and six density sub-blocks live in LDS for
// the whole workgroup.  Global f64 atomics are issued once per tile element at the end (coalesced
// rows) instead of once per quartet element, which is what the chip-wide atomic rate of MI355X
// demands (MI355X_MICROARCH.md "Global float atomics").
//
// Inside the workgroup a quartet is evaluated by T = NFI*NFJ lanes ("row lanes", one per bra
// Cartesian pair (ci,cj)); G = 256/T quartets are in flight.  Per primitive combination:
//   phase A  3*NROOTS "job" lanes of each quartet compute one Rys root and run the transfer
//            recurrence (TRR) of one (root, axis) into LDS  t[root][axis][a<=LIJ][c<=LKL];
//   phase B  every row lane contracts t with its own bra horizontal-recurrence weights, runs the ket
//            horizontal recurrence in registers and accumulates its E = CW*NFL integrals
//            (compile-time indices only: no scratch, no LDS traffic in the inner product loop).
// The six contractions then go to the LDS Fock tiles; J_kl, K_jk, K_jl are first summed in
// registers across consecutive quartets that share the destination block.
//
// Mathematics (what is computed) follows the reference kernels
//   /root/reference/jqc/backend/jk/1q1t.cu:86-94 (symmetry factors), :174-242 (primitive prefactors,
//   seeds), :250-330 (TRR), :336-382 (HRR), :423-638 (six contractions and their destinations) and
//   /root/reference/jqc/backend/jk/screen_jk_tasks.cu:202-261 (per-quartet screening predicate);
// the decomposition above replaces the reference's jk/1qnt.cu design and is this build's own.
#include "jk_common.h"
#include "jk_axis.h"

constexpr int ts_of(int l) { return l <= 2 ? 4 : (l == 3 ? 2 : 1); }
constexpr int TSI = ts_of(LI), TSJ = ts_of(LJ), TSK = ts_of(LK), TSL = ts_of(LL);
constexpr int NQ = TSI * TSJ * TSK * TSL;
constexpr int T = NFI * NFJ;
constexpr int G = 256 / T;
#ifndef ECAP
#define ECAP 64
#endif
#ifndef TILE_1Q
#define TILE_1Q 0   // 1: one quartet per lane inside the tile (small classes); 0: T row lanes per quartet
#endif
#ifndef ABLATE
#define ABLATE 0    // timing-only builds (wrong results): 1 skip phase A, 2 skip phase B, 4 skip contraction, 8 no barriers
#endif
#ifndef RYS_LDS_MAX
#define RYS_LDS_MAX 28672   // stage the class's Chebyshev table in LDS when it is at most this many bytes (nroots <= 5 in f64)
#endif
#ifndef MINW
#define MINW (TILE_1Q ? 2 : 1)   // waves/SIMD the register allocator leaves room for (measured best: 2 for the
                                  // lane-per-quartet mode, 1 for the row-lane mode; profiles/r01_*)
#endif
constexpr int pick_nch()
{
    for (int n = 1; n <= NFK; n++)
        if (NFK % n == 0 && (NFK / n) * NFL <= ECAP) return n;
    return NFK;
}
constexpr int NCH = pick_nch();
constexpr int CW = NFK / NCH;
constexpr int E = CW * NFL;
constexpr int WI = TSI * NFI, WJ = TSJ * NFJ, WK = TSK * NFK, WL = TSL * NFL;
constexpr int NT2 = (LIJ + 1) * (LKL + 1);
constexpr int RYS_TAB = (2 * NROOTS + 14) * NROOTS * NCOEF * 2;          // Chebyshev table of this class, in reals
constexpr bool RYS_IN_LDS = RYS_TAB * (int)sizeof(real) <= RYS_LDS_MAX;
static_assert(T <= 256 && G >= 1 && NQ <= 256, "tile geometry");

// Rys root `r` only (same tables and branches as rys_roots in jk_common.h)
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

template <typename TT>
__device__ __forceinline__ void stage_tile(TT* __restrict__ dst, const real* __restrict__ dm, const int nao, const int r0,
                                           const int c0, const int NR, const int NC, const int tid)
{
    for (int idx = tid; idx < NR * NC; idx += 256) {
        const int r = idx / NC, c = idx - r * NC;
        dst[idx] = (r0 + r < nao && c0 + c < nao) ? dm[(size_t)(r0 + r) * nao + c0 + c] : real(0);
    }
}

__device__ __forceinline__ void flush_tile(const double* __restrict__ src, double* __restrict__ out, const int nao,
                                           const int r0, const int c0, const int NR, const int NC, const int tid)
{
    for (int idx = tid; idx < NR * NC; idx += 256) {
        const int r = idx / NC, c = idx - r * NC;
        const double v = src[idx];
        if (v != 0.0 && r0 + r < nao && c0 + c < nao) atomic_add_f64(out + (size_t)(r0 + r) * nao + c0 + c, v);
    }
}

__device__ __forceinline__ void lds_add(double* p, double v) { atomicAdd(p, v); }   // ds_add_f64

#ifndef KNAME
#define KNAME jk_tile
#endif
extern "C" __global__ void __launch_bounds__(256, MINW)
KNAME(const int nao, const real* __restrict__ basis, const real* __restrict__ dm, double* __restrict__ vj,
        double* __restrict__ vk, const real omega, const int* __restrict__ tasks, const int ntasks,
        const unsigned* __restrict__ tpair_sh, const float* __restrict__ tpair_q, const float* __restrict__ q_cond,
        const float* __restrict__ log_dm, const int nbas, const float cut_lo, const float cut_hi,
        const float log_max_dm, const int n_dm, const real* __restrict__ rys_cheb, const real* __restrict__ rys_large,
        unsigned long long* __restrict__ counter)
{
    __shared__ int s_task[8];
    __shared__ int s_nact;
    __shared__ unsigned s_wcnt[4];
    __shared__ unsigned short s_act[NQ];
    __shared__ real sDij[WJ * WI], sDkl[WL * WK], sDik[WI * WK], sDil[WI * WL], sDjk[WJ * WK], sDjl[WJ * WL];
    __shared__ double sJij[WJ * WI], sJkl[WL * WK], sKik[WI * WK], sKil[WI * WL], sKjk[WJ * WK], sKjl[WJ * WL];
#if !TILE_1Q
    __shared__ real sT[G * NROOTS * 3 * NT2];
#endif
    // shell rows of the four tiles and per-primitive-pair prefactors {c_a c_b K_ab, 1/(a+b), a+b}:
    // every exp / reciprocal of the pair prefactors is evaluated once per workgroup, not once per quartet
    __shared__ real sBas[(TSI + TSJ + TSK + TSL) * BASIS_STRIDE];
    __shared__ real sPB[TSI * TSJ * 9 * 3], sPK[TSK * TSL * 9 * 3];
    // Rys Chebyshev table of the class: every lane reads 28 coefficients of ITS OWN x-interval per root, i.e. 64
    // different cache lines per wave instruction from global memory; from LDS the same gather costs a few cycles
    __shared__ real sRys[RYS_IN_LDS ? RYS_TAB : 1];

    const int tid = threadIdx.x;
    if (tid == 0) {
        int lo = 0, hi = ntasks - 1;
        const int b = blockIdx.x;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tasks[mid * 8 + 5] <= b) lo = mid; else hi = mid - 1;
        }
        for (int n = 0; n < 8; n++) s_task[n] = tasks[lo * 8 + n];
    }
    __syncthreads();
    const int nkl = s_task[3];
    const int lb = blockIdx.x - s_task[5];
    const int bij = lb / nkl, bkl = lb - bij * nkl;
    const unsigned pij = tpair_sh[s_task[0] + bij], pkl = tpair_sh[s_task[2] + bkl];
    if (tpair_q[s_task[0] + bij] + tpair_q[s_task[2] + bkl] + log_max_dm <= cut_lo) return;
    const int ish0 = pij >> 16, jsh0 = pij & 0xffff, ksh0 = pkl >> 16, lsh0 = pkl & 0xffff;

    // ---- per-quartet screening inside the tile pair, compaction of the survivors (wave64 ballots)
    bool keep = false;
    if (tid < NQ) {
        const int a = tid % TSI, b = (tid / TSI) % TSJ, d = (tid / (TSI * TSJ)) % TSL, c = tid / (TSI * TSJ * TSL);
        const int ish = ish0 + a, jsh = jsh0 + b, ksh = ksh0 + c, lsh = lsh0 + d;
        if (ish >= jsh && ksh >= lsh && ish * nbas + jsh >= ksh * nbas + lsh) {
            const float q = q_cond[ish * nbas + jsh] + q_cond[ksh * nbas + lsh];
            float dl = -36.8f;
#if DO_K
            dl = fmaxf(dl, log_dm[ish * nbas + ksh]);
            dl = fmaxf(dl, log_dm[jsh * nbas + ksh]);
            dl = fmaxf(dl, log_dm[ish * nbas + lsh]);
            dl = fmaxf(dl, log_dm[jsh * nbas + lsh]);
#endif
#if DO_J
            dl = fmaxf(dl, log_dm[ish * nbas + jsh]);
            dl = fmaxf(dl, log_dm[ksh * nbas + lsh]);
#endif
            const float dq = q + dl;
            keep = dq > cut_lo && dq <= cut_hi;
        }
    }
    {
        const int wave = tid >> 6, lane = tid & 63;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        unsigned off = 0;
        for (int x = 0; x < wave; x++) off += s_wcnt[x];
        if (keep) s_act[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tid;
        if (tid == 0) s_nact = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int nact = s_nact;
    if (nact == 0) return;
    if (tid == 0 && counter) atomicAdd(counter + s_task[6], (unsigned long long)nact);   // per task row (slot 6)

    const real* __restrict__ bi0 = basis + ish0 * BASIS_STRIDE;
    const real* __restrict__ bj0 = basis + jsh0 * BASIS_STRIDE;
    const real* __restrict__ bk0 = basis + ksh0 * BASIS_STRIDE;
    const real* __restrict__ bl0 = basis + lsh0 * BASIS_STRIDE;
    const int npi = (int)bi0[10], npj = (int)bj0[10], npk = (int)bk0[10], npl = (int)bl0[10];
    const int i0 = (int)bi0[3], j0 = (int)bj0[3], k0 = (int)bk0[3], l0 = (int)bl0[3];
    constexpr int OFF_J = TSI * BASIS_STRIDE, OFF_K = (TSI + TSJ) * BASIS_STRIDE, OFF_L = (TSI + TSJ + TSK) * BASIS_STRIDE;
    for (int n = tid; n < (TSI + TSJ + TSK + TSL) * BASIS_STRIDE; n += 256) {
        const int sl = n / BASIS_STRIDE, w = n - sl * BASIS_STRIDE;
        const int sh = sl < TSI ? ish0 + sl : sl < TSI + TSJ ? jsh0 + sl - TSI
                     : sl < TSI + TSJ + TSK ? ksh0 + sl - TSI - TSJ : lsh0 + sl - TSI - TSJ - TSK;
        sBas[n] = basis[sh * BASIS_STRIDE + w];
    }
    if (RYS_IN_LDS)
        for (int n = tid; n < RYS_TAB; n += 256) sRys[n] = rys_cheb[n];
    const real* cheb_tab = RYS_IN_LDS ? sRys : rys_cheb;
    __syncthreads();
    for (int n = tid; n < (TSI * TSJ + TSK * TSL) * 9; n += 256) {
        const bool bra = n < TSI * TSJ * 9;
        const int m = bra ? n : n - TSI * TSJ * 9;
        const int pr = m / 9, pp = m - pr * 9, p1 = pp / 3, p2 = pp - p1 * 3;
        const real* s1 = bra ? sBas + (pr / TSJ) * BASIS_STRIDE : sBas + OFF_K + (pr / TSL) * BASIS_STRIDE;
        const real* s2 = bra ? sBas + OFF_J + (pr % TSJ) * BASIS_STRIDE : sBas + OFF_L + (pr % TSL) * BASIS_STRIDE;
        const real dx = s2[0] - s1[0], dy = s2[1] - s1[1], dz = s2[2] - s1[2];
        const real a1 = s1[5 + 2 * p1], a2 = s2[5 + 2 * p2];
        const real asum = a1 + a2, inv = fast_rcp(asum);
        const real val = s1[4 + 2 * p1] * s2[4 + 2 * p2] * exp(-a1 * a2 * inv * (dx * dx + dy * dy + dz * dz));
        real* dst = (bra ? sPB : sPK) + m * 3;
        dst[0] = val; dst[1] = inv; dst[2] = asum;
    }
    __syncthreads();

#if !TILE_1Q
    const int slot = tid / T, t = tid - slot * T;
    const bool lane_on = slot < G;
    const int per = (nact + G - 1) / G;
    const int ci = t / NFJ, cj = t - ci * NFJ;
    const int ibra[3] = {TI.x[ci], TI.y[ci], TI.z[ci]};
    const int jbra[3] = {TJ.x[cj], TJ.y[cj], TJ.z[cj]};
    real* __restrict__ myT = sT + slot * (NROOTS * 3 * NT2);
    const size_t nao2 = (size_t)nao * nao;

#else
    const size_t nao2 = (size_t)nao * nao;
#endif
    for (int idm = 0; idm < n_dm; idm++) {
        const real* __restrict__ D = dm + idm * nao2;
        __syncthreads();
        // ---- stage the six density sub-blocks, clear the six Fock sub-blocks
#if DO_J
        stage_tile(sDij, D, nao, j0, i0, WJ, WI, tid);
        stage_tile(sDkl, D, nao, l0, k0, WL, WK, tid);
        for (int n = tid; n < WJ * WI; n += 256) sJij[n] = 0;
        for (int n = tid; n < WL * WK; n += 256) sJkl[n] = 0;
#endif
#if DO_K
        stage_tile(sDik, D, nao, i0, k0, WI, WK, tid);
        stage_tile(sDil, D, nao, i0, l0, WI, WL, tid);
        stage_tile(sDjk, D, nao, j0, k0, WJ, WK, tid);
        stage_tile(sDjl, D, nao, j0, l0, WJ, WL, tid);
        for (int n = tid; n < WI * WK; n += 256) sKik[n] = 0;
        for (int n = tid; n < WI * WL; n += 256) sKil[n] = 0;
        for (int n = tid; n < WJ * WK; n += 256) sKjk[n] = 0;
        for (int n = tid; n < WJ * WL; n += 256) sKjl[n] = 0;
#endif
        __syncthreads();

#if TILE_1Q
        // ---------------- one quartet per lane: everything in registers, then LDS Fock tiles
        if (tid < nact) {
            const int qd = s_act[tid];
            const int a = qd % TSI, b = (qd / TSI) % TSJ, d = (qd / (TSI * TSJ)) % TSL, c = qd / (TSI * TSJ * TSL);
            const int ish = ish0 + a, jsh = jsh0 + b, ksh = ksh0 + c, lsh = lsh0 + d;
            const real* bi = sBas + a * BASIS_STRIDE;
            const real* bj = sBas + OFF_J + b * BASIS_STRIDE;
            const real* bk = sBas + OFF_K + c * BASIS_STRIDE;
            const real* bl = sBas + OFF_L + d * BASIS_STRIDE;
            const real* pb = sPB + (a * TSJ + b) * 27;
            const real* pk = sPK + (c * TSL + d) * 27;
            const real rix = bi[0], riy = bi[1], riz = bi[2];
            const real rkx = bk[0], rky = bk[1], rkz = bk[2];
            const real rij[3] = {bj[0] - rix, bj[1] - riy, bj[2] - riz};
            const real rkl[3] = {bl[0] - rkx, bl[1] - rky, bl[2] - rkz};
            real fac = real(34.98683665524972497);
            if (ish == jsh) fac *= real(0.5);
            if (ksh == lsh) fac *= real(0.5);
            if (ish == ksh && jsh == lsh) fac *= real(0.5);
            real I[NINT];
#pragma unroll
            for (int n = 0; n < NINT; n++) I[n] = 0;
            for (int kp = 0; kp < npk; kp++)
            for (int lp = 0; lp < npl; lp++) {
                const real ckcl = pk[(kp * 3 + lp) * 3], inv_akl = pk[(kp * 3 + lp) * 3 + 1], akl = pk[(kp * 3 + lp) * 3 + 2];
                const real al_akl = bl[5 + 2 * lp] * inv_akl;
                for (int ip = 0; ip < npi; ip++)
                for (int jp = 0; jp < npj; jp++) {
                    const real inv_aij = pb[(ip * 3 + jp) * 3 + 1], aij = pb[(ip * 3 + jp) * 3 + 2];
                    const real aj_aij = bj[5 + 2 * jp] * inv_aij;
                    const real cicj = fac * pb[(ip * 3 + jp) * 3];
                    const real rpa[3] = {rij[0] * aj_aij, rij[1] * aj_aij, rij[2] * aj_aij};
                    const real rqc[3] = {rkl[0] * al_akl, rkl[1] * al_akl, rkl[2] * al_akl};
                    const real rpq[3] = {rpa[0] + rix - rqc[0] - rkx, rpa[1] + riy - rqc[1] - rky,
                                         rpa[2] + riz - rqc[2] - rkz};
                    const real rr = rpq[0] * rpq[0] + rpq[1] * rpq[1] + rpq[2] * rpq[2];
                    const real sinv = fast_rsqrt(aij + akl);
                    const real inv = sinv * sinv;
                    const real theta = aij * akl * inv;
                    const real gy0 = cicj * inv_aij * inv_akl * sinv;
#pragma clang loop unroll(disable)
                    for (int ir = 0; ir < NROOTS; ir++) {
                        // one root at a time: evaluating all roots at once keeps 28 table coefficients per root live
                        real t2, wt;
                        rys_root_one(rr, theta, omega, ir, cheb_tab, rys_large, t2, wt);
                        const real rt_aa = t2 * inv;
                        const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                        const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                        const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                        const real b00 = real(0.5) * rt_aa;
                        real gx[GSIZE], gy[GSIZE], gz[GSIZE];
                        axis_integrals(ckcl, rpa[0] - rt_aij * rpq[0], rqc[0] + rt_akl * rpq[0], b10, b01, b00, rij[0], rkl[0], gx);
                        axis_integrals(gy0, rpa[1] - rt_aij * rpq[1], rqc[1] + rt_akl * rpq[1], b10, b01, b00, rij[1], rkl[1], gy);
                        axis_integrals(wt, rpa[2] - rt_aij * rpq[2], rqc[2] + rt_akl * rpq[2], b10, b01, b00, rij[2], rkl[2], gz);
#pragma unroll
                        for (int i = 0; i < NFI; i++)
#pragma unroll
                        for (int j = 0; j < NFJ; j++)
#pragma unroll
                        for (int k = 0; k < NFK; k++)
#pragma unroll
                        for (int l = 0; l < NFL; l++) {
                            const int ax = TI.x[i] * GS_I + TJ.x[j] * GS_J + TK.x[k] * GS_K + TL.x[l];
                            const int ay = TI.y[i] * GS_I + TJ.y[j] * GS_J + TK.y[k] * GS_K + TL.y[l];
                            const int az = TI.z[i] * GS_I + TJ.z[j] * GS_J + TK.z[k] * GS_K + TL.z[l];
                            I[((i * NFJ + j) * NFK + k) * NFL + l] += gx[ax] * gy[ay] * gz[az];
                        }
                    }
                }
            }
            const int iA = a * NFI, jA = b * NFJ, kA = c * NFK, lA = d * NFL;
            if (ABLATE & 4) {
                real s = 0;
                for (int n = 0; n < NINT; n++) s += I[n];
                if (s == real(1.2345)) lds_add(&sJij[0], (double)s);
            } else {
#if DO_J
            {
                real jkl[NFK * NFL], dkl[NFK * NFL];
#pragma unroll
                for (int k = 0; k < NFK; k++)
#pragma unroll
                    for (int l = 0; l < NFL; l++) { jkl[k * NFL + l] = 0; dkl[k * NFL + l] = sDkl[(lA + l) * WK + kA + k]; }
#pragma unroll
                for (int i = 0; i < NFI; i++)
#pragma unroll
                    for (int j = 0; j < NFJ; j++) {
                        const real dij = sDij[(jA + j) * WI + iA + i];
                        real s = 0;
#pragma unroll
                        for (int n = 0; n < NFK * NFL; n++) {
                            const real v = I[(i * NFJ + j) * NFK * NFL + n];
                            s += v * dkl[n];
                            jkl[n] += v * dij;
                        }
                        lds_add(&sJij[(jA + j) * WI + iA + i], (double)s);
                    }
#pragma unroll
                for (int k = 0; k < NFK; k++)
#pragma unroll
                    for (int l = 0; l < NFL; l++) lds_add(&sJkl[(lA + l) * WK + kA + k], (double)jkl[k * NFL + l]);
            }
#endif
#if DO_K
            {
                real kjk[NFJ * NFK], kjl[NFJ * NFL], djk[NFJ * NFK], djl[NFJ * NFL];
#pragma unroll
                for (int j = 0; j < NFJ; j++) {
#pragma unroll
                    for (int k = 0; k < NFK; k++) { kjk[j * NFK + k] = 0; djk[j * NFK + k] = sDjk[(jA + j) * WK + kA + k]; }
#pragma unroll
                    for (int l = 0; l < NFL; l++) { kjl[j * NFL + l] = 0; djl[j * NFL + l] = sDjl[(jA + j) * WL + lA + l]; }
                }
#pragma unroll
                for (int i = 0; i < NFI; i++) {
                    real kik[NFK], kil[NFL], dik[NFK], dil[NFL];
#pragma unroll
                    for (int k = 0; k < NFK; k++) { kik[k] = 0; dik[k] = sDik[(iA + i) * WK + kA + k]; }
#pragma unroll
                    for (int l = 0; l < NFL; l++) { kil[l] = 0; dil[l] = sDil[(iA + i) * WL + lA + l]; }
#pragma unroll
                    for (int j = 0; j < NFJ; j++)
#pragma unroll
                        for (int k = 0; k < NFK; k++)
#pragma unroll
                            for (int l = 0; l < NFL; l++) {
                                const real v = I[((i * NFJ + j) * NFK + k) * NFL + l];
                                kik[k] += v * djl[j * NFL + l];
                                kil[l] += v * djk[j * NFK + k];
                                kjk[j * NFK + k] += v * dil[l];
                                kjl[j * NFL + l] += v * dik[k];
                            }
#pragma unroll
                    for (int k = 0; k < NFK; k++) lds_add(&sKik[(iA + i) * WK + kA + k], (double)kik[k]);
#pragma unroll
                    for (int l = 0; l < NFL; l++) lds_add(&sKil[(iA + i) * WL + lA + l], (double)kil[l]);
                }
#pragma unroll
                for (int j = 0; j < NFJ; j++) {
#pragma unroll
                    for (int k = 0; k < NFK; k++) lds_add(&sKjk[(jA + j) * WK + kA + k], (double)kjk[j * NFK + k]);
#pragma unroll
                    for (int l = 0; l < NFL; l++) lds_add(&sKjl[(jA + j) * WL + lA + l], (double)kjl[j * NFL + l]);
                }
            }
#endif
            }
        }
#else
#pragma unroll
        for (int CH = 0; CH < NCH; CH++) {
            // register accumulators carried across consecutive quartets of this lane
            double jkl_acc[E], kjk_acc[CW], kjl_acc[NFL];
#pragma unroll
            for (int e = 0; e < E; e++) jkl_acc[e] = 0;
#pragma unroll
            for (int n = 0; n < CW; n++) kjk_acc[n] = 0;
#pragma unroll
            for (int n = 0; n < NFL; n++) kjl_acc[n] = 0;
            int key_kl = -1, key_jk = -1, key_jl = -1;      // local ids of the block the accumulators belong to
            int pjA = 0;                                     // jA of the accumulators (bra side of K_jk/K_jl)

            for (int step = 0; step < per; step++) {
                const int qi = slot * per + step;
                const bool on = lane_on && qi < nact;
                int a = 0, b = 0, c = 0, d = 0;
                if (on) {
                    const int qd = s_act[qi];
                    a = qd % TSI; b = (qd / TSI) % TSJ; d = (qd / (TSI * TSJ)) % TSL; c = qd / (TSI * TSJ * TSL);
                }
                const int ish = ish0 + a, jsh = jsh0 + b, ksh = ksh0 + c, lsh = lsh0 + d;
                const real* bi = sBas + a * BASIS_STRIDE;
                const real* bj = sBas + OFF_J + b * BASIS_STRIDE;
                const real* bk = sBas + OFF_K + c * BASIS_STRIDE;
                const real* bl = sBas + OFF_L + d * BASIS_STRIDE;
                const real* pb = sPB + (a * TSJ + b) * 27;
                const real* pk = sPK + (c * TSL + d) * 27;
                const real rix = bi[0], riy = bi[1], riz = bi[2];
                const real rkx = bk[0], rky = bk[1], rkz = bk[2];
                const real rij[3] = {bj[0] - rix, bj[1] - riy, bj[2] - riz};
                const real rkl[3] = {bl[0] - rkx, bl[1] - rky, bl[2] - rkz};
                real fac = real(34.98683665524972497);
                if (ish == jsh) fac *= real(0.5);
                if (ksh == lsh) fac *= real(0.5);
                if (ish == ksh && jsh == lsh) fac *= real(0.5);

                // bra HRR as a weighted sum over TRR rows: g(i,j) = sum_m C(j,m) (Ri-Rj)^(j-m) t[i+m]
                real wb[3][LJ + 1];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const int ja = jbra[ax];
                    const real ab = -rij[ax];
                    real pw = 1;      // ab^(ja-m), built downwards from m = ja
                    int binom = 1;    // C(ja, m)
#pragma unroll
                    for (int m = LJ; m >= 0; m--) {
                        if (m > ja) { wb[ax][m] = 0; continue; }
                        wb[ax][m] = pw * binom;
                        pw *= ab;
                        binom = binom * m / (ja - m + 1);
                    }
                }

                real acc[E];
#pragma unroll
                for (int e = 0; e < E; e++) acc[e] = 0;

                for (int kp = 0; kp < npk; kp++)
                for (int lp = 0; lp < npl; lp++)
                for (int ip = 0; ip < npi; ip++)
                for (int jp = 0; jp < npj; jp++) {
                    // ---------------- phase A: job lanes: one root + one axis TRR each
                    if (on && !(ABLATE & 1)) {
                        for (int job = t; job < 3 * NROOTS; job += T) {
                            const int r = job / 3, ax = job - r * 3;
                            const real ckcl = pk[(kp * 3 + lp) * 3], inv_akl = pk[(kp * 3 + lp) * 3 + 1], akl = pk[(kp * 3 + lp) * 3 + 2];
                            const real cicj = pb[(ip * 3 + jp) * 3], inv_aij = pb[(ip * 3 + jp) * 3 + 1], aij = pb[(ip * 3 + jp) * 3 + 2];
                            const real al_akl = bl[5 + 2 * lp] * inv_akl, aj_aij = bj[5 + 2 * jp] * inv_aij;
                            const real rpq0 = rij[0] * aj_aij + rix - rkl[0] * al_akl - rkx;
                            const real rpq1 = rij[1] * aj_aij + riy - rkl[1] * al_akl - rky;
                            const real rpq2 = rij[2] * aj_aij + riz - rkl[2] * al_akl - rkz;
                            const real rr = rpq0 * rpq0 + rpq1 * rpq1 + rpq2 * rpq2;
                            const real sinv = fast_rsqrt(aij + akl);
                            const real inv = sinv * sinv;
                            const real theta = aij * akl * inv;
                            real t2, wt;
                            if (ABLATE & 16) { t2 = real(0.3) + real(0.01) * r; wt = real(0.5); }
                            else rys_root_one(rr, theta, omega, r, cheb_tab, rys_large, t2, wt);
                            const real rt_aa = t2 * inv;
                            const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                            const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                            const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                            const real b00 = real(0.5) * rt_aa;
                            const real rij_a = ax == 0 ? rij[0] : ax == 1 ? rij[1] : rij[2];
                            const real rkl_a = ax == 0 ? rkl[0] : ax == 1 ? rkl[1] : rkl[2];
                            const real rpq_a = ax == 0 ? rpq0 : ax == 1 ? rpq1 : rpq2;
                            const real c0 = rij_a * aj_aij - rt_aij * rpq_a;
                            const real cp = rkl_a * al_akl + rt_akl * rpq_a;
                            real g0;
                            if (ax == 0) g0 = ckcl;
                            else if (ax == 1) g0 = fac * cicj * inv_aij * inv_akl * sinv;
                            else g0 = wt;
                            real tt[LIJ + 1][LKL + 1];
                            tt[0][0] = g0;
                            if (ABLATE & 32) {
#pragma unroll
                                for (int q = 0; q <= LIJ; q++)
#pragma unroll
                                    for (int cc = 0; cc <= LKL; cc++) tt[q][cc] = g0 + c0 * q + cp * cc;
                            } else {
                            if (LIJ > 0) {
                                tt[1][0] = c0 * g0;
#pragma unroll
                                for (int q = 1; q < LIJ; q++) tt[q + 1][0] = c0 * tt[q][0] + q * b10 * tt[q - 1][0];
                            }
#pragma unroll
                            for (int cc = 0; cc < LKL; cc++) {
#pragma unroll
                                for (int q = 0; q <= LIJ; q++) {
                                    real v = cp * tt[q][cc];
                                    if (cc > 0) v += cc * b01 * tt[q][cc - 1];
                                    if (q > 0) v += q * b00 * tt[q - 1][cc];
                                    tt[q][cc + 1] = v;
                                }
                            }
                            }
                            real* __restrict__ dst = myT + (r * 3 + ax) * NT2;
#pragma unroll
                            for (int q = 0; q <= LIJ; q++)
#pragma unroll
                                for (int cc = 0; cc <= LKL; cc++) dst[q * (LKL + 1) + cc] = tt[q][cc];
                        }
                    }
                    if (!(ABLATE & 8)) __syncthreads();
                    // ---------------- phase B: row lanes: own bra slice, ket HRR, integral accumulation
                    if (on && !(ABLATE & 2)) {
                        for (int r = 0; r < NROOTS; r++) {
                            real gk[3][LK + 1][LL + 1];
#pragma unroll
                            for (int ax = 0; ax < 3; ax++) {
                                const real* __restrict__ tp = myT + (r * 3 + ax) * NT2 + ibra[ax] * (LKL + 1);
                                real w[LKL + 1];
#pragma unroll
                                for (int cc = 0; cc <= LKL; cc++) w[cc] = wb[ax][0] * tp[cc];
#pragma unroll
                                for (int m = 1; m <= LJ; m++)
#pragma unroll
                                    for (int cc = 0; cc <= LKL; cc++) w[cc] += wb[ax][m] * tp[m * (LKL + 1) + cc];
#pragma unroll
                                for (int l = 0; l <= LL; l++) {
#pragma unroll
                                    for (int k = 0; k <= LK; k++) gk[ax][k][l] = w[k];
                                    if (l < LL) {
#pragma unroll
                                        for (int cc = 0; cc < LKL - l; cc++) w[cc] = w[cc + 1] - rkl[ax] * w[cc];
                                    }
                                }
                            }
#pragma unroll
                            for (int kk = 0; kk < CW; kk++)
#pragma unroll
                                for (int cl = 0; cl < NFL; cl++) {
                                    const int ck = CH * CW + kk;
                                    acc[kk * NFL + cl] += gk[0][TK.x[ck]][TL.x[cl]] * gk[1][TK.y[ck]][TL.y[cl]] *
                                                          gk[2][TK.z[ck]][TL.z[cl]];
                                }
                        }
                    }
                    if (!(ABLATE & 8)) __syncthreads();
                }
                if (ABLATE & 4) {
                    if (on) { real s = 0; for (int e = 0; e < E; e++) s += acc[e]; if (s == real(1.2345)) lds_add(&sJij[0], (double)s); }
                    continue;
                }

                // ---------------- contraction with the density sub-blocks, accumulation in the LDS Fock tiles
                const int iA = a * NFI + ci, jA = b * NFJ + cj;
                const int kb = c * NFK + CH * CW, lbs = d * NFL;
                const int nk_kl = on ? c * TSL + d : -1, nk_jk = on ? b * TSK + c : -1, nk_jl = on ? b * TSL + d : -1;
#if DO_J
                if (key_kl >= 0 && key_kl != nk_kl) {
                    const int pk = (key_kl / TSL) * NFK + CH * CW, pl = (key_kl % TSL) * NFL;
#pragma unroll
                    for (int kk = 0; kk < CW; kk++)
#pragma unroll
                        for (int cl = 0; cl < NFL; cl++) {
                            lds_add(&sJkl[(pl + cl) * WK + pk + kk], jkl_acc[kk * NFL + cl]);
                            jkl_acc[kk * NFL + cl] = 0;
                        }
                }
                key_kl = nk_kl;
                if (on) {
                    const real dij = sDij[jA * WI + iA];
                    real s = 0;
#pragma unroll
                    for (int kk = 0; kk < CW; kk++)
#pragma unroll
                        for (int cl = 0; cl < NFL; cl++) {
                            const real v = acc[kk * NFL + cl];
                            s += v * sDkl[(lbs + cl) * WK + kb + kk];
                            jkl_acc[kk * NFL + cl] += (double)(v * dij);
                        }
                    lds_add(&sJij[jA * WI + iA], (double)s);
                }
#endif
#if DO_K
                if (key_jk >= 0 && key_jk != nk_jk) {
                    const int pk = (key_jk % TSK) * NFK + CH * CW;
#pragma unroll
                    for (int kk = 0; kk < CW; kk++) { lds_add(&sKjk[pjA * WK + pk + kk], kjk_acc[kk]); kjk_acc[kk] = 0; }
                }
                if (key_jl >= 0 && key_jl != nk_jl) {
                    const int pl = (key_jl % TSL) * NFL;
#pragma unroll
                    for (int cl = 0; cl < NFL; cl++) { lds_add(&sKjl[pjA * WL + pl + cl], kjl_acc[cl]); kjl_acc[cl] = 0; }
                }
                key_jk = nk_jk;
                key_jl = nk_jl;
                pjA = jA;
                if (on) {
                    real kil[NFL];
#pragma unroll
                    for (int cl = 0; cl < NFL; cl++) kil[cl] = 0;
#pragma unroll
                    for (int kk = 0; kk < CW; kk++) {
                        real s_ik = 0, s_jk = 0;
                        const real djk = sDjk[jA * WK + kb + kk], dik = sDik[iA * WK + kb + kk];
#pragma unroll
                        for (int cl = 0; cl < NFL; cl++) {
                            const real v = acc[kk * NFL + cl];
                            s_ik += v * sDjl[jA * WL + lbs + cl];
                            s_jk += v * sDil[iA * WL + lbs + cl];
                            kil[cl] += v * djk;
                            kjl_acc[cl] += (double)(v * dik);
                        }
                        lds_add(&sKik[iA * WK + kb + kk], (double)s_ik);
                        kjk_acc[kk] += (double)s_jk;
                    }
#pragma unroll
                    for (int cl = 0; cl < NFL; cl++) lds_add(&sKil[iA * WL + lbs + cl], (double)kil[cl]);
                }
#endif
            }
            // final flush of the carried accumulators
#if DO_J
            if (key_kl >= 0) {
                const int pk = (key_kl / TSL) * NFK + CH * CW, pl = (key_kl % TSL) * NFL;
#pragma unroll
                for (int kk = 0; kk < CW; kk++)
#pragma unroll
                    for (int cl = 0; cl < NFL; cl++) lds_add(&sJkl[(pl + cl) * WK + pk + kk], jkl_acc[kk * NFL + cl]);
            }
#endif
#if DO_K
            if (key_jk >= 0) {
                const int pk = (key_jk % TSK) * NFK + CH * CW;
#pragma unroll
                for (int kk = 0; kk < CW; kk++) lds_add(&sKjk[pjA * WK + pk + kk], kjk_acc[kk]);
            }
            if (key_jl >= 0) {
                const int pl = (key_jl % TSL) * NFL;
#pragma unroll
                for (int cl = 0; cl < NFL; cl++) lds_add(&sKjl[pjA * WL + pl + cl], kjl_acc[cl]);
            }
#endif
        }
#endif  // TILE_1Q
        __syncthreads();
        // ---- one coalesced pass of global f64 atomics per Fock sub-block
#if DO_J
        {
            double* __restrict__ J = vj + idm * nao2;
            flush_tile(sJij, J, nao, j0, i0, WJ, WI, tid);
            flush_tile(sJkl, J, nao, l0, k0, WL, WK, tid);
        }
#endif
#if DO_K
        {
            double* __restrict__ K = vk + idm * nao2;
            flush_tile(sKik, K, nao, i0, k0, WI, WK, tid);
            flush_tile(sKil, K, nao, i0, l0, WI, WL, tid);
            flush_tile(sKjk, K, nao, j0, k0, WJ, WK, tid);
            flush_tile(sKjl, K, nao, j0, l0, WJ, WL, tid);
        }
#endif
    }
}
