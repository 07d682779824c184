// Compile-time Cartesian index tables and the per-axis 1-D integral recurrences (TRR + HRR).
// Mathematics: reference /root/reference/jqc/backend/jk/1q1t.cu:250-382; organisation is this build's own.
#pragma once
#include "jk_common.h"

template <int L> struct CartTab { int x[nf_of(L)]; int y[nf_of(L)]; int z[nf_of(L)]; };
template <int L> constexpr CartTab<L> make_tab()
{
    CartTab<L> t{};
    for (int c = 0; c < nf_of(L); c++) {
        CartPow p = cart_pow(L, c);
        t.x[c] = p.x; t.y[c] = p.y; t.z[c] = p.z;
    }
    return t;
}
static constexpr CartTab<LI> TI = make_tab<LI>();
static constexpr CartTab<LJ> TJ = make_tab<LJ>();
static constexpr CartTab<LK> TK = make_tab<LK>();
static constexpr CartTab<LL> TL = make_tab<LL>();

constexpr int GS_L = 1;
constexpr int GS_K = LL + 1;
constexpr int GS_J = GS_K * (LK + 1);
constexpr int GS_I = GS_J * (LJ + 1);
constexpr int GSIZE = GS_I * (LI + 1);

#ifndef BLOCK
#define BLOCK 256
#endif
// classes whose integral block is too large to live in registers fall back to rolled loops
#if !defined(UNROLL_ALL)
#define UNROLL_ALL (NINT <= 1296)
#endif
#if UNROLL_ALL
#define UNROLL _Pragma("unroll")
#else
#define UNROLL _Pragma("nounroll")
#endif

// 1-D integrals of one axis for one root: TRR in (a,c), HRR into j, HRR into l.
// out[((i*(LJ+1)+j)*(LK+1)+k)*(LL+1)+l]
// (R: double / float, or a 2-vector of floats holding two quartets per lane: jk_tile.hip MIXED)
template <typename R = real>
__device__ __forceinline__ void axis_integrals(R g0, R c0, R cp, R b10, R b01, R b00,
                                               R rij, R rkl, R* __restrict__ out)
{
    typedef R real;
    real t[LIJ + 1][LKL + 1];
    t[0][0] = g0;
    if (LIJ > 0) {
        t[1][0] = c0 * g0;
#pragma unroll
        for (int a = 1; a < LIJ; a++) t[a + 1][0] = c0 * t[a][0] + float(a) * b10 * t[a - 1][0];
    }
#pragma unroll
    for (int c = 0; c < LKL; c++) {
#pragma unroll
        for (int a = 0; a <= LIJ; a++) {
            real v = cp * t[a][c];
            if (c > 0) v += float(c) * b01 * t[a][c - 1];
            if (a > 0) v += float(a) * b00 * t[a - 1][c];
            t[a][c + 1] = v;
        }
    }
    // HRR on the bra, in place over a: after step j, h[a][.] holds (a, j) for a <= LIJ - j
    real h[LJ + 1][LI + 1][LKL + 1];
    {
        real w[LIJ + 1][LKL + 1];
#pragma unroll
        for (int a = 0; a <= LIJ; a++)
#pragma unroll
            for (int c = 0; c <= LKL; c++) w[a][c] = t[a][c];
#pragma unroll
        for (int j = 0; j <= LJ; j++) {
#pragma unroll
            for (int i = 0; i <= LI; i++)
#pragma unroll
                for (int c = 0; c <= LKL; c++) h[j][i][c] = w[i][c];
            if (j < LJ) {
#pragma unroll
                for (int a = 0; a < LIJ - j; a++)
#pragma unroll
                    for (int c = 0; c <= LKL; c++) w[a][c] = w[a + 1][c] - rij * w[a][c];
            }
        }
    }
#pragma unroll
    for (int i = 0; i <= LI; i++)
#pragma unroll
        for (int j = 0; j <= LJ; j++) {
            real w[LKL + 1];
#pragma unroll
            for (int c = 0; c <= LKL; c++) w[c] = h[j][i][c];
#pragma unroll
            for (int l = 0; l <= LL; l++) {
#pragma unroll
                for (int k = 0; k <= LK; k++) out[i * GS_I + j * GS_J + k * GS_K + l] = w[k];
                if (l < LL) {
#pragma unroll
                    for (int c = 0; c < LKL - l; c++) w[c] = w[c + 1] - rkl * w[c];
                }
            }
        }
}

