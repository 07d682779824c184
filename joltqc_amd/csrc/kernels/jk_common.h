// Shared device code of the J/K kernels (gfx950 only).  Compiled through hiprtc or hipcc --genco with
//   -DLI= -DLJ= -DLK= -DLL=   angular momenta of the four shells of the class
//   -DDO_J= -DDO_K=           which matrices to build
//   -DRYS_LR=0|1              1: long-range erf(omega r)/r kernel (reference rys_type > 0)
//   -DFP32=0|1                arithmetic type of the integral evaluation (accumulation is always f64)
// Mathematics follows the reference's rys_1q1t_vjk (/root/reference/jqc/backend/jk/1q1t.cu:45-644) and
// rys_roots (jqc/backend/rys/rys_roots.cu:30-160); the code organisation (2-D TRR array, separate
// HRR stages with compile-time index maps, run-time primitive counts) is this build's own.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#ifndef FP32
#define FP32 0
#endif
#if FP32
typedef float real;
#else
typedef double real;
#endif

#define BASIS_STRIDE 12
#define NCOEF 14

constexpr int nf_of(int l) { return (l + 1) * (l + 2) / 2; }

constexpr int LIJ = LI + LJ;
constexpr int LKL = LK + LL;
constexpr int NROOTS = (LIJ + LKL) / 2 + 1;
constexpr int NFI = nf_of(LI), NFJ = nf_of(LJ), NFK = nf_of(LK), NFL = nf_of(LL);
constexpr int NINT = NFI * NFJ * NFK * NFL;

// Cartesian exponents of component c of a shell with angular momentum l (libcint order:
// lx descending, then ly descending; /root/reference/jqc/backend/util.py:21-36).
struct CartPow { int x, y, z; };
constexpr CartPow cart_pow(int l, int c)
{
    int n = 0;
    for (int lx = l; lx >= 0; lx--)
        for (int ly = l - lx; ly >= 0; ly--) {
            if (n == c) return CartPow{lx, ly, l - lx - ly};
            n++;
        }
    return CartPow{0, 0, 0};
}

// ---------------------------------------------------------------------------------------------
// Rys roots (t^2) and weights.  `tab` points at this class's tables inside the shared blob
// (layout: joltqc_amd/backend/rys.py): cheb[2n+14][n][14][2] followed by large[n][2].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rys_roots(real x, real theta, real omega, const real* __restrict__ cheb,
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
    if (x >= real(5 * NROOTS + 35)) {
        const real isx = rsqrt(x);
        const real ix = isx * isx;
#pragma unroll
        for (int i = 0; i < NROOTS; i++) {
            rw[2 * i] = large[2 * i] * ix * tf;
            rw[2 * i + 1] = large[2 * i + 1] * isx * stf;
        }
        return;
    }
    const int it = (int)(x * real(0.4));
    const real u = (x - real(2.5) * it) * real(0.8) - real(1);
    const real u2 = u + u;
    const real* __restrict__ c = cheb + it * (NROOTS * NCOEF * 2);
#pragma unroll
    for (int i = 0; i < NROOTS; i++, c += NCOEF * 2) {
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

// Fast reciprocal / reciprocal square root: hardware estimate + Newton steps (error < 2 ulp in f64,
// < 1 ulp in f32), 5-7 instructions instead of the ~12-instruction IEEE division sequence.
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = fma(r, fma(-h * r, r, 0.5), r);
    r = fma(r, fma(-h * r, r, 0.5), r);
    return r;
}
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

__device__ __forceinline__ void atomic_add_f64(double* p, double v)
{
    // lowers to global_atomic_add_f64 (no CAS loop) on gfx950.  ATOMIC_AGENT=1: device (agent) scope instead of the
    // system scope of unsafeAtomicAdd (no sc1 bit): the Fock matrix is read by nobody before the kernel boundary
#if defined(ATOMIC_AGENT) && ATOMIC_AGENT
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    unsafeAtomicAdd(p, v);
#endif
}
