// J/K kernel, tiled form for gfx950.  Entry point: jk_tile.
//
// One 256-thread workgroup owns ONE bra shell-tile pair and walks a chunk of consecutive ket shell-tile
// pairs of the Schwarz-sorted ket list (task row = rectangle of the two lists).  Per (bra pair, ket pair)
// up to TSI*TSJ*TSK*TSL shell quartets are evaluated; their six density sub-blocks and six Fock
// sub-blocks live in LDS.  Bra-side data (shell rows, primitive-pair prefactors, D_ij, the Rys table and
// the J_ij accumulator) is staged once per workgroup, ket-side data once per ket pair.  Global f64
// atomics are issued once per tile element (coalesced rows) instead of once per quartet element, which
// is what the chip-wide atomic rate of MI355X demands (MI355X_MICROARCH.md "Global float atomics").
//
// Two compute modes (compile time):
//  TILE_1Q=1  one quartet per lane, all integrals of the quartet in registers (classes up to ~180 integrals).
//             NKS = 2/4/8 stages and screens that many ket pairs per iteration, so the survivor queue fills the 256
//             lanes where a tile pair has few candidates; with NKS >= 4 LDS allows one workgroup per CU and the
//             compiler allocates up to 512 registers per lane (no scratch) -- the fastest form for the d/f classes;
//  TILE_1Q=0  a quartet is evaluated by T = NFI*NFJ "row lanes" (one per bra Cartesian pair (ci,cj)),
//             G = 256/T quartets in flight.  Per primitive combination
//     phase A  G*3*NROOTS "job" lanes (packed into the first waves of the workgroup) compute one Rys
//              root and run the transfer recurrence (TRR) of one (root, axis) into LDS
//              t[root][axis][a<=LIJ][c<=LKL] (the double-buffered schedule that overlapped phase A of the
//              next combination with phase B is switched off: TRR_DOUBLE_BUFFER);
//     phase B  every row lane contracts t with its own bra horizontal-recurrence weights, runs the ket
//              horizontal recurrence in registers and accumulates its E = CW*NFL integrals
//              (compile-time indices only: no scratch, no LDS traffic in the inner product loop).
//   The six contractions then go to the LDS Fock tiles; J_kl, K_jk, K_jl are first summed in registers
//   across consecutive quartets that share the destination block.
//   HB=1 ("h form", round 6): phase A also runs the bra horizontal recurrence, once per (quartet, root, axis), and leaves
//   h[root][axis][i][j][c] in LDS; a phase-B lane = (bra component i, group of j components) reads one row per (axis,
//   j component), runs only the ket recurrence and multiplies -- the other row-lane forms redo the bra recurrence in every
//   lane, root and integral chunk (hardware / model flop 1.7-2.6 in the round-5 counters).
//
// Mathematics (what is computed) follows the reference kernels
//   /root/reference/jqc/backend/jk/1q1t.cu:86-94 (symmetry factors), :174-242 (primitive prefactors,
//   seeds), :250-330 (TRR), :336-382 (HRR), :423-638 (six contractions and their destinations) and
//   /root/reference/jqc/backend/jk/screen_jk_tasks.cu:202-261 (per-quartet screening predicate);
// the decomposition above replaces the reference's jk/1qnt.cu design and is this build's own.
#include "jk_common.h"
#include "jk_axis.h"

// shells per tile edge, by angular momentum (host side: joltqc_amd/constants.py tile_width; the library passes -DTSWn)
#ifndef TSW0
#define TSW0 8
#endif
#ifndef TSW1
#define TSW1 4
#endif
#ifndef TSW2
#define TSW2 4
#endif
#ifndef TSW3
#define TSW3 2
#endif
#ifndef TSW4
#define TSW4 1
#endif
constexpr int ts_of(int l) { return l == 0 ? TSW0 : l == 1 ? TSW1 : l == 2 ? TSW2 : l == 3 ? TSW3 : TSW4; }
constexpr int TSI = ts_of(LI), TSJ = ts_of(LJ), TSK = ts_of(LK), TSL = ts_of(LL);
constexpr int NQ = TSI * TSJ * TSK * TSL;
#ifndef CJR
#define CJR 0       // row-lane mode with the bra j components in REGISTERS: lane = bra component ci only (T = nf_i lanes per
                    // quartet), each lane runs the bra HRR for every cj and holds nf_j times more integrals.  For classes with a
                    // small ket block (few integrals per (ci,cj) lane) the LDS reads of phase B are shared by nf_j times more
                    // products and K_ik / K_il / J_ij need no cross-lane sum at all.
#endif
#ifdef CTWO
#define CTWO_ CTWO
#else
#define CTWO_ 0
#endif
#ifdef CORD
#define CORD_ CORD
#else
#define CORD_ 0
#endif
#ifdef STAGE_ALL
#define STAGE_ALL_ STAGE_ALL
#else
#define STAGE_ALL_ 0
#endif
#ifndef SKIP_EMPTY
#define SKIP_EMPTY 1 // a ket tile pair whose smallest shell-pair index lies above the largest one of the bra tile pair holds no canonical
                    // quartet ((ij) >= (kl), reference screen_jk_tasks.cu:202-239): it is neither staged nor screened.  In the classes
                    // whose bra and ket lists coincide ((ps|ps), (dp|dp), ...) 44 % of the iterations are of that kind
                    // (profiles/r02_survivor_statistics.txt).
#endif
#ifndef MIXED
#define MIXED 0     // lane-per-quartet mode, FP64 build: 1 = mixed precision inside ONE launch.  A tile pair is staged and screened once;
                    // quartets whose estimate lies above cut_hi are evaluated in FP64 (one per lane, the code below), those in
                    // (cut_lo, cut_hi] in FP32 with TWO quartets per lane held as 2-vectors: the recurrences, the integral products
                    // and the contractions are v_pk_mul / v_pk_add / v_pk_fma_f32, i.e. twice the FP64 rate in the register
                    // footprint of one FP64 quartet (scalar FP32 VALU runs at the FP64 rate on gfx950).  Accumulation into the LDS
                    // Fock tiles stays FP64 (reference: outputs are always f64, jk/1q1t.cu:49-50,455; the split into an fp32 and
                    // an fp64 kernel launch per class is jqc/pyscf/jk.py:293-328 with the estimate of jk/screen_jk_tasks.cu:241-261).
#endif
#ifndef NDM
#define NDM 1       // density matrices contracted against ONE evaluation of the integrals (1 or 2): the kernel walks the n_dm
                    // matrices of a call in groups of NDM; D and Fock tiles of a group live in LDS side by side.  Reference:
                    // every density matrix is contracted against the same integral block (jk/1q1t.cu:423-638, 1qnt.cu:488).
#endif
#ifndef QUAD
#define QUAD 0      // lane-per-quartet mode for classes with a p shell (index X = the last p among l, k, j, i): a quartet is worked on by
                    // ONE QUAD of lanes (DPP quad = 4 consecutive lanes), 64 quartets per pass of a workgroup.  Lanes c = 0, 1, 2 of
                    // the quad own Cartesian axis c: lane c runs the transfer + horizontal recurrences of axis c only (a third of
                    // the recurrence work, nothing redundant) and owns the integrals whose X component is p_c, i.e. a third of the
                    // integral block.  Such an integral is  g_c[X-power 1] * g_(c+1)[X-power 0] * g_(c+2)[X-power 0]:  the lane needs its
                    // own 1-D array and the X-power-0 HALF of its two neighbours' arrays, which it fetches register to register with
                    // two quad-permute DPP moves per double (no LDS array, no barrier).  Every lane enumerates its block in axes
                    // ROTATED so that its own axis comes first (then the code is the same for the three lanes); the rotated
                    // component labels are turned back into AO offsets by compile-time permutation tables selected by c.
                    // Each lane evaluates ONE Rys root per primitive combination (lane r of the quad: root r, so the fourth lane
                    // is of use when there are four roots) and the root in work is broadcast inside the quad.  Purpose: the
                    // 100-180-integral classes fit 256 registers (no AGPR copies, second workgroup per CU); price: the fourth
                    // lane idles otherwise, and blocks that do not carry X are added to the LDS tiles by three lanes.
#endif
#ifndef ORED
#define ORED 0      // row-lane mode with the j components in registers (CJR): the three outputs that are summed over the bra
                    // component i -- J_kl, K_jk, K_jl, i.e. over the LANES of a quartet -- are not added to the LDS Fock tiles by
                    // every lane (all T lanes of a quartet, and the neighbouring quartets of the wave, hit the SAME address in
                    // one instruction: ds_add_f64 serialises them, ~140 cycles per instruction), but written to a per-wave
                    // scratch area [value][lane] (conflict-free), summed by one OWNER lane per value, and added once per
                    // quartet with all lanes of the instruction on different addresses.  The scratch aliases the TRR array,
                    // which is dead during the contraction.
#endif
#ifndef ORED_HOIST
#define ORED_HOIST 1  // owner reduction: the owner's quartets are decoded once per step, the reads of a pass are issued together, sums pairwise
#endif
#ifndef RSPLIT
#define RSPLIT 1    // row-lane mode: the Rys roots of a primitive combination go through phase A / phase B in RSPLIT groups, so the TRR
                    // array holds NROOTS / RSPLIT roots per quartet: half the LDS of the largest array of these kernels (2121: 61 of
                    // 100 KB), i.e. room for a second workgroup per CU
#endif
#ifndef PAROOT
#define PAROOT 0    // row-lane mode: a phase-A job = (quartet, root) and runs the transfer recurrence of all three axes, so the
                    // Rys root and the primitive prefactors are evaluated once instead of three times
#endif
#ifndef HB
#define HB 0        // row-lane mode, "h form": phase A runs the transfer recurrence AND the bra horizontal recurrence, once per (quartet,
                    // root, axis), and leaves h[root][axis][i][j][c] (i <= LI, j <= LJ, c <= LKL) in LDS; a phase-B lane then reads ONE row
                    // (LKL + 1 reals) per (axis, j component), runs only the cheap ket recurrence and multiplies.  The plain row-lane
                    // forms redo the bra recurrence in every lane, root and integral chunk (hardware / model flop 1.7-2.6).
                    // Lane = (bra component ci, j-group jg): HEJ consecutive j components per lane, NJG = NFJ / HEJ lanes per ci, so the
                    // integral block of a lane is sized by HEJ instead of by chunks over k (no second pass through phase A).
#endif
#ifndef HEJ
#define HEJ 1       // HB: bra j components per lane = the largest divisor of NFJ that is <= HEJ
#endif
constexpr int pick_hej()
{
    int best = 1;
    for (int n = 1; n <= NFJ && n <= HEJ; n++)
        if (NFJ % n == 0) best = n;
    return best;
}
constexpr int EJ = HB ? pick_hej() : CJR ? NFJ : 1;  // bra j components held per lane
constexpr int NJG = NFJ / EJ;                        // lanes per bra component ci
constexpr int T = NFI * NJG;
static_assert(!HB || !CJR, "HB: CJR is its HEJ = NFJ case");
#ifndef TBLOCK
#define TBLOCK 256   // threads per workgroup (512: two waves per SIMD share one set of LDS tiles; row-lane mode only)
#endif
#ifndef WSYNC
#define WSYNC 0     // row-lane mode, T <= 64: every quartet lives inside ONE wave (phase A by lanes of the same wave), so the
                    // step loop needs no workgroup barrier and the waves of a workgroup drift apart (LDS atomics of one
                    // overlap the arithmetic of another); costs the 64 % T lanes left over in every wave
#endif
#ifndef ECAP
#define ECAP 64
#endif
constexpr int pick_nch()
{
    for (int n = 1; n <= NFK; n++)
        if (NFK % n == 0 && (NFK / n) * NFL * EJ <= ECAP) return n;
    return NFK;
}
constexpr int NCH = pick_nch();
#ifndef KW
#define KW 0        // row-lane mode with the owner reduction, classes whose integral block needs NCH > 1 chunks over the ket components k:
                    // the chunks are worked on by DIFFERENT WAVES at the same time instead of one after the other.  A workgroup has
                    // NCH x 4 waves (TBLOCK = 256 NCH); wave w owns the quartet slots of wave group w % 4 and chunk w / 4, so the NCH
                    // waves of a group share the quartets' recurrence arrays in LDS: phase A runs ONCE per (step, primitive combination)
                    // instead of once per chunk, its jobs dealt over all NCH x 256 lanes, and -- one set of Fock / density tiles per CU
                    // instead of two -- LDS holds every Rys root of the combination, so one pass instead of RSPLIT.
#endif
constexpr int KPARTS = KW ? NCH : 1;              // waves that share a quartet slot group
constexpr int NWAVE = TBLOCK / 64;
constexpr int NWAVE_S = NWAVE / KPARTS;           // wave groups that own quartet slots
constexpr int GW = T <= 64 ? 64 / T : 0;          // quartets per wave (WSYNC)
constexpr bool WMAP = WSYNC || (ORED && !TILE_1Q && T <= 64);     // lane -> quartet map with whole quartets per wave
constexpr int G = WMAP ? NWAVE_S * GW : TBLOCK / T;
static_assert(!KW || (!TILE_1Q && ORED && !WSYNC && T <= 64 && NCH > 1 && NWAVE == 4 * NCH), "KW: owner-reduction builds with workgroup-wide steps, TBLOCK = 256 NCH");
#ifndef TILE_1Q
#define TILE_1Q 0   // 1: one quartet per lane inside the tile (small classes); 0: T row lanes per quartet
#endif
#ifndef RYS_LDS_MAX
#define RYS_LDS_MAX 28672   // stage the class's Chebyshev table in LDS when it is at most this many bytes (nroots <= 5 in f64)
#endif
#ifndef NKS
#define NKS 1       // ket tile pairs staged per iteration (lane-per-quartet mode: 2 or 4 where a tile pair has few candidates)
#endif
#ifndef MINW
#define MINW 2      // waves/SIMD the register allocator leaves room for.  Never 1: builds with more than 256 registers per lane
                    // (AGPR spill space) gave wrong results in a few classes (tools/verify_scheme.py)
#endif
#ifndef UNROLL_B
#define UNROLL_B (KW ? 0 : 1)   // 1: unroll the root loop of phase B (loads of root r+1 overlap the products of root r); KW builds hold every
                                // root of a combination in LDS: their unrolled five-root loop spills 540-620 B and runs 5x slower (rolled: 116 B)
#endif
#ifndef ST_LDS_MAX
#define ST_LDS_MAX 40960   // double-buffer the TRR array (phase A of the next combination overlaps phase B) up to this size
#endif
constexpr int CW = NFK / NCH;
constexpr int E = EJ * CW * NFL;                     // integrals per lane and chunk: e = (cj * CW + kk) * NFL + cl
constexpr int WI = TSI * NFI, WJ = TSJ * NFJ, WK = TSK * NFK, WL = TSL * NFL;
constexpr int NT2 = HB ? (LI + 1) * (LJ + 1) * (LKL + 1) : (LIJ + 1) * (LKL + 1);    // reals per (root, axis) in the LDS array
constexpr int NRH = (NROOTS + RSPLIT - 1) / RSPLIT;                       // roots per phase-A / phase-B pass
constexpr int NJOB = G * 3 * NRH;                                         // phase-A jobs per pass
// The TRR array is single-buffered.  A double-buffered schedule (phase A of the next primitive combination issued before
// phase B of the current one, one barrier per combination; code paths under NBUF > 1 below) measured 3-8 % faster, but it
// gives wrong J/K in a few classes ((fd|fp), several g classes) on large inputs -- found by tools/verify_scheme.py, not
// understood yet (no LDS race found by inspection; extra barriers do not cure it) -- so it stays disabled.
#ifndef TRR_DOUBLE_BUFFER
#define TRR_DOUBLE_BUFFER 0
#endif
#ifndef TPAD
#define TPAD 0      // extra reals per quartet slot of the TRR array (bank spread of the phase-B reads across quartets)
#endif
constexpr int TRR_SLOT = NRH * 3 * NT2 + TPAD;                              // reals per quartet slot
constexpr bool USE_ORED = ORED && !TILE_1Q && T <= 64;
constexpr int NBUF = (TRR_DOUBLE_BUFFER && !WSYNC && !USE_ORED && 2 * G * TRR_SLOT * (int)sizeof(real) <= ST_LDS_MAX) ? 2 : 1;
// owner reduction (ORED): values per lane and step, scratch rows of RSTR doubles (odd: the owner lanes read column-wise)
// rows: J_kl [CW * NFL], K_jk [EJ * CW], K_jl [EJ * NFL]; lane = (ci, cj) form in addition K_ik [CW], K_il [NFL] (summed over cj)
constexpr int NE0 = DO_J ? CW * NFL : 0, NE1 = DO_K ? EJ * CW : 0, NE2 = DO_K ? EJ * NFL : 0;
constexpr int NE3 = (DO_K && !CJR && !(HB && NJG == 1)) ? CW : 0, NE4 = (DO_K && !CJR && !(HB && NJG == 1)) ? NFL : 0, NPART = NE0 + NE1 + NE2 + NE3 + NE4;
constexpr int RSTR = 66;
constexpr int RDBL = 8 / (int)sizeof(real);                                 // reals per double
#ifndef RGMIN_ROWS
#define RGMIN_ROWS 16
#endif
constexpr int RG_MIN = NPART < RGMIN_ROWS ? NPART : RGMIN_ROWS;              // rows the scratch of a wave holds at least
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }
// reals of the TRR array that belong to one wave (WSYNC: its GW slots; otherwise a quarter of the whole array)
constexpr int WREG0 = WMAP ? GW * TRR_SLOT : (G * TRR_SLOT + NWAVE - 1) / NWAVE;
// (KW: the KPARTS waves of a slot group split the group's region for their owner-reduction scratch)
constexpr int WPART = USE_ORED ? ((cmax((WREG0 + KPARTS - 1) / KPARTS, RG_MIN * RSTR * RDBL) + 3) & ~3) : WREG0;
constexpr int WREG = USE_ORED ? KPARTS * WPART : WREG0;
constexpr int ST_LEN = USE_ORED ? cmax(NWAVE_S * WREG, NBUF * G * TRR_SLOT) : NBUF * G * TRR_SLOT;
constexpr int RG = cmin(cmin(WPART / (RSTR * RDBL), NPART), 64);
constexpr int NPASS = RG > 0 ? (NPART + RG - 1) / RG : 0;
constexpr int NH = RG > 0 ? 64 / RG : 1;                                    // owner lanes per scratch row (they split the quartets)
// offset of quartet slot `sl` in the TRR array
__device__ __forceinline__ int trr_off(const int sl)
{
    constexpr int gw = GW > 0 ? GW : 1;
    return WMAP ? (sl / gw) * WREG + (sl % gw) * TRR_SLOT : sl * TRR_SLOT;
}
static_assert(!WSYNC || (T <= 64 && !TILE_1Q), "WSYNC needs a quartet to fit one wave");
static_assert(RSPLIT == 1 || (!TILE_1Q && NBUF == 1), "RSPLIT: row-lane mode, single-buffered TRR array");
static_assert(!CJR || !TILE_1Q, "CJR is a variant of the row-lane mode");
static_assert(!HB || (!TILE_1Q && ORED && T <= 64), "HB: row-lane builds with the owner reduction, a quartet inside one wave");
static_assert(NDM == 1 || NDM == 2, "density matrices per integral evaluation");
static_assert(NDM == 1 || ((TILE_1Q || USE_ORED) && !CTWO_ && !CORD_ && !STAGE_ALL_), "NDM > 1: lane-per-quartet or owner-reduction builds");
#if WSYNC
// ordering of LDS traffic inside one wave is kept by the hardware (one in-order DS queue per wave); the compiler only
// has to keep the program order of the accesses
#define STEP_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define STEP_SYNC() __syncthreads()
#endif
constexpr int RYS_TAB = (2 * NROOTS + 14) * NROOTS * NCOEF * 2;          // Chebyshev table of this class, in reals
constexpr bool RYS_IN_LDS = RYS_TAB * (int)sizeof(real) <= RYS_LDS_MAX;
static_assert(T <= TBLOCK && G >= 1 && NQ <= 65535, "tile geometry");
static_assert((TSI + TSJ) * BASIS_STRIDE <= TBLOCK && (TSK + TSL) * BASIS_STRIDE <= TBLOCK, "shell rows of a tile pair are staged by one pass");
static_assert(TBLOCK == 256 || !TILE_1Q, "the lane-per-quartet mode uses 256 threads");
constexpr int KS_SHIFT = NKS == 1 ? 16 : NKS == 2 ? 15 : NKS == 4 ? 14 : 13;     // queue entry = candidate id | ket slot << KS_SHIFT
static_assert(NKS == 1 || (TILE_1Q && (NKS == 2 || NKS == 4 || NKS == 8) && NQ <= (1 << KS_SHIFT)),
              "several ket pairs per iteration: lane-per-quartet mode, 16-bit queue entries");
static_assert(!MIXED || (TILE_1Q && !FP32 && NDM == 1 && !STAGE_ALL_), "MIXED: FP64 lane-per-quartet builds, one density matrix per evaluation");
typedef float v2f __attribute__((ext_vector_type(2)));

// Timing-only ablations of the lane-per-quartet mode (WRONG results; tools/ablate.py): bit 0 no Rys table gather, bit 1 no
// LDS atomics (sums kept alive in a register), bit 2 no density reads from LDS, bit 3 no integral evaluation, bit 4 no
// compute phase at all (staging + screening + flush only)
#ifndef ABL
#define ABL 0
#endif
#ifndef QIL
#define QIL 0       // lane-per-quartet mode: 1 = the survivor queue is read with a stride (consecutive lanes take entries of
                    // different ket slots / far-apart candidates: fewer lanes of one wave instruction on the same LDS Fock
                    // element) and J_ij, the one tile every quartet of the workgroup shares, is kept in JREP replicas
#endif
#ifndef JREP
#define JREP (QIL ? 4 : 1)
#endif
#ifndef STAGE_ALL
#define STAGE_ALL 0  // 1: issue every global load of every ket slot of an iteration (tiles, prefactors, screening gathers) before
                     // the first one is used
#endif
#ifndef RYS_SPLIT
#define RYS_SPLIT 0
#endif
#ifndef CTWO
#define CTWO 0      // lane-per-quartet mode: 1 = contraction in two sweeps over the integral block -- first the outputs indexed by
                    // the bra component i (J_ij, K_ik, K_il: emitted row by row), then the three accumulated over i (J_kl, K_jk,
                    // K_jl) -- so that only half of the density values and accumulators are live at a time (registers)
#endif
#ifndef CORD
#define CORD 0      // lane-per-quartet mode: 1 = contraction with every density read of a row issued before the row's LDS atomics
                    // of the PREVIOUS row (an LDS read queued behind a same-address atomic waits for its serialised lanes)
#endif
// Rys root `r` only (same tables and branches as rys_roots in jk_common.h)
__device__ __forceinline__ void rys_root_one(real x, real theta, real omega, const int r, const real* cheb,
                                             const real* __restrict__ large, real& root, real& weight)
{
    real tf = 1, stf = 1;
    x *= theta;
#if ABL & 1
    root = x * real(1e-3) + real(0.3) * (r + 1); weight = real(0.5) + x * real(1e-4);
    return;
#endif
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
#if RYS_SPLIT
    // root and weight polynomials one after the other: 14 coefficients in flight instead of 28 (register pressure)
    {
        real b1 = 0, b2 = 0;
#pragma unroll
        for (int k = NCOEF - 1; k >= 1; k--) { const real t = c[2 * k] + u2 * b1 - b2; b2 = b1; b1 = t; }
        root = (c[0] + u * b1 - b2) * tf;
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        real b1 = 0, b2 = 0;
#pragma unroll
        for (int k = NCOEF - 1; k >= 1; k--) { const real t = c[2 * k + 1] + u2 * b1 - b2; b2 = b1; b1 = t; }
        weight = (c[1] + u * b1 - b2) * stf;
    }
#else
    real br1 = 0, br2 = 0, bw1 = 0, bw2 = 0;
#pragma unroll
    for (int k = NCOEF - 1; k >= 1; k--) {
        real t = c[2 * k] + u2 * br1 - br2; br2 = br1; br1 = t;
        t = c[2 * k + 1] + u2 * bw1 - bw2; bw2 = bw1; bw1 = t;
    }
    root = (c[0] + u * br1 - br2) * tf;
    weight = (c[1] + u * bw1 - bw2) * stf;
#endif
}

#if MIXED
// Rys root `r` of TWO quartets at once (FP32, one per vector component): same tables and branches as rys_root_one.  The table
// row of each component is gathered separately (its own x interval); the Clenshaw recurrences run packed.  `cheb` is the FP32
// copy of the class's table (LDS) or, for classes whose table stays in L2, the FP64 table converted on the fly.
template <typename TAB>
__device__ __forceinline__ void rys_root_pk(v2f x, const v2f theta, const float omega, const int r, const TAB* cheb,
                                            const double* __restrict__ large, v2f& root, v2f& weight)
{
    v2f tf = {1.f, 1.f}, stf = {1.f, 1.f};
    x *= theta;
#if RYS_LR
    {
        const float w2 = omega * omega;
        tf.x = w2 * __builtin_amdgcn_rcpf(w2 + theta.x);
        tf.y = w2 * __builtin_amdgcn_rcpf(w2 + theta.y);
        x *= tf;
        stf.x = __builtin_amdgcn_sqrtf(tf.x);
        stf.y = __builtin_amdgcn_sqrtf(tf.y);
    }
#endif
    const float lim = float(5 * NROOTS + 35);
    const bool big0 = x.x >= lim, big1 = x.y >= lim;
    // (a component beyond the table takes the asymptotic form below; its polynomial is evaluated on interval 0 and dropped)
    const int it0 = big0 ? 0 : (int)(x.x * 0.4f), it1 = big1 ? 0 : (int)(x.y * 0.4f);
    const v2f u = {(x.x - 2.5f * it0) * 0.8f - 1.f, (x.y - 2.5f * it1) * 0.8f - 1.f};
    const v2f u2 = u + u;
    const TAB* c0 = cheb + (it0 * NROOTS + r) * (NCOEF * 2);
    const TAB* c1 = cheb + (it1 * NROOTS + r) * (NCOEF * 2);
    v2f br1 = {0.f, 0.f}, br2 = {0.f, 0.f}, bw1 = {0.f, 0.f}, bw2 = {0.f, 0.f};
#pragma unroll
    for (int k = NCOEF - 1; k >= 1; k--) {
        const v2f cr = {(float)c0[2 * k], (float)c1[2 * k]}, cw = {(float)c0[2 * k + 1], (float)c1[2 * k + 1]};
        v2f t = cr + u2 * br1 - br2; br2 = br1; br1 = t;
        t = cw + u2 * bw1 - bw2; bw2 = bw1; bw1 = t;
    }
    {
        const v2f cr = {(float)c0[0], (float)c1[0]}, cw = {(float)c0[1], (float)c1[1]};
        root = (cr + u * br1 - br2) * tf;
        weight = (cw + u * bw1 - bw2) * stf;
    }
    if (big0 || big1) {
        const float lr = (float)large[2 * r], lw = (float)large[2 * r + 1];
        if (big0) { const float isx = __builtin_amdgcn_rsqf(x.x); root.x = lr * isx * isx * tf.x; weight.x = lw * isx * stf.x; }
        if (big1) { const float isx = __builtin_amdgcn_rsqf(x.y); root.y = lr * isx * isx * tf.y; weight.y = lw * isx * stf.y; }
    }
}
#endif

#if QUAD
// split index X and the rotated-component -> AO-component tables
constexpr int XS = LL == 1 ? 3 : LK == 1 ? 2 : LJ == 1 ? 1 : LI == 1 ? 0 : -1;
static_assert(XS >= 0, "QUAD: the class needs a p shell");
static_assert(TILE_1Q && !MIXED && NROOTS <= 4, "QUAD: lane-per-quartet builds up to four Rys roots");
constexpr int NXI = XS == 0 ? 1 : NFI, NXJ = XS == 1 ? 1 : NFJ, NXK = XS == 2 ? 1 : NFK, NXL = XS == 3 ? 1 : NFL;
constexpr int NINTQ = NXI * NXJ * NXK * NXL;              // integrals per lane
// chunks of the lane's block over the components of a second index (QY = 0..3 = i, j, k, l; -1: none): QNCH passes, each with its own
// evaluation of the recurrences and roots, for the classes whose third of the block would not fit the registers
#ifndef QNCH
#define QNCH 1
#endif
#ifndef QY
#define QY (-1)
#endif

static_assert(QNCH == 1 || (QY >= 0 && QY <= 3 && QY != XS), "QUAD chunks: over an index other than the split one");
constexpr int CI = QY == 0 ? NXI / QNCH : NXI, CJ = QY == 1 ? NXJ / QNCH : NXJ, CK = QY == 2 ? NXK / QNCH : NXK, CL = QY == 3 ? NXL / QNCH : NXL;
static_assert(CI * CJ * CK * CL * QNCH == NINTQ, "QUAD chunks must divide the component count of their index");
constexpr int NINTC = CI * CJ * CK * CL;                  // integrals per lane and chunk
constexpr int GS_X = XS == 0 ? GS_I : XS == 1 ? GS_J : XS == 2 ? GS_K : GS_L;
// component n of a shell of angular momentum l, read with the axes rotated by c (own axis first): exponents (p, q, r) on
// (a0, a1, a2) = (c, c+1, c+2) mod 3  ->  index of the Cartesian component with those exponents on (x, y, z)
constexpr int rot_comp(const int l, const int c, const int n)
{
    const CartPow p = cart_pow(l, n);
    const int ex = c == 0 ? p.x : c == 1 ? p.z : p.y;
    const int ey = c == 0 ? p.y : c == 1 ? p.x : p.z;
    return (l - ex) * (l - ex + 1) / 2 + (l - ex - ey);
}
// quad permutes (DPP): lane c of a quad reads lane (c + 1) % 3 / (c + 2) % 3 (lane 3 mirrors lane 0)
#define DPP_ROT1 0x49   /* quad_perm:[1,2,0,1] */
#define DPP_ROT2 0x92   /* quad_perm:[2,0,1,2] */
template <int CTRL> __device__ __forceinline__ double dpp_quad(const double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_quad(const float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// value of lane r of the quad (r wave-uniform)
__device__ __forceinline__ real quad_bcast(const real v, const int r)
{
    switch (r) {
    case 0: return dpp_quad<0x00>(v);
    case 1: return dpp_quad<0x55>(v);
    case 2: return dpp_quad<0xaa>(v);
    default: return dpp_quad<0xff>(v);
    }
}
#endif

#if HB
// bra horizontal recurrence of one axis on the transfer-recurrence array tt[a][c] (a <= LIJ), written to LDS as h[i][j][c]:
// (i, j + 1) = (i + 1, j) - (Rj - Ri) (i, j)   (reference jk/1q1t.cu:336-360)
__device__ __forceinline__ void bra_hrr_store(real (&tt)[LIJ + 1][LKL + 1], const real rij, real* __restrict__ dst)
{
#pragma unroll
    for (int j = 0; j <= LJ; j++) {
#pragma unroll
        for (int i = 0; i <= LI; i++)
#pragma unroll
            for (int cc = 0; cc <= LKL; cc++) dst[(i * (LJ + 1) + j) * (LKL + 1) + cc] = tt[i][cc];
        if (j < LJ) {
#pragma unroll
            for (int a = 0; a < LIJ - j; a++)
#pragma unroll
                for (int cc = 0; cc <= LKL; cc++) tt[a][cc] = tt[a + 1][cc] - rij * tt[a][cc];
        }
    }
}
#endif

// Staging is split into "issue every global load" and "write LDS": all loads of a workgroup's staging step are in
// flight together (one L2 round trip), instead of one round trip per tile as a load->store loop would cost.
template <int NR, int NC>
struct TileRegs { real v[(NR * NC + TBLOCK - 1) / TBLOCK]; };
template <int NR, int NC>
__device__ __forceinline__ void tile_load(TileRegs<NR, NC>& t, const real* __restrict__ dm, const int nao, const int r0,
                                          const int c0, const int tid)
{
#pragma unroll
    for (int u = 0; u < (NR * NC + TBLOCK - 1) / TBLOCK; u++) {
        const int idx = tid + u * TBLOCK;
        const int r = idx / NC, c = idx - r * NC;
        t.v[u] = (idx < NR * NC && r0 + r < nao && c0 + c < nao) ? dm[(size_t)(r0 + r) * nao + c0 + c] : real(0);
    }
}
template <int NR, int NC>
__device__ __forceinline__ void tile_store(real* __restrict__ dst, const TileRegs<NR, NC>& t, const int tid)
{
#pragma unroll
    for (int u = 0; u < (NR * NC + TBLOCK - 1) / TBLOCK; u++) {
        const int idx = tid + u * TBLOCK;
        if (idx < NR * NC) dst[idx] = t.v[u];
    }
}

// add the tile to the global matrix and clear it (the thread that flushes an element is the one that clears it)
#ifndef FLUSH_UNROLL
#define FLUSH_UNROLL TILE_1Q   // 1: straight-line flush of a tile (its size is compile-time) instead of a rolled loop.  The compiler
                         // puts an `s_waitcnt vmcnt(0)` in front of every LDS store that follows a global atomic, i.e. every
                         // element waits for the previous atomic to complete at L2; unrolled, a tile's LDS reads are issued
                         // together and its few waits overlap.  Measured on the 112-atom run: -4.7 % over the 30
                         // lane-per-quartet classes ((dp|ps) -11 %); 2 (every tile read first, all atomics last) costs scratch
                         // and gains only 1 % (profiles/r02_ab_flush_staging_tile1q_112atoms.txt); row-lane kernels: 1 %, left rolled.
#endif
#if FLUSH_UNROLL
template <int NR, int NC>
__device__ __forceinline__ void flush_tile_t(double* __restrict__ src, double* __restrict__ out, const int nao,
                                             const int r0, const int c0, const int tid)
{
    constexpr int NU = (NR * NC + TBLOCK - 1) / TBLOCK;
    double v[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int idx = tid + u * TBLOCK;
        v[u] = idx < NR * NC ? src[idx] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int idx = tid + u * TBLOCK;
        const int r = idx / NC, c = idx - r * NC;
        if (v[u] != 0.0) {
            src[idx] = 0.0;
#ifndef NO_FLUSH
            if (r0 + r < nao && c0 + c < nao) atomic_add_f64(out + (size_t)(r0 + r) * nao + c0 + c, v[u]);
#endif
        }
    }
}
#define flush_tile(src, out, nao, r0, c0, NR, NC, tid) flush_tile_t<NR, NC>(src, out, nao, r0, c0, tid)
// two-step form: every LDS element of every tile is read (and cleared) first, the global atomics of all tiles follow with no
// LDS access between them (the compiler puts an `s_waitcnt vmcnt(0)` in front of an LDS store that follows a global atomic)
template <int NR, int NC> struct FlushRegs { double v[(NR * NC + TBLOCK - 1) / TBLOCK]; };
template <int NR, int NC>
__device__ __forceinline__ void flush_read(FlushRegs<NR, NC>& f, double* __restrict__ src, const int tid)
{
#pragma unroll
    for (int u = 0; u < (NR * NC + TBLOCK - 1) / TBLOCK; u++) {
        const int idx = tid + u * TBLOCK;
        f.v[u] = idx < NR * NC ? src[idx] : 0.0;
        if (f.v[u] != 0.0) src[idx] = 0.0;
    }
}
template <int NR, int NC>
__device__ __forceinline__ void flush_atomics(const FlushRegs<NR, NC>& f, double* __restrict__ out, const int nao, const int r0,
                                              const int c0, const int tid)
{
#pragma unroll
    for (int u = 0; u < (NR * NC + TBLOCK - 1) / TBLOCK; u++) {
        const int idx = tid + u * TBLOCK;
        const int r = idx / NC, c = idx - r * NC;
#ifndef NO_FLUSH
        if (f.v[u] != 0.0 && r0 + r < nao && c0 + c < nao) atomic_add_f64(out + (size_t)(r0 + r) * nao + c0 + c, f.v[u]);
#endif
    }
}
#else
__device__ __forceinline__ void flush_tile(double* __restrict__ src, double* __restrict__ out, const int nao,
                                           const int r0, const int c0, const int NR, const int NC, const int tid)
{
    for (int idx = tid; idx < NR * NC; idx += TBLOCK) {
        const int r = idx / NC, c = idx - r * NC;
        const double v = src[idx];
        if (v != 0.0) {
            src[idx] = 0.0;
#ifndef NO_FLUSH   // timing-only ablation (wrong results): no global atomics
            if (r0 + r < nao && c0 + c < nao) atomic_add_f64(out + (size_t)(r0 + r) * nao + c0 + c, v);
#endif
        }
    }
}
#endif

__device__ __forceinline__ void lds_add(double* p, double v) { atomicAdd(p, v); }   // ds_add_f64

// candidate id -> ket shell index c inside the tile.  Lane-per-quartet mode skews c by (a + b + d): the 64 quartets of
// a wave then spread evenly over the 16 targets of EVERY Fock sub-block (4 lanes per LDS address instead of 16 for
// J_kl / K_ik / K_jk with the plain order); the row-lane mode keeps the plain order (its register accumulators
// rely on consecutive quartets sharing the ket pair).
#if TILE_1Q
#define QC(craw, a, b, d) (((craw) + (a) + (b) + (d)) % TSK)
#else
#define QC(craw, a, b, d) (craw)
#endif
#ifndef STAMPS
#define STAMPS 0    // diagnostic build: wave 0 adds the cycles it spends per phase to counter[-1-phase] (tools/stamps_profile.py)
#endif
#if STAMPS
#define STAMP(k) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_last; st_last = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
#ifndef KNAME
#define KNAME jk_tile
#endif
// The kernel arguments as they lie in the kernarg segment (same order / natural alignment as the signature below).
// The ket loop re-reads the pointers it needs from there (scalar loads, K$ hits) at the start of its staging and
// flush sections instead of keeping ~15 pointers + scalars live in SGPRs across the compute phase: the compute
// phase is what runs out of SGPRs (exec-mask stack + uniform loop state), and every SGPR held for the staging code
// is one more spilled to a VGPR lane there.
#ifndef KARG_RELOAD
#define KARG_RELOAD 1
#endif
#define AS4 __attribute__((address_space(4)))
struct KArgs {
    int nao; const real* basis; const real* dm; double* vj; double* vk; real omega; const int* tasks; int ntasks;
    const unsigned* tpair_sh; const float* tpair_q; const float* q_cond; const float* log_dm; int nbas;
    float cut_lo, cut_hi, log_max_dm; int n_dm; const real* rys_cheb; const real* rys_large;
    unsigned long long* counter; const int* blk_index; const unsigned* tpair_ao; const unsigned* tpair_pp;
    const real* pair_tab; unsigned long long* counter32;
};
__device__ __forceinline__ const KArgs AS4* kargs()
{
    const KArgs AS4* p = (const KArgs AS4*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));          // opaque: the loads below it stay where they are written
    return p;
}
extern "C" __global__ void __launch_bounds__(TBLOCK, MINW)
KNAME(const int nao, const real* __restrict__ basis, const real* __restrict__ dm, double* __restrict__ vj,
        double* __restrict__ vk, const real omega, const int* __restrict__ tasks, const int ntasks,
        const unsigned* __restrict__ tpair_sh, const float* __restrict__ tpair_q, const float* __restrict__ q_cond,
        const float* __restrict__ log_dm, const int nbas, const float cut_lo, const float cut_hi,
        const float log_max_dm, const int n_dm, const real* __restrict__ rys_cheb, const real* __restrict__ rys_large,
        unsigned long long* __restrict__ counter, const int* __restrict__ blk_index,
        const unsigned* __restrict__ tpair_ao, const unsigned* __restrict__ tpair_pp, const real* __restrict__ pair_tab,
        unsigned long long* __restrict__ counter32)       // (MIXED builds: quartets evaluated in FP32, per task row; otherwise unused)
{
    __shared__ unsigned s_nact[2];              // survivors of the current tile pair (double-buffered by iteration parity)
    // NKS ket tile pairs are staged and screened per iteration (lane-per-quartet mode only): classes with few candidates
    // per tile pair fill the 256 lanes from several ket pairs; queue entry = candidate id | ket slot << KS_SHIFT
    __shared__ unsigned short s_act[NKS * NQ];  // their candidate ids, appended wave by wave
    // (NDM > 1: [dm][ket slot][tile])
    __shared__ real sDij[NDM * WJ * WI], sDkl[NDM * NKS * WL * WK], sDik[NDM * NKS * WI * WK], sDil[NDM * NKS * WI * WL], sDjk[NDM * NKS * WJ * WK], sDjl[NDM * NKS * WJ * WL];
    __shared__ double sJij[NDM * (TILE_1Q ? JREP : 1) * WJ * WI], sJkl[NDM * NKS * WL * WK], sKik[NDM * NKS * WI * WK], sKil[NDM * NKS * WI * WL], sKjk[NDM * NKS * WJ * WK], sKjl[NDM * NKS * WJ * WL];
#if !TILE_1Q
    __shared__ __attribute__((aligned(16))) real sT[ST_LEN];
#endif
    // shell rows of the four tiles and per-primitive-pair prefactors {c_a c_b K_ab, 1/(a+b), a+b}:
    // every exp / reciprocal of the pair prefactors is evaluated once per tile pair, not once per quartet
    __shared__ real sBas[(TSI + TSJ + NKS * (TSK + TSL)) * BASIS_STRIDE];
    __shared__ real sPB[TSI * TSJ * 9 * 3], sPK[NKS * TSK * TSL * 9 * 3];
    // Rys Chebyshev table of the class: every lane reads 28 coefficients of ITS OWN x-interval per root, i.e. 64
    // different cache lines per wave instruction from global memory; from LDS the same gather costs a few cycles
    __shared__ real sRys[RYS_IN_LDS ? RYS_TAB : 1];
#if MIXED
    __shared__ unsigned s_nact32[2];            // FP32-window survivors: appended to s_act from its END downwards
    __shared__ float sRys32[RYS_IN_LDS ? RYS_TAB : 1];
    constexpr int QEND = NKS * NQ;
    unsigned nq32_done = 0;
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wave_s = wave_u % NWAVE_S, kpart = wave_u / NWAVE_S;      // slot-owning wave group, chunk of this wave (KW; otherwise wave, 0)
#if STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    // ---- which (task row, bra pair, ket chunk): coarse index per 256 workgroups + one wave-wide probe of the
    //      following rows (no chain of dependent loads as a binary search would need)
    int row;
    {
        const int b = blockIdx.x;
        row = blk_index[b >> 8];
        for (;;) {
            const int r = row + 1 + lane;
            const bool p = r < ntasks && tasks[r * 8 + 5] <= b;
            const int c = __popcll(__ballot(p));
            row += c;
            if (c < 64) break;
        }
        row = __builtin_amdgcn_readfirstlane(row);
    }
    STAMP(0);
    const int* __restrict__ tk = tasks + row * 8;
    const int ij0 = tk[0], kl0 = tk[2], nkl = tk[3], nchunk = tk[4], kchunk = tk[7] & 0xffff;
    // small launches: the candidates of one (bra pair, ket chunk) are dealt to nsplit workgroups (contiguous id ranges)
    const int nsplit = tk[7] >> 16;
    const int lb0 = blockIdx.x - tk[5];
    const int lb = lb0 / nsplit, sid = lb0 - lb * nsplit;
    const int cand_lo = NQ * sid / nsplit, cand_hi = NQ * (sid + 1) / nsplit;
    const int bij = lb / nchunk, ch = lb - bij * nchunk;
    const int kt0 = ch * kchunk, kt1 = min(nkl, kt0 + kchunk);
    // (the host may round the chunk count of a task row up to a multiple of 8 and start every row at a multiple of 8, so that
    //  workgroup b and ket chunk b mod 8 always meet on the same XCD -- blocks are dealt round-robin over the 8 XCDs -- and the
    //  ket-side tables of a chunk stay in ONE L2: the surplus workgroups have nothing to do)
    if (kt0 >= nkl || bij >= tk[1]) return;
    const unsigned pij = tpair_sh[ij0 + bij], aoij = tpair_ao[ij0 + bij];
    const float qij = tpair_q[ij0 + bij] + log_max_dm;
    if (qij + tpair_q[kl0 + kt0] <= cut_lo) return;          // ket list is sorted: nothing in this chunk survives
    const int ish0 = pij >> 16, jsh0 = pij & 0xffff;
    const int i0 = aoij >> 16, j0 = aoij & 0xffff;           // first AO of the two tiles
    constexpr int OFF_J = TSI * BASIS_STRIDE, OFF_K = (TSI + TSJ) * BASIS_STRIDE, OFF_L = (TSI + TSJ + TSK) * BASIS_STRIDE;
    constexpr int KSTR = (TSK + TSL) * BASIS_STRIDE;          // ket slot stride in sBas
    constexpr int NRYS = RYS_IN_LDS ? (RYS_TAB + TBLOCK - 1) / TBLOCK : 1;
    // ---- bra side, once per workgroup: issue every load, then write LDS
    {
        real rb = 0, rrys[NRYS];
        if (tid < (TSI + TSJ) * BASIS_STRIDE) {
            const int sl = tid / BASIS_STRIDE, w = tid - sl * BASIS_STRIDE;
            rb = basis[(sl < TSI ? ish0 + sl : jsh0 + sl - TSI) * BASIS_STRIDE + w];
        }
        // primitive-pair prefactors of the bra tile pair: a contiguous block of the per-geometry table
        constexpr int NPB = (TSI * TSJ * 27 + TBLOCK - 1) / TBLOCK;
        real rpb[NPB];
        const real* __restrict__ ppb = pair_tab + (size_t)tpair_pp[ij0 + bij] * 27;
#pragma unroll
        for (int u = 0; u < NPB; u++) rpb[u] = tid + u * TBLOCK < TSI * TSJ * 27 ? ppb[tid + u * TBLOCK] : real(0);
        if (RYS_IN_LDS) {
#pragma unroll
            for (int u = 0; u < NRYS; u++) rrys[u] = tid + u * TBLOCK < RYS_TAB ? rys_cheb[tid + u * TBLOCK] : real(0);
        }
#if DO_J
        for (int n = tid; n < NDM * (TILE_1Q ? JREP : 1) * WJ * WI; n += TBLOCK) sJij[n] = 0;
        for (int n = tid; n < NDM * NKS * WL * WK; n += TBLOCK) sJkl[n] = 0;
#endif
#if DO_K
        for (int n = tid; n < NDM * NKS * WI * WK; n += TBLOCK) sKik[n] = 0;
        for (int n = tid; n < NDM * NKS * WI * WL; n += TBLOCK) sKil[n] = 0;
        for (int n = tid; n < NDM * NKS * WJ * WK; n += TBLOCK) sKjk[n] = 0;
        for (int n = tid; n < NDM * NKS * WJ * WL; n += TBLOCK) sKjl[n] = 0;
#endif
        if (tid < 2) s_nact[tid] = 0;
#if MIXED
        if (tid < 2) s_nact32[tid] = 0;
#endif
        if (tid < (TSI + TSJ) * BASIS_STRIDE) sBas[tid] = rb;
#pragma unroll
        for (int u = 0; u < NPB; u++)
            if (tid + u * TBLOCK < TSI * TSJ * 27) sPB[tid + u * TBLOCK] = rpb[u];
        if (RYS_IN_LDS) {
#pragma unroll
            for (int u = 0; u < NRYS; u++)
                if (tid + u * TBLOCK < RYS_TAB) {
                    sRys[tid + u * TBLOCK] = rrys[u];
#if MIXED
                    sRys32[tid + u * TBLOCK] = (float)rrys[u];
#endif
                }
        }
    }
    const real* cheb_tab = RYS_IN_LDS ? sRys : rys_cheb;
    __syncthreads();            // counters and Fock tiles are clear before any wave appends / accumulates
    STAMP(1);

#if !TILE_1Q
    // WMAP: whole quartets per wave (wave-local steps and / or the owner reduction); otherwise quartets packed over the workgroup
    const int qslot = WMAP ? lane / T : 0;
    const int slot = WMAP ? wave_s * GW + qslot : tid / T;
    const int t = WMAP ? lane - qslot * T : tid - slot * T;
    const bool lane_on = WMAP ? qslot < GW : slot < G;
#if CJR
    const int ci = t, cj = 0;
#elif HB
    const int ci = t / NJG, jg = t - ci * NJG, cj = jg * EJ;       // cj: the lane's first j component
#else
    const int ci = t / NFJ, cj = t - ci * NFJ;
#endif
    const int ibra[3] = {TI.x[ci], TI.y[ci], TI.z[ci]};
    const int jbra[3] = {TJ.x[cj], TJ.y[cj], TJ.z[cj]};
#if HB
    // row of h[i][j][.] the lane reads for its j component e on each axis (NJG > 1: the j powers of a lane are run-time values)
    int hrow[EJ][3];
#pragma unroll
    for (int e = 0; e < EJ; e++) {
        const int cje = cj + e;
        hrow[e][0] = (ibra[0] * (LJ + 1) + TJ.x[cje]) * (LKL + 1);
        hrow[e][1] = (ibra[1] * (LJ + 1) + TJ.y[cje]) * (LKL + 1);
        hrow[e][2] = (ibra[2] * (LJ + 1) + TJ.z[cje]) * (LKL + 1);
    }
#endif
#endif
    const size_t nao2 = (size_t)nao * nao;
    unsigned nq_done = 0;
    int parity = 0;

#if KARG_RELOAD
    for (int idm = 0; idm < kargs()->n_dm; idm += NDM) {
        const int ndm_grp = NDM > 1 ? kargs()->n_dm - idm : 1;      // density matrices of this group that exist
#else
    for (int idm = 0; idm < n_dm; idm += NDM) {
        const real* __restrict__ D = dm + idm * nao2;
        const int ndm_grp = NDM > 1 ? n_dm - idm : 1;
#endif
#if DO_J
        __syncthreads();
        {
#if KARG_RELOAD
            const KArgs AS4* kd = kargs();
            const int nao = kd->nao;
            const real* __restrict__ D = kd->dm + idm * ((size_t)nao * nao);
#endif
            const int ndm_here = ndm_grp;
            TileRegs<WJ, WI> r[NDM];
#pragma unroll
            for (int dmi = 0; dmi < NDM; dmi++)       // (a group's missing matrix reads as zero: nao = 0 fails every bounds test)
                tile_load(r[dmi], D + (dmi < ndm_here ? dmi : 0) * ((size_t)nao * nao), dmi < ndm_here ? nao : 0, j0, i0, tid);
#pragma unroll
            for (int dmi = 0; dmi < NDM; dmi++) tile_store(sDij + dmi * (WJ * WI), r[dmi], tid);
        }
        STAMP(2);
#endif
        for (int kt = kt0; kt < kt1; kt += NKS) {
#if KARG_RELOAD
            const KArgs AS4* ka = kargs();
            const int nao = ka->nao, nbas = ka->nbas;
            const float cut_lo = ka->cut_lo, cut_hi = ka->cut_hi;
            const unsigned* __restrict__ tpair_sh = ka->tpair_sh;
            const unsigned* __restrict__ tpair_ao = ka->tpair_ao;
            const unsigned* __restrict__ tpair_pp = ka->tpair_pp;
            const float* __restrict__ tpair_q = ka->tpair_q;
            const float* __restrict__ q_cond = ka->q_cond;
            const float* __restrict__ log_dm = ka->log_dm;
            const real* __restrict__ basis = ka->basis;
            const real* __restrict__ pair_tab = ka->pair_tab;
            const real* __restrict__ D = ka->dm + idm * ((size_t)nao * nao);
#endif
            if (qij + tpair_q[kl0 + kt] <= cut_lo) break;
            int ksh0s[NKS], lsh0s[NKS], k0s[NKS], l0s[NKS];
            bool kval[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                // (the ket list is sorted by its bound: once a pair fails the cut every later one does)
                kval[ks] = ks == 0 || (kt + ks < kt1 && qij + tpair_q[kl0 + kt + ks] > cut_lo);
                const unsigned pkl = kval[ks] ? tpair_sh[kl0 + kt + ks] : 0u, aokl = kval[ks] ? tpair_ao[kl0 + kt + ks] : 0u;
                ksh0s[ks] = pkl >> 16; lsh0s[ks] = pkl & 0xffff;
                k0s[ks] = aokl >> 16; l0s[ks] = aokl & 0xffff;
            }
#if SKIP_EMPTY
            {
                const int bra_max = (ish0 + TSI - 1) * nbas + jsh0 + TSJ - 1;
                bool any = false;
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    kval[ks] = kval[ks] && ksh0s[ks] * nbas + lsh0s[ks] <= bra_max;
                    any = any || kval[ks];
                }
                if (!any) continue;
            }
#endif
            parity ^= 1;
            // index arithmetic of the staging / flush loops is re-derived from an opaque copy of the thread id: keeps
            // the (cheap) loop-invariant addresses from being hoisted over the compute phase and spilled there
            int tid_s = tid;
            asm volatile("" : "+v"(tid_s));
#define tid tid_s
            if (tid == 0) s_nact[parity ^ 1] = 0;          // next iteration's counter (last read before this point)
#if MIXED
            if (tid == 0) s_nact32[parity ^ 1] = 0;
#endif
#if STAGE_ALL
            // ---- every global load of EVERY ket slot is issued before the first use of any of them: one L2 round trip per
            //      iteration instead of one per slot (the screening of a slot waits for its gathers, and a vmcnt wait covers
            //      every older load as well)
            constexpr int NPK = (TSK * TSL * 27 + TBLOCK - 1) / TBLOCK;
            constexpr int NR = (NQ + TBLOCK - 1) / TBLOCK;           // screening rounds per slot
            real rb_[NKS], rpk_[NKS][NPK];
#if DO_J
            TileRegs<WL, WK> rkl_[NKS];
#endif
#if DO_K
            TileRegs<WI, WK> rik_[NKS];
            TileRegs<WI, WL> ril_[NKS];
            TileRegs<WJ, WK> rjk_[NKS];
            TileRegs<WJ, WL> rjl_[NKS];
#endif
            float sq_[NKS][NR], sdv_[NKS][NR][6];
            bool can_[NKS][NR];
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                rb_[ks] = 0;
                if (!kval[ks]) continue;
                const int ksh0 = ksh0s[ks], lsh0 = lsh0s[ks], k0 = k0s[ks], l0 = l0s[ks];
                if (tid < (TSK + TSL) * BASIS_STRIDE) {
                    const int sl = tid / BASIS_STRIDE, w = tid - sl * BASIS_STRIDE;
                    rb_[ks] = basis[(sl < TSK ? ksh0 + sl : lsh0 + sl - TSK) * BASIS_STRIDE + w];
                }
                const real* __restrict__ ppk = pair_tab + (size_t)tpair_pp[kl0 + kt + ks] * 27;
#pragma unroll
                for (int u = 0; u < NPK; u++) rpk_[ks][u] = tid + u * TBLOCK < TSK * TSL * 27 ? ppk[tid + u * TBLOCK] : real(0);
#if DO_J
                tile_load(rkl_[ks], D, nao, l0, k0, tid);
#endif
#if DO_K
                tile_load(rik_[ks], D, nao, i0, k0, tid);
                tile_load(ril_[ks], D, nao, i0, l0, tid);
                tile_load(rjk_[ks], D, nao, j0, k0, tid);
                tile_load(rjl_[ks], D, nao, j0, l0, tid);
#endif
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const int cd = cand_lo + r * TBLOCK + tid;
                    can_[ks][r] = false;
                    sq_[ks][r] = 0;
#pragma unroll
                    for (int n = 0; n < 6; n++) sdv_[ks][r][n] = -36.8f;
                    if (cd < cand_hi) {
                        const int a = cd % TSI, b = (cd / TSI) % TSJ, d = (cd / (TSI * TSJ)) % TSL;
                        const int c = QC(cd / (TSI * TSJ * TSL), a, b, d);
                        const int ish = ish0 + a, jsh = jsh0 + b, ksh = ksh0 + c, lsh = lsh0 + d;
                        if (ish >= jsh && ksh >= lsh && ish * nbas + jsh >= ksh * nbas + lsh) {
                            can_[ks][r] = true;
                            sq_[ks][r] = q_cond[ish * nbas + jsh] + q_cond[ksh * nbas + lsh];
#if DO_K
                            sdv_[ks][r][0] = log_dm[ish * nbas + ksh];
                            sdv_[ks][r][1] = log_dm[jsh * nbas + ksh];
                            sdv_[ks][r][2] = log_dm[ish * nbas + lsh];
                            sdv_[ks][r][3] = log_dm[jsh * nbas + lsh];
#endif
#if DO_J
                            sdv_[ks][r][4] = log_dm[ish * nbas + jsh];
                            sdv_[ks][r][5] = log_dm[ksh * nbas + lsh];
#endif
                        }
                    }
                }
            }
            STAMP(3);
            // ---- screening predicate, survivors of every wave appended to the queue through one LDS counter
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                if (!kval[ks]) continue;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (cand_lo + r * TBLOCK >= cand_hi) continue;
                    const int cd = cand_lo + r * TBLOCK + tid;
                    float sd = -36.8f;
#pragma unroll
                    for (int n = 0; n < 6; n++) sd = fmaxf(sd, sdv_[ks][r][n]);
                    const float dq = sq_[ks][r] + sd;
                    const bool keep = can_[ks][r] && dq > cut_lo && dq <= cut_hi;
                    const unsigned long long m = __ballot(keep);
                    if (m) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(&s_nact[parity], (unsigned)__popcll(m));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (keep) s_act[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(cd | (ks << KS_SHIFT));
                    }
                }
            }
            // ---- write LDS
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                if (!kval[ks]) continue;
                if (tid < (TSK + TSL) * BASIS_STRIDE) sBas[OFF_K + ks * KSTR + tid] = rb_[ks];
#pragma unroll
                for (int u = 0; u < NPK; u++)
                    if (tid + u * TBLOCK < TSK * TSL * 27) sPK[ks * (TSK * TSL * 27) + tid + u * TBLOCK] = rpk_[ks][u];
#if DO_J
                tile_store(sDkl + ks * (WL * WK), rkl_[ks], tid);
#endif
#if DO_K
                tile_store(sDik + ks * (WI * WK), rik_[ks], tid);
                tile_store(sDil + ks * (WI * WL), ril_[ks], tid);
                tile_store(sDjk + ks * (WJ * WK), rjk_[ks], tid);
                tile_store(sDjl + ks * (WJ * WL), rjl_[ks], tid);
#endif
            }
#else
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                if (!kval[ks]) continue;
                const int ksh0 = ksh0s[ks], lsh0 = lsh0s[ks], k0 = k0s[ks], l0 = l0s[ks];
                // ---- issue: ket shell rows, primitive-pair inputs, five density sub-blocks
                real rb = 0;
                if (tid < (TSK + TSL) * BASIS_STRIDE) {
                    const int sl = tid / BASIS_STRIDE, w = tid - sl * BASIS_STRIDE;
                    rb = basis[(sl < TSK ? ksh0 + sl : lsh0 + sl - TSK) * BASIS_STRIDE + w];
                }
                constexpr int NPK = (TSK * TSL * 27 + TBLOCK - 1) / TBLOCK;
                real rpk[NPK];
                const real* __restrict__ ppk = pair_tab + (size_t)tpair_pp[kl0 + kt + ks] * 27;
#pragma unroll
                for (int u = 0; u < NPK; u++) rpk[u] = tid + u * TBLOCK < TSK * TSL * 27 ? ppk[tid + u * TBLOCK] : real(0);
            const int ndm_here = ndm_grp;
#if DO_J
                TileRegs<WL, WK> rkl[NDM];
#pragma unroll
                for (int dmi = 0; dmi < NDM; dmi++)
                    tile_load(rkl[dmi], D + (dmi < ndm_here ? dmi : 0) * ((size_t)nao * nao), dmi < ndm_here ? nao : 0, l0, k0, tid);
#endif
#if DO_K
                TileRegs<WI, WK> rik[NDM];
                TileRegs<WI, WL> ril[NDM];
                TileRegs<WJ, WK> rjk[NDM];
                TileRegs<WJ, WL> rjl[NDM];
#pragma unroll
                for (int dmi = 0; dmi < NDM; dmi++) {
                    const real* __restrict__ Dd = D + (dmi < ndm_here ? dmi : 0) * ((size_t)nao * nao);
                    const int naod = dmi < ndm_here ? nao : 0;
                    tile_load(rik[dmi], Dd, naod, i0, k0, tid);
                    tile_load(ril[dmi], Dd, naod, i0, l0, tid);
                    tile_load(rjk[dmi], Dd, naod, j0, k0, tid);
                    tile_load(rjl[dmi], Dd, naod, j0, l0, tid);
                }
#endif
                STAMP(3);
                // ---- per-quartet screening of the NQ candidates of the tile pair (wave64 ballots); every wave appends
                //      its survivors to the queue through one LDS counter
#pragma unroll 2
                for (int cand0 = cand_lo; cand0 < cand_hi; cand0 += TBLOCK) {
                    const int cd = cand0 + tid;
                    bool keep = false;
#if MIXED
                    bool keep32 = false;
#endif
                    if (cd < cand_hi) {
                        const int a = cd % TSI, b = (cd / TSI) % TSJ, d = (cd / (TSI * TSJ)) % TSL;
                        const int c = QC(cd / (TSI * TSJ * TSL), a, b, d);
                        const int ish = ish0 + a, jsh = jsh0 + b, ksh = ksh0 + c, lsh = lsh0 + d;
                        if (ish >= jsh && ksh >= lsh && ish * nbas + jsh >= ksh * nbas + lsh) {
                            const float sq = q_cond[ish * nbas + jsh] + q_cond[ksh * nbas + lsh];
                            float sd = -36.8f;
#if DO_K
                            sd = fmaxf(sd, log_dm[ish * nbas + ksh]);
                            sd = fmaxf(sd, log_dm[jsh * nbas + ksh]);
                            sd = fmaxf(sd, log_dm[ish * nbas + lsh]);
                            sd = fmaxf(sd, log_dm[jsh * nbas + lsh]);
#endif
#if DO_J
                            sd = fmaxf(sd, log_dm[ish * nbas + jsh]);
                            sd = fmaxf(sd, log_dm[ksh * nbas + lsh]);
#endif
                            const float dq = sq + sd;
#if MIXED
                            keep = dq > cut_hi;                          // FP64 phase
                            keep32 = dq > cut_lo && dq <= cut_hi;        // FP32 phase (two quartets per lane)
#else
                            keep = dq > cut_lo && dq <= cut_hi;
#endif
                        }
                    }
                    const unsigned long long m = __ballot(keep);
                    if (m) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(&s_nact[parity], (unsigned)__popcll(m));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (keep) s_act[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(cd | (ks << KS_SHIFT));
                    }
#if MIXED
                    const unsigned long long m32 = __ballot(keep32);
                    if (m32) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(&s_nact32[parity], (unsigned)__popcll(m32));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (keep32) s_act[QEND - 1 - (base + __popcll(m32 & ((1ull << lane) - 1ull)))] = (unsigned short)(cd | (ks << KS_SHIFT));
                    }
#endif
                }
                // ---- write LDS
                if (tid < (TSK + TSL) * BASIS_STRIDE) sBas[OFF_K + ks * KSTR + tid] = rb;
#pragma unroll
                for (int u = 0; u < NPK; u++)
                    if (tid + u * TBLOCK < TSK * TSL * 27) sPK[ks * (TSK * TSL * 27) + tid + u * TBLOCK] = rpk[u];
#pragma unroll
                for (int dmi = 0; dmi < NDM; dmi++) {
#if DO_J
                    tile_store(sDkl + (dmi * NKS + ks) * (WL * WK), rkl[dmi], tid);
#endif
#if DO_K
                    tile_store(sDik + (dmi * NKS + ks) * (WI * WK), rik[dmi], tid);
                    tile_store(sDil + (dmi * NKS + ks) * (WI * WL), ril[dmi], tid);
                    tile_store(sDjk + (dmi * NKS + ks) * (WJ * WK), rjk[dmi], tid);
                    tile_store(sDjl + (dmi * NKS + ks) * (WJ * WL), rjl[dmi], tid);
#endif
                }
            }
#endif  // STAGE_ALL
#undef tid
            const int ksh0 = ksh0s[0], lsh0 = lsh0s[0], k0 = k0s[0], l0 = l0s[0];      // (row-lane mode: one ket pair)
            STAMP(4);
            __syncthreads();
            STAMP(5);
            const int nact = __builtin_amdgcn_readfirstlane((int)s_nact[parity]);
#if MIXED
            const int nact32 = __builtin_amdgcn_readfirstlane((int)s_nact32[parity]);
            if (nact == 0 && nact32 == 0) continue;
            if (idm == 0) nq32_done += nact32;
#else
            if (nact == 0) continue;
#endif
            const int npi = __builtin_amdgcn_readfirstlane((int)sBas[10]), npj = __builtin_amdgcn_readfirstlane((int)sBas[OFF_J + 10]);
            int kv0 = 0;                       // first staged ket slot (every slot of a task row has the same primitive counts)
#pragma unroll
            for (int ks = NKS - 1; ks >= 0; ks--)
                if (kval[ks]) kv0 = ks;
            const int npk = __builtin_amdgcn_readfirstlane((int)sBas[OFF_K + kv0 * KSTR + 10]), npl = __builtin_amdgcn_readfirstlane((int)sBas[OFF_L + kv0 * KSTR + 10]);
            if (idm == 0) nq_done += nact;

#if TILE_1Q
            // ---------------- one quartet per lane: everything in registers, then LDS Fock tiles
#if ABL & 2
            double abl_sink = 0;
#define LDS_ADD(p, v) (abl_sink += (double)(v))
#else
#define LDS_ADD(p, v) lds_add(p, v)
#endif
#if ABL & 4
#define DLD(x) real(0.37)
#else
#define DLD(x) (x)
#endif
#if QUAD
            // ---------------- one quartet per QUAD of lanes (see QUAD above)
            {
            const int qc = tid & 3;                        // lane of the quad
            const int ax0 = qc == 3 ? 0 : qc;              // own axis (lane 3: a copy of lane 0 that adds nothing)
            const bool q_on = qc < 3;
            const int rme = qc < NROOTS ? qc : NROOTS - 1; // the Rys root this lane evaluates
            for (int q1 = tid >> 2; q1 < nact; q1 += TBLOCK / 4) {
                const int qe = s_act[q1];
                const int ks = NKS > 1 ? qe >> KS_SHIFT : 0, qd = NKS > 1 ? qe & ((1 << KS_SHIFT) - 1) : qe;
                const int a = qd % TSI, b = (qd / TSI) % TSJ, d = (qd / (TSI * TSJ)) % TSL, c = QC(qd / (TSI * TSJ * TSL), a, b, d);
                int kshb = ksh0s[0], lshb = lsh0s[0];
#pragma unroll
                for (int u = 1; u < NKS; u++)
                    if (ks == u) { kshb = ksh0s[u]; lshb = lsh0s[u]; }
                const int ish = ish0 + a, jsh = jsh0 + b, ksh = kshb + c, lsh = lshb + d;
                const real* bi = sBas + a * BASIS_STRIDE;
                const real* bj = sBas + OFF_J + b * BASIS_STRIDE;
                const real* bk = sBas + OFF_K + ks * KSTR + c * BASIS_STRIDE;
                const real* bl = sBas + OFF_L + ks * KSTR + d * BASIS_STRIDE;
                const real* pb = sPB + (a * TSJ + b) * 27;
                const real* pk = sPK + (ks * TSK * TSL + c * TSL + d) * 27;
                // own-axis geometry only
                const real ri_o = bi[ax0], rk_o = bk[ax0];
                const real rij_o = bj[ax0] - ri_o, rkl_o = bl[ax0] - rk_o;
                real fac = real(34.98683665524972497);
                if (ish == jsh) fac *= real(0.5);
                if (ksh == lsh) fac *= real(0.5);
                if (ish == ksh && jsh == lsh) fac *= real(0.5);
                // the lane's third of the block is evaluated in QNCH chunks over the components of index QY (recurrences and roots
                // redone per chunk: the price of keeping <= ~60 integrals per lane for the 270-540-integral classes)
#pragma unroll
                for (int ch = 0; ch < QNCH; ch++) {
                const int i0 = QY == 0 ? ch * CI : 0, j0 = QY == 1 ? ch * CJ : 0, k0 = QY == 2 ? ch * CK : 0, l0 = QY == 3 ? ch * CL : 0;
                real I[NINTC];
#pragma unroll
                for (int n = 0; n < NINTC; n++) I[n] = 0;
                for (int kp = 0; kp < npk; kp++)
                for (int lp = 0; lp < npl; lp++) {
                    const real ckcl = pk[(kp * 3 + lp) * 3], inv_akl = pk[(kp * 3 + lp) * 3 + 1], akl = pk[(kp * 3 + lp) * 3 + 2];
                    const real al_akl = bl[5 + 2 * lp] * inv_akl;
                    const real rqc_o = rkl_o * al_akl;
                    for (int ip = 0; ip < npi; ip++)
                    for (int jp = 0; jp < npj; jp++) {
                        const real inv_aij = pb[(ip * 3 + jp) * 3 + 1], aij = pb[(ip * 3 + jp) * 3 + 2];
                        const real aj_aij = bj[5 + 2 * jp] * inv_aij;
                        const real cicj = fac * pb[(ip * 3 + jp) * 3];
                        const real rpa_o = rij_o * aj_aij;
                        const real rpq_o = rpa_o + ri_o - rqc_o - rk_o;
                        const real s2 = rpq_o * rpq_o;
                        const real rr = s2 + dpp_quad<DPP_ROT1>(s2) + dpp_quad<DPP_ROT2>(s2);     // |P - Q|^2 from the three axis lanes
                        const real sinv = fast_rsqrt(aij + akl);
                        const real inv = sinv * sinv;
                        const real theta = aij * akl * inv;
                        const real gy0 = cicj * inv_aij * inv_akl * sinv;
                        real t2m, wtm;
                        rys_root_one(rr, theta, omega, rme, cheb_tab, rys_large, t2m, wtm);
#pragma clang loop unroll(disable)
                        for (int ir = 0; ir < NROOTS; ir++) {
                            const real t2 = quad_bcast(t2m, ir), wt = quad_bcast(wtm, ir);
                            const real rt_aa = t2 * inv;
                            const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                            const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                            const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                            const real b00 = real(0.5) * rt_aa;
                            const real seed = ax0 == 0 ? ckcl : ax0 == 1 ? gy0 : wt;
                            real g0[GSIZE], g1[GSIZE], g2[GSIZE];
                            axis_integrals(seed, rpa_o - rt_aij * rpq_o, rqc_o + rt_akl * rpq_o, b10, b01, b00, rij_o, rkl_o, g0);
                            // X-power-0 halves of the two neighbouring axes
#pragma unroll
                            for (int e = 0; e < GSIZE; e++)
                                if ((e / GS_X) % 2 == 0) { g1[e] = dpp_quad<DPP_ROT1>(g0[e]); g2[e] = dpp_quad<DPP_ROT2>(g0[e]); }
#pragma unroll
                            for (int i = 0; i < CI; i++)
#pragma unroll
                            for (int j = 0; j < CJ; j++)
#pragma unroll
                            for (int k = 0; k < CK; k++)
#pragma unroll
                            for (int l = 0; l < CL; l++) {
                                const int e0 = TI.x[i0 + i] * GS_I + TJ.x[j0 + j] * GS_J + TK.x[k0 + k] * GS_K + TL.x[l0 + l];
                                const int e1 = TI.y[i0 + i] * GS_I + TJ.y[j0 + j] * GS_J + TK.y[k0 + k] * GS_K + TL.y[l0 + l];
                                const int e2 = TI.z[i0 + i] * GS_I + TJ.z[j0 + j] * GS_J + TK.z[k0 + k] * GS_K + TL.z[l0 + l];
                                I[((i * CJ + j) * CK + k) * CL + l] += g0[e0] * g1[e1] * g2[e2];
                            }
                        }
                    }
                }
                // ---- AO offsets of the rotated components: index X carries component ax0, the others the rotation table
                int oi[CI], oj[CJ], ok[CK], ol[CL];
#pragma unroll
                for (int n = 0; n < CI; n++) oi[n] = a * NFI + (XS == 0 ? ax0 : ax0 == 0 ? i0 + n : ax0 == 1 ? rot_comp(LI, 1, i0 + n) : rot_comp(LI, 2, i0 + n));
#pragma unroll
                for (int n = 0; n < CJ; n++) oj[n] = b * NFJ + (XS == 1 ? ax0 : ax0 == 0 ? j0 + n : ax0 == 1 ? rot_comp(LJ, 1, j0 + n) : rot_comp(LJ, 2, j0 + n));
#pragma unroll
                for (int n = 0; n < CK; n++) ok[n] = c * NFK + (XS == 2 ? ax0 : ax0 == 0 ? k0 + n : ax0 == 1 ? rot_comp(LK, 1, k0 + n) : rot_comp(LK, 2, k0 + n));
#pragma unroll
                for (int n = 0; n < CL; n++) ol[n] = d * NFL + (XS == 3 ? ax0 : ax0 == 0 ? l0 + n : ax0 == 1 ? rot_comp(LL, 1, l0 + n) : rot_comp(LL, 2, l0 + n));
                if (q_on) {
#if NDM > 1
#pragma unroll 1
                for (int dmi = 0; dmi < NDM && dmi < ndm_grp; dmi++) {
#else
                {
                constexpr int dmi = 0;
#endif
                const real* const qDij = sDij + dmi * (WJ * WI);
                double* const qJij = sJij + dmi * ((TILE_1Q ? JREP : 1) * WJ * WI);
                const real* qDkl = sDkl + (dmi * NKS + ks) * (WL * WK); const real* qDik = sDik + (dmi * NKS + ks) * (WI * WK);
                const real* qDil = sDil + (dmi * NKS + ks) * (WI * WL); const real* qDjk = sDjk + (dmi * NKS + ks) * (WJ * WK);
                const real* qDjl = sDjl + (dmi * NKS + ks) * (WJ * WL);
                double* qJkl = sJkl + (dmi * NKS + ks) * (WL * WK); double* qKik = sKik + (dmi * NKS + ks) * (WI * WK);
                double* qKil = sKil + (dmi * NKS + ks) * (WI * WL); double* qKjk = sKjk + (dmi * NKS + ks) * (WJ * WK);
                double* qKjl = sKjl + (dmi * NKS + ks) * (WJ * WL);
#if DO_J
                {
                    real jkl[CK * CL], dkl[CK * CL];
#pragma unroll
                    for (int k = 0; k < CK; k++)
#pragma unroll
                        for (int l = 0; l < CL; l++) { jkl[k * CL + l] = 0; dkl[k * CL + l] = qDkl[ol[l] * WK + ok[k]]; }
#pragma unroll
                    for (int i = 0; i < CI; i++)
#pragma unroll
                        for (int j = 0; j < CJ; j++) {
                            const real dij = qDij[oj[j] * WI + oi[i]];
                            real sj = 0;
#pragma unroll
                            for (int n = 0; n < CK * CL; n++) {
                                const real v = I[(i * CJ + j) * CK * CL + n];
                                sj += v * dkl[n];
                                jkl[n] += v * dij;
                            }
                            lds_add(&qJij[oj[j] * WI + oi[i]], (double)sj);
                        }
#pragma unroll
                    for (int k = 0; k < CK; k++)
#pragma unroll
                        for (int l = 0; l < CL; l++) lds_add(&qJkl[ol[l] * WK + ok[k]], (double)jkl[k * CL + l]);
                }
#endif
#if DO_K
                {
                    real kjk[CJ * CK], kjl[CJ * CL], djk[CJ * CK], djl[CJ * CL];
#pragma unroll
                    for (int j = 0; j < CJ; j++) {
#pragma unroll
                        for (int k = 0; k < CK; k++) { kjk[j * CK + k] = 0; djk[j * CK + k] = qDjk[oj[j] * WK + ok[k]]; }
#pragma unroll
                        for (int l = 0; l < CL; l++) { kjl[j * CL + l] = 0; djl[j * CL + l] = qDjl[oj[j] * WL + ol[l]]; }
                    }
#pragma unroll
                    for (int i = 0; i < CI; i++) {
                        real kik[CK], kil[CL], dik[CK], dil[CL];
#pragma unroll
                        for (int k = 0; k < CK; k++) { kik[k] = 0; dik[k] = qDik[oi[i] * WK + ok[k]]; }
#pragma unroll
                        for (int l = 0; l < CL; l++) { kil[l] = 0; dil[l] = qDil[oi[i] * WL + ol[l]]; }
#pragma unroll
                        for (int j = 0; j < CJ; j++)
#pragma unroll
                            for (int k = 0; k < CK; k++)
#pragma unroll
                                for (int l = 0; l < CL; l++) {
                                    const real v = I[((i * CJ + j) * CK + k) * CL + l];
                                    kik[k] += v * djl[j * CL + l];
                                    kil[l] += v * djk[j * CK + k];
                                    kjk[j * CK + k] += v * dil[l];
                                    kjl[j * CL + l] += v * dik[k];
                                }
#pragma unroll
                        for (int k = 0; k < CK; k++) lds_add(&qKik[oi[i] * WK + ok[k]], (double)kik[k]);
#pragma unroll
                        for (int l = 0; l < CL; l++) lds_add(&qKil[oi[i] * WL + ol[l]], (double)kil[l]);
                    }
#pragma unroll
                    for (int j = 0; j < CJ; j++) {
#pragma unroll
                        for (int k = 0; k < CK; k++) lds_add(&qKjk[oj[j] * WK + ok[k]], (double)kjk[j * CK + k]);
#pragma unroll
                        for (int l = 0; l < CL; l++) lds_add(&qKjl[oj[j] * WL + ol[l]], (double)kjl[j * CL + l]);
                    }
                }
#endif
                }
                }
                }   // chunk
            }
            }
#else   // !QUAD
#if QIL
            // strided read: lane l takes entry (l % QS) * qchunk + l / QS, so the QS = 4 neighbours of a lane group come from
            // four far-apart quarters of the queue (different ket slots when NKS > 1)
            constexpr int QS = 4;
            const int qchunk = (nact + QS - 1) / QS;
            double* const sJij_r = sJij + (lane & (JREP - 1)) * (WJ * WI);
            for (int q0 = tid; q0 < ((ABL & 16) ? 0 : QS * qchunk); q0 += TBLOCK) {
                const int q1 = (q0 % QS) * qchunk + q0 / QS;
                if (q1 >= nact) continue;
                const int qe = s_act[q1];
#else
            double* const sJij_r = sJij;
            for (int q1 = tid; q1 < ((ABL & 16) ? 0 : nact); q1 += TBLOCK) {
                const int qe = s_act[q1];
#endif
#if STAMPS
                // (diagnostic) slot 13: active lanes of wave 0 summed over its batches; slot 10: time from the end of the previous
                // batch to here (queue read, loop control)
                if (tid == 0) st_acc[13] += __popcll(__ballot(true));
                STAMP(10);
#endif
                const int ks = NKS > 1 ? qe >> KS_SHIFT : 0, qd = NKS > 1 ? qe & ((1 << KS_SHIFT) - 1) : qe;
                const int a = qd % TSI, b = (qd / TSI) % TSJ, d = (qd / (TSI * TSJ)) % TSL, c = QC(qd / (TSI * TSJ * TSL), a, b, d);
                int kshb = ksh0s[0], lshb = lsh0s[0];
#pragma unroll
                for (int u = 1; u < NKS; u++)
                    if (ks == u) { kshb = ksh0s[u]; lshb = lsh0s[u]; }
                const int ish = ish0 + a, jsh = jsh0 + b, ksh = kshb + c, lsh = lshb + d;
                const real* bi = sBas + a * BASIS_STRIDE;
                const real* bj = sBas + OFF_J + b * BASIS_STRIDE;
                const real* bk = sBas + OFF_K + ks * KSTR + c * BASIS_STRIDE;
                const real* bl = sBas + OFF_L + ks * KSTR + d * BASIS_STRIDE;
                const real* pb = sPB + (a * TSJ + b) * 27;
                const real* pk = sPK + (ks * TSK * TSL + c * TSL + d) * 27;
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
#if ABL & 8
#pragma unroll
                for (int n = 0; n < NINT; n++) I[n] = fac * (rij[0] + real(n + 1)) * rkl[1];
#endif
                for (int kp = 0; kp < ((ABL & 8) ? 0 : npk); kp++)
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
                STAMP(11);          // (diagnostic) integral evaluation of this batch
                const int iA = a * NFI, jA = b * NFJ, kA = c * NFK, lA = d * NFL;
#if NDM > 1
#pragma unroll 1
                for (int dmi = 0; dmi < NDM && dmi < ndm_grp; dmi++) {
#else
                {
                constexpr int dmi = 0;
#endif
                // views of this density matrix's tiles: D_ij / J_ij, and the ket slot of the ket-dependent ones
                const real* const sDij_v = sDij + dmi * (WJ * WI);
                double* const sJij_v = sJij_r + dmi * ((TILE_1Q ? JREP : 1) * WJ * WI);
                const real* sDkl_q = sDkl + (dmi * NKS + ks) * (WL * WK); const real* sDik_q = sDik + (dmi * NKS + ks) * (WI * WK);
                const real* sDil_q = sDil + (dmi * NKS + ks) * (WI * WL); const real* sDjk_q = sDjk + (dmi * NKS + ks) * (WJ * WK);
                const real* sDjl_q = sDjl + (dmi * NKS + ks) * (WJ * WL);
                double* sJkl_q = sJkl + (dmi * NKS + ks) * (WL * WK); double* sKik_q = sKik + (dmi * NKS + ks) * (WI * WK);
                double* sKil_q = sKil + (dmi * NKS + ks) * (WI * WL); double* sKjk_q = sKjk + (dmi * NKS + ks) * (WJ * WK);
                double* sKjl_q = sKjl + (dmi * NKS + ks) * (WJ * WL);
                {
                const real* const sDij = sDij_v;          // (shadow the arrays: the code below is written for one matrix)
                double* const sJij_r = sJij_v;
#if CTWO
                {
                    // ---- sweep 1: outputs of row i (J_ij, K_ik, K_il); live: D_kl, D_jl, D_jk
                    {
                        real dkl[NFK * NFL], djk[NFJ * NFK], djl[NFJ * NFL];
#pragma unroll
                        for (int k = 0; k < NFK; k++)
#pragma unroll
                            for (int l = 0; l < NFL; l++) dkl[k * NFL + l] = DO_J ? DLD(sDkl_q[(lA + l) * WK + kA + k]) : real(0);
#pragma unroll
                        for (int j = 0; j < NFJ; j++) {
#pragma unroll
                            for (int k = 0; k < NFK; k++) djk[j * NFK + k] = DO_K ? DLD(sDjk_q[(jA + j) * WK + kA + k]) : real(0);
#pragma unroll
                            for (int l = 0; l < NFL; l++) djl[j * NFL + l] = DO_K ? DLD(sDjl_q[(jA + j) * WL + lA + l]) : real(0);
                        }
#pragma unroll
                        for (int i = 0; i < NFI; i++) {
                            real sij[NFJ], kik[NFK], kil[NFL];
#pragma unroll
                            for (int j = 0; j < NFJ; j++) sij[j] = 0;
#pragma unroll
                            for (int k = 0; k < NFK; k++) kik[k] = 0;
#pragma unroll
                            for (int l = 0; l < NFL; l++) kil[l] = 0;
#pragma unroll
                            for (int j = 0; j < NFJ; j++)
#pragma unroll
                                for (int k = 0; k < NFK; k++)
#pragma unroll
                                    for (int l = 0; l < NFL; l++) {
                                        const real v = I[((i * NFJ + j) * NFK + k) * NFL + l];
#if DO_J
                                        sij[j] += v * dkl[k * NFL + l];
#endif
#if DO_K
                                        kik[k] += v * djl[j * NFL + l];
                                        kil[l] += v * djk[j * NFK + k];
#endif
                                    }
#if DO_J
#pragma unroll
                            for (int j = 0; j < NFJ; j++) LDS_ADD(&sJij_r[(jA + j) * WI + iA + i], (double)sij[j]);
#endif
#if DO_K
#pragma unroll
                            for (int k = 0; k < NFK; k++) LDS_ADD(&sKik_q[(iA + i) * WK + kA + k], (double)kik[k]);
#pragma unroll
                            for (int l = 0; l < NFL; l++) LDS_ADD(&sKil_q[(iA + i) * WL + lA + l], (double)kil[l]);
#endif
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- sweep 2: outputs accumulated over i (J_kl, K_jk, K_jl); live: the accumulators and one row of D_ij, D_ik, D_il
                    {
                        real jkl[NFK * NFL], kjk[NFJ * NFK], kjl[NFJ * NFL];
#pragma unroll
                        for (int n = 0; n < NFK * NFL; n++) jkl[n] = 0;
#pragma unroll
                        for (int n = 0; n < NFJ * NFK; n++) kjk[n] = 0;
#pragma unroll
                        for (int n = 0; n < NFJ * NFL; n++) kjl[n] = 0;
#pragma unroll
                        for (int i = 0; i < NFI; i++) {
                            real dij[NFJ], dik[NFK], dil[NFL];
#pragma unroll
                            for (int j = 0; j < NFJ; j++) dij[j] = DO_J ? DLD(sDij[(jA + j) * WI + iA + i]) : real(0);
#pragma unroll
                            for (int k = 0; k < NFK; k++) dik[k] = DO_K ? DLD(sDik_q[(iA + i) * WK + kA + k]) : real(0);
#pragma unroll
                            for (int l = 0; l < NFL; l++) dil[l] = DO_K ? DLD(sDil_q[(iA + i) * WL + lA + l]) : real(0);
#pragma unroll
                            for (int j = 0; j < NFJ; j++)
#pragma unroll
                                for (int k = 0; k < NFK; k++)
#pragma unroll
                                    for (int l = 0; l < NFL; l++) {
                                        const real v = I[((i * NFJ + j) * NFK + k) * NFL + l];
#if DO_J
                                        jkl[k * NFL + l] += v * dij[j];
#endif
#if DO_K
                                        kjk[j * NFK + k] += v * dil[l];
                                        kjl[j * NFL + l] += v * dik[k];
#endif
                                    }
                        }
#if DO_J
#pragma unroll
                        for (int k = 0; k < NFK; k++)
#pragma unroll
                            for (int l = 0; l < NFL; l++) LDS_ADD(&sJkl_q[(lA + l) * WK + kA + k], (double)jkl[k * NFL + l]);
#endif
#if DO_K
#pragma unroll
                        for (int j = 0; j < NFJ; j++) {
#pragma unroll
                            for (int k = 0; k < NFK; k++) LDS_ADD(&sKjk_q[(jA + j) * WK + kA + k], (double)kjk[j * NFK + k]);
#pragma unroll
                            for (int l = 0; l < NFL; l++) LDS_ADD(&sKjl_q[(jA + j) * WL + lA + l], (double)kjl[j * NFL + l]);
                        }
#endif
                    }
                }
#elif CORD
                {
                    // ---- contraction, row by row in i: the density reads of row i + 1 are issued BEFORE the LDS atomics of
                    //      row i (the DS queue is in order: a read queued behind a same-address atomic waits for its lanes)
                    real jkl[NFK * NFL], dkl[NFK * NFL], kjk[NFJ * NFK], kjl[NFJ * NFL], djk[NFJ * NFK], djl[NFJ * NFL];
#pragma unroll
                    for (int k = 0; k < NFK; k++)
#pragma unroll
                        for (int l = 0; l < NFL; l++) { jkl[k * NFL + l] = 0; dkl[k * NFL + l] = DO_J ? DLD(sDkl_q[(lA + l) * WK + kA + k]) : real(0); }
#pragma unroll
                    for (int j = 0; j < NFJ; j++) {
#pragma unroll
                        for (int k = 0; k < NFK; k++) { kjk[j * NFK + k] = 0; djk[j * NFK + k] = DO_K ? DLD(sDjk_q[(jA + j) * WK + kA + k]) : real(0); }
#pragma unroll
                        for (int l = 0; l < NFL; l++) { kjl[j * NFL + l] = 0; djl[j * NFL + l] = DO_K ? DLD(sDjl_q[(jA + j) * WL + lA + l]) : real(0); }
                    }
                    real dij_n[NFJ], dik_n[NFK], dil_n[NFL];
                    auto load_row = [&](const int i) {
#pragma unroll
                        for (int j = 0; j < NFJ; j++) dij_n[j] = DO_J ? DLD(sDij[(jA + j) * WI + iA + i]) : real(0);
#pragma unroll
                        for (int k = 0; k < NFK; k++) dik_n[k] = DO_K ? DLD(sDik_q[(iA + i) * WK + kA + k]) : real(0);
#pragma unroll
                        for (int l = 0; l < NFL; l++) dil_n[l] = DO_K ? DLD(sDil_q[(iA + i) * WL + lA + l]) : real(0);
                    };
                    load_row(0);
#pragma unroll
                    for (int i = 0; i < NFI; i++) {
                        real dij[NFJ], dik[NFK], dil[NFL], sij[NFJ], kik[NFK], kil[NFL];
#pragma unroll
                        for (int j = 0; j < NFJ; j++) { dij[j] = dij_n[j]; sij[j] = 0; }
#pragma unroll
                        for (int k = 0; k < NFK; k++) { dik[k] = dik_n[k]; kik[k] = 0; }
#pragma unroll
                        for (int l = 0; l < NFL; l++) { dil[l] = dil_n[l]; kil[l] = 0; }
                        if (i + 1 < NFI) load_row(i + 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < NFJ; j++)
#pragma unroll
                            for (int k = 0; k < NFK; k++)
#pragma unroll
                                for (int l = 0; l < NFL; l++) {
                                    const real v = I[((i * NFJ + j) * NFK + k) * NFL + l];
#if DO_J
                                    sij[j] += v * dkl[k * NFL + l];
                                    jkl[k * NFL + l] += v * dij[j];
#endif
#if DO_K
                                    kik[k] += v * djl[j * NFL + l];
                                    kil[l] += v * djk[j * NFK + k];
                                    kjk[j * NFK + k] += v * dil[l];
                                    kjl[j * NFL + l] += v * dik[k];
#endif
                                }
                        __builtin_amdgcn_sched_barrier(0);
#if DO_J
#pragma unroll
                        for (int j = 0; j < NFJ; j++) LDS_ADD(&sJij_r[(jA + j) * WI + iA + i], (double)sij[j]);
#endif
#if DO_K
#pragma unroll
                        for (int k = 0; k < NFK; k++) LDS_ADD(&sKik_q[(iA + i) * WK + kA + k], (double)kik[k]);
#pragma unroll
                        for (int l = 0; l < NFL; l++) LDS_ADD(&sKil_q[(iA + i) * WL + lA + l], (double)kil[l]);
#endif
                    }
#if DO_J
#pragma unroll
                    for (int k = 0; k < NFK; k++)
#pragma unroll
                        for (int l = 0; l < NFL; l++) LDS_ADD(&sJkl_q[(lA + l) * WK + kA + k], (double)jkl[k * NFL + l]);
#endif
#if DO_K
#pragma unroll
                    for (int j = 0; j < NFJ; j++) {
#pragma unroll
                        for (int k = 0; k < NFK; k++) LDS_ADD(&sKjk_q[(jA + j) * WK + kA + k], (double)kjk[j * NFK + k]);
#pragma unroll
                        for (int l = 0; l < NFL; l++) LDS_ADD(&sKjl_q[(jA + j) * WL + lA + l], (double)kjl[j * NFL + l]);
                    }
#endif
                }
#else
#if DO_J
                {
                    real jkl[NFK * NFL], dkl[NFK * NFL];
#pragma unroll
                    for (int k = 0; k < NFK; k++)
#pragma unroll
                        for (int l = 0; l < NFL; l++) { jkl[k * NFL + l] = 0; dkl[k * NFL + l] = DLD(sDkl_q[(lA + l) * WK + kA + k]); }
#pragma unroll
                    for (int i = 0; i < NFI; i++)
#pragma unroll
                        for (int j = 0; j < NFJ; j++) {
                            const real dij = DLD(sDij[(jA + j) * WI + iA + i]);
                            real s = 0;
#pragma unroll
                            for (int n = 0; n < NFK * NFL; n++) {
                                const real v = I[(i * NFJ + j) * NFK * NFL + n];
                                s += v * dkl[n];
                                jkl[n] += v * dij;
                            }
                            LDS_ADD(&sJij_r[(jA + j) * WI + iA + i], (double)s);
                        }
#pragma unroll
                    for (int k = 0; k < NFK; k++)
#pragma unroll
                        for (int l = 0; l < NFL; l++) LDS_ADD(&sJkl_q[(lA + l) * WK + kA + k], (double)jkl[k * NFL + l]);
                }
#endif
#if DO_K
                {
                    real kjk[NFJ * NFK], kjl[NFJ * NFL], djk[NFJ * NFK], djl[NFJ * NFL];
#pragma unroll
                    for (int j = 0; j < NFJ; j++) {
#pragma unroll
                        for (int k = 0; k < NFK; k++) { kjk[j * NFK + k] = 0; djk[j * NFK + k] = DLD(sDjk_q[(jA + j) * WK + kA + k]); }
#pragma unroll
                        for (int l = 0; l < NFL; l++) { kjl[j * NFL + l] = 0; djl[j * NFL + l] = DLD(sDjl_q[(jA + j) * WL + lA + l]); }
                    }
#pragma unroll
                    for (int i = 0; i < NFI; i++) {
                        real kik[NFK], kil[NFL], dik[NFK], dil[NFL];
#pragma unroll
                        for (int k = 0; k < NFK; k++) { kik[k] = 0; dik[k] = DLD(sDik_q[(iA + i) * WK + kA + k]); }
#pragma unroll
                        for (int l = 0; l < NFL; l++) { kil[l] = 0; dil[l] = DLD(sDil_q[(iA + i) * WL + lA + l]); }
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
                        for (int k = 0; k < NFK; k++) LDS_ADD(&sKik_q[(iA + i) * WK + kA + k], (double)kik[k]);
#pragma unroll
                        for (int l = 0; l < NFL; l++) LDS_ADD(&sKil_q[(iA + i) * WL + lA + l], (double)kil[l]);
                    }
#pragma unroll
                    for (int j = 0; j < NFJ; j++) {
#pragma unroll
                        for (int k = 0; k < NFK; k++) LDS_ADD(&sKjk_q[(jA + j) * WK + kA + k], (double)kjk[j * NFK + k]);
#pragma unroll
                        for (int l = 0; l < NFL; l++) LDS_ADD(&sKjl_q[(jA + j) * WL + lA + l], (double)kjl[j * NFL + l]);
                    }
                }
#endif
            #endif  // CORD
                }
                }
                STAMP(12);          // (diagnostic) contraction + LDS atomics of this batch
}
#endif  // QUAD
#if MIXED
            // ---------------- FP32 phase of a MIXED build: TWO quartets per lane, every per-quartet quantity a 2-vector (component
            //                  x = queue entry 2p, y = entry 2p + 1 of the FP32 survivors, which grow downwards from the end of
            //                  s_act).  Same formulas, loop order and destinations as the FP64 phase above (reference
            //                  jk/1q1t.cu:174-638); shell data, prefactors and density tiles are the FP64 LDS copies converted
            //                  on load, the Fock tiles are FP64.  An odd survivor count leaves the y component of the last lane
            //                  with a zero prefactor and no LDS atomics.
#define PK2(p0, p1, n) ((v2f){(float)(p0)[n], (float)(p1)[n]})
            // (pairs are dealt from the LAST lane downwards: the FP64 phase fills the workgroup's waves from the front, and a wave runs
            //  the two phases one after the other -- dealt from the same end, waves 0-1 would run both and waves 2-3 neither)
            for (int p2 = TBLOCK - 1 - tid; 2 * p2 < nact32; p2 += TBLOCK) {
                const bool two = 2 * p2 + 1 < nact32;
                const int qe0 = s_act[QEND - 1 - 2 * p2];
                const int qe1 = two ? s_act[QEND - 2 - 2 * p2] : qe0;
                int ksv[2], av[2], bv[2], cv[2], dv[2];
                v2f fac;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int qe = h ? qe1 : qe0;
                    ksv[h] = NKS > 1 ? qe >> KS_SHIFT : 0;
                    const int qd = NKS > 1 ? qe & ((1 << KS_SHIFT) - 1) : qe;
                    av[h] = qd % TSI; bv[h] = (qd / TSI) % TSJ; dv[h] = (qd / (TSI * TSJ)) % TSL;
                    cv[h] = QC(qd / (TSI * TSJ * TSL), av[h], bv[h], dv[h]);
                    int kshb = ksh0s[0], lshb = lsh0s[0];
#pragma unroll
                    for (int u = 1; u < NKS; u++)
                        if (ksv[h] == u) { kshb = ksh0s[u]; lshb = lsh0s[u]; }
                    const int ish = ish0 + av[h], jsh = jsh0 + bv[h], ksh = kshb + cv[h], lsh = lshb + dv[h];
                    float f = 34.98683665524972497f;
                    if (ish == jsh) f *= 0.5f;
                    if (ksh == lsh) f *= 0.5f;
                    if (ish == ksh && jsh == lsh) f *= 0.5f;
                    if (h == 0) fac.x = f; else fac.y = two ? f : 0.f;
                }
                const real* bi0 = sBas + av[0] * BASIS_STRIDE, * bi1 = sBas + av[1] * BASIS_STRIDE;
                const real* bj0 = sBas + OFF_J + bv[0] * BASIS_STRIDE, * bj1 = sBas + OFF_J + bv[1] * BASIS_STRIDE;
                const real* bk0 = sBas + OFF_K + ksv[0] * KSTR + cv[0] * BASIS_STRIDE, * bk1 = sBas + OFF_K + ksv[1] * KSTR + cv[1] * BASIS_STRIDE;
                const real* bl0 = sBas + OFF_L + ksv[0] * KSTR + dv[0] * BASIS_STRIDE, * bl1 = sBas + OFF_L + ksv[1] * KSTR + dv[1] * BASIS_STRIDE;
                const real* pb0 = sPB + (av[0] * TSJ + bv[0]) * 27, * pb1 = sPB + (av[1] * TSJ + bv[1]) * 27;
                const real* pk0 = sPK + (ksv[0] * TSK * TSL + cv[0] * TSL + dv[0]) * 27, * pk1 = sPK + (ksv[1] * TSK * TSL + cv[1] * TSL + dv[1]) * 27;
                const v2f rix = PK2(bi0, bi1, 0), riy = PK2(bi0, bi1, 1), riz = PK2(bi0, bi1, 2);
                const v2f rkx = PK2(bk0, bk1, 0), rky = PK2(bk0, bk1, 1), rkz = PK2(bk0, bk1, 2);
                // (differences formed in FP64, then rounded: the shell centres are large numbers next to their distances)
                const v2f rij[3] = {{(float)(bj0[0] - bi0[0]), (float)(bj1[0] - bi1[0])}, {(float)(bj0[1] - bi0[1]), (float)(bj1[1] - bi1[1])},
                                    {(float)(bj0[2] - bi0[2]), (float)(bj1[2] - bi1[2])}};
                const v2f rkl[3] = {{(float)(bl0[0] - bk0[0]), (float)(bl1[0] - bk1[0])}, {(float)(bl0[1] - bk0[1]), (float)(bl1[1] - bk1[1])},
                                    {(float)(bl0[2] - bk0[2]), (float)(bl1[2] - bk1[2])}};
                const v2f rik[3] = {{(float)(bi0[0] - bk0[0]), (float)(bi1[0] - bk1[0])}, {(float)(bi0[1] - bk0[1]), (float)(bi1[1] - bk1[1])},
                                    {(float)(bi0[2] - bk0[2]), (float)(bi1[2] - bk1[2])}};
                (void)rix; (void)riy; (void)riz; (void)rkx; (void)rky; (void)rkz;
                v2f I[NINT];
#pragma unroll
                for (int n = 0; n < NINT; n++) I[n] = (v2f){0.f, 0.f};
                const float omega_f = (float)omega;
                for (int kp = 0; kp < npk; kp++)
                for (int lp = 0; lp < npl; lp++) {
                    const v2f ckcl = PK2(pk0, pk1, (kp * 3 + lp) * 3), inv_akl = PK2(pk0, pk1, (kp * 3 + lp) * 3 + 1), akl = PK2(pk0, pk1, (kp * 3 + lp) * 3 + 2);
                    const v2f al_akl = PK2(bl0, bl1, 5 + 2 * lp) * inv_akl;
                    for (int ip = 0; ip < npi; ip++)
                    for (int jp = 0; jp < npj; jp++) {
                        const v2f inv_aij = PK2(pb0, pb1, (ip * 3 + jp) * 3 + 1), aij = PK2(pb0, pb1, (ip * 3 + jp) * 3 + 2);
                        const v2f aj_aij = PK2(bj0, bj1, 5 + 2 * jp) * inv_aij;
                        const v2f cicj = fac * PK2(pb0, pb1, (ip * 3 + jp) * 3);
                        const v2f rpa[3] = {rij[0] * aj_aij, rij[1] * aj_aij, rij[2] * aj_aij};
                        const v2f rqc[3] = {rkl[0] * al_akl, rkl[1] * al_akl, rkl[2] * al_akl};
                        const v2f rpq[3] = {rpa[0] + rik[0] - rqc[0], rpa[1] + rik[1] - rqc[1], rpa[2] + rik[2] - rqc[2]};
                        const v2f rr = rpq[0] * rpq[0] + rpq[1] * rpq[1] + rpq[2] * rpq[2];
                        const v2f asum = aij + akl;
                        const v2f sinv = {__builtin_amdgcn_rsqf(asum.x), __builtin_amdgcn_rsqf(asum.y)};
                        const v2f inv = sinv * sinv;
                        const v2f theta = aij * akl * inv;
                        const v2f gy0 = cicj * inv_aij * inv_akl * sinv;
#pragma clang loop unroll(disable)
                        for (int ir = 0; ir < NROOTS; ir++) {
                            v2f t2, wt;
                            if (RYS_IN_LDS) rys_root_pk(rr, theta, omega_f, ir, sRys32, rys_large, t2, wt);
                            else rys_root_pk(rr, theta, omega_f, ir, rys_cheb, rys_large, t2, wt);
                            const v2f rt_aa = t2 * inv;
                            const v2f rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                            const v2f b10 = 0.5f * inv_aij * (1.f - rt_aij);
                            const v2f b01 = 0.5f * inv_akl * (1.f - rt_akl);
                            const v2f b00 = 0.5f * rt_aa;
                            v2f gx[GSIZE], gy[GSIZE], gz[GSIZE];
                            axis_integrals<v2f>(ckcl, rpa[0] - rt_aij * rpq[0], rqc[0] + rt_akl * rpq[0], b10, b01, b00, rij[0], rkl[0], gx);
                            axis_integrals<v2f>(gy0, rpa[1] - rt_aij * rpq[1], rqc[1] + rt_akl * rpq[1], b10, b01, b00, rij[1], rkl[1], gy);
                            axis_integrals<v2f>(wt, rpa[2] - rt_aij * rpq[2], rqc[2] + rt_akl * rpq[2], b10, b01, b00, rij[2], rkl[2], gz);
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
                // ---- contraction with the density tiles of each component's own (bra, ket slot) position
                const int iA0 = av[0] * NFI, jA0 = bv[0] * NFJ, kA0 = cv[0] * NFK, lA0 = dv[0] * NFL;
                const int iA1 = av[1] * NFI, jA1 = bv[1] * NFJ, kA1 = cv[1] * NFK, lA1 = dv[1] * NFL;
#if DO_J
                {
                    const real* dkl0 = sDkl + ksv[0] * (WL * WK), * dkl1 = sDkl + ksv[1] * (WL * WK);
                    double* jkl0 = sJkl + ksv[0] * (WL * WK), * jkl1 = sJkl + ksv[1] * (WL * WK);
                    v2f jkl[NFK * NFL], dkl[NFK * NFL];
#pragma unroll
                    for (int k = 0; k < NFK; k++)
#pragma unroll
                        for (int l = 0; l < NFL; l++) {
                            jkl[k * NFL + l] = (v2f){0.f, 0.f};
                            dkl[k * NFL + l] = (v2f){(float)dkl0[(lA0 + l) * WK + kA0 + k], (float)dkl1[(lA1 + l) * WK + kA1 + k]};
                        }
#pragma unroll
                    for (int i = 0; i < NFI; i++)
#pragma unroll
                        for (int j = 0; j < NFJ; j++) {
                            const v2f dij = {(float)sDij[(jA0 + j) * WI + iA0 + i], (float)sDij[(jA1 + j) * WI + iA1 + i]};
                            v2f sacc = {0.f, 0.f};
#pragma unroll
                            for (int n = 0; n < NFK * NFL; n++) {
                                const v2f v = I[(i * NFJ + j) * NFK * NFL + n];
                                sacc += v * dkl[n];
                                jkl[n] += v * dij;
                            }
                            LDS_ADD(&sJij_r[(jA0 + j) * WI + iA0 + i], (double)sacc.x);
                            if (two) LDS_ADD(&sJij_r[(jA1 + j) * WI + iA1 + i], (double)sacc.y);
                        }
#pragma unroll
                    for (int k = 0; k < NFK; k++)
#pragma unroll
                        for (int l = 0; l < NFL; l++) {
                            LDS_ADD(&jkl0[(lA0 + l) * WK + kA0 + k], (double)jkl[k * NFL + l].x);
                            if (two) LDS_ADD(&jkl1[(lA1 + l) * WK + kA1 + k], (double)jkl[k * NFL + l].y);
                        }
                }
#endif
#if DO_K
                {
                    const real* dik0 = sDik + ksv[0] * (WI * WK), * dik1 = sDik + ksv[1] * (WI * WK);
                    const real* dil0 = sDil + ksv[0] * (WI * WL), * dil1 = sDil + ksv[1] * (WI * WL);
                    const real* djk0 = sDjk + ksv[0] * (WJ * WK), * djk1 = sDjk + ksv[1] * (WJ * WK);
                    const real* djl0 = sDjl + ksv[0] * (WJ * WL), * djl1 = sDjl + ksv[1] * (WJ * WL);
                    double* kik0 = sKik + ksv[0] * (WI * WK), * kik1 = sKik + ksv[1] * (WI * WK);
                    double* kil0 = sKil + ksv[0] * (WI * WL), * kil1 = sKil + ksv[1] * (WI * WL);
                    double* kjk0 = sKjk + ksv[0] * (WJ * WK), * kjk1 = sKjk + ksv[1] * (WJ * WK);
                    double* kjl0 = sKjl + ksv[0] * (WJ * WL), * kjl1 = sKjl + ksv[1] * (WJ * WL);
                    v2f kjk[NFJ * NFK], kjl[NFJ * NFL], djk[NFJ * NFK], djl[NFJ * NFL];
#pragma unroll
                    for (int j = 0; j < NFJ; j++) {
#pragma unroll
                        for (int k = 0; k < NFK; k++) {
                            kjk[j * NFK + k] = (v2f){0.f, 0.f};
                            djk[j * NFK + k] = (v2f){(float)djk0[(jA0 + j) * WK + kA0 + k], (float)djk1[(jA1 + j) * WK + kA1 + k]};
                        }
#pragma unroll
                        for (int l = 0; l < NFL; l++) {
                            kjl[j * NFL + l] = (v2f){0.f, 0.f};
                            djl[j * NFL + l] = (v2f){(float)djl0[(jA0 + j) * WL + lA0 + l], (float)djl1[(jA1 + j) * WL + lA1 + l]};
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NFI; i++) {
                        v2f kik[NFK], kil[NFL], dik[NFK], dil[NFL];
#pragma unroll
                        for (int k = 0; k < NFK; k++) {
                            kik[k] = (v2f){0.f, 0.f};
                            dik[k] = (v2f){(float)dik0[(iA0 + i) * WK + kA0 + k], (float)dik1[(iA1 + i) * WK + kA1 + k]};
                        }
#pragma unroll
                        for (int l = 0; l < NFL; l++) {
                            kil[l] = (v2f){0.f, 0.f};
                            dil[l] = (v2f){(float)dil0[(iA0 + i) * WL + lA0 + l], (float)dil1[(iA1 + i) * WL + lA1 + l]};
                        }
#pragma unroll
                        for (int j = 0; j < NFJ; j++)
#pragma unroll
                            for (int k = 0; k < NFK; k++)
#pragma unroll
                                for (int l = 0; l < NFL; l++) {
                                    const v2f v = I[((i * NFJ + j) * NFK + k) * NFL + l];
                                    kik[k] += v * djl[j * NFL + l];
                                    kil[l] += v * djk[j * NFK + k];
                                    kjk[j * NFK + k] += v * dil[l];
                                    kjl[j * NFL + l] += v * dik[k];
                                }
#pragma unroll
                        for (int k = 0; k < NFK; k++) {
                            LDS_ADD(&kik0[(iA0 + i) * WK + kA0 + k], (double)kik[k].x);
                            if (two) LDS_ADD(&kik1[(iA1 + i) * WK + kA1 + k], (double)kik[k].y);
                        }
#pragma unroll
                        for (int l = 0; l < NFL; l++) {
                            LDS_ADD(&kil0[(iA0 + i) * WL + lA0 + l], (double)kil[l].x);
                            if (two) LDS_ADD(&kil1[(iA1 + i) * WL + lA1 + l], (double)kil[l].y);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NFJ; j++) {
#pragma unroll
                        for (int k = 0; k < NFK; k++) {
                            LDS_ADD(&kjk0[(jA0 + j) * WK + kA0 + k], (double)kjk[j * NFK + k].x);
                            if (two) LDS_ADD(&kjk1[(jA1 + j) * WK + kA1 + k], (double)kjk[j * NFK + k].y);
                        }
#pragma unroll
                        for (int l = 0; l < NFL; l++) {
                            LDS_ADD(&kjl0[(jA0 + j) * WL + lA0 + l], (double)kjl[j * NFL + l].x);
                            if (two) LDS_ADD(&kjl1[(jA1 + j) * WL + lA1 + l], (double)kjl[j * NFL + l].y);
                        }
                    }
                }
#endif
            }
#undef PK2
#endif  // MIXED
#if ABL & 2
            if (abl_sink == 1.2345e300) sJij[0] = abl_sink;
#endif
#else   // ---------------- row-lane mode
            const int per = (nact + G - 1) / G;
            const int ncomb = npk * npl * npi * npj;
            const int nitem = per * ncomb;            // (step, primitive combination) pairs, flattened
            // phase A of item `m` into buffer m % NBUF
            auto phase_a = [&](const int m, const int rh = 0) {
                const int step = m / ncomb;
                int cmb = m - step * ncomb;
                const int jp = cmb % npj; cmb /= npj;
                const int ip = cmb % npi; cmb /= npi;
                const int lp = cmb % npl;
                const int kp = cmb / npl;
                real* __restrict__ buf = sT + (NBUF > 1 ? (m & 1) : 0) * (G * TRR_SLOT);
#if PAROOT
#if WSYNC
                for (int job = lane; job < GW * NRH; job += 64) {
                    const int sl = job / NRH, rloc = job - sl * NRH, r = rh * NRH + rloc;
                    const int sa = wave_s * GW + sl;
                    if (RSPLIT > 1 && r >= NROOTS) continue;
#else
                for (int job = tid; job < G * NRH; job += TBLOCK) {
                    const int sa = job / NRH, rloc = job - sa * NRH, r = rh * NRH + rloc;
                    if (RSPLIT > 1 && r >= NROOTS) continue;
#endif
                    const int qa = sa * per + step;
                    if (qa >= nact) continue;
                    const int qd = s_act[qa];
                    const int a = qd % TSI, b = (qd / TSI) % TSJ, d = (qd / (TSI * TSJ)) % TSL, c = QC(qd / (TSI * TSJ * TSL), a, b, d);
                    const real* bi = sBas + a * BASIS_STRIDE;
                    const real* bj = sBas + OFF_J + b * BASIS_STRIDE;
                    const real* bk = sBas + OFF_K + c * BASIS_STRIDE;
                    const real* bl = sBas + OFF_L + d * BASIS_STRIDE;
                    const real* pb = sPB + (a * TSJ + b) * 27 + (ip * 3 + jp) * 3;
                    const real* pk = sPK + (c * TSL + d) * 27 + (kp * 3 + lp) * 3;
                    const real ckcl = pk[0], inv_akl = pk[1], akl = pk[2];
                    const real cicj = pb[0], inv_aij = pb[1], aij = pb[2];
                    const real al_akl = bl[5 + 2 * lp] * inv_akl, aj_aij = bj[5 + 2 * jp] * inv_aij;
                    const real rijv[3] = {bj[0] - bi[0], bj[1] - bi[1], bj[2] - bi[2]};
                    const real rklv[3] = {bl[0] - bk[0], bl[1] - bk[1], bl[2] - bk[2]};
                    const real rpqv[3] = {rijv[0] * aj_aij + bi[0] - rklv[0] * al_akl - bk[0],
                                          rijv[1] * aj_aij + bi[1] - rklv[1] * al_akl - bk[1],
                                          rijv[2] * aj_aij + bi[2] - rklv[2] * al_akl - bk[2]};
                    const real rr = rpqv[0] * rpqv[0] + rpqv[1] * rpqv[1] + rpqv[2] * rpqv[2];
                    const real sinv = fast_rsqrt(aij + akl);
                    const real inv = sinv * sinv;
                    const real theta = aij * akl * inv;
                    real t2, wt;
                    rys_root_one(rr, theta, omega, r, cheb_tab, rys_large, t2, wt);
                    const real rt_aa = t2 * inv;
                    const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                    const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                    const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                    const real b00 = real(0.5) * rt_aa;
                    real fac = real(34.98683665524972497);
                    {
                        const int ish = ish0 + a, jsh = jsh0 + b, ksh = ksh0 + c, lsh = lsh0 + d;
                        if (ish == jsh) fac *= real(0.5);
                        if (ksh == lsh) fac *= real(0.5);
                        if (ish == ksh && jsh == lsh) fac *= real(0.5);
                    }
                    const real g0v[3] = {ckcl, fac * cicj * inv_aij * inv_akl * sinv, wt};
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) {
                        const real c0 = rijv[ax] * aj_aij - rt_aij * rpqv[ax];
                        const real cp = rklv[ax] * al_akl + rt_akl * rpqv[ax];
                        real tt[LIJ + 1][LKL + 1];
                        tt[0][0] = g0v[ax];
                        if (LIJ > 0) {
                            tt[1][0] = c0 * g0v[ax];
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
                        real* __restrict__ dst = buf + trr_off(sa) + (rloc * 3 + ax) * NT2;
#if HB
                        bra_hrr_store(tt, rijv[ax], dst);
#else
#pragma unroll
                        for (int q = 0; q <= LIJ; q++)
#pragma unroll
                            for (int cc = 0; cc <= LKL; cc++) dst[q * (LKL + 1) + cc] = tt[q][cc];
#endif
                    }
                }
#else
#if WSYNC
                for (int job = lane; job < GW * 3 * NRH; job += 64) {
                    const int sl = job / (3 * NRH), rem = job - sl * (3 * NRH);
                    const int sa = wave_s * GW + sl;
#else
                for (int job = tid; job < NJOB; job += TBLOCK) {
                    const int sa = job / (3 * NRH), rem = job - sa * (3 * NRH);
#endif
                    const int rloc = rem / 3, ax = rem - rloc * 3, r = rh * NRH + rloc;
                    if (RSPLIT > 1 && r >= NROOTS) continue;
                    const int qa = sa * per + step;
                    if (qa >= nact) continue;
                    const int qd = s_act[qa];
                    const int a = qd % TSI, b = (qd / TSI) % TSJ, d = (qd / (TSI * TSJ)) % TSL, c = QC(qd / (TSI * TSJ * TSL), a, b, d);
                    const real* bi = sBas + a * BASIS_STRIDE;
                    const real* bj = sBas + OFF_J + b * BASIS_STRIDE;
                    const real* bk = sBas + OFF_K + c * BASIS_STRIDE;
                    const real* bl = sBas + OFF_L + d * BASIS_STRIDE;
                    const real* pb = sPB + (a * TSJ + b) * 27 + (ip * 3 + jp) * 3;
                    const real* pk = sPK + (c * TSL + d) * 27 + (kp * 3 + lp) * 3;
                    const real ckcl = pk[0], inv_akl = pk[1], akl = pk[2];
                    const real cicj = pb[0], inv_aij = pb[1], aij = pb[2];
                    const real al_akl = bl[5 + 2 * lp] * inv_akl, aj_aij = bj[5 + 2 * jp] * inv_aij;
                    const real rij0 = bj[0] - bi[0], rij1 = bj[1] - bi[1], rij2 = bj[2] - bi[2];
                    const real rkl0 = bl[0] - bk[0], rkl1 = bl[1] - bk[1], rkl2 = bl[2] - bk[2];
                    const real rpq0 = rij0 * aj_aij + bi[0] - rkl0 * al_akl - bk[0];
                    const real rpq1 = rij1 * aj_aij + bi[1] - rkl1 * al_akl - bk[1];
                    const real rpq2 = rij2 * aj_aij + bi[2] - rkl2 * al_akl - bk[2];
                    const real rr = rpq0 * rpq0 + rpq1 * rpq1 + rpq2 * rpq2;
                    const real sinv = fast_rsqrt(aij + akl);
                    const real inv = sinv * sinv;
                    const real theta = aij * akl * inv;
                    real t2, wt;
                    rys_root_one(rr, theta, omega, r, cheb_tab, rys_large, t2, wt);
                    const real rt_aa = t2 * inv;
                    const real rt_aij = rt_aa * akl, rt_akl = rt_aa * aij;
                    const real b10 = real(0.5) * inv_aij * (real(1) - rt_aij);
                    const real b01 = real(0.5) * inv_akl * (real(1) - rt_akl);
                    const real b00 = real(0.5) * rt_aa;
                    const real rij_a = ax == 0 ? rij0 : ax == 1 ? rij1 : rij2;
                    const real rkl_a = ax == 0 ? rkl0 : ax == 1 ? rkl1 : rkl2;
                    const real rpq_a = ax == 0 ? rpq0 : ax == 1 ? rpq1 : rpq2;
                    const real c0 = rij_a * aj_aij - rt_aij * rpq_a;
                    const real cp = rkl_a * al_akl + rt_akl * rpq_a;
                    real g0;
                    if (ax == 0) g0 = ckcl;
                    else if (ax == 1) {
                        const int ish = ish0 + a, jsh = jsh0 + b, ksh = ksh0 + c, lsh = lsh0 + d;
                        real fac = real(34.98683665524972497);
                        if (ish == jsh) fac *= real(0.5);
                        if (ksh == lsh) fac *= real(0.5);
                        if (ish == ksh && jsh == lsh) fac *= real(0.5);
                        g0 = fac * cicj * inv_aij * inv_akl * sinv;
                    } else g0 = wt;
                    real tt[LIJ + 1][LKL + 1];
                    tt[0][0] = g0;
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
                    real* __restrict__ dst = buf + trr_off(sa) + (rloc * 3 + ax) * NT2;
#if HB
                    bra_hrr_store(tt, rij_a, dst);
#else
#pragma unroll
                    for (int q = 0; q <= LIJ; q++)
#pragma unroll
                        for (int cc = 0; cc <= LKL; cc++) dst[q * (LKL + 1) + cc] = tt[q][cc];
#endif
                }
#endif  // PAROOT
            };

#pragma unroll
            for (int CH = 0; CH < NCH; CH++) {
                if (KW && CH != kpart) continue;          // (KW: this wave's chunk only; the loops of the chunks have the same barriers)
                // register accumulators carried across consecutive quartets of this lane that share the destination block
                double jkl_acc[CW * NFL], kjk_acc[EJ * CW], kjl_acc[EJ * NFL];
#pragma unroll
                for (int e = 0; e < CW * NFL; e++) jkl_acc[e] = 0;
#pragma unroll
                for (int n = 0; n < EJ * CW; n++) kjk_acc[n] = 0;
#pragma unroll
                for (int n = 0; n < EJ * NFL; n++) kjl_acc[n] = 0;

                int item = 0;
                if (NBUF > 1) {
                    phase_a(0);
                    STEP_SYNC();
                }
                for (int step = 0; step < per; step++) {
                    const int qi = slot * per + step;
                    const bool on = lane_on && qi < nact;
                    int a = 0, b = 0, c = 0, d = 0;
                    if (on) {
                        const int qd = s_act[qi];
                        a = qd % TSI; b = (qd / TSI) % TSJ; d = (qd / (TSI * TSJ)) % TSL; c = QC(qd / (TSI * TSJ * TSL), a, b, d);
                    }
                    const real* bi = sBas + a * BASIS_STRIDE;
                    const real* bj = sBas + OFF_J + b * BASIS_STRIDE;
                    const real* bk = sBas + OFF_K + c * BASIS_STRIDE;
                    const real* bl = sBas + OFF_L + d * BASIS_STRIDE;
                    const real rij[3] = {bj[0] - bi[0], bj[1] - bi[1], bj[2] - bi[2]};
                    const real rkl[3] = {bl[0] - bk[0], bl[1] - bk[1], bl[2] - bk[2]};

                    // bra HRR as a weighted sum over TRR rows: g(i,j) = sum_m C(j,m) (Ri-Rj)^(j-m) t[i+m]
                    real wb[3][LJ + 1];
#pragma unroll
                    for (int ax = 0; ax < 3 && !HB && !CJR; ax++) {
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

                    for (int cmb = 0; cmb < ncomb; cmb++, item++) {
                        STAMP(13);
#pragma unroll
                        for (int rh = 0; rh < RSPLIT; rh++) {
                        if (NBUF > 1) {
                            if (item + 1 < nitem) phase_a(item + 1);
                        } else {
                            phase_a(item, rh);
                            STEP_SYNC();
                        }
                        STAMP(10);
                        // ---------------- phase B: row lanes: own bra slice, ket HRR, integral accumulation
                        if (on) {
                            const real* __restrict__ myT = sT + (NBUF > 1 ? (item & 1) : 0) * (G * TRR_SLOT) + trr_off(slot);
#if UNROLL_B == 1
#pragma unroll
#elif UNROLL_B == 2
#pragma unroll 2
#else
#pragma clang loop unroll(disable)
#endif
                            for (int r = 0; r < NRH; r++) {
                                if (RSPLIT > 1 && rh * NRH + r >= NROOTS) break;
#if CJR
                                // bra HRR for every j power and the ket HRR in registers: gk[axis][j][k][l]
                                real gk[3][LJ + 1][LK + 1][LL + 1];
#pragma unroll
                                for (int ax = 0; ax < 3; ax++) {
                                    const real* tp = myT + (r * 3 + ax) * NT2 + ibra[ax] * (LKL + 1);
                                    real h[LJ + 1][LKL + 1];
#pragma unroll
                                    for (int m = 0; m <= LJ; m++)
#pragma unroll
                                        for (int cc = 0; cc <= LKL; cc++) h[m][cc] = tp[m * (LKL + 1) + cc];
#pragma unroll
                                    for (int j = 0; j <= LJ; j++) {
                                        real w[LKL + 1];
#pragma unroll
                                        for (int cc = 0; cc <= LKL; cc++) w[cc] = h[0][cc];
#pragma unroll
                                        for (int l = 0; l <= LL; l++) {
#pragma unroll
                                            for (int k = 0; k <= LK; k++) gk[ax][j][k][l] = w[k];
                                            if (l < LL) {
#pragma unroll
                                                for (int cc = 0; cc < LKL - l; cc++) w[cc] = w[cc + 1] - rkl[ax] * w[cc];
                                            }
                                        }
                                        if (j < LJ) {
#pragma unroll
                                            for (int m = 0; m < LJ - j; m++)
#pragma unroll
                                                for (int cc = 0; cc <= LKL; cc++) h[m][cc] = h[m + 1][cc] - rij[ax] * h[m][cc];
                                        }
                                    }
                                }
#pragma unroll
                                for (int oj = 0; oj < NFJ; oj++)
#pragma unroll
                                    for (int kk = 0; kk < CW; kk++)
#pragma unroll
                                        for (int cl = 0; cl < NFL; cl++) {
                                            const int ck = CH * CW + kk;
                                            acc[(oj * CW + kk) * NFL + cl] += gk[0][TJ.x[oj]][TK.x[ck]][TL.x[cl]] *
                                                                              gk[1][TJ.y[oj]][TK.y[ck]][TL.y[cl]] *
                                                                              gk[2][TJ.z[oj]][TK.z[ck]][TL.z[cl]];
                                        }
                            }
#elif HB
                                if (NJG == 1) {
                                    // every j component in the lane: rows (i_ax, j = 0..LJ) of each axis, ket HRR, gk[axis][j][k][l]
                                    real gk[3][LJ + 1][LK + 1][LL + 1];
#pragma unroll
                                    for (int ax = 0; ax < 3; ax++) {
                                        const real* __restrict__ tp = myT + (r * 3 + ax) * NT2 + ibra[ax] * ((LJ + 1) * (LKL + 1));
#pragma unroll
                                        for (int j = 0; j <= LJ; j++) {
                                            real w[LKL + 1];
#pragma unroll
                                            for (int cc = 0; cc <= LKL; cc++) w[cc] = tp[j * (LKL + 1) + cc];
#pragma unroll
                                            for (int l = 0; l <= LL; l++) {
#pragma unroll
                                                for (int k = 0; k <= LK; k++) gk[ax][j][k][l] = w[k];
                                                if (l < LL) {
#pragma unroll
                                                    for (int cc = 0; cc < LKL - l; cc++) w[cc] = w[cc + 1] - rkl[ax] * w[cc];
                                                }
                                            }
                                        }
                                    }
#pragma unroll
                                    for (int oj = 0; oj < NFJ; oj++)
#pragma unroll
                                        for (int kk = 0; kk < CW; kk++)
#pragma unroll
                                            for (int cl = 0; cl < NFL; cl++) {
                                                const int ck = CH * CW + kk;
                                                acc[(oj * CW + kk) * NFL + cl] += gk[0][TJ.x[oj]][TK.x[ck]][TL.x[cl]] *
                                                                                  gk[1][TJ.y[oj]][TK.y[ck]][TL.y[cl]] *
                                                                                  gk[2][TJ.z[oj]][TK.z[ck]][TL.z[cl]];
                                            }
                                } else {
                                    // lane = (ci, j-group): per j component one row per axis (run-time row offsets), ket HRR, products
#pragma unroll
                                    for (int e = 0; e < EJ; e++) {
                                        real gk[3][LK + 1][LL + 1];
#pragma unroll
                                        for (int ax = 0; ax < 3; ax++) {
                                            const real* __restrict__ tp = myT + (r * 3 + ax) * NT2 + hrow[e][ax];
                                            real w[LKL + 1];
#pragma unroll
                                            for (int cc = 0; cc <= LKL; cc++) w[cc] = tp[cc];
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
                                                acc[(e * CW + kk) * NFL + cl] += gk[0][TK.x[ck]][TL.x[cl]] * gk[1][TK.y[ck]][TL.y[cl]] *
                                                                                 gk[2][TK.z[ck]][TL.z[cl]];
                                            }
                                    }
                                }
                            }
#else
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
#endif
                        }
                        STAMP(11);
                        STEP_SYNC();
                        STAMP(12);
                        }
                    }

                    // ---------------- contraction with the density sub-blocks.  All LDS reads and arithmetic first, every
                    // LDS atomic of the step at the end: the DS queue of a CU is in order, so a read issued behind a
                    // same-address f64 atomic waits for its serialised lanes (~9 cycles each, tools/micro/lds_atomic_bench.hip);
                    // issued last, the atomics drain under the next step's arithmetic.
                    const int iA = a * NFI + ci, jA = b * NFJ + cj;
                    const int kb = c * NFK + CH * CW, lbs = d * NFL;
                    // does the next quartet of this lane still belong to the same (k,l) / (j,k) / (j,l) block?
                    bool keep_kl = false, keep_jk = false, keep_jl = false;
                    if (on && step + 1 < per && qi + 1 < nact) {
                        const int q2 = s_act[qi + 1];
                        const int a2 = q2 % TSI, b2 = (q2 / TSI) % TSJ, d2 = (q2 / (TSI * TSJ)) % TSL;
                        const int c2 = QC(q2 / (TSI * TSJ * TSL), a2, b2, d2);
                        keep_kl = c2 == c && d2 == d;
                        keep_jk = b2 == b && c2 == c;
                        keep_jl = b2 == b && d2 == d;
                    }
#if CJR
#if NDM > 1
#pragma unroll 1
                    for (int dmi = 0; dmi < NDM && dmi < ndm_grp; dmi++) {
#else
                    {
                    constexpr int dmi = 0;
#endif
                    // this density matrix's tiles (row-lane mode: one ket slot); the code below is written for one matrix
                    const real* const sDij_v = sDij + dmi * (WJ * WI); const real* const sDkl_v = sDkl + dmi * (WL * WK);
                    const real* const sDik_v = sDik + dmi * (WI * WK); const real* const sDil_v = sDil + dmi * (WI * WL);
                    const real* const sDjk_v = sDjk + dmi * (WJ * WK); const real* const sDjl_v = sDjl + dmi * (WJ * WL);
                    double* const sJij_v = sJij + dmi * (WJ * WI); double* const sJkl_v = sJkl + dmi * (WL * WK);
                    double* const sKik_v = sKik + dmi * (WI * WK); double* const sKil_v = sKil + dmi * (WI * WL);
                    double* const sKjk_v = sKjk + dmi * (WJ * WK); double* const sKjl_v = sKjl + dmi * (WJ * WL);
                    {
                    const real* const sDij = sDij_v; const real* const sDkl = sDkl_v; const real* const sDik = sDik_v;
                    const real* const sDil = sDil_v; const real* const sDjk = sDjk_v; const real* const sDjl = sDjl_v;
                    double* const sJij = sJij_v; double* const sJkl = sJkl_v; double* const sKik = sKik_v;
                    double* const sKil = sKil_v; double* const sKjk = sKjk_v; double* const sKjl = sKjl_v;
#if ORED
#pragma unroll
                    for (int e = 0; e < CW * NFL; e++) jkl_acc[e] = 0;
#pragma unroll
                    for (int n = 0; n < EJ * CW; n++) kjk_acc[n] = 0;
#pragma unroll
                    for (int n = 0; n < EJ * NFL; n++) kjl_acc[n] = 0;
#endif
                    if (on) {
                        // lane = bra component ci, registers = (cj, k, l): J_ij, K_ik, K_il are complete in the lane
                        const int jA0 = b * NFJ;
                        real s_ij[NFJ], s_ik[CW], s_il[NFL];
#pragma unroll
                        for (int kk = 0; kk < CW; kk++) s_ik[kk] = 0;
#pragma unroll
                        for (int cl = 0; cl < NFL; cl++) s_il[cl] = 0;
#pragma unroll
                        for (int oj = 0; oj < NFJ; oj++) {
                            real sj = 0;
#if DO_J
                            const real dij = sDij[(jA0 + oj) * WI + iA];
#endif
#pragma unroll
                            for (int kk = 0; kk < CW; kk++) {
#if DO_K
                                const real djk = sDjk[(jA0 + oj) * WK + kb + kk], dik = sDik[iA * WK + kb + kk];
                                real s_jk = 0;
#endif
#pragma unroll
                                for (int cl = 0; cl < NFL; cl++) {
                                    const real v = acc[(oj * CW + kk) * NFL + cl];
#if DO_J
                                    sj += v * sDkl[(lbs + cl) * WK + kb + kk];
                                    jkl_acc[kk * NFL + cl] += (double)(v * dij);
#endif
#if DO_K
                                    s_ik[kk] += v * sDjl[(jA0 + oj) * WL + lbs + cl];
                                    s_il[cl] += v * djk;
                                    s_jk += v * sDil[iA * WL + lbs + cl];
                                    kjl_acc[oj * NFL + cl] += (double)(v * dik);
#endif
                                }
#if DO_K
                                kjk_acc[oj * CW + kk] += (double)s_jk;
#endif
                            }
                            s_ij[oj] = sj;
                        }
                        // ---- atomics of the step
#if DO_J
#pragma unroll
                        for (int oj = 0; oj < NFJ; oj++) lds_add(&sJij[(jA0 + oj) * WI + iA], (double)s_ij[oj]);
                        if (!ORED && !keep_kl) {
#pragma unroll
                            for (int kk = 0; kk < CW; kk++)
#pragma unroll
                                for (int cl = 0; cl < NFL; cl++) {
                                    lds_add(&sJkl[(lbs + cl) * WK + kb + kk], jkl_acc[kk * NFL + cl]);
                                    jkl_acc[kk * NFL + cl] = 0;
                                }
                        }
#endif
#if DO_K
#pragma unroll
                        for (int kk = 0; kk < CW; kk++) lds_add(&sKik[iA * WK + kb + kk], (double)s_ik[kk]);
#pragma unroll
                        for (int cl = 0; cl < NFL; cl++) lds_add(&sKil[iA * WL + lbs + cl], (double)s_il[cl]);
                        if (!ORED && !keep_jk) {
#pragma unroll
                            for (int oj = 0; oj < NFJ; oj++)
#pragma unroll
                                for (int kk = 0; kk < CW; kk++) {
                                    lds_add(&sKjk[(jA0 + oj) * WK + kb + kk], kjk_acc[oj * CW + kk]);
                                    kjk_acc[oj * CW + kk] = 0;
                                }
                        }
                        if (!ORED && !keep_jl) {
#pragma unroll
                            for (int oj = 0; oj < NFJ; oj++)
#pragma unroll
                                for (int cl = 0; cl < NFL; cl++) {
                                    lds_add(&sKjl[(jA0 + oj) * WL + lbs + cl], kjl_acc[oj * NFL + cl]);
                                    kjl_acc[oj * NFL + cl] = 0;
                                }
                        }
#endif
                    }
#if ORED
                    STAMP(14);
                    // ---- owner reduction of the values summed over the lanes of a quartet (J_kl, K_jk, K_jl of this step).
                    //      Scratch of this wave: red[row][lane] inside its own part of the TRR array (dead until the next
                    //      phase A).  Row r of a pass is summed by NH owner lanes, each over the quartets qs = h, h + NH, ...
                    {
                        double* __restrict__ red = (double*)(sT + wave_s * WREG + kpart * WPART);
                        const int ox = lane % RG, oh = lane / RG;
                        const double* __restrict__ row = red + ox * RSTR + oh * T;
#if ORED_HOIST
                        // the owner's quartets and their destination blocks do not change from pass to pass: decoded once per step
                        constexpr int KN = (GW + NH - 1) / NH;
                        bool okk[KN];
                        int o0[KN], o1[KN], o2[KN];
#pragma unroll
                        for (int k = 0; k < KN; k++) {
                            const int qs = oh + k * NH;
                            const int qo = (wave_s * GW + qs) * per + step;
                            okk[k] = oh < NH && qs < GW && qo < nact;
                            const int qd2 = s_act[okk[k] ? qo : 0];
                            const int a2 = qd2 % TSI, b2 = (qd2 / TSI) % TSJ, d2 = (qd2 / (TSI * TSJ)) % TSL;
                            const int c2 = QC(qd2 / (TSI * TSJ * TSL), a2, b2, d2);
                            o0[k] = (d2 * NFL) * WK + c2 * NFK;
                            o1[k] = (b2 * NFJ) * WK + c2 * NFK;
                            o2[k] = (b2 * NFJ) * WL + d2 * NFL;
                        }
#pragma unroll
                        for (int ps = 0; ps < NPASS; ps++) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int r = 0; r < RG; r++) {
                                const int e = ps * RG + r;                  // (compile-time)
                                if (e < NPART)
                                    red[r * RSTR + lane] = e < NE0 ? jkl_acc[e < NE0 ? e : 0]
                                                         : e < NE0 + NE1 ? kjk_acc[(e >= NE0 && e < NE0 + NE1) ? e - NE0 : 0]
                                                                         : kjl_acc[e >= NE0 + NE1 ? e - NE0 - NE1 : 0];
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            const int e = ps * RG + ox;
                            int blk = 2, off;
                            {
                                const int e1 = e - NE0, e2 = e - NE0 - NE1;
                                off = (e2 / NFL) * WL + e2 % NFL;                                  // K_jl[oj][cl]
                                if (e < NE0 + NE1) { blk = 1; off = (e1 / CW) * WK + CH * CW + e1 % CW; }     // K_jk[oj][kk]
                                if (e < NE0) { blk = 0; off = (e % NFL) * WK + CH * CW + e / NFL; }           // J_kl[cl][kk]  (e = kk * NFL + cl)
                            }
                            double* const base = blk == 0 ? sJkl : blk == 1 ? sKjk : sKjl;
                            // every read of the pass first (one LDS round trip for all the owner's quartets), pairwise sums, atomics last
                            double v[KN][T];
#pragma unroll
                            for (int k = 0; k < KN; k++)
#pragma unroll
                                for (int u = 0; u < T; u++) v[k][u] = row[(oh + k * NH < GW ? k * NH * T : 0) + u];
#pragma unroll
                            for (int k = 0; k < KN; k++) {
#pragma unroll
                                for (int w = 1; w < T; w *= 2)
#pragma unroll
                                    for (int u = 0; u + w < T; u += 2 * w) v[k][u] += v[k][u + w];
                            }
#pragma unroll
                            for (int k = 0; k < KN; k++)
                                if (okk[k] && e < NPART) lds_add(base + (blk == 0 ? o0[k] : blk == 1 ? o1[k] : o2[k]) + off, v[k][0]);
                        }
#else
#pragma unroll
                        for (int ps = 0; ps < NPASS; ps++) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int r = 0; r < RG; r++) {
                                const int e = ps * RG + r;                  // (compile-time)
                                if (e < NPART)
                                    red[r * RSTR + lane] = e < NE0 ? jkl_acc[e < NE0 ? e : 0]
                                                         : e < NE0 + NE1 ? kjk_acc[(e >= NE0 && e < NE0 + NE1) ? e - NE0 : 0]
                                                                         : kjl_acc[e >= NE0 + NE1 ? e - NE0 - NE1 : 0];
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            const int e = ps * RG + ox;
                            const bool own = oh < NH && e < NPART;
                            int blk = 2, off;
                            {
                                const int e1 = e - NE0, e2 = e - NE0 - NE1;
                                off = (e2 / NFL) * WL + e2 % NFL;                                  // K_jl[oj][cl]
                                if (e < NE0 + NE1) { blk = 1; off = (e1 / CW) * WK + CH * CW + e1 % CW; }     // K_jk[oj][kk]
                                if (e < NE0) { blk = 0; off = (e % NFL) * WK + CH * CW + e / NFL; }           // J_kl[cl][kk]  (e = kk * NFL + cl)
                            }
#pragma unroll
                            for (int k = 0; k < (GW + NH - 1) / NH; k++) {
                                const int qs = oh + k * NH;
                                const int qo = (wave_s * GW + qs) * per + step;
                                if (own && qs < GW && qo < nact) {
                                    double v[T];
#pragma unroll
                                    for (int u = 0; u < T; u++) v[u] = row[k * NH * T + u];
                                    const int qd2 = s_act[qo];
                                    const int a2 = qd2 % TSI, b2 = (qd2 / TSI) % TSJ, d2 = (qd2 / (TSI * TSJ)) % TSL;
                                    const int c2 = QC(qd2 / (TSI * TSJ * TSL), a2, b2, d2);
                                    double* dst = blk == 0 ? sJkl + (d2 * NFL) * WK + c2 * NFK + off
                                                : blk == 1 ? sKjk + (b2 * NFJ) * WK + c2 * NFK + off
                                                           : sKjl + (b2 * NFJ) * WL + d2 * NFL + off;
                                    double sum = 0;
#pragma unroll
                                    for (int u = 0; u < T; u++) sum += v[u];
                                    lds_add(dst, sum);
                                }
                            }
                        }
#endif
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
#if !WSYNC
                    __syncthreads();          // the next phase A overwrites the scratch (it aliases the TRR array)
#endif
#endif
                    }
                    }
                }
            }
#else
#if NDM > 1
#pragma unroll 1
                    for (int dmi = 0; dmi < NDM && dmi < ndm_grp; dmi++) {
#else
                    {
                    constexpr int dmi = 0;
#endif
                    // this density matrix's tiles (row-lane mode: one ket slot); the code below is written for one matrix
                    const real* const sDij_v = sDij + dmi * (WJ * WI); const real* const sDkl_v = sDkl + dmi * (WL * WK);
                    const real* const sDik_v = sDik + dmi * (WI * WK); const real* const sDil_v = sDil + dmi * (WI * WL);
                    const real* const sDjk_v = sDjk + dmi * (WJ * WK); const real* const sDjl_v = sDjl + dmi * (WJ * WL);
                    double* const sJij_v = sJij + dmi * (WJ * WI); double* const sJkl_v = sJkl + dmi * (WL * WK);
                    double* const sKik_v = sKik + dmi * (WI * WK); double* const sKil_v = sKil + dmi * (WI * WL);
                    double* const sKjk_v = sKjk + dmi * (WJ * WK); double* const sKjl_v = sKjl + dmi * (WJ * WL);
                    {
                    const real* const sDij = sDij_v; const real* const sDkl = sDkl_v; const real* const sDik = sDik_v;
                    const real* const sDil = sDil_v; const real* const sDjk = sDjk_v; const real* const sDjl = sDjl_v;
                    double* const sJij = sJij_v; double* const sJkl = sJkl_v; double* const sKik = sKik_v;
                    double* const sKil = sKil_v; double* const sKjk = sKjk_v; double* const sKjl = sKjl_v;
                    real s_ij = 0, s_ik[CW], kil[NFL];
                    if (USE_ORED) {          // step-local sums; lanes without a quartet contribute zeros to the owner sums
#pragma unroll
                        for (int e = 0; e < CW * NFL; e++) jkl_acc[e] = 0;
#pragma unroll
                        for (int kk = 0; kk < EJ * CW; kk++) kjk_acc[kk] = 0;
#pragma unroll
                        for (int cl = 0; cl < EJ * NFL; cl++) kjl_acc[cl] = 0;
#pragma unroll
                        for (int kk = 0; kk < CW; kk++) s_ik[kk] = 0;
#pragma unroll
                        for (int cl = 0; cl < NFL; cl++) kil[cl] = 0;
                    }
#if HB
                    if (on) {
                        // lane = (ci, j-group), registers = (e, k, l): J_ij is complete in the lane; K_ik / K_il are sums over the lanes of
                        // one ci (complete when NJG == 1), J_kl / K_jk / K_jl over the lanes of the quartet / of one j-group
                        real s_ije[EJ];
#pragma unroll
                        for (int e = 0; e < EJ; e++) {
                            real sj = 0;
#if DO_J
                            const real dij = sDij[(jA + e) * WI + iA];
#endif
#pragma unroll
                            for (int kk = 0; kk < CW; kk++) {
#if DO_K
                                const real djk = sDjk[(jA + e) * WK + kb + kk], dik = sDik[iA * WK + kb + kk];
                                real s_jk = 0;
#endif
#pragma unroll
                                for (int cl = 0; cl < NFL; cl++) {
                                    const real v = acc[(e * CW + kk) * NFL + cl];
#if DO_J
                                    sj += v * sDkl[(lbs + cl) * WK + kb + kk];
                                    jkl_acc[kk * NFL + cl] += (double)(v * dij);
#endif
#if DO_K
                                    s_ik[kk] += v * sDjl[(jA + e) * WL + lbs + cl];
                                    kil[cl] += v * djk;
                                    s_jk += v * sDil[iA * WL + lbs + cl];
                                    kjl_acc[e * NFL + cl] += (double)(v * dik);
#endif
                                }
#if DO_K
                                kjk_acc[e * CW + kk] += (double)s_jk;
#endif
                            }
                            s_ije[e] = sj;
                        }
#if DO_J
#pragma unroll
                        for (int e = 0; e < EJ; e++) lds_add(&sJij[(jA + e) * WI + iA], (double)s_ije[e]);
#endif
#if DO_K
                        if (NJG == 1) {
#pragma unroll
                            for (int kk = 0; kk < CW; kk++) lds_add(&sKik[iA * WK + kb + kk], (double)s_ik[kk]);
#pragma unroll
                            for (int cl = 0; cl < NFL; cl++) lds_add(&sKil[iA * WL + lbs + cl], (double)kil[cl]);
                        }
#endif
                    }
#else
                    if (on) {
#if DO_J
                        {
                            const real dij = sDij[jA * WI + iA];
#pragma unroll
                            for (int kk = 0; kk < CW; kk++)
#pragma unroll
                                for (int cl = 0; cl < NFL; cl++) {
                                    const real v = acc[kk * NFL + cl];
                                    s_ij += v * sDkl[(lbs + cl) * WK + kb + kk];
                                    jkl_acc[kk * NFL + cl] += (double)(v * dij);
                                }
                        }
#endif
#if DO_K
#pragma unroll
                        for (int cl = 0; cl < NFL; cl++) kil[cl] = 0;
#pragma unroll
                        for (int kk = 0; kk < CW; kk++) {
                            real sk = 0, s_jk = 0;
                            const real djk = sDjk[jA * WK + kb + kk], dik = sDik[iA * WK + kb + kk];
#pragma unroll
                            for (int cl = 0; cl < NFL; cl++) {
                                const real v = acc[kk * NFL + cl];
                                sk += v * sDjl[jA * WL + lbs + cl];
                                s_jk += v * sDil[iA * WL + lbs + cl];
                                kil[cl] += v * djk;
                                kjl_acc[cl] += (double)(v * dik);
                            }
                            s_ik[kk] = sk;
                            kjk_acc[kk] += (double)s_jk;
                        }
#endif
                        // ---- atomics of the step
#if DO_J
                        lds_add(&sJij[jA * WI + iA], (double)s_ij);
                        if (!USE_ORED && !keep_kl) {
#pragma unroll
                            for (int kk = 0; kk < CW; kk++)
#pragma unroll
                                for (int cl = 0; cl < NFL; cl++) {
                                    lds_add(&sJkl[(lbs + cl) * WK + kb + kk], jkl_acc[kk * NFL + cl]);
                                    jkl_acc[kk * NFL + cl] = 0;
                                }
                        }
#endif
#if DO_K
                        if (!USE_ORED) {
#pragma unroll
                            for (int kk = 0; kk < CW; kk++) lds_add(&sKik[iA * WK + kb + kk], (double)s_ik[kk]);
#pragma unroll
                            for (int cl = 0; cl < NFL; cl++) lds_add(&sKil[iA * WL + lbs + cl], (double)kil[cl]);
                            if (!keep_jk) {
#pragma unroll
                                for (int kk = 0; kk < CW; kk++) { lds_add(&sKjk[jA * WK + kb + kk], kjk_acc[kk]); kjk_acc[kk] = 0; }
                            }
                            if (!keep_jl) {
#pragma unroll
                                for (int cl = 0; cl < NFL; cl++) { lds_add(&sKjl[jA * WL + lbs + cl], kjl_acc[cl]); kjl_acc[cl] = 0; }
                            }
                        }
#endif
                    }
#endif  // HB
                    STAMP(14);
                    if (USE_ORED) {
                        // ---- owner reduction, lane = (ci, cj) form (HB: lane = (ci, j-group), NJG groups of EJ components): every output but J_ij is a sum over lanes of the quartet --
                        //      J_kl over all T lanes (type A), K_jk / K_jl over the NFI lanes of one cj (type B), K_ik / K_il over the
                        //      NFJ lanes of one ci (type C).  Same scratch as above: red[row][lane] of this wave.
                        constexpr int rg = RG > 0 ? RG : 1;
                        constexpr int OB = NE0, OC = NE0 + NE1 + NE2;
                        double* __restrict__ red = (double*)(sT + wave_s * WREG + kpart * WPART);
#if ORED_HOIST
                        // Owner lanes are tied to ONE quartet slot of the wave (lane % GW) for the whole step: its destination blocks are decoded
                        // once; lane / GW enumerates the (row, group) sums of a pass.  Per pass: every read first, pairwise sums, atomics last.
                        constexpr int gw = GW > 0 ? GW : 1;
                        constexpr int NLQ = 64 / gw;                      // owner lanes per quartet slot
                        const int oq = lane % gw, om = lane / gw;
                        const int qo_ = (wave_s * gw + oq) * per + step;
                        const bool oq_ok = om < NLQ && qo_ < nact;
                        int bJkl, bKjk, bKjl, bKik, bKil;
                        {
                            const int qd2 = s_act[oq_ok ? qo_ : 0];
                            const int a2 = qd2 % TSI, b2 = (qd2 / TSI) % TSJ, d2 = (qd2 / (TSI * TSJ)) % TSL;
                            const int c2 = QC(qd2 / (TSI * TSJ * TSL), a2, b2, d2);
                            bJkl = (d2 * NFL) * WK + c2 * NFK + CH * CW;
                            bKjk = (b2 * NFJ) * WK + c2 * NFK + CH * CW;
                            bKjl = (b2 * NFJ) * WL + d2 * NFL;
                            bKik = (a2 * NFI) * WK + c2 * NFK + CH * CW;
                            bKil = (a2 * NFI) * WL + d2 * NFL;
                        }
                        const double* __restrict__ qsrc = red + oq * T;
#pragma unroll
                        for (int ps = 0; ps < NPASS; ps++) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int r = 0; r < rg; r++) {
                                const int e = ps * rg + r;                  // (compile-time)
                                if (e < NPART) {
                                    double val;
                                    if (e < OB) val = jkl_acc[e < OB ? e : 0];
                                    else if (e < OB + NE1) val = kjk_acc[(e >= OB && e < OB + NE1) ? e - OB : 0];
                                    else if (e < OC) val = kjl_acc[(e >= OB + NE1 && e < OC) ? e - OB - NE1 : 0];
                                    else if (e < OC + NE3) val = (double)s_ik[(e >= OC && e < OC + NE3) ? e - OC : 0];
                                    else val = (double)kil[e >= OC + NE3 ? e - OC - NE3 : 0];
                                    red[r * RSTR + lane] = val;
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            constexpr int p_lo = 0;                          // (rows of a pass start at scratch row 0)
                            const int g_lo = ps * rg, g_hi = (ps * rg + rg < NPART) ? ps * rg + rg : NPART;
                            // rows of this pass by type (compile-time bounds)
                            const int a_lo = g_lo, a_hi = g_hi < OB ? g_hi : OB, na = a_hi > a_lo ? a_hi - a_lo : 0;
                            const int b_lo = g_lo > OB ? g_lo : OB, b_hi = g_hi < OC ? g_hi : OC, nb = b_hi > b_lo ? b_hi - b_lo : 0;
                            const int c_lo = g_lo > OC ? g_lo : OC, c_hi = g_hi, nc = c_hi > c_lo ? c_hi - c_lo : 0;
                            constexpr int KA = (rg + NLQ - 1) / NLQ, KB = (rg * NJG + NLQ - 1) / NLQ, KC = (rg * NFI + NLQ - 1) / NLQ;
                            double va[KA][T], vb[KB][NFI], vc[KC][NJG];
                            // ---- reads
#pragma unroll
                            for (int k = 0; k < KA; k++) {
                                if (k * NLQ >= na) break;
                                const int m = om + k * NLQ, rr = m < na ? m : 0;
                                const double* __restrict__ src = qsrc + (a_lo + rr - g_lo + p_lo) * RSTR;
#pragma unroll
                                for (int u = 0; u < T; u++) va[k][u] = src[u];
                            }
#pragma unroll
                            for (int k = 0; k < KB; k++) {
                                if (k * NLQ >= nb * NJG) break;
                                const int m0 = om + k * NLQ, m = m0 < nb * NJG ? m0 : 0;
                                const int nbd = nb > 0 ? nb : 1;
                                const int rr = m % nbd, grp = m / nbd;
                                const double* __restrict__ src = qsrc + (b_lo + rr - g_lo + p_lo) * RSTR + grp;
#pragma unroll
                                for (int u = 0; u < NFI; u++) vb[k][u] = src[u * NJG];
                            }
#pragma unroll
                            for (int k = 0; k < KC; k++) {
                                if (k * NLQ >= nc * NFI) break;
                                const int m0 = om + k * NLQ, m = m0 < nc * NFI ? m0 : 0;
                                const int ncd = nc > 0 ? nc : 1;
                                const int rr = m % ncd, grp = m / ncd;
                                const double* __restrict__ src = qsrc + (c_lo + rr - g_lo + p_lo) * RSTR + grp * NJG;
#pragma unroll
                                for (int u = 0; u < NJG; u++) vc[k][u] = src[u];
                            }
                            // ---- pairwise sums + atomics
#pragma unroll
                            for (int k = 0; k < KA; k++) {
                                if (k * NLQ >= na) break;
#pragma unroll
                                for (int w = 1; w < T; w *= 2)
#pragma unroll
                                    for (int u = 0; u + w < T; u += 2 * w) va[k][u] += va[k][u + w];
                                const int m = om + k * NLQ, e = a_lo + m;                    // J_kl[cl][kk], e = kk * NFL + cl
                                if (oq_ok && m < na) lds_add(sJkl + bJkl + (e % NFL) * WK + e / NFL, va[k][0]);
                            }
#pragma unroll
                            for (int k = 0; k < KB; k++) {
                                if (k * NLQ >= nb * NJG) break;
#pragma unroll
                                for (int w = 1; w < NFI; w *= 2)
#pragma unroll
                                    for (int u = 0; u + w < NFI; u += 2 * w) vb[k][u] += vb[k][u + w];
                                const int m = om + k * NLQ;
                                const int nbd = nb > 0 ? nb : 1;
                                const int rr = m % nbd, grp = m / nbd;
                                const int eb = b_lo + rr - OB, eb2 = eb - NE1;    // K_jk[e][kk] (e * CW + kk) or K_jl[e][cl] (NE1 + e * NFL + cl) of j component grp * EJ + e
                                double* dst = eb < NE1 ? sKjk + bKjk + (grp * EJ + eb / CW) * WK + eb % CW
                                                       : sKjl + bKjl + (grp * EJ + eb2 / NFL) * WL + eb2 % NFL;
                                if (oq_ok && m < nb * NJG) lds_add(dst, vb[k][0]);
                            }
#pragma unroll
                            for (int k = 0; k < KC; k++) {
                                if (k * NLQ >= nc * NFI) break;
#pragma unroll
                                for (int w = 1; w < NJG; w *= 2)
#pragma unroll
                                    for (int u = 0; u + w < NJG; u += 2 * w) vc[k][u] += vc[k][u + w];
                                const int m = om + k * NLQ;
                                const int ncd = nc > 0 ? nc : 1;
                                const int rr = m % ncd, grp = m / ncd;
                                const int ec = c_lo + rr - OC;
                                double* dst = ec < NE3 ? sKik + bKik + grp * WK + ec : sKil + bKil + grp * WL + (ec - NE3);
                                if (oq_ok && m < nc * NFI) lds_add(dst, vc[k][0]);
                            }
                        }
#else
#pragma unroll
                        for (int ps = 0; ps < NPASS; ps++) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int r = 0; r < rg; r++) {
                                const int e = ps * rg + r;                  // (compile-time)
                                if (e < NPART) {
                                    double val;
                                    if (e < OB) val = jkl_acc[e < OB ? e : 0];
                                    else if (e < OB + NE1) val = kjk_acc[(e >= OB && e < OB + NE1) ? e - OB : 0];
                                    else if (e < OC) val = kjl_acc[(e >= OB + NE1 && e < OC) ? e - OB - NE1 : 0];
                                    else if (e < OC + NE3) val = (double)s_ik[(e >= OC && e < OC + NE3) ? e - OC : 0];
                                    else val = (double)kil[e >= OC + NE3 ? e - OC - NE3 : 0];
                                    red[r * RSTR + lane] = val;
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            const int p_lo = ps * rg, p_hi = (ps * rg + rg < NPART) ? ps * rg + rg : NPART;
                            // ---- type A rows of this pass: J_kl
                            {
                                const int lo = p_lo, hi = p_hi < OB ? p_hi : OB, n = hi - lo;
                                if (n > 0) {
                                    const int nd = n > 0 ? n : 1;
#pragma unroll
                                    for (int k = 0; k < (nd * GW + 63) / 64; k++) {
                                        const int task = lane + 64 * k, rr = task % nd, qs = task / nd;
                                        const int qo = (wave_s * GW + qs) * per + step;
                                        if (task < n * GW && qo < nact) {
                                            const double* __restrict__ src = red + (lo + rr - p_lo) * RSTR + qs * T;
                                            double v[T];
#pragma unroll
                                            for (int u = 0; u < T; u++) v[u] = src[u];
                                            const int qd2 = s_act[qo];
                                            const int a2 = qd2 % TSI, b2 = (qd2 / TSI) % TSJ, d2 = (qd2 / (TSI * TSJ)) % TSL;
                                            const int c2 = QC(qd2 / (TSI * TSJ * TSL), a2, b2, d2);
                                            const int e = lo + rr;
                                            double sum = 0;
#pragma unroll
                                            for (int u = 0; u < T; u++) sum += v[u];
                                            lds_add(sJkl + (d2 * NFL + e % NFL) * WK + c2 * NFK + CH * CW + e / NFL, sum);
                                        }
                                    }
                                }
                            }
                            // ---- type B rows: K_jk [CW], K_jl [NFL]; group = cj, members = the NFI lanes ci * NFJ + cj
                            {
                                const int lo = p_lo > OB ? p_lo : OB, hi = p_hi < OC ? p_hi : OC, n = hi - lo;
                                if (n > 0) {
                                    const int nd = n > 0 ? n : 1;
#pragma unroll
                                    for (int k = 0; k < (nd * GW * NJG + 63) / 64; k++) {
                                        const int task = lane + 64 * k, rr = task % nd, rest = task / nd;
                                        const int grp = rest % NJG, qs = rest / NJG;
                                        const int qo = (wave_s * GW + qs) * per + step;
                                        if (task < n * GW * NJG && qo < nact) {
                                            const double* __restrict__ src = red + (lo + rr - p_lo) * RSTR + qs * T + grp;
                                            double v[NFI];
#pragma unroll
                                            for (int u = 0; u < NFI; u++) v[u] = src[u * NJG];
                                            const int qd2 = s_act[qo];
                                            const int a2 = qd2 % TSI, b2 = (qd2 / TSI) % TSJ, d2 = (qd2 / (TSI * TSJ)) % TSL;
                                            const int c2 = QC(qd2 / (TSI * TSJ * TSL), a2, b2, d2);
                                            const int eb = lo + rr - OB;
                                            double sum = 0;
#pragma unroll
                                            for (int u = 0; u < NFI; u++) sum += v[u];
                                            // row eb = K_jk[e][kk] (e * CW + kk) or K_jl[e][cl] (NE1 + e * NFL + cl) of j component grp * EJ + e
                                            const int eb2 = eb - NE1;
                                            double* dst = eb < NE1 ? sKjk + (b2 * NFJ + grp * EJ + eb / CW) * WK + c2 * NFK + CH * CW + eb % CW
                                                                   : sKjl + (b2 * NFJ + grp * EJ + eb2 / NFL) * WL + d2 * NFL + eb2 % NFL;
                                            lds_add(dst, sum);
                                        }
                                    }
                                }
                            }
                            // ---- type C rows: K_ik [CW], K_il [NFL]; group = ci, members = the NFJ lanes ci * NFJ + cj
                            {
                                const int lo = p_lo > OC ? p_lo : OC, hi = p_hi, n = hi - lo;
                                if (n > 0) {
                                    const int nd = n > 0 ? n : 1;
#pragma unroll
                                    for (int k = 0; k < (nd * GW * NFI + 63) / 64; k++) {
                                        const int task = lane + 64 * k, rr = task % nd, rest = task / nd;
                                        const int grp = rest % NFI, qs = rest / NFI;
                                        const int qo = (wave_s * GW + qs) * per + step;
                                        if (task < n * GW * NFI && qo < nact) {
                                            const double* __restrict__ src = red + (lo + rr - p_lo) * RSTR + qs * T + grp * NJG;
                                            double v[NJG];
#pragma unroll
                                            for (int u = 0; u < NJG; u++) v[u] = src[u];
                                            const int qd2 = s_act[qo];
                                            const int a2 = qd2 % TSI, b2 = (qd2 / TSI) % TSJ, d2 = (qd2 / (TSI * TSJ)) % TSL;
                                            const int c2 = QC(qd2 / (TSI * TSJ * TSL), a2, b2, d2);
                                            const int ec = lo + rr - OC;
                                            double sum = 0;
#pragma unroll
                                            for (int u = 0; u < NJG; u++) sum += v[u];
                                            double* dst = ec < NE3 ? sKik + (a2 * NFI + grp) * WK + c2 * NFK + CH * CW + ec
                                                                   : sKil + (a2 * NFI + grp) * WL + d2 * NFL + (ec - NE3);
                                            lds_add(dst, sum);
                                        }
                                    }
                                }
                            }
                        }
#endif
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
#if !WSYNC
                        __syncthreads();          // the next phase A overwrites the scratch (it aliases the TRR array)
#endif
                    }
                    }
                    }
                }
            }
#endif  // CJR
#endif  // TILE_1Q
            STAMP(6);
            __syncthreads();
            STAMP(7);
            int tid_f = tid;
            asm volatile("" : "+v"(tid_f));
#define tid tid_f
            {
            // ---- ket-dependent Fock sub-blocks of this tile pair: one coalesced pass of global f64 atomics each
#if KARG_RELOAD
            const KArgs AS4* kf = kargs();
            const int nao = kf->nao;
            const size_t nao2 = (size_t)nao * nao;
            double* __restrict__ vj = kf->vj;
            double* __restrict__ vk = kf->vk;
#endif
#if FLUSH_UNROLL == 2
            {
#if DO_J
                FlushRegs<WL, WK> fkl[NKS];
#endif
#if DO_K
                FlushRegs<WI, WK> fik[NKS];
                FlushRegs<WI, WL> fil[NKS];
                FlushRegs<WJ, WK> fjk[NKS];
                FlushRegs<WJ, WL> fjl[NKS];
#endif
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    if (!kval[ks]) continue;
#if DO_J
                    flush_read(fkl[ks], sJkl + ks * (WL * WK), tid);
#endif
#if DO_K
                    flush_read(fik[ks], sKik + ks * (WI * WK), tid);
                    flush_read(fil[ks], sKil + ks * (WI * WL), tid);
                    flush_read(fjk[ks], sKjk + ks * (WJ * WK), tid);
                    flush_read(fjl[ks], sKjl + ks * (WJ * WL), tid);
#endif
                }
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    if (!kval[ks]) continue;
                    const int k0 = k0s[ks], l0 = l0s[ks];
#if DO_J
                    flush_atomics(fkl[ks], vj + idm * nao2, nao, l0, k0, tid);
#endif
#if DO_K
                    double* __restrict__ K = vk + idm * nao2;
                    flush_atomics(fik[ks], K, nao, i0, k0, tid);
                    flush_atomics(fil[ks], K, nao, i0, l0, tid);
                    flush_atomics(fjk[ks], K, nao, j0, k0, tid);
                    flush_atomics(fjl[ks], K, nao, j0, l0, tid);
#endif
                }
            }
#else
            const int ndm_here = ndm_grp;
#pragma unroll
            for (int dmi = 0; dmi < NDM; dmi++) {
            if (dmi >= ndm_here) break;
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                if (!kval[ks]) continue;
                const int k0 = k0s[ks], l0 = l0s[ks];
#if DO_J
                flush_tile(sJkl + (dmi * NKS + ks) * (WL * WK), vj + (idm + dmi) * nao2, nao, l0, k0, WL, WK, tid);
#endif
#if DO_K
                double* __restrict__ K = vk + (idm + dmi) * nao2;
                flush_tile(sKik + (dmi * NKS + ks) * (WI * WK), K, nao, i0, k0, WI, WK, tid);
                flush_tile(sKil + (dmi * NKS + ks) * (WI * WL), K, nao, i0, l0, WI, WL, tid);
                flush_tile(sKjk + (dmi * NKS + ks) * (WJ * WK), K, nao, j0, k0, WJ, WK, tid);
                flush_tile(sKjl + (dmi * NKS + ks) * (WJ * WL), K, nao, j0, l0, WJ, WL, tid);
#endif
            }
            }
#endif
            }
#undef tid
            STAMP(8);
        }
#if DO_J
        __syncthreads();
        {
#if KARG_RELOAD
            const KArgs AS4* kf = kargs();
            const int nao = kf->nao;
            const size_t nao2 = (size_t)nao * nao;
            double* __restrict__ vj = kf->vj;
#endif
            // J_ij: summed over the whole ket chunk (lane-per-quartet mode: JREP replicas)
            const int ndm_here = ndm_grp;
            for (int dmi = 0; dmi < NDM && dmi < ndm_here; dmi++)
                for (int rep = 0; rep < (TILE_1Q ? JREP : 1); rep++)
                    flush_tile(sJij + (dmi * (TILE_1Q ? JREP : 1) + rep) * (WJ * WI), vj + (idm + dmi) * nao2, nao, j0, i0, WJ, WI, tid);
        }
#endif
        STAMP(9);
    }
#if STAMPS
    if (tid == 0 && counter) {
        for (int k = 0; k < 15; k++) atomicAdd(counter - 1 - k, st_acc[k]);
        atomicAdd(counter - 16, 1ull);
    }
#endif
#if KARG_RELOAD
    {
        const KArgs AS4* ke = kargs();
        unsigned long long* cnt = ke->counter;
        if (tid == 0 && cnt && nq_done) atomicAdd(cnt + ke->tasks[row * 8 + 6], (unsigned long long)nq_done);
#if MIXED
        unsigned long long* cnt32 = ke->counter32;
        if (tid == 0 && cnt32 && nq32_done) atomicAdd(cnt32 + ke->tasks[row * 8 + 6], (unsigned long long)nq32_done);
#endif
    }
#else
    if (tid == 0 && counter && nq_done) atomicAdd(counter + tk[6], (unsigned long long)nq_done);   // per task row (slot 6)
#if MIXED
    if (tid == 0 && counter32 && nq32_done) atomicAdd(counter32 + tk[6], (unsigned long long)nq32_done);
#endif
#endif
}
