"""Development helper: SCREEN_LDS option of jk_tile.hip -- the eight small tables the screening predicate reads per candidate
(two Schwarz blocks, six density-bound blocks) are loaded once per tile pair with the staging loads and the candidates are screened
from LDS, instead of eight global gathers per candidate.  usage: python tools/dev/apply_screen_lds_patch.py <jk_tile.hip>"""
import sys
p = sys.argv[1]
s = open(p).read()
assert "SCREEN_LDS" not in s, "already patched"

s = s.replace('''#ifndef ORED
#define ORED 0''', '''#ifndef SCREEN_LDS
#define SCREEN_LDS 0 // 1: the screening predicate reads its Schwarz bounds and density bounds from LDS: the blocks of q_cond / log_dm that
                    // belong to the tile pair (<= 8 x 8 entries each: q_ij, ld_ij once per workgroup; q_kl, ld_ik, ld_jk, ld_il, ld_jl, ld_kl
                    // per ket slot) are fetched with the staging loads (two coalesced loads per thread and slot) and every candidate
                    // is then screened from LDS -- instead of eight dependent global gathers per candidate and round (1 024 candidates
                    // of a (ps|ps) tile pair: 32 gathers per thread and slot, two exposed L2 round trips).  Same values, same
                    // survivors; costs one more workgroup barrier per iteration.
#endif
#ifndef ORED
#define ORED 0''', 1)

# LDS arrays
s = s.replace('''    __shared__ real sBas[(TSI + TSJ + NKS * (TSK + TSL)) * BASIS_STRIDE];''', '''#if SCREEN_LDS
    constexpr int SC_KL = 0, SC_IK = SC_KL + TSK * TSL, SC_JK = SC_IK + TSI * TSK, SC_IL = SC_JK + TSJ * TSK, SC_JL = SC_IL + TSI * TSL,
                  SC_DKL = SC_JL + TSJ * TSL, SC_N = SC_DKL + TSK * TSL;
    __shared__ float sScr[NKS * SC_N], sScrB[2 * TSI * TSJ];
#endif
    __shared__ real sBas[(TSI + TSJ + NKS * (TSK + TSL)) * BASIS_STRIDE];''', 1)

# bra-side blocks: at workgroup start, next to the bra shell rows
s = s.replace('''        if (tid < 2) s_nact[tid] = 0;
#if MIXED
        if (tid < 2) s_nact32[tid] = 0;
#endif
        if (tid < (TSI + TSJ) * BASIS_STRIDE) sBas[tid] = rb;''', '''        if (tid < 2) s_nact[tid] = 0;
#if MIXED
        if (tid < 2) s_nact32[tid] = 0;
#endif
#if SCREEN_LDS
        for (int n = tid; n < 2 * TSI * TSJ; n += TBLOCK) {
            const int e = n % (TSI * TSJ), a = e / TSJ, b = e - a * TSJ;
            sScrB[n] = (n < TSI * TSJ ? q_cond : log_dm)[(ish0 + a) * nbas + jsh0 + b];
        }
#endif
        if (tid < (TSI + TSJ) * BASIS_STRIDE) sBas[tid] = rb;''', 1)

# per-slot: loads of the screening blocks + replace the gather screening + store
old_scr_start = '''                STAMP(3);
                // ---- per-quartet screening of the NQ candidates of the tile pair (wave64 ballots); every wave appends
                //      its survivors to the queue through one LDS counter
#pragma unroll 2
                for (int cand0 = cand_lo; cand0 < cand_hi; cand0 += TBLOCK) {'''
assert s.count(old_scr_start) == 1
new_scr_start = '''#if SCREEN_LDS
                constexpr int NSC = (SC_N + TBLOCK - 1) / TBLOCK;
                float rsc[NSC];
#pragma unroll
                for (int u = 0; u < NSC; u++) {
                    const int n = tid + u * TBLOCK;
                    rsc[u] = 0.f;
                    if (n < SC_IK) { const int c = n / TSL, d = n - c * TSL; rsc[u] = q_cond[(ksh0 + c) * nbas + lsh0 + d]; }
                    else if (n < SC_JK) { const int e = n - SC_IK, a = e / TSK, c = e - a * TSK; rsc[u] = log_dm[(ish0 + a) * nbas + ksh0 + c]; }
                    else if (n < SC_IL) { const int e = n - SC_JK, b = e / TSK, c = e - b * TSK; rsc[u] = log_dm[(jsh0 + b) * nbas + ksh0 + c]; }
                    else if (n < SC_JL) { const int e = n - SC_IL, a = e / TSL, d = e - a * TSL; rsc[u] = log_dm[(ish0 + a) * nbas + lsh0 + d]; }
                    else if (n < SC_DKL) { const int e = n - SC_JL, b = e / TSL, d = e - b * TSL; rsc[u] = log_dm[(jsh0 + b) * nbas + lsh0 + d]; }
                    else if (n < SC_N) { const int e = n - SC_DKL, c = e / TSL, d = e - c * TSL; rsc[u] = log_dm[(ksh0 + c) * nbas + lsh0 + d]; }
                }
#endif
                STAMP(3);
                // ---- per-quartet screening of the NQ candidates of the tile pair (wave64 ballots); every wave appends
                //      its survivors to the queue through one LDS counter
#pragma unroll 2
                for (int cand0 = cand_lo; cand0 < (SCREEN_LDS ? cand_lo : cand_hi); cand0 += TBLOCK) {'''
s = s.replace(old_scr_start, new_scr_start, 1)

old_store = '''                // ---- write LDS
                if (tid < (TSK + TSL) * BASIS_STRIDE) sBas[OFF_K + ks * KSTR + tid] = rb;
#pragma unroll
                for (int u = 0; u < NPK; u++)
                    if (tid + u * TBLOCK < TSK * TSL * 27) sPK[ks * (TSK * TSL * 27) + tid + u * TBLOCK] = rpk[u];
#pragma unroll
                for (int dmi = 0; dmi < NDM; dmi++) {
#if DO_J
                    tile_store(sDkl + (dmi * NKS + ks) * (WL * WK), rkl[dmi], tid);
#endif'''
assert s.count(old_store) == 1
s = s.replace(old_store, '''                // ---- write LDS
#if SCREEN_LDS
#pragma unroll
                for (int u = 0; u < NSC; u++)
                    if (tid + u * TBLOCK < SC_N) sScr[ks * SC_N + tid + u * TBLOCK] = rsc[u];
#endif
                if (tid < (TSK + TSL) * BASIS_STRIDE) sBas[OFF_K + ks * KSTR + tid] = rb;
#pragma unroll
                for (int u = 0; u < NPK; u++)
                    if (tid + u * TBLOCK < TSK * TSL * 27) sPK[ks * (TSK * TSL * 27) + tid + u * TBLOCK] = rpk[u];
#pragma unroll
                for (int dmi = 0; dmi < NDM; dmi++) {
#if DO_J
                    tile_store(sDkl + (dmi * NKS + ks) * (WL * WK), rkl[dmi], tid);
#endif''', 1)

# after the slot loop (non-STAGE_ALL branch): the LDS screening pass
old_end = '''#endif  // STAGE_ALL
#undef tid'''
assert s.count(old_end) == 1
s = s.replace(old_end, '''#if SCREEN_LDS
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                if (!kval[ks]) continue;
                const int ksh0 = ksh0s[ks], lsh0 = lsh0s[ks];
                const float* sc = sScr + ks * SC_N;
#pragma unroll 2
                for (int cand0 = cand_lo; cand0 < cand_hi; cand0 += TBLOCK) {
                    const int cd = cand0 + tid;
                    bool keep = false;
                    if (cd < cand_hi) {
                        const int a = cd % TSI, b = (cd / TSI) % TSJ, d = (cd / (TSI * TSJ)) % TSL;
                        const int c = QC(cd / (TSI * TSJ * TSL), a, b, d);
                        const int ish = ish0 + a, jsh = jsh0 + b, ksh = ksh0 + c, lsh = lsh0 + d;
                        if (ish >= jsh && ksh >= lsh && ish * nbas + jsh >= ksh * nbas + lsh) {
                            const float sq = sScrB[a * TSJ + b] + sc[SC_KL + c * TSL + d];
                            float sd = -36.8f;
#if DO_K
                            sd = fmaxf(sd, sc[SC_IK + a * TSK + c]);
                            sd = fmaxf(sd, sc[SC_JK + b * TSK + c]);
                            sd = fmaxf(sd, sc[SC_IL + a * TSL + d]);
                            sd = fmaxf(sd, sc[SC_JL + b * TSL + d]);
#endif
#if DO_J
                            sd = fmaxf(sd, sScrB[TSI * TSJ + a * TSJ + b]);
                            sd = fmaxf(sd, sc[SC_DKL + c * TSL + d]);
#endif
                            const float dq = sq + sd;
                            keep = dq > cut_lo && dq <= cut_hi;
                        }
                    }
                    const unsigned long long m = __ballot(keep);
                    if (m) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(&s_nact[parity], (unsigned)__popcll(m));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (keep) s_act[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(cd | (ks << KS_SHIFT));
                    }
                }
            }
#endif
#endif  // STAGE_ALL
#undef tid''', 1)
s = s.replace('''static_assert(!MIXED || (TILE_1Q && !FP32 && NDM == 1 && !STAGE_ALL_), "MIXED: FP64 lane-per-quartet builds, one density matrix per evaluation");''','''static_assert(!MIXED || (TILE_1Q && !FP32 && NDM == 1 && !STAGE_ALL_), "MIXED: FP64 lane-per-quartet builds, one density matrix per evaluation");
static_assert(!SCREEN_LDS || (!MIXED && !STAGE_ALL_), "SCREEN_LDS: plain staging path");''')
open(p, 'w').write(s)
print("patched", p)
