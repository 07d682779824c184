"""Algorithmic FLOP model of the J/K path (loop structure of the reference's
``jqc/backend/jk/1q1t.cu:174-405, :423-638``).  Counts mul/add as 1 and FMA as 2 for ONE dispatched
shell quartet of class ``(li,lj,lk,ll; npi,npj,npk,npl)``:

    F = N_p (F_prim + F_rys + n_r F_root) + F_con
    F_prim = 45,  F_rys = 76 n_r
    F_root = 3 [6 (lij+1)(lkl+1) + 2 (lkl+1) sum_{j<lj}(lij-j) + 2 (li+1)(lj+1) sum_{l<ll}(lkl-l)] + 3 N_int
    F_con  = n_dm 2 N_int (2 [do_j] + 4 [do_k])
"""

FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X datasheet: 256 CU x 4 SIMD x 16 FP64 FMA lanes/clk x 2 flop x 2.4 GHz
FP64_MFMA_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64: the datasheet's FP64 matrix rate equals the vector rate
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0
# Measured on an MI355X of this pool (tools/micro/fp64_fma_peak.hip, profiles/r05_fp64_fma_peak_microbench.json: chains of
# independent v_fma_f64 / v_pk_fma_f32 / v_mfma_f64_16x16x4_f64, every CU, best of 1-8 waves per SIMD) -- SURVEY.md 8(d) asks
# for the measured FMA rate as the denominator; the bench line carries both.  The FP64 vector rate depends on occupancy:
# 59.2 TFLOP/s at one wave per SIMD, 68.6 at two, 75.3 at four, 76.2 at eight (an FMA's latency is not covered by one wave).
FP64_VALU_PEAK_MEASURED_TFLOPS = 76.2
FP64_VALU_PEAK_MEASURED_BY_WAVES = {1: 59.2, 2: 68.6, 4: 75.3, 8: 76.2}
FP64_MFMA_PEAK_MEASURED_TFLOPS = 72.1
FP32_PACKED_VALU_PEAK_MEASURED_TFLOPS = 141.4


def nf(l):
    return (l + 1) * (l + 2) // 2


def quartet_flops(ang, nprim=(1, 1, 1, 1), n_dm=1, do_j=True, do_k=True):
    li, lj, lk, ll = ang
    lij, lkl = li + lj, lk + ll
    n_r = (lij + lkl) // 2 + 1
    n_int = nf(li) * nf(lj) * nf(lk) * nf(ll)
    n_p = nprim[0] * nprim[1] * nprim[2] * nprim[3]
    f_root = 3 * (6 * (lij + 1) * (lkl + 1) + 2 * (lkl + 1) * sum(lij - j for j in range(lj))
                  + 2 * (li + 1) * (lj + 1) * sum(lkl - l for l in range(ll))) + 3 * n_int
    f_con = n_dm * 2 * n_int * (2 * int(do_j) + 4 * int(do_k))
    return n_p * (45 + 76 * n_r + n_r * f_root) + f_con


def quartet_bytes(ang, n_dm=1):
    """Algorithmic bytes per quartet: 8 B of indices + the six density blocks read + as many f64 adds."""
    a = [nf(l) for l in ang]
    blocks = a[0] * a[1] + a[2] * a[3] + a[0] * a[2] + a[0] * a[3] + a[1] * a[2] + a[1] * a[3]
    return 8 + 2 * 8 * blocks * n_dm
