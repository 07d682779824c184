"""ctypes front-end of oracle/libjk_oracle.so (CPU restatement of the reference J/K arithmetic).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_BLOB = None


def build(force=False):
    so = os.path.join(_HERE, "libjk_oracle.so")
    src = os.path.join(_HERE, "jk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libjk_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB, _BLOB
    if _LIB is None:
        L = ctypes.CDLL(build())
        dp = ctypes.POINTER(ctypes.c_double)
        L.jqc_oracle_set_rys.argtypes = [dp]
        L.jqc_oracle_rys_roots.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, dp]
        L.jqc_oracle_eri_block.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp]
        L.jqc_oracle_jk.argtypes = [ctypes.c_int, dp, ctypes.c_int, dp, dp, dp, ctypes.c_double,
                                    ctypes.POINTER(ctypes.c_uint16), ctypes.c_long, ctypes.c_int, ctypes.c_int]
        L.jqc_oracle_schwarz.argtypes = [dp, ctypes.c_int, ctypes.c_double, dp]
        L.jqc_oracle_jk_mt.argtypes = L.jqc_oracle_jk.argtypes + [ctypes.c_int]
        L.jqc_oracle_jk_dense.argtypes = [ctypes.c_int, dp, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_int, dp, dp, dp,
                                          ctypes.c_double, dp, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int]
        L.jqc_oracle_jk_dense.restype = ctypes.c_long
        L.jqc_oracle_jk_bench.argtypes = [ctypes.c_int, dp, dp, ctypes.c_double, ctypes.POINTER(ctypes.c_uint16), ctypes.c_long,
                                          ctypes.c_int, ctypes.c_int]
        L.jqc_oracle_jk_bench.restype = ctypes.c_double
        L.jqc_oracle_vv10.argtypes = [ctypes.c_int, dp, dp, dp, ctypes.c_int, dp, dp, dp, dp, dp, dp, dp]
        from joltqc_amd.backend.rys import pack_tables  # data file only (numbers), no product code path
        _BLOB = np.array(pack_tables())
        L.jqc_oracle_set_rys(_BLOB.ctypes.data_as(dp))
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def rys_roots(n, x, theta=1.0, omega=0.0):
    rw = np.zeros(2 * n)
    lib().jqc_oracle_rys_roots(n, float(x), float(theta), float(omega), _dp(rw))
    return rw[0::2].copy(), rw[1::2].copy()


def nf(l):
    return (l + 1) * (l + 2) // 2


def eri_block(basis, ish, jsh, ksh, lsh, omega=0.0):
    basis = np.ascontiguousarray(basis, dtype=np.float64)
    ls = [int(basis[s, 11]) for s in (ish, jsh, ksh, lsh)]
    out = np.zeros([nf(l) for l in ls])
    lib().jqc_oracle_eri_block(_dp(basis), ish, jsh, ksh, lsh, float(omega), _dp(out))
    return out


def jk_raw(basis, dm, quartets, omega=0.0, do_j=True, do_k=True, nthreads=1):
    """Raw (pre-epilogue) vj, vk for internal-order Cartesian density matrices dm[n_dm,nao,nao].
    ``nthreads`` > 1: the quartet list is dealt to OpenMP threads (private accumulators, summed at the end)."""
    basis = np.ascontiguousarray(basis, dtype=np.float64)
    dm = np.ascontiguousarray(dm, dtype=np.float64)
    if dm.ndim == 2:
        dm = dm[None]
    n_dm, nao, _ = dm.shape
    q = np.ascontiguousarray(quartets, dtype=np.uint16).reshape(-1, 4)
    vj = np.zeros_like(dm)
    vk = np.zeros_like(dm)
    lib().jqc_oracle_jk_mt(nao, _dp(basis), n_dm, _dp(dm), _dp(vj), _dp(vk), float(omega or 0.0),
                           q.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), q.shape[0], int(do_j), int(do_k), int(nthreads))
    return vj, vk


def jk_bench(basis, dm, quartets, reps=1, omega=0.0, nthreads=1):
    """bench.py's CPU-baseline leg: ERI blocks + the six contractions of every listed quartet, ``reps`` passes, digested
    into thread-local blocks (no shared Fock matrix: see jk_oracle.c).  Returns the checksum."""
    basis = np.ascontiguousarray(basis, dtype=np.float64)
    dm = np.ascontiguousarray(dm, dtype=np.float64)
    q = np.ascontiguousarray(quartets, dtype=np.uint16).reshape(-1, 4)
    return lib().jqc_oracle_jk_bench(dm.shape[-1], _dp(basis), _dp(dm), float(omega or 0.0),
                                     q.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), q.shape[0], int(reps), int(nthreads))


def jk_raw_dense(basis, dm, skip, omega=0.0, do_j=True, do_k=True, cutoff=None, nthreads=None):
    """Raw vj, vk over EVERY canonical quartet of the shells with ``skip[s] == False``, generated inside the C loop
    (OpenMP).  ``cutoff``: drop quartets whose Schwarz estimate Q_ij Q_kl max|D| is below it (the checker's own,
    much looser screening for big cases; None = no screening at all).  Returns (vj, vk, number of quartets)."""
    basis = np.ascontiguousarray(basis, dtype=np.float64)
    dm = np.ascontiguousarray(dm, dtype=np.float64)
    if dm.ndim == 2:
        dm = dm[None]
    n_dm, nao, _ = dm.shape
    nbas = basis.shape[0]
    sk = np.ascontiguousarray(np.asarray(skip, dtype=np.uint8))
    vj, vk = np.zeros_like(dm), np.zeros_like(dm)
    lq = None
    log_cut, log_dmax = -1e300, 0.0
    if cutoff is not None:
        lq = np.log(schwarz(basis, omega) + 1e-300)
        log_cut = float(np.log(cutoff))
        log_dmax = float(np.log(np.abs(dm).max() + 1e-300))
    nt = int(nthreads or os.cpu_count() or 1)
    n = lib().jqc_oracle_jk_dense(nao, _dp(basis), nbas, sk.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), n_dm, _dp(dm),
                                  _dp(vj), _dp(vk), float(omega or 0.0), _dp(lq) if lq is not None else None, log_dmax,
                                  log_cut, int(do_j), int(do_k), nt)
    return vj, vk, int(n)


def schwarz(basis, omega=0.0):
    basis = np.ascontiguousarray(basis, dtype=np.float64)
    n = basis.shape[0]
    out = np.zeros((n, n))
    lib().jqc_oracle_schwarz(_dp(basis), n, float(omega or 0.0), _dp(out))
    return out


def all_quartets(nbas):
    """Every canonical quartet i>=j, k>=l, (ij)>=(kl) (no screening), uint16[n,4]."""
    out = []
    for i in range(nbas):
        for j in range(i + 1):
            for k in range(i + 1):
                for l in range(k + 1):
                    if i * nbas + j >= k * nbas + l:
                        out.append((i, j, k, l))
    return np.array(out, dtype=np.uint16).reshape(-1, 4)


def vv10_sums(coords, vvcoords, W0, K, W0p, Kp, RpW):
    """F, U, W of the VV10 double sum, signature of oracle/dft.py:vv10_kernel (jqc_oracle_vv10: reference vv10.cu:86-117 in FP64, OpenMP over the outer points)."""
    c = np.ascontiguousarray(coords, dtype=np.float64)
    v = np.ascontiguousarray(vvcoords, dtype=np.float64)
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (W0, K, W0p, Kp, RpW)]
    n, m = c.shape[0], v.shape[0]
    F, U, W = np.empty(n), np.empty(n), np.empty(n)
    lib().jqc_oracle_vv10(n, _dp(c), _dp(a[0]), _dp(a[1]), m, _dp(v), _dp(a[2]), _dp(a[3]), _dp(a[4]), _dp(F), _dp(U), _dp(W))
    return F, U, W
