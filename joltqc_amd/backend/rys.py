"""Rys-quadrature tables: load the mpmath-generated coefficients and pack them into one flat
float64 blob shared by the HIP kernels (device copy) and the CPU oracle.

Blob layout (doubles):
    [0]            nmax (= 9)
    [2n-1], [2n]   offsets (in doubles, from blob start) of cheb_n and large_n, n = 1..nmax
    cheb_n         [2n+14 intervals][n roots][14 coefficients][2 = {root, weight}]
    large_n        [n][2]
Root/weight at x (after the theta/omega scaling of ``rys_roots``, reference
``jqc/backend/rys/rys_roots.cu:30-160``):  it = int(0.4 x), u = 0.8 (x - 2.5 it) - 1,
value = c_0 + sum_{k>=1} c_k T_k(u);  for x >= 5n+35:  root = r_i / x, weight = v_i / sqrt(x).
"""
import os
from functools import lru_cache

import numpy as np

NMAX = 9
NCOEF = 14
_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "rys_tables.npz")


@lru_cache(maxsize=1)
def pack_tables():
    d = np.load(_DATA)
    head = np.zeros(2 * NMAX + 1)
    head[0] = NMAX
    parts = []
    off = head.size
    for n in range(1, NMAX + 1):
        cheb = np.ascontiguousarray(d[f"cheb_{n}"], dtype=np.float64)
        large = np.ascontiguousarray(d[f"large_{n}"], dtype=np.float64)
        assert cheb.shape == (2 * n + 14, n, NCOEF, 2) and large.shape == (n, 2)
        head[2 * n - 1] = off
        off += cheb.size
        head[2 * n] = off
        off += large.size
        parts += [cheb.ravel(), large.ravel()]
    blob = np.concatenate([head] + parts)
    blob.setflags(write=False)
    return blob
