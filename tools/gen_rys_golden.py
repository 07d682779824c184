#!/usr/bin/env python3
"""Writes tests/golden/rys_mpmath.json: Rys roots/weights at scattered x from an 80-digit mpmath
Golub-Welsch computation (tools/gen_rys_tables.py:rys_mp), independent of the Chebyshev fit."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_rys_tables import rys_mp

rng = np.random.default_rng(2025)
out = []
for n in range(1, 10):
    xs = list(rng.uniform(0, 5 * n + 35, 6)) + [0.0, 1e-8, 2.5, 5 * n + 34.9999, 5 * n + 35.0, 5 * n + 50.0, 300.0]
    for x in xs:
        r, w = rys_mp(n, x)
        out.append({"n": n, "x": float(x), "roots": [float(v) for v in r], "weights": [float(v) for v in w]})
with open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rys_mpmath.json"), "w") as f:
    json.dump(out, f)
print(len(out), "vectors")
