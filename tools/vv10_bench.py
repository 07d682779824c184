"""VV10 kernel throughput (pair evaluations/s; ~30 flop per pair) on synthetic NLC grids."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from joltqc_amd.backend import lib as _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
L = _lib.lib(); dev = _lib.require_gpu()
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.rand(*s, dtype=torch.float64, device=dev, generator=g)
co, ci = (r(3, n) * 30).contiguous(), (r(3, n) * 30).contiguous()
W0, K, W0p, Kp, RpW = r(n) + 0.5, r(n) + 0.5, r(n) + 0.5, r(n) + 0.5, r(n) * 1e-3
F, U, W = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
for fp32 in (1, 3, 0):        # FP32 inner loop with / without the denominator test (include/jqc_hip.h), FP64
    for it in range(2):
        torch.cuda.synchronize(); t = time.time()
        _lib.check(L.jqc_vv10(F.data_ptr(), U.data_ptr(), W.data_ptr(), ci.data_ptr(), co.data_ptr(), W0p.data_ptr(), W0.data_ptr(),
                              K.data_ptr(), Kp.data_ptr(), RpW.data_ptr(), n, n, fp32, _lib.stream_ptr()))
        torch.cuda.synchronize(); dt = time.time() - t
    print(f"vv10 {('fp64', 'fp32', '', 'fp32 no test')[fp32]} inner: N={n}  {dt*1e3:.2f} ms  {n*n/dt:.3e} pairs/s  {30.0*n*n/dt/1e12:.2f} TFLOP/s "
          f"(30 flop/pair model) = {30.0*n*n/dt/157.3e12:.1%} of the FP32 vector peak; sums F {float(F.sum()):.10e} U {float(U.sum()):.10e} W {float(W.sum()):.10e}")
