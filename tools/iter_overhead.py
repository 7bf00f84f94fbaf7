#!/usr/bin/env python3
"""How much of a solve-loop step is not kernel time: cycles back-to-back vs the policy+cycle+norm loop."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
x = np.linspace(0, 1, n); rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
for prec, name in ((_lib.MG_PREC_DOUBLE, "double"), (_lib.MG_PREC_SINGLE, "single"), (_lib.MG_PREC_ADAPTIVE, "adaptive")):
    eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), smoother=0, omega=0.8, precision=prec)
    eng.set_rhs(rhs); eng.set_solution(None); eng.cycle(3); eng.synchronize()
    K = 20
    t0 = time.perf_counter(); eng.cycle(K); eng.synchronize(); t1 = time.perf_counter()
    eng.set_solution(None); eng.synchronize()
    t2 = time.perf_counter(); r = eng.iterate(0.0, K); t3 = time.perf_counter()
    t4 = time.perf_counter()
    for _ in range(K): eng.residual_norm()
    t5 = time.perf_counter()
    print(f"{name:9s} n={n}: cycle x{K}: {(t1-t0)/K*1e6:7.1f} us/cycle | iterate x{K}: {(t3-t2)/K*1e6:7.1f} us/step | residual_norm: {(t5-t4)/K*1e6:6.1f} us | codes {r['precision_codes']}")
    eng.close()
