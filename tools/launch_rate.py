#!/usr/bin/env python3
"""Is the host launch path the bottleneck?  Time until mg_cycle(K) RETURNS (all launches queued) vs until the GPU is done."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
for n in (4097, 1025, 257):
    x = np.linspace(0, 1, n); rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
    eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), smoother=0, omega=0.8)
    eng.set_rhs(rhs); eng.set_solution(None); eng.cycle(3); eng.synchronize()
    K = 50
    t0 = time.perf_counter(); eng.cycle(K); t1 = time.perf_counter(); eng.synchronize(); t2 = time.perf_counter()
    ms = eng.time_op("cycle", 0, np.float64, K)
    print(f"n={n}: queued in {(t1-t0)/K*1e6:6.1f} us/cycle, done in {(t2-t0)/K*1e6:6.1f} us/cycle, hipEvent {ms*1e3:6.1f} us/cycle, levels {eng.num_levels}")
    eng.close()
