#!/usr/bin/env python3
"""Spanning leg (speculate = 2) against up leg + down leg (speculate = 1): kernel times and cycle times.  python3 tools/span_probe.py [n] [cycles]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = np.linspace(0, 1, n)
rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
for prec, name in ((_lib.MG_PREC_DOUBLE, "double"), (_lib.MG_PREC_SINGLE_MANAGED, "single (fp64 coarsest)")):
    dt = np.float64 if prec == _lib.MG_PREC_DOUBLE else np.float32
    eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), smoother=_lib.MG_JACOBI, omega=0.8, precision=prec, speculate=2)
    eng.set_rhs(rhs); eng.set_solution(None); eng.cycle(1)
    row = [f"{name:24s}"]
    for op in ("down_leg", "up_leg", "span_leg", "span_leg_nomid"):
        row.append(f"{op} {eng.time_op(op, 0, dt, 20) * 1e3:7.1f} us")
    print("  ".join(row), flush=True)
    eng.close()
    for spec in (1, 2):
        for tol in (0.0, 1e-300):
            eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), smoother=_lib.MG_JACOBI, omega=0.8, precision=prec, speculate=spec)
            eng.set_rhs(rhs); eng.set_solution(None); eng.iterate(tol, 3)
            eng.set_solution(None)
            r = eng.iterate(tol, cycles)
            print(f"    speculate={spec} tol={tol:g}: {r['solve_seconds'] / r['iterations'] * 1e6:8.1f} us/cycle  {n * n / (r['solve_seconds'] / r['iterations']) / 1e9:6.2f} GDoF/s"
                  f"  ||r|| -> {r['residual_history'][-1]:.3e}", flush=True)
            eng.close()
