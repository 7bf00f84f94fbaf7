#!/usr/bin/env python3
"""Variable-coefficient legs at level 0: LDS-tiled (fused = 1) vs register-blocked (fused = 2).  python3 tools/var_leg_probe.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
x = np.linspace(0, 1, n)
rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
a = 1.0 + 0.5 * np.sin(2 * np.pi * x)[:, None] * np.cos(2 * np.pi * x)[None, :]
for sm, name in ((_lib.MG_JACOBI, "jacobi"), (_lib.MG_RBGS, "rbgs")):
    for fused in (1, 2):
        eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), smoother=sm, omega=0.8 if sm == 0 else 1.0,
                                 precision=_lib.MG_PREC_ADAPTIVE, fused=fused)
        eng.set_coefficient(a); eng.set_rhs(rhs); eng.set_solution(None); eng.cycle(1)
        row = [f"{name} fused={fused}:"]
        for dt in (np.float32, np.float64):
            for op in ("down_leg", "up_leg"):
                row.append(f"{np.dtype(dt).name} {op} {eng.time_op(op, 0, dt, 10) * 1e3:8.1f} us")
        print("  ".join(row), flush=True)
        eng.close()
