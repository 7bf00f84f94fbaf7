#!/usr/bin/env python3
"""A few V(2,2) weighted-Jacobi fp64 cycles at n^2 for `rocprofv3 --kernel-trace` runs, and -- given the trace csv -- the
kernels of the LAST cycle in launch order with their durations (which level costs what inside one cycle).

    python3 tools/vcycle_probe.py [n] [cycles]
    python3 tools/vcycle_probe.py --trace kernel_trace.csv
"""
import csv
import os
import sys

import numpy as np

if len(sys.argv) > 2 and sys.argv[1] == "--trace":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tails = [k for k, r in enumerate(rows) if "coarse_tail_kernel" in r["Kernel_Name"]]
    lo = tails[-2] + 1 if len(tails) > 1 else 0
    # from the launch after the previous cycle's tail ... to the end
    seq = rows[lo:]
    t0 = int(seq[0]["Start_Timestamp"])
    prev_end = None
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        name = name[:name.find("(")] if "(" in name else name
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"+{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.2f} us  gap {gap:5.2f}  grid {r.get('Grid_Size', '?'):>8}  {name[:110]}")
        prev_end = e
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 6
x = np.linspace(0, 1, n)
rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), precision=_lib.MG_PREC_DOUBLE)
eng.set_rhs(rhs)
eng.set_solution(None)
r = eng.iterate(0.0, cycles)
print(f"{n}^2 V(2,2) Jacobi fp64: {r['solve_seconds'] / cycles * 1e3:.3f} ms/cycle, ||r|| -> {r['residual_history'][-1]:.2e}")
eng.close()
