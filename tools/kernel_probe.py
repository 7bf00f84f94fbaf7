#!/usr/bin/env python3
"""Launch the hot kernels of the 4097^2 hierarchy a few times (for rocprofv3 --pmc / --kernel-trace runs).

    python3 tools/kernel_probe.py [n] [reps] [ops...]      ops from: jacobi rbgs residual residual_norm restrict prolong cycle
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib      # noqa: E402

FUSED = int(os.environ.get("MG_FUSED", "1"))      # 1 LDS-tiled legs, 2 register-blocked legs (include/mghip.h mg_config.fused)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ops = sys.argv[3:] or ["jacobi", "sweeps2", "down_leg", "up_leg", "residual", "residual_norm", "restrict", "prolong"]
x = np.linspace(0, 1, n)
rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
RB = [o for o in ops if o.startswith("rb:")]
for smoother, names in ((_lib.MG_JACOBI, [o for o in ops if o != "rbgs" and not o.startswith("rb:")]),
                        (_lib.MG_RBGS, [o for o in ops if o == "rbgs"] + [o[3:] for o in RB])):
    if not names:
        continue
    print("--- smoother:", "jacobi" if smoother == _lib.MG_JACOBI else "red-black GS")
    eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), smoother=smoother,
                             omega=0.8 if smoother == _lib.MG_JACOBI else 1.0, precision=_lib.MG_PREC_ADAPTIVE, fused=FUSED)
    eng.set_rhs(rhs)
    eng.set_solution(None)
    eng.cycle(1)
    for dt, w in ((np.float32, 4), (np.float64, 8)):
        for op in names:
            try:
                ms = eng.time_op(op, 0, dt, reps)
            except Exception as exc:                      # e.g. no spanning leg for this hierarchy / precision
                print(f"{op:14s} {np.dtype(dt).name}: not available ({exc})")
                continue
            print(f"{op:14s} {np.dtype(dt).name}: {ms * 1e3:9.2f} us/launch   ({n * n * w / ms / 1e6:8.1f} GB/s per word/DoF)")
    eng.close()
