import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib, distributed as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
px, py = 2, 1
NX, NY = px * (n - 1) + 1, py * (n - 1) + 1
domain = (0.0, float(px), 0.0, float(py))
for managed, dt in ((True, np.float32), (False, np.float64)):
    ops = D.HipOps(dt, torch.device("cuda", 0), managed_single=managed)
    s = D.DistributedMultigrid(NX, NY, px, py, range(px * py), ops, None, domain=domain, smoother="jacobi", omega=0.8, cycle="V", pre=2, post=2)
    s.set_problem(lambda b: D.sine_rhs_block(b, domain))
    h = [s.residual_norm()]
    for _ in range(6):
        s.cycle(0); h.append(s.residual_norm())
    print("virtual", dt.__name__, s.mode, s.Ld, ["%.3e" % v for v in h])
    s.close()
x = np.linspace(domain[0], domain[1], NX); y = np.linspace(domain[2], domain[3], NY)
rhs = 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]
eng = mg.MultigridEngine(NX, NY, domain=domain, max_levels=mg.default_max_levels(NX, NY), precision=_lib.MG_PREC_SINGLE_MANAGED)
u, r = eng.solve(rhs, tol=0.0, max_iterations=6)
print("engine", ["%.3e" % v for v in r["residual_history"]])
