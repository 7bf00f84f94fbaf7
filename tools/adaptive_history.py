import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
n=4097
x=np.linspace(0,1,n)
rhs=2*np.pi**2*np.sin(np.pi*x)[:,None]*np.sin(np.pi*x)[None,:]
eng=mg.MultigridEngine(n,n,max_levels=mg.default_max_levels(n,n),precision=_lib.MG_PREC_ADAPTIVE, switch_threshold=1e-6)
eng.set_rhs(rhs); eng.set_solution(None)
r=eng.iterate(0.0, 10)
print([f"{v:.3e}" for v in r["residual_history"]]); print(r.get("precision_history"))
