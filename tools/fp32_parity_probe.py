"""How far the fp32 / adaptive_ref golden solves are from the reference's outputs (ulp of max|u|, history ratios):
the numbers behind the assertions of tests/test_gpu_solver.py::test_golden_histories_and_solutions."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_solver as T

g = np.load(os.path.join(ROOT, "tests", "golden", "solves.npz"))
for key in sorted(k for k in g.files if k.endswith("__hist")):
    if not (key.endswith("_float32__hist") or key.endswith("_adaptive_ref__hist") or key.endswith("_mixed__hist")):
        continue
    u, info, pm = T._run(g, key)
    ref_u = g[key.replace("__hist", "__u")]
    hist, ref = np.array(info["residual_history"]), g[key]
    d = np.abs(u.astype(np.float64) - ref_u.astype(np.float64))
    eps = np.finfo(np.float32).eps if not key.endswith("_mixed__hist") else np.finfo(np.float64).eps
    print(key, "dtype", u.dtype, ref_u.dtype, "max|du|/max|u| =", d.max() / np.abs(ref_u).max(), "in eps:", d.max() / np.abs(ref_u).max() / eps,
          "n_diff", int((d > 0).sum()), "of", d.size)
    n = min(len(hist), len(ref))
    print("   hist rel diff:", np.array2string(np.abs(hist[:n] - ref[:n]) / ref[:n], precision=2), "len", len(hist), len(ref))
