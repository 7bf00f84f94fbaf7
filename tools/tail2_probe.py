"""Register-resident coarse tail (tail=1) against the LDS tail (tail=2) and per-level launches (tail=0): bit-for-bit check of
a few solves per smoother / cycle / precision, then the W(2,2) and V(2,2) cycle times at a large size.

    python3 tools/tail2_probe.py [n_timing] [cycles]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib


def rhs_of(n):
    x = np.linspace(0.0, 1.0, n)
    rng = np.random.default_rng(n)
    return 2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :] + 0.1 * rng.standard_normal((n, n))


bad = 0
for n in (33, 65, 129, 257):
    for cyc in ("V", "W", "F"):
        for sm, om in ((_lib.MG_JACOBI, 0.8), (_lib.MG_RBGS, 1.0), (_lib.MG_RBGS, 1.15)):
            for prec in (_lib.MG_PREC_DOUBLE, _lib.MG_PREC_SINGLE, _lib.MG_PREC_SINGLE_MANAGED, _lib.MG_PREC_MIXED_LEVELS):
                for direct in (False, True):
                    if cyc == "F" and n > 129:
                        continue
                    res = {}
                    for tail in (1, 2):
                        e = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle=cyc, smoother=sm, omega=om,
                                               precision=prec, tail=tail, coarse_direct=direct, coarse_maxit=60)
                        f = rhs_of(n)
                        if prec == _lib.MG_PREC_SINGLE:
                            f = f.astype(np.float32)
                        u, r = e.solve(f, tol=0.0, max_iterations=3)
                        res[tail] = (u, r["residual_history"])
                        e.close()
                    same = np.array_equal(res[1][0], res[2][0]) and res[1][1] == res[2][1]
                    if not same:
                        bad += 1
                        d = np.max(np.abs(res[1][0].astype(np.float64) - res[2][0])) / np.max(np.abs(res[2][0]))
                        print(f"MISMATCH n={n} {cyc} sm={sm} om={om} prec={prec} direct={direct}: rel {d:.3e} hist {res[1][1]} vs {res[2][1]}")
print("bitwise check:", "OK" if bad == 0 else f"{bad} mismatches")


def coef_of(n):
    x = np.linspace(0.0, 1.0, n)
    return 1.0 + 0.5 * np.sin(3 * np.pi * x)[:, None] * np.cos(2 * np.pi * x)[None, :] + 0.25 * x[:, None]


vbad = 0
for n in (33, 65, 129, 257):
    for cyc in ("V", "W", "F"):
        for sm, om in ((_lib.MG_JACOBI, 0.8), (_lib.MG_RBGS, 1.0), (_lib.MG_RBGS, 1.15)):
            for prec in (_lib.MG_PREC_DOUBLE, _lib.MG_PREC_SINGLE, _lib.MG_PREC_SINGLE_MANAGED, _lib.MG_PREC_MIXED_LEVELS):
                for direct in (False, True):
                    for shift in (0.0, 3.0):
                        if cyc == "F" and n > 129:
                            continue
                        res = {}
                        for tail in (1, 2):
                            e = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle=cyc, smoother=sm, omega=om,
                                                   precision=prec, tail=tail, coarse_direct=direct, coarse_maxit=60)
                            e.set_coefficient(coef_of(n))
                            if shift:
                                e.set_shift(shift)
                            f = rhs_of(n)
                            if prec == _lib.MG_PREC_SINGLE:
                                f = f.astype(np.float32)
                            u, r = e.solve(f, tol=0.0, max_iterations=3)
                            res[tail] = (u, r["residual_history"])
                            e.close()
                        same = np.array_equal(res[1][0], res[2][0]) and res[1][1] == res[2][1]
                        if not same:
                            vbad += 1
                            d = np.max(np.abs(res[1][0].astype(np.float64) - res[2][0])) / np.max(np.abs(res[2][0]))
                            print(f"VAR MISMATCH n={n} {cyc} sm={sm} om={om} prec={prec} direct={direct} shift={shift}: rel {d:.3e} "
                                  f"hist {res[1][1]} vs {res[2][1]}")
print("variable-coefficient bitwise check:", "OK" if vbad == 0 else f"{vbad} mismatches")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = rhs_of(n) * 0 + 2 * np.pi**2 * np.sin(np.pi * np.linspace(0, 1, n))[:, None] * np.sin(np.pi * np.linspace(0, 1, n))[None, :]
for cyc, sm, om, name in (("W", _lib.MG_RBGS, 1.0, "W(2,2) rbgs"), ("V", _lib.MG_RBGS, 1.0, "V(2,2) rbgs"), ("V", _lib.MG_JACOBI, 0.8, "V(2,2) jacobi"),
                          ("W", _lib.MG_JACOBI, 0.8, "W(2,2) jacobi")):
    for tail, direct in ((1, None), (1, False), (2, None), (2, False)):
        e = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle=cyc, smoother=sm, omega=om, tail=tail, coarse_direct=direct)
        e.set_rhs(f)
        e.set_solution(None)
        e.iterate(0.0, 2)
        e.set_solution(None)
        r = e.iterate(0.0, cycles)
        print(f"{n}^2 fp64 {name:14s} tail={tail} direct={direct}: {r['solve_seconds'] / cycles * 1e3:8.3f} ms/cycle   ||r|| -> {r['residual_history'][-1]:.3e}", flush=True)
        e.close()
