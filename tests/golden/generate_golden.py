#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py

It imports the reference's CPU path from /root/reference/src in its one
self-consistent configuration: Grid,
LaplacianOperator(coefficient=-1.0), RestrictionOperator("full_weighting"),
ProlongationOperator("bilinear"), MultigridSolver with the smoother named per
case.  Outputs are data only (inputs + the reference's outputs), written as
compressed .npz files next to this script.
"""

import logging
import os
import sys
import time

import numpy as np

REF = os.environ.get("MG_REFERENCE_SRC", "/root/reference/src")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
logging.disable(logging.CRITICAL)

from multigrid.core.grid import Grid                                    # noqa: E402
from multigrid.core.precision import PrecisionManager                   # noqa: E402
from multigrid.operators.laplacian import LaplacianOperator             # noqa: E402
from multigrid.operators.transfer import RestrictionOperator, ProlongationOperator  # noqa: E402
from multigrid.solvers.multigrid import MultigridSolver                 # noqa: E402
from multigrid.solvers.smoothers import (JacobiSmoother, WeightedJacobiSmoother,      # noqa: E402
                                          GaussSeidelSmoother)
from multigrid.solvers.iterative import EnhancedJacobiSolver            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OP = LaplacianOperator(coefficient=-1.0)
R_FW = RestrictionOperator("full_weighting")
P_BL = ProlongationOperator("bilinear")


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ---------------------------------------------------------------- per-op ----
def gen_ops():
    """Element-wise operators on seeded random fields WITH non-zero boundary
    values (pins quirks F9/F10), square and non-square, dyadic and non-dyadic h."""
    cases = [
        ("sq17", 17, 17, (0.0, 1.0, 0.0, 1.0)),
        ("sq33", 33, 33, (0.0, 1.0, 0.0, 1.0)),
        ("rect17x33", 17, 33, (0.0, 1.0, 0.0, 1.0)),
        ("rect33x9", 33, 9, (0.0, 1.0, 0.0, 1.0)),
        ("nondyadic21x13", 21, 13, (0.0, 1.5, -0.2, 0.5)),
        ("min5", 5, 5, (0.0, 1.0, 0.0, 1.0)),
    ]
    out = {}
    rng = np.random.default_rng(20251205)
    for tag, nx, ny, dom in cases:
        for dt in (np.float64, np.float32):
            g = Grid(nx, ny, domain=dom, dtype=dt)
            u = rng.standard_normal((nx, ny)).astype(dt)
            f = rng.standard_normal((nx, ny)).astype(dt)
            k = f"{tag}_{np.dtype(dt).name}"
            out[f"{k}__domain"] = np.array(dom, dtype=np.float64)
            out[f"{k}__u"] = u
            out[f"{k}__f"] = f
            out[f"{k}__apply"] = OP.apply(g, u)
            out[f"{k}__residual"] = OP.residual(g, u, f)
            out[f"{k}__norm"] = np.array(g.l2_norm(OP.residual(g, u, f)), dtype=np.float64)
            out[f"{k}__jacobi_w23_nu1"] = JacobiSmoother().smooth(g, OP, u, f, 1)
            out[f"{k}__jacobi_w08_nu2"] = WeightedJacobiSmoother().smooth(g, OP, u, f, 2)
            out[f"{k}__vjacobi_w08_nu2"] = EnhancedJacobiSolver(relaxation_parameter=0.8).smooth(g, OP, u, f, 2)
            out[f"{k}__rbgs_w10_nu1"] = GaussSeidelSmoother(red_black=True).smooth(g, OP, u, f, 1)
            out[f"{k}__rbgs_w15_nu2"] = GaussSeidelSmoother(red_black=True, relaxation_parameter=1.5).smooth(g, OP, u, f, 2)
            out[f"{k}__lexgs_w10_nu1"] = GaussSeidelSmoother().smooth(g, OP, u, f, 1)
            if (nx - 1) % 2 == 0 and (ny - 1) % 2 == 0 and nx >= 5 and ny >= 5:
                gc = g.coarsen()
                out[f"{k}__restrict_fw"] = R_FW.apply(g, u, gc)
                e = rng.standard_normal(gc.shape).astype(dt)
                out[f"{k}__e"] = e
                out[f"{k}__prolong"] = P_BL.apply(gc, e, g)
    save("ops.npz", **out)


# ------------------------------------------------------------- full solves --
def sine_rhs(g):
    return 2 * np.pi**2 * np.sin(np.pi * g.X) * np.sin(np.pi * g.Y)


def make_smoother(name):
    if name == "jacobi08":
        return WeightedJacobiSmoother()
    if name == "vjacobi08":
        return EnhancedJacobiSolver(relaxation_parameter=0.8)
    if name == "jacobi23":
        return JacobiSmoother()
    if name == "rbgs":
        return GaussSeidelSmoother(red_black=True)
    if name == "rbgs15":
        return GaussSeidelSmoother(red_black=True, relaxation_parameter=1.15)
    if name == "lexgs":
        return None           # MultigridSolver default (solvers/multigrid.py:112-117)
    raise ValueError(name)


def run_solve(n, levels, cycle, smoother, dtype=np.float64, tol=1e-10, maxit=30,
              pm=None, pre=2, post=2, ny=None, domain=(0.0, 1.0, 0.0, 1.0), rhs=None, u0=None):
    g = Grid(n, ny or n, domain=domain, dtype=dtype)
    s = MultigridSolver(max_levels=levels, max_iterations=maxit, tolerance=tol, cycle_type=cycle,
                        pre_smooth_iterations=pre, post_smooth_iterations=post)
    s.setup(g, OP, R_FW, P_BL, smoother=make_smoother(smoother))
    f = sine_rhs(g).astype(dtype) if rhs is None else rhs
    t = time.time()
    u, info = s.solve(g, OP, f, initial_guess=u0, precision_manager=pm)
    dt = time.time() - t
    return u, info, dt


def gen_solves():
    out = {}
    meta = []
    plan = [
        # n, levels, cycle, smoother, keep_u
        (33, 4, "V", "jacobi08", True), (33, 4, "W", "jacobi08", True),
        (33, 4, "V", "rbgs", True), (33, 4, "W", "rbgs", True),
        (33, 4, "V", "lexgs", True), (33, 4, "F", "rbgs", True),
        (33, 4, "V", "jacobi23", True), (33, 4, "V", "rbgs15", True),
        (65, 5, "V", "jacobi08", True), (65, 5, "W", "jacobi08", True),
        (65, 5, "V", "rbgs", True), (65, 5, "W", "rbgs", True),
        (65, 4, "V", "vjacobi08", True),          # class-default max_levels: coarsest 9x9
        (129, 6, "V", "vjacobi08", True), (129, 6, "W", "rbgs", True),
        (129, 6, "W", "vjacobi08", False), (129, 6, "V", "rbgs", False),
    ]
    for n, L, cyc, sm, keep in plan:
        u, info, dt = run_solve(n, L, cyc, sm)
        k = f"n{n}_L{L}_{cyc}_{sm}_float64"
        out[f"{k}__hist"] = np.array(info["residual_history"], dtype=np.float64)
        out[f"{k}__umax"] = np.array(np.max(np.abs(u)))
        if keep:
            out[f"{k}__u"] = u
        meta.append(k)
        print(f"  {k}: {info['iterations']} its, final {info['final_residual']:.3e}, {dt:.2f}s")

    # asymmetric V(1,2) and V(3,0) sweeps, non-square grid, non-dyadic domain
    u, info, _ = run_solve(65, 5, "V", "vjacobi08", pre=1, post=2)
    out["n65_L5_V12_vjacobi08_float64__hist"] = np.array(info["residual_history"]); out["n65_L5_V12_vjacobi08_float64__u"] = u
    u, info, _ = run_solve(65, 4, "V", "rbgs", ny=33)
    out["n65x33_L4_V_rbgs_float64__hist"] = np.array(info["residual_history"]); out["n65x33_L4_V_rbgs_float64__u"] = u
    u, info, _ = run_solve(33, 4, "V", "vjacobi08", ny=65, domain=(0.0, 1.5, -0.2, 0.5))
    out["n33x65_nondyadic_L4_V_vjacobi08_float64__hist"] = np.array(info["residual_history"])
    out["n33x65_nondyadic_L4_V_vjacobi08_float64__u"] = u

    # random rhs with non-zero boundary values + non-zero initial guess (F10 plateau)
    rng = np.random.default_rng(7)
    rhs = rng.standard_normal((33, 33)); u0 = rng.standard_normal((33, 33))
    u, info, _ = run_solve(33, 4, "V", "rbgs", rhs=rhs, u0=u0, maxit=8)
    out["n33_random_rhs"] = rhs; out["n33_random_u0"] = u0
    out["n33_random_L4_V_rbgs_float64__hist"] = np.array(info["residual_history"]); out["n33_random_L4_V_rbgs_float64__u"] = u

    # fp32 grid (Grid(dtype=float32)), NumPy-vectorised smoother
    u, info, _ = run_solve(129, 6, "V", "vjacobi08", dtype=np.float32, maxit=12)
    out["n129_L6_V_vjacobi08_float32__hist"] = np.array(info["residual_history"], dtype=np.float64)
    out["n129_L6_V_vjacobi08_float32__u"] = u
    u, info, _ = run_solve(65, 5, "V", "rbgs", dtype=np.float32, maxit=12)
    out["n65_L5_V_rbgs_float32__hist"] = np.array(info["residual_history"], dtype=np.float64)
    out["n65_L5_V_rbgs_float32__u"] = u

    # per-level MIXED (PrecisionManager('mixed')): fine half fp64, coarse half fp32
    pm = PrecisionManager(default_precision="mixed")
    u, info, _ = run_solve(129, 6, "V", "vjacobi08", pm=pm)
    out["n129_L6_V_vjacobi08_mixed__hist"] = np.array(info["residual_history"]); out["n129_L6_V_vjacobi08_mixed__u"] = u
    pm = PrecisionManager(default_precision="mixed")
    u, info, _ = run_solve(65, 5, "W", "rbgs", pm=pm)
    out["n65_L5_W_rbgs_mixed__hist"] = np.array(info["residual_history"]); out["n65_L5_W_rbgs_mixed__u"] = u

    # the reference's own ADAPTIVE rule (default PrecisionManager): downgrades to fp32 at
    # iteration 1 and never recovers -- pinned so our policy restatement matches.
    pm = PrecisionManager()
    u, info, _ = run_solve(129, 6, "V", "vjacobi08", pm=pm, maxit=12)
    out["n129_L6_V_vjacobi08_adaptive_ref__hist"] = np.array(info["residual_history"], dtype=np.float64)
    out["n129_L6_V_vjacobi08_adaptive_ref__u"] = u
    out["n129_L6_V_vjacobi08_adaptive_ref__precisions"] = np.array(
        [p.value for p in pm.precision_history])
    save("solves.npz", **out)


def gen_large():
    """BASELINE config 2: 1025^2 fp64 V(2,2) Jacobi 0.8, 9 levels.  The solution
    (8 MB) is not committed: history, a strided sample and norms are."""
    u, info, dt = run_solve(1025, 9, "V", "vjacobi08", maxit=16)
    g = Grid(1025, 1025)
    exact = np.sin(np.pi * g.X) * np.sin(np.pi * g.Y)
    save("large_1025.npz",
         hist=np.array(info["residual_history"]),
         u_sample=u[::32, ::32].copy(),
         u_linf=np.array(np.max(np.abs(u))),
         u_l2=np.array(np.sqrt(np.sum(u * u))),
         err_linf=np.array(np.max(np.abs(u - exact))),
         seconds_per_cycle=np.array(dt / info["iterations"]))
    print(f"  1025^2: {info['iterations']} cycles, {dt / info['iterations']:.2f} s/cycle (reference CPU path, 1 core)")


def gen_large4097():
    """The bench size itself (BASELINE config 3's grid): two V(2,2) Jacobi-0.8 cycles of the reference's own CPU solver
    at 4097^2 fp64, 11 levels -- its seconds per cycle is the `reference_cpu_captured` figure bench.py quotes
    (gpu/gpu_benchmark.py:246-248 metric), the history and a strided sample pin the GPU engine at full size."""
    u, info, dt = run_solve(4097, 11, "V", "vjacobi08", maxit=2, tol=0.0)
    save("large_4097.npz",
         hist=np.array(info["residual_history"]),
         u_sample=u[::128, ::128].copy(),
         u_linf=np.array(np.max(np.abs(u))),
         u_l2=np.array(np.sqrt(np.sum(u * u))),
         seconds_per_cycle=np.array(dt / info["iterations"]),
         cores=np.array(1))
    print(f"  4097^2: {info['iterations']} cycles, {dt / info['iterations']:.2f} s/cycle (reference CPU path, 1 core)")


from heat_inputs import heat_cases, heat_config                          # noqa: E402  (tests/golden/heat_inputs.py)


def gen_heat():
    """applications/heat_equation.py run as is: a few fixed-dt steps per scheme (the implicit ones are the reference's
    100 Gauss-Seidel sweeps per step), plus one adaptive Crank-Nicolson integration."""
    from multigrid.applications import heat_equation as ref_heat
    out = {}
    for name, (n, alpha, scheme, dt, steps, bc_kind, with_source) in heat_cases().items():
        g = Grid(n, n)
        hs = ref_heat.HeatEquationSolver(heat_config(ref_heat, alpha, bc_kind, with_source), g)
        u = hs.set_initial_condition().copy()
        out[f"{name}__u0"] = u.copy()
        sch = ref_heat.TimeSteppingScheme(scheme)
        if dt is None:
            dt = 0.2 * g.hx**2 / alpha if scheme == "explicit_euler" else 0.1 * g.hx
        for k in range(steps):
            u = hs._single_time_step(u, dt, sch)
            hs.current_time += dt
            out[f"{name}__u{k + 1}"] = u.copy()
        out[f"{name}__dt"] = np.array(dt)
    # adaptive step doubling, Crank-Nicolson, 17^2
    g = Grid(17, 17)
    hs = ref_heat.HeatEquationSolver(heat_config(ref_heat, 1.0, "zero", False), g)
    hs.set_initial_condition()
    res = hs.solve_time_dependent(t_final=0.02, dt_initial=0.004, scheme=ref_heat.TimeSteppingScheme.CRANK_NICOLSON,
                                  adaptive=True, error_tolerance=2e-3)
    out["adaptive17__final"] = res["final_solution"]
    out["adaptive17__times"] = np.array(res["time_history"])
    out["adaptive17__dts"] = np.array(res["dt_history"])
    out["adaptive17__steps"] = np.array(res["total_steps"])
    save("heat.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "solves", "large", "heat"]
    if "heat" in which:
        gen_heat()
    if "ops" in which:
        gen_ops()
    if "solves" in which:
        gen_solves()
    if "large" in which:
        gen_large()
    if "large4097" in which:
        gen_large4097()
