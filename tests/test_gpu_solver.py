"""GPU parity of the whole V/W/F-cycle driver against the reference's golden residual histories and
final solutions (tests/golden/solves.npz, produced by the reference), through
MultigridSolver.setup/solve -> mg_create/mg_solve.

north_star tolerances: 1e-12 relative l-inf (fp64), 1e-5 (mixed)."""
import re

import numpy as np
import pytest

import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
from oracle import mg_oracle as O

pytestmark = pytest.mark.gpu

SOLVE_RE = re.compile(r"n(\d+)(?:x(\d+))?_(?:(nondyadic|random)_)?L(\d+)_([VWF])(\d\d)?_([a-z0-9]+)_(float64|float32|mixed|adaptive_ref)__hist")


def _smoother(name):
    return {"jacobi08": lambda: mg.WeightedJacobiSmoother(),
            "vjacobi08": lambda: mg.EnhancedJacobiSolver(relaxation_parameter=0.8),
            "jacobi23": lambda: mg.JacobiSmoother(),
            "rbgs": lambda: mg.GaussSeidelSmoother(red_black=True),
            "rbgs15": lambda: mg.GaussSeidelSmoother(red_black=True, relaxation_parameter=1.15),
            # the lexgs goldens were produced by the reference's DEFAULT smoother (setup(smoother=None),
            # solvers/multigrid.py:112-117): go through the same default here
            "lexgs": lambda: None}[name]()


def _run(g, key, coarse_direct=None):
    m = SOLVE_RE.fullmatch(key)
    nx = int(m.group(1)); ny = int(m.group(2) or nx)
    special, L, cyc, vv, sm, prec = m.group(3), int(m.group(4)), m.group(5), m.group(6), m.group(7), m.group(8)
    pre, post = (int(vv[0]), int(vv[1])) if vv else (2, 2)
    dom = (0.0, 1.5, -0.2, 0.5) if special == "nondyadic" else (0.0, 1.0, 0.0, 1.0)
    dtype = np.float32 if prec == "float32" else np.float64
    grid = mg.Grid(nx, ny, dom, dtype)
    if special == "random":
        rhs, u0, maxit = g["n33_random_rhs"], g["n33_random_u0"], 8
    else:
        rhs, u0 = O.sine_rhs(nx, ny, dom, dtype).astype(dtype), None
        maxit = 12 if prec in ("float32", "adaptive_ref") else 30
    pm = None
    if prec == "mixed":
        pm = mg.PrecisionManager(default_precision="mixed")
    elif prec == "adaptive_ref":
        pm = mg.PrecisionManager()
    op = mg.LaplacianOperator(coefficient=-1.0)
    s = mg.MultigridSolver(max_levels=L, max_iterations=maxit, tolerance=1e-10, cycle_type=cyc,
                           pre_smooth_iterations=pre, post_smooth_iterations=post, coarse_direct=coarse_direct)
    s.setup(grid, op, mg.RestrictionOperator("full_weighting"), mg.ProlongationOperator("bilinear"), smoother=_smoother(sm))
    u, info = s.solve(grid, op, rhs, u0, pm)
    s.cleanup()
    return u, info, pm


def _keys(g):
    return sorted(k for k in g.files if k.endswith("__hist"))


@pytest.mark.parametrize("idx", range(28))
def test_golden_histories_and_solutions(golden_solves, idx):
    """coarse_direct=False: the reference's coarsest-grid iteration, reproduced sweep for sweep -- the strict parity run."""
    g = golden_solves
    keys = _keys(g)
    if idx >= len(keys):
        pytest.skip("fewer golden solves than slots")
    key = keys[idx]
    ref_hist = g[key]
    u, info, pm = _run(g, key, coarse_direct=False)
    hist = np.array(info["residual_history"])
    ukey = key.replace("__hist", "__u")
    if key.endswith("_float64__hist"):
        assert len(hist) == len(ref_hist), (key, hist, ref_hist)
        # the norm is an fp64 reduction in a different order: 1e-10 relative, floor at fp64 round-off of the norm
        np.testing.assert_allclose(hist, ref_hist, rtol=1e-9, atol=5e-14, err_msg=key)
        assert info["iterations"] == len(ref_hist) and info["converged"] == bool(ref_hist[-1] < 1e-10)
        if ukey in g.files:
            ref_u = g[ukey]
            rel = np.max(np.abs(u - ref_u)) / np.max(np.abs(ref_u))
            assert rel <= 1e-12, (key, rel)
    elif key.endswith("_mixed__hist"):
        # per-level mixed: every element-wise operator is bit-exact in both precisions and the coarsest solve stops on the
        # same sweep, so the iterate is the reference's bit for bit; the norm differs by its fp64 reduction order only
        assert len(hist) == len(ref_hist)
        np.testing.assert_allclose(hist, ref_hist, rtol=1e-12, atol=5e-14, err_msg=key)
        np.testing.assert_array_equal(u, g[ukey])
    elif key.endswith("_float32__hist"):
        # Grid(dtype=float32): the reference reduces its norm in fp32 (pairwise), we accumulate in fp64 -- the only
        # difference (measured <= 3e-6 relative, tools/fp32_parity_probe.py); the fp32 coarsest solve never meets 1e-12
        # on either side (1000 sweeps), so the iterate is the reference's bit for bit, stalled at the same floor
        assert len(hist) == len(ref_hist)
        np.testing.assert_allclose(hist, ref_hist, rtol=2e-5, err_msg=key)
        assert u.dtype == np.float32
        np.testing.assert_array_equal(u, g[ukey])
    else:   # adaptive_ref: the reference's own rule drops to fp32 at iteration 1 and never returns (F11)
        assert [p.value for p in pm.precision_history] == list(g[key.replace("__hist", "__precisions")])
        assert len(hist) == len(ref_hist) and not info["converged"]
        np.testing.assert_allclose(hist, ref_hist, rtol=2e-5, err_msg=key)
        assert u.dtype == np.float32 and g[ukey].dtype == np.float32
        np.testing.assert_array_equal(u, g[ukey])


@pytest.mark.parametrize("idx", range(28))
def test_golden_solves_with_the_default_coarsest_solve(golden_solves, idx):
    """The default (coarse_direct=None): V-cycles iterate on the coarsest grid like the reference and must meet the strict
    bars above; W- and F-cycles solve its nine unknowns directly.  A direct solve meets coarse_tolerance (1e-12) exactly
    where the reference's iteration stops anywhere below it, so histories agree to max(1e-9 relative, coarse_tolerance
    absolute) -- the accuracy the reference configured its own coarse solver for -- and iterates to 1e-12 relative l-inf
    (fp64) / 1e-5 (fp32, mixed), north_star's bars."""
    g = golden_solves
    keys = _keys(g)
    if idx >= len(keys):
        pytest.skip("fewer golden solves than slots")
    key = keys[idx]
    m = SOLVE_RE.fullmatch(key)
    if m is None or m.group(5) == "V" or key.endswith("adaptive_ref__hist"):
        pytest.skip("V-cycles take the iteration under the default too (covered by the strict run)")
    ref_hist = g[key]
    u, info, pm = _run(g, key, coarse_direct="auto")
    hist = np.array(info["residual_history"])
    assert len(hist) == len(ref_hist), (key, hist, ref_hist)
    loose = key.endswith("_float32__hist")
    np.testing.assert_allclose(hist, ref_hist, rtol=2e-5 if loose else 1e-9, atol=1e-12, err_msg=key)
    ukey = key.replace("__hist", "__u")
    if ukey in g.files:
        ref_u = g[ukey]
        rel = np.max(np.abs(u.astype(np.float64) - ref_u)) / np.max(np.abs(ref_u))
        assert rel <= (1e-5 if (loose or key.endswith("_mixed__hist")) else 1e-12), (key, rel)


def test_info_dict_contract(golden_solves):
    """Keys of solvers/base.py:159-171 + solvers/multigrid.py:382-389."""
    u, info, _ = _run(golden_solves, "n33_L4_V_jacobi08_float64__hist")
    for k in ["converged", "iterations", "final_residual", "convergence_rate", "residual_history", "total_time",
              "average_time_per_iteration", "precision_levels_used", "cycle_type", "num_levels", "grid_hierarchy",
              "level_timings", "pre_smooth_iterations", "post_smooth_iterations"]:
        assert k in info, k
    assert info["grid_hierarchy"] == [(33, 33), (17, 17), (9, 9), (5, 5)] and info["num_levels"] == 4
    assert info["precision_levels_used"] == ["double"]


def test_config2_1025_fp64_history(golden_large):
    """BASELINE config 2: 1025^2 fp64 V(2,2) Jacobi 0.8, 9 levels: the reference's 16-cycle history
    (stalls at 2.0e-10), a strided sample of its solution, and its norms."""
    n = 1025
    grid = mg.Grid(n, n)
    op = mg.LaplacianOperator(coefficient=-1.0)
    s = mg.MultigridSolver(max_levels=9, max_iterations=16, tolerance=1e-10)
    s.setup(grid, op, mg.RestrictionOperator(), mg.ProlongationOperator(), smoother=mg.EnhancedJacobiSolver(relaxation_parameter=0.8))
    u, info = s.solve(grid, op, O.sine_rhs(n, n))
    s.cleanup()
    ref = golden_large["hist"]
    hist = np.array(info["residual_history"])
    assert len(hist) == len(ref) == 16
    np.testing.assert_allclose(hist[:11], ref[:11], rtol=1e-7)        # above the round-off floor of the norm
    assert np.all(np.abs(hist[11:] - ref[11:]) < 5e-11)                # at the floor (2e-10): same plateau
    rel = np.max(np.abs(u[::32, ::32] - golden_large["u_sample"])) / float(golden_large["u_linf"])
    assert rel <= 1e-12, rel
    np.testing.assert_allclose(np.max(np.abs(u)), float(golden_large["u_linf"]), rtol=1e-12)
    np.testing.assert_allclose(np.sqrt(np.sum(u * u)), float(golden_large["u_l2"]), rtol=1e-12)


def test_config3_4097_adaptive_vs_fp64():
    """BASELINE config 3: 4097^2 adaptive fp32->fp64 (switch_threshold 1e-6) against the fp64 run of
    the same engine: <= 1e-5 relative l-inf (north_star).  At h = 1/4096 an fp32 residual bottoms out at
    eps32 diag(A) ||u|| ~ 0.2 ||r_0||: less than one cycle of use, so the policy declines the fp32 phase a priori
    (switch_reason "fp32_skipped", include/mghip.h) and the solve is the double solve, bit for bit."""
    n = 4097
    f = lambda x, y: 2 * np.pi**2 * np.sin(np.pi * x) * np.sin(np.pi * y)
    prob = mg.PoissonProblem(f, nx=n, ny=n, analytical_solution=lambda x, y: np.sin(np.pi * x) * np.sin(np.pi * y))
    u64, i64 = mg.MixedPrecisionMultigrid("double", tolerance=1e-7, max_iterations=25).solve(prob)
    ua, ia = mg.MixedPrecisionMultigrid("adaptive", switch_threshold=1e-6, tolerance=1e-7, max_iterations=40).solve(prob)
    assert i64["converged"] and ia["converged"]
    assert set(ia["precision_levels_used"]) == {"float64"} and ia["switch_reason"] == "fp32_skipped"
    assert ia["iterations"] == i64["iterations"] and np.array_equal(ua, u64)
    assert np.max(np.abs(ua - u64)) / np.max(np.abs(u64)) <= 1e-5
    assert ia["max_error"] < 1e-6 and i64["max_error"] < 1e-6          # O(h^2) discretisation error


def test_config3_shape_1025_adaptive_switches_and_matches_fp64():
    """The same configuration at 1025^2, where the fp32 floor (0.013 ||r_0||) leaves the fp32 phase two cycles: the phase is
    taken, ends for one of the three reasons the policy knows, and the result matches the double solve to <= 1e-5."""
    n = 1025
    f = lambda x, y: 2 * np.pi**2 * np.sin(np.pi * x) * np.sin(np.pi * y)
    prob = mg.PoissonProblem(f, nx=n, ny=n, analytical_solution=lambda x, y: np.sin(np.pi * x) * np.sin(np.pi * y))
    u64, i64 = mg.MixedPrecisionMultigrid("double", tolerance=1e-7, max_iterations=25).solve(prob)
    ua, ia = mg.MixedPrecisionMultigrid("adaptive", switch_threshold=1e-6, tolerance=1e-7, max_iterations=40).solve(prob)
    assert i64["converged"] and ia["converged"]
    assert set(ia["precision_levels_used"]) == {"float32", "float64"}
    assert ia["switch_reason"] in ("threshold", "stagnation", "fp32_floor")
    assert np.max(np.abs(ua - u64)) / np.max(np.abs(u64)) <= 1e-5
    assert ia["max_error"] < 1e-5 and i64["max_error"] < 1e-5


def test_device_resident_stepping_equals_solve():
    n = 257
    rhs = O.sine_rhs(n, n)
    eng = mg.MultigridEngine(n, n, max_levels=7, smoother=0, omega=0.8)
    u, r = eng.solve(rhs, tol=1e-30, max_iterations=5)
    eng.set_rhs(rhs); eng.set_solution(None)
    hist = []
    for _ in range(5):
        eng.cycle(1)
        hist.append(eng.residual_norm())
    np.testing.assert_array_equal(eng.get_solution(), u)
    np.testing.assert_array_equal(hist, r["residual_history"])
    eng.close()


@pytest.mark.parametrize("sm,omega", [("jacobi", 0.8), ("rbgs", 1.0), ("rbgs", 1.15)])
@pytest.mark.parametrize("prec", ["double", "single", "mixed", "adaptive"])
@pytest.mark.parametrize("n,cyc,pre,post", [(257, "V", 2, 2), (129, "W", 2, 2), (513, "V", 1, 1), (129, "V", 3, 0), (65, "F", 0, 4),
                                            ((97, 193), "V", 2, 1), (1025, "V", 2, 2)])
def test_fused_legs_equal_one_launch_per_operator(prec, n, cyc, pre, post, sm, omega):
    """The fused down/up legs (temporal blocking in LDS) must reproduce the operator-by-operator cycle bit for bit,
    including the norm they accumulate on the way (to reduction round-off)."""
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
    nx, ny = (n, n) if isinstance(n, int) else n
    code = {"double": _lib.MG_PREC_DOUBLE, "single": _lib.MG_PREC_SINGLE, "mixed": _lib.MG_PREC_MIXED_LEVELS,
            "adaptive": _lib.MG_PREC_ADAPTIVE}[prec]
    rng = np.random.default_rng(nx + ny + pre)
    rhs = O.sine_rhs(nx, ny) + 0.05 * rng.standard_normal((nx, ny))
    u0 = rng.standard_normal((nx, ny))
    res = []
    for fused, tail in ((True, True), (True, False), (False, False)):
        eng = mg.MultigridEngine(nx, ny, max_levels=mg.default_max_levels(nx, ny), cycle=cyc, pre=pre, post=post,
                                 smoother=_lib.MG_JACOBI if sm == "jacobi" else _lib.MG_RBGS, omega=omega, precision=code,
                                 switch_threshold=1e-3, coarse_maxit=60, fused=fused, tail=tail, speculate=tail)
        u, r = eng.solve(rhs, u0, tol=1e-30, max_iterations=6)
        eng.close()
        res.append((u, r))
    (ut, rt), (uf, rf), (uu, ru) = res
    np.testing.assert_array_equal(uf, uu)
    np.testing.assert_array_equal(ut, uu)
    np.testing.assert_allclose(rf["residual_history"], ru["residual_history"], rtol=1e-11)
    np.testing.assert_allclose(rt["residual_history"], ru["residual_history"], rtol=1e-11)
    assert rf["precision_codes"] == ru["precision_codes"] == rt["precision_codes"]


@pytest.mark.parametrize("sm,omega", [("jacobi", 0.8), ("rbgs", 1.0), ("rbgs", 1.15)])
@pytest.mark.parametrize("prec", ["double", "single", "mixed", "adaptive"])
@pytest.mark.parametrize("n,cyc,pre,post", [(257, "V", 2, 2), (129, "W", 2, 2), (513, "V", 1, 1), (129, "V", 3, 0), (65, "F", 0, 4),
                                            ((97, 193), "V", 2, 1), ((449, 131), "V", 2, 2), (2049, "V", 2, 2)])
def test_register_blocked_legs_equal_lds_tiled_legs(prec, n, cyc, pre, post, sm, omega):
    """fused = 3 runs every level on the register-blocked legs (iterate in registers, strip edges through a small LDS
    exchange, lateral neighbours by DPP, full weighting in registers), fused = 2 the large levels only: both must
    reproduce the LDS-tiled legs (fused = 1, themselves equal to one launch per operator) bit for bit."""
    nx, ny = (n, n) if isinstance(n, int) else n
    code = {"double": _lib.MG_PREC_DOUBLE, "single": _lib.MG_PREC_SINGLE, "mixed": _lib.MG_PREC_MIXED_LEVELS,
            "adaptive": _lib.MG_PREC_ADAPTIVE}[prec]
    rng = np.random.default_rng(nx + 3 * ny + pre)
    rhs = O.sine_rhs(nx, ny) + 0.05 * rng.standard_normal((nx, ny))
    u0 = rng.standard_normal((nx, ny))
    res = []
    for fused in (3, 2, 1):
        eng = mg.MultigridEngine(nx, ny, max_levels=mg.default_max_levels(nx, ny), cycle=cyc, pre=pre, post=post,
                                 smoother=_lib.MG_JACOBI if sm == "jacobi" else _lib.MG_RBGS, omega=omega, precision=code,
                                 switch_threshold=1e-3, coarse_maxit=60, fused=fused)
        u, r = eng.solve(rhs, u0, tol=1e-30, max_iterations=4)
        eng.close()
        res.append((u, r))
    (u3, r3), (u2, r2), (u1, r1) = res
    np.testing.assert_array_equal(u3, u1)
    np.testing.assert_array_equal(u2, u1)
    np.testing.assert_allclose(r3["residual_history"], r1["residual_history"], rtol=1e-11)
    np.testing.assert_allclose(r2["residual_history"], r1["residual_history"], rtol=1e-11)
    assert r3["precision_codes"] == r1["precision_codes"]


@pytest.mark.parametrize("prec,thr", [("double", 1e-6), ("adaptive", 1e-3), ("adaptive", 1e-6), ("mixed", 1e-6)])
def test_speculative_launching_changes_nothing(prec, thr):
    """mg_iterate queues the front part of cycle k+1 while ||r_k|| is in flight; stopping on tolerance and precision
    switches must leave exactly the iterate, history and switch points of the one-cycle-at-a-time loop."""
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
    code = {"double": _lib.MG_PREC_DOUBLE, "mixed": _lib.MG_PREC_MIXED_LEVELS, "adaptive": _lib.MG_PREC_ADAPTIVE}[prec]
    n = 257
    rhs = O.sine_rhs(n, n)
    out = []
    for spec in (True, False):
        eng = mg.MultigridEngine(n, n, max_levels=7, smoother=_lib.MG_JACOBI, omega=0.8, precision=code, switch_threshold=thr,
                                 speculate=spec)
        u, r = eng.solve(rhs, tol=1e-9, max_iterations=40)
        u2, r2 = eng.solve(rhs, tol=1e-30, max_iterations=7)        # ends on max_iter with a front part never queued
        eng.close()
        out.append((u, r, u2, r2))
    (ua, ra, ua2, ra2), (ub, rb, ub2, rb2) = out
    assert ra["converged"] and ra["iterations"] == rb["iterations"] and ra["precision_codes"] == rb["precision_codes"]
    np.testing.assert_array_equal(ua, ub); np.testing.assert_array_equal(ua2, ub2)
    np.testing.assert_array_equal(ra["residual_history"], rb["residual_history"])
    np.testing.assert_array_equal(ra2["residual_history"], rb2["residual_history"])


@pytest.mark.parametrize("prec", ["double", "adaptive", "single"])
def test_repeated_solves_and_adaptive_switches_leave_no_trace(prec):
    """What the adaptive path keeps between solves -- the norm of the zero iterate per right-hand side, the injected coarse
    rhs rings per working precision, the ring-only ping-pong partner after a precision switch, a new guess stored straight
    into the fp64 iterate -- must not be visible: a second solve of the same right-hand side, a solve after another
    right-hand side, and a solve from a non-zero guess all equal a fresh engine's, bit for bit."""
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
    code = {"double": _lib.MG_PREC_DOUBLE, "single": _lib.MG_PREC_SINGLE_MANAGED, "adaptive": _lib.MG_PREC_ADAPTIVE}[prec]
    n = 513
    rng = np.random.default_rng(3)
    rhs_a, rhs_b = O.sine_rhs(n, n), rng.standard_normal((n, n))
    u0 = rng.standard_normal((n, n))
    kw = dict(max_levels=8, smoother=_lib.MG_JACOBI, omega=0.8, precision=code, switch_threshold=1e-4)

    def fresh(rhs, guess, iters):
        eng = mg.MultigridEngine(n, n, **kw)
        eng.set_rhs(rhs); eng.set_solution(guess)
        r = eng.iterate(0.0, iters)
        u = eng.get_solution()
        eng.close()
        return u, r

    eng = mg.MultigridEngine(n, n, **kw)
    eng.set_rhs(rhs_a)
    runs, resident = [], rhs_a
    for rhs, guess, iters in ((rhs_a, None, 3), (rhs_a, None, 9), (rhs_b, None, 4), (rhs_a, u0, 9), (rhs_a, None, 2)):
        if rhs is not resident:
            eng.set_rhs(rhs)
            resident = rhs
        eng.set_solution(guess)
        r = eng.iterate(0.0, iters)
        runs.append((rhs, guess, iters, eng.get_solution(), r))
    eng.close()
    for rhs, guess, iters, u, r in runs:
        u_ref, r_ref = fresh(rhs, guess, iters)
        np.testing.assert_array_equal(u, u_ref)
        assert r["residual_history"] == r_ref["residual_history"] and r["initial_residual"] == r_ref["initial_residual"]
        assert r["precision_codes"] == r_ref["precision_codes"]
    if prec == "adaptive":
        assert 0 in runs[1][4]["precision_codes"] and 1 in runs[1][4]["precision_codes"]      # the 9-cycle solve did switch


def test_bench_size_4097_two_cycles_equal_oracle():
    """At the bench size itself: two V(2,2) Jacobi cycles of the fused fp64 engine at 4097^2 against the NumPy oracle
    (about 1 s per oracle cycle) -- bit-identical iterate, norms to reduction round-off."""
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
    n = 4097
    rhs = O.sine_rhs(n, n)
    ref = O.MGOracle(n, n, max_levels=11, cycle="V", smoother="jacobi", omega=0.8, jacobi_form="vectorized")
    ref.rhs[0] = rhs.copy()
    u_ref = np.zeros_like(rhs)
    h_ref = []
    for _ in range(2):
        u_ref = ref.cycle_once(u_ref, 0)
        h_ref.append(ref.residual_norm(u_ref, rhs, 0))
    eng = mg.MultigridEngine(n, n, max_levels=11, smoother=_lib.MG_JACOBI, omega=0.8)
    u, r = eng.solve(rhs, tol=0.0, max_iterations=2)
    eng.close()
    np.testing.assert_array_equal(u, u_ref)
    np.testing.assert_allclose(r["residual_history"], h_ref, rtol=1e-11)


def test_bench_size_4097_equals_the_reference_capture(golden_large4097):
    """The reference's OWN CPU solver at the bench size (two V(2,2) Jacobi-0.8 cycles at 4097^2 fp64, 27 s each on one core:
    tests/golden/large_4097.npz): residual history, a strided sample of the iterate and its norms, <= 1e-12."""
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
    n = 4097
    g = golden_large4097
    eng = mg.MultigridEngine(n, n, max_levels=11, smoother=_lib.MG_JACOBI, omega=0.8)
    u, r = eng.solve(O.sine_rhs(n, n), tol=0.0, max_iterations=2)
    eng.close()
    np.testing.assert_allclose(r["residual_history"], g["hist"], rtol=1e-10)
    assert np.max(np.abs(u[::128, ::128] - g["u_sample"])) <= 1e-12 * float(g["u_linf"])
    np.testing.assert_allclose(np.max(np.abs(u)), float(g["u_linf"]), rtol=1e-12)
    np.testing.assert_allclose(np.sqrt(np.sum(u * u)), float(g["u_l2"]), rtol=1e-12)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("n,cyc,kind,omega,ncyc", [(129, "V", "jacobi", 0.8, 1), (65, "W", "rbgs", 1.0, 2), (257, "V", "rbgs", 1.0, 1)])
def test_fmg_initial_guess_equals_oracle(n, cyc, kind, omega, ncyc, fused):
    """Full-multigrid start (solvers/advanced_multigrid.py:626-683): restricted rhs hierarchy, coarsest solve, prolongate +
    cycles per level.  Bit-identical to the oracle restatement; one FMG pass reaches discretisation-level accuracy."""
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
    rhs = O.sine_rhs(n, n)
    rhs[0, :] = rhs[-1, :] = rhs[:, 0] = rhs[:, -1] = 0.0
    levels = mg.default_max_levels(n, n)
    ref = O.MGOracle(n, n, max_levels=levels, cycle=cyc, smoother=kind, omega=omega, jacobi_form="vectorized")
    u_ref = ref.fmg_init(rhs, ncyc)
    eng = mg.MultigridEngine(n, n, max_levels=levels, cycle=cyc, smoother=_lib.MG_JACOBI if kind == "jacobi" else _lib.MG_RBGS,
                             omega=omega, fused=fused)
    eng.set_rhs(rhs); eng.set_solution(None); eng.fmg(ncyc)
    u = eng.get_solution()
    assert np.max(np.abs(u - u_ref)) <= 1e-13 * np.max(np.abs(u_ref))
    x = np.linspace(0, 1, n)
    exact = np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
    assert np.max(np.abs(u - exact)) < 3.0 * (np.pi**2 / 12.0) * (1.0 / (n - 1))**2 * 2      # within a small factor of the h^2 truncation error
    eng.close()
    # through the solver front-end: an FMG start saves cycles
    prob = mg.PoissonProblem(lambda x, y: 2 * np.pi**2 * np.sin(np.pi * x) * np.sin(np.pi * y), nx=n, ny=n)
    _, plain = mg.MixedPrecisionMultigrid("double", tolerance=1e-8, smoother="jacobi" if kind == "jacobi" else "gauss_seidel").solve(prob)
    _, fmg = mg.MixedPrecisionMultigrid("double", tolerance=1e-8, smoother="jacobi" if kind == "jacobi" else "gauss_seidel", use_fmg=True).solve(prob)
    assert fmg["converged"] and fmg["iterations"] < plain["iterations"]


@pytest.mark.parametrize("sm,fused", [(0, True), (1, True), (1, False), (2, False)])
def test_fmg_ignores_call_history(sm, fused):
    """mg_fmg builds its guess from the rhs and the Dirichlet ring alone: the interior of an earlier iterate (an initial
    guess, or whatever previous cycles left in the ping-pong buffers) must not leak into it."""
    n = 129
    rng = np.random.default_rng(5)
    rhs = O.sine_rhs(n, n)
    ring = np.zeros((n, n)); ring[0, :] = rng.standard_normal(n); ring[:, -1] = rng.standard_normal(n)
    junk = ring + np.pad(rng.standard_normal((n - 2, n - 2)), 1)
    eng = mg.MultigridEngine(n, n, max_levels=6, smoother=sm, omega=0.8 if sm == 0 else 1.0, fused=fused)
    eng.set_rhs(rhs)
    eng.set_solution(ring); eng.fmg(1)
    u_clean = eng.get_solution()
    eng.set_solution(junk); eng.cycle(3); eng.fmg(1)
    u_after = eng.get_solution()
    eng.close()
    np.testing.assert_array_equal(u_after[0, :], ring[0, :]); np.testing.assert_array_equal(u_after[:, -1], ring[:, -1])
    np.testing.assert_array_equal(u_after, u_clean)


def test_default_smoother_is_the_references_lexicographic_gs(golden_solves):
    """setup(smoother=None) == setup(smoother=GaussSeidelSmoother()) (solvers/multigrid.py:112-117), not red-black."""
    grid = mg.Grid(33, 33)
    op = mg.LaplacianOperator(coefficient=-1.0)
    rhs = O.sine_rhs(33, 33)
    out = []
    for sm in (None, mg.GaussSeidelSmoother(), mg.GaussSeidelSmoother(red_black=True)):
        s = mg.MultigridSolver(max_levels=4, max_iterations=30, tolerance=1e-10)
        s.setup(grid, op, mg.RestrictionOperator("full_weighting"), mg.ProlongationOperator("bilinear"), smoother=sm)
        assert s.smoother.red_black == (sm is not None and sm.red_black)
        out.append(s.solve(grid, op, rhs))
        s.cleanup()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert out[0][1]["residual_history"] == out[1][1]["residual_history"]
    assert out[0][1]["residual_history"] != out[2][1]["residual_history"]
    np.testing.assert_allclose(out[0][1]["residual_history"], golden_solves["n33_L4_V_lexgs_float64__hist"], rtol=1e-9, atol=5e-14)


def test_config5_size_16385_mixed_w_rbgs_single_gpu():
    """BASELINE config 5's problem on ONE GPU (268 M unknowns, per-level mixed precision, W(2,2) red-black GS):
    the fused legs equal the one-launch-per-operator cycle bit for bit and the cycle contracts like it does on
    small grids (h-independent convergence)."""
    n = 16385
    x = np.linspace(0.0, 1.0, n)
    rhs = (2 * np.pi**2) * np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
    out = []
    for fused in (2, 0):          # register-blocked legs on the large levels (LDS-tiled below, LDS tail) vs one launch per operator
        eng = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle="W", smoother=_lib.MG_RBGS, omega=1.0,
                                 precision=_lib.MG_PREC_MIXED_LEVELS, fused=fused)
        u, r = eng.solve(rhs, tol=0.0, max_iterations=3)
        eng.close()
        out.append((u, r["residual_history"]))
    assert np.array_equal(out[0][0], out[1][0])
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-12)
    h = out[0][1]
    # the sine mode is so smooth on this grid that one W-cycle takes ||r|| from 9.87 to ~3e-7; the fp64 rounding
    # floor eps * ||A|| ~ 1e-16 * 4 / h^2 sits at a few 1e-8
    assert h[0] < 1e-6 and h[1] < 0.5 * h[0] and h[2] < 1.1 * h[1] and h[2] < 1e-7, h
    exact = np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :]
    assert np.max(np.abs(out[0][0] - exact)) < 1e-6


@pytest.mark.parametrize("key", ["n33_L4_W_jacobi08_float64__hist", "n65_L5_W_rbgs_float64__hist", "n129_L6_W_rbgs_float64__hist",
                                 "n65_L5_W_jacobi08_float64__hist", "n129_L6_V_vjacobi08_float64__hist"])
def test_direct_coarsest_solve_stays_within_the_parity_bar(golden_solves, key):
    """mg_config.coarse_direct = 1 replaces the reference's Gauss-Seidel iteration on the 5 x 5 coarsest grid (to 1e-12,
    solvers/multigrid.py:119-124) by the exact solve of its nine unknowns.  Not bit-identical by construction -- the
    iteration may leave an error of ||A^-1|| 1e-12 / h ~ 2e-13 per visit -- but the reference's W-cycle goldens are
    reproduced within north_star's 1e-12 relative l-inf, with the same iteration count."""
    m = SOLVE_RE.fullmatch(key)
    n, L, cyc, sm = int(m.group(1)), int(m.group(4)), m.group(5), m.group(7)
    kind, omega = (_lib.MG_RBGS, 1.0) if sm == "rbgs" else (_lib.MG_JACOBI, 0.8)
    rhs = O.sine_rhs(n, n)
    ref_hist, ref_u = golden_solves[key], golden_solves[key.replace("__hist", "__u")] if key.replace("__hist", "__u") in golden_solves.files else None
    out = {}
    for direct in (True, False):
        eng = mg.MultigridEngine(n, n, max_levels=L, cycle=cyc, smoother=kind, omega=omega, coarse_direct=direct)
        out[direct] = eng.solve(rhs, tol=1e-10, max_iterations=30)
        eng.close()
    u, r = out[True]
    assert r["iterations"] == len(ref_hist) and r["last_coarse_sweeps"] == 0 and out[False][1]["last_coarse_sweeps"] >= 1
    np.testing.assert_allclose(r["residual_history"], ref_hist, rtol=1e-6, atol=5e-13)
    if ref_u is not None:
        assert np.max(np.abs(u - ref_u)) / np.max(np.abs(ref_u)) <= 1e-12
    assert np.max(np.abs(u - out[False][0])) / np.max(np.abs(u)) <= 1e-12 and not np.array_equal(u, out[False][0])
