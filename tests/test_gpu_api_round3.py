"""Round 3: the decomposed engine behind the reference's multi-GPU solver API, the GPU-side class names the reference's own
callers import (gpu/__init__.py:3-6, applications/poisson_solver.py:15-19), the register-resident coarse tail, and the
defects ADVICE r02 listed -- on a real MI355X, through the C ABI."""
import ctypes as C
import os
import time

import numpy as np
import pytest

import mixed_precision_multigrid_solvers_for_pdes_amd as mg
from mixed_precision_multigrid_solvers_for_pdes_amd import _lib
from mixed_precision_multigrid_solvers_for_pdes_amd import distributed as D
from oracle import mg_oracle as O

pytestmark = pytest.mark.gpu


def _plugins():
    return mg.LaplacianOperator(coefficient=-1.0), mg.RestrictionOperator("full_weighting"), mg.ProlongationOperator("bilinear")


# ---------------------------------------------------------------- DistributedMultigridSolver (gpu/multi_gpu.py:301-750) ----
@pytest.mark.parametrize("ids,strategy,key,n,levels,cyc,smoother,agg", [
    ([0, 0], "stripe", "n65_L5_V_jacobi08_float64", 65, 5, "V", "jacobi", 17),
    ([0, 0, 0, 0], "checkerboard", "n129_L6_W_rbgs_float64", 129, 6, "W", "gauss_seidel", 33),
    ([0, 0, 0, 0], "block", "n129_L6_V_vjacobi08_float64", 129, 6, "V", "jacobi", 33),
])
def test_distributed_solver_api_virtual_ranks_equal_reference_golden(golden_solves, ids, strategy, key, n, levels, cyc, smoother, agg):
    """setup / solve through the facade class on px x py virtual ranks with the real kernels (native cycle plans included):
    the reference's own golden solve -- history to 1e-9, iterate to 1e-12 -- and the single-GPU solver's iterate bit for bit."""
    op, R, P = _plugins()
    grid = mg.Grid(n, n)
    s = mg.DistributedMultigridSolver(device_ids=ids, decomposition_strategy=strategy, agglomerate_at=agg, max_levels=levels,
                                      max_iterations=30, tolerance=1e-10, cycle_type=cyc, smoother=smoother)
    s.setup(grid, op, R, P)
    rhs = O.sine_rhs(n, n)
    u, info = s.solve(grid, op, rhs)
    ref_hist, ref_u = golden_solves[key + "__hist"], golden_solves[key + "__u"]
    assert info["iterations"] == len(ref_hist) and info["converged"]
    np.testing.assert_allclose(info["residual_history"], ref_hist, rtol=1e-9, atol=5e-14)
    assert float(np.max(np.abs(u - ref_u)) / np.max(np.abs(ref_u))) <= 1e-12
    assert info["n_gpus"] == len(ids) and info["decomposed"] and info["exchanges_per_cycle"] > 0
    assert info["native_plan_cycles"] >= info["iterations"] - 1            # the first cycle records, the rest replay
    single = mg.GPUMultigridSolver(max_levels=levels, max_iterations=30, tolerance=1e-10, cycle_type=cyc, smoother=smoother)
    single.setup(grid, op, R, P)
    u1, info1 = single.solve(grid, op, rhs)
    np.testing.assert_array_equal(u, u1)
    np.testing.assert_allclose(info["residual_history"], info1["residual_history"][1:], rtol=1e-12)
    single.cleanup()
    s.cleanup()


def test_distributed_solver_adaptive_policy_equals_the_single_gpu_engine():
    """precision_manager = the one-way adaptive rule, 2 x 2 virtual ranks at 513^2: same trajectory (precision per cycle,
    history) and same final iterate as the single-GPU engine's adaptive solve."""
    n, levels = 513, 8
    op, R, P = _plugins()
    grid = mg.Grid(n, n)
    rhs = O.sine_rhs(n, n)

    def pm():
        m = mg.PrecisionManager(default_precision="double", adaptive=True, convergence_threshold=1e-4)
        m.reference_rule = False
        return m
    s = mg.DistributedMultigridSolver(device_ids=[0] * 4, decomposition_strategy="block", agglomerate_at=65, max_levels=levels,
                                      max_iterations=14, tolerance=1e-9, smoother="jacobi")
    s.setup(grid, op, R, P)
    u, info = s.solve(grid, op, rhs, precision_manager=pm())
    single = mg.GPUMultigridSolver(max_levels=levels, max_iterations=14, tolerance=1e-9, smoother="jacobi")
    single.setup(grid, op, R, P)
    u1, info1 = single.solve(grid, op, rhs, precision_manager=pm())
    assert info["iterations"] == info1["iterations"] and info["precision_switches"] >= 1
    np.testing.assert_allclose(info["residual_history"], info1["residual_history"][1:], rtol=1e-9)
    assert float(np.max(np.abs(u - u1)) / np.max(np.abs(u1))) <= 1e-12
    assert set(info["precision_levels_used"]) == set(info1["precision_levels_used"]) == {"float32", "float64"}
    single.cleanup()
    s.cleanup()


def test_facade_n_gpus_switch_and_small_grid_fallback():
    """MixedPrecisionMultigrid(n_gpus=4): the decomposed solve behind the README facade equals the single-GPU one; a grid
    no level of which can be cut runs the single-GPU engine on every rank (decomposed: False), never a CPU path."""
    f = lambda x, y: 2 * np.pi**2 * np.sin(np.pi * x) * np.sin(np.pi * y)
    prob = mg.PoissonProblem(f, nx=257, ny=257)
    u4, i4 = mg.MixedPrecisionMultigrid("double", n_gpus=4, agglomerate_at=65, tolerance=1e-9).solve(prob)
    u1, i1 = mg.MixedPrecisionMultigrid("double", tolerance=1e-9).solve(prob)
    np.testing.assert_array_equal(u4, u1)
    assert i4["n_gpus"] == 4 and i4["process_grid"] == (2, 2) and i4["iterations"] == i1["iterations"]
    op, R, P = _plugins()
    s = mg.DistributedMultigridSolver(device_ids=[0, 0], max_levels=4, tolerance=1e-9)
    g = mg.Grid(21, 21)                                   # 20 cells: nothing to cut evenly below the first level
    s.setup(g, op, R, P)
    u, info = s.solve(g, op, O.sine_rhs(21, 21))
    assert info["decomposed"] is False and info["process_grid"] == (1, 1) and np.all(np.isfinite(u))
    s.cleanup()


def test_multi_gpu_solver_dict_api():
    """MultiGPUSolver.domain_decomposition_solve (gpu/multi_gpu_solver.py:244-340): the dict its callers read."""
    n = 129
    s = mg.MultiGPUSolver(4, mg.DecompositionType.BLOCK_2D, max_levels=6, max_iterations=20, tolerance=1e-9, agglomerate_at=33)
    res = s.domain_decomposition_solve(mg.Grid(n, n), O.sine_rhs(n, n))
    for k in ("solution", "converged", "iterations", "final_residual", "residual_history", "solve_time", "num_gpus_used",
              "domain_decomposition", "performance_stats"):
        assert k in res, k
    assert res["converged"] and res["num_gpus_used"] == 4 and res["domain_decomposition"] == "block_2d"
    x = np.linspace(0, 1, n)
    assert np.max(np.abs(res["solution"] - np.sin(np.pi * x)[:, None] * np.sin(np.pi * x)[None, :])) < 1e-4
    s.cleanup()


# ---------------------------------------------------------------- GPUCommunicationAvoidingMultigrid, precision / memory managers ----
def test_gpu_ca_multigrid_fmg_start_and_poisson_solver_type():
    """gpu/gpu_solver.py:504-798: use_fmg starts from a full-multigrid guess (mg_fmg) when no initial guess is given;
    applications/poisson_solver.py:90-101 builds it for solver_type='gpu_ca_multigrid'."""
    n = 257
    op, R, P = _plugins()
    grid = mg.Grid(n, n)
    rhs = O.sine_rhs(n, n)
    res = {}
    for fmg in (False, True):
        s = mg.GPUCommunicationAvoidingMultigrid(max_levels=7, max_iterations=30, tolerance=1e-8, use_fmg=fmg, smoother="jacobi")
        s.setup(grid, op, R, P)
        u, info = s.solve(grid, op, rhs)
        assert info["ca_optimizations"] and info["fmg_used"] is fmg and info["ca_stats"]["fmg_initializations"] == int(fmg)
        assert "ca_optimizations" in s.get_performance_statistics()
        res[fmg] = (u, info)
        s.cleanup()
    # the FMG start is worth cycles: its initial residual is orders below ||f|| and the solve needs fewer iterations
    assert res[True][1]["initial_residual"] < 1e-2 * res[False][1]["initial_residual"]
    assert res[True][1]["iterations"] < res[False][1]["iterations"]
    assert np.max(np.abs(res[True][0] - res[False][0])) < 1e-7
    ps = mg.PoissonSolver2D(solver_type="gpu_ca_multigrid", max_levels=7, tolerance=1e-8)
    assert isinstance(ps.solver, mg.GPUCommunicationAvoidingMultigrid) and ps.solver.use_fmg


def test_gpu_precision_manager_policy_and_engine_bridge():
    pm = mg.GPUPrecisionManager()
    assert pm.current_precision == mg.GPUPrecisionLevel.MIXED_TC and pm.tensor_core_available is False
    assert pm.get_optimal_dtype("smoothing", 3) == np.float32 and pm.get_optimal_dtype("residual", 0) == np.float32
    assert pm.update_precision_adaptive(1e-9) and pm.current_precision == mg.GPUPrecisionLevel.SINGLE       # < 1e-8: leave MIXED_TC
    d = mg.GPUPrecisionManager(default_precision="double")
    assert d.get_optimal_dtype("smoothing", 2) == np.float64
    assert d.update_precision_adaptive(1.0) and d.current_precision == mg.GPUPrecisionLevel.SINGLE         # > 1e-2: downgrade
    st = d.get_precision_statistics()
    assert st["precision_switches"] == 1 and st["tensor_core_available"] is False and "precision_distribution" in st
    # the bridge: the engine policy a manager stands for, and a solve under it
    assert mg.GPUPrecisionManager(default_precision="double").to_engine_policy().current_precision == mg.PrecisionLevel.DOUBLE
    pol = mg.GPUPrecisionManager(default_precision="mixed_tc").to_engine_policy(switch_threshold=1e-4)
    assert pol.adaptive and pol.reference_rule is False
    n = 257
    op, R, P = _plugins()
    s = mg.GPUMultigridSolver(max_levels=7, max_iterations=20, tolerance=1e-9)
    s.setup(mg.Grid(n, n), op, R, P)
    u, info = s.solve(mg.Grid(n, n), op, O.sine_rhs(n, n), precision_manager=pol)
    assert info["converged"] and set(info["precision_levels_used"]) == {"float32", "float64"}
    s.cleanup()


def test_gpu_memory_manager_pool_and_pitched_fields():
    mm = mg.GPUMemoryManager(device_id=0, max_pool_size_mb=64.0)
    a = np.random.default_rng(0).standard_normal((129, 257))
    d = mm.to_gpu(a)
    assert tuple(d.shape) == a.shape and d.stride(0) * 8 % 512 == 0 and d.data_ptr() % 16 == 0     # pitched like mg_pitch_elems
    np.testing.assert_array_equal(mm.to_cpu(d), a)
    # a pooled field is what the device-pointer entry points take: residual of it through mg_dev_residual
    f = mm.to_gpu(np.zeros_like(a))
    r = mm.allocate_like_gpu(a)
    import torch
    g = mg.Grid(129, 257)
    _lib.check(_lib.load().mg_dev_residual(_lib.MG_F64, 129, 257, d.stride(0), g.hx, g.hy, -1.0, C.c_void_p(d.data_ptr()),
                                           C.c_void_p(f.data_ptr()), C.c_void_p(r.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(mm.to_cpu(r), O.residual(a, np.zeros_like(a), g.hx, g.hy, -1.0))
    pool = mm.memory_pool
    before = pool.get_statistics()
    pool.deallocate(r)
    r2 = mm.allocate_gpu_array(a.shape, np.float64)
    after = pool.get_statistics()
    assert r2.data_ptr() == r.data_ptr() and after["cache_hits"] == before["cache_hits"] + 1 and after["hit_rate"] > 0
    assert float(r2.abs().max()) == 0.0                                   # zero_fill
    assert mm.get_memory_usage()["gpu_memory_info"]["total_gpu_memory"] > 2**36
    mm.cleanup()
    assert mg.memory_manager.check_gpu_availability()["gpu_count"] >= 1


# ---------------------------------------------------------------- the register-resident coarse tail ----
@pytest.mark.parametrize("n", [33, 65, 129, 257])
@pytest.mark.parametrize("cyc,sm,om", [("V", _lib.MG_JACOBI, 0.8), ("W", _lib.MG_RBGS, 1.0), ("W", _lib.MG_JACOBI, 2.0 / 3.0),
                                       ("F", _lib.MG_RBGS, 1.15)])
def test_register_tail_equals_lds_tail_and_per_level_launches(n, cyc, sm, om):
    """mg_config.tail: 1 the register-resident one-workgroup tail (csrc/mg_tail_kernels.hpp), 2 the LDS one, 0 a launch pair
    per level.  Same arithmetic per cell: iterates and histories bit for bit, in every precision layout, with the
    reference's coarsest iteration and with the direct coarsest solve."""
    if cyc == "F" and n > 129:
        pytest.skip("the reference's F-cycle visits level l 2^(L-l-2) times")
    rng = np.random.default_rng(n)
    f64 = O.sine_rhs(n, n) + 0.1 * rng.standard_normal((n, n))
    for prec in (_lib.MG_PREC_DOUBLE, _lib.MG_PREC_SINGLE, _lib.MG_PREC_SINGLE_MANAGED, _lib.MG_PREC_MIXED_LEVELS):
        for direct in (False, True):
            res = {}
            for tail in (1, 2, 0):
                if tail == 0 and direct:
                    continue                      # per-level launches always iterate on the coarsest grid
                e = mg.MultigridEngine(n, n, max_levels=mg.default_max_levels(n, n), cycle=cyc, smoother=sm, omega=om,
                                       precision=prec, tail=tail, coarse_direct=direct, coarse_maxit=60)
                f = f64.astype(np.float32) if prec == _lib.MG_PREC_SINGLE else f64
                u, r = e.solve(f, tol=0.0, max_iterations=3)
                res[tail] = (u, r["residual_history"], r["last_coarse_sweeps"])
                e.close()
            for tail in res:
                np.testing.assert_array_equal(res[tail][0], res[1][0], err_msg=f"tail={tail} prec={prec} direct={direct}")
                assert res[tail][1] == res[1][1], (tail, prec, direct)
            assert (res[1][2] == 0) == direct


def test_register_tail_with_a_helmholtz_shift_and_odd_sweep_counts():
    """the shifted operator has no exact reciprocal diagonal (the kernel's DIV variant), V(1,3) and V(3,0) sweep counts"""
    n = 129
    f = O.sine_rhs(n, n)
    for pre, post, sigma in ((1, 3, 37.5), (3, 0, 0.0), (0, 2, 1e3)):
        out = {}
        for tail in (1, 2):
            e = mg.MultigridEngine(n, n, max_levels=6, cycle="W", smoother=_lib.MG_RBGS, omega=1.0, pre=pre, post=post, tail=tail,
                                   coarse_direct=False)
            e.set_shift(sigma)
            out[tail] = e.solve(f, tol=0.0, max_iterations=2)
            e.close()
        np.testing.assert_array_equal(out[1][0], out[2][0])
        assert out[1][1]["residual_history"] == out[2][1]["residual_history"]


def test_default_coarsest_solve_is_direct_and_the_iteration_can_be_asked_for():
    """coarse_direct="auto" (the shipped default): the nine unknowns of the 5 x 5 grid directly, in every cycle shape;
    False: the reference's iteration (a positive sweep count comes back)"""
    n = 129
    f = O.sine_rhs(n, n)
    for cyc in ("V", "W", "F"):
        for direct, expect_direct in (("auto", True), (True, True), (False, False)):
            e = mg.MultigridEngine(n, n, max_levels=6, cycle=cyc, smoother=_lib.MG_RBGS, omega=1.0, coarse_direct=direct)
            u, r = e.solve(f, tol=1e-10, max_iterations=30)
            assert (r["last_coarse_sweeps"] == 0) == expect_direct and r["converged"]
            e.close()


@pytest.mark.parametrize("nx,ny,dom", [(129, 65, (0.0, 2.0, 0.0, 1.0)), (65, 129, (0.0, 1.0, 0.0, 2.0)), (257, 129, (0.0, 1.0, 0.0, 1.0)), (49, 25, (0.0, 2.0, 0.0, 1.0))])
@pytest.mark.parametrize("cyc,kind", [("V", _lib.MG_JACOBI), ("W", _lib.MG_RBGS)])
def test_direct_coarsest_solve_on_grids_that_do_not_end_in_5x5(nx, ny, dom, cyc, kind):
    """A 2:1 grid ends in 9 x 5 (21 unknowns), 49 x 25 in 7 x 4 (10): the LDS tail solves any coarsest grid of at most 64 unknowns
    directly from its inverse in device memory; iterates stay within 1e-12 of the reference's iteration, histories within
    max(1e-9 relative, coarse_tolerance absolute) -- the bar of the 5 x 5 case"""
    x = np.linspace(dom[0], dom[1], nx); y = np.linspace(dom[2], dom[3], ny)
    f = np.sin(np.pi * x / dom[1])[:, None] * np.sin(np.pi * y / dom[3])[None, :] + 0.05 * np.random.default_rng(nx).standard_normal((nx, ny))
    res = {}
    for direct in (False, "auto"):
        e = mg.MultigridEngine(nx, ny, domain=dom, max_levels=mg.default_max_levels(nx, ny), cycle=cyc, smoother=kind,
                               omega=0.8 if kind == _lib.MG_JACOBI else 1.0, coarse_direct=direct)
        assert e.shapes[-1] != (5, 5) and (e.shapes[-1][0] - 2) * (e.shapes[-1][1] - 2) <= 64
        u, r = e.solve(f, tol=0.0, max_iterations=6)
        res[direct] = (u, r)
        e.close()
    assert res["auto"][1]["last_coarse_sweeps"] == 0 and res[False][1]["last_coarse_sweeps"] > 0
    assert np.max(np.abs(res["auto"][0] - res[False][0])) <= 1e-12 * np.max(np.abs(res[False][0]))
    np.testing.assert_allclose(res["auto"][1]["residual_history"], res[False][1]["residual_history"], rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------- ADVICE r02 ----
def test_fmg_under_defect_correction_is_used_not_discarded():
    """with MG_PREC_DEFECT the FMG guess was built in the fp32 hierarchy and thrown away.  Now the FMG pass runs
    on the error equation and its result is added to the fp64 iterate: the first residual is orders below ||f||."""
    n = 257
    f = O.sine_rhs(n, n)
    init = {}
    for fmg in (0, 1):
        e = mg.MultigridEngine(n, n, max_levels=7, precision=_lib.MG_PREC_DEFECT, fmg_cycles=fmg)
        u, r = e.solve(f, tol=1e-9, max_iterations=20)
        init[fmg] = (r["initial_residual"], r["iterations"], u)
        e.close()
    assert init[1][0] < 1e-2 * init[0][0] and init[1][1] < init[0][1]
    assert np.max(np.abs(init[1][2] - init[0][2])) < 1e-7
    # Dirichlet data survive the FMG start (it works on the error equation around them)
    e = mg.MultigridEngine(65, 65, max_levels=5, precision=_lib.MG_PREC_DEFECT)
    u0 = np.zeros((65, 65)); u0[0, :] = 1.0; u0[:, -1] = -2.0
    e.set_rhs(np.zeros((65, 65))); e.set_solution(u0); e.fmg(1)
    u = e.get_solution()
    np.testing.assert_array_equal(u[0, :], u0[0, :]); np.testing.assert_array_equal(u[:, -1], u0[:, -1])
    assert abs(u[32, 32]) > 1e-3                       # the harmonic extension reached the interior
    with pytest.raises(ValueError, match="constant-coefficient"):
        e.set_coefficient(np.ones((65, 65)))
    e.close()


def test_time_op_on_the_fine_level_clears_the_zero_iterate_flag():
    """mg_set_solution(NULL) -> mg_time_op (which rewrites the fine iterate) -> mg_residual_norm returned the
    cached ||f|| of the ZERO iterate."""
    n = 129
    e = mg.MultigridEngine(n, n, max_levels=6)
    f = O.sine_rhs(n, n)
    e.set_rhs(f)
    e.set_solution(None)
    n0 = e.residual_norm()
    e.time_op("sweeps2", 0, np.float64, reps=3)
    n1 = e.residual_norm()
    u = e.get_solution()
    g = mg.Grid(n, n)
    assert n1 < n0 and abs(n1 - O.l2_norm(O.residual(u, f, g.hx, g.hy, -1.0), g.hx, g.hy)) <= 1e-12 * n0
    e.close()


def test_plan_wait_deadline_returns_promptly_and_the_work_still_finishes():
    """MG_PLAN_TIMEOUT_S.  A plan whose queued work outlasts the deadline makes mg_plan_wait return
    MG_ERR_TIMEOUT (PlanTimeout) promptly; nothing is cancelled (the rank's driver then leaves the process)."""
    import torch
    from mixed_precision_multigrid_solvers_for_pdes_amd import dist_plan
    dev = torch.device("cuda", 0)
    a = torch.zeros((4096, 4096), dtype=torch.float64, device=dev)
    b = torch.ones((4096, 4096), dtype=torch.float64, device=dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    rec = dist_plan.PlanRecorder()
    for k in range(1500):                                  # ~1500 x 128 MB of copies: ~0.1 s of queued work, 20 x the deadline
        rec.copy2d(a if k % 2 == 0 else b, b if k % 2 == 0 else a)
    rec.add(acc, acc)
    rec.result(acc)
    plan = dist_plan.CyclePlan(rec, None, 0)
    st = torch.cuda.current_stream().cuda_stream
    old = os.environ.get("MG_PLAN_TIMEOUT_S")
    os.environ["MG_PLAN_TIMEOUT_S"] = "0.005"
    try:
        plan.run_async(st, st)
        t0 = time.time()
        with pytest.raises(_lib.PlanTimeout):
            plan.wait()
        assert time.time() - t0 < 1.0
    finally:
        if old is None:
            del os.environ["MG_PLAN_TIMEOUT_S"]
        else:
            os.environ["MG_PLAN_TIMEOUT_S"] = old
    torch.cuda.synchronize()                               # the queued work was not cancelled: it completes
    assert plan.wait() == 0.0                              # ... and the result can still be collected afterwards
    plan.close()


def test_eager_cycle_after_native_cycles_settles_the_queued_front_part():
    """cycle(0, zero_u=True) / a cycle through the Python driver after native ones used to leave the queued front
    part and the pending norm behind; the next native cycle then continued from stale arrays."""
    import torch
    n, px, py = 513, 2, 2
    dev = torch.device("cuda", 0)
    rhs = O.sine_rhs(n, n)
    cut = lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny]

    def run(interleave):
        s = D.DistributedMultigrid(n, n, px, py, range(px * py), D.HipOps(np.float64, dev), None, agglomerate_at=65)
        s.set_problem(cut)
        hist = []
        for k in range(5):
            if interleave and k == 3:
                s.native, was = False, s.native           # one cycle through the Python driver in the middle
                s.cycle(0)
                s.native = was
            else:
                s.cycle(0)
            hist.append(s.residual_norm())
        u = {r: s.local_solution(r)[1].copy() for r in s.ranks}
        nat = s.native_cycles
        s.close()
        return hist, u, nat
    h0, u0, nat0 = run(False)
    h1, u1, nat1 = run(True)
    assert nat0 >= 4 and nat1 >= 3
    assert h0 == h1
    for r in u0:
        np.testing.assert_array_equal(u0[r], u1[r])
