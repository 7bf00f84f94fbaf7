"""The decomposed engine behind the reference's solver API, on CPU: DistributedMultigridSolver.setup / solve
(gpu/multi_gpu.py:348, 540-607) over (a) virtual ranks in one process and (b) two real processes over torch.distributed /
gloo, with the NumPy stand-in kernels of tests/dist_helpers.py.  Scatter, the mg_iterate loop (tolerance, max_iterations,
precision policy), gather and the info dict are the code a GPU run executes; the answer must be the reference's own golden
solve (tests/golden/solves.npz, written by the reference's CPU MultigridSolver): history to 1e-9, iterate to 1e-12."""
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mixed_precision_multigrid_solvers_for_pdes_amd as mg                          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd import distributed as D          # noqa: E402
from mixed_precision_multigrid_solvers_for_pdes_amd.multi_gpu import process_grid_for  # noqa: E402
from oracle import mg_oracle as O                                                      # noqa: E402
import dist_helpers as H                                                               # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "solves.npz"))
INFO_KEYS = ("converged", "iterations", "final_residual", "convergence_rate", "residual_history", "total_time",
             "average_time_per_iteration", "precision_levels_used", "cycle_type", "num_levels", "grid_hierarchy",
             "level_timings", "pre_smooth_iterations", "post_smooth_iterations", "initial_residual", "n_gpus", "process_grid",
             "exchanges_per_cycle", "total_iterations", "average_iterations", "num_devices", "distributed_solve_time",
             "decomposition_strategy", "device_stats")


def _factory(dtype, managed_single=False, mixed=False):
    return H.NumpyOps(dtype, mixed=mixed)


def _solver(world_ids, strategy, n, levels, cyc, smoother, agg, maxit=30, tol=1e-10):
    return mg.DistributedMultigridSolver(device_ids=world_ids, decomposition_strategy=strategy, agglomerate_at=agg,
                                         ops_factory=_factory, max_levels=levels, max_iterations=maxit, tolerance=tol,
                                         cycle_type=cyc, smoother=smoother)


def _plugins():
    return mg.LaplacianOperator(coefficient=-1.0), mg.RestrictionOperator("full_weighting"), mg.ProlongationOperator("bilinear")


def test_decomposition_strategies():
    assert process_grid_for("stripe", 4) == (4, 1) and process_grid_for("checkerboard", 4) == (2, 2)
    assert process_grid_for("block", 8) == (4, 2) and process_grid_for(mg.DecompositionType.STRIP_Y, 2) == (1, 2)
    with pytest.raises(ValueError):
        process_grid_for("checkerboard", 8)                       # gpu/multi_gpu.py:434-436
    with pytest.raises(ValueError):
        process_grid_for("diagonal", 2)


@pytest.mark.parametrize("ids,strategy,key,n,levels,cyc,smoother", [
    ([0, 0], "stripe", "n65_L5_V_jacobi08_float64", 65, 5, "V", "jacobi"),
    ([0, 0, 0, 0], "checkerboard", "n65_L5_W_jacobi08_float64", 65, 5, "W", "jacobi"),
    ([0, 0, 0, 0], "block", "n129_L6_W_rbgs_float64", 129, 6, "W", "gauss_seidel"),
])
def test_virtual_ranks_solve_equals_reference_golden(ids, strategy, key, n, levels, cyc, smoother):
    op, R, P = _plugins()
    grid = mg.Grid(n, n)
    s = _solver(ids, strategy, n, levels, cyc, smoother, agg=17)
    s.setup(grid, op, R, P)
    rhs = O.sine_rhs(n, n)
    keep = rhs.copy()
    u, info = s.solve(grid, op, rhs)
    np.testing.assert_array_equal(rhs, keep)                       # the caller's array is not touched
    ref_hist, ref_u = GOLD[key + "__hist"], GOLD[key + "__u"]
    assert info["iterations"] == len(ref_hist) and info["converged"]
    np.testing.assert_allclose(info["residual_history"], ref_hist, rtol=1e-9, atol=5e-14)
    assert float(np.max(np.abs(u - ref_u)) / np.max(np.abs(ref_u))) <= 1e-12
    for k in INFO_KEYS:
        assert k in info, k
    assert info["n_gpus"] == len(ids) and info["process_grid"] == process_grid_for(strategy, len(ids)) and info["decomposed"]
    assert info["num_levels"] == levels and info["grid_hierarchy"][0] == (n, n) and info["grid_hierarchy"][-1] == (5, 5)
    assert info["exchanges_per_cycle"] > 0 and info["distributed_levels"] >= 1
    assert set(s.communication_graph) == set(range(len(ids))) and all(s.communication_graph.values())
    # a second solve on the same hierarchy, from an initial guess (Dirichlet data + the converged interior): converges at once
    u2, info2 = s.solve(grid, op, rhs, initial_guess=u)
    assert info2["iterations"] <= 2 and info2["converged"]
    s.cleanup()


def test_solve_before_setup_and_bad_plugins():
    op, R, P = _plugins()
    s = _solver([0, 0], "stripe", 65, 5, "V", "jacobi", agg=17)
    with pytest.raises(ValueError, match="not properly setup"):
        s.solve(mg.Grid(65, 65), op, np.zeros((65, 65)))
    with pytest.raises(NotImplementedError):
        s.setup(mg.Grid(65, 65), op, mg.RestrictionOperator("injection"), P)
    with pytest.raises(ValueError, match="Unknown smoother"):
        mg.DistributedMultigridSolver(device_ids=[0, 0], ops_factory=_factory, smoother="ilu")


def test_adaptive_policy_through_the_api_equals_the_oracle_trajectory():
    """precision_manager = the one-way adaptive rule: the decomposed loop must take the single-domain trajectory -- start in
    double, drop to single on the large first residual, promote at ||r|| < 10 thr -- and end on the same iterate."""
    n, levels, thr = 129, 6, 1e-3
    op, R, P = _plugins()
    grid = mg.Grid(n, n)
    s = _solver([0, 0, 0, 0], "checkerboard", n, levels, "V", "jacobi", agg=33, maxit=12, tol=1e-9)
    s.setup(grid, op, R, P)
    pm = mg.PrecisionManager(default_precision="double", adaptive=True, convergence_threshold=thr)
    pm.reference_rule = False
    rhs = O.sine_rhs(n, n)
    u, info = s.solve(grid, op, rhs, precision_manager=pm)
    # single domain, same policy by hand: fp32 cycles until ||r|| < 10 thr, then fp64 (the NumPy stand-in's fp32 solver is the
    # all-fp32 hierarchy of Grid(dtype=float32); the device kernels keep the coarsest level and the interpolation in fp64)
    o32 = O.MGOracle(n, n, (0.0, 1.0, 0.0, 1.0), np.float32, -1.0, levels, "V", 2, 2, "jacobi", 0.8, "vectorized")
    o64 = O.MGOracle(n, n, (0.0, 1.0, 0.0, 1.0), np.float64, -1.0, levels, "V", 2, 2, "jacobi", 0.8, "vectorized")
    pol = D.AdaptivePolicy(thr)
    o64.rhs[0] = rhs.copy()
    v = np.zeros((n, n))
    rn = o64.residual_norm(v, rhs, 0)
    hist, phases = [], []
    for _ in range(12):
        ph = pol.before_cycle(rn)
        if ph == "f32":
            o32.rhs[0] = rhs.astype(np.float32)
            v = o32.cycle_once(v.astype(np.float32), 0)
            rn = o32.residual_norm(v, rhs.astype(np.float32), 0)
        else:
            o64.rhs[0] = rhs.copy()
            v = o64.cycle_once(v.astype(np.float64), 0)
            rn = o64.residual_norm(v, rhs, 0)
        pol.after_cycle(rn)
        if pol.floor_due():                      # the fp32 residual floor, once, from the iterate of the first fp32 cycle
            h = 1.0 / (n - 1)
            pol.set_floor(4.0 / (h * h), float(np.sqrt(h * h * np.sum(v.astype(np.float64)**2))))
        hist.append(rn); phases.append(ph)
        if rn < 1e-9:
            break
    assert phases[0] == "f32" and phases[-1] == "f64" and info["precision_switches"] == 2 and pol.reason in ("threshold", "fp32_floor")
    assert info["iterations"] == len(hist)
    np.testing.assert_allclose(info["residual_history"], hist, rtol=2e-5)        # fp32 norms: fp32 vs fp64 accumulation
    assert float(np.max(np.abs(u - v)) / np.max(np.abs(v))) <= 1e-12
    assert set(info["precision_levels_used"]) == {"float32", "float64"}
    assert pm.current_precision == mg.PrecisionLevel.DOUBLE
    s.cleanup()


# ---------------------------------------------------------------------------------------- gloo, world 2 ----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _api_worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, levels = 65, 5
    op, R, P = _plugins()
    grid = mg.Grid(n, n)
    s = mg.DistributedMultigridSolver(decomposition_strategy="stripe", agglomerate_at=17, ops_factory=_factory, max_levels=levels,
                                      max_iterations=30, tolerance=1e-10, cycle_type="V", smoother="jacobi")
    assert s.world == world and s.ranks == [rank]
    s.setup(grid, op, R, P)
    u, info = s.solve(grid, op, O.sine_rhs(n, n))
    # the facade switch: MixedPrecisionMultigrid picks the decomposed solver up from the process group
    if rank == 0:
        np.savez(out_path, u=u, hist=np.array(info["residual_history"]), n_gpus=info["n_gpus"], grid=np.array(info["process_grid"]),
                 its=info["iterations"], exch=info["exchanges_per_cycle"])
    s.cleanup()
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_solve_through_the_api_equals_reference_golden(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "api.npz")
    mp.spawn(_api_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = np.load(out)
    key = "n65_L5_V_jacobi08_float64"
    np.testing.assert_allclose(res["hist"], GOLD[key + "__hist"], rtol=1e-9, atol=5e-14)
    ref_u = GOLD[key + "__u"]
    assert float(np.max(np.abs(res["u"] - ref_u)) / np.max(np.abs(ref_u))) <= 1e-12
    assert int(res["n_gpus"]) == 2 and tuple(res["grid"]) == (2, 1) and int(res["its"]) == len(GOLD[key + "__hist"])
    # fused legs: the fine iterate + one distributed coarse rhs per cycle (+ one exchange for the initial residual norm)
    assert abs(float(res["exch"]) * int(res["its"]) - (2 * int(res["its"]) + 1)) < 1e-9


# ---------------------------------------------------------------------------------------- launcher supervision ----
def test_supervise_takes_the_other_ranks_down_when_one_fails():
    """bench.py's launcher blocked on rank 0 while a sibling had died.  supervise() polls every child and
    terminates the rest as soon as one exits non-zero."""
    sys.path.insert(0, ROOT)
    import bench
    env = dict(os.environ)
    hang = ([sys.executable, "-c", "import time, sys; print('{\"x\": 1}', flush=True); time.sleep(120)"], env)
    die = ([sys.executable, "-c", "import sys, time; time.sleep(0.5); sys.exit(3)"], env)
    t0 = time.time()
    codes, out0 = bench.supervise([hang, die], grace_s=3.0)
    assert time.time() - t0 < 20.0
    assert codes[1] == 3 and codes[0] != 0 and '{"x": 1}' in out0
    ok = ([sys.executable, "-c", "print('{\"y\": 2}')"], env)
    codes, out0 = bench.supervise([ok, ok])
    assert codes == [0, 0] and '{"y": 2}' in out0
