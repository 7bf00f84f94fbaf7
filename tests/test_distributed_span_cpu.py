"""The spanning scheme of the decomposed driver (DistributedMultigrid(span=True)): the level-0 up legs of cycle k and the down
legs of cycle k + 1 as one operation per block (ops.span_leg), the level-0 ghost exchange moved to the pre-smoothed iterate.
CPU: NumPy stand-in kernels (its span_leg is up leg + down leg), virtual ranks and gloo processes; owned cells must equal the
single-domain oracle bit for bit whatever mix of "queue the next front part" (mid) and "stop here" (back) a solve goes through."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mixed_precision_multigrid_solvers_for_pdes_amd import distributed as D      # noqa: E402
import dist_helpers as H                                                           # noqa: E402
from test_distributed_cpu import _oracle, _rhs, _u0, _plans_mirror_each_other      # noqa: E402


class SpanOps(H.NumpyOps):
    span_min_cells = 0                      # the library serves blocks above ~1100^2 cells; the stand-in every block


def _solver(NX, NY, px, py, ranks, dist, levels, cyc, omega, agg, span):
    rhs, u0 = _rhs(NX, NY, (0.0, 1.0, 0.0, 1.0)), _u0(NX, NY)
    s = D.DistributedMultigrid(NX, NY, px, py, ranks, SpanOps(), dist, max_levels=levels, cycle=cyc, smoother="jacobi", omega=omega,
                               agglomerate_at=agg, coarse_maxit=40, mode="fused", span=span)
    s.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    return s


@pytest.mark.parametrize("px,py,NX,NY,agg,cyc,omega,pattern", [
    (2, 1, 129, 65, 33, "V", 0.8, (True, True, True, True)),
    (2, 1, 129, 65, 33, "W", 0.8, (True, False, True, False)),
    (2, 2, 129, 129, 33, "V", 0.8, (False, True, True, False)),
    (2, 2, 129, 129, 33, "F", 2 / 3, (True, True, False)),
    (4, 2, 257, 129, 33, "V", 0.8, (True, True, True)),
    (2, 2, 129, 129, 65, "V", 0.8, (True, False, True, True)),        # one distributed level: the replicated engine right below
    (2, 2, 129, 129, 65, "W", 0.8, (True, True)),
])
def test_spanning_scheme_virtual_ranks_equal_single_domain(px, py, NX, NY, agg, cyc, omega, pattern):
    levels = len(D.hierarchy_shapes(NX, NY, 99))
    u_ref, h_ref = _oracle(NX, NY, levels, cyc, "jacobi", omega, len(pattern))
    s = _solver(NX, NY, px, py, range(px * py), None, levels, cyc, omega, agg, True)
    assert s._span_usable()
    hist = []
    for spec in pattern:
        s.speculate = spec
        s.cycle(0)
        hist.append(s.residual_norm())
    np.testing.assert_allclose(hist, h_ref, rtol=1e-13)
    s._settle()                              # a queued front part is dropped: u is the iterate
    np.testing.assert_array_equal(H.assemble(s, NX, NY), u_ref)
    if cyc == "V" and all(pattern):          # once: u; per part (the front, then one mid per cycle): the pre-smoothed level-0 iterate + one per distributed coarse rhs
        assert s.exchanges == 1 + (len(pattern) + 1) * (1 + (s.Ld - 1))


def test_new_problem_after_a_queued_front_part_and_two_solves():
    NX = NY = 129
    levels = len(D.hierarchy_shapes(NX, NY, 99))
    u_ref, h_ref = _oracle(NX, NY, levels, "V", "jacobi", 0.8, 3)
    s = _solver(NX, NY, 2, 2, range(4), None, levels, "V", 0.8, 33, True)
    for _ in range(2):
        s.cycle(0)
    rhs, u0 = _rhs(NX, NY, (0.0, 1.0, 0.0, 1.0)), _u0(NX, NY)
    s.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    hist = []
    for _ in range(3):
        s.cycle(0)
        hist.append(s.residual_norm())
    s._settle()
    np.testing.assert_array_equal(H.assemble(s, NX, NY), u_ref)
    np.testing.assert_allclose(hist, h_ref, rtol=1e-13)


def test_decomposed_solve_loop_with_the_spanning_scheme_stops_on_the_tolerance():
    """DecomposedSolve (what bench.py --gpus N and DistributedMultigridSolver.solve run): same history and iterate with and
    without the spanning scheme, including the cycle on which the tolerance is met"""
    NX = NY = 129
    levels = len(D.hierarchy_shapes(NX, NY, 99))
    out = {}
    for span in (False, True):
        s = _solver(NX, NY, 2, 2, range(4), None, levels, "V", 0.8, 33, span)
        solve = D.DecomposedSolve({"f64": s}, "fixed")
        rhs, u0 = _rhs(NX, NY, (0.0, 1.0, 0.0, 1.0)), _u0(NX, NY)
        solve.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
        probe, _, _ = solve.run(0.0, 5)
        solve.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
        hist, _, conv = solve.run(1.5 * probe[3], 10)
        s._settle()
        out[span] = (hist, conv, H.assemble(s, NX, NY))
    assert out[True][1] and out[False][1] and len(out[True][0]) == len(out[False][0]) <= 4
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-13)
    np.testing.assert_array_equal(out[True][2], out[False][2])


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, px, py, out_path):
    import torch.distributed as dist
    from mixed_precision_multigrid_solvers_for_pdes_amd import dist_plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    NX = NY = 129
    s = _solver(NX, NY, px, py, [rank], dist, 5, "V", 0.8, 33, True)
    hist = []
    for spec in (True, True, False):
        s.speculate = spec
        s.cycle(0)
        hist.append(s.residual_norm())
    b, u = s.local_solution(rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, (b.gx0, b.gy0, b.i_lo, b.i_hi, b.j_lo, b.j_hi, u))
    # what the plans of the spanning scheme hand to RCCL: front, mid (legs | exchange + lower levels), back -- recorded on every
    # rank; the communication groups must mirror each other rank by rank
    ok = True
    for part in (lambda: (s._sp_front("t"), s._sp_lower("t")), lambda: (s._sp_legs("t", "s"), s.allreduce_sum(s._last_norm_parts)),
                 lambda: s._sp_lower("s"), lambda: (s._sp_legs("s", None), s.allreduce_sum(s._last_norm_parts))):
        s._rec = dist_plan.PlanRecorder()
        part()
        sig = s._rec.signature()
        s._rec = None
        sigs = [None] * world
        dist.all_gather_object(sigs, sig)
        if rank == 0:
            ok = ok and _plans_mirror_each_other(sigs)
    if rank == 0:
        full = np.full((NX, NY), np.nan)
        for gx0, gy0, i_lo, i_hi, j_lo, j_hi, ul in gathered:
            full[gx0 + i_lo:gx0 + i_hi, gy0 + j_lo:gy0 + j_hi] = ul[i_lo:i_hi, j_lo:j_hi]
        np.savez(out_path, u=full, hist=np.array(hist), plan_ok=ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_multiprocess_spanning_scheme(tmp_path, world):
    import torch.multiprocessing as mp
    px, py = D.process_grid(world)
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), px, py, out), nprocs=world, join=True)
    res = np.load(out)
    u_ref, h_ref = _oracle(129, 129, 5, "V", "jacobi", 0.8, 3)
    np.testing.assert_array_equal(res["u"], u_ref)
    np.testing.assert_allclose(res["hist"], h_ref, rtol=1e-13)
    assert bool(res["plan_ok"])
