"""The N > 1 path on CPU: (a) every rank of a px x py decomposition executed in ONE process (virtual ranks,
in-process halo copies) and (b) real multi-process runs over torch.distributed/gloo with world_size 2 and 4.
Kernels are replaced by the NumPy stand-in of tests/dist_helpers.py; the result must equal the single-domain
oracle BIT FOR BIT (Jacobi and globally coloured RBGS are decomposition-invariant), the norm to round-off."""
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
from oracle import mg_oracle as O                                                  # noqa: E402
import dist_helpers as H                                                           # noqa: E402


def test_block_bookkeeping():
    assert D.process_grid(1) == (1, 1) and D.process_grid(2) == (2, 1) and D.process_grid(4) == (2, 2) and D.process_grid(8) == (4, 2)
    b0, b1 = D.Block(8193, 4097, 2, 1, 0, 0), D.Block(8193, 4097, 2, 1, 1, 0)
    assert (b0.gx0, b0.lnx, b0.lny, b0.sides) == (0, 4098, 4097, 1 | 4 | 8)
    assert (b1.gx0, b1.lnx, b1.lny, b1.sides) == (4096, 4097, 4097, 2 | 4 | 8)
    assert (b0.i_lo, b0.i_hi, b1.i_lo, b1.i_hi) == (0, 4097, 1, 4097)        # rows 0..4096 | 4097..8192: disjoint cover
    # exclusive windows tile the grid exactly once
    cover = np.zeros((65, 33), dtype=int)
    for r in range(8):
        b = D.Block(65, 33, 4, 2, *divmod(r, 2))
        cover[b.gx0 + b.i_lo:b.gx0 + b.i_hi, b.gy0 + b.j_lo:b.gy0 + b.j_hi] += 1
        assert b.gx0 % 2 == 0 and b.gy0 % 2 == 0
    assert np.all(cover == 1)
    shapes = D.hierarchy_shapes(16385, 8193, 13)
    assert shapes[-1] == (9, 5) and D.distributed_levels(shapes, 4, 2, 1025) == 4       # 16385, 8193, 4097, 2049 stay distributed
    shapes = D.hierarchy_shapes(8193, 8193, 12)
    assert D.distributed_levels(shapes, 2, 2, 1025) == 3
    assert D.distributed_levels(D.hierarchy_shapes(65, 65, 5), 2, 2, 17) == 2              # 65, 33 distributed; 17 replicated
    assert D.distributed_levels(D.hierarchy_shapes(65, 65, 5), 2, 2, 9) == 3


def _oracle(NX, NY, levels, cyc, kind, omega, ncycles, domain=(0.0, 1.0, 0.0, 1.0)):
    mg = O.MGOracle(NX, NY, domain, np.float64, -1.0, levels, cyc, 2, 2, kind, omega, "vectorized", coarse_maxit=40)
    rhs = _rhs(NX, NY, domain)
    mg.rhs[0] = rhs.copy()
    u = _u0(NX, NY)
    hist = []
    for _ in range(ncycles):
        u = mg.cycle_once(u, 0)
        hist.append(mg.residual_norm(u, rhs, 0))
    return u, hist


def _rhs(NX, NY, domain):
    rng = np.random.default_rng(NX * 31 + NY)
    return O.sine_rhs(NX, NY, domain) + 0.1 * rng.standard_normal((NX, NY))       # non-zero boundary values too


def _u0(NX, NY):
    rng = np.random.default_rng(NX + 7 * NY)
    return rng.standard_normal((NX, NY))                                           # non-zero Dirichlet data + guess


def _run_ranks(NX, NY, px, py, ranks, dist, levels, cyc, kind, omega, ncycles, agg, domain=(0.0, 1.0, 0.0, 1.0), mode="per_operator"):
    rhs, u0 = _rhs(NX, NY, domain), _u0(NX, NY)
    s = D.DistributedMultigrid(NX, NY, px, py, ranks, H.NumpyOps(), dist, domain=domain, max_levels=levels, cycle=cyc,
                               smoother=kind, omega=omega, agglomerate_at=agg, coarse_maxit=40, mode=mode)   # rhs has boundary noise: the
    # coarsest solve never meets its tolerance and would burn 1000 sweeps per visit
    s.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny],
                  lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    hist = []
    for _ in range(ncycles):
        s.cycle(0)
        hist.append(s.residual_norm())
    return s, hist


@pytest.mark.parametrize("px,py", [(2, 1), (1, 2), (2, 2), (4, 2)])
@pytest.mark.parametrize("cyc,kind,omega", [("V", "jacobi", 0.8), ("W", "rbgs", 1.0), ("V", "rbgs", 1.15), ("F", "jacobi", 2 / 3)])
def test_virtual_ranks_equal_single_domain(px, py, cyc, kind, omega):
    NX, NY, levels = 65, 33 if py == 1 else 65, 4
    u_ref, h_ref = _oracle(NX, NY, levels, cyc, kind, omega, 3)
    s, hist = _run_ranks(NX, NY, px, py, range(px * py), None, levels, cyc, kind, omega, 3, agg=17)
    assert s.Ld == 2 and s.L == levels
    np.testing.assert_array_equal(H.assemble(s, NX, NY), u_ref)
    np.testing.assert_allclose(hist, h_ref, rtol=1e-13)


@pytest.mark.parametrize("px,py,NX,NY,agg", [(2, 1, 129, 65, 33), (1, 2, 65, 129, 33), (2, 2, 129, 129, 33), (4, 2, 257, 129, 33), (2, 2, 129, 129, 65)])
@pytest.mark.parametrize("cyc,omega", [("V", 0.8), ("W", 0.8), ("F", 2 / 3)])
def test_fused_mode_virtual_ranks_equal_single_domain(px, py, NX, NY, agg, cyc, omega):
    """Communication-avoiding mode: ghost zones of 7 cells, fused legs, one exchange of the iterate per cycle and one
    of each coarse rhs.  Owned cells must still equal the single-domain oracle bit for bit, and the norm the up leg
    accumulates must be the global residual norm."""
    levels = D.hierarchy_shapes(NX, NY, 99).__len__()
    u_ref, h_ref = _oracle(NX, NY, levels, cyc, "jacobi", omega, 3)
    s, hist = _run_ranks(NX, NY, px, py, range(px * py), None, levels, cyc, "jacobi", omega, 3, agg=agg, mode="fused")
    assert s.mode == "fused" and s.G == 7 and s.Ld >= 1
    np.testing.assert_array_equal(H.assemble(s, NX, NY), u_ref)
    np.testing.assert_allclose(hist, h_ref, rtol=1e-13)
    if cyc == "V":
        assert s.exchanges == 3 * (1 + (s.Ld - 1))          # per cycle: the fine iterate + one per distributed coarse rhs


@pytest.mark.parametrize("px,py,NX,NY,agg", [(2, 1, 257, 129, 65), (2, 2, 257, 257, 65), (1, 2, 129, 257, 33)])
@pytest.mark.parametrize("cyc,omega", [("V", 1.0), ("W", 1.15)])
def test_fused_mode_red_black_virtual_ranks_equal_single_domain(px, py, NX, NY, agg, cyc, omega):
    """Red-black GS in the communication-avoiding mode: a colour pass costs one ghost cell, G = 13."""
    levels = len(D.hierarchy_shapes(NX, NY, 99))
    u_ref, h_ref = _oracle(NX, NY, levels, cyc, "rbgs", omega, 3)
    s, hist = _run_ranks(NX, NY, px, py, range(px * py), None, levels, cyc, "rbgs", omega, 3, agg=agg, mode="fused")
    assert s.mode == "fused" and s.G == 13 and s.Ld >= 1
    np.testing.assert_array_equal(H.assemble(s, NX, NY), u_ref)
    np.testing.assert_allclose(hist, h_ref, rtol=1e-13)


def test_fused_mode_block_bookkeeping():
    b0, b1 = D.Block(8193, 4097, 2, 1, 0, 0, 7), D.Block(8193, 4097, 2, 1, 1, 0, 7)
    assert (b0.gx0, b0.lnx, b0.oi_lo, b0.oi_hi) == (0, 4097 + 7, 1, 4096)          # rows 0 .. 4096+7
    assert (b1.gx0, b1.lnx, b1.oi_lo, b1.oi_hi) == (4096 - 6, 4097 + 6, 7, 7 + 4094)   # rows 4090 .. 8192
    c1 = D.Block(4097, 2049, 2, 1, 1, 0, 7)
    assert b1.coarse_offsets(c1) == (3, 0) and b0.coarse_offsets(D.Block(4097, 2049, 2, 1, 0, 0, 7)) == (0, 0)
    assert b1.gx0 % 2 == 0 and c1.gx0 % 2 == 0
    assert D.distributed_levels(D.hierarchy_shapes(8193, 8193, 12), 2, 2, 1025, 7) == 3


def test_virtual_ranks_rectangular_cells_and_single_distributed_level():
    dom = (0.0, 2.0, 0.0, 1.0)
    u_ref, h_ref = _oracle(129, 33, 4, "V", "jacobi", 0.8, 2, dom)
    s, hist = _run_ranks(129, 33, 2, 1, range(2), None, 4, "V", "jacobi", 0.8, 2, agg=65, domain=dom)
    assert s.Ld == 1
    np.testing.assert_array_equal(H.assemble(s, 129, 33), u_ref)
    np.testing.assert_allclose(hist, h_ref, rtol=1e-13)


# ---------------------------------------------------------------------------------------- gloo ----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _plans_mirror_each_other(sigs):
    """sigs[p] = PlanRecorder.signature() of rank p.  Same number and kinds of communication steps on every rank, equal
    collective sizes, and inside p2p group k every send p -> q (n bytes) has exactly one receive on q from p of n bytes."""
    world = len(sigs)
    if len({len(s) for s in sigs}) != 1 or len(sigs[0]) == 0:
        return False
    for k in range(len(sigs[0])):
        kinds = {s[k][0] for s in sigs}
        if len(kinds) != 1:
            return False
        if kinds != {"p2p"}:
            if len({s[k][1] for s in sigs}) != 1:
                return False
            continue
        sends = sorted((p, peer, n) for p in range(world) for kind, peer, n in sigs[p][k][1] if kind == "send")
        recvs = sorted((peer, p, n) for p in range(world) for kind, peer, n in sigs[p][k][1] if kind == "recv")
        if sends != recvs or not sends or len(set((a, b) for a, b, _ in sends)) != len(sends):
            return False
    return True


def test_plan_signature_checker_rejects_mismatches():
    ok = [[("p2p", (("send", 1, 8), ("recv", 1, 8))), ("allreduce", 1)], [("p2p", (("send", 0, 8), ("recv", 0, 8))), ("allreduce", 1)]]
    assert _plans_mirror_each_other(ok)
    bad_size = [ok[0], [("p2p", (("send", 0, 8), ("recv", 0, 16))), ("allreduce", 1)]]
    missing = [ok[0], [("p2p", (("send", 0, 8),)), ("allreduce", 1)]]
    shifted = [ok[0], [("allreduce", 1), ("p2p", (("send", 0, 8), ("recv", 0, 8)))]]
    assert not _plans_mirror_each_other(bad_size) and not _plans_mirror_each_other(missing) and not _plans_mirror_each_other(shifted)


def _worker(rank, world, port, px, py, cyc, kind, omega, out_path, mode="per_operator"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    NX = NY = 65 if mode == "per_operator" else 129
    s, hist = _run_ranks(NX, NY, px, py, [rank], dist, 4 if mode == "per_operator" else 5, cyc, kind, omega, 3,
                         agg=17 if mode == "per_operator" else 33, mode=mode)
    b, u = s.local_solution(rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, (b.gx0, b.gy0, b.i_lo, b.i_hi, b.j_lo, b.j_hi, u))
    plan_ok = True
    if s.mode == "fused":
        # what a native cycle plan of this rank would hand to RCCL (dist_plan.PlanRecorder): one more cycle with the
        # recorder attached; every rank's communication groups must mirror its peers', group by group
        from mixed_precision_multigrid_solvers_for_pdes_amd import dist_plan
        s._rec = dist_plan.PlanRecorder()
        s._last_norm_parts = None
        s._cycle_fused(0, False)
        s.allreduce_sum(s._last_norm_parts)
        sig = s._rec.signature()
        s._rec = None
        sigs = [None] * world
        dist.all_gather_object(sigs, sig)
        if rank == 0:
            plan_ok = _plans_mirror_each_other(sigs)
    if rank == 0:
        full = np.full((NX, NY), np.nan)
        for gx0, gy0, i_lo, i_hi, j_lo, j_hi, ul in gathered:
            full[gx0 + i_lo:gx0 + i_hi, gy0 + j_lo:gy0 + j_hi] = ul[i_lo:i_hi, j_lo:j_hi]
        np.savez(out_path, u=full, hist=np.array(hist), plan_ok=plan_ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cyc,kind,omega", [(2, "V", "jacobi", 0.8), (2, "W", "rbgs", 1.0), (4, "V", "rbgs", 1.0), (4, "W", "jacobi", 0.8),
                                                  (8, "V", "jacobi", 0.8)])
def test_gloo_multiprocess_equals_single_domain(tmp_path, world, cyc, kind, omega):
    import torch.multiprocessing as mp
    px, py = D.process_grid(world)
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), px, py, cyc, kind, omega, out), nprocs=world, join=True)
    res = np.load(out)
    u_ref, h_ref = _oracle(65, 65, 4, cyc, kind, omega, 3)
    np.testing.assert_array_equal(res["u"], u_ref)
    np.testing.assert_allclose(res["hist"], h_ref, rtol=1e-13)


@pytest.mark.parametrize("world,cyc", [(2, "V"), (4, "V"), (4, "W"), (8, "V")])       # 8: the 4 x 2 grid of the 8-GPU node
def test_gloo_multiprocess_fused_mode(tmp_path, world, cyc):
    import torch.multiprocessing as mp
    px, py = D.process_grid(world)
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), px, py, cyc, "jacobi", 0.8, out, "fused"), nprocs=world, join=True)
    res = np.load(out)
    u_ref, h_ref = _oracle(129, 129, 5, cyc, "jacobi", 0.8, 3)
    np.testing.assert_array_equal(res["u"], u_ref)
    np.testing.assert_allclose(res["hist"], h_ref, rtol=1e-13)
    assert bool(res["plan_ok"])            # the RCCL steps a native plan would replay pair up across the ranks


def test_precision_switch_between_two_decomposed_solvers():
    """bench.py --gpus N runs the adaptive policy on two solvers that share the decomposition (fp32 and fp64):
    take_iterate_from must hand over the whole local iterate, ghost zone included, so that the continued cycles
    equal the single-domain oracle that casts its iterate at the same point."""
    NX = NY = 129
    dom = (0.0, 1.0, 0.0, 1.0)
    levels = len(D.hierarchy_shapes(NX, NY, 99))
    rhs64 = O.sine_rhs(NX, NY, dom)
    # single domain: 2 cycles in fp32, cast, 2 cycles in fp64
    o32 = O.MGOracle(NX, NY, dom, np.float32, -1.0, levels, "V", 2, 2, "jacobi", 0.8, "vectorized")
    o64 = O.MGOracle(NX, NY, dom, np.float64, -1.0, levels, "V", 2, 2, "jacobi", 0.8, "vectorized")
    o32.rhs[0] = rhs64.astype(np.float32); o64.rhs[0] = rhs64.copy()
    u = np.zeros((NX, NY), dtype=np.float32)
    for _ in range(2):
        u = o32.cycle_once(u, 0)
    u = u.astype(np.float64)
    h_ref = []
    for _ in range(2):
        u = o64.cycle_once(u, 0)
        h_ref.append(o64.residual_norm(u, rhs64, 0))
    # decomposed, 2 x 2 virtual ranks, fused mode
    sol = {}
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        sol[name] = D.DistributedMultigrid(NX, NY, 2, 2, range(4), H.NumpyOps(dt), None, domain=dom, max_levels=levels, cycle="V",
                                           smoother="jacobi", omega=0.8, agglomerate_at=33, mode="fused")
        sol[name].set_problem(lambda b: rhs64[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    for _ in range(2):
        sol["f32"].cycle(0)
    sol["f64"].take_iterate_from(sol["f32"])
    hist = []
    for _ in range(2):
        sol["f64"].cycle(0)
        hist.append(sol["f64"].residual_norm())
    np.testing.assert_array_equal(H.assemble(sol["f64"], NX, NY), u)
    np.testing.assert_allclose(hist, h_ref, rtol=1e-12)


def test_adaptive_policy_host_port():
    """distributed.AdaptivePolicy = csrc/mghip.hip adapt() (one-way rule): the trajectory of the 4097^2 bench problem
    (fp32 stagnates at its residual floor, the stagnation rule promotes after five cycles) and the threshold path."""
    p = D.AdaptivePolicy(1e-6)
    assert p.before_cycle(9.87) == "f32"                       # large first residual: single
    p.set_floor(1.0, 1e-3 / D.AdaptivePolicy.EPS32)            # a floor estimate far below the trajectory: the other rules decide
    floor = [2.15, 2.05, 2.04, 2.045, 2.05]
    for rn in floor:
        p.after_cycle(rn)
        if rn is not floor[-1]:
            assert p.before_cycle(rn) == "f32"
    assert p.before_cycle(floor[-1]) == "f64" and p.promoted    # mean ratio of the last five > 0.9
    for rn in (0.2, 9.9, 1e-9):
        p.after_cycle(rn)
        assert p.before_cycle(rn) == "f64"                      # promoted for good
    q = D.AdaptivePolicy(1e-3)
    assert q.before_cycle(5.0) == "f32"
    q.after_cycle(5e-3)
    assert q.before_cycle(5e-3) == "f64"                        # below 10 thr
    r = D.AdaptivePolicy(1e-6)
    assert r.before_cycle(5e-5) == "f64"                        # small first residual: stay in double
    assert not D.stagnating([1.0, 0.1, 0.01, 0.001, 0.0001]) and D.stagnating([1.0, 1.0, 1.0, 1.0, 1.0])


def test_adaptive_policy_predicts_its_own_switch():
    """switch_likely(): asked BEFORE a cycle whether the norm that cycle will produce changes the precision (the decomposed
    driver then does not queue the next cycle's front part behind it).  It fires for the first fp32 cycle (the fp32 residual
    floor is evaluated from the iterate that cycle leaves and usually ends the phase), for the cycle that takes the norm
    below 10 thr, for the one whose norm fills the stagnation window -- never in double and never after the promotion."""
    def run(thr, norms, floor):
        p, fired, rn = D.AdaptivePolicy(thr), [], norms[0]
        for k, nxt in enumerate(norms[1:]):
            p.before_cycle(rn)
            fired.append(p.switch_likely())
            rn = nxt
            p.after_cycle(rn)
            if p.floor_due():
                p.set_floor(1.0, floor / D.AdaptivePolicy.EPS32)
        return p, fired
    # the 4097^2 bench problem: eps32 * diag * ||u|| = 2.0, the first fp32 cycle lands on 2.15: promoted at once
    p, fired = run(1e-6, [13.96, 2.150, 8.5e-2, 1.0e-2, 1.4e-3], floor=2.0)
    assert fired == [True, False, False, False] and p.promoted and p.reason == "fp32_floor" and p.phase == "f64"
    # the same trajectory with a floor estimate that is far too low: the five-norm stagnation rule still catches it
    p, fired = run(1e-6, [13.96, 2.150, 1.808, 1.803, 1.806, 1.809, 8.5e-2, 1.0e-2, 1.4e-3], floor=1e-3)
    assert fired == [True, False, False, False, True, False, False, False] and p.promoted and p.reason == "stagnation"
    # healthy contraction by 0.1 per cycle towards 10 thr = 1e-2, floor far below
    p, fired = run(1e-3, [50.0, 5.0, 0.5, 0.05, 0.005, 5e-4, 5e-5], floor=1e-6)
    assert fired == [True, False, False, True, False, False] and p.promoted and p.phase == "f64" and p.reason == "threshold"


def test_bench_py_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (how the driver's scaling run may start it): the parent spawns the
    two ranks, waits and relays rank 0's ONE JSON line.  Rehearsed on CPU over gloo with the NumPy stand-in kernels
    (MG_BENCH_OPS test hook); the decomposition, policy, timing and reporting code is the one the GPU run executes."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MG_DIST_BACKEND="gloo", MG_BENCH_OPS="dist_helpers:NumpyOps", OMP_NUM_THREADS="1",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, env.get("PYTHONPATH", "")]))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--grid-n", "65", "--agglomerate-at", "17"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "iterations", "residual_floor", "iterations_to_floor"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dd2x1"
    assert d["config"]["grid"] == [129, 65]
    assert abs(d["value"] - 129 * 65 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) < 1e-6 * d["value"]
    assert 0.0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["unit"] == "GB/s"
    # the solve itself: starts in double, drops to single on the large first residual (AdaptivePolicy), contracts
    assert d["cycles_fp32"] >= 1 and d["residual_first"] < 0.2 * d["residual_initial"]
    # the run diagnoses itself: who took part, which driver ran the cycles, where a cycle's time went
    assert d["ranks_seen"] == 2 and sorted(w["rank"] for w in d["rank_devices"]) == [0, 1]
    drv = d["driver"]
    assert drv["native_plan_cycles"] == 0 and drv["python_cycles"] == 3 and drv["fallback"] is None      # gloo: no RCCL plans
    assert drv["rccl_multi_rank_replay"].startswith("not exercised")
    assert drv["spanning_scheme"] and not any(drv["spanning_scheme"].values())     # follows the native plans: off in this rehearsal
    assert all(c["ran"] is False for c in drv["selfcheck"].values())               # nothing to replay on this backend
    ph = d["phases_ms_per_cycle"]
    for k in ("legs", "halo_copy", "halo_exchange", "coarse_allgather", "replicated_engine", "allreduce"):
        assert k in ph and ph[k] >= 0.0, k
    assert ph["legs"] > 0.0 and ph["halo_exchange"] > 0.0 and ph["replicated_engine"] > 0.0 and ph["cycles"] == 3
    # a rank that fails must fail the launcher
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--grid-n", "64", "--agglomerate-at", "17"], env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


# ---------------------------------------------------------------- variable coefficient, per-level mixed precision ----
def _a_at(NX, NY):
    """a(x, y) = 1 + 0.5 sin(2 pi x) cos(2 pi y) at global fine-grid indices."""
    x, y = np.linspace(0.0, 1.0, NX), np.linspace(0.0, 1.0, NY)
    return lambda ix, iy: 1.0 + 0.5 * np.sin(2 * np.pi * x[ix])[:, None] * np.cos(2 * np.pi * y[iy])[None, :]


def _var_oracle(NX, NY, levels, cyc, kind, omega, ncycles, mixed):
    a = _a_at(NX, NY)(np.arange(NX), np.arange(NY))
    mgo = O.VarMGOracle(a, (0.0, 1.0, 0.0, 1.0), np.float64, -1.0, levels, cyc, 2, 2, kind, omega, "vectorized", coarse_maxit=40)
    pm = O.OraclePrecision("mixed") if mixed else None
    rhs = _rhs(NX, NY, (0.0, 1.0, 0.0, 1.0))
    mgo.rhs[0] = rhs.copy()
    u = _u0(NX, NY)
    hist = []
    for _ in range(ncycles):
        u = mgo.cycle_once(u, 0, pm)
        mgo.rhs[0] = rhs.copy()                 # the mixed cycle converts rhs[0] in place on entry; same values, fp64 here
        hist.append(mgo.residual_norm(u.astype(np.float64), rhs, 0))
    return u, hist


@pytest.mark.parametrize("px,py,NX,NY,agg,cyc,kind,omega,mixed", [
    (2, 1, 129, 65, 33, "V", "jacobi", 0.8, False), (2, 2, 129, 129, 33, "W", "jacobi", 0.8, False),
    (2, 2, 257, 257, 65, "W", "rbgs", 1.0, False), (2, 1, 257, 129, 65, "V", "rbgs", 1.15, False),
    # per-level mixed: 257^2 has 7 levels, split at level 3 (33^2); agg = 17 keeps 257..33 decomposed: an fp32 DISTRIBUTED level
    (2, 2, 257, 257, 17, "V", "jacobi", 0.8, True), (2, 2, 257, 257, 65, "W", "jacobi", 0.8, True), (4, 2, 513, 257, 17, "V", "jacobi", 0.8, True)])
def test_variable_coefficient_and_mixed_levels_virtual_ranks(px, py, NX, NY, agg, cyc, kind, omega, mixed):
    """BASELINE config 5's ingredients in the decomposed driver: -div(a grad u) with a coefficient field per block
    (ghost zone filled from the function, no exchange) and per-level mixed precision on decomposed AND replicated
    levels.  Owned cells equal the single-domain oracle (VarMGOracle, our own restatement: parity unpinned) bit for bit."""
    levels = len(D.hierarchy_shapes(NX, NY, 99))
    u_ref, h_ref = _var_oracle(NX, NY, levels, cyc, kind, omega, 2, mixed)
    rhs, u0 = _rhs(NX, NY, (0.0, 1.0, 0.0, 1.0)), _u0(NX, NY)
    s = D.DistributedMultigrid(NX, NY, px, py, range(px * py), H.NumpyOps(np.float64, mixed=mixed), None, max_levels=levels, cycle=cyc,
                               smoother=kind, omega=omega, agglomerate_at=agg, coarse_maxit=40, mode="fused")
    s.set_coefficient(_a_at(NX, NY))
    s.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    if mixed:
        assert s.ldt[0] == np.float64 and s.ldt[s.split] == np.float32 and s.ldt[-1] == np.float64
        if agg == 17:
            assert s.Ld > s.split                        # an fp32 level is decomposed
    n0 = s.residual_norm()                               # the stand-alone norm path of the variable-coefficient operator
    mg0 = O.VarMGOracle(_a_at(NX, NY)(np.arange(NX), np.arange(NY)), max_levels=levels)
    np.testing.assert_allclose(n0, mg0.residual_norm(u0, rhs, 0), rtol=1e-13)
    hist = []
    for _ in range(2):
        s.cycle(0)
        hist.append(s.residual_norm())
    np.testing.assert_array_equal(H.assemble(s, NX, NY), u_ref)
    np.testing.assert_allclose(hist, h_ref, rtol=1e-13)


def test_mixed_levels_constant_coefficient_virtual_ranks():
    """Per-level mixed precision alone (PrecisionManager('mixed')): decomposed == MGOracle with the per-level policy."""
    NX = NY = 257
    levels = len(D.hierarchy_shapes(NX, NY, 99))
    mgo = O.MGOracle(NX, NY, (0.0, 1.0, 0.0, 1.0), np.float64, -1.0, levels, "V", 2, 2, "rbgs", 1.0, "vectorized", coarse_maxit=40)
    pm = O.OraclePrecision("mixed")
    rhs, u0 = _rhs(NX, NY, (0.0, 1.0, 0.0, 1.0)), _u0(NX, NY)
    u = u0.copy()
    for _ in range(2):
        mgo.rhs[0] = rhs.copy()
        u = mgo.cycle_once(u, 0, pm)
    s = D.DistributedMultigrid(NX, NY, 2, 2, range(4), H.NumpyOps(np.float64, mixed=True), None, max_levels=levels, cycle="V",
                               smoother="rbgs", omega=1.0, agglomerate_at=17, coarse_maxit=40, mode="fused")
    s.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    for _ in range(2):
        s.cycle(0)
    np.testing.assert_array_equal(H.assemble(s, NX, NY), u)


def _var_worker(rank, world, port, px, py, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    NX = NY = 257
    levels = len(D.hierarchy_shapes(NX, NY, 99))
    rhs, u0 = _rhs(NX, NY, (0.0, 1.0, 0.0, 1.0)), _u0(NX, NY)
    s = D.DistributedMultigrid(NX, NY, px, py, [rank], H.NumpyOps(np.float64, mixed=True), dist, max_levels=levels, cycle="W",
                               smoother="rbgs", omega=1.0, agglomerate_at=33, coarse_maxit=40, mode="fused")
    s.set_coefficient(_a_at(NX, NY))
    s.set_problem(lambda b: rhs[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny], lambda b: u0[b.gx0:b.gx0 + b.lnx, b.gy0:b.gy0 + b.lny])
    hist = []
    for _ in range(2):
        s.cycle(0)
        hist.append(s.residual_norm())
    b, u = s.local_solution(rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, (b.gx0, b.gy0, b.i_lo, b.i_hi, b.j_lo, b.j_hi, u))
    plan_ok = True
    if s.mode == "fused":
        # what a native cycle plan of this rank would hand to RCCL (dist_plan.PlanRecorder): one more cycle with the
        # recorder attached; every rank's communication groups must mirror its peers', group by group
        from mixed_precision_multigrid_solvers_for_pdes_amd import dist_plan
        s._rec = dist_plan.PlanRecorder()
        s._last_norm_parts = None
        s._cycle_fused(0, False)
        s.allreduce_sum(s._last_norm_parts)
        sig = s._rec.signature()
        s._rec = None
        sigs = [None] * world
        dist.all_gather_object(sigs, sig)
        if rank == 0:
            plan_ok = _plans_mirror_each_other(sigs)
    if rank == 0:
        full = np.full((NX, NY), np.nan)
        for gx0, gy0, i_lo, i_hi, j_lo, j_hi, ul in gathered:
            full[gx0 + i_lo:gx0 + i_hi, gy0 + j_lo:gy0 + j_hi] = ul[i_lo:i_hi, j_lo:j_hi]
        np.savez(out_path, u=full, hist=np.array(hist), plan_ok=plan_ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_multiprocess_config5_ingredients(tmp_path, world):
    """Config 5's cycle (variable coefficient, per-level mixed, W(2,2) red-black GS, fused legs, ghost width 13) over
    real processes and gloo."""
    import torch.multiprocessing as mp
    px, py = D.process_grid(world)
    out = str(tmp_path / "res.npz")
    mp.spawn(_var_worker, args=(world, _free_port(), px, py, out), nprocs=world, join=True)
    res = np.load(out)
    u_ref, h_ref = _var_oracle(257, 257, len(D.hierarchy_shapes(257, 257, 99)), "W", "rbgs", 1.0, 2, True)
    np.testing.assert_array_equal(res["u"], u_ref)
    np.testing.assert_allclose(res["hist"], h_ref, rtol=1e-13)
    assert bool(res["plan_ok"])            # W-cycle re-visits included: the recorded RCCL steps pair up across the ranks
