#!/usr/bin/env python3
"""bench.py -- MDoF/s per V-cycle on 2-D Poisson (BASELINE.json metric), MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload at N=1: BASELINE config 3, the configuration north_star quotes the roofline target on:
2-D Poisson 4097^2, V(2,2) weighted Jacobi (omega 0.8), 11 levels (coarsest 5x5), adaptive
fp32 -> fp64 with switch_threshold 1e-6, f = 2 pi^2 sin(pi x) sin(pi y), u0 = 0 (synthetic).
One step = one V-cycle of the solve loop (policy check, cycle, ||r||) with all fields resident in HBM.
N > 1: the same workload per GPU on a px x py block decomposition (weak scaling), one process per GPU.  Started
under torch.distributed.run the ranks come from the environment; started as plain `python bench.py --gpus N` this
process only spawns the N ranks (before anything touches the GPU), waits and relays rank 0's line.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6300 achievable)
HBM_ACHIEVABLE_GBS = 6300.0


def sine_rhs(nx, ny, dtype=None):
    import numpy as np
    x = np.linspace(0.0, 1.0, nx)
    y = np.linspace(0.0, 1.0, ny)
    return (2 * np.pi**2 * np.sin(np.pi * x)[:, None] * np.sin(np.pi * y)[None, :]).astype(dtype or np.float64)


def cpu_baseline(n, levels, seconds_budget=12.0):
    """The oracle timed on this host on a bounded sample of the same workload: whole V(2,2) Jacobi cycles at n^2 in
    fp64.  Preferred: the C restatement (oracle/mg_oracle.c, OpenMP over all host cores it is given); fallback: the
    NumPy restatement on one core.  Both reproduce the reference's CPU arithmetic (tests/test_oracle_golden.py)."""
    import numpy as np
    from oracle import mg_oracle as O
    rhs = O.sine_rhs(n, n)
    try:
        from oracle.c_oracle import COracle
        # one GPU's share of the host: at most 16 threads (a 128-thread team on the shared box runs 3x SLOWER)
        share = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        co = COracle(n, n, max_levels=levels, cycle="V", smoother="jacobi", omega=0.8, threads=share)
        co.set_problem(rhs)
        co.cycle()                                   # warm-up (page faults, thread pool)
        cycles, t0 = 0, time.time()
        while True:
            co.cycle()
            cycles += 1
            el = time.time() - t0
            if el > seconds_budget or cycles >= 2000:
                break
        threads = co.threads
        co.close()
        return {"value": n * n * cycles / el / 1e6, "unit": "MDoF/s", "cores": threads, "kind": "port",
                "sample": f"{cycles} V(2,2) Jacobi cycles of the {n}^2 fp64 problem, C oracle (OpenMP, {threads} threads), {el:.1f} s"}
    except Exception as exc:                          # no compiler / OpenMP on this host
        mgo = O.MGOracle(n, n, max_levels=levels, cycle="V", smoother="jacobi", omega=0.8, jacobi_form="vectorized")
        mgo.rhs[0] = rhs.copy()
        u = np.zeros_like(rhs)
        cycles, t0 = 0, time.time()
        while True:
            u = mgo.cycle_once(u, 0)
            cycles += 1
            el = time.time() - t0
            if el > seconds_budget or cycles >= 8:
                break
        return {"value": n * n * cycles / el / 1e6, "unit": "MDoF/s", "cores": 1, "kind": "port",
                "sample": f"{cycles} V(2,2) Jacobi cycles of the {n}^2 fp64 problem, NumPy oracle ({type(exc).__name__}: C oracle unavailable), {el:.1f} s"}


def reference_cpu_captured():
    """The reference's OWN CPU V-cycle (solvers/multigrid.py with the NumPy Jacobi twin, oracle configuration), timed in
    the build container by tests/golden/generate_golden.py -- the reference cannot travel to the GPU box, so its figure
    is a committed capture, at the bench size itself when tests/golden/large_4097.npz exists."""
    import numpy as np
    for n, name in ((4097, "large_4097.npz"), (1025, "large_1025.npz")):
        path = os.path.join(ROOT, "tests", "golden", name)
        if os.path.exists(path):
            spc = float(np.load(path)["seconds_per_cycle"])
            return {"value": n * n / spc / 1e6, "unit": "MDoF/s", "cores": 1, "kind": "reference", "seconds_per_cycle": spc,
                    "sample": f"the reference's own MultigridSolver (LaplacianOperator(-1), EnhancedJacobiSolver 0.8, V(2,2)) at "
                              f"{n}^2 fp64: {spc:.2f} s/cycle on 1 core of the build container (tests/golden/{name}, written by "
                              f"tests/golden/generate_golden.py); it cannot travel to the GPU box"}
    return None


def first_below(vals, thr):
    for k, v in enumerate(vals):
        if v < thr:
            return k + 1
    return None


def floor_of(hist):
    """(floor, iterations_to_floor): the plateau the reference's absolute h-scaled norm stalls on = the
    median of the last five entries; reached at the first cycle within 2x of it."""
    tail = sorted(hist[-5:])
    floor = tail[len(tail) // 2]
    return floor, first_below(hist, 2.0 * floor)


def supervise(commands, poll_s=0.2, grace_s=5.0):
    """Run the rank processes to the end: commands = [(argv, env), ...]; rank 0's stdout is captured (through a temporary
    file, so a long line cannot block it), the others' is dropped.  As soon as ONE rank exits non-zero the remaining ranks
    are terminated (SIGTERM, then SIGKILL after `grace_s`): a rank whose peer died would otherwise wait for it in a
    collective until somebody kills the job.  Returns (exit codes, rank-0 stdout)."""
    import tempfile
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r, (argv, env) in enumerate(commands):
            procs.append(subprocess.Popen(argv, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
        failed = False
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                failed = True
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(poll_s)
        if failed:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + grace_s
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        codes = [p.wait() for p in procs]
        out0.seek(0)
        return codes, out0.read()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has made
    no GPU call and makes none), supervise them (a failing rank takes the others down), relay rank 0's JSON line, fail if
    any rank fails."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    commands = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        commands.append(([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env))
    codes, out0 = supervise(commands)
    line = None
    for ln in (out0 or "").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            line = ln
    if line is not None:
        print(line, flush=True)
    if any(codes) or line is None:
        sys.stderr.write(f"bench.py: rank exit codes {codes}, rank-0 line {'found' if line else 'missing'}\n")
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid-n", dest="n", type=int, default=4097, help="grid points per direction per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--agglomerate-at", dest="agglomerate_at", type=int, default=1025,
                    help="N > 1: levels of at most this many points per direction are solved replicated on every GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        from mixed_precision_multigrid_solvers_for_pdes_amd import distributed as dist_mg
        return dist_mg.bench_main(args, rank, local_rank, world)

    import numpy as np
    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    import mixed_precision_multigrid_solvers_for_pdes_amd as mg
    from mixed_precision_multigrid_solvers_for_pdes_amd import _lib

    n, K, W = args.n, args.steps, args.warmup
    levels = mg.default_max_levels(n, n)
    eng = mg.MultigridEngine(n, n, max_levels=levels, cycle="V", pre=2, post=2, smoother=_lib.MG_JACOBI, omega=0.8,
                             precision=_lib.MG_PREC_ADAPTIVE, switch_threshold=1e-6, adaptive_reference_rule=False,
                             device=local_rank)
    eng.set_rhs(sine_rhs(n, n))
    eng.set_solution(None)
    if W > 0:
        eng.iterate(tol=0.0, max_iterations=W)            # untimed warm-up steps
    eng.set_solution(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = eng.iterate(tol=0.0, max_iterations=K)            # exactly K steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = n * n * K / dt / 1e6
    hist = r["residual_history"]
    codes = r["precision_codes"]
    r0 = r["initial_residual"]
    # iterations to tolerance (north_star: "iterations-to-1e-10").  The reference's absolute, h-scaled norm cannot reach
    # 1e-10 for n >= 1025 in fp64: an untimed 40-cycle run of the same solve gives the plateau it stalls
    # on and the cycle that reaches it, next to the relative and 1e-9 forms.
    eng.set_solution(None)
    long_hist = eng.iterate(tol=0.0, max_iterations=max(40, K))["residual_history"]
    floor, its_floor = floor_of(long_hist)
    its_rel = first_below([v / r0 for v in long_hist], 1e-10)
    its_abs9 = first_below(long_hist, 1e-9)
    its_abs10 = first_below(long_hist, 1e-10)

    # roofline leg: hipEvent-timed launches of the level-0 kernels on the engine's own stream (mg_time_op).
    # `moved` = the bytes a launch MUST move (w = bytes per word): Jacobi sweep 3w (read u, rhs; write u'); a fused leg
    # 3.25w (read u, rhs [+ the coarse e, w/4]; write u' [+ the coarse rhs, w/4]) whatever it computes on chip.
    # achieved = moved / launch time, frac = achieved / 8 TB/s <= 1.  The per-operator accounting of SURVEY 8(d)
    # (what the same work costs as one launch per operator) is kept as unfused_equivalent_*.
    reps = 30
    nn = n * n

    def leg(op, dtype, w, moved_words, unfused_words):
        ms = eng.time_op(op, 0, dtype, reps)
        gbs = moved_words * w * nn / (ms * 1e-3) / 1e9
        return {"launch_ms": ms, "bytes_per_launch": int(moved_words * w * nn), "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                "frac_of_achievable": gbs / HBM_ACHIEVABLE_GBS,
                "unfused_equivalent_bytes": int(unfused_words * w * nn),
                "unfused_equivalent_gbs": unfused_words * w * nn / (ms * 1e-3) / 1e9}
    kern = {}
    for name, dtype, w in (("f32", np.float32, 4), ("f64", np.float64, 8)):
        ws = 3 * w * n * mg_pitch(_lib, dtype, n) / 2**20
        kern[name] = {
            # jacobi_kernel<T,1,..>, one sweep.  `jacobi_sweep` ping-pongs ONE {u, rhs, t} set (fp32: 208 MiB, resident in
            # the 256 MiB Infinity Cache across back-to-back launches); `jacobi_sweep_hbm` rotates independent sets of
            # > 768 MiB in total, so every launch reads from HBM proper: the figure north_star's 70 % target is stated on
            "jacobi_sweep": dict(leg("jacobi", dtype, w, 3.0, 3.0), working_set_mib=ws,
                                 served_from="Infinity Cache (MALL)" if ws < 256 else "HBM"),
            "jacobi_sweep_hbm": dict(leg("jacobi_hbm", dtype, w, 3.0, 3.0), served_from="HBM (rotating sets > 768 MiB)"),
            # the same traffic with no stencil at all (c = a + b over the same rotating sets): the memory system's ceiling
            "stream_hbm": dict(leg("stream_hbm", dtype, w, 3.0, 3.0), served_from="HBM (rotating sets > 768 MiB)"),
            "jacobi_2sweeps": leg("sweeps2", dtype, w, 3.0, 6.0),                    # fused_jacobi_kernel<T,2,false,0,..>
            "down_leg": leg("down_leg", dtype, w, 3.25, 2 * 3.0 + 2.25),             # 2 sweeps + residual + restriction
            "up_leg": leg("up_leg", dtype, w, 3.25, 2.25 + 2 * 3.0 + 2.0),           # prolong-add + 2 sweeps + norm
        }
        # the spanning leg (up leg of cycle k + down leg of cycle k + 1 in one launch, mg_config.speculate = 2): reads u, rhs,
        # e/4; writes u' and rhs_coarse/4 (3.5 w), plus the iterate in between (4.5 w) when a cycle may end the solve
        try:
            kern[name]["span_leg"] = leg("span_leg", dtype, w, 4.5, 2.25 + 4 * 3.0 + 2.0 + 2.25)
            kern[name]["span_leg_nomid"] = leg("span_leg_nomid", dtype, w, 3.5, 2.25 + 4 * 3.0 + 2.0 + 2.25)
        except Exception:
            pass                                                                     # this hierarchy runs the two-launch form
    f32_cycles = sum(1 for c in codes if c == 0)
    f64_cycles = K - f32_cycles
    # the timed region is mg_iterate(tol = 0, K): with a spanning leg every cycle but the first and the last is one
    # span_leg_nomid launch on level 0 (nothing can end the solve, the iterate in between is not stored)
    def level0_ms(p):
        k = kern[p]
        return k["span_leg_nomid"]["launch_ms"] if "span_leg_nomid" in k else k["down_leg"]["launch_ms"] + k["up_leg"]["launch_ms"]
    t32, t64 = f32_cycles * level0_ms("f32"), f64_cycles * level0_ms("f64")
    dom_p = "f64" if t64 >= t32 else "f32"
    if "span_leg_nomid" in kern[dom_p]:
        dom_k = "span_leg_nomid"
    else:
        dom_k = "up_leg" if kern[dom_p]["up_leg"]["launch_ms"] >= kern[dom_p]["down_leg"]["launch_ms"] else "down_leg"
    dom = kern[dom_p][dom_k]
    traffic, traffic_source, traffic_build = None, None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    from mixed_precision_multigrid_solvers_for_pdes_amd import _build
    build_now = _build.source_hash()
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            traffic = pmc["kernels"][f"{dom_k}_{dom_p}_{n}"]["hbm_bytes_per_launch_corrected"]
            traffic_build = pmc.get("source_hash")
            traffic_source = ("profiles/pmc_latest.json: stored rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel "
                              "(FETCH doubled per the gfx950 note); not measured in this run -- traffic_build is the source hash "
                              "of the library those passes ran, build the one of this run")
        except Exception:
            traffic = None
    roof = {"bound": "hbm",
            "kernel": (f"rb_span_kernel {dom_p} at {n}^2 (level 0: up leg of cycle k + down leg of cycle k + 1 in one launch, the "
                       "iterate in between not stored): the largest time share of the timed region") if dom_k.startswith("span")
                      else f"rb_leg_kernel {dom_k} {dom_p} at {n}^2 (level 0, register-blocked fused leg): the largest time share of the timed region",
            "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": traffic,
            "traffic_source": traffic_source, "traffic_build": traffic_build, "build": build_now,
            "traffic_is_from_this_build": bool(traffic_build) and traffic_build == build_now,
            "launch_ms": dom["launch_ms"], "bytes_per_launch": dom["bytes_per_launch"],
            "frac_of_achievable": dom["frac_of_achievable"], "achievable_gbs": HBM_ACHIEVABLE_GBS,
            "unfused_equivalent_bytes": dom["unfused_equivalent_bytes"], "unfused_equivalent_gbs": dom["unfused_equivalent_gbs"],
            "smoother_hbm": {"kernel": f"jacobi_kernel f32 at {n}^2, operands rotating through > 768 MiB (HBM proper)",
                             "achieved": kern["f32"]["jacobi_sweep_hbm"]["achieved"], "frac": kern["f32"]["jacobi_sweep_hbm"]["frac"],
                             "target_frac": 0.70,
                             "stream_ceiling_gbs": kern["f32"]["stream_hbm"]["achieved"],
                             "frac_of_stream_ceiling": kern["f32"]["jacobi_sweep_hbm"]["achieved"] / kern["f32"]["stream_hbm"]["achieved"],
                             "infinity_cache_resident_frac": kern["f32"]["jacobi_sweep"]["frac"],
                             "note": "stream_ceiling = a stencil-free c = a + b kernel over the same rotating buffers, timed in this "
                                     "run: what the memory system delivers for 2 reads + 1 write of this size; the sweep runs "
                                     "at frac_of_stream_ceiling of it"},
            "note": "achieved = bytes the launch must move (3.5 words/DoF for the spanning leg, 3.25 for a fused leg, 3 for a sweep) / hipEvent-timed "
                    "launch time; unfused_equivalent_* prices the same work as one launch per operator (SURVEY 8d) and may "
                    "exceed the HBM peak",
            "kernels": kern}

    # ---- time to solution: cycles and device time to the plateau of the absolute norm, per policy ----
    def to_floor(engine):
        engine.set_solution(None)
        h40 = engine.iterate(tol=0.0, max_iterations=40)["residual_history"]
        fl, its = floor_of(h40)
        engine.set_solution(None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rr = engine.iterate(tol=0.0, max_iterations=its)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        return {"iterations_to_floor": its, "time_to_floor_ms": el * 1e3, "residual_floor": fl, "ms_per_cycle": el / its * 1e3,
                "cycles_fp32": sum(1 for c in rr["precision_codes"] if c == 0), "switch_reason": rr.get("switch_reason"),
                "fp32_floor_estimate": rr.get("fp32_floor")}
    tts = {"adaptive": to_floor(eng)}
    for name, prec in (("double", _lib.MG_PREC_DOUBLE), ("defect", _lib.MG_PREC_DEFECT)):
        e2 = mg.MultigridEngine(n, n, max_levels=levels, cycle="V", pre=2, post=2, smoother=_lib.MG_JACOBI, omega=0.8, precision=prec,
                                device=local_rank)
        e2.set_rhs(sine_rhs(n, n))
        e2.iterate(tol=0.0, max_iterations=2)            # warm-up
        tts[name] = to_floor(e2)
        e2.close()
    tts["note"] = ("floor = the plateau of the reference's absolute h-scaled norm (SURVEY F10); time = device-resident mg_iterate to "
                   "the first cycle within 2x of it.  adaptive: the fp32 phase is entered only where its residual floor eps32 * diag(A) "
                   "* ||u||, bounded a priori by eps32 * diag(A) / lambda_min * ||r_0||, leaves it at least two cycles (switch_reason "
                   "fp32_skipped otherwise: 4097^2 and larger), and ends as soon as ||r|| is within 2x of that floor; defect: fp32 "
                   "cycles on the error equation of an fp64 iterate")
    # ---- the smoother against the stream ceiling by size: does the ceiling climb with launch length? ----
    sweep = []
    for m in (4097, 8193, 16385):
        try:
            if m == n:
                e3 = None
                a, b = kern["f32"]["jacobi_sweep_hbm"], kern["f32"]["stream_hbm"]
                ms_j, ms_s = a["launch_ms"], b["launch_ms"]
            else:
                e3 = mg.MultigridEngine(m, m, max_levels=3, cycle="V", smoother=_lib.MG_JACOBI, omega=0.8, precision=_lib.MG_PREC_SINGLE,
                                        device=local_rank)
                e3.set_rhs(sine_rhs(m, m, np.float32))
                ms_j, ms_s = e3.time_op("jacobi_hbm", 0, np.float32, 10), e3.time_op("stream_hbm", 0, np.float32, 10)
                e3.close()
            byts = 3.0 * 4 * m * m
            sweep.append({"n": m, "dtype": "f32", "jacobi_sweep_hbm_ms": ms_j, "jacobi_sweep_hbm_gbs": byts / (ms_j * 1e-3) / 1e9,
                          "jacobi_frac_of_8TBs": byts / (ms_j * 1e-3) / 1e9 / HBM_PEAK_GBS, "stream_hbm_ms": ms_s,
                          "stream_hbm_gbs": byts / (ms_s * 1e-3) / 1e9, "stream_frac_of_8TBs": byts / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS})
        except Exception as exc:                          # a box without the memory for the 16385^2 rotating sets
            sweep.append({"n": m, "error": repr(exc)})
    out = {
        "metric": "MDoF/s per V-cycle on 2D Poisson", "value": value, "unit": "MDoF/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f64" if f32_cycles == 0 else "f32" if f64_cycles == 0 else "f32->f64") + " (adaptive policy)", "data": "synthetic",
        "config": {"workload": f"2D Poisson {n}^2 adaptive fp32->fp64 (switch_threshold=1e-6), V(2,2) weighted-Jacobi "
                               f"omega=0.8, {levels} levels, 1xMI355X", "grid": [n, n], "levels": levels,
                   "cycle": "V(2,2)", "smoother": "jacobi", "parallelism": "1 GPU",
                   # engine defaults of this run (include/mghip.h, mg_config): the up leg of cycle k and the down leg of cycle k + 1
                   # of the finest level are one launch; the 5 x 5 coarsest grid is solved directly (within 1e-12 of the reference's
                   # iteration to coarse_tol); the coarse levels from 65^2 down run in one register-resident workgroup
                   "engine": {"speculate": int(eng.cfg.speculate), "coarse_direct": int(eng.cfg.coarse_direct), "tail": int(eng.cfg.tail),
                              "fused": int(eng.cfg.fused)}},
        "cycles_fp32": f32_cycles, "cycles_fp64": f64_cycles,
        "residual_initial": r0, "residual_first": hist[0], "residual_last": hist[-1],
        "iterations": K,
        "iterations_to_1e-10_absolute": its_abs10, "iterations_to_1e-10_relative": its_rel, "iterations_to_1e-9_absolute": its_abs9,
        "residual_floor": floor, "iterations_to_floor": its_floor, "time_to_floor_ms": tts["adaptive"]["time_to_floor_ms"],
        "switch_reason": tts["adaptive"]["switch_reason"], "time_to_solution": tts,
        "smoother_size_sweep": {"rows": sweep,
                                "note": "single Jacobi sweep and the stencil-free c = a + b stream, fp32, operands rotating through "
                                        "> 768 MiB (HBM proper), by grid size: if the fractions climb with n the 4097^2 figures are "
                                        "launch-length-limited (a 25-40 us launch spends a visible share ramping up and draining); "
                                        "if they do not, ~0.6 of 8 TB/s is what 2 reads + 1 write get from this memory system"},
        "reference_cpu_captured": reference_cpu_captured(),
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, levels)
    eng.close()
    print(json.dumps(out))
    return 0


def mg_pitch(_lib, dtype, n):
    import ctypes as C
    ld = C.c_int(0)
    _lib.check(_lib.load().mg_pitch_elems(_lib.dtype_code(dtype), n, C.byref(ld)))
    return ld.value


if __name__ == "__main__":
    sys.exit(main() or 0)
