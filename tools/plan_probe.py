#!/usr/bin/env python3
"""Native cycle plans (dist_plan.py) against the Python driver: virtual ranks of a px x py decomposition on ONE GPU.
  plan_probe.py px py n [dtype]      per-cycle wall time of both drivers + bit-for-bit comparison of iterate and norms
  plan_probe.py rccl                 one-rank RCCL communicator through mg_comm_init: all-reduce, all-gather and a
                                     send/recv-to-self group replayed from a plan (run it under `timeout`)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixed_precision_multigrid_solvers_for_pdes_amd import distributed as D, dist_plan as P, _lib


def rccl_selftest():
    dev = torch.device("cuda", 0)
    comm = P.RcclComm.single(0)
    print("communicator up", flush=True)
    a = torch.tensor([1.5], dtype=torch.float64, device=dev)
    b = torch.tensor([2.25], dtype=torch.float64, device=dev)
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    src = torch.arange(64, dtype=torch.float32, device=dev).reshape(4, 16).contiguous()
    dst = torch.zeros_like(src)
    gat = torch.zeros_like(src)
    rec = P.PlanRecorder()
    rec.add(out, a, b)
    rec.allreduce(out)
    rec.group([(0, src)], [(0, dst)])
    rec.allgather(src, gat)
    rec.result(out)
    plan = P.CyclePlan(rec, comm, 0)
    s = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()              # a communication stream of its own: the calls recorded on the compute stream are fenced over to it
    for k in range(3):
        dst.zero_(); gat.zero_()
        v = plan.run(s, side.cuda_stream if k else s)
        torch.cuda.synchronize()
        assert v == 3.75, v
        assert torch.equal(dst, src) and torch.equal(gat, src)
        print(f"run {k}: all-reduce {v}, self send/recv and all-gather exact", flush=True)
    plan.close(); comm.close()
    print("RCCL_SELFTEST_OK")


def main():
    if sys.argv[1] == "rccl":
        return rccl_selftest()
    px, py, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dtype = np.dtype(sys.argv[4]) if len(sys.argv) > 4 else np.dtype(np.float64)
    NX, NY = px * (n - 1) + 1, py * (n - 1) + 1
    dom = (0.0, float(px), 0.0, float(py))
    out = {}
    for native in (False, True):
        ops = D.HipOps(dtype, torch.device("cuda", 0), managed_single=(dtype == np.float32))
        s = D.DistributedMultigrid(NX, NY, px, py, range(px * py), ops, None, domain=dom, smoother="jacobi", omega=0.8, native=native)
        s.set_problem(lambda b: D.sine_rhs_block(b, dom))
        hist = []
        for _ in range(3):
            s.cycle(0); hist.append(s.residual_norm())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            s.cycle(0); hist.append(s.residual_norm())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        u = []
        for r in s.ranks:                      # the cells the rank owns (a queued front part refreshes ghost cells early)
            b, full = s.local_solution(r)
            u.append(full[b.i_lo:b.i_hi, b.j_lo:b.j_hi].copy())
        out[native] = (hist, u)
        plans = s._all_plans()
        copies = tuple(sum(q.copy_launches()[k] for q in plans) for k in (0, 1))
        nops = sum(q.n for q in plans)
        print(f"native={native}: {px}x{py} virtual ranks, {n}^2 each, Ld={s.Ld}: {dt*1e3:.3f} ms/cycle -> {dt*1e3/(px*py):.3f} ms per rank-cycle"
              f" (native cycles {s.native_cycles}, plan ops {nops}, {copies[0]} copies in {copies[1]} launches)", flush=True)
        s.close()
    same = out[False][0] == out[True][0] and all(np.array_equal(a, b) for a, b in zip(out[False][1], out[True][1]))
    print("bit-for-bit:", same, "(norms", out[False][0] == out[True][0], ")")
    assert same


main()
