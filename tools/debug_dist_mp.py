"""Multi-process rehearsal of DistributedMultigrid on ONE GPU over gloo: residual histories for a few configurations.
    MG_DIST_SAME_DEVICE=1 python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/debug_dist_mp.py 1025
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from mixed_precision_multigrid_solvers_for_pdes_amd import distributed as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
px, py = D.process_grid(world)
NX, NY = px * (n - 1) + 1, py * (n - 1) + 1
domain = (0.0, float(px), 0.0, float(py))
for mode, overlap, dt, managed in (("per_operator", False, np.float64, False), ("fused", False, np.float64, False),
                                   ("fused", True, np.float64, False), ("fused", True, np.float32, True)):
    ops = D.HipOps(dt, torch.device("cuda", 0), managed_single=managed)
    s = D.DistributedMultigrid(NX, NY, px, py, [rank], ops, dist, domain=domain, smoother="jacobi", omega=0.8, cycle="V", pre=2, post=2,
                               mode=mode, overlap=overlap)
    s.set_problem(lambda b: D.sine_rhs_block(b, domain))
    h = [s.residual_norm()]
    for _ in range(5):
        s.cycle(0); h.append(s.residual_norm())
    if rank == 0:
        print(f"{px}x{py} {mode:12s} overlap={overlap} {np.dtype(dt).name} Ld={s.Ld}:", ["%.3e" % v for v in h], flush=True)
    s.close()
dist.destroy_process_group()
