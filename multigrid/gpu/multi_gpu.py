"""multigrid.gpu.multi_gpu (reference: src/multigrid/gpu/multi_gpu.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.multi_gpu import DistributedMultigridSolver, MultiGPUManager   # noqa: F401
