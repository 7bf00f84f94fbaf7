"""multigrid.gpu.gpu_precision (reference: src/multigrid/gpu/gpu_precision.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.gpu_precision import GPUPrecisionLevel, GPUPrecisionManager   # noqa: F401
