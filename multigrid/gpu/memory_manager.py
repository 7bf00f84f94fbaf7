"""multigrid.gpu.memory_manager (reference: src/multigrid/gpu/memory_manager.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.memory_manager import GPUMemoryBlock, GPUMemoryManager, GPUMemoryPool, check_gpu_availability   # noqa: F401
