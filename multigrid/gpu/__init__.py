"""multigrid.gpu (reference: src/multigrid/gpu/__init__.py:3-6): the device-side classes.  GPUPerformanceProfiler and
GPUBenchmarkSuite (:7-8) are measurement harnesses, replaced by rocprofv3 + bench.py (out of scope)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.memory_manager import GPUMemoryManager, GPUMemoryPool                                  # noqa: F401
from mixed_precision_multigrid_solvers_for_pdes_amd.gpu_kernels import HIPKernels as CUDAKernels, SmoothingKernels, TransferKernels        # noqa: F401
from mixed_precision_multigrid_solvers_for_pdes_amd.solver import GPUCommunicationAvoidingMultigrid, GPUMultigridSolver                    # noqa: F401
from mixed_precision_multigrid_solvers_for_pdes_amd.gpu_precision import GPUPrecisionManager                                              # noqa: F401
from mixed_precision_multigrid_solvers_for_pdes_amd.multi_gpu import DistributedMultigridSolver, MultiGPUManager, MultiGPUSolver          # noqa: F401
from mixed_precision_multigrid_solvers_for_pdes_amd import MultigridEngine                                                                # noqa: F401

__all__ = ["GPUMemoryManager", "GPUMemoryPool", "CUDAKernels", "SmoothingKernels", "TransferKernels", "GPUMultigridSolver",
           "GPUCommunicationAvoidingMultigrid", "GPUPrecisionManager", "DistributedMultigridSolver", "MultiGPUManager",
           "MultiGPUSolver", "MultigridEngine"]
