"""multigrid.gpu.cuda_kernels (reference: src/multigrid/gpu/cuda_kernels.py): the kernel wrapper classes, on HIP."""
from mixed_precision_multigrid_solvers_for_pdes_amd.gpu_kernels import (HIPKernels as CUDAKernels, MixedPrecisionKernels,   # noqa: F401
                                                                        SmoothingKernels, TransferKernels)
