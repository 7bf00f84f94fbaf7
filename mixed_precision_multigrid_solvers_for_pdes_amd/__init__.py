"""MI355X-native mixed-precision geometric multigrid: the V/W-cycle hot path of
Tani843/Mixed_Precision_Multigrid_Solvers_for_PDEs behind that project's own Python API.

Host side (this package): metadata, policy and the plugin classes of the reference
(Grid, LaplacianOperator, RestrictionOperator, ProlongationOperator, smoothers, PrecisionManager,
MultigridSolver, GPUMultigridSolver) plus the README facade (MixedPrecisionMultigrid,
PoissonProblem).  Device side: csrc/ -> lib/libmghip.so, reached through the C ABI in
include/mghip.h.  There is no CPU fallback for the device path."""
from .grid import Grid
from .operators import (BaseOperator, DiffusionOperator, HelmholtzOperator, LaplacianOperator, ProlongationOperator,
                        RestrictionOperator)
from .precision import PrecisionLevel, PrecisionManager
from .smoothers import (BaseSolver, ConvergenceHistory, EnhancedJacobiSolver, GaussSeidelSmoother,
                        IterativeSolver, JacobiSmoother, WeightedJacobiSmoother)
from .solver import GPUCommunicationAvoidingMultigrid, GPUMultigridSolver, MultigridCycle, MultigridSolver
from .engine import MultigridEngine
from .gpu_kernels import MixedPrecisionKernels, SmoothingKernels, TransferKernels
from .gpu_precision import GPUPrecisionLevel, GPUPrecisionManager
from .memory_manager import GPUMemoryManager, GPUMemoryPool
from .multi_gpu import DecompositionType, DistributedMultigridSolver, MultiGPUManager, MultiGPUSolver
from .facade import MixedPrecisionMultigrid, PoissonProblem, default_max_levels
from . import applications, heat_equation
from .heat_equation import HeatEquationConfig, HeatEquationSolver, TimeSteppingScheme
from .applications import MultigridPreconditioner, PoissonSolver2D

__all__ = [
    "Grid", "BaseOperator", "LaplacianOperator", "DiffusionOperator", "HelmholtzOperator", "RestrictionOperator", "ProlongationOperator",
    "PrecisionLevel", "PrecisionManager", "BaseSolver", "ConvergenceHistory", "IterativeSolver",
    "JacobiSmoother", "WeightedJacobiSmoother", "EnhancedJacobiSolver", "GaussSeidelSmoother",
    "MultigridSolver", "GPUMultigridSolver", "GPUCommunicationAvoidingMultigrid", "MultigridCycle", "MultigridEngine",
    "GPUPrecisionManager", "GPUPrecisionLevel", "GPUMemoryManager", "GPUMemoryPool",
    "DistributedMultigridSolver", "MultiGPUSolver", "MultiGPUManager", "DecompositionType",
    "SmoothingKernels", "TransferKernels", "MixedPrecisionKernels",
    "MixedPrecisionMultigrid", "PoissonProblem", "default_max_levels",
    "PoissonSolver2D", "MultigridPreconditioner", "applications", "heat_equation", "HeatEquationSolver",
    "HeatEquationConfig", "TimeSteppingScheme",
]
__version__ = "0.1.0"
