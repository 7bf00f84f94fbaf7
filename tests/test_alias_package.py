"""The reference's import paths resolve to the MI355X-native classes (drop-in namespace `multigrid`)."""
import importlib

import pytest

import mixed_precision_multigrid_solvers_for_pdes_amd as mg

PATHS = [
    ("multigrid", ["Grid", "LaplacianOperator", "MultigridSolver", "PrecisionManager"]),
    ("multigrid.core", ["Grid", "PrecisionManager", "PrecisionLevel"]),
    ("multigrid.core.grid", ["Grid"]),
    ("multigrid.core.precision", ["PrecisionManager", "PrecisionLevel"]),
    ("multigrid.operators", ["LaplacianOperator", "RestrictionOperator", "ProlongationOperator"]),
    ("multigrid.operators.laplacian", ["LaplacianOperator"]),
    ("multigrid.operators.transfer", ["RestrictionOperator", "ProlongationOperator"]),
    ("multigrid.operators.base", ["BaseOperator"]),
    ("multigrid.solvers", ["MultigridSolver", "MixedPrecisionMultigrid", "JacobiSmoother", "GaussSeidelSmoother"]),
    ("multigrid.solvers.multigrid", ["MultigridSolver", "MultigridCycle"]),
    ("multigrid.solvers.smoothers", ["JacobiSmoother", "WeightedJacobiSmoother", "GaussSeidelSmoother"]),
    ("multigrid.solvers.iterative", ["EnhancedJacobiSolver"]),
    ("multigrid.solvers.base", ["BaseSolver", "IterativeSolver"]),
    ("multigrid.gpu", ["GPUMemoryManager", "GPUMemoryPool", "CUDAKernels", "SmoothingKernels", "TransferKernels", "GPUMultigridSolver",
                       "GPUCommunicationAvoidingMultigrid", "GPUPrecisionManager"]),          # gpu/__init__.py:3-6
    ("multigrid.gpu.gpu_solver", ["GPUMultigridSolver", "GPUCommunicationAvoidingMultigrid"]),
    ("multigrid.gpu.gpu_precision", ["GPUPrecisionManager", "GPUPrecisionLevel"]),
    ("multigrid.gpu.memory_manager", ["GPUMemoryManager", "GPUMemoryPool", "check_gpu_availability"]),
    ("multigrid.gpu.multi_gpu", ["DistributedMultigridSolver", "MultiGPUManager"]),
    ("multigrid.gpu.multi_gpu_solver", ["MultiGPUSolver", "DecompositionType"]),
    ("multigrid.gpu.cuda_kernels", ["CUDAKernels", "SmoothingKernels", "TransferKernels", "MixedPrecisionKernels"]),
    ("multigrid.problems", ["PoissonProblem"]),
    ("multigrid.applications", ["PoissonSolver2D", "PoissonProblem", "HeatEquationSolver"]),
    ("multigrid.applications.poisson_solver", ["PoissonSolver2D", "PoissonProblem"]),
    ("multigrid.applications.heat_equation", ["HeatEquationSolver", "HeatEquationConfig", "TimeSteppingScheme", "BoundaryType",
                                              "BoundaryCondition", "create_gaussian_initial_condition"]),
    ("multigrid.preconditioning", ["MultigridPreconditioner"]),
    ("multigrid.preconditioning.multigrid_preconditioner", ["MultigridPreconditioner"]),
]


@pytest.mark.parametrize("module,names", PATHS)
def test_reference_import_paths(module, names):
    m = importlib.import_module(module)
    assert m.__file__ and "/root/reference" not in m.__file__
    for n in names:
        obj = getattr(m, n)
        assert obj.__module__.startswith("mixed_precision_multigrid_solvers_for_pdes_amd"), (module, n, obj.__module__)


def test_readme_facade_names():
    from multigrid.solvers import MixedPrecisionMultigrid
    from multigrid.problems import PoissonProblem
    assert MixedPrecisionMultigrid is mg.MixedPrecisionMultigrid and PoissonProblem is mg.PoissonProblem


def test_the_reference_callers_import_lines_bind_to_the_hip_path():
    """applications/poisson_solver.py:15-19 of the reference guards exactly this line with `except ImportError` and then runs
    its CPU path SILENTLY; gpu/__init__.py:3-6 are the package's own imports.  Executed verbatim against this repo's
    `multigrid` namespace they must resolve -- to the MI355X classes."""
    ns = {}
    exec("from multigrid.gpu.gpu_solver import GPUMultigridSolver, GPUCommunicationAvoidingMultigrid", ns)
    exec("from multigrid.gpu import GPUMemoryManager, GPUMemoryPool, CUDAKernels, SmoothingKernels, TransferKernels", ns)
    exec("from multigrid.gpu import GPUPrecisionManager", ns)
    exec("from multigrid.gpu.multi_gpu import DistributedMultigridSolver", ns)
    assert ns["GPUCommunicationAvoidingMultigrid"] is mg.GPUCommunicationAvoidingMultigrid
    assert issubclass(ns["GPUCommunicationAvoidingMultigrid"], ns["GPUMultigridSolver"])
    assert ns["DistributedMultigridSolver"] is mg.DistributedMultigridSolver
    import multigrid
    assert multigrid.GPU_AVAILABLE is True
