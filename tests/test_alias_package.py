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
    ("multigrid.gpu", ["GPUMultigridSolver"]),
    ("multigrid.gpu.gpu_solver", ["GPUMultigridSolver"]),
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
