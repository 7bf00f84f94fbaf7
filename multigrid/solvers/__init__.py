"""multigrid.solvers (reference: src/multigrid/solvers/__init__.py:1-21 + the README's MixedPrecisionMultigrid)."""
from mixed_precision_multigrid_solvers_for_pdes_amd import (                     # noqa: F401
    BaseSolver, EnhancedJacobiSolver, GaussSeidelSmoother, GPUMultigridSolver, IterativeSolver, JacobiSmoother,
    MixedPrecisionMultigrid, MultigridCycle, MultigridSolver, WeightedJacobiSmoother)
