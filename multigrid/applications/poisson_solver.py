"""multigrid.applications.poisson_solver (reference: src/multigrid/applications/poisson_solver.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.applications import PoissonProblem, PoissonSolver2D   # noqa: F401
