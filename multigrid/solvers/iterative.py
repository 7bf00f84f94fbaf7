"""multigrid.solvers.iterative (reference: src/multigrid/solvers/iterative.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.smoothers import EnhancedJacobiSolver   # noqa: F401
