"""multigrid.solvers.base (reference: src/multigrid/solvers/base.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.smoothers import BaseSolver, ConvergenceHistory, IterativeSolver   # noqa: F401
