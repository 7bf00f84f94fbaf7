"""multigrid.solvers.multigrid (reference: src/multigrid/solvers/multigrid.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.solver import MultigridCycle, MultigridSolver   # noqa: F401
