"""multigrid.solvers.smoothers (reference: src/multigrid/solvers/smoothers.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.smoothers import GaussSeidelSmoother, JacobiSmoother, WeightedJacobiSmoother   # noqa: F401
