"""multigrid.preconditioning.multigrid_preconditioner (reference: src/multigrid/preconditioning/multigrid_preconditioner.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.applications import MultigridPreconditioner   # noqa: F401
