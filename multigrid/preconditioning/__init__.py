"""multigrid.preconditioning (reference: src/multigrid/preconditioning/__init__.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.applications import MultigridPreconditioner   # noqa: F401
