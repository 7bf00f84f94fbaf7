"""multigrid.core (reference: src/multigrid/core/__init__.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd import Grid, PrecisionLevel, PrecisionManager   # noqa: F401
