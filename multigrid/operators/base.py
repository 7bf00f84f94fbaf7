"""multigrid.operators.base (reference: src/multigrid/operators/base.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.operators import BaseOperator   # noqa: F401
