"""multigrid.operators.transfer (reference: src/multigrid/operators/transfer.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.operators import ProlongationOperator, RestrictionOperator   # noqa: F401
