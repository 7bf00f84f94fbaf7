"""multigrid.operators.laplacian (reference: src/multigrid/operators/laplacian.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.operators import LaplacianOperator   # noqa: F401
