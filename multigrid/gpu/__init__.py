"""multigrid.gpu (reference: src/multigrid/gpu/__init__.py): the device-side driver classes."""
from mixed_precision_multigrid_solvers_for_pdes_amd import GPUMultigridSolver, MultigridEngine      # noqa: F401
