"""multigrid.applications.heat_equation (reference: src/multigrid/applications/heat_equation.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.heat_equation import (   # noqa: F401
    BoundaryCondition, BoundaryType, HeatEquationConfig, HeatEquationSolver, TimeSteppingScheme,
    create_gaussian_initial_condition, create_time_dependent_boundary)
