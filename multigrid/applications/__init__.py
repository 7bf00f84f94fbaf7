"""multigrid.applications (reference: src/multigrid/applications/): the dataclass PoissonProblem + PoissonSolver2D
(poisson_solver.py) and the heat-equation time stepper (heat_equation.py)."""
from mixed_precision_multigrid_solvers_for_pdes_amd.applications import PoissonProblem, PoissonSolver2D   # noqa: F401
from mixed_precision_multigrid_solvers_for_pdes_amd.heat_equation import (   # noqa: F401
    BoundaryCondition, BoundaryType, HeatEquationConfig, HeatEquationSolver, TimeSteppingScheme)
