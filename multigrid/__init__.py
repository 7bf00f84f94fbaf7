"""`multigrid` -- the reference's import namespace, served by the MI355X-native engine.

    from multigrid.solvers import MixedPrecisionMultigrid      # README.md:73-93 of the reference
    from multigrid.problems import PoissonProblem
    from multigrid import Grid, LaplacianOperator, MultigridSolver, ...

Everything is re-exported from mixed_precision_multigrid_solvers_for_pdes_amd."""
from mixed_precision_multigrid_solvers_for_pdes_amd import *          # noqa: F401,F403
from mixed_precision_multigrid_solvers_for_pdes_amd import __all__, __version__   # noqa: F401

GPU_AVAILABLE = True      # the device path is the product; it raises at use if no MI355X is visible
