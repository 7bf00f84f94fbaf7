"""multigrid.problems (README.md:75 of the reference; the package does not exist there)."""
from mixed_precision_multigrid_solvers_for_pdes_amd import PoissonProblem        # noqa: F401
