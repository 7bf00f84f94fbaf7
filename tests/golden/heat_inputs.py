"""Inputs of the heat-equation golden cases, shared by the generator (which runs the REFERENCE on them) and by the
tests (which run the oracle / the GPU product on them).  Data only: case table + configuration builder."""
import numpy as np


def heat_cases():
    """name -> (n, alpha, scheme, dt, steps, bc-kind, source?)  -- inputs shared with tests/test_heat_golden.py."""
    return {
        "explicit33": (33, 0.1, "explicit_euler", None, 5, "dirichlet_t", True),
        "implicit17": (17, 1.0, "implicit_euler", None, 3, "zero", False),
        "implicit33_src": (33, 0.5, "implicit_euler", 0.004, 2, "dirichlet_t", True),
        "cn17": (17, 1.0, "crank_nicolson", None, 3, "zero", False),
        "cn33_mixed_bc": (33, 0.25, "crank_nicolson", 0.005, 2, "mixed", True),
    }


def heat_config(mod, alpha, bc_kind, with_source):
    """The same configuration objects for the reference module and for ours (`mod` provides the classes)."""
    BC, BT = mod.BoundaryCondition, mod.BoundaryType
    wave = lambda x, y, t: 0.3 * np.sin(2 * np.pi * 1.5 * t) + 0.0 * x * y          # noqa: E731
    flux = lambda x, y, t: 0.2 * np.cos(np.pi * x) * (1.0 + t) + 0.0 * y            # noqa: E731
    zero = lambda x, y, t: 0.0                                                       # noqa: E731
    if bc_kind == "zero":
        bcs = None
    elif bc_kind == "dirichlet_t":
        bcs = {"left": BC(BT.DIRICHLET, wave), "right": BC(BT.DIRICHLET, zero),
               "bottom": BC(BT.DIRICHLET, zero), "top": BC(BT.DIRICHLET, wave)}
    else:   # mixed: Dirichlet(t) left, Neumann right and top, Robin... only 'left' exists in the reference: Dirichlet bottom
        bcs = {"left": BC(BT.DIRICHLET, wave), "right": BC(BT.NEUMANN, flux),
               "bottom": BC(BT.DIRICHLET, zero), "top": BC(BT.NEUMANN, flux)}
    src = (lambda x, y, t: np.sin(np.pi * x) * np.cos(2 * np.pi * y) * np.exp(-t)) if with_source else None
    return mod.HeatEquationConfig(thermal_diffusivity=alpha,
                                  initial_condition=mod.create_gaussian_initial_condition((0.4, 0.55), 0.12, 1.0),
                                  source_term=src, boundary_conditions=bcs)
