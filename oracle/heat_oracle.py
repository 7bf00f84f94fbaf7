"""CPU oracle of the reference's heat-equation time stepper -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

NumPy restatement of applications/heat_equation.py (reference lines cited per function), including the part the
GPU product replaces: `_solve_helmholtz` (:459-497), the reference's lexicographic Gauss-Seidel relaxation of
(-Laplace_h + lambda) u = rhs whose stop test never fires (sign error in the tested residual, :474), i.e. exactly
`sweeps` = 100 sweeps per implicit step.  PINNED: tests/golden/heat.npz holds outputs of the reference itself
(tests/golden/generate_golden.py heat) and tests/test_heat_golden.py checks this file against them.
With sweeps -> infinity the same function converges to the solution of the linear system, which is what the
product's multigrid solve returns; the tests use both ends.

Callables (initial condition, source, boundary values) are evaluated on coordinate arrays; the reference evaluates
them point by point, which gives the same numbers for elementwise NumPy callables.
"""
import numpy as np


def _eval(fn, *args):
    shape = np.broadcast(*[np.asarray(a) for a in args]).shape
    out = np.asarray(fn(*args), dtype=np.float64)
    return np.broadcast_to(out, shape).copy()


def compute_laplacian(u, hx, hy):                                   # heat_equation.py:430-442
    lap = np.zeros_like(u)
    lap[1:-1, 1:-1] = ((u[:-2, 1:-1] - 2 * u[1:-1, 1:-1] + u[2:, 1:-1]) / hx**2
                       + (u[1:-1, :-2] - 2 * u[1:-1, 1:-1] + u[1:-1, 2:]) / hy**2)
    return lap


def helmholtz_gs(rhs, lam, u0, hx, hy, sweeps=100, tol=1e-10):       # heat_equation.py:459-497
    """Lexicographic GS for (-Laplace + lam) u = rhs; the reference uses hx for both directions (:488-493).
    Anti-diagonal order = the reference's double loop (see mg_oracle.lexgs_sweep).  Returns (u, sweeps done)."""
    u = u0.copy()
    nx, ny = u.shape
    hx2 = hx**2
    done = 0
    for _ in range(sweeps):
        # the reference's stop test, with its sign error kept (:473-484); zero only for rhs = 0, u = 0
        res = rhs - compute_laplacian(u, hx, hy) + lam * u
        if np.linalg.norm(res[1:-1, 1:-1]) < tol:
            break
        for s in range(2, nx + ny - 3):
            i = np.arange(max(1, s - (ny - 2)), min(nx - 2, s - 1) + 1)
            j = s - i
            u[i, j] = (rhs[i, j] + (u[i - 1, j] + u[i + 1, j] + u[i, j - 1] + u[i, j + 1]) / hx2) / (4 / hx2 + lam)
        done += 1
    return u, done


class HeatOracle:
    """HeatEquationSolver (heat_equation.py:75-600) with `sweeps` Gauss-Seidel sweeps per implicit solve."""

    def __init__(self, config, nx, ny, domain=(0.0, 1.0, 0.0, 1.0), sweeps=100):
        self.cfg = config
        self.nx, self.ny = nx, ny
        self.hx = (domain[1] - domain[0]) / (nx - 1)
        self.hy = (domain[3] - domain[2]) / (ny - 1)
        self.x = np.linspace(domain[0], domain[1], nx)
        self.y = np.linspace(domain[2], domain[3], ny)
        self.sweeps = sweeps
        self.t = 0.0
        self.u = None

    # -- boundary conditions (heat_equation.py:499-599) -----------------------------------
    def _edge(self, loc):
        x, y = self.x, self.y
        return {"left": (np.s_[0, :], np.s_[1, :], x[0], y), "right": (np.s_[-1, :], np.s_[-2, :], x[-1], y),
                "bottom": (np.s_[:, 0], np.s_[:, 1], x, y[0]), "top": (np.s_[:, -1], np.s_[:, -2], x, y[-1])}[loc]

    def apply_bcs(self, u, t):
        h = min(self.hx, self.hy)
        for loc in ("left", "right", "bottom", "top"):
            bc = self.cfg.boundary_conditions.get(loc)
            if not bc:
                continue
            edge, inner, ex, ey = self._edge(loc)
            kind = bc.boundary_type.value
            if kind == "dirichlet":
                u[edge] = _eval(bc.evaluate, ex, ey, t)
            elif kind == "neumann":                                       # :548-562
                sign = -1.0 if loc in ("left", "bottom") else 1.0
                u[edge] = u[inner] + sign * (h * _eval(bc.evaluate, ex, ey, t))
            elif kind == "robin" and loc == "left":                       # :564-577
                u[edge] = (_eval(bc.evaluate, ex, ey, t) + bc.beta * u[inner] / h) / (bc.alpha + bc.beta / h)

    def apply_bcs_to_rhs(self, rhs, t):                                   # :579-599
        for loc, bc in self.cfg.boundary_conditions.items():
            if bc.boundary_type.value == "dirichlet":
                edge, _, ex, ey = self._edge(loc)
                rhs[edge] = _eval(bc.evaluate, ex, ey, t)

    # -- pieces ---------------------------------------------------------------------------
    def source(self, t):                                                  # :444-457
        if self.cfg.source_term is None:
            return np.zeros((self.nx, self.ny))
        return _eval(self.cfg.source_term, self.x[:, None], self.y[None, :], t)

    def set_initial_condition(self, u0=None):                             # :120-153
        if u0 is not None:
            self.u = np.array(u0, dtype=np.float64)
        elif self.cfg.initial_condition is not None:
            self.u = _eval(self.cfg.initial_condition, self.x[:, None], self.y[None, :])
        else:
            self.u = np.zeros((self.nx, self.ny))
        self.apply_bcs(self.u, 0.0)
        self.t = 0.0
        return self.u

    def solve_helmholtz(self, rhs, lam, guess):
        return helmholtz_gs(rhs, lam, guess, self.hx, self.hy, self.sweeps)[0]

    # -- steps (heat_equation.py:155-266) ---------------------------------------------------
    def step(self, u_old, dt, scheme):
        a = self.cfg.thermal_diffusivity
        if scheme == "explicit_euler":
            u = u_old + dt * (a * compute_laplacian(u_old, self.hx, self.hy) + self.source(self.t))
            self.apply_bcs(u, self.t + dt)
            return u
        if scheme == "implicit_euler":
            rhs = u_old + dt * self.source(self.t + dt)
            self.apply_bcs_to_rhs(rhs, self.t + dt)
            u = self.solve_helmholtz(rhs / (dt * a), 1.0 / (dt * a), u_old)
        elif scheme == "crank_nicolson":
            rhs = (u_old + dt * a * compute_laplacian(u_old, self.hx, self.hy) / 2
                   + dt * (self.source(self.t) + self.source(self.t + dt)) / 2)
            self.apply_bcs_to_rhs(rhs, self.t + dt)
            u = self.solve_helmholtz(2.0 * rhs / (dt * a), 2.0 / (dt * a), u_old)
        else:
            raise ValueError(f"Unsupported time stepping scheme: {scheme}")
        self.apply_bcs(u, self.t + dt)
        return u

    def adaptive_step(self, u_old, dt, tol, scheme):                      # :268-330
        u_full = u_old
        for _ in range(10):
            u_full = self.step(u_old, dt, scheme)
            u_half = self.step(self.step(u_old, dt / 2, scheme), dt / 2, scheme)
            if scheme in ("explicit_euler", "implicit_euler"):
                err, order = np.linalg.norm(u_half - u_full), 1
            else:
                err, order = np.linalg.norm(u_half - u_full) / 3.0, 2
            if err < tol:
                return u_half, dt
            dt = max(dt / 4, dt * 0.8 * (tol / err) ** (1 / (order + 1)))
        return u_full, dt

    def solve_time_dependent(self, t_final, dt, scheme, adaptive=True, tol=1e-4):   # :332-417
        times, dts, steps = [0.0], [], 0
        while self.t < t_final:
            if self.t + dt > t_final:
                dt = t_final - self.t
            if adaptive and scheme != "explicit_euler":
                self.u, dt = self.adaptive_step(self.u, dt, tol, scheme)
            else:
                self.u = self.step(self.u, dt, scheme)
            self.t += dt
            steps += 1
            times.append(self.t)
            dts.append(dt)
        return {"final_solution": self.u, "time_history": times, "dt_history": dts, "total_steps": steps}
