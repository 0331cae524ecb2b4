"""Implicit time integrators of the reference's transient solves, on device-resident vectors.

Mirrors `time_integrator/ImplicitTimeIntegrator.cpp`, `ImplicitEuler.cpp:5-33`, `ImplicitNewmark.cpp:5-41` and
`BDF.cpp:8-160`: an integrator keeps the previous positions / velocities / accelerations, offers the predictor
`x_tilde()` and `acceleration_scaling()` that the inertia term and the weights of the other forms are built from
(`state/StateSolveNonlinear.cpp:78-84`, `solver/forms/InertiaForm.cpp`), and `update_quantities(x)` after the solve.
The vectors are torch tensors (they stay on the GPU between steps) or numpy arrays; the arithmetic is the reference's,
expression by expression.

`TransientSolver` is the step loop of `State::solve_transient_tensor_nonlinear` (`state/StateSolveNonlinear.cpp:22-46`)
for one engine: per step it hands x_tilde and the scaling to the engine's inertia term and runs the device-resident
Newton solve starting at the previous solution.
"""
from __future__ import annotations

from collections import deque


def _copy(a):
    return a.clone() if hasattr(a, "clone") else a.copy()

BDF_ALPHAS = (
    (1.0,),
    (4.0 / 3.0, -1.0 / 3.0),
    (18.0 / 11.0, -9.0 / 11.0, 2.0 / 11.0),
    (48.0 / 25.0, -36.0 / 25.0, 16.0 / 25.0, -3.0 / 25.0),
    (300.0 / 137.0, -300.0 / 137.0, 200.0 / 137.0, -75.0 / 137.0, 12.0 / 137.0),
    (360.0 / 147.0, -450.0 / 147.0, 400.0 / 147.0, -225.0 / 147.0, 72.0 / 147.0, -10.0 / 147.0),
)
BDF_BETAS = (1.0, 2.0 / 3.0, 6.0 / 11.0, 12.0 / 25.0, 60.0 / 137.0, 60.0 / 147.0)


class ImplicitTimeIntegrator:
    """Common state: the newest previous quantities first, as the reference's deques."""

    def __init__(self):
        self.x_prevs, self.v_prevs, self.a_prevs = deque(), deque(), deque()
        self._dt = 0.0

    def init(self, x_prev, v_prev, a_prev, dt: float):
        if not dt > 0:
            raise ValueError("dt must be positive")
        self.x_prevs, self.v_prevs, self.a_prevs = deque([_copy(x_prev)]), deque([_copy(v_prev)]), deque([_copy(a_prev)])
        self._dt = float(dt)

    def dt(self) -> float:
        return self._dt

    def x_prev(self):
        return self.x_prevs[0]

    def v_prev(self):
        return self.v_prevs[0]

    def a_prev(self):
        return self.a_prevs[0]

    def update_quantities(self, x):
        v = self.compute_velocity(x)
        a = self.compute_acceleration(v)
        self.x_prevs[0], self.v_prevs[0], self.a_prevs[0] = _copy(x), v, a


class ImplicitEuler(ImplicitTimeIntegrator):
    def x_tilde(self):
        return self.x_prev() + self.dt() * self.v_prev()

    def compute_velocity(self, x):
        return (x - self.x_prev()) / self.dt()

    def compute_acceleration(self, v):
        return (v - self.v_prev()) / self.dt()

    def acceleration_scaling(self) -> float:
        return self.dt() * self.dt()


class ImplicitNewmark(ImplicitTimeIntegrator):
    def __init__(self, gamma: float = 0.5, beta: float = 0.25):
        super().__init__()
        self.gamma, self.beta = float(gamma), float(beta)

    def x_tilde(self):
        dt = self.dt()
        return self.x_prev() + dt * (self.v_prev() + dt * (0.5 - self.beta) * self.a_prev())

    def compute_velocity(self, x):
        c, dt = self.gamma / self.beta, self.dt()
        return c / dt * (x - self.x_prev()) + (1 - c) * self.v_prev() + (1 - c / 2) * dt * self.a_prev()

    def compute_acceleration(self, v):
        dt = self.dt()
        return (v - self.v_prev() - (1 - self.gamma) * dt * self.a_prev()) / (self.gamma * dt)

    def acceleration_scaling(self) -> float:
        return self.beta * self.dt() * self.dt()


class BDF(ImplicitTimeIntegrator):
    """Backward differentiation formulas of order 1..6; the history grows until `steps` previous states are known."""

    def __init__(self, steps: int = 1):
        super().__init__()
        self.steps = steps if 1 <= steps <= 6 else 1  # the reference warns and falls back to 1

    def _weighted(self, prevs):
        alpha = BDF_ALPHAS[len(prevs) - 1]
        s = alpha[0] * prevs[0]
        for i in range(1, len(prevs)):
            s = s + alpha[i] * prevs[i]
        return s

    def _beta(self) -> float:
        return BDF_BETAS[len(self.x_prevs) - 1]

    def x_tilde(self):
        return self._weighted(self.x_prevs) + self._beta() * self.dt() * self._weighted(self.v_prevs)

    def compute_velocity(self, x):
        return (x - self._weighted(self.x_prevs)) / (self._beta() * self.dt())

    def compute_acceleration(self, v):
        return (v - self._weighted(self.v_prevs)) / (self._beta() * self.dt())

    def acceleration_scaling(self) -> float:
        b = self._beta()
        return b * b * self.dt() * self.dt()

    def update_quantities(self, x):
        v = self.compute_velocity(x)
        a = self.compute_acceleration(v)
        self.x_prevs.appendleft(_copy(x))
        self.v_prevs.appendleft(v)
        self.a_prevs.appendleft(a)
        if len(self.x_prevs) > self.steps:
            self.x_prevs.pop(), self.v_prevs.pop(), self.a_prevs.pop()


def construct_time_integrator(params):
    """ImplicitTimeIntegrator::construct_time_integrator: a name or {"type": ..., parameters} (the reference also
    accepts its historical spellings)."""
    kind = params["type"] if isinstance(params, dict) else params
    if kind in ("implict_euler", "ImplicitEuler"):
        return ImplicitEuler()
    if kind in ("implict_newmark", "ImplicitNewmark"):
        p = params if isinstance(params, dict) else {}
        return ImplicitNewmark(p.get("gamma", 0.5), p.get("beta", 0.25))
    if kind == "BDF":
        return BDF(params.get("steps", 1) if isinstance(params, dict) else 1)
    raise RuntimeError(f"Unknown time integrator ({kind})")


class TransientSolver:
    """Device-resident transient nonlinear solve: engine (mesh, material, Dirichlet dofs set; inertia_set_mass called),
    integrator (initialised), rhs = body-force vector on the device or None."""

    def __init__(self, engine, integrator, rhs=None, **newton_params):
        self.engine, self.integrator, self.rhs, self.newton_params = engine, integrator, rhs, newton_params

    def step(self):
        """One time step; returns (x, Newton info).  x is a new tensor, the integrator's history is updated."""
        import torch
        ti, eng = self.integrator, self.engine
        x_tilde = ti.x_tilde().contiguous()
        x = ti.x_prev().clone()  # the previous solution is the initial guess
        torch.cuda.current_stream().synchronize()  # the engine enqueues on its own stream
        eng.inertia_enable(ti.acceleration_scaling(), x_tilde.data_ptr())
        try:
            info = eng.newton_solve_dev(x.data_ptr(), self.rhs.data_ptr() if self.rhs is not None else 0, **self.newton_params)
        finally:
            eng.inertia_disable()
        ti.update_quantities(x)
        return x, info
