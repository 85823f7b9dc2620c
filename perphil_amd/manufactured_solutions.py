"""
Manufactured pressures — mirror of reference ``src/perphil/utils/manufactured_solutions.py``:
``exact_expressions`` (:7-53, 2D) and ``exact_expressions_3d`` (:56-94).  The "expressions" are
callables ``f(points[n, dim]) -> values[n]`` (NumPy, fp64); they supply the Dirichlet data of every
benchmark configuration.  Velocities are returned as callables to ``[n, dim]`` arrays.
"""
from __future__ import annotations

import math
from typing import Callable, Tuple

import numpy as np

from . import fd
from .parameters import DPPParameters

PointFn = Callable[[np.ndarray], np.ndarray]


class MMSPressure:
    """Callable manufactured pressure p1 (field 0) or p2 (field 1); carries its parameters so that the
    device error-norm kernel (``pph_error_norms_mms``) can evaluate the same closed form."""

    def __init__(self, fn: PointFn, field: int, dim: int, k1: float, k2: float, beta: float, mu: float):
        self._fn, self.field, self.dim = fn, field, dim
        self.k1, self.k2, self.beta, self.mu = k1, k2, beta, mu

    @property
    def params(self):
        """What the values depend on (fd.DirichletBC keys its cache of evaluated boundary data on it)."""
        return (self.field, self.dim, self.k1, self.k2, self.beta, self.mu)

    def __call__(self, X: np.ndarray) -> np.ndarray:
        return self._fn(X)


def exact_expressions(mesh: fd.Mesh, dpp_params: DPPParameters) -> Tuple[PointFn, PointFn, PointFn, PointFn]:
    """(u1, p1, u2, p2) on the unit square (manufactured_solutions.py:39-51)."""
    k1, k2 = float(dpp_params.k1), float(dpp_params.k2)
    beta, mu, eta = float(dpp_params.beta), float(dpp_params.mu), float(dpp_params.eta)
    pi = math.pi

    def p1(X):
        x, y = X[:, 0], X[:, 1]
        return (mu / pi) * np.exp(pi * x) * np.sin(pi * y) - (mu / (beta * k1)) * np.exp(eta * y)

    def p2(X):
        x, y = X[:, 0], X[:, 1]
        return (mu / pi) * np.exp(pi * x) * np.sin(pi * y) + (mu / (beta * k2)) * np.exp(eta * y)

    def u1(X):
        x, y = X[:, 0], X[:, 1]
        return np.stack([-k1 * np.exp(pi * x) * np.sin(pi * y),
                         -k1 * (np.exp(pi * x) * np.cos(pi * y) - (eta / (beta * k1)) * np.exp(eta * y))], axis=1)

    def u2(X):
        x, y = X[:, 0], X[:, 1]
        return np.stack([-k2 * np.exp(pi * x) * np.sin(pi * y),
                         -k2 * (np.exp(pi * x) * np.cos(pi * y) + (eta / (beta * k2)) * np.exp(eta * y))], axis=1)

    return u1, MMSPressure(p1, 0, 2, k1, k2, beta, mu), u2, MMSPressure(p2, 1, 2, k1, k2, beta, mu)


def exact_expressions_3d(mesh: fd.Mesh, dpp_params: DPPParameters) -> Tuple[PointFn, PointFn, PointFn, PointFn]:
    """(u1, p1, u2, p2) on the unit cube (manufactured_solutions.py:87-94)."""
    k1, k2 = float(dpp_params.k1), float(dpp_params.k2)
    beta, mu, eta = float(dpp_params.beta), float(dpp_params.mu), float(dpp_params.eta)
    pi = math.pi

    def _p(X, sign, k):
        x, y, z = X[:, 0], X[:, 1], X[:, 2]
        return (mu / pi) * np.exp(pi * x) * (np.sin(pi * y) + np.sin(pi * z)) + sign * (mu / (beta * k)) * (
            np.exp(eta * y) + np.exp(eta * z))

    def _grad(X, sign, k):
        x, y, z = X[:, 0], X[:, 1], X[:, 2]
        ex = np.exp(pi * x)
        c = sign * (mu / (beta * k)) * eta
        return np.stack([mu * ex * (np.sin(pi * y) + np.sin(pi * z)),
                         mu * ex * np.cos(pi * y) + c * np.exp(eta * y),
                         mu * ex * np.cos(pi * z) + c * np.exp(eta * z)], axis=1)

    p1 = MMSPressure(lambda X: _p(X, -1.0, k1), 0, 3, k1, k2, beta, mu)
    p2 = MMSPressure(lambda X: _p(X, +1.0, k2), 1, 3, k1, k2, beta, mu)
    u1 = lambda X: -(k1 / mu) * _grad(X, -1.0, k1)
    u2 = lambda X: -(k2 / mu) * _grad(X, +1.0, k2)
    return u1, p1, u2, p2


def interpolate_exact(mesh: fd.Mesh, velocity_space, pressure_space, dpp_params: DPPParameters):
    """Nodal interpolants (u1, p1, u2, p2) (manufactured_solutions.py:97-135); velocities are
    returned as ``[n, dim]`` arrays, pressures as ``Function``s."""
    ex = exact_expressions if mesh.dim == 2 else exact_expressions_3d
    u1, p1, u2, p2 = ex(mesh, dpp_params)
    X = mesh.local_node_coordinates()
    return (u1(X), fd.Function(pressure_space, p1(X), name="p1_exact"),
            u2(X), fd.Function(pressure_space, p2(X), name="p2_exact"))
