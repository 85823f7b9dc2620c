"""
Post-processing — mirror of reference ``src/perphil/utils/postprocessing.py``:
``split_dpp_solution`` (:6-31), ``calculate_darcy_velocity_from_pressure`` (:34-63), ``slice_along_x`` (:66-86), ``l2_error`` (:89-105), ``h1_seminorm_error``
(:108-124).  The two error norms run on the device: a Gauss rule per cell on the isoparametric map, with the
manufactured pressure evaluated in closed form at every quadrature point (``pph_error_norms_mms``) or, for any other
exact field, with samples the caller's callable provides at those points (``pph_quadrature_points`` /
``pph_error_norms_sampled``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import fd
from .manufactured_solutions import MMSPressure


def split_dpp_solution(dpp_solution: fd.Function) -> Tuple[fd.Function, fd.Function]:
    """(p1_h, p2_h) as independent Functions; ValueError unless the space is a 2-field mixed space."""
    W = dpp_solution.function_space()
    if not hasattr(W, "num_sub_spaces") or W.num_sub_spaces() != 2:
        raise ValueError(f"Expected a 2-field MixedFunctionSpace, got {type(W)}")
    p1 = fd.Function(W.sub(0), dpp_solution.sub(0).vector().copy(), name="p1_h")
    p2 = fd.Function(W.sub(1), dpp_solution.sub(1).vector().copy(), name="p2_h")
    return p1, p2


def calculate_darcy_velocity_from_pressure(pressure_field: fd.Function, conductivity,
                                           velocity_space: Optional[fd.FunctionSpace] = None,
                                           degree: int = 1) -> fd.Function:
    """
    u = -conductivity * grad(p_h), L2-projected onto the CG-1 vector space (``pph_darcy_velocity``:
    node-centred right-hand side kernel + one mass-matrix CG solve per component on the device).
    Coefficients are node-major: ``u.vector().reshape(-1, dim)[node]`` is the velocity at a vertex.
    """
    mesh = pressure_field.function_space().mesh()
    if mesh.distributed:
        pressure_field = pressure_field.gather()      # (collective) the projection runs on the serial twin of the mesh
        mesh, velocity_space = pressure_field.function_space().mesh(), None
    if velocity_space is None:
        velocity_space = fd.VectorFunctionSpace(mesh, "CG", degree)
    if velocity_space.degree != 1 or velocity_space.mesh() is not mesh:
        raise NotImplementedError("the velocity space must be the CG-1 vector space of the pressure's mesh")
    if not isinstance(conductivity, (int, float, fd.Constant)):
        raise NotImplementedError("conductivity must be a constant")
    u = mesh.context().darcy_velocity(pressure_field.vector(), float(conductivity))
    return fd.Function(velocity_space, u.reshape(-1), name="velocity")


def slice_along_x(scalar_field: fd.Function, x_value: float) -> Tuple[np.ndarray, np.ndarray]:
    """(y_points, values) of a scalar field along the vertical line x = x_value (a grid line)."""
    mesh = scalar_field.function_space().mesh()
    if mesh.dim != 2:
        raise NotImplementedError("slice_along_x is a 2D utility (as in the reference)")
    y_points = np.arange(mesh.ny + 1) / mesh.ny
    values = np.array([scalar_field.at((x_value, y)) for y in y_points])
    return y_points, values


def _norms(numerical: fd.Function, exact_expr, quadrature_points: int):
    """Both norms on the device.  `exact_expr` may be what the reference's UFL argument can be: a manufactured pressure
    (closed form evaluated in the kernel), any callable of point arrays `f(X[m, dim]) -> [m]` (optionally with a
    `.grad(X) -> [m, dim]` attribute; central differences otherwise), a CG-1 `Function` on the same mesh, or a
    `Constant` / number."""
    mesh = numerical.function_space().mesh()
    if mesh.distributed:
        # post-processing is not on the sharded path: the whole function on the serial twin of the mesh (collective)
        numerical = numerical.gather()
        if isinstance(exact_expr, fd.Function):
            exact_expr = exact_expr.gather()
        mesh = numerical.function_space().mesh()
    ctx = mesh.context()
    if isinstance(exact_expr, MMSPressure):
        if exact_expr.dim != mesh.dim:
            raise ValueError("exact expression and mesh have different dimensions")
        e = exact_expr
        return ctx.error_norms_mms(e.field, numerical.vector(), e.k1, e.k2, e.beta, e.mu, quadrature_points)
    if isinstance(exact_expr, fd.Function):
        if exact_expr.function_space().mesh() is not mesh:
            raise ValueError("both functions must live on the same mesh")
        diff = np.asarray(numerical.vector(), dtype=np.float64) - np.asarray(exact_expr.vector(), dtype=np.float64)
        return ctx.error_norms_sampled(diff, None, None, min(quadrature_points, 3))   # CG-1 difference: exact with 2-3 points
    if isinstance(exact_expr, (int, float, fd.Constant)):
        c = float(exact_expr)
        return ctx.error_norms_sampled(numerical.vector(), lambda X: np.full(X.shape[0], c), lambda X: np.zeros_like(X),
                                       quadrature_points)
    if callable(exact_expr):
        return ctx.error_norms_sampled(numerical.vector(), exact_expr, getattr(exact_expr, "grad", None), quadrature_points)
    raise TypeError(f"cannot evaluate an exact expression of type {type(exact_expr).__name__}")


def l2_error(numerical: fd.Function, exact_expr, quadrature_points: int = 6) -> float:
    """||numerical - exact||_{L2} (reference postprocessing.py:89-105)."""
    return float(_norms(numerical, exact_expr, quadrature_points)[0])


def h1_seminorm_error(numerical: fd.Function, exact_expr, quadrature_points: int = 6) -> float:
    """|numerical - exact|_{H1} (reference postprocessing.py:108-124)."""
    return float(_norms(numerical, exact_expr, quadrature_points)[1])
