"""``create_function_spaces`` — mirror of reference ``src/perphil/forms/spaces.py:5-36``."""
from __future__ import annotations

from typing import Tuple

from . import fd


def create_function_spaces(
    mesh: fd.Mesh,
    velocity_deg: int = 1,
    pressure_deg: int = 1,
    velocity_family: str = "CG",
    pressure_family: str = "CG",
) -> Tuple[fd.VectorFunctionSpace, fd.FunctionSpace]:
    """(U, V): CG vector space for velocities, CG scalar space for the pressures."""
    U = fd.VectorFunctionSpace(mesh, velocity_family, velocity_deg)
    V = fd.FunctionSpace(mesh, pressure_family, pressure_deg)
    return U, V
