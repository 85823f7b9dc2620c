"""``create_mesh`` — mirror of reference ``src/perphil/mesh/builtin.py:4-20``."""
from __future__ import annotations

from . import fd


def create_mesh(num_x: int, num_y: int, quadrilateral: bool = True) -> fd.Mesh:
    """2D unit-square mesh: quads (default) or "left"-diagonal triangles."""
    return fd.UnitSquareMesh(num_x, num_y, quadrilateral=quadrilateral)
