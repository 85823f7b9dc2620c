"""
Minimal stand-ins for the Firedrake objects that cross the reference's ``solve_dpp`` boundary
(``import firedrake as fd`` in reference ``src/perphil/solvers/solver.py:3``): ``Constant``,
``UnitSquareMesh`` / ``UnitCubeMesh``, ``FunctionSpace`` / ``VectorFunctionSpace`` /
``MixedFunctionSpace``, ``Function``, ``DirichletBC``.  They hold *descriptions* (sizes, kinds,
boundary data); all arithmetic happens in the HIP library behind ``perphil_amd._ffi``.

Numbering (documented, differs from Firedrake's DMPlex numbering which cannot be reproduced):
node (i,j,k) -> i + (nx+1)*(j + (ny+1)*k); mixed dof = field*n + node (field-major, as pinned by
reference ``src/perphil/experiments/iterative_bench.py:323-324``).
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, Optional, Sequence, Tuple, Union

import numpy as np

pi = math.pi

CELL_QUAD, CELL_TRI, CELL_HEX, CELL_TET = 0, 1, 2, 3


class Constant:
    """Scalar constant (stand-in for ``fd.Constant``); ``float(c)`` works, arithmetic yields Constants."""

    ufl_shape = ()

    def __init__(self, value):
        self._v = float(value)

    def __float__(self):
        return self._v

    def values(self):
        return np.array([self._v])

    def assign(self, value):
        self._v = float(value)
        return self

    def _b(self, other, op):
        return Constant(op(self._v, float(other)))

    def __add__(self, o): return self._b(o, lambda a, b: a + b)
    def __radd__(self, o): return self._b(o, lambda a, b: b + a)
    def __sub__(self, o): return self._b(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._b(o, lambda a, b: b - a)
    def __mul__(self, o): return self._b(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._b(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._b(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._b(o, lambda a, b: b / a)
    def __neg__(self): return Constant(-self._v)
    def __pow__(self, o): return self._b(o, lambda a, b: a ** b)

    def __repr__(self):
        return f"Constant({self._v!r})"


def sqrt(x):
    return Constant(math.sqrt(float(x))) if isinstance(x, Constant) else math.sqrt(x)


class Mesh:
    """Structured unit square / unit cube; lexicographic vertex numbering."""

    def __init__(self, dim: int, kind: int, nx: int, ny: int, nz: int = 0):
        if dim not in (2, 3):
            raise ValueError("dim must be 2 or 3")
        if min(nx, ny) < 1 or (dim == 3 and nz < 1):
            raise ValueError("need at least one cell per direction")
        self.dim, self.kind, self.nx, self.ny, self.nz = dim, kind, int(nx), int(ny), int(nz)
        self._ctx = None  # device context, created on first solve

    # -- sizes ----------------------------------------------------------------------------------
    @property
    def node_dims(self) -> Tuple[int, int, int]:
        return self.nx + 1, self.ny + 1, (self.nz + 1 if self.dim == 3 else 1)

    def num_vertices(self) -> int:
        px, py, pz = self.node_dims
        return px * py * pz

    def num_cells(self) -> int:
        boxes = self.nx * self.ny * (self.nz if self.dim == 3 else 1)
        return boxes * {CELL_QUAD: 1, CELL_TRI: 2, CELL_HEX: 1, CELL_TET: 6}[self.kind]

    def geometric_dimension(self) -> int:
        return self.dim

    # -- geometry (host side, closed form; only boundary coordinates are needed by the hot path) --
    def node_coordinates(self, nodes: Optional[np.ndarray] = None) -> np.ndarray:
        px, py, _ = self.node_dims
        ids = np.arange(self.num_vertices(), dtype=np.int64) if nodes is None else np.asarray(nodes, dtype=np.int64)
        i, j = ids % px, (ids // px) % py
        cols = [i / self.nx, j / self.ny]
        if self.dim == 3:
            cols.append((ids // (px * py)) / self.nz)
        return np.stack(cols, axis=1).astype(np.float64)

    def boundary_nodes(self) -> np.ndarray:
        """Sorted vertex ids with the "on_boundary" marker."""
        px, py, pz = self.node_dims
        if self.dim == 2:
            m = np.zeros((py, px), dtype=bool)
            m[0, :] = m[-1, :] = True
            m[:, 0] = m[:, -1] = True
        else:
            m = np.zeros((pz, py, px), dtype=bool)
            m[0] = m[-1] = True
            m[:, 0, :] = m[:, -1, :] = True
            m[:, :, 0] = m[:, :, -1] = True
        return np.nonzero(m.ravel())[0].astype(np.int64)

    def context(self, device: int = 0):
        """Device context holding this mesh (cell->dof map, CSR pattern, K and M are cached there)."""
        from . import _ffi

        if self._ctx is None or self._ctx.device != device:
            ctx = _ffi.Context(device)
            ctx.mesh_build(self.dim, self.kind, self.nx, self.ny, self.nz)
            self._ctx = ctx
        return self._ctx


def UnitSquareMesh(nx: int, ny: int, quadrilateral: bool = False, **_ignored) -> Mesh:
    """``fd.UnitSquareMesh``: quads, or triangles with the "left" diagonal (Firedrake's default)."""
    return Mesh(2, CELL_QUAD if quadrilateral else CELL_TRI, nx, ny)


def UnitCubeMesh(nx: int, ny: int, nz: int, hexahedral: bool = False, **_ignored) -> Mesh:
    """``fd.UnitCubeMesh``: hexes, or six Kuhn tetrahedra per cube."""
    return Mesh(3, CELL_HEX if hexahedral else CELL_TET, nx, ny, nz)


class FunctionSpace:
    """CG-1 scalar space: one dof per mesh vertex."""

    def __init__(self, mesh: Mesh, family: str = "CG", degree: int = 1, name: Optional[str] = None):
        if family not in ("CG", "Lagrange", "P", "Q") or degree != 1:
            raise NotImplementedError("the MI355X path implements the conforming CG-1 pressure space only")
        self._mesh, self.family, self.degree, self.name = mesh, "CG", 1, name
        self.index: Optional[int] = None
        self.parent: Optional["MixedFunctionSpace"] = None

    def mesh(self) -> Mesh:
        return self._mesh

    def dim(self) -> int:
        return self._mesh.num_vertices()

    def num_sub_spaces(self) -> int:
        return 1

    def __mul__(self, other: "FunctionSpace") -> "MixedFunctionSpace":
        return MixedFunctionSpace((self, other))


class VectorFunctionSpace(FunctionSpace):
    """CG-1 vector space (velocity space U of ``create_function_spaces``; not on the hot path)."""

    def __init__(self, mesh: Mesh, family: str = "CG", degree: int = 1, name: Optional[str] = None):
        super().__init__(mesh, family, degree, name)
        self.value_size = mesh.dim

    def dim(self) -> int:
        return self._mesh.num_vertices() * self.value_size


class _IndexedSubSpace(FunctionSpace):
    def __init__(self, parent: "MixedFunctionSpace", index: int, base: FunctionSpace):
        super().__init__(base.mesh(), base.family, base.degree, base.name)
        self.parent, self.index = parent, index


class MixedFunctionSpace:
    """W = V x V; dofs field-major."""

    def __init__(self, spaces: Sequence[FunctionSpace]):
        spaces = tuple(spaces)
        if len(spaces) < 1:
            raise ValueError("need at least one sub space")
        m = spaces[0].mesh()
        if any(s.mesh() is not m for s in spaces):
            raise ValueError("all sub spaces must live on the same mesh")
        self._subs = tuple(_IndexedSubSpace(self, i, s) for i, s in enumerate(spaces))

    def num_sub_spaces(self) -> int:
        return len(self._subs)

    def sub(self, i: int) -> FunctionSpace:
        return self._subs[i]

    def __iter__(self):
        return iter(self._subs)

    def mesh(self) -> Mesh:
        return self._subs[0].mesh()

    def dim(self) -> int:
        return sum(s.dim() for s in self._subs)


class _Dat:
    def __init__(self, arr: np.ndarray):
        self.data = arr

    @property
    def data_ro(self):
        return self.data


class Function:
    """Nodal coefficient vector on a (mixed) space; ``sub(i)`` / ``subfunctions`` are views."""

    def __init__(self, space, val: Optional[np.ndarray] = None, name: Optional[str] = None):
        self._space, self.name = space, name
        n = space.dim()
        if val is None:
            val = np.zeros(n, dtype=np.float64)
        if val.shape != (n,):
            raise ValueError(f"expected {n} coefficients, got {val.shape}")
        self._val = val
        self.dat = _Dat(self._val)

    def function_space(self):
        return self._space

    def vector(self) -> np.ndarray:
        return self._val

    def sub(self, i: int) -> "Function":
        if not isinstance(self._space, MixedFunctionSpace):
            raise IndexError("not a mixed function")
        off = sum(self._space.sub(k).dim() for k in range(i))
        V = self._space.sub(i)
        return Function(V, self._val[off:off + V.dim()], name=f"{self.name or 'w'}[{i}]")

    @property
    def subfunctions(self) -> Tuple["Function", ...]:
        return tuple(self.sub(i) for i in range(self._space.num_sub_spaces()))

    def split(self) -> Tuple["Function", ...]:
        return self.subfunctions

    def assign(self, other) -> "Function":
        self._val[:] = other._val if isinstance(other, Function) else float(other)
        return self

    def interpolate(self, expr) -> "Function":
        self._val[:] = evaluate(expr, self._space.mesh(), None)
        return self

    def at(self, point: Sequence[float]) -> float:
        """Value at a point that coincides with a mesh vertex (what ``slice_along_x`` needs)."""
        mesh = self._space.mesh()
        dims = (mesh.nx, mesh.ny, mesh.nz)[: mesh.dim]
        idx = []
        for c, nc in zip(point, dims):
            t = c * nc
            if abs(t - round(t)) > 1e-9:
                raise NotImplementedError("Function.at is available at mesh vertices only")
            idx.append(int(round(t)))
        px, py, _ = mesh.node_dims
        node = idx[0] + px * (idx[1] + (py * idx[2] if mesh.dim == 3 else 0))
        return float(self._val[node])


Expr = Union[float, Constant, np.ndarray, Callable[[np.ndarray], np.ndarray], Function]


def evaluate(expr: Expr, mesh: Mesh, nodes: Optional[np.ndarray]) -> np.ndarray:
    """Values of a boundary/initial datum at mesh vertices ``nodes`` (all vertices if None)."""
    count = mesh.num_vertices() if nodes is None else len(nodes)
    if isinstance(expr, Function):
        return expr.vector() if nodes is None else expr.vector()[nodes]
    if isinstance(expr, (int, float, Constant)):
        return np.full(count, float(expr))
    if isinstance(expr, np.ndarray):
        if expr.shape != (mesh.num_vertices(),):
            raise ValueError("nodal array must have one value per mesh vertex")
        return expr if nodes is None else expr[nodes]
    if callable(expr):
        vals = np.asarray(expr(mesh.node_coordinates(nodes)), dtype=np.float64)
        if vals.shape != (count,):
            raise ValueError("expression must return one value per point")
        return vals
    raise TypeError(f"cannot evaluate boundary datum of type {type(expr)}")


class DirichletBC:
    """``fd.DirichletBC(W.sub(i), value, "on_boundary")``."""

    def __init__(self, V: FunctionSpace, value: Expr, sub_domain="on_boundary"):
        if sub_domain != "on_boundary":
            raise NotImplementedError('only the "on_boundary" marker is supported')
        self._V, self.value, self.sub_domain = V, value, sub_domain

    def function_space(self) -> FunctionSpace:
        return self._V

    @property
    def field(self) -> int:
        return 0 if self._V.index is None else int(self._V.index)

    def nodes_and_values(self) -> Tuple[np.ndarray, np.ndarray]:
        """Boundary nodes and their values.  A callable datum (the manufactured pressures) is evaluated once per
        condition object: every solve with the same conditions used to re-evaluate exp / sin at 394 k boundary nodes
        (256^3: ~20 ms per field and call).  Constants, arrays and Functions are read afresh (they can be reassigned)."""
        mesh = self._V.mesh()
        if callable(self.value) and not isinstance(self.value, (Function, Constant)):
            if getattr(self, "_cache", None) is None:
                nodes = mesh.boundary_nodes()
                self._cache = (nodes, evaluate(self.value, mesh, nodes))
            return self._cache
        nodes = mesh.boundary_nodes()
        return nodes, evaluate(self.value, mesh, nodes)
