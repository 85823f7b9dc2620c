"""
ORACLE — test infrastructure only.  NOT part of the product path.

CPU (NumPy/SciPy) restatement of the geometric-multigrid preconditioner that the HIP path uses for
the scalar blocks A11 = (k1/mu)K + (beta/mu)M and A22 = (k2/mu)K + (beta/mu)M in place of the LU
block solves of the reference's field-split / Picard configurations
(reference ``src/perphil/solvers/parameters.py:30-37`` FIELDSPLIT_LU_PARAMS, ``:79-85``
PICARD_LU_SOLVER_PARAMS; block operators from ``src/perphil/forms/dpp.py:196-203``).

The reference holds no golden for a multigrid cycle (it uses MUMPS LU there): this file pins only
the HIP kernels' arithmetic (same hierarchy, same Chebyshev smoother, same transfer weights), while
the *result* of a multigrid-preconditioned solve is pinned against the direct solution of
``dpp_oracle`` (which is itself pinned by the reference's goldens).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.sparse as sp

from . import dpp_oracle as o


def _stencil(kind: int):
    """Offsets (dx,dy,dz) of the sparsity stencil of a cell kind, ascending (dz,dy,dx)."""
    out = []
    if kind == o.CELL_QUAD:
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                out.append((dx, dy, 0))
    elif kind == o.CELL_TRI:
        out = [(0, -1, 0), (1, -1, 0), (-1, 0, 0), (0, 0, 0), (1, 0, 0), (-1, 1, 0), (0, 1, 0)]
    elif kind == o.CELL_HEX:
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    out.append((dx, dy, dz))
    else:
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if (dx >= 0 and dy >= 0 and dz >= 0) or (dx <= 0 and dy <= 0 and dz <= 0):
                        out.append((dx, dy, dz))
    return out


def transfer_weight(kind: int, d) -> float:
    """Weight of fine node 2C+d in the restriction to coarse node C (= prolongation weight)."""
    nz = sum(1 for t in d if t != 0)
    if kind in (o.CELL_QUAD, o.CELL_HEX):
        return 0.5 ** nz
    return 1.0 if nz == 0 else 0.5


def prolongation(kind: int, cdims, fdims) -> sp.csr_matrix:
    """P (n_fine x n_coarse): multilinear interpolation for Q1, edge-midpoint averaging for the
    P1 Kuhn / left-diagonal triangulations (their refinement is again Kuhn / left-diagonal)."""
    pxc, pyc, pzc = cdims
    pxf, pyf, pzf = fdims
    rows, cols, vals = [], [], []
    K, J, I = np.meshgrid(np.arange(pzc), np.arange(pyc), np.arange(pxc), indexing="ij")
    C = (I + pxc * (J + pyc * K)).ravel()
    I, J, K = I.ravel(), J.ravel(), K.ravel()
    for d in _stencil(kind):
        fi, fj, fk = 2 * I + d[0], 2 * J + d[1], 2 * K + d[2]
        ok = (fi >= 0) & (fi < pxf) & (fj >= 0) & (fj < pyf) & (fk >= 0) & (fk < pzf)
        rows.append((fi + pxf * (fj + pyf * fk))[ok])
        cols.append(C[ok])
        vals.append(np.full(int(ok.sum()), transfer_weight(kind, d)))
    P = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(pxf * pyf * pzf, pxc * pyc * pzc)).tocsr()
    return P


@dataclass
class Level:
    A: sp.csr_matrix
    dinv: np.ndarray
    mask: np.ndarray  # True on constrained dofs
    lam: float  # upper bound of the spectrum of D^-1 A (max abs row sum)
    P: Optional[sp.csr_matrix] = None  # to this level from the next coarser one


def eliminate(A: sp.csr_matrix, mask: np.ndarray) -> sp.csr_matrix:
    return o.eliminate_dirichlet(A, np.nonzero(mask)[0])


def build_hierarchy(dim: int, kind: int, nx: int, ny: int, nz: int, coefK: float, coefM: float,
                    mask_fine: np.ndarray, min_cells: int = 2) -> List[Level]:
    """Rediscretised operators coefK*K + coefM*M on nx/2^l meshes, Dirichlet mask injected."""
    levels: List[Level] = []
    mask = mask_fine.astype(bool)
    dims_prev = None
    while True:
        mesh = o.build_mesh(dim, kind, nx, ny, nz)
        K, M = o.assemble_scalar(mesh)
        A = eliminate((coefK * K + coefM * M).tocsr(), mask)
        d = A.diagonal()
        lam = float(np.max(np.asarray(abs(A).sum(axis=1)).ravel() / d))
        lv = Level(A, 1.0 / d, mask.copy(), lam)
        dims = (nx + 1, ny + 1, (nz + 1) if dim == 3 else 1)
        if dims_prev is not None:
            levels[-1].P = prolongation(kind, dims, dims_prev)
        levels.append(lv)
        can = nx % 2 == 0 and ny % 2 == 0 and (dim == 2 or nz % 2 == 0)
        can = can and min(nx, ny, nz if dim == 3 else nx) // 2 >= min_cells
        if not can:
            break
        # inject the mask: coarse node C <- fine node 2C
        px, py = nx + 1, ny + 1
        pz = nz + 1 if dim == 3 else 1
        m3 = mask.reshape(pz, py, px)
        mask = (m3[::2, ::2, ::2] if dim == 3 else m3[:, ::2, ::2]).ravel().copy()
        dims_prev = dims
        nx, ny = nx // 2, ny // 2
        if dim == 3:
            nz //= 2
    return levels


CHEB_LOWER = 0.25  # smoother targets [CHEB_LOWER*lam, lam]


def chebyshev(lv: Level, b: np.ndarray, x: Optional[np.ndarray], steps: int) -> np.ndarray:
    """`steps` Chebyshev-Jacobi steps on A x = b; x=None means zero initial guess."""
    lo, hi = CHEB_LOWER * lv.lam, lv.lam
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma = theta / delta
    rho = 1.0 / sigma
    if x is None:
        x = np.zeros_like(b)
        r = b.copy()
    else:
        r = b - lv.A @ x
    d = lv.dinv * r / theta
    x = x + d
    for _ in range(1, steps):
        r = r - lv.A @ d
        rho_new = 1.0 / (2.0 * sigma - rho)
        d = (rho_new * rho) * d + (2.0 * rho_new / delta) * (lv.dinv * r)
        rho = rho_new
        x = x + d
    return x


def coarse_solve(lv: Level, b: np.ndarray) -> np.ndarray:
    r = o.pcg(lv.A, b, lambda v: lv.dinv * v, rtol=1e-12, atol=1e-300, max_it=500)
    return r.x


def vcycle(levels: List[Level], b: np.ndarray, steps: int = 2, l: int = 0) -> np.ndarray:
    lv = levels[l]
    if l == len(levels) - 1:
        return coarse_solve(lv, b)
    x = chebyshev(lv, b, None, steps)
    r = b - lv.A @ x
    r[lv.mask] = 0.0
    bc = lv.P.T @ r
    bc[levels[l + 1].mask] = 0.0
    xc = vcycle(levels, bc, steps, l + 1)
    corr = lv.P @ xc
    corr[lv.mask] = 0.0
    x = x + corr
    return chebyshev(lv, b, x, steps)
