"""
ORACLE — test infrastructure only.  NOT part of the product path.

CPU (NumPy/SciPy) restatement of the reference's DPP hot path: CG-1 assembly of the
two-pressure double-porosity/permeability system and the Krylov / Picard solve.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; ``perphil_amd`` never does (the product fails loudly without its HIP
library).

What is restated, and from where (paths relative to the reference checkout):

* operator        ``src/perphil/forms/dpp.py:27,57-58,89-90,129-130``
                  A = [[a K + b M, -b M], [-b M, c K + b M]],  a=k1/mu, b=beta/mu, c=k2/mu,  rhs 0
* Picard split    ``src/perphil/forms/dpp.py:196-203`` (``dpp_delayed_form``)
* coefficients    ``src/perphil/models/dpp/parameters.py:26-52`` (defaults, ``eta``)
* boundary data   ``src/perphil/utils/manufactured_solutions.py:39-51`` (2D), ``:87-88`` (3D)
* tolerances      ``src/perphil/solvers/parameters.py:12-18`` (rtol 1e-8, atol 1e-12, max_it 50000)
* dof layout      ``src/perphil/experiments/iterative_bench.py:323-324`` (field-major: all p1, then all p2)
* CSR export      ``src/perphil/solvers/conditioning.py:85-86``; condition number ``:134-154``
* meshes          ``src/perphil/mesh/builtin.py:20`` (UnitSquareMesh quads / "left"-diagonal
                  triangles), ``src/perphil/experiments/petsc_profiling_3d.py:31`` (UnitCubeMesh,
                  6 Kuhn tets per cube), ``notebooks/condition-number-study-3d.py:66`` (hexes)

The arithmetic itself lives in third-party packages absent from the reference tree and from
this image (Firedrake 2025.10.x, PETSc 3.24.0, MUMPS): their published algorithms (Galerkin
FEM with exact quadrature on affine/multilinear cells; symmetric Dirichlet elimination with
unit diagonal; left-preconditioned restarted GMRES(30) tested on the preconditioned residual)
are restated here.

PINNING: this oracle is pinned against the reference's committed study outputs (stored
notebook cells and result CSVs) — see ``tests/golden/reference_goldens.json`` and
``tests/test_oracle_goldens.py`` (G1..G9 of SURVEY.md §8c).  All of those are invariant under
dof renumbering, which matters because Firedrake's DMPlex numbering is not reproducible here;
the numbering used below is lexicographic: node (i,j,k) -> i + (nx+1)*(j + (ny+1)*k),
global index = field*n + node.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

CELL_QUAD, CELL_TRI, CELL_HEX, CELL_TET = 0, 1, 2, 3
_NODES_PER_CELL = {CELL_QUAD: 4, CELL_TRI: 3, CELL_HEX: 8, CELL_TET: 4}


# --------------------------------------------------------------------------------------
# mesh: lexicographic structured unit square / cube
# --------------------------------------------------------------------------------------
@dataclass
class Mesh:
    dim: int
    kind: int
    nx: int
    ny: int
    nz: int
    coords: np.ndarray  # [n, dim] float64
    cells: np.ndarray  # [ncell, m] int32

    @property
    def num_nodes(self) -> int:
        return self.coords.shape[0]

    @property
    def num_cells(self) -> int:
        return self.cells.shape[0]


def build_mesh(dim: int, kind: int, nx: int, ny: int, nz: int = 0) -> Mesh:
    """Unit square (dim 2: quads or left-diagonal triangles) / unit cube (dim 3: hexes or 6 Kuhn
    tets per cube sharing the diagonal v0-v7)."""
    if dim == 2:
        px, py = nx + 1, ny + 1
        ii, jj = np.meshgrid(np.arange(px), np.arange(py), indexing="xy")
        coords = np.stack([ii.ravel() / nx, jj.ravel() / ny], axis=1).astype(np.float64)
        ci, cj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
        v0 = (ci + px * cj).ravel()
        v1, v2, v3 = v0 + 1, v0 + px, v0 + px + 1  # +x, +y, +x+y
        if kind == CELL_QUAD:
            cells = np.stack([v0, v1, v2, v3], axis=1)
        elif kind == CELL_TRI:
            # "left" diagonal joins (i+1,j) with (i,j+1)
            cells = np.stack([np.stack([v0, v1, v2], 1), np.stack([v1, v3, v2], 1)], axis=1).reshape(-1, 3)
        else:
            raise ValueError("2D kinds: quad, tri")
    elif dim == 3:
        px, py, pz = nx + 1, ny + 1, nz + 1
        kk, jj, ii = np.meshgrid(np.arange(pz), np.arange(py), np.arange(px), indexing="ij")
        coords = np.stack([ii.ravel() / nx, jj.ravel() / ny, kk.ravel() / nz], axis=1).astype(np.float64)
        ck, cj, ci = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        v0 = (ci + px * (cj + py * ck)).ravel()
        dx, dy, dz = 1, px, px * py
        v = [v0, v0 + dx, v0 + dy, v0 + dx + dy, v0 + dz, v0 + dx + dz, v0 + dy + dz, v0 + dx + dy + dz]
        if kind == CELL_HEX:
            cells = np.stack(v, axis=1)
        elif kind == CELL_TET:
            tets = [(0, 1, 3, 7), (0, 1, 7, 5), (0, 5, 7, 4), (0, 3, 2, 7), (0, 6, 4, 7), (0, 2, 6, 7)]
            cells = np.stack([np.stack([v[a] for a in t], 1) for t in tets], axis=1).reshape(-1, 4)
        else:
            raise ValueError("3D kinds: hex, tet")
    else:
        raise ValueError("dim must be 2 or 3")
    return Mesh(dim, kind, nx, ny, nz, coords, cells.astype(np.int32))


def boundary_nodes(mesh: Mesh) -> np.ndarray:
    """Sorted node ids on the boundary of the unit square/cube ("on_boundary")."""
    px, py = mesh.nx + 1, mesh.ny + 1
    n = mesh.num_nodes
    idx = np.arange(n)
    i, j = idx % px, (idx // px) % py
    on = (i == 0) | (i == mesh.nx) | (j == 0) | (j == mesh.ny)
    if mesh.dim == 3:
        k = idx // (px * py)
        on |= (k == 0) | (k == mesh.nz)
    return idx[on].astype(np.int64)


# --------------------------------------------------------------------------------------
# element matrices
# --------------------------------------------------------------------------------------
def _gauss2():
    g = 1.0 / math.sqrt(3.0)
    return np.array([-g, g]), np.array([1.0, 1.0])


def element_matrices(mesh: Mesh) -> Tuple[np.ndarray, np.ndarray]:
    """Per-cell stiffness K_e = int grad(phi_a).grad(phi_b) and mass M_e = int phi_a phi_b.
    Multilinear cells: 2-point Gauss per direction on the isoparametric map (exact on the
    uniform grid); simplices: constant gradients, M_e = vol/((d+1)(d+2)) (1 + I)."""
    X = mesh.coords[mesh.cells]  # [nc, m, d]
    nc, m, d = X.shape
    if mesh.kind in (CELL_QUAD, CELL_HEX):
        pts, wts = _gauss2()
        K = np.zeros((nc, m, m))
        M = np.zeros((nc, m, m))
        grids = np.meshgrid(*([np.arange(2)] * d), indexing="ij")
        for q in zip(*[g.ravel() for g in grids]):
            xi = [pts[t] for t in q]
            w = float(np.prod([wts[t] for t in q]))
            # local node a has reference corner bits (a&1, a>>1&1, a>>2&1)
            N = np.ones(m)
            dN = np.ones((m, d))
            for a in range(m):
                for c in range(d):
                    s = 1.0 if (a >> c) & 1 else -1.0
                    N[a] *= 0.5 * (1.0 + s * xi[c])
                    for e in range(d):
                        dN[a, e] *= (0.5 * s) if e == c else 0.5 * (1.0 + s * xi[c])
            J = np.einsum("ae,cad->ced", dN, X)  # J[c, e, d] = d x_d / d xi_e
            detJ = np.linalg.det(J)
            Jinv = np.linalg.inv(J)  # [c, d, e]
            G = np.einsum("cde,ae->cad", Jinv, dN)  # physical gradients [c, a, d]
            K += (w * detJ)[:, None, None] * np.einsum("cad,cbd->cab", G, G)
            M += (w * detJ)[:, None, None] * np.outer(N, N)[None]
        return K, M
    # simplices
    E = X[:, 1:, :] - X[:, :1, :]  # [nc, d, d] edge vectors as rows
    detE = np.linalg.det(E)
    fact = 2.0 if d == 2 else 6.0
    vol = np.abs(detE) / fact
    Einv = np.linalg.inv(E)  # columns = gradients of barycentric lambda_1..d
    G = np.zeros((nc, m, d))
    G[:, 1:, :] = np.transpose(Einv, (0, 2, 1))
    G[:, 0, :] = -G[:, 1:, :].sum(axis=1)
    K = vol[:, None, None] * np.einsum("cad,cbd->cab", G, G)
    M = (vol / ((d + 1) * (d + 2)))[:, None, None] * (np.ones((m, m)) + np.eye(m))[None]
    return K, M


def assemble_scalar(mesh: Mesh) -> Tuple[sp.csr_matrix, sp.csr_matrix]:
    """Scalar stiffness K and mass M (n x n CSR, sorted columns, identical patterns)."""
    Ke, Me = element_matrices(mesh)
    n = mesh.num_nodes
    m = mesh.cells.shape[1]
    rows = np.repeat(mesh.cells, m, axis=1).ravel()
    cols = np.tile(mesh.cells, (1, m)).ravel()
    K = sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    M = sp.coo_matrix((Me.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    K.sum_duplicates(); M.sum_duplicates()
    K.sort_indices(); M.sort_indices()
    return K, M


# --------------------------------------------------------------------------------------
# model parameters / manufactured boundary data
# --------------------------------------------------------------------------------------
@dataclass
class Params:
    """``DPPParameters`` (models/dpp/parameters.py:26-30): k2 defaults to k1/scale_contrast."""
    k1: float = 1.0
    k2: Optional[float] = None
    beta: float = 1.0
    mu: float = 1.0
    scale_contrast: float = 1e2

    def __post_init__(self):
        if self.k2 is None:
            self.k2 = self.k1 / self.scale_contrast

    @property
    def eta(self) -> float:  # parameters.py:52
        return math.sqrt(self.beta * (self.k1 + self.k2) / (self.k1 * self.k2))

    @property
    def abc(self) -> Tuple[float, float, float]:
        return self.k1 / self.mu, self.beta / self.mu, self.k2 / self.mu


def exact_pressures(coords: np.ndarray, p: Params) -> Tuple[np.ndarray, np.ndarray]:
    """Manufactured p1, p2 at points (manufactured_solutions.py:39-51 in 2D, :87-88 in 3D)."""
    x, y = coords[:, 0], coords[:, 1]
    eta = p.eta
    if coords.shape[1] == 2:
        common = (p.mu / math.pi) * np.exp(math.pi * x) * np.sin(math.pi * y)
        e = np.exp(eta * y)
    else:
        z = coords[:, 2]
        common = (p.mu / math.pi) * np.exp(math.pi * x) * (np.sin(math.pi * y) + np.sin(math.pi * z))
        e = np.exp(eta * y) + np.exp(eta * z)
    return common - (p.mu / (p.beta * p.k1)) * e, common + (p.mu / (p.beta * p.k2)) * e


# --------------------------------------------------------------------------------------
# monolithic system with Dirichlet elimination
# --------------------------------------------------------------------------------------
@dataclass
class System:
    A_full: sp.csr_matrix  # 2n x 2n, no BCs
    A: sp.csr_matrix  # after symmetric elimination (pattern kept, explicit zeros)
    u0: np.ndarray  # boundary values on boundary dofs, 0 inside
    rhs: np.ndarray  # -(A_full u0) on interior, 0 on boundary
    bc_dofs: np.ndarray
    n: int  # dofs per field


def monolithic_matrix(K: sp.csr_matrix, M: sp.csr_matrix, p: Params) -> sp.csr_matrix:
    a, b, c = p.abc
    A = sp.bmat([[a * K + b * M, -b * M], [-b * M, c * K + b * M]], format="csr")
    A.sort_indices()
    return A


def eliminate_dirichlet(A_full: sp.csr_matrix, bc_dofs: np.ndarray) -> sp.csr_matrix:
    """Zero boundary rows and columns, unit diagonal; sparsity pattern kept (explicit zeros)."""
    A = A_full.copy().tocsr()
    N = A.shape[0]
    isbc = np.zeros(N, dtype=bool)
    isbc[bc_dofs] = True
    rows = np.repeat(np.arange(N), np.diff(A.indptr))
    kill = isbc[rows] | isbc[A.indices]
    A.data[kill] = 0.0
    diag = kill & (rows == A.indices)
    A.data[diag] = 1.0
    return A


def build_system(mesh: Mesh, p: Params, g1: Optional[np.ndarray] = None, g2: Optional[np.ndarray] = None,
                 mms: bool = True, mask1: Optional[np.ndarray] = None, mask2: Optional[np.ndarray] = None) -> System:
    """Monolithic DPP system with Dirichlet data on the whole boundary of both fields.
    ``mms=True`` takes the data from the manufactured solution; otherwise ``g1``/``g2`` are
    nodal arrays (only boundary entries are read); None -> homogeneous.  ``mask1``/``mask2`` (boolean nodal
    arrays) constrain an arbitrary node set per field instead of the boundary (same elimination), with the
    values of ``g1``/``g2`` there."""
    K, M = assemble_scalar(mesh)
    n = mesh.num_nodes
    A_full = monolithic_matrix(K, M, p)
    if mask1 is not None or mask2 is not None:
        b1 = np.nonzero(mask1)[0] if mask1 is not None else np.zeros(0, np.int64)
        b2 = np.nonzero(mask2)[0] if mask2 is not None else np.zeros(0, np.int64)
        bc = np.concatenate([b1, b2 + n]).astype(np.int64)
        u0 = np.zeros(2 * n)
        if g1 is not None:
            u0[b1] = g1[b1]
        if g2 is not None:
            u0[n + b2] = g2[b2]
        F = A_full @ u0
        F[bc] = 0.0
        return System(A_full, eliminate_dirichlet(A_full, bc), u0, -F, bc, n)
    b = boundary_nodes(mesh)
    bc = np.concatenate([b, b + n])
    u0 = np.zeros(2 * n)
    if mms:
        e1, e2 = exact_pressures(mesh.coords, p)
        u0[b], u0[n + b] = e1[b], e2[b]
    else:
        if g1 is not None:
            u0[b] = g1[b]
        if g2 is not None:
            u0[n + b] = g2[b]
    F = A_full @ u0
    F[bc] = 0.0
    return System(A_full, eliminate_dirichlet(A_full, bc), u0, -F, bc, n)


# --------------------------------------------------------------------------------------
# Krylov solvers with PETSc semantics (left PC, preconditioned residual test)
# --------------------------------------------------------------------------------------
@dataclass
class KspResult:
    x: np.ndarray
    its: int
    resnorm: float
    history: list
    converged: bool


def gmres(A, b, M_apply: Optional[Callable] = None, rtol=1e-8, atol=1e-12, max_it=50000, restart=30,
          x0: Optional[np.ndarray] = None) -> KspResult:
    """Left-preconditioned restarted GMRES, modified Gram-Schmidt, Givens residual recurrence;
    stops when ||P^-1 r|| <= max(rtol*||P^-1 b||, atol); counts every inner step."""
    N = b.shape[0]
    prec = (lambda v: v) if M_apply is None else M_apply
    x = np.zeros(N) if x0 is None else x0.copy()
    r = prec(b - A @ x) if x0 is not None else prec(b)
    beta = float(np.linalg.norm(r))
    bnorm = float(np.linalg.norm(prec(b))) if x0 is not None else beta
    tol = max(rtol * bnorm, atol)
    hist = [beta]
    its = 0
    if beta <= tol:
        return KspResult(x, 0, beta, hist, True)
    while its < max_it:
        V = np.zeros((restart + 1, N))
        H = np.zeros((restart + 1, restart))
        cs, sn = np.zeros(restart), np.zeros(restart)
        g = np.zeros(restart + 1)
        g[0] = beta
        V[0] = r / beta
        k_used = 0
        done = False
        for k in range(restart):
            w = prec(A @ V[k])
            for i in range(k + 1):
                H[i, k] = float(np.dot(V[i], w))
                w -= H[i, k] * V[i]
            H[k + 1, k] = float(np.linalg.norm(w))
            if H[k + 1, k] > 0.0:
                V[k + 1] = w / H[k + 1, k]
            for i in range(k):
                t = cs[i] * H[i, k] + sn[i] * H[i + 1, k]
                H[i + 1, k] = -sn[i] * H[i, k] + cs[i] * H[i + 1, k]
                H[i, k] = t
            d = math.hypot(H[k, k], H[k + 1, k])
            cs[k], sn[k] = H[k, k] / d, H[k + 1, k] / d
            H[k, k], H[k + 1, k] = d, 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            its += 1
            k_used = k + 1
            res = abs(g[k + 1])
            hist.append(res)
            if res <= tol or its >= max_it:
                done = True
                break
        y = np.linalg.solve(np.triu(H[:k_used, :k_used]), g[:k_used])
        x = x + V[:k_used].T @ y
        if done:
            return KspResult(x, its, hist[-1], hist, hist[-1] <= tol)
        r = prec(b - A @ x)
        beta = float(np.linalg.norm(r))
    return KspResult(x, its, hist[-1], hist, False)


def pcg(A, b, M_apply: Optional[Callable] = None, rtol=1e-8, atol=1e-12, max_it=50000,
        x0: Optional[np.ndarray] = None, reduction: float = 0.0, norm: str = "preconditioned") -> KspResult:
    """Preconditioned CG.  ``norm="preconditioned"`` (PETSc default): tests ||z||_2, z = P^-1 r;
    ``norm="unpreconditioned"`` (KSP_NORM_UNPRECONDITIONED): tests ||r||_2, known before the preconditioner is
    applied.  ``reduction`` > 0 also accepts a drop of that norm by that factor from its value at the start of this
    solve (inexact Picard sweeps)."""
    prec = (lambda v: v) if M_apply is None else M_apply
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x if x0 is not None else b.copy()
    if norm == "none":
        # KSP_NORM_NONE: exactly max_it iterations, no test (ksp_norm_type none + ksp_max_it)
        z = prec(r)
        p = z.copy()
        rz = float(np.dot(r, z))
        for it in range(max_it):
            Ap = A @ p
            alpha = rz / float(np.dot(p, Ap))
            x += alpha * p
            r -= alpha * Ap
            if it == max_it - 1:
                break
            z = prec(r)
            rz_new = float(np.dot(r, z))
            p = z + (rz_new / rz) * p
            rz = rz_new
        res = float(np.linalg.norm(r))
        return KspResult(x, max_it, res, [res], True)
    if norm == "unpreconditioned":
        res = float(np.linalg.norm(r))
        tol = max(rtol * float(np.linalg.norm(b)), atol, reduction * res)
        hist = [res]
        if res <= tol:
            return KspResult(x, 0, res, hist, True)
        z = prec(r)
        p = z.copy()
        rz = float(np.dot(r, z))
        its = 0
        while its < max_it:
            Ap = A @ p
            alpha = rz / float(np.dot(p, Ap))
            x += alpha * p
            r -= alpha * Ap
            its += 1
            res = float(np.linalg.norm(r))
            hist.append(res)
            if res <= tol:
                return KspResult(x, its, res, hist, True)
            z = prec(r)
            rz_new = float(np.dot(r, z))
            p = z + (rz_new / rz) * p
            rz = rz_new
        return KspResult(x, its, hist[-1], hist, False)
    z = prec(r)
    z0 = prec(b) if x0 is not None else z
    res = float(np.linalg.norm(z))
    tol = max(rtol * float(np.linalg.norm(z0)), atol, reduction * res)
    hist = [res]
    if res <= tol:
        return KspResult(x, 0, res, hist, True)
    p = z.copy()
    rz = float(np.dot(r, z))
    its = 0
    while its < max_it:
        Ap = A @ p
        alpha = rz / float(np.dot(p, Ap))
        x += alpha * p
        r -= alpha * Ap
        z = prec(r)
        its += 1
        res = float(np.linalg.norm(z))
        hist.append(res)
        if res <= tol:
            return KspResult(x, its, res, hist, True)
        rz_new = float(np.dot(r, z))
        p = z + (rz_new / rz) * p
        rz = rz_new
    return KspResult(x, its, hist[-1], hist, False)


def jacobi_apply(A: sp.csr_matrix) -> Callable:
    dinv = 1.0 / A.diagonal()
    return lambda v: dinv * v


def block2_jacobi_apply(A: sp.csr_matrix, n: int) -> Callable:
    """Node-block (2x2, fields coupled at one node) Jacobi."""
    d11 = A.diagonal()[:n]
    d22 = A.diagonal()[n:]
    d12 = np.asarray(A[:n, n:].diagonal())
    d21 = np.asarray(A[n:, :n].diagonal())
    det = d11 * d22 - d12 * d21

    def apply(v):
        v1, v2 = v[:n], v[n:]
        return np.concatenate([(d22 * v1 - d12 * v2) / det, (d11 * v2 - d21 * v1) / det])

    return apply


def fieldsplit_multiplicative_apply(A: sp.csr_matrix, n: int) -> Callable:
    """PETSc pc_fieldsplit_type multiplicative with exact (LU) block solves
    (solvers/parameters.py:30-37): z1 = A11^-1 r1 ; z2 = A22^-1 (r2 - A21 z1)."""
    A = A.tocsr()
    lu11 = spla.splu(A[:n, :n].tocsc())
    lu22 = spla.splu(A[n:, n:].tocsc())
    A21 = A[n:, :n]

    def apply(v):
        z1 = lu11.solve(v[:n])
        z2 = lu22.solve(v[n:] - A21 @ z1)
        return np.concatenate([z1, z2])

    return apply


def ilu0(A: sp.csr_matrix) -> sp.csr_matrix:
    """ILU(0) in the natural (row) ordering on the pattern of A, explicit zeros of the pattern kept (PETSc's
    pc_type ilu with pc_factor_levels 0, reference src/perphil/solvers/parameters.py:27; PETSc's algorithm: IKJ
    Gaussian elimination restricted to the pattern).  Returns one CSR holding the strict lower part of L (unit
    diagonal implied) and U.  Plain Python loops over the rows: small systems only (oracle)."""
    A = sp.csr_matrix(A).copy()
    A.sort_indices()
    ip, ix, v = A.indptr, A.indices, A.data.astype(np.float64).copy()
    n = A.shape[0]
    diag = np.full(n, -1, dtype=np.int64)
    for i in range(n):
        cols = ix[ip[i]:ip[i + 1]]
        pos = np.searchsorted(cols, i)
        assert pos < cols.size and cols[pos] == i, "ILU(0) needs a structurally non-zero diagonal"
        diag[i] = ip[i] + pos
    for i in range(n):
        lo, hi = ip[i], ip[i + 1]
        where = {int(c): lo + q for q, c in enumerate(ix[lo:hi])}
        for kk in range(lo, diag[i]):
            k = int(ix[kk])
            piv = v[kk] / v[diag[k]]
            v[kk] = piv
            for jj in range(diag[k] + 1, ip[k + 1]):
                pos = where.get(int(ix[jj]))
                if pos is not None:
                    v[pos] -= piv * v[jj]
    return sp.csr_matrix((v, ix.copy(), ip.copy()), shape=A.shape)


def ilu0_apply(A: sp.csr_matrix) -> Callable:
    """z = (L U)^-1 r with the ILU(0) factors of A (forward / backward substitution)."""
    LU = ilu0(A)
    L = sp.tril(LU, k=-1, format="csr") + sp.identity(A.shape[0], format="csr")
    U = sp.triu(LU, k=0, format="csr")
    return lambda r: spla.spsolve_triangular(U, spla.spsolve_triangular(L, r, lower=True, unit_diagonal=True), lower=False)


def fieldsplit_ilu_gmres_apply(A: sp.csr_matrix, n: int, rtol=1e-8, atol=1e-12, max_it=50000, preonly=False) -> Callable:
    """pc_fieldsplit multiplicative whose block solves are GMRES + ILU(0) (FIELDSPLIT_GMRES_ILU_PARAMS,
    parameters.py:50-57) or, with `preonly`, one ILU(0) application (iterative_bench.make_fieldsplit_params_with)."""
    A = A.tocsr()
    A11, A22, A21 = A[:n, :n].tocsr(), A[n:, n:].tocsr(), A[n:, :n]
    p11, p22 = ilu0_apply(A11), ilu0_apply(A22)

    def block(Ab, pb, r):
        return pb(r) if preonly else gmres(Ab, r, pb, rtol=rtol, atol=atol, max_it=max_it).x

    def apply(v):
        z1 = block(A11, p11, v[:n])
        z2 = block(A22, p22, v[n:] - A21 @ z1)
        return np.concatenate([z1, z2])

    return apply


def solve_direct(sys_: System) -> np.ndarray:
    du = spla.spsolve(sys_.A.tocsc(), sys_.rhs)
    return sys_.u0 + du


def picard(sys_: System, rtol=1e-8, atol=1e-12, max_it=50000, inner: str = "direct",
           inner_rtol=1e-10) -> Tuple[np.ndarray, int, float, list]:
    """Block Picard (fixed-stress / block Gauss-Seidel) per ``dpp_delayed_form``
    (forms/dpp.py:196-203): A11 p1 = r1 - A12 p2_old ; A22 p2 = r2 - A21 p1_new.
    Iterates on the correction du with homogeneous BCs; stops on the monolithic residual
    ||rhs - A du|| <= max(rtol*||rhs||, atol)."""
    n, A, b = sys_.n, sys_.A.tocsr(), sys_.rhs
    A11, A12, A21, A22 = A[:n, :n], A[:n, n:], A[n:, :n], A[n:, n:]
    if inner == "direct":
        s11 = spla.splu(A11.tocsc()).solve
        s22 = spla.splu(A22.tocsc()).solve
    else:
        s11 = lambda r: pcg(A11, r, jacobi_apply(A11), rtol=inner_rtol).x
        s22 = lambda r: pcg(A22, r, jacobi_apply(A22), rtol=inner_rtol).x
    du = np.zeros(2 * n)
    r0 = float(np.linalg.norm(b))
    tol = max(rtol * r0, atol)
    hist = [r0]
    its = 0
    while hist[-1] > tol and its < max_it:
        du[:n] = s11(b[:n] - A12 @ du[n:])
        du[n:] = s22(b[n:] - A21 @ du[:n])
        its += 1
        hist.append(float(np.linalg.norm(b - A @ du)))
    return sys_.u0 + du, its, hist[-1], hist


# --------------------------------------------------------------------------------------
# analysis helpers (conditioning.py:134-154)
# --------------------------------------------------------------------------------------
def condition_number(A: sp.spmatrix, zero_tol: float = 1e-7) -> float:
    s = np.linalg.svd(A.toarray(), compute_uv=False)
    s = s[s > zero_tol]
    return float(s.max() / s.min())


def slice_along_x(mesh: Mesh, field: np.ndarray, x_value: float) -> Tuple[np.ndarray, np.ndarray]:
    """Nodal values on the vertical line x = x_value (must be a grid line); 2D only
    (utils/postprocessing.py:66-86)."""
    i = int(round(x_value * mesh.nx))
    assert abs(i / mesh.nx - x_value) < 1e-14
    ids = i + (mesh.nx + 1) * np.arange(mesh.ny + 1)
    return mesh.coords[ids, 1], field[ids]


# --------------------------------------------------------------------------------------
# error norms (utils/postprocessing.py:89-124) with a high-order tensor Gauss rule
# --------------------------------------------------------------------------------------
def error_norms(mesh: Mesh, ph: np.ndarray, exact: Callable, exact_grad: Callable, nq: int = 6) -> Tuple[float, float]:
    """L2 and H1-seminorm errors of the nodal CG-1 field ``ph``: tensor Gauss rule on quads / hexes, the same rule
    collapsed onto the simplex (Duffy transform) on triangles / tetrahedra."""
    d = mesh.dim
    if mesh.kind in (CELL_TRI, CELL_TET):
        pts, wts = np.polynomial.legendre.leggauss(nq)
        pts, wts = 0.5 * (pts + 1.0), 0.5 * wts
        X = mesh.coords[mesh.cells]
        U = ph[mesh.cells]
        E = X[:, 1:, :] - X[:, :1, :]
        dU = U[:, 1:] - U[:, :1]
        gh = np.linalg.solve(E, dU[:, :, None])[:, :, 0]
        detE = np.abs(np.linalg.det(E))
        l2 = h1 = 0.0
        grids = np.meshgrid(*([np.arange(nq)] * d), indexing="ij")
        for q in zip(*[g.ravel() for g in grids]):
            u_, v_ = pts[q[0]], pts[q[1]]
            lam = [u_, v_ * (1.0 - u_)]
            w = wts[q[0]] * wts[q[1]] * (1.0 - u_)
            if d == 3:
                w_ = pts[q[2]]
                lam.append(w_ * (1.0 - u_) * (1.0 - v_))
                w *= wts[q[2]] * (1.0 - u_) * (1.0 - v_)
            lam = np.array(lam)
            xq = X[:, 0, :] + np.einsum("r,crd->cd", lam, E)
            uh = U[:, 0] + dU @ lam
            l2 += float(np.sum(w * detE * (uh - exact(xq)) ** 2))
            h1 += float(np.sum(w * detE * np.sum((gh - exact_grad(xq)) ** 2, axis=1)))
        return math.sqrt(l2), math.sqrt(h1)
    pts, wts = np.polynomial.legendre.leggauss(nq)
    X = mesh.coords[mesh.cells]
    U = ph[mesh.cells]
    m = X.shape[1]
    l2 = 0.0
    h1 = 0.0
    grids = np.meshgrid(*([np.arange(nq)] * d), indexing="ij")
    for q in zip(*[g.ravel() for g in grids]):
        xi = [pts[t] for t in q]
        w = float(np.prod([wts[t] for t in q]))
        N = np.ones(m)
        dN = np.ones((m, d))
        for a in range(m):
            for c in range(d):
                s = 1.0 if (a >> c) & 1 else -1.0
                N[a] *= 0.5 * (1.0 + s * xi[c])
                for e in range(d):
                    dN[a, e] *= (0.5 * s) if e == c else 0.5 * (1.0 + s * xi[c])
        xq = np.einsum("a,cad->cd", N, X)
        J = np.einsum("ae,cad->ced", dN, X)
        detJ = np.linalg.det(J)
        G = np.einsum("cde,ae->cad", np.linalg.inv(J), dN)
        uh = U @ N
        guh = np.einsum("ca,cad->cd", U, G)
        l2 += float(np.sum(w * detJ * (uh - exact(xq)) ** 2))
        h1 += float(np.sum(w * detJ * np.sum((guh - exact_grad(xq)) ** 2, axis=1)))
    return math.sqrt(l2), math.sqrt(h1)


# --------------------------------------------------------------------------------------
# Darcy velocity (utils/postprocessing.py:34-63): fd.project(-k grad(p_h), VectorFunctionSpace(mesh, "CG", 1))
# --------------------------------------------------------------------------------------
def darcy_velocity(mesh: Mesh, ph: np.ndarray, conductivity: float, nq: int = 3) -> np.ndarray:
    """L2 projection of -k grad(p_h) onto CG-1 vectors: solve M u_d = b_d (sparse direct); returns [n, dim]."""
    d = mesh.dim
    n = mesh.num_nodes
    _, M = assemble_scalar(mesh)
    X = mesh.coords[mesh.cells]
    U = ph[mesh.cells]
    m = X.shape[1]
    b = np.zeros((n, d))
    if mesh.kind in (CELL_QUAD, CELL_HEX):
        pts, wts = np.polynomial.legendre.leggauss(nq)
        grids = np.meshgrid(*([np.arange(nq)] * d), indexing="ij")
        for q in zip(*[g.ravel() for g in grids]):
            xi = [pts[t] for t in q]
            w = float(np.prod([wts[t] for t in q]))
            N = np.ones(m)
            dN = np.ones((m, d))
            for a in range(m):
                for c in range(d):
                    s = 1.0 if (a >> c) & 1 else -1.0
                    N[a] *= 0.5 * (1.0 + s * xi[c])
                    for e in range(d):
                        dN[a, e] *= (0.5 * s) if e == c else 0.5 * (1.0 + s * xi[c])
            J = np.einsum("ae,cad->ced", dN, X)
            detJ = np.abs(np.linalg.det(J))
            G = np.einsum("cde,ae->cad", np.linalg.inv(J), dN)
            gp = np.einsum("ca,cad->cd", U, G)
            contrib = (-conductivity * w) * detJ[:, None, None] * N[None, :, None] * gp[:, None, :]
            for e in range(d):
                np.add.at(b[:, e], mesh.cells.ravel(), contrib[:, :, e].ravel())
    else:
        E = X[:, 1:, :] - X[:, :1, :]                       # [cell][r][d] edge vectors
        dP = U[:, 1:] - U[:, :1]
        gp = np.linalg.solve(E, dP[:, :, None])[:, :, 0]    # E grad = dP
        vol = np.abs(np.linalg.det(E)) / math.factorial(d)
        contrib = (-conductivity / (d + 1)) * vol[:, None, None] * np.ones((1, m, 1)) * gp[:, None, :]
        for e in range(d):
            np.add.at(b[:, e], mesh.cells.ravel(), contrib[:, :, e].ravel())
    lu = spla.splu(M.tocsc())
    return np.column_stack([lu.solve(b[:, e]) for e in range(d)])
