"""
Matrix export + condition numbers — mirror of reference ``src/perphil/solvers/conditioning.py``
(SURVEY.md §8f rank 4): ``get_matrix_data_from_form`` (:66-102) assembles on the device and returns the
SciPy CSR the reference extracts from PETSc (``getValuesCSR`` + ``eliminate_zeros``);
``calculate_condition_number`` (:105-218) is the same dense-SVD / extreme-singular-value computation the
reference runs in SciPy on the host (it is analysis code there as well, not part of the solve).
"""
from __future__ import annotations

from typing import List, Optional

import attr
import numpy as np
from scipy.sparse import csr_matrix

from . import _ffi, fd
from .forms import DPPBilinearForm, ScalarBlockForm

DEFAULT_CONDITION_NUMBER_TOLERANCE = 1e-7


@attr.define(frozen=True)
class MatrixData:
    """Same fields as the reference's MatrixData minus the PETSc handles (there is no PETSc here)."""
    is_symmetric: bool
    sparse_csr_data: csr_matrix
    number_of_nonzero_entries: int
    number_of_dofs: int
    symmetry_tolerance: float


def _set_bcs(ctx, space, bcs: List[fd.DirichletBC]) -> None:
    empty = (np.zeros(0, np.int64), np.zeros(0))
    per_field = {0: empty, 1: empty}
    for bc in bcs or []:
        per_field[bc.field] = bc.nodes_and_values()
    for f in (0, 1):
        ctx.set_dirichlet(f, *per_field[f])


def get_matrix_data_from_form(form, boundary_conditions: List[fd.DirichletBC], symmetry_tolerance: float = 1e-8) -> MatrixData:
    """Assemble `form` (monolithic DPP form or one Picard block) with the BCs and export it as SciPy CSR."""
    if form.space.mesh().distributed:
        raise NotImplementedError("matrix export is an analysis path on the whole matrix: build the mesh with "
                                  "comm=fd.COMM_SELF under torch.distributed")
    if isinstance(form, DPPBilinearForm):
        mesh = form.space.mesh()
        ctx = mesh.context()
        _set_bcs(ctx, form.space, boundary_conditions)
        ctx.assemble(form.k1, form.k2, form.beta, form.mu, monolithic=True)
        csr = ctx.csr(_ffi.MAT_MONO)
    elif isinstance(form, ScalarBlockForm) and form.rank == 2:
        mesh = form.space.mesh()
        ctx = mesh.context()
        # a scalar block (coef_K K + coef_M M): assemble the pair with the block's coefficients on its own field
        bcs = [fd.DirichletBC(_as_field(bc, form.field), bc.value, bc.sub_domain) for bc in boundary_conditions or []]
        _set_bcs(ctx, form.space, bcs)
        k = form.coef_K if form.coef_K > 0 else 1.0
        ctx.assemble(k, k, form.coef_M, 1.0, monolithic=False)
        csr = ctx.csr(_ffi.MAT_A11 if form.field == 0 else _ffi.MAT_A22)
    else:
        raise TypeError(f"cannot assemble {type(form)}")
    csr = csr_matrix(csr)
    csr.eliminate_zeros()
    asym = abs(csr - csr.T)
    is_symmetric = bool((asym.max() if asym.nnz else 0.0) <= symmetry_tolerance)
    return MatrixData(is_symmetric, csr, int(csr.nnz), int(csr.shape[0]), symmetry_tolerance)


class _FieldView(fd.FunctionSpace):
    def __init__(self, V, field):
        super().__init__(V.mesh(), "CG", 1)
        self.index = field


def _as_field(bc: fd.DirichletBC, field: int):
    return _FieldView(bc.function_space(), field)


def assemble_bilinear_form(form, boundary_conditions: List[fd.DirichletBC]):
    """The assembled operator of `form` with the BCs applied (reference conditioning.py:51-63 returns the Firedrake
    matrix; here: the SciPy CSR exported from the device, explicit zeros of the eliminated pattern kept, like a
    PETSc aij matrix)."""
    if isinstance(form, DPPBilinearForm):
        ctx = form.space.mesh().context()
        _set_bcs(ctx, form.space, boundary_conditions)
        ctx.assemble(form.k1, form.k2, form.beta, form.mu, monolithic=True)
        return csr_matrix(ctx.csr(_ffi.MAT_MONO))
    return get_matrix_data_from_form(form, boundary_conditions).sparse_csr_data


def _dense_singular_values(A: csr_matrix) -> np.ndarray:
    return np.linalg.svd(A.toarray(), compute_uv=False)


def _extreme_singular_values(A: csr_matrix):
    """sigma_max and sigma_min without the full spectrum, in the order of the reference's sparse branch
    (conditioning.py:155-205): sigma_max from ARPACK `svds(which="LM")`; sigma_min from `svds(which="SM", tol=1e-8)`,
    failing that from the smallest eigenvalue of the normal equations A^T A (`eigsh(which="SM")`), failing that from a
    dense SVD - and any failure of the sigma_max call falls back to a dense SVD as well.  Returns (nan, None) semantics
    like the reference: sigma_min None when every route failed."""
    from scipy.sparse.linalg import eigsh, svds

    A = A.astype(np.float64)
    try:
        smax = float(np.max(svds(A, k=1, which="LM", maxiter=10000, return_singular_vectors=False, solver="arpack")))
    except Exception:
        sv = _dense_singular_values(A)
        smax = float(sv.max()) if sv.size else float("nan")
    smin = None
    try:
        smin = float(np.min(svds(A, k=1, which="SM", maxiter=20000, return_singular_vectors=False, solver="arpack", tol=1e-8)))
    except Exception:
        try:
            lam = eigsh((A.T @ A), k=1, which="SM", return_eigenvectors=False)
            smin = float(np.sqrt(max(float(lam[0]), 0.0)))
        except Exception:
            sv = _dense_singular_values(A)
            if sv.size:
                smin = float(sv.min())
    return smax, smin


def calculate_condition_number(scipy_csr_sparse_matrix: csr_matrix, num_singular_values: Optional[int] = None,
                               use_sparse: bool = False, zero_tol: float = DEFAULT_CONDITION_NUMBER_TOLERANCE,
                               num_of_factors: Optional[int] = None) -> float:
    """sigma_max / sigma_min over singular values above `zero_tol` (reference conditioning.py:105-218).  Dense SVD
    unless `use_sparse` with a positive `num_singular_values` (alias `num_of_factors`) below min(shape) - 1: then only
    the two extreme singular values are computed iteratively, as in the reference's sparse branch (:155-218; inf when
    the smallest one does not exceed the tolerance)."""
    A = scipy_csr_sparse_matrix
    k = num_singular_values if num_singular_values is not None else num_of_factors
    nmin = min(A.shape)
    if nmin == 0:
        return float("nan")
    if use_sparse and k is not None and 0 < int(k) < nmin - 1:
        smax, smin = _extreme_singular_values(csr_matrix(A))
        if smin is None or not np.isfinite(smax):
            return float("nan")
        return float("inf") if smin <= zero_tol else float(smax / smin)
    s = np.linalg.svd(A.toarray() if hasattr(A, "toarray") else np.asarray(A), compute_uv=False)
    s = s[s > zero_tol]
    if s.size == 0:
        return float("inf")
    return float(s.max() / s.min())
