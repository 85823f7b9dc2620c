"""
Experiment harness — mirror of reference ``src/perphil/experiments/iterative_bench.py`` (SURVEY.md §8f
rank 3): ``Approach`` (:31-48), ``SolveResult`` (:51-76), ``build_mesh`` / ``build_spaces`` /
``default_bcs`` / ``default_model_params`` (:79-133), ``params_for`` (:157-188), ``solve_on_mesh``
(:191-252), ``assemble_monolithic_matrix`` (:255-287), ``estimate_condition_numbers`` (:290-337).
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Dict, List, Optional, Tuple

import numpy as np
import scipy.sparse as sp

from . import conditioning, fd, solver_parameters as solver_params
from .forms import dpp_form
from .mesh import create_mesh
from .parameters import DPPParameters
from .spaces import create_function_spaces


class Approach(str, Enum):
    PLAIN_GMRES = "GMRES"
    GMRES_ILU = "GMRES + ILU PC"
    SS_GMRES = "Scale-Splitting GMRES"
    SS_GMRES_ILU = "Scale-Splitting GMRES + ILU PC"
    PICARD_MUMPS = "Scaling-Splitting Picard with MUMPS"
    MONOLITHIC_MUMPS = "Monolithic LU with MUMPS"


@dataclass(frozen=True)
class SolveResult:
    approach: Approach
    nx: int
    ny: int
    iteration_number: int
    residual_error: float
    fields: Optional[Tuple[fd.Function, fd.Function]] = None


def build_mesh(nx: int, ny: int, quadrilateral: bool = True) -> fd.Mesh:
    return create_mesh(nx, ny, quadrilateral=quadrilateral)


def build_spaces(mesh: fd.Mesh):
    U, V = create_function_spaces(mesh)
    return U, V, fd.MixedFunctionSpace((V, V))


def default_bcs(W) -> List[fd.DirichletBC]:
    return [fd.DirichletBC(W.sub(0), fd.Constant(0.0), "on_boundary"), fd.DirichletBC(W.sub(1), fd.Constant(0.0), "on_boundary")]


def default_model_params() -> DPPParameters:
    return DPPParameters(k1=1.0, k2=1.0 / 1e2, beta=1.0, mu=1.0)


def make_fieldsplit_params_with(block_pc: str = "lu") -> Dict:
    """Field-split GMRES options with the given block preconditioner on both blocks ('lu' or 'ilu'); for 'ilu' the
    block solves are a single preconditioner application (reference iterative_bench.py:134-154)."""
    opts = {**solver_params.FIELDSPLIT_LU_PARAMS, "ksp_type": "gmres"}
    if block_pc.lower() != "lu":
        for i in (0, 1):
            opts[f"fieldsplit_{i}_pc_type"] = block_pc
            opts.setdefault(f"fieldsplit_{i}_ksp_type", "preonly")
    return opts


def params_for(approach: Approach) -> Dict:
    if approach == Approach.PLAIN_GMRES:
        return solver_params.PLAIN_GMRES_PARAMS.copy()
    if approach == Approach.GMRES_ILU:
        return solver_params.GMRES_ILU_PARAMS.copy()
    if approach == Approach.SS_GMRES:
        return {**solver_params.GMRES_PARAMS, **solver_params.FIELDSPLIT_LU_PARAMS}
    if approach == Approach.SS_GMRES_ILU:
        return {**solver_params.GMRES_PARAMS, **solver_params.FIELDSPLIT_GMRES_ILU_PARAMS}
    if approach == Approach.MONOLITHIC_MUMPS:
        return solver_params.LINEAR_SOLVER_PARAMS.copy()
    if approach == Approach.PICARD_MUMPS:
        return solver_params.PICARD_LU_SOLVER_PARAMS.copy()
    raise ValueError(f"Unknown approach: {approach}")


def solve_on_mesh(W, approach: Approach, params: Optional[DPPParameters] = None,
                  bcs: Optional[List[fd.DirichletBC]] = None) -> SolveResult:
    from .solver import solve_dpp, solve_dpp_nonlinear

    params = params or default_model_params()
    bcs = bcs or default_bcs(W)
    sp_dict = params_for(approach)
    if approach == Approach.PICARD_MUMPS:
        sol = solve_dpp_nonlinear(W, params, bcs=bcs, solver_parameters=sp_dict)
    else:
        sol = solve_dpp(W, params, bcs=bcs, solver_parameters=sp_dict)
    f1, f2 = sol.solution.split()
    return SolveResult(approach=approach, nx=-1, ny=-1, iteration_number=int(sol.iteration_number),
                       residual_error=float(sol.residual_error), fields=(f1, f2))


def assemble_monolithic_matrix(W, params: Optional[DPPParameters] = None,
                               bcs: Optional[List[fd.DirichletBC]] = None) -> Tuple[sp.csr_matrix, int, int]:
    params = params or default_model_params()
    bcs = bcs or default_bcs(W)
    a, _L = dpp_form(W, params)
    md = conditioning.get_matrix_data_from_form(a, bcs)
    return md.sparse_csr_data, W.sub(0).dim(), W.sub(1).dim()


def estimate_condition_numbers(W, params: Optional[DPPParameters] = None, bcs: Optional[List[fd.DirichletBC]] = None,
                               num_of_factors: Optional[int] = 50, use_sparse: bool = True) -> Dict[str, float]:
    csr, n0, n1 = assemble_monolithic_matrix(W, params=params, bcs=bcs)
    cond = conditioning.calculate_condition_number
    kw = {"num_singular_values": num_of_factors, "use_sparse": use_sparse}
    return {"monolithic": cond(csr, **kw), "macro": cond(csr[:n0, :n0].tocsr(), **kw),
            "micro": cond(csr[n0:n0 + n1, n0:n0 + n1].tocsr(), **kw)}


def l2_errors_against_reference(W, fields: Tuple[fd.Function, fd.Function],
                                ref_fields: Tuple[fd.Function, fd.Function]) -> Tuple[float, float]:
    """L2 norms of p1 - r1 and p2 - r2 for CG-1 fields on the mesh of W (reference iterative_bench.py:340-362:
    sqrt(assemble((p - r)^2 dx))).  For CG-1 differences that integral is exactly d^T M d with the mass matrix M,
    which the device assembles (pph_get_csr K/M export)."""
    from . import _ffi

    ctx = W.mesh().context()
    M = ctx.csr(_ffi.MAT_M)
    out = []
    for p, r in zip(fields, ref_fields):
        d = np.asarray(p.vector(), dtype=np.float64) - np.asarray(r.vector(), dtype=np.float64)
        out.append(float(np.sqrt(max(d @ (M @ d), 0.0))))
    return out[0], out[1]
