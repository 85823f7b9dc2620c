"""
Performance harness — mirror of reference ``src/perphil/experiments/petsc_profiling_3d.py``
(``run_perf_once_3d`` :43-200, ``run_perf_sweep_3d`` :203-235, ``save_perf_csv`` :238-240) and of the 2D entry points
``run_perf_once`` / ``run_perf_sweep`` of ``src/perphil/experiments/petsc_profiling.py`` (:637-855): same protocol
(one warm-up solve, ``repeats`` timed solves, one more solve for iteration count / residual) and the same
flat row schema, so the reference notebooks' plotting code can consume the output unchanged.  PETSc's
``-log_view`` events become HIP-event timers of the library:

    time_SNESJacobianEval <- K/M integration + Dirichlet elimination / block formation
    time_KSPSolve         <- the solve (Krylov / Picard, preconditioner set-up included)
    time_MatMult          <- sum of the event-timed CSR SpMV launches;  flops_MatMult = 2 nnz per launch
    time_PCSetUp / time_PCApply / time_SNESFunctionEval / MatAssembly*: not separated (0.0)
"""
from __future__ import annotations

import resource
import time
from typing import Any, Dict, List, Optional

from . import fd
from .iterative_bench import Approach, params_for
from .manufactured_solutions import exact_expressions_3d
from .parameters import DPPParameters
from .solver import solve_dpp, solve_dpp_nonlinear

_EVENTS = ["SNESFunctionEval", "MatAssemblyEnd", "PCSetUp", "SNESJacobianEval", "PCApply", "KSPSolve", "SNESSolve",
           "MatMult", "MatAssemblyBegin"]


def _default_model_params() -> DPPParameters:
    return DPPParameters(k1=1.0, k2=1.0 / 1e2, beta=1.0, mu=1.0)


def _run_perf(mesh, W, bcs, params, approach: Approach, eager: bool, repeats: int, backend: str, nx: int, ny: int) -> Dict[str, Any]:
    """The measurement protocol both harnesses share (reference petsc_profiling.py:700-800, petsc_profiling_3d.py:73-200):
    optional warm-up solve, `repeats` timed solves with re-assembly, one more solve for iterations / residual."""
    solve = solve_dpp_nonlinear if approach == Approach.PICARD_MUMPS else solve_dpp
    opts = params_for(approach)
    ctx = mesh.context()
    if eager:
        solve(W, params, bcs=bcs, solver_parameters={**opts})
    rss_before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    ctx.set_option("time_spmv", 1)
    times = {e: 0.0 for e in _EVENTS}
    flops = {e: 0.0 for e in _EVENTS}
    t0 = time.perf_counter()
    for _ in range(max(1, repeats)):
        ctx.set_option("invalidate_KM", 1)   # the reference re-assembles the Jacobian in every solve
        sol = solve(W, params, bcs=bcs, solver_parameters={**opts})
        tm = sol.info["timers"]
        times["SNESJacobianEval"] += 1e-3 * (tm["assemble_ms"] + tm["bc_blocks_ms"])
        times["KSPSolve"] += 1e-3 * tm["solve_ms"]
        times["MatMult"] += 1e-3 * (tm["spmv_ms"] + tm["spmv_dot_ms"])
        # bytes = 12 nnz + 20 nrows per launch  =>  flops = 2 nnz ~ bytes / 6 (row term neglected)
        flops["MatMult"] += (tm["spmv_bytes"] + tm["spmv_dot_bytes"]) / 6.0
    wall = time.perf_counter() - t0
    ctx.set_option("time_spmv", 0)
    times["SNESSolve"] = times["SNESJacobianEval"] + times["KSPSolve"]
    sol = solve(W, params, bcs=bcs, solver_parameters={**opts})
    rss_after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    row: Dict[str, Any] = {
        "approach": approach.value, "nx": nx, "ny": ny, "dofs": int(W.dim()), "num_cells": int(mesh.num_cells()),
        "iterations": int(sol.iteration_number), "residual": float(sol.residual_error),
        "time_total": float(wall / max(1, repeats)), "time_total_repeats": float(wall),
    }
    for k in _EVENTS:
        row[f"time_{k}"] = float(times[k])
    for k in _EVENTS:
        row[f"flops_{k}"] = float(flops[k])
        row[f"mflops_{k}"] = float(flops[k] / times[k] / 1e6) if times[k] > 0 else 0.0
    row["flops_total"] = float(sum(flops.values()))
    row["mem_rss_peak_kb"] = float(rss_after)
    row["mem_rss_delta_kb"] = float(max(0, rss_after - rss_before))
    row["backend"] = backend
    row["repeats"] = repeats
    return row


def run_perf_once_3d(nx: int, approach: Approach, eager: bool = True, logical_events: Optional[List[str]] = None,
                     repeats: int = 5, backend: str = "hip-events", hexahedral: bool = False) -> Dict[str, Any]:
    mesh = fd.UnitCubeMesh(nx, nx, nx, hexahedral=hexahedral)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    params = _default_model_params()
    _u1, p1e, _u2, p2e = exact_expressions_3d(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    return _run_perf(mesh, W, bcs, params, approach, eager, repeats, backend, nx, nx)


def run_perf_once(nx: int, ny: int, approach: Approach, eager: bool = True, logical_events: Optional[List[str]] = None,
                  force_nonzero_rhs: bool = False, bc_values: Optional[List[float]] = None, repeats: int = 5,
                  backend: str = "hip-events", use_manufactured: bool = True) -> Dict[str, Any]:
    """2D counterpart (reference src/perphil/experiments/petsc_profiling.py:637-800): Q1 quadrilaterals, manufactured
    pressures as Dirichlet data by default, constants (`bc_values`, default [1, 0]) with `force_nonzero_rhs`, else
    homogeneous.  Returns the flat row of the reference's `PerfResult.to_dict()`."""
    from .iterative_bench import build_mesh, build_spaces, default_bcs, default_model_params
    from .manufactured_solutions import exact_expressions

    mesh = build_mesh(nx, ny, quadrilateral=True)
    _, _, W = build_spaces(mesh)
    params = default_model_params()
    if use_manufactured:
        _u1, p1e, _u2, p2e = exact_expressions(mesh, params)
        bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    elif force_nonzero_rhs:
        v = bc_values or [1.0, 0.0]
        bcs = [fd.DirichletBC(W.sub(0), fd.Constant(v[0]), "on_boundary"), fd.DirichletBC(W.sub(1), fd.Constant(v[1]), "on_boundary")]
    else:
        bcs = default_bcs(W)
    row = _run_perf(mesh, W, bcs, params, approach, eager, repeats, backend, nx, ny)
    row["metadata"] = {"backend": backend, "repeats": repeats, "library": "perphil_amd / libperphil_hip.so"}
    return row


def run_perf_sweep(mesh_sizes: List[int], approaches: List[Approach], logical_events: Optional[List[str]] = None,
                   eager: bool = True, force_nonzero_rhs: bool = True, bc_values: Optional[List[float]] = None,
                   repeats: int = 5, backend: str = "hip-events", use_manufactured: bool = True):
    """reference petsc_profiling.py:803-855 (ny = nx)."""
    import pandas as pd

    rows = [run_perf_once(nx, nx, ap, eager=eager, force_nonzero_rhs=force_nonzero_rhs, bc_values=bc_values, repeats=repeats,
                          backend=backend, use_manufactured=use_manufactured) for nx in mesh_sizes for ap in approaches]
    return pd.DataFrame(rows)


def run_perf_sweep_3d(mesh_sizes: List[int], approaches: List[Approach], logical_events: Optional[List[str]] = None,
                      eager: bool = True, repeats: int = 5, backend: str = "hip-events", hexahedral: bool = False):
    import pandas as pd

    rows = [run_perf_once_3d(nx, ap, eager=eager, repeats=repeats, backend=backend, hexahedral=hexahedral)
            for nx in mesh_sizes for ap in approaches]
    return pd.DataFrame(rows)


def save_perf_csv(df, path: str) -> None:
    import os

    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    df.to_csv(path, index=False)


def save_perf_json(df, path: str) -> None:
    """Rows of the sweep as a JSON list of records (reference petsc_profiling_3d.py:238-241)."""
    import json
    import os

    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(df.to_dict(orient="records"), f, indent=2)
