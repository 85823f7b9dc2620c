"""The reference's whole published solver sweep on the HIP path:
every row of notebooks/results-conforming-2d/petsc_profiling/petsc_perf_breakdown.csv (35 rows: 2D Q1, N = 4 .. 256) and
notebooks/results-conforming-3d/petsc_profiling/petsc_perf_breakdown_3d.csv (40 rows: Kuhn tets, nx = 4 .. 40), values
in tests/golden/reference_goldens.json, run through the public harness (perphil_amd.profiling_3d: run_perf_once /
run_perf_once_3d -> solve_dpp with the reference's option dictionaries).

Pinned: dofs and num_cells exact; plain GMRES(30) iteration counts (PETSc and the device use classical Gram-Schmidt: exact
on the small meshes, within 1 % where thousands of restarts amplify the last bits) and final residual norms; GMRES + ILU(0)
counts exact on 2D Q1 (the lexicographic numbering gives the reference's factors there; on tets the DMPlex numbering is
not reproducible: count unpinned, the solve must converge); field-split GMRES = 4 outer iterations (both block-solver
variants) with PETSc's residual norm; MUMPS rows = 1 iteration, residual 0.  tools/r3_reference_sweep.py writes the same
rows, with timings, to profiles/r03_reference_sweep_{2d,3d}.csv."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as _f:
    _G = json.load(_f)
ROWS_2D = [(r["approach"], int(r["nx"])) for r in _G["G7_G9_perf_2d_q1"]]
ROWS_3D = [(r["approach"], int(r["nx"])) for r in _G["G6_G9_perf_3d_tets"]]


def _golden(key, approach, nx):
    return next(r for r in _G[key] if r["approach"] == approach and int(r["nx"]) == nx)


def check_row(row, g, dim):
    """Assertions of one sweep row against the reference's (shared with tools/r3_reference_sweep.py)."""
    assert row["dofs"] == g["dofs"] and row["num_cells"] == g["num_cells"]
    ap, its, ref = g["approach"], row["iterations"], g["iterations"]
    if ap == "GMRES":
        assert abs(its - ref) <= max(1, ref // 100), (its, ref)
        # (the norm of the LAST iterate: on the two smallest meshes it is far below the tolerance already and follows the
        # last bits of the orthogonalisation and of the matrix entries - 2.3e-7 (round 3) / 5.1e-7 (round 4: canonical edges) against
        # 3.1e-7, i.e. 1e-12 of the first residual; from N = 16 on the norms agree to three digits)
        assert row["residual"] == pytest.approx(g["residual"], rel=1.0 if g["dofs"] < 300 else 5e-3)
    elif ap == "GMRES + ILU PC":
        if dim == 2:
            assert its == ref, (its, ref)
            assert row["residual"] == pytest.approx(g["residual"], rel=1e-3)
        else:
            assert 1 <= its <= 2 * ref + 4, (its, ref)    # ordering-dependent factors: parity of the count unpinned
    elif ap.startswith("Scale-Splitting GMRES"):
        assert its == ref == 4
        assert row["residual"] == pytest.approx(g["residual"], rel=1e-3)
    elif ap == "Monolithic LU with MUMPS":
        assert its == 1 and row["residual"] == 0.0
    else:
        raise AssertionError(f"unknown approach {ap}")


@pytest.mark.parametrize("approach,nx", ROWS_2D)
def test_reference_sweep_2d_row(approach, nx):
    from perphil_amd.iterative_bench import Approach
    from perphil_amd.profiling_3d import run_perf_once

    row = run_perf_once(nx, nx, Approach(approach), eager=False, repeats=1)
    check_row(row, _golden("G7_G9_perf_2d_q1", approach, nx), 2)


@pytest.mark.parametrize("approach,nx", ROWS_3D)
def test_reference_sweep_3d_row(approach, nx):
    from perphil_amd.iterative_bench import Approach
    from perphil_amd.profiling_3d import run_perf_once_3d

    row = run_perf_once_3d(nx, Approach(approach), eager=False, repeats=1)
    check_row(row, _golden("G6_G9_perf_3d_tets", approach, nx), 3)


@pytest.mark.parametrize("N", [int(r["N"]) for r in _G["G5_conditioning_3d_hex"] if int(r["N"]) <= 12])
def test_G5_conditioning_3d_hex_through_the_public_api(N):
    """notebooks/results-conforming-3d/conditioning/conditioning_3d.csv:2-8 (reference
    notebooks/condition-number-study-3d.py: UnitCubeMesh(N, N, N, hexahedral=True), manufactured Dirichlet data) through
    get_matrix_data_from_form / estimate_condition_numbers on the device-assembled matrix: kappa of the monolithic, macro
    and micro systems to 1e-9 of the stored values for N = 4 .. 12 (dense SVD of the exported CSR: 4 394 dofs at N = 12;
    N = 14 and 16 - 6 750 / 9 826 dofs, one and three and a half minutes of host SVD - are left to
    tools/r3_conditioning_3d.py, profiles/r03_conditioning_3d.csv), block sizes exact; the ARPACK branch (extreme
    singular values only) to 1e-6 on the two smallest meshes."""
    import perphil_amd as pa
    from perphil_amd import fd
    from perphil_amd.iterative_bench import estimate_condition_numbers
    from perphil_amd.manufactured_solutions import exact_expressions_3d

    g = next(r for r in _G["G5_conditioning_3d_hex"] if int(r["N"]) == N)
    mesh = fd.UnitCubeMesh(N, N, N, hexahedral=True)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    params = pa.DPPParameters(k1=1.0, k2=1.0 / 1e2, beta=1.0, mu=1.0)
    _u1, p1e, _u2, p2e = exact_expressions_3d(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    assert W.dim() == int(g["n_dofs"]) and W.sub(0).dim() == int(g["n0"]) and W.sub(1).dim() == int(g["n1"])
    cols = (("monolithic", "cond_monolithic"), ("macro", "cond_macro"), ("micro", "cond_micro"))
    # the reference's call (notebooks/condition-number-study-3d.py:47-48, 89-99): num_of_factors = 0, use_sparse = True,
    # i.e. the dense SVD branch (conditioning.py:134-154)
    ref_call = estimate_condition_numbers(W, params=params, bcs=bcs, use_sparse=True, num_of_factors=0)
    for k, col in cols:
        assert ref_call[k] == pytest.approx(g[col], rel=1e-9), (k, ref_call[k], g[col])
    if N <= 6:
        sparse = estimate_condition_numbers(W, params=params, bcs=bcs, use_sparse=True, num_of_factors=50)   # ARPACK branch
        for k, col in cols:
            assert sparse[k] == pytest.approx(g[col], rel=1e-6), (k, sparse[k], g[col])


# ---- notebooks/results-conforming-2d/convergence.csv + convergence_eoc.csv through convergence_2d ----
ROWS_CONV = [(r["solver"], int(r["N"])) for r in _G["G10_convergence_2d"]]
_CONV_DONE = {}


def _convergence_row(solver, N):
    import perphil_amd as pa
    from perphil_amd import convergence_2d as c2
    from perphil_amd.iterative_bench import Approach

    if (solver, N) not in _CONV_DONE:
        spec = c2.approach_solvers([Approach(solver)])[0]
        _CONV_DONE[(solver, N)] = c2.run_one(N=N, solver=spec, quad=True, degree=1, params=pa.DPPParameters())
    return _CONV_DONE[(solver, N)]


def check_convergence_row(row, g):
    """One row of the h-convergence study against the reference's (shared with tools/r3_convergence_2d.py): iteration
    counts exact (plain GMRES included: 10 ... 11 765); the four error norms to 2e-9 from N = 8 on and to 1e-6 on the 4 x 4
    mesh (measured: 7.7e-8 there, <= 1.2e-10 elsewhere - the reference integrates the UFL error with the quadrature degree
    its form compiler estimates, the device with 6 Gauss points per direction, and the difference is the coarse-mesh
    quadrature error of the former).  The rows of the UNPRECONDITIONED GMRES carry an iterate that met ksp_rtol 1e-8 and
    nothing more: its error norms follow the last bits of the matrix (round 4, canonical edges: 2.2e-9 at N = 16) - 2e-8 there."""
    assert list(row.keys()) == _G["convergence_csv_columns"]
    assert row["h"] == g["h"] and row["degree"] == g["degree"] and row["quad"] == g["quad"]
    name, its, ref = g["solver"], row["it"], int(g["it"])
    if name == "GMRES":
        assert its == ref, (its, ref)
        assert row["res"] == pytest.approx(g["res"], rel=1.0 if N_dofs(g) < 300 else 5e-3)   # (N = 4: 1e-12 of the first residual)
    elif name == "Monolithic LU with MUMPS":
        assert its == ref == 1
    else:
        assert its == ref, (its, ref)
        assert row["res"] == pytest.approx(g["res"], rel=1e-3)
    for k in ("e1_L2", "e2_L2", "e1_H1s", "e2_H1s"):
        assert row[k] == pytest.approx(g[k], rel=1e-6 if int(g["N"]) == 4 else (2e-8 if name == "GMRES" else 2e-9)), (k, row[k], g[k])


def N_dofs(g):
    return 2 * (int(g["N"]) + 1) ** 2


@pytest.mark.parametrize("solver,N", ROWS_CONV)
def test_convergence_2d_row(solver, N):
    g = next(r for r in _G["G10_convergence_2d"] if r["solver"] == solver and int(r["N"]) == N)
    check_convergence_row(_convergence_row(solver, N), g)


def test_convergence_2d_observed_orders():
    """convergence_eoc.csv: the least-squares orders over N = 4 .. 128 (1.939 in L2, 0.9448 in the H1 seminorm) for every
    solver and both pressures, to 1e-7 (measured 8e-9)."""
    from perphil_amd import convergence_2d as c2

    rows = [_convergence_row(s, N) for s, N in ROWS_CONV]
    got = {(r["solver"], r["err"]): r["slope"] for r in c2.observed_orders(rows)}
    assert len(got) == len(_G["G10_convergence_2d_eoc"]) == 20
    for r in _G["G10_convergence_2d_eoc"]:
        assert got[(r["solver"], r["err"])] == pytest.approx(r["slope"], rel=1e-7), r
