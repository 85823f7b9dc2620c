"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's goldens.

Bars: bit-exact for index work (cell->dof map, CSR row pointers and columns, boundary
sets); fp64 values to 1e-12 relative to the largest entry (element integrals are summed in a
different order on the GPU: atomics + shuffles); iterative solutions to the tolerance the solver was
asked for; iteration counts equal to the goldens within +-1 (different but equivalent
orthogonalisation order)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import dpp_oracle as o
from oracle import dpp_mg_oracle as mgo

pytestmark = pytest.mark.gpu

P = o.Params(k1=1.0, k2=0.01, beta=1.0, mu=1.0)
VAL_RTOL = 1e-12

CASES = [
    (2, o.CELL_QUAD, 5, 3, 0),
    (2, o.CELL_TRI, 4, 6, 0),
    (3, o.CELL_HEX, 3, 4, 5),
    (3, o.CELL_TET, 4, 3, 2),
    (2, o.CELL_QUAD, 16, 16, 0),   # BASELINE config 1
    (3, o.CELL_HEX, 9, 9, 9),      # more than one workgroup batch, ragged last batch
]


def _ffi():
    from perphil_amd import _ffi

    return _ffi


def _setup(make, dim, kind, nx, ny, nz, params=P, mms=True, monolithic=True):
    ctx = make()
    ctx.mesh_build(dim, kind, nx, ny, nz)
    om = o.build_mesh(dim, kind, nx, ny, nz)
    osys = o.build_system(om, params, mms=mms)
    b = o.boundary_nodes(om)
    if mms:
        e1, e2 = o.exact_pressures(om.coords, params)
        ctx.set_dirichlet(0, b, e1[b])
        ctx.set_dirichlet(1, b, e2[b])
    else:
        ctx.set_dirichlet(0, b, np.zeros(len(b)))
        ctx.set_dirichlet(1, b, np.zeros(len(b)))
    ctx.assemble(params.k1, params.k2, params.beta, params.mu, monolithic=monolithic)
    return ctx, om, osys


def _cfg(**kw):
    f = _ffi()
    c = f.SolverCfg()
    c.ksp_type, c.pc_type, c.restart, c.max_it = f.KSP_GMRES, f.PC_NONE, 30, 50000
    c.rtol, c.atol = 1e-8, 1e-12
    c.inner_ksp_type, c.inner_pc_type, c.inner_max_it = f.KSP_CG, f.PC_JACOBI, 50000
    c.inner_rtol, c.inner_atol = 1e-12, 1e-300
    c.picard, c.picard_rtol, c.picard_atol, c.picard_max_it = 0, 1e-8, 1e-12, 100
    c.mg_smooth = 2
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _structural_pattern(om):
    m = om.cells.shape[1]
    rows = np.repeat(om.cells, m, axis=1).ravel()
    cols = np.tile(om.cells, (1, m)).ravel()
    S = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(om.num_nodes,) * 2).tocsr()
    S.sum_duplicates()
    S.sort_indices()
    return S


@pytest.mark.parametrize("dim,kind,nx,ny,nz", CASES)
def test_mesh_dofmap_pattern_bit_exact(gpu_ctx_factory, dim, kind, nx, ny, nz):
    ctx = gpu_ctx_factory()
    ctx.mesh_build(dim, kind, nx, ny, nz)
    om = o.build_mesh(dim, kind, nx, ny, nz)
    assert (ctx.n, ctx.ncell, ctx.m) == (om.num_nodes, om.num_cells, om.cells.shape[1])
    np.testing.assert_array_equal(ctx.dofmap(), om.cells)
    np.testing.assert_array_equal(ctx.coords(), om.coords)
    b = o.boundary_nodes(om)
    ctx.set_dirichlet(0, b, np.zeros(len(b)))
    ctx.set_dirichlet(1, b, np.zeros(len(b)))
    ctx.assemble(1.0, 0.01, 1.0, 1.0, monolithic=True)
    S = _structural_pattern(om)
    K = ctx.csr(_ffi().MAT_K)
    np.testing.assert_array_equal(K.indptr, S.indptr)
    np.testing.assert_array_equal(K.indices, S.indices)
    assert ctx.nnzb == S.nnz
    A = ctx.csr(_ffi().MAT_MONO)
    n = om.num_nodes
    Sm = sp.bmat([[S, S], [S, S]], format="csr")
    Sm.sort_indices()
    np.testing.assert_array_equal(A.indptr, Sm.indptr)
    np.testing.assert_array_equal(A.indices, Sm.indices)
    assert A.shape == (2 * n, 2 * n)


@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(2, o.CELL_TRI, 5, 4, 0), (3, o.CELL_TET, 4, 3, 5), (3, o.CELL_TET, 7, 7, 7)])
def test_simplex_scatter_and_gather_assembly_agree(gpu_ctx_factory, dim, kind, nx, ny, nz):
    f = _ffi()
    om = o.build_mesh(dim, kind, nx, ny, nz)
    Ko, Mo = o.assemble_scalar(om)
    out = []
    for mode in (0, 2, 2):
        ctx = gpu_ctx_factory()
        ctx.mesh_build(dim, kind, nx, ny, nz)
        ctx.set_option("asm_kernel", mode)
        b = o.boundary_nodes(om)
        ctx.set_dirichlet(0, b, np.zeros(len(b)))
        ctx.set_dirichlet(1, b, np.zeros(len(b)))
        ctx.assemble(1.0, 0.01, 1.0, 1.0, monolithic=False)
        K, M = ctx.csr(f.MAT_K), ctx.csr(f.MAT_M)
        assert abs(K - Ko).max() <= VAL_RTOL * abs(Ko).max() and abs(M - Mo).max() <= VAL_RTOL * abs(Mo).max()
        out.append((K.data.copy(), M.data.copy()))
    np.testing.assert_array_equal(out[1][0], out[2][0])   # gather: bitwise reproducible
    np.testing.assert_array_equal(out[1][1], out[2][1])


@pytest.mark.parametrize("nx,ny,nz", [(5, 3, 0), (3, 4, 5), (9, 9, 9)])
def test_scatter_and_gather_assembly_agree(gpu_ctx_factory, nx, ny, nz):
    """The deterministic node-centred gather kernel (default) and the cell-centred atomic scatter-add
    kernel integrate the same element matrices; the gather result is bitwise reproducible."""
    f = _ffi()
    dim, kind = (2, o.CELL_QUAD) if nz == 0 else (3, o.CELL_HEX)
    om = o.build_mesh(dim, kind, nx, ny, nz)
    Ko, Mo = o.assemble_scalar(om)
    res = {}
    for mode in (0, 1, 1, 2, 2):
        ctx = gpu_ctx_factory()
        ctx.mesh_build(dim, kind, nx, ny, nz)
        ctx.set_option("asm_kernel", mode)
        b = o.boundary_nodes(om)
        ctx.set_dirichlet(0, b, np.zeros(len(b)))
        ctx.set_dirichlet(1, b, np.zeros(len(b)))
        ctx.assemble(1.0, 0.01, 1.0, 1.0, monolithic=False)
        K, M = ctx.csr(f.MAT_K), ctx.csr(f.MAT_M)
        assert abs(K - Ko).max() <= VAL_RTOL * abs(Ko).max() and abs(M - Mo).max() <= VAL_RTOL * abs(Mo).max()
        res.setdefault(mode, []).append((K.data.copy(), M.data.copy()))
    np.testing.assert_array_equal(res[1][0][0], res[1][1][0])   # bitwise reproducible
    np.testing.assert_array_equal(res[1][0][1], res[1][1][1])
    np.testing.assert_array_equal(res[2][0][0], res[2][1][0])
    np.testing.assert_array_equal(res[2][0][1], res[2][1][1])


@pytest.mark.parametrize("dim,kind,nx,ny,nz", CASES)
def test_K_M_blocks_rhs_match_oracle(gpu_ctx_factory, dim, kind, nx, ny, nz):
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, dim, kind, nx, ny, nz)
    Ko, Mo = o.assemble_scalar(om)
    for which, ref in ((f.MAT_K, Ko), (f.MAT_M, Mo)):
        G = ctx.csr(which)
        assert abs(G - ref).max() <= VAL_RTOL * abs(ref).max()
    n = osys.n
    A = ctx.csr(f.MAT_MONO)
    assert abs(A - osys.A).max() <= VAL_RTOL * abs(osys.A).max()
    for which, ref in ((f.MAT_A11, osys.A[:n, :n]), (f.MAT_A22, osys.A[n:, n:]), (f.MAT_A12, osys.A[:n, n:]),
                       (f.MAT_A21, osys.A[n:, :n])):
        G = ctx.csr(which)
        assert abs(G - ref).max() <= VAL_RTOL * max(abs(ref).max(), 1.0)
    rhs, u0 = ctx.rhs()
    np.testing.assert_array_equal(u0, osys.u0)
    assert np.abs(rhs - osys.rhs).max() <= VAL_RTOL * np.abs(osys.rhs).max()
    # Dirichlet rows/cols: exact identity
    Ad = A.toarray()
    for d in osys.bc_dofs[:: max(1, len(osys.bc_dofs) // 50)]:
        row = Ad[d].copy(); row[d] -= 1.0
        col = Ad[:, d].copy(); col[d] -= 1.0
        assert not row.any() and not col.any()


def test_G1_initial_residual_and_spmv(gpu_ctx_factory, goldens):
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, 2, o.CELL_QUAD, 10, 10, 0)
    rhs, _ = ctx.rhs()
    assert np.linalg.norm(rhs) == pytest.approx(goldens["G1_initial_residual_10x10"], rel=1e-12)
    rng = np.random.default_rng(20260313)
    x = rng.uniform(-1, 1, 2 * osys.n)
    ref = osys.A @ x
    # the CSR SpMV kernel at every lanes-per-row setting (its round-1 A/B variants are gone from the tree: DESIGN.md appendix A)
    for lanes in (0, 4, 8, 16, 32, 64):
        ctx.set_option("spmv_lanes", lanes)
        y = ctx.spmv(f.MAT_MONO, x)
        assert np.abs(y - ref).max() <= 1e-13 * np.abs(ref).max(), lanes
    ctx.set_option("spmv_lanes", 0)
    with pytest.raises(ValueError):
        ctx.set_option("spmv_kernel", 17)       # (an option the library no longer knows)
    xs = x[: osys.n]
    for which, ref in ((f.MAT_A11, osys.A[: osys.n, : osys.n]), (f.MAT_A21, osys.A[osys.n:, : osys.n])):
        y = ctx.spmv(which, xs)
        r = ref @ xs
        assert np.abs(y - r).max() <= 1e-13 * max(np.abs(r).max(), 1.0)


def _perf(goldens, key, approach, nx):
    return next(r for r in goldens[key] if r["approach"] == approach and r["nx"] == nx)


@pytest.mark.parametrize("nx", [4, 8, 16])
def test_G7_plain_gmres_iterations_2d(gpu_ctx_factory, goldens, nx):
    g = _perf(goldens, "G7_G9_perf_2d_q1", "GMRES", nx)
    ctx, om, osys = _setup(gpu_ctx_factory, 2, o.CELL_QUAD, nx, nx, 0)
    x, info, hist = ctx.solve(_cfg(), hist_cap=400)
    assert 2 * ctx.n == g["dofs"]
    assert abs(info.iterations - g["iterations"]) <= 1 and info.converged
    ref = o.gmres(osys.A, osys.rhs)
    k = min(20, len(hist) - 2, len(ref.history) - 2)  # the last entries depend on the orthogonalisation order
    np.testing.assert_allclose(hist[:k], ref.history[:k], rtol=1e-8)
    ud = o.solve_direct(osys)
    assert np.abs(x - ud).max() / np.abs(ud).max() < 1e-6


@pytest.mark.parametrize("nx", [4, 8])
def test_G6_plain_gmres_iterations_3d_tets(gpu_ctx_factory, goldens, nx):
    g = _perf(goldens, "G6_G9_perf_3d_tets", "GMRES", nx)
    ctx, om, osys = _setup(gpu_ctx_factory, 3, o.CELL_TET, nx, nx, nx)
    x, info, _ = ctx.solve(_cfg())
    assert 2 * ctx.n == g["dofs"] and ctx.ncell == g["num_cells"]
    assert abs(info.iterations - g["iterations"]) <= max(1, g["iterations"] // 100)
    assert info.resnorm == pytest.approx(g["residual"], rel=0.15)


def test_G2_direct_equivalent_slice(gpu_ctx_factory, goldens):
    """solve_dpp with LINEAR_SOLVER_PARAMS through the public API (BASELINE config-1 style)."""
    import perphil_amd as pa
    from perphil_amd import fd, solver_parameters as spar

    mesh = pa.create_mesh(10, 10, quadrilateral=True)
    _, V = pa.create_function_spaces(mesh)
    W = V * V
    params = pa.DPPParameters(k1=1.0, k2=1 / 1e2, beta=1.0, mu=1)
    _, p1e, _, p2e = pa.exact_expressions(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    sol = pa.solve_dpp(W, params, bcs, solver_parameters=spar.LINEAR_SOLVER_PARAMS)
    assert isinstance(sol, pa.Solution) and sol.iteration_number == 1 and sol.residual_error == 0.0
    g = goldens["G2_slice_x05_monolithic_10x10"]
    p1 = [sol.solution.sub(0).at((0.5, y)) for y in g["y"]]
    p2 = [sol.solution.sub(1).at((0.5, y)) for y in g["y"]]
    np.testing.assert_allclose(p1, g["p1"], rtol=5e-9)
    np.testing.assert_allclose(p2, g["p2"], rtol=5e-9)
    om = o.build_mesh(2, o.CELL_QUAD, 10, 10)
    ud = o.solve_direct(o.build_system(om, P))
    assert np.abs(sol.solution.vector() - ud).max() / np.abs(ud).max() < 1e-10


@pytest.mark.parametrize("pc", ["jacobi", "block2", "none"])
def test_cg_matches_oracle(gpu_ctx_factory, pc):
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, 3, o.CELL_HEX, 6, 6, 6)
    code = {"jacobi": f.PC_JACOBI, "block2": f.PC_BLOCK2, "none": f.PC_NONE}[pc]
    x, info, hist = ctx.solve(_cfg(ksp_type=f.KSP_CG, pc_type=code), hist_cap=2000)
    apply = {"jacobi": o.jacobi_apply(osys.A), "block2": o.block2_jacobi_apply(osys.A, osys.n), "none": None}[pc]
    ref = o.pcg(osys.A, osys.rhs, apply)
    assert abs(info.iterations - ref.its) <= 2 and info.converged
    np.testing.assert_allclose(hist[:10], ref.history[:10], rtol=1e-9)
    ud = o.solve_direct(osys)
    assert np.abs(x - ud).max() / np.abs(ud).max() < 1e-6


@pytest.mark.parametrize("dim,kind,nx", [(2, o.CELL_QUAD, 16), (3, o.CELL_TET, 8), (3, o.CELL_HEX, 8)])
def test_G9_fieldsplit_gmres(gpu_ctx_factory, goldens, dim, kind, nx):
    """GMRES + multiplicative field-split with (near-)exact block solves: 4 iterations at every
    mesh size in the reference; preconditioned residuals match PETSc's."""
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, dim, kind, nx, nx, nx if dim == 3 else 0)
    for inner_pc in (f.PC_JACOBI, f.PC_MG):
        x, info, hist = ctx.solve(_cfg(pc_type=f.PC_FIELDSPLIT, inner_pc_type=inner_pc, inner_rtol=1e-12), hist_cap=16)
        assert info.iterations == 4 and info.converged
        ref = o.gmres(osys.A, osys.rhs, o.fieldsplit_multiplicative_apply(osys.A, osys.n))
        np.testing.assert_allclose(hist, ref.history, rtol=1e-5)
        if kind != o.CELL_HEX:
            key = "G7_G9_perf_2d_q1" if dim == 2 else "G6_G9_perf_3d_tets"
            g = _perf(goldens, key, "Scale-Splitting GMRES", nx)
            assert info.iterations == g["iterations"]
            assert info.resnorm == pytest.approx(g["residual"], rel=1e-4)
        ud = o.solve_direct(osys)
        assert np.abs(x - ud).max() / np.abs(ud).max() < 1e-7


@pytest.mark.parametrize("inner", ["jacobi", "mg"])
def test_picard_fixed_point(gpu_ctx_factory, goldens, inner):
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, 2, o.CELL_QUAD, 10, 10, 0, monolithic=False)
    code = f.PC_JACOBI if inner == "jacobi" else f.PC_MG
    x, info, hist = ctx.solve(_cfg(picard=1, inner_pc_type=code, inner_rtol=1e-12), hist_cap=64)
    _, its, res, ohist = o.picard(osys)
    assert info.iterations == its and info.converged
    np.testing.assert_allclose(hist, ohist, rtol=1e-5)
    ud = o.solve_direct(osys)
    assert np.abs(x - ud).max() / np.abs(ud).max() < 1e-8
    g = goldens["G11_slice_x05_picard_10x10"]
    _, p1 = o.slice_along_x(om, x[: osys.n], 0.5)
    np.testing.assert_allclose(p1, g["p1"], rtol=3e-4)


@pytest.mark.parametrize("dim,kind,n,k2", [(3, o.CELL_HEX, 16, 1e-2), (3, o.CELL_HEX, 16, 1e-4), (3, o.CELL_TET, 8, 1e-2),
                                           (2, o.CELL_QUAD, 32, 1e-2), (2, o.CELL_TRI, 32, 1e-2)])
def test_mg_pcg_iterations_match_mg_oracle(gpu_ctx_factory, dim, kind, n, k2):
    """Multigrid-preconditioned CG block solves: same iteration counts as the NumPy restatement of
    the same cycle (inner iteration total of one field-split application = one Picard sweep)."""
    f = _ffi()
    params = o.Params(k1=1.0, k2=k2)
    nz = n if dim == 3 else 0
    ctx, om, osys = _setup(gpu_ctx_factory, dim, kind, n, n, nz, params=params, monolithic=False)
    x, info, hist = ctx.solve(_cfg(picard=1, picard_max_it=1, inner_pc_type=f.PC_MG, inner_rtol=1e-10),
                              hist_cap=4, raise_on_diverged=False)
    a, b, c = params.abc
    mask = np.zeros(osys.n, bool)
    mask[o.boundary_nodes(om)] = True
    nn = osys.n
    A = osys.A.tocsr()
    L1 = mgo.build_hierarchy(dim, kind, n, n, nz, a, b, mask)
    L2 = mgo.build_hierarchy(dim, kind, n, n, nz, c, b, mask)
    r1 = o.pcg(A[:nn, :nn], osys.rhs[:nn], lambda v: mgo.vcycle(L1, v, 2), rtol=1e-10)
    r2 = o.pcg(A[nn:, nn:], osys.rhs[nn:] - A[nn:, :nn] @ r1.x, lambda v: mgo.vcycle(L2, v, 2), rtol=1e-10)
    assert abs(info.inner_iterations - (r1.its + r2.its)) <= 1
    du = np.concatenate([r1.x, r2.x])
    assert np.abs(x - (osys.u0 + du)).max() <= 1e-7 * np.abs(osys.u0 + du).max()


def test_public_api_inexact_picard_preset(gpu_ctx_factory):
    """solve_dpp_nonlinear with PICARD_MG_INEXACT_SOLVER_PARAMS (the benchmark's algorithm through the public API)
    reaches the fixed point of the exact-block-solve preset with far fewer block iterations."""
    import perphil_amd as pa
    from perphil_amd import fd, solver_parameters as spar

    mesh = fd.UnitCubeMesh(16, 16, 16, hexahedral=True)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    params = pa.DPPParameters(k1=1.0, k2=0.01)
    _, p1e, _, p2e = pa.exact_expressions_3d(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    exact = pa.solve_dpp_nonlinear(W, params, bcs, solver_parameters=spar.PICARD_MG_SOLVER_PARAMS)
    inexact = pa.solve_dpp_nonlinear(W, params, bcs, solver_parameters=spar.PICARD_MG_INEXACT_SOLVER_PARAMS)
    a, b = exact.solution.vector(), inexact.solution.vector()
    assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max()
    assert inexact.info["inner_iterations"] < exact.info["inner_iterations"] / 2


def test_public_api_solvers_and_errors(gpu_ctx_factory):
    """solve_dpp / solve_dpp_nonlinear through the mirror of the reference API
    (reference solvers/_tests/test_solver.py:24-50)."""
    import perphil_amd as pa
    from perphil_amd import fd, solver_parameters as spar

    mesh = pa.create_mesh(2, 2, quadrilateral=True)
    _, V = pa.create_function_spaces(mesh)
    W = fd.MixedFunctionSpace((V, V))
    bcs = [fd.DirichletBC(W.sub(0), fd.Constant(0.0), "on_boundary"), fd.DirichletBC(W.sub(1), fd.Constant(0.0), "on_boundary")]
    sol = pa.solve_dpp(W, pa.DPPParameters(), bcs=bcs)
    assert isinstance(sol, pa.Solution) and sol.iteration_number >= 0
    assert not sol.solution.vector().any()   # homogeneous data -> zero solution
    sol = pa.solve_dpp_nonlinear(W, pa.DPPParameters(), bcs=bcs)
    assert isinstance(sol, pa.Solution) and sol.iteration_number >= 0
    # 3D manufactured problem through the API with the reference's option dictionaries
    mesh = fd.UnitCubeMesh(8, 8, 8)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    params = pa.DPPParameters(k1=1.0, k2=1.0 / 1e2, beta=1.0, mu=1.0)
    _, p1e, _, p2e = pa.exact_expressions_3d(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    om = o.build_mesh(3, o.CELL_TET, 8, 8, 8)
    ud = o.solve_direct(o.build_system(om, P))
    import warnings as _w

    for sp_dict, nonlinear in ((spar.PLAIN_GMRES_PARAMS, False), ({**spar.GMRES_PARAMS, **spar.FIELDSPLIT_LU_PARAMS}, False),
                               (spar.LINEAR_SOLVER_PARAMS, False), (spar.PICARD_LU_SOLVER_PARAMS, True),
                               (spar.CG_BLOCK_JACOBI_PARAMS, False), (spar.PICARD_MG_SOLVER_PARAMS, True),
                               # every option set of the reference runs its stated preconditioner, without a warning
                               (spar.GMRES_ILU_PARAMS, False), ({**spar.GMRES_PARAMS, **spar.FIELDSPLIT_GMRES_ILU_PARAMS}, False),
                               ({**spar.GMRES_PARAMS, **spar.FIELDSPLIT_GMRES_PARAMS}, False),
                               (spar.PICARD_GMRES_ILU_SOLVER_PARAMS, True), (spar.GMRES_JACOBI_PARAMS, False)):
        _w.simplefilter("error")     # a UserWarning (substituted algorithm) fails the test
        fn = pa.solve_dpp_nonlinear if nonlinear else pa.solve_dpp
        sol = fn(W, params, bcs, solver_parameters=sp_dict)
        err = np.abs(sol.solution.vector() - ud).max() / np.abs(ud).max()
        assert err < 1e-6, (sp_dict, err)
    _w.resetwarnings()
    ctx = mesh.context()
    f = _ffi()
    with pytest.raises(ValueError):
        ctx.solve(_cfg(restart=31))
    with pytest.raises(ValueError):
        ctx.set_dirichlet(0, np.array([10 ** 9]), np.array([0.0]))
    with pytest.raises(ValueError):
        gpu_ctx_factory().mesh_build(3, f.CELL_QUAD, 2, 2, 2)


def test_full_size_properties_64cubed(gpu_ctx_factory):
    """BASELINE config 2 size (64^3 Q1): size-independent properties instead of an oracle run."""
    f = _ffi()
    N = 64
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, f.CELL_HEX, N, N, N)
    n = ctx.n
    assert n == (N + 1) ** 3 and ctx.nnzb == (3 * N + 1) ** 3
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=True)
    b = mesh.boundary_nodes()
    X = mesh.node_coordinates(b)
    e1, e2 = o.exact_pressures(X, P)
    ctx.set_dirichlet(0, b, e1)
    ctx.set_dirichlet(1, b, e2)
    ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=True)
    ones = np.ones(n)
    assert np.abs(ctx.spmv(f.MAT_K, ones)).max() < 1e-12          # constants in the kernel of K
    assert ctx.spmv(f.MAT_M, ones).sum() == pytest.approx(1.0, rel=1e-12)   # volume of the unit cube
    rng = np.random.default_rng(7)
    x, y = rng.uniform(-1, 1, 2 * n), rng.uniform(-1, 1, 2 * n)
    Ax, Ay = ctx.spmv(f.MAT_MONO, x), ctx.spmv(f.MAT_MONO, y)
    assert abs(y @ Ax - x @ Ay) <= 1e-11 * abs(y @ Ax)            # symmetry
    assert x @ Ax > 0                                             # positive definite
    bc = np.concatenate([b, b + n])
    np.testing.assert_array_equal(Ax[bc], x[bc])                  # identity rows on Dirichlet dofs
    # field-split GMRES with multigrid block solves: 4 iterations also at this size (G9)
    xs, info, _ = ctx.solve(_cfg(pc_type=f.PC_FIELDSPLIT, inner_pc_type=f.PC_MG, inner_rtol=1e-12))
    assert info.iterations == 4 and info.converged
    r, u0 = ctx.rhs()
    res = r - ctx.spmv(f.MAT_MONO, xs - u0)
    assert np.linalg.norm(res) <= 1e-7 * np.linalg.norm(r)
    # the Picard loop reaches the same fixed point
    xp, pinfo, _ = ctx.solve(_cfg(picard=1, inner_pc_type=f.PC_MG, inner_rtol=1e-10, picard_rtol=1e-10))
    assert pinfo.converged and np.abs(xp - xs).max() <= 1e-7 * np.abs(xs).max()
    # nodal error of the CG-1 solution against the manufactured solution is O(h^2)-small
    Xall = mesh.node_coordinates()
    ex1, ex2 = o.exact_pressures(Xall, P)
    assert np.abs(xs[:n] - ex1).max() / np.abs(ex1).max() < 5e-3
    assert np.abs(xs[n:] - ex2).max() / np.abs(ex2).max() < 5e-3


@pytest.mark.parametrize("hexa,n", [(False, 12), (True, 12)])
def test_high_contrast_gmres_fieldsplit(gpu_ctx_factory, hexa, n):
    """BASELINE config 5 in small: k1/k2 = 1e4 (eta = 100, boundary data up to ~1e47), GMRES with the
    multiplicative field-split preconditioner and multigrid-CG block solves, through the public API."""
    import perphil_amd as pa
    from perphil_amd import fd, solver_parameters as spar

    mesh = fd.UnitCubeMesh(n, n, n, hexahedral=hexa)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    params = pa.DPPParameters(k1=1.0, k2=1e-4, beta=1.0, mu=1.0)
    _, p1e, _, p2e = pa.exact_expressions_3d(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    sol = pa.solve_dpp(W, params, bcs, solver_parameters=spar.FIELDSPLIT_MG_PARAMS)
    om = o.build_mesh(3, o.CELL_HEX if hexa else o.CELL_TET, n, n, n)
    osys = o.build_system(om, o.Params(k1=1.0, k2=1e-4))
    ud = o.solve_direct(osys)
    assert sol.iteration_number <= 6
    assert np.abs(sol.solution.vector() - ud).max() / np.abs(ud).max() < 1e-7
    ref = o.gmres(osys.A, osys.rhs, o.fieldsplit_multiplicative_apply(osys.A, osys.n))
    assert sol.iteration_number == ref.its


def test_G10_error_norms_on_device(gpu_ctx_factory, goldens):
    """l2_error / h1_seminorm_error through the public API against the reference's convergence.csv
    (2D Q1, MUMPS rows) and against the oracle's high-order quadrature."""
    import perphil_amd as pa
    from perphil_amd import fd, solver_parameters as spar
    from perphil_amd.postprocessing import l2_error, h1_seminorm_error, split_dpp_solution

    params = pa.DPPParameters()   # convergence_2d.py uses the defaults
    for N in (16, 32):
        g = next(r for r in goldens["G10_convergence_2d"] if r["solver"] == "Monolithic LU with MUMPS" and int(r["N"]) == N)
        mesh = pa.create_mesh(N, N, quadrilateral=True)
        _, V = pa.create_function_spaces(mesh)
        W = fd.MixedFunctionSpace((V, V))
        _, p1e, _, p2e = pa.exact_expressions(mesh, params)
        bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
        sol = pa.solve_dpp(W, params, bcs, solver_parameters=spar.LINEAR_SOLVER_PARAMS)
        p1h, p2h = split_dpp_solution(sol.solution)
        vals = (l2_error(p1h, p1e), l2_error(p2h, p2e), h1_seminorm_error(p1h, p1e), h1_seminorm_error(p2h, p2e))
        for v, key in zip(vals, ("e1_L2", "e2_L2", "e1_H1s", "e2_H1s")):
            assert v == pytest.approx(g[key], rel=2e-9), (N, key)    # measured <= 5e-12 (tests/test_reference_sweep.py: all 30 rows)
    # against the oracle's quadrature of the same fields (tight)
    om = o.build_mesh(2, o.CELL_QUAD, 32, 32)
    Pd = o.Params()
    import math
    eta, pi = Pd.eta, math.pi
    ex = lambda X: (Pd.mu / pi) * np.exp(pi * X[:, 0]) * np.sin(pi * X[:, 1]) - (Pd.mu / (Pd.beta * Pd.k1)) * np.exp(eta * X[:, 1])
    gr = lambda X: np.stack([Pd.mu * np.exp(pi * X[:, 0]) * np.sin(pi * X[:, 1]),
                             Pd.mu * np.exp(pi * X[:, 0]) * np.cos(pi * X[:, 1]) - (Pd.mu / (Pd.beta * Pd.k1)) * eta * np.exp(eta * X[:, 1])], 1)
    e1, h1 = o.error_norms(om, p1h.vector(), ex, gr, nq=6)
    assert vals[0] == pytest.approx(e1, rel=1e-10) and vals[2] == pytest.approx(h1, rel=1e-10)
    # 3D hex: second-order L2 convergence of the manufactured problem
    errs = []
    for N in (8, 16):
        mesh = fd.UnitCubeMesh(N, N, N, hexahedral=True)
        V = fd.FunctionSpace(mesh, "CG", 1)
        W = V * V
        p3 = pa.DPPParameters(k1=1.0, k2=0.01)
        _, p1e, _, p2e = pa.exact_expressions_3d(mesh, p3)
        bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
        sol = pa.solve_dpp(W, p3, bcs, solver_parameters=spar.FIELDSPLIT_MG_PARAMS)
        errs.append(l2_error(sol.solution.sub(0), p1e))
    assert 1.6 < np.log2(errs[0] / errs[1]) < 2.3


def test_conditioning_and_harness_api(gpu_ctx_factory, goldens):
    """get_matrix_data_from_form / calculate_condition_number / Approach harness (reference
    solvers/_tests/test_conditioning.py:16-56, experiments/_tests/test_iterative_bench.py:16-29) and the
    reference's conditioning goldens G3 / G4 through the public API."""
    import perphil_amd as pa
    from perphil_amd import fd, conditioning, iterative_bench as ib

    mesh = pa.create_mesh(10, 10, quadrilateral=True)
    _, V = pa.create_function_spaces(mesh)
    W = V * V
    params = pa.DPPParameters(k1=1.0, k2=1 / 1e2, beta=1.0, mu=1)
    _, p1e, _, p2e = pa.exact_expressions(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    a, _ = pa.dpp_form(W, params)
    md = conditioning.get_matrix_data_from_form(a, bcs)
    assert md.is_symmetric and md.number_of_dofs == W.dim() and md.number_of_nonzero_entries == md.sparse_csr_data.nnz
    g = goldens["G3_condition_numbers_10x10"]
    assert conditioning.calculate_condition_number(md.sparse_csr_data, None) == pytest.approx(g["monolithic"], rel=1e-10)
    (am, _), (ai, _) = pa.dpp_delayed_form(V, V, params, fd.Function(V), fd.Function(V))
    assert conditioning.calculate_condition_number(conditioning.get_matrix_data_from_form(am, [bcs[0]]).sparse_csr_data) == \
        pytest.approx(g["macro"], rel=1e-10)
    assert conditioning.calculate_condition_number(conditioning.get_matrix_data_from_form(ai, [bcs[1]]).sparse_csr_data) == \
        pytest.approx(g["micro"], rel=1e-10)
    # harness: homogeneous BCs, conditioning.csv row N=8
    _, _, W8 = ib.build_spaces(ib.build_mesh(8, 8))
    c = ib.estimate_condition_numbers(W8)
    g4 = goldens["G4_conditioning_2d"][1]
    assert c["monolithic"] == pytest.approx(g4["cond_monolithic"], rel=1e-10)
    assert c["macro"] == pytest.approx(g4["cond_macro"], rel=1e-10) and c["micro"] == pytest.approx(g4["cond_micro"], rel=1e-10)
    assert ib.params_for(ib.Approach.SS_GMRES)["pc_type"] == "fieldsplit"
    res = ib.solve_on_mesh(W8, ib.Approach.SS_GMRES)
    assert isinstance(res, ib.SolveResult) and res.iteration_number >= 0 and res.fields is not None


def test_G8_residual_histories_on_the_gpu(gpu_ctx_factory, goldens):
    """the reference's stored KSP residual histories of the 10 x 10 problem
    (notebooks/conforming-galerkin-fem-operator-splitting-2D-perphil.ipynb:412-553 plain GMRES(30), 141 steps; :815-822
    GMRES + multiplicative field-split with LU blocks, 6 steps) against the histories the HIP solver returns through the C
    ABI (pph_solve's `hist`), not only against the oracle: plain GMRES to 1e-9 inside the first restart cycle (PETSc
    and the GPU: classical Gram-Schmidt) and on the same curve to the end; field-split to 1e-6 on every entry."""
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, 2, o.CELL_QUAD, 10, 10, 0)
    ref = np.array(goldens["G8_gmres_history_10x10"])
    cfg = _cfg(ksp_type=f.KSP_GMRES, pc_type=f.PC_NONE, rtol=1e-12, atol=1e-50, max_it=1000)
    x, info, hist = ctx.solve(cfg, hist_cap=256)
    assert info.converged and abs(info.iterations - (len(ref) - 1)) <= 1
    assert hist[0] == pytest.approx(goldens["G1_initial_residual_10x10"], rel=1e-12)
    np.testing.assert_allclose(hist[:31], ref[:31], rtol=1e-9)
    m = min(len(hist), len(ref))
    # later restart cycles drift with the rounding of the orthogonalisation but stay on the curve: every one of the 141
    # entries within 0.15 of a decade (the worst two: 0.10), 13 decades covered
    np.testing.assert_allclose(np.log10(hist[:m]), np.log10(ref[:m]), atol=0.15)
    ref = np.array(goldens["G8_fieldsplit_lu_history_10x10"])
    cfg = _cfg(ksp_type=f.KSP_GMRES, pc_type=f.PC_FIELDSPLIT, rtol=1e-12, atol=1e-50, inner_ksp_type=f.KSP_CG,
               inner_pc_type=f.PC_MG, inner_rtol=1e-13)
    x, info, hist = ctx.solve(cfg, hist_cap=32)
    assert info.converged and info.iterations == len(ref) - 1
    np.testing.assert_allclose(hist[:6], ref[:6], rtol=1e-6)
    np.testing.assert_allclose(hist[6], ref[6], rtol=1e-3)      # 15 decades below the first entry: rounding of the block solves


@pytest.mark.parametrize("row", [0, 1, 2, 3])
def test_G4_conditioning_rows_on_the_gpu(gpu_ctx_factory, goldens, row):
    """conditioning.csv rows N = 4, 8, 16, 32 (notebooks/results-conforming-2d/conditioning/conditioning.csv:2-5)
    through estimate_condition_numbers on the device-assembled matrices (N = 64, the fifth row, is a dense SVD of 8 450
    dofs - minutes of host time: tools/r3_conditioning_3d.py's kind of run, not a test)."""
    from perphil_amd import iterative_bench as ib

    g4 = goldens["G4_conditioning_2d"][row]
    N = int(g4["N"]) if "N" in g4 else int(g4["nx"])
    _, _, W = ib.build_spaces(ib.build_mesh(N, N))
    c = ib.estimate_condition_numbers(W)
    assert c["monolithic"] == pytest.approx(g4["cond_monolithic"], rel=1e-9)
    assert c["macro"] == pytest.approx(g4["cond_macro"], rel=1e-9) and c["micro"] == pytest.approx(g4["cond_micro"], rel=1e-9)


def test_perf_harness_row_schema(gpu_ctx_factory, goldens):
    """run_perf_once_3d mirrors the reference's flat row: same column names, dofs / cells / iteration
    counts of the committed CSV for the rows that are algorithm-independent (reference
    experiments/_tests/test_petsc_profiling.py:16-58)."""
    import warnings
    from perphil_amd.iterative_bench import Approach
    from perphil_amd.profiling_3d import run_perf_once_3d, run_perf_sweep_3d

    cols = [c for c in goldens["perf_csv_columns_3d"]]
    row = run_perf_once_3d(4, Approach.PLAIN_GMRES, repeats=2)
    assert list(row.keys()) == cols
    g = next(r for r in goldens["G6_G9_perf_3d_tets"] if r["approach"] == "GMRES" and r["nx"] == 4)
    assert row["dofs"] == g["dofs"] and row["num_cells"] == g["num_cells"] and abs(row["iterations"] - g["iterations"]) <= 1
    assert row["time_MatMult"] > 0 and row["time_KSPSolve"] >= row["time_MatMult"] and row["mflops_MatMult"] > 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        df = run_perf_sweep_3d([4, 8], [Approach.SS_GMRES, Approach.MONOLITHIC_MUMPS], repeats=1)
    assert list(df.columns) == cols and len(df) == 4
    assert list(df[df.approach == "Scale-Splitting GMRES"].iterations) == [4, 4]     # G9
    assert list(df[df.approach == "Monolithic LU with MUMPS"].iterations) == [1, 1]


@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(3, o.CELL_HEX, 12, 8, 4), (3, o.CELL_HEX, 9, 5, 7), (3, o.CELL_TET, 6, 10, 4),
                                               (2, o.CELL_QUAD, 24, 8, 0), (2, o.CELL_TRI, 7, 9, 0), (3, o.CELL_HEX, 1, 1, 1),
                                               (2, o.CELL_QUAD, 1, 3, 0)])
def test_ragged_meshes_all_solvers(gpu_ctx_factory, dim, kind, nx, ny, nz):
    """Non-cubic, odd and minimal meshes: partial multigrid hierarchies (coarsening stops at the first odd
    direction; single level = polynomial preconditioner), every solver family against the direct solution."""
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, dim, kind, nx, ny, nz)
    ud = o.solve_direct(osys)
    scale = max(np.abs(ud).max(), 1e-300)
    for cfg in (_cfg(picard=1, inner_pc_type=f.PC_MG, inner_rtol=1e-12, picard_rtol=1e-10),
                _cfg(picard=1, inner_pc_type=f.PC_MG, inner_rtol=1e-12, picard_rtol=1e-10, inner_reduction=1e-2, mg_smooth=1),
                _cfg(pc_type=f.PC_FIELDSPLIT, inner_pc_type=f.PC_MG, inner_rtol=1e-12, rtol=1e-10),
                _cfg(ksp_type=f.KSP_CG, pc_type=f.PC_BLOCK2, rtol=1e-10),
                _cfg(pc_type=f.PC_JACOBI, rtol=1e-10)):
        x, info, _ = ctx.solve(cfg)
        assert info.converged
        assert np.abs(x - ud).max() / scale < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(2, o.CELL_QUAD, 7, 5, 0), (2, o.CELL_TRI, 6, 9, 0), (3, o.CELL_HEX, 5, 4, 6),
                                               (3, o.CELL_TET, 4, 6, 3), (3, o.CELL_HEX, 1, 1, 1)])
def test_darcy_velocity_projection_matches_oracle(gpu_ctx_factory, dim, kind, nx, ny, nz):
    """calculate_darcy_velocity_from_pressure (reference postprocessing.py:34-63): device projection vs the
    oracle's sparse-direct projection on a random nodal field, and exactness for a linear pressure."""
    ctx = gpu_ctx_factory()
    ctx.mesh_build(dim, kind, nx, ny, nz)
    om = o.build_mesh(dim, kind, nx, ny, nz)
    rng = np.random.default_rng(11)
    p = rng.standard_normal(om.num_nodes)
    u = ctx.darcy_velocity(p, 0.37)
    ref = o.darcy_velocity(om, p, 0.37)
    assert u.shape == ref.shape == (om.num_nodes, dim)
    np.testing.assert_allclose(u, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    a = np.array([0.5, -1.25, 2.0])[:dim]
    u = ctx.darcy_velocity(om.coords[:, :dim] @ a + 3.0, 2.0)
    np.testing.assert_allclose(u, np.tile(-2.0 * a, (om.num_nodes, 1)), rtol=0, atol=1e-10)


@pytest.mark.gpu
def test_darcy_velocity_public_api(gpu_ctx_factory):
    import perphil_amd as pa
    from perphil_amd import fd
    from perphil_amd.postprocessing import calculate_darcy_velocity_from_pressure

    mesh = pa.create_mesh(6, 4, quadrilateral=True)
    U, V = pa.create_function_spaces(mesh)
    ph = fd.Function(V).interpolate(lambda X: 2.0 * X[:, 0] - X[:, 1])
    vel = calculate_darcy_velocity_from_pressure(ph, fd.Constant(0.5), velocity_space=U)
    assert vel.function_space() is U
    np.testing.assert_allclose(vel.vector().reshape(-1, 2), np.tile([-1.0, 0.5], (V.dim(), 1)), atol=1e-11)
    vel2 = calculate_darcy_velocity_from_pressure(ph, 0.5)
    np.testing.assert_array_equal(vel2.vector(), vel.vector())


@pytest.mark.gpu
@pytest.mark.parametrize("hexa,k2", [(True, 1e-2), (False, 1e-4)])
def test_full_size_properties_256cubed(gpu_ctx_factory, hexa, k2):
    """BASELINE configs 4 (256^3 Q1, k1/k2 = 1e2) and 5 (256^3 Kuhn tets, k1/k2 = 1e4) at full size, through
    size-independent properties: closed-form sizes, kernel of K / volume from M, symmetry of the blocks,
    A12 == A21, identity Dirichlet rows, an independently recomputed true residual of the converged Picard
    solve, and O(h^2) agreement with the manufactured solution."""
    f = _ffi()
    N = 256
    Pp = o.Params(k1=1.0, k2=k2)
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, f.CELL_HEX if hexa else f.CELL_TET, N, N, N)
    n = ctx.n
    assert n == (N + 1) ** 3
    nnz_tet = n + 2 * (3 * N * (N + 1) ** 2 + 3 * N * N * (N + 1) + N ** 3)     # SURVEY.md section 8 size table
    assert ctx.nnzb == ((3 * N + 1) ** 3 if hexa else nnz_tet)
    assert ctx.ncell == (N ** 3 if hexa else 6 * N ** 3)
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=hexa)
    b = mesh.boundary_nodes()
    assert b.size == 6 * N * N + 2
    e1, e2 = o.exact_pressures(mesh.node_coordinates(b), Pp)
    ctx.set_dirichlet(0, b, e1)
    ctx.set_dirichlet(1, b, e2)
    ctx.assemble(Pp.k1, Pp.k2, Pp.beta, Pp.mu, monolithic=False)
    ones = np.ones(n)
    assert np.abs(ctx.spmv(f.MAT_K, ones)).max() < 1e-11
    assert ctx.spmv(f.MAT_M, ones).sum() == pytest.approx(1.0, rel=1e-11)
    rng = np.random.default_rng(256)
    x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    for which in (f.MAT_A11, f.MAT_A22):
        Ax, Ay = ctx.spmv(which, x), ctx.spmv(which, y)
        assert abs(y @ Ax - x @ Ay) <= 1e-10 * abs(y @ Ax)
        assert x @ Ax > 0
        np.testing.assert_array_equal(Ax[b], x[b])
    A12x = ctx.spmv(f.MAT_A12, x)
    np.testing.assert_array_equal(A12x, ctx.spmv(f.MAT_A21, x))
    assert not A12x[b].any()
    # the bench's algorithm: inexact block-Picard with multigrid-CG block solves
    xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                 inner_reduction=1e-1, inner_norm=1, mg_smooth=1, picard_rtol=1e-8, picard_max_it=200))
    assert info.converged
    r, u0 = ctx.rhs()
    d = xs - u0
    res1 = r[:n] - ctx.spmv(f.MAT_A11, d[:n]) - ctx.spmv(f.MAT_A12, d[n:])
    res2 = r[n:] - ctx.spmv(f.MAT_A21, d[:n]) - ctx.spmv(f.MAT_A22, d[n:])
    true_res = np.sqrt(res1 @ res1 + res2 @ res2)
    assert true_res <= 1.05e-8 * np.linalg.norm(r)
    assert true_res == pytest.approx(info.resnorm, rel=1e-3)
    np.testing.assert_array_equal(xs[b], e1)
    np.testing.assert_array_equal(xs[n + b], e2)
    ex1, ex2 = o.exact_pressures(mesh.node_coordinates(), Pp)
    tol = 5e-4 if hexa else 2e-2      # config 5: boundary layer exp(100 y) resolved with eta*h = 0.39
    assert np.abs(xs[:n] - ex1).max() / np.abs(ex1).max() < tol
    assert np.abs(xs[n:] - ex2).max() / np.abs(ex2).max() < tol


@pytest.mark.gpu
def test_config5_256cubed_tets_gmres_fieldsplit(gpu_ctx_factory, goldens):
    """BASELINE config 5 as stated, at full size on one GPU: 256^3 Kuhn P1 tetrahedra (100.7 M cells), k1/k2 = 1e4
    (k2 = 1e-4, eta = 100), GMRES + multiplicative field-split (reference src/perphil/solvers/parameters.py:30-37,
    block LU -> multigrid-CG block solves).  Checks: 4 outer iterations like the reference at every size (golden G9,
    notebooks/results-conforming-3d/petsc_profiling/petsc_perf_breakdown_3d.csv:39), the TRUE residual recomputed
    through pph_spmv on the blocks <= 1e-8 ||F(u0)||, and agreement with the block-Picard solution to 1e-6."""
    f = _ffi()
    N = 256
    Pp = o.Params(k1=1.0, k2=1e-4)
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, f.CELL_TET, N, N, N)
    n = ctx.n
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=False)
    b = mesh.boundary_nodes()
    e1, e2 = o.exact_pressures(mesh.node_coordinates(b), Pp)
    ctx.set_dirichlet(0, b, e1)
    ctx.set_dirichlet(1, b, e2)
    ctx.assemble(Pp.k1, Pp.k2, Pp.beta, Pp.mu, monolithic=True)
    g9 = {r["iterations"] for r in goldens["G6_G9_perf_3d_tets"] if r["approach"] == "Scale-Splitting GMRES"}
    assert g9 == {4}            # the reference: 4 outer iterations at every mesh size
    cfg = _cfg(ksp_type=f.KSP_GMRES, pc_type=f.PC_FIELDSPLIT, rtol=1e-8, atol=1e-12, inner_ksp_type=f.KSP_CG,
               inner_pc_type=f.PC_MG, inner_rtol=1e-10, mg_smooth=2)
    xs, info, hist = ctx.solve(cfg, hist_cap=16)
    assert info.converged and info.iterations == 4
    # PETSc's criterion (left preconditioning): the PRECONDITIONED residual dropped by ksp_rtol
    assert info.resnorm <= 1e-8 * hist[0]
    r, u0 = ctx.rhs()

    def true_residual(x):
        d = x - u0
        res1 = r[:n] - ctx.spmv(f.MAT_A11, d[:n]) - ctx.spmv(f.MAT_A12, d[n:])
        res2 = r[n:] - ctx.spmv(f.MAT_A21, d[:n]) - ctx.spmv(f.MAT_A22, d[n:])
        return np.sqrt(res1 @ res1 + res2 @ res2)

    # the unpreconditioned residual of that iterate sits a decade above (the block scaling at k1/k2 = 1e4 separates
    # the two norms); one more decade of ksp_rtol brings the TRUE residual under 1e-8 ||F(u0)||
    assert true_residual(xs) <= 1e-6 * np.linalg.norm(r)
    cfg.rtol = 1e-10
    xs, info, _ = ctx.solve(cfg)
    assert info.converged and info.iterations <= 6
    assert true_residual(xs) <= 1e-8 * np.linalg.norm(r)
    xp, pinfo, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                  inner_reduction=1e-1, inner_norm=1, mg_smooth=1, picard_rtol=1e-9, picard_max_it=200))
    assert pinfo.converged
    assert np.abs(xs - xp).max() <= 1e-6 * np.abs(xp).max()


@pytest.mark.gpu
@pytest.mark.parametrize("hexa,N,norm,red", [(True, 96, 0, 1e-2), (False, 48, 0, 1e-2), (True, 96, 1, 1e-1), (False, 48, 1, 1e-1),
                                             (True, 40, 1, 3e-2),
                                             # BASELINE config 3's size with the bench's algorithm (tile assembly on every
                                             # level down to 33^3, 2.1 M rows per block: 57 M entries compared per block)
                                             (True, 128, 1, 1e-1),
                                             # no inner convergence test (ksp_norm_type none): exactly `red` CG iterations
                                             # per block solve (launch-only sweeps)
                                             (True, 96, 2, 1), (False, 48, 2, 1), (True, 40, 2, 2), (True, 64, 2, 3)])
def test_hip_matches_cpu_port_mid_size(gpu_ctx_factory, hexa, N, norm, red):
    """HIP path vs the C/OpenMP restatement (oracle/dpp_cpu.c) at sizes the NumPy oracle cannot reach in seconds:
    K/M/blocks entry for entry, right-hand side, and the bench's inexact-Picard solve (same sweeps, same number
    of CG iterations, same solution); norm 2 = the launch-only variant with a fixed iteration count per block solve."""
    from oracle import dpp_cpu as cpu

    cpu.set_threads(8)     # fixed reduction order of the C port, whatever the host offers
    f = _ffi()
    kind = f.CELL_HEX if hexa else f.CELL_TET
    S = cpu.CpuSystem(3, kind, N, N, N)
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, kind, N, N, N)
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=hexa)
    b = mesh.boundary_nodes()
    e1, e2 = o.exact_pressures(mesh.node_coordinates(b), P)
    for tgt in (S, ctx):
        tgt.set_dirichlet(0, b, e1)
        tgt.set_dirichlet(1, b, e2)
    S.assemble(P.k1, P.k2, P.beta, P.mu)
    ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
    for wc, wg in ((cpu.MAT_K, f.MAT_K), (cpu.MAT_M, f.MAT_M), (cpu.MAT_A11, f.MAT_A11), (cpu.MAT_A22, f.MAT_A22),
                   (cpu.MAT_A12, f.MAT_A12)):
        ref, got = S.csr(wc), ctx.csr(wg)
        np.testing.assert_array_equal(got.indptr, ref.indptr)
        np.testing.assert_array_equal(got.indices, ref.indices)
        np.testing.assert_allclose(got.data, ref.data, rtol=0, atol=1e-12 * np.abs(ref.data).max())
    r_ref, _ = S.rhs()
    r, _ = ctx.rhs()
    np.testing.assert_allclose(r, r_ref, rtol=0, atol=1e-12 * np.abs(r_ref).max())
    S.mg_setup()
    if norm == 2:
        kits = int(red)
        x_ref, sweeps, inner, res = S.picard(reduction=0.0, inner_norm=2, inner_max_it=kits)
        xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                     inner_max_it=kits, inner_norm=2, mg_smooth=1, picard_rtol=1e-8, picard_max_it=100))
    else:
        x_ref, sweeps, inner, res = S.picard(reduction=red, inner_norm=norm)
        xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                     inner_reduction=red, inner_norm=norm, mg_smooth=1, picard_rtol=1e-8, picard_max_it=100))
    assert info.converged and info.iterations == sweeps and info.inner_iterations == inner
    assert np.abs(xs - x_ref).max() <= 1e-9 * np.abs(x_ref).max()
    assert info.resnorm == pytest.approx(res, rel=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(2, o.CELL_QUAD, 9, 6, 0), (2, o.CELL_QUAD, 70, 33, 0), (3, o.CELL_HEX, 6, 5, 4),
                                               (3, o.CELL_HEX, 1, 1, 1), (3, o.CELL_HEX, 35, 18, 9), (3, o.CELL_HEX, 64, 64, 64)])
def test_node_assembly_kernel_equals_tile_kernel(gpu_ctx_factory, dim, kind, nx, ny, nz):
    """k_asm_node2 (one thread per node, registers only; default on box meshes) against k_asm_tile (LDS element rows).  The
    node kernel integrates a uniform box on its CANONICAL edges h e_d (every edge of the mesh was checked against them to one
    rounding of a coordinate at mesh build, MeshData::uniform), the tile kernel on the stored coordinates i / nx, whose
    spacing wobbles by that rounding: relative to an edge of length 1 / n that is n x 2^-52.  Operators and right-hand side
    therefore agree to (n + 4) x 2^-52 of the largest entry (1e-15 when the cell counts are powers of two: then i / n is
    exact), u0 exactly, the multigrid-preconditioned Picard solve with the same sweeps and CG iterations and the same
    solution to 1e-12; both against the oracle's matrix on the small meshes; asm_uniform 0 (stored coordinates in the node
    kernel too) restores 1e-15 on every mesh."""
    f = _ffi()
    om = o.build_mesh(dim, kind, nx, ny, nz) if nx * max(ny, 1) * max(nz, 1) <= 6000 else None
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(nx, ny, nz, hexahedral=True) if dim == 3 else fdm.UnitSquareMesh(nx, ny, quadrilateral=True)
    b = mesh.boundary_nodes()
    g1, g2 = o.exact_pressures(mesh.node_coordinates(b), P)
    out = []
    for node, uni in ((1, 1), (0, 1), (1, 0)):
        ctx = gpu_ctx_factory()
        ctx.set_option("asm_node", node)
        ctx.set_option("asm_uniform", uni)
        ctx.set_option("asm_tile", 2)
        ctx.mesh_build(dim, kind, nx, ny, nz)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b[: len(b) // 2], g2[: len(b) // 2])     # different Dirichlet sets: A21 stored on its own
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
        mats = [ctx.csr(w) for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12, f.MAT_A21)]
        rhs, u0 = ctx.rhs()
        xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                     inner_reduction=1e-1, inner_norm=1, mg_smooth=1, picard_rtol=1e-8))
        out.append((mats, rhs, u0, xs, (info.iterations, info.inner_iterations)))
        ctx.close()
    exact = all(v & (v - 1) == 0 for v in (nx, ny, nz) if v > 0)
    tol = 1e-15 if exact else (max(nx, ny, nz) + 4) * 2.0 ** -52
    for A, B in zip(out[0][0], out[1][0]):
        np.testing.assert_array_equal(A.indptr, B.indptr)
        np.testing.assert_array_equal(A.indices, B.indices)
        np.testing.assert_allclose(A.data, B.data, rtol=0, atol=tol * np.abs(B.data).max())
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=tol * np.abs(out[1][1]).max())
    for A, B in zip(out[2][0], out[1][0]):          # the node kernel on the stored coordinates: the tile kernel's entries
        np.testing.assert_allclose(A.data, B.data, rtol=0, atol=1e-15 * np.abs(B.data).max())
    np.testing.assert_allclose(out[2][1], out[1][1], rtol=0, atol=1e-15 * np.abs(out[1][1]).max())
    np.testing.assert_array_equal(out[0][2], out[1][2])
    np.testing.assert_allclose(out[0][3], out[1][3], rtol=0, atol=1e-12 * np.abs(out[1][3]).max())
    assert out[0][4] == out[1][4]
    if om is not None:
        # the oracle with the same (differing) Dirichlet sets
        m1, m2 = np.zeros(om.num_nodes, bool), np.zeros(om.num_nodes, bool)
        m1[b] = True
        m2[b[: len(b) // 2]] = True
        e1, e2 = o.exact_pressures(om.coords, P)
        osys = o.build_system(om, P, g1=e1, g2=e2, mms=False, mask1=m1, mask2=m2)
        n = osys.n
        A = osys.A.tocsr()
        for blk, (r0, c0) in zip(out[0][0], ((0, 0), (n, n), (0, n), (n, 0))):
            assert abs(blk - A[r0:r0 + n, c0:c0 + n]).max() <= 1e-12 * abs(A).max()
        np.testing.assert_allclose(out[0][1], osys.rhs, rtol=0, atol=1e-12 * np.abs(osys.rhs).max())


@pytest.mark.gpu
def test_launch_only_sweeps_report_a_breakdown_instead_of_nans(gpu_ctx_factory):
    """inner_norm 2 (exactly k CG iterations per block solve, every scalar on the device): a block whose residual is
    exactly zero (no coupling, homogeneous data on field 0) gives p.Ap = 0 in every one of its solves.  The update kernel
    must leave x and r alone (no 0 / 0), count the event on the device and the solve must report it as inner_failed -
    like the host-scalar loop, which flags the same case as a breakdown - while field 1 still converges."""
    f = _ffi()
    N = 8
    om = o.build_mesh(3, o.CELL_HEX, N, N, N)
    b = o.boundary_nodes(om)
    _, e2 = o.exact_pressures(om.coords, P)
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, f.CELL_HEX, N, N, N)
    ctx.set_dirichlet(0, b, np.zeros(len(b)))
    ctx.set_dirichlet(1, b, e2[b])
    ctx.assemble(P.k1, P.k2, 0.0, P.mu, monolithic=False)      # beta = 0: the blocks do not couple
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_max_it=3, inner_norm=2,
               mg_smooth=1, picard_rtol=1e-8, picard_max_it=100)
    xs, info, _ = ctx.solve(cfg)
    n = ctx.n
    assert np.all(np.isfinite(xs)) and info.converged
    assert np.all(xs[:n] == 0.0)
    assert info.inner_failed == 1
    A22 = ctx.csr(f.MAT_A22)
    rhs, u0 = ctx.rhs()
    r2 = rhs[n:] - A22 @ (xs[n:] - u0[n:])
    assert np.linalg.norm(r2) <= 1e-8 * np.linalg.norm(rhs)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(2, o.CELL_QUAD, 12, 9, 0), (2, o.CELL_TRI, 7, 10, 0), (3, o.CELL_HEX, 6, 5, 4),
                                               (3, o.CELL_TET, 4, 5, 3)])
def test_error_norms_for_an_arbitrary_exact_expression(gpu_ctx_factory, dim, kind, nx, ny, nz):
    """l2_error / h1_seminorm_error take any exact expression like the reference's (utils/postprocessing.py:89-124 accept a
    UFL expression): a Python callable of point arrays with a gradient attribute, the same callable without one (central
    differences), a Constant, and another CG-1 Function - checked against the oracle's quadrature (o.error_norms) with a
    NON-manufactured field, on all four cell kinds; chunked evaluation gives the same sums."""
    import perphil_amd as pa
    from perphil_amd import fd
    from perphil_amd.postprocessing import h1_seminorm_error, l2_error

    mesh = (pa.create_mesh(nx, ny, quadrilateral=(kind == o.CELL_QUAD)) if dim == 2
            else fd.UnitCubeMesh(nx, ny, nz, hexahedral=(kind == o.CELL_HEX)))
    V = fd.FunctionSpace(mesh, "CG", 1)
    om = o.build_mesh(dim, kind, nx, ny, nz)

    def ex(X):
        r = np.cos(2.0 * X[:, 0]) * np.exp(-X[:, 1]) + 0.3 * X[:, 0] * X[:, 1]
        return r + (np.sin(1.5 * X[:, 2]) if dim == 3 else 0.0)

    def gr(X):
        g = [-2.0 * np.sin(2.0 * X[:, 0]) * np.exp(-X[:, 1]) + 0.3 * X[:, 1],
             -np.cos(2.0 * X[:, 0]) * np.exp(-X[:, 1]) + 0.3 * X[:, 0]]
        if dim == 3:
            g.append(1.5 * np.cos(1.5 * X[:, 2]))
        return np.stack(g, axis=1)

    rng = np.random.default_rng(3)
    uh = ex(om.coords) + 1e-2 * rng.uniform(-1, 1, om.num_nodes)       # not the interpolant: a real error
    fh = fd.Function(V, uh, name="uh")
    rl2, rh1 = o.error_norms(om, uh, ex, gr, nq=5)

    class WithGrad:
        def __call__(self, X): return ex(X)
        def grad(self, X): return gr(X)

    assert l2_error(fh, WithGrad(), 5) == pytest.approx(rl2, rel=1e-10)
    assert h1_seminorm_error(fh, WithGrad(), 5) == pytest.approx(rh1, rel=1e-10)
    assert l2_error(fh, ex, 5) == pytest.approx(rl2, rel=1e-10)                 # gradient not needed
    assert h1_seminorm_error(fh, ex, 5) == pytest.approx(rh1, rel=1e-6)         # central differences
    ctx = mesh.context()
    a = ctx.error_norms_sampled(uh, ex, gr, 5, chunk_cells=7)
    assert a[0] == pytest.approx(rl2, rel=1e-10) and a[1] == pytest.approx(rh1, rel=1e-10)
    c2, ch = o.error_norms(om, uh, lambda X: np.full(X.shape[0], 0.25), lambda X: np.zeros_like(X), nq=4)
    assert l2_error(fh, fd.Constant(0.25), 4) == pytest.approx(c2, rel=1e-10)
    assert h1_seminorm_error(fh, 0.25, 4) == pytest.approx(ch, rel=1e-10)
    vh = ex(om.coords)
    d2, dh = o.error_norms(om, uh - vh, lambda X: np.zeros(X.shape[0]), lambda X: np.zeros_like(X), nq=3)
    gh = fd.Function(V, vh, name="vh")
    assert l2_error(fh, gh) == pytest.approx(d2, rel=1e-10) and h1_seminorm_error(fh, gh) == pytest.approx(dh, rel=1e-10)


@pytest.mark.gpu
def test_spmv_long_rows_and_residual_forms(gpu_ctx_factory):
    """Monolithic 3D rows (up to 54 entries) take more than one 32-entry step of the aligned-wide kernels; the
    CG and residual entry points (fused p.Ap, b - Ax) are covered through a Jacobi-CG solve against the oracle."""
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, 3, o.CELL_HEX, 5, 4, 3)
    ctx.set_option("op_format", 0)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, 2 * osys.n)
    ref = osys.A @ x
    y = ctx.spmv(f.MAT_MONO, x)
    assert np.abs(y - ref).max() <= 1e-13 * np.abs(ref).max()
    xs, info, _ = ctx.solve(_cfg(ksp_type=f.KSP_CG, pc_type=f.PC_JACOBI, rtol=1e-10))
    res = o.pcg(osys.A, osys.rhs, o.jacobi_apply(osys.A), rtol=1e-10)
    assert info.converged and abs(info.iterations - res.its) <= 1
    assert np.abs(xs - (osys.u0 + res.x)).max() <= 1e-8 * np.abs(xs).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(2, o.CELL_QUAD, 1, 1, 0), (2, o.CELL_QUAD, 2, 5, 0), (2, o.CELL_TRI, 1, 3, 0),
                                               (2, o.CELL_TRI, 7, 4, 0), (3, o.CELL_HEX, 1, 1, 1), (3, o.CELL_HEX, 2, 1, 3),
                                               (3, o.CELL_HEX, 9, 7, 5), (3, o.CELL_TET, 1, 2, 1), (3, o.CELL_TET, 5, 3, 4),
                                               (2, o.CELL_QUAD, 40, 33, 0), (3, o.CELL_HEX, 21, 18, 10)])
def test_stencil_ell_format_equals_csr(gpu_ctx_factory, dim, kind, nx, ny, nz):
    """The stencil-ELL operator format (default) against the CSR format on the same mesh, tiny boxes (stencil slots
    clipped on every side, px < 3) and ragged ones included: the exported CSR blocks are identical entry for entry
    (the fused assembly writes the stencil-ELL arrays, pph_get_csr converts), the products of all blocks agree with
    SciPy's, and both formats solve to the same iterates."""
    f = _ffi()
    om = o.build_mesh(dim, kind, nx, ny, nz)
    b = o.boundary_nodes(om)
    g1, g2 = o.exact_pressures(om.coords[b], P)
    res = {}
    for fmt in (0, 1):
        ctx = gpu_ctx_factory()
        ctx.set_option("op_format", fmt)
        ctx.set_option("asm_tile", 0)      # same element-row arithmetic in both runs
        ctx.mesh_build(dim, kind, nx, ny, nz)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
        rng = np.random.default_rng(7)
        x = rng.uniform(-1, 1, om.num_nodes)
        prods = {w: ctx.spmv(w, x) for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12, f.MAT_A21, f.MAT_K, f.MAT_M)}
        mats = {w: ctx.csr(w) for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12, f.MAT_A21, f.MAT_K, f.MAT_M)}
        for w in mats:
            ref = mats[w] @ x
            assert np.abs(prods[w] - ref).max() <= 1e-13 * max(np.abs(ref).max(), 1e-300), (fmt, w)
        xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                     inner_reduction=1e-1, inner_norm=1, mg_smooth=1, picard_rtol=1e-9))
        res[fmt] = (mats, xs, info.iterations, info.inner_iterations)
    import scipy.sparse as sps

    for w in res[0][0]:
        np.testing.assert_array_equal(res[1][0][w].indptr, res[0][0][w].indptr)
        np.testing.assert_array_equal(res[1][0][w].indices, res[0][0][w].indices)
        # diagonal and upper entries: the stored values, bit for bit; lower entries are read from their mirror
        # (symmetric storage): equal to the row-wise computed ones up to the rounding of (g_a w) g_b vs (g_b w) g_a
        np.testing.assert_array_equal(sps.triu(res[1][0][w]).toarray(), sps.triu(res[0][0][w]).toarray())
        scale = max(np.abs(res[0][0][w].data).max(), 1e-300)
        np.testing.assert_allclose(res[1][0][w].data, res[0][0][w].data, rtol=0, atol=1e-14 * scale)
        if w in (f.MAT_A11, f.MAT_A22, f.MAT_A12, f.MAT_A21):      # the blocks (K, M are integrated straight into CSR)
            assert abs(res[1][0][w] - res[1][0][w].T).max() == 0.0  # exactly symmetric operators
    assert res[0][2:] == res[1][2:]
    np.testing.assert_allclose(res[1][1], res[0][1], rtol=0, atol=1e-10 * np.abs(res[0][1]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [0, 1, 2])
@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(2, o.CELL_QUAD, 9, 6, 0), (3, o.CELL_HEX, 6, 5, 4), (3, o.CELL_HEX, 16, 16, 16),
                                               (2, o.CELL_QUAD, 37, 19, 0), (3, o.CELL_HEX, 19, 9, 5),
                                               (2, o.CELL_TRI, 8, 6, 0), (3, o.CELL_TET, 6, 4, 8)])
def test_fused_assembly_equals_two_step(gpu_ctx_factory, dim, kind, nx, ny, nz, tile):
    """The fused node-centred pass (blocks, lifted right-hand side, smoother diagonal / bound straight from the
    element rows) against the two-step path (K, M, then k_lift_rhs / k_blocks / k_diag_lam): same entries, and the
    multigrid-preconditioned solve takes the same iterations; with asm_keep_km = 0 K and M are integrated on demand.
    tile = 0: the two-pass fused kernels (same arithmetic as the two-step path: bitwise equal); tile = 1: the
    single-pass tile kernel, which forms the element rows from geometry factors (another association order: equal to
    1e-13 of the largest entry; simplices have no tile kernel and stay bitwise); tile = 2: the tile kernel with
    asm_affine = 0, i.e. its general pass (Jacobian per Gauss point) instead of the once-per-cell factor of cells with
    equal parallel edges - which every cell of these box meshes is."""
    if tile == 2 and kind in (o.CELL_TRI, o.CELL_TET):
        pytest.skip("simplices have no tile kernel")
    f = _ffi()
    om = o.build_mesh(dim, kind, nx, ny, nz)
    b = o.boundary_nodes(om)
    g1, g2 = o.exact_pressures(om.coords[b], P)
    out = {}
    exact = (tile == 0) or kind in (o.CELL_TRI, o.CELL_TET)
    for mode, (fused, keep) in {"two-step": (0, 1), "fused": (1, 1), "fused-nokeep": (1, 0)}.items():
        ctx = gpu_ctx_factory()
        ctx.set_option("asm_tile", 2 * min(tile, 1))   # 2: the tile kernel on every level, whatever its size
        ctx.set_option("asm_affine", 0 if tile == 2 else 1)
        ctx.set_option("asm_fused", fused)
        ctx.set_option("asm_keep_km", keep)
        ctx.mesh_build(dim, kind, nx, ny, nz)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=True)
        mats = {w: ctx.csr(w) for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12, f.MAT_A21, f.MAT_MONO)}
        rhs, u0 = ctx.rhs()
        xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                     inner_reduction=1e-1, inner_norm=1, mg_smooth=1, picard_rtol=1e-8))
        K, M = ctx.csr(f.MAT_K), ctx.csr(f.MAT_M)       # after the solve: on-demand integration when not kept
        out[mode] = (mats, rhs, u0, xs, (info.iterations, info.inner_iterations), K, M)
    ref = out["two-step"]
    for mode in ("fused", "fused-nokeep"):
        got = out[mode]
        for w in ref[0]:
            np.testing.assert_array_equal(got[0][w].indices, ref[0][w].indices)
            if exact:
                # same arithmetic: the stored (diagonal + upper) entries bit for bit; the fused path's lower entries are
                # read from their mirror (symmetric stencil-ELL storage): rounding of (g_a w) g_b vs (g_b w) g_a apart
                if w != f.MAT_MONO:      # (the monolithic matrix holds whole coupling blocks above its diagonal)
                    np.testing.assert_array_equal(sp.triu(got[0][w]).toarray() if got[0][w].shape[0] < 3000 else sp.triu(got[0][w]).data,
                                                  sp.triu(ref[0][w]).toarray() if ref[0][w].shape[0] < 3000 else sp.triu(ref[0][w]).data)
                np.testing.assert_allclose(got[0][w].data, ref[0][w].data, rtol=0, atol=1e-14 * np.abs(ref[0][w].data).max())
            else:
                np.testing.assert_allclose(got[0][w].data, ref[0][w].data, rtol=0, atol=1e-13 * np.abs(ref[0][w].data).max())
        # the lifting sums are reduced in a different lane order: last-bit differences only
        np.testing.assert_allclose(got[1], ref[1], rtol=0, atol=(1e-14 if exact else 1e-12) * np.abs(ref[1]).max())
        np.testing.assert_array_equal(got[2], ref[2])
        assert got[4] == ref[4]
        np.testing.assert_allclose(got[3], ref[3], rtol=0, atol=1e-12 * np.abs(ref[3]).max())
        for q in (5, 6):
            if exact or mode == "fused-nokeep":    # not kept: integrated on demand by the two-pass kernels
                np.testing.assert_array_equal(got[q].data, ref[q].data)
            else:
                np.testing.assert_allclose(got[q].data, ref[q].data, rtol=0, atol=1e-13 * np.abs(ref[q].data).max())
    osys = o.build_system(om, P)
    assert abs(ref[0][f.MAT_MONO] - osys.A).max() <= 1e-12 * abs(osys.A).max()


@pytest.mark.gpu
def test_config3_128cubed_picard_with_jacobi_block_solves(gpu_ctx_factory):
    """BASELINE config 3 at full size: 128^3 Q1, Picard-split with Jacobi-preconditioned CG block solves (the fused
    CG-update kernel path), against the multigrid-preconditioned Picard solve of the same system and an
    independently recomputed residual."""
    f = _ffi()
    N = 128
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, f.CELL_HEX, N, N, N)
    n = ctx.n
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=True)
    b = mesh.boundary_nodes()
    e1, e2 = o.exact_pressures(mesh.node_coordinates(b), P)
    ctx.set_dirichlet(0, b, e1)
    ctx.set_dirichlet(1, b, e2)
    ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
    xj, ij, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_JACOBI, inner_rtol=1e-10,
                               inner_reduction=1e-2, picard_rtol=1e-8, picard_max_it=100))
    assert ij.converged and ij.iterations <= 12
    r, u0 = ctx.rhs()
    d = xj - u0
    res1 = r[:n] - ctx.spmv(f.MAT_A11, d[:n]) - ctx.spmv(f.MAT_A12, d[n:])
    res2 = r[n:] - ctx.spmv(f.MAT_A21, d[:n]) - ctx.spmv(f.MAT_A22, d[n:])
    assert np.sqrt(res1 @ res1 + res2 @ res2) <= 1.05e-8 * np.linalg.norm(r)
    xm, im, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                               inner_reduction=1e-1, inner_norm=1, mg_smooth=1, picard_rtol=1e-8, picard_max_it=100))
    assert im.converged
    assert np.abs(xj - xm).max() <= 2e-6 * np.abs(xm).max()
    assert ij.inner_iterations > 20 * im.inner_iterations      # what the multigrid block solves buy at this size


@pytest.mark.gpu
@pytest.mark.parametrize("pc", ["block2", "jacobi"])
def test_config2_64cubed_monolithic_cg(gpu_ctx_factory, pc):
    """BASELINE config 2 at full size: 64^3 Q1, monolithic CSR, CG with the 2x2 node-block Jacobi (or point Jacobi)
    preconditioner; the solution is checked through an independently recomputed residual and against the
    field-split GMRES solve of the same system."""
    f = _ffi()
    N = 64
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, f.CELL_HEX, N, N, N)
    n = ctx.n
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=True)
    b = mesh.boundary_nodes()
    e1, e2 = o.exact_pressures(mesh.node_coordinates(b), P)
    ctx.set_dirichlet(0, b, e1)
    ctx.set_dirichlet(1, b, e2)
    ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=True)
    xs, info, hist = ctx.solve(_cfg(ksp_type=f.KSP_CG, pc_type=f.PC_BLOCK2 if pc == "block2" else f.PC_JACOBI, rtol=1e-10),
                               hist_cap=4096)
    assert info.converged and 50 < info.iterations < 2000
    r, u0 = ctx.rhs()
    res = r - ctx.spmv(f.MAT_MONO, xs - u0)
    assert np.linalg.norm(res) <= 1e-7 * np.linalg.norm(r)
    xg, ig, _ = ctx.solve(_cfg(pc_type=f.PC_FIELDSPLIT, inner_pc_type=f.PC_MG, inner_rtol=1e-12, rtol=1e-10))
    assert ig.converged and np.abs(xs - xg).max() <= 1e-6 * np.abs(xg).max()


@pytest.mark.gpu
def test_option_paths_agree(gpu_ctx_factory):
    """Alternative code paths kept behind pph_set_option give the same solve: host-driven coarsest CG vs the
    on-chip tail of the cycle, CSR operators vs the stencil-ELL default, the general
    kernel-per-operation V-cycle vs the fused one, eager vs graph-replayed iterations, synchronising vs polled
    fetches, two-pass vs tile assembly."""
    f = _ffi()
    ref = None
    for opts in ({}, {"coarse_on_device": 0}, {"asm_ring": 200}, {"op_format": 0}, {"mg_fused": 0}, {"use_graphs": 0}, {"use_graphs": 2}, {"fetch_spin": 0},
                 {"mg_tail_rows": 50}, {"mg_tail_rows": 0}, {"asm_tile": 0}, {"asm_tile": 2}, {"sell_rpt": 1}, {"sell_sym": 0},
                 {"sell_group": 4}):
        ctx, om, osys = _setup(gpu_ctx_factory, 3, o.CELL_HEX, 12, 8, 16)
        for k, v in opts.items():
            ctx.set_option(k, v)
        if opts:                      # assembly-side options take effect at the next assembly
            ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=True)
        xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                     inner_reduction=1e-1, inner_norm=1, mg_smooth=1, picard_rtol=1e-9))
        assert info.converged
        if ref is None:
            ref = (xs, info.iterations, info.inner_iterations)
        else:
            assert (info.iterations, info.inner_iterations) == ref[1:], opts
            np.testing.assert_allclose(xs, ref[0], rtol=0, atol=1e-10 * np.abs(ref[0]).max(), err_msg=str(opts))


@pytest.mark.gpu
def test_launch_only_sweeps_eager_equals_graph_replay(gpu_ctx_factory):
    """Block solves without a convergence test (inner_norm 2): the warm sweeps are a pure launch sequence, run eagerly by
    default and replayed from a captured graph with use_graphs 2 - same sweeps, iterations and solution bit for bit."""
    f = _ffi()
    out = []
    for graphs in (1, 2):
        ctx, om, osys = _setup(gpu_ctx_factory, 3, o.CELL_HEX, 16, 16, 16)
        ctx.set_option("use_graphs", graphs)
        xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_norm=2, inner_max_it=1,
                                     mg_smooth=1, picard_rtol=1e-9))
        assert info.converged
        out.append((xs, info.iterations, info.inner_iterations))
    assert out[0][1:] == out[1][1:]
    np.testing.assert_array_equal(out[0][0], out[1][0])


@pytest.mark.gpu
@pytest.mark.parametrize("dim,kind,ns", [(2, o.CELL_TRI, (8, 16)), (3, o.CELL_TET, (6, 12))])
def test_error_norms_on_simplices(gpu_ctx_factory, dim, kind, ns):
    """l2_error / h1_seminorm_error on P1 triangles and Kuhn tetrahedra: device quadrature against the oracle's
    (same collapsed Gauss rule; the reference stores no simplex error norms - parity unpinned beyond that) and
    second / first order convergence of the manufactured problem."""
    import math
    f = _ffi()
    Pd = o.Params()
    eta, pi = Pd.eta, math.pi
    if dim == 2:
        ex = lambda X: (Pd.mu / pi) * np.exp(pi * X[:, 0]) * np.sin(pi * X[:, 1]) - (Pd.mu / (Pd.beta * Pd.k1)) * np.exp(eta * X[:, 1])
        gr = lambda X: np.stack([Pd.mu * np.exp(pi * X[:, 0]) * np.sin(pi * X[:, 1]),
                                 Pd.mu * np.exp(pi * X[:, 0]) * np.cos(pi * X[:, 1])
                                 - (Pd.mu / (Pd.beta * Pd.k1)) * eta * np.exp(eta * X[:, 1])], 1)
    else:
        ex = lambda X: ((Pd.mu / pi) * np.exp(pi * X[:, 0]) * (np.sin(pi * X[:, 1]) + np.sin(pi * X[:, 2]))
                        - (Pd.mu / (Pd.beta * Pd.k1)) * (np.exp(eta * X[:, 1]) + np.exp(eta * X[:, 2])))
        gr = lambda X: np.stack([Pd.mu * np.exp(pi * X[:, 0]) * (np.sin(pi * X[:, 1]) + np.sin(pi * X[:, 2])),
                                 Pd.mu * np.exp(pi * X[:, 0]) * np.cos(pi * X[:, 1]) - (Pd.mu / (Pd.beta * Pd.k1)) * eta * np.exp(eta * X[:, 1]),
                                 Pd.mu * np.exp(pi * X[:, 0]) * np.cos(pi * X[:, 2]) - (Pd.mu / (Pd.beta * Pd.k1)) * eta * np.exp(eta * X[:, 2])], 1)
    errs = []
    for n in ns:
        nz = n if dim == 3 else 0
        ctx, om, osys = _setup(gpu_ctx_factory, dim, kind, n, n, nz)
        xs, info, _ = ctx.solve(_cfg(pc_type=f.PC_FIELDSPLIT, inner_pc_type=f.PC_MG, inner_rtol=1e-12, rtol=1e-11))
        assert info.converged
        p1h = xs[: osys.n]
        l2, h1 = ctx.error_norms_mms(0, p1h, Pd.k1, Pd.k2, Pd.beta, Pd.mu, 5)
        rl2, rh1 = o.error_norms(om, p1h, ex, gr, nq=5)
        assert l2 == pytest.approx(rl2, rel=1e-10) and h1 == pytest.approx(rh1, rel=1e-10)
        errs.append((l2, h1))
    assert 1.6 < np.log2(errs[0][0] / errs[1][0]) < 2.4       # L2: O(h^2)
    assert 0.7 < np.log2(errs[0][1] / errs[1][1]) < 1.3       # H1 seminorm: O(h)


@pytest.mark.gpu
def test_assembly_and_solve_on_random_small_shapes(gpu_ctx_factory):
    """Seeded sweep over small, odd-shaped meshes of all four cell kinds, with partial Dirichlet sets that differ
    between the two fields: monolithic matrix and right-hand side against the oracle (1e-12 of the largest entry),
    and the GMRES + field-split solve against the direct solution.  Exercises the tails of every kernel (last
    partial batch, one-cell directions, rows at corners, non-aliased A21)."""
    f = _ffi()
    rng = np.random.default_rng(20260313)
    kinds = [(2, o.CELL_QUAD), (2, o.CELL_TRI), (3, o.CELL_HEX), (3, o.CELL_TET)]
    for it in range(24):
        dim, kind = kinds[it % 4]
        nx, ny = int(rng.integers(1, 8)), int(rng.integers(1, 8))
        nz = int(rng.integers(1, 8)) if dim == 3 else 0
        om = o.build_mesh(dim, kind, nx, ny, nz)
        bnd = o.boundary_nodes(om)
        e1, e2 = o.exact_pressures(om.coords, P)
        if it % 3 == 2:
            # field 1 constrained on part of the boundary only: the two Dirichlet sets differ
            keep = bnd[om.coords[bnd, 0] < 0.75]
            if keep.size == 0:
                keep = bnd
            sets = (bnd, keep)
        else:
            sets = (bnd, bnd)
        ctx = gpu_ctx_factory()
        ctx.mesh_build(dim, kind, nx, ny, nz)
        ctx.set_dirichlet(0, sets[0], e1[sets[0]])
        ctx.set_dirichlet(1, sets[1], e2[sets[1]])
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=True)
        g1 = np.zeros(om.num_nodes); g2 = np.zeros(om.num_nodes)
        g1[sets[0]] = e1[sets[0]]; g2[sets[1]] = e2[sets[1]]
        m1 = np.zeros(om.num_nodes, bool); m2 = np.zeros(om.num_nodes, bool)
        m1[sets[0]] = True; m2[sets[1]] = True
        osys = o.build_system(om, P, g1=g1, g2=g2, mask1=m1, mask2=m2)
        A = ctx.csr(f.MAT_MONO)
        assert abs(A - osys.A).max() <= 1e-12 * abs(osys.A).max(), (it, dim, kind, nx, ny, nz)
        r, u0 = ctx.rhs()
        np.testing.assert_allclose(r, osys.rhs, rtol=0, atol=1e-12 * max(np.abs(osys.rhs).max(), 1.0))
        np.testing.assert_array_equal(u0, osys.u0)
        xs, info, _ = ctx.solve(_cfg(pc_type=f.PC_FIELDSPLIT, inner_pc_type=f.PC_MG, inner_rtol=1e-13, rtol=1e-12))
        ud = o.solve_direct(osys)
        assert info.converged and np.abs(xs - ud).max() <= 1e-8 * max(np.abs(ud).max(), 1.0), (it, dim, kind, nx, ny, nz)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [o.CELL_HEX, o.CELL_TET])
def test_default_path_is_bitwise_reproducible(gpu_ctx_factory, kind):
    """Two independent contexts, default options (fused assembly, MFMA contraction, device-side coarse solve,
    device-resident CG scalars): identical blocks, right-hand side, iteration counts and solution bits."""
    f = _ffi()
    runs = []
    for _ in range(2):
        ctx, om, osys = _setup(gpu_ctx_factory, 3, kind, 20, 12, 16, monolithic=False)
        mats = [ctx.csr(w).data.copy() for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12)]
        rhs, _ = ctx.rhs()
        xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10,
                                     inner_reduction=1e-1, inner_norm=1, mg_smooth=1, picard_rtol=1e-8))
        assert info.converged
        runs.append((mats, rhs, xs, info.iterations, info.inner_iterations, info.resnorm))
    a, b = runs
    for x, y in zip(a[0], b[0]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    assert a[3:] == b[3:]


@pytest.mark.gpu
@pytest.mark.parametrize("hexa,shape", [(True, (77, 45, 53)), (False, (31, 47, 29)), (True, (130, 6, 34))])
def test_odd_mid_size_shapes_match_cpu_port(gpu_ctx_factory, hexa, shape):
    """Shapes that are no multiple of any batch size (tails of every kernel) at a size only the C port reaches:
    blocks and right-hand side entry for entry, and the Jacobi-CG Picard solve against the C port's solution
    (odd meshes have no coarser level: the comparison uses the Jacobi-preconditioned block solves of both)."""
    from oracle import dpp_cpu as cpu

    cpu.set_threads(8)     # fixed reduction order of the C port, whatever the host offers
    f = _ffi()
    nx, ny, nz = shape
    kind = f.CELL_HEX if hexa else f.CELL_TET
    S = cpu.CpuSystem(3, kind, nx, ny, nz)
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, kind, nx, ny, nz)
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(nx, ny, nz, hexahedral=hexa)
    b = mesh.boundary_nodes()
    e1, e2 = o.exact_pressures(mesh.node_coordinates(b), P)
    for tgt in (S, ctx):
        tgt.set_dirichlet(0, b, e1)
        tgt.set_dirichlet(1, b, e2)
    S.assemble(P.k1, P.k2, P.beta, P.mu)
    ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
    for wc, wg in ((cpu.MAT_A11, f.MAT_A11), (cpu.MAT_A22, f.MAT_A22), (cpu.MAT_A12, f.MAT_A12), (cpu.MAT_A21, f.MAT_A21)):
        ref, got = S.csr(wc), ctx.csr(wg)
        np.testing.assert_array_equal(got.indptr, ref.indptr)
        np.testing.assert_array_equal(got.indices, ref.indices)
        np.testing.assert_allclose(got.data, ref.data, rtol=0, atol=1e-12 * np.abs(ref.data).max())
    r_ref, _ = S.rhs()
    r, _ = ctx.rhs()
    np.testing.assert_allclose(r, r_ref, rtol=0, atol=1e-12 * np.abs(r_ref).max())
    S.mg_setup()
    x_ref, sweeps, inner, res = S.picard(pc=cpu.PC_JACOBI, inner_rtol=1e-10, reduction=0.0, inner_norm=0, rtol=1e-9)
    xs, info, _ = ctx.solve(_cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_JACOBI, inner_rtol=1e-10,
                                 picard_rtol=1e-9, picard_max_it=100))
    assert info.converged and sweeps > 0
    assert info.iterations == sweeps
    assert np.abs(xs - x_ref).max() <= 1e-7 * np.abs(x_ref).max()


@pytest.mark.gpu
def test_truncated_coarsest_solve_is_reported(gpu_ctx_factory):
    """A mesh that coarsens only once (34 -> 17 cells: 18^3 = 5832 coarse rows, beyond the on-chip tail) solves its
    coarsest level with the host-driven Jacobi-CG; when that solve stops at its iteration limit the cycle is a
    truncated, non-stationary preconditioner: the solve must say so (pph_solve_info.inner_failed, a warning from
    solve_dpp) instead of passing silently; with the default limit the flag stays clear."""
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, 3, o.CELL_HEX, 34, 34, 34, monolithic=False)
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_reduction=1e-1,
               inner_norm=1, mg_smooth=1, picard_rtol=1e-8)
    xs, info, _ = ctx.solve(cfg)
    assert info.converged and not info.inner_failed
    ctx.set_option("coarse_max_it", 2)
    xs2, info2, _ = ctx.solve(cfg, raise_on_diverged=False)
    assert info2.inner_failed


@pytest.mark.gpu
@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(2, o.CELL_QUAD, 4, 4, 0), (2, o.CELL_QUAD, 8, 8, 0), (2, o.CELL_QUAD, 16, 16, 0),
                                               (2, o.CELL_TRI, 7, 5, 0), (3, o.CELL_HEX, 5, 4, 3), (3, o.CELL_TET, 4, 4, 4),
                                               (3, o.CELL_TET, 8, 8, 8), (2, o.CELL_QUAD, 1, 2, 0)])
def test_ilu0_gmres_matches_oracle(gpu_ctx_factory, goldens, dim, kind, nx, ny, nz):
    """GMRES + ILU(0) (GMRES_ILU_PARAMS, reference parameters.py:27) with the level-scheduled device factorisation
    against the oracle's sequential ILU(0): same iteration count (+-1: different Gram-Schmidt variant), same
    solution.  On 2D Q1 meshes the lexicographic numbering gives the reference's factors: its recorded iteration
    counts (petsc_perf_breakdown.csv 'GMRES + ILU PC': 5, 7, 11 at N = 4, 8, 16) are reproduced; elsewhere the
    reference's DMPlex numbering gives other factors (parity of the counts unpinned)."""
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, dim, kind, nx, ny, nz)
    xs, info, hist = ctx.solve(_cfg(ksp_type=f.KSP_GMRES, pc_type=f.PC_ILU), hist_cap=64)
    ref = o.gmres(osys.A, osys.rhs, o.ilu0_apply(osys.A))
    assert info.converged and abs(info.iterations - ref.its) <= 1
    assert np.abs(xs - (osys.u0 + ref.x)).max() <= 1e-7 * max(np.abs(xs).max(), 1e-300)
    ud = o.solve_direct(osys)
    assert np.abs(xs - ud).max() <= 1e-6 * np.abs(ud).max()
    n0 = min(len(hist), len(ref.history), 4)
    np.testing.assert_allclose(hist[:n0], ref.history[:n0], rtol=1e-8)     # same factors: same preconditioned residuals
    if kind == o.CELL_QUAD and nx == ny and nx in (4, 8, 16):
        g = _perf(goldens, "G7_G9_perf_2d_q1", "GMRES + ILU PC", nx)
        assert info.iterations == g["iterations"]


@pytest.mark.gpu
@pytest.mark.parametrize("dim,kind,nx", [(2, o.CELL_QUAD, 8), (3, o.CELL_TET, 4), (3, o.CELL_TET, 8)])
def test_fieldsplit_gmres_ilu_blocks(gpu_ctx_factory, goldens, dim, kind, nx):
    """FIELDSPLIT_GMRES_ILU_PARAMS (reference parameters.py:50-57): GMRES + multiplicative field-split whose block
    solves are GMRES(30) + ILU(0) on A11 / A22: 4 outer iterations like the reference at every size (golden
    'Scale-Splitting GMRES + ILU PC'), solution = direct solution; the preonly + ILU block variant of
    make_fieldsplit_params_with("ilu") against the oracle's."""
    f = _ffi()
    ctx, om, osys = _setup(gpu_ctx_factory, dim, kind, nx, nx, nx if dim == 3 else 0)
    cfg = _cfg(ksp_type=f.KSP_GMRES, pc_type=f.PC_FIELDSPLIT, inner_ksp_type=f.KSP_GMRES, inner_pc_type=f.PC_ILU,
               inner_rtol=1e-8, inner_atol=1e-12)
    xs, info, _ = ctx.solve(cfg)
    key = "G7_G9_perf_2d_q1" if dim == 2 else "G6_G9_perf_3d_tets"
    g = _perf(goldens, key, "Scale-Splitting GMRES + ILU PC", nx)
    assert info.converged and info.iterations == g["iterations"] == 4
    ud = o.solve_direct(osys)
    assert np.abs(xs - ud).max() <= 1e-6 * np.abs(ud).max()
    cfg = _cfg(ksp_type=f.KSP_GMRES, pc_type=f.PC_FIELDSPLIT, inner_ksp_type=f.KSP_PREONLY, inner_pc_type=f.PC_ILU)
    xs, info, _ = ctx.solve(cfg)
    ref = o.gmres(osys.A, osys.rhs, o.fieldsplit_ilu_gmres_apply(osys.A, osys.n, preonly=True))
    assert info.converged and abs(info.iterations - ref.its) <= 1
    assert np.abs(xs - ud).max() <= 1e-6 * np.abs(ud).max()


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,nz", [(32, 16, 8), (64, 64, 64), (16, 128, 32), (40, 36, 30), (15, 15, 15), (96, 80, 72)])
def test_row_dictionary_products_are_bitwise_the_stored_ones(gpu_ctx_factory, nx, ny, nz):
    """Option sell_dict (pph_internal.h: struct SellDict): the distinct rows of a stencil-ELL block stored once, a 2-byte
    class per row, the product's coefficients from LDS.  Same x loads, same order of the sums: y = A x BIT-identical to the
    stored-value kernel for the three blocks, a whole Picard solve (every epilogue mode, the multigrid levels' dictionaries
    included) with the same sweeps / iterations; with the plain kernel's grid also the same residual history bit for bit.
    A re-assembly with other coefficients keeps the classes (table re-read, every row checked again); more distinct rows
    than the cap, or a failed check on the device, leave the stored values in charge - same results again.  Rows repeat
    bit for bit when the node spacing is exact in binary (cells per direction a power of two: 27 interior / next-to-boundary
    classes + the Dirichlet row); with 40 x 36 x 30 cells the rounding of i / 40 makes 1 574 distinct rows and the
    dictionary is refused - the same assertions then hold on the stored values.
    Round 4: the node assembly kernel integrates a uniform box on its canonical edges (MeshData::uniform), so the rows repeat
    bit for bit at EVERY cell count - 40 x 36 x 30, 15^3 and 96 x 80 x 72 carry the same 28 (+1) classes as 64^3; with
    asm_uniform 0 (stored coordinates) the old behaviour - more distinct rows than the cap, refused automatically - is back
    and is checked on 40 x 36 x 30."""
    exact = True
    f = _ffi()
    import perphil_amd.fd as fdm

    mesh = fdm.UnitCubeMesh(nx, ny, nz, hexahedral=True)
    b = mesh.boundary_nodes()
    g1, g2 = o.exact_pressures(mesh.node_coordinates(b), P)
    rng = np.random.default_rng(21)
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_reduction=1e-1, inner_norm=1,
               mg_smooth=1, picard_rtol=1e-8)
    x = None
    res = {}
    for name in ("plain", "dict", "dict_class_loads", "dict_plain_grid", "cap", "poisoned"):
        ctx = gpu_ctx_factory()
        ctx.set_option("sell_zwalk_min_chunks", 1)
        if name != "plain":
            ctx.set_option("sell_dict", 1)
            ctx.set_option("sell_dict_min_rows", 1)
        if name == "dict_class_loads":    # the walk kernel that loads the classes of every plane (round 4: sell_dict_zconst)
            ctx.set_option("sell_dict_zconst", 0)
        if name == "dict_plain_grid":     # k_spmv_sell<DICT> in the stored-value kernel's chunk order and grid
            ctx.set_option("sell_dict_walk", 0)
            ctx.set_option("sell_dict_blocks", 0)
        if name == "cap":
            ctx.set_option("sell_dict_cap", 4)
        ctx.mesh_build(3, f.CELL_HEX, nx, ny, nz)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        ctx.assemble(P.k1, 3.0 * P.k2, P.beta, P.mu, monolithic=False)
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)      # the re-assembly path: classes kept, table re-read
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)      # ... and with unchanged values: checked by the assembly kernel itself
        t = ctx.timers()
        if name in ("dict", "dict_class_loads", "dict_plain_grid", "poisoned") and exact:
            assert t["dict_operators"] >= 3 and t["dict_status"] == 1 and 8 <= t["dict_classes"] <= 64, t
            # the class of an in-plane position is the same on the planes 2 .. pz - 3 of a box with one Dirichlet set per face:
            # established on the class array at the build, used by the walk kernel (no class loads there) from 8 planes on
            assert t["dict_zconst"] == (name in ("dict", "poisoned") and nz >= 7), (name, t)
        elif name in ("dict", "dict_plain_grid", "poisoned"):
            assert (t["dict_operators"] == 0 and t["dict_status"] == -1) or t["dict_classes"] <= 256, t
        elif name == "cap":
            assert t["dict_operators"] == 0 and t["dict_status"] == -1, t
        else:
            assert t["dict_operators"] == 0 and t["dict_classes"] == 0
        if name == "poisoned":
            ctx.set_option("sell_dict_poison", 1)
        if x is None:
            x = rng.uniform(-1, 1, ctx.n)
        ys = [ctx.spmv(w, x) for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12)]
        xs, info, hist = ctx.solve(cfg, hist_cap=64)
        assert info.converged
        if name == "dict" and exact:
            assert ctx.timers()["dict_operators"] >= 3
            # the same context through the two-step assembly (K, M, then elimination kernels): other kernels rewrite the stored
            # values, the dictionaries must not survive that
            ctx.set_option("asm_fused", 0)
            ctx.assemble(P.k1, 2.0 * P.k2, P.beta, P.mu, monolithic=False)
            assert ctx.timers()["dict_operators"] == 0
            A11 = ctx.csr(f.MAT_A11)
            y2 = ctx.spmv(f.MAT_A11, x)
            ref = A11 @ x
            assert np.abs(y2 - ref).max() <= 1e-13 * np.abs(ref).max()
        res[name] = (ys, xs, (info.iterations, info.inner_iterations), hist)
        ctx.close()
    ys0, xs0, its0, hist0 = res["plain"]
    for name in ("dict", "dict_class_loads", "dict_plain_grid", "cap", "poisoned"):
        ys, xs, its, hist = res[name]
        for ya, yb in zip(ys0, ys):
            np.testing.assert_array_equal(ya, yb)
        assert its == its0, name
        np.testing.assert_allclose(xs, xs0, rtol=0, atol=1e-12 * np.abs(xs0).max())
        np.testing.assert_allclose(hist, hist0, rtol=1e-6)
    np.testing.assert_array_equal(res["dict_class_loads"][1], res["dict"][1])     # (same grid, same sums: the solve bit for bit)
    np.testing.assert_array_equal(res["dict_class_loads"][3], res["dict"][3])
    for name in ("dict_plain_grid", "cap"):
        np.testing.assert_array_equal(res[name][1], xs0)
        np.testing.assert_array_equal(res[name][3], hist0)
    if (nx, ny, nz) == (40, 36, 30):
        # rows that genuinely differ (the stored coordinates i / 40: 1 574 distinct rows) are still refused by themselves
        ctx = gpu_ctx_factory()
        ctx.set_option("asm_uniform", 0)
        ctx.set_option("sell_dict_min_rows", 1)
        ctx.mesh_build(3, f.CELL_HEX, nx, ny, nz)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
        t = ctx.timers()
        assert t["dict_operators"] == 0 and t["dict_status"] == -1, t
        xs, info, _ = ctx.solve(cfg)
        assert info.converged and (info.iterations, info.inner_iterations) == its0
        np.testing.assert_allclose(xs, xs0, rtol=0, atol=1e-11 * np.abs(xs0).max())
        ctx.close()


@pytest.mark.gpu
def test_row_dictionary_with_classes_that_change_along_z(gpu_ctx_factory):
    """Round 4 (sell_dict_zconst): the walk kernel takes the classes of the interior planes from the plane below only when the
    class array says so (k_dict_zconst at the build).  Dirichlet data on the lower half of the boundary only: the rows next
    to the faces change their class half way up, the flag stays off, the classes are loaded on every plane, and products and
    the solve are those of the stored values bit for bit."""
    f = _ffi()
    import perphil_amd.fd as fdm

    N = 16
    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=True)
    b_all = mesh.boundary_nodes()
    xyz = mesh.node_coordinates(b_all)
    b = b_all[xyz[:, 2] <= 0.5]
    g1, g2 = o.exact_pressures(mesh.node_coordinates(b), P)
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_reduction=1e-1, inner_norm=1,
               mg_smooth=1, picard_rtol=1e-8)
    rng = np.random.default_rng(5)
    x = None
    res = {}
    for name in ("plain", "dict"):
        ctx = gpu_ctx_factory()
        ctx.set_option("sell_zwalk_min_chunks", 1)
        ctx.set_option("sell_dict", 1 if name == "dict" else 0)
        ctx.set_option("sell_dict_min_rows", 1)
        ctx.mesh_build(3, f.CELL_HEX, N, N, N)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
        t = ctx.timers()
        if name == "dict":
            assert t["dict_operators"] >= 3 and t["dict_status"] == 1 and not t["dict_zconst"], t
        if x is None:
            x = rng.uniform(-1, 1, ctx.n)
        ys = [ctx.spmv(w, x) for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12)]
        xs, info, _ = ctx.solve(cfg)
        assert info.converged
        res[name] = (ys, xs, (info.iterations, info.inner_iterations))
        ctx.close()
    for ya, yb in zip(res["plain"][0], res["dict"][0]):
        np.testing.assert_array_equal(ya, yb)
    assert res["plain"][2] == res["dict"][2]
    np.testing.assert_allclose(res["dict"][1], res["plain"][1], rtol=0, atol=1e-12 * np.abs(res["plain"][1]).max())


@pytest.mark.gpu
def test_row_dictionary_comes_back_after_being_switched_off_and_retires_after_a_device_refusal(gpu_ctx_factory):
    """(medium) sell_dict 1 -> assemble -> 0 -> assemble -> 1 -> assemble must rebuild the dictionaries - only a
    real refusal (too many distinct rows, a failed check) is remembered per mesh and Dirichlet set; (low) a refusal that
    happens ON THE DEVICE in a re-assembly check reaches the host at the end of the next solve (one word in mapped host
    memory), which then retires the dictionary instead of running every product through the dictionary kernel's one-row
    fallback and counting 2 B per row."""
    f = _ffi()
    import perphil_amd.fd as fdm

    N = 16
    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=True)
    b = mesh.boundary_nodes()
    g1, g2 = o.exact_pressures(mesh.node_coordinates(b), P)
    ctx = gpu_ctx_factory()
    ctx.set_option("sell_dict_min_rows", 1)
    ctx.mesh_build(3, f.CELL_HEX, N, N, N)
    ctx.set_dirichlet(0, b, g1)
    ctx.set_dirichlet(1, b, g2)
    asm = lambda: ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
    asm()
    builds = ctx.timers()["dict_builds"]
    assert ctx.timers()["dict_operators"] >= 3 and builds >= 3 and ctx.timers()["dict_build_ms"] > 0
    ctx.set_option("sell_dict", 0)
    asm()
    assert ctx.timers()["dict_operators"] == 0
    ctx.set_option("sell_dict", 1)
    asm()
    assert ctx.timers()["dict_operators"] >= 3 and ctx.timers()["dict_builds"] >= builds + 3
    nb = ctx.timers()["dict_builds"]
    asm()                                                  # an ordinary re-assembly builds nothing
    assert ctx.timers()["dict_builds"] == nb
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_reduction=1e-1, inner_norm=1,
               mg_smooth=1, picard_rtol=1e-8)
    x0, info0, _ = ctx.solve(cfg)
    before = ctx.timers()["dict_operators"]
    assert info0.converged and before >= 3
    ctx.set_option("sell_dict_poison", 2)                  # refusal on the device + the alarm word a check kernel raises
    x1, info1, _ = ctx.solve(cfg)                          # this solve's launches take the stored values by themselves ...
    assert info1.converged and np.array_equal(x1, x0)
    t = ctx.timers()                                       # ... and at its end the host retired the fine blocks' dictionaries
    assert t["dict_status"] == -2 and t["dict_operators"] == before - 3, t
    x2, info2, _ = ctx.solve(cfg)                          # plain kernels now
    assert info2.converged and np.array_equal(x2, x0)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["interior", "near_boundary"])
def test_fused_dictionary_check_refuses_a_row_that_left_its_class(gpu_ctx_factory, which):
    """Round 4: the per-assembly check of the row dictionaries runs inside the assembly kernel (pph_sell.hip, "check fused into
    the assembly": tables from the representative rows assembled first, every stored entry compared with its class's entry
    before it is stored - the rows of general-form waves read back once -, class adjacencies checked on the tables).  A re-assembly
    with other coefficients keeps the dictionaries without any k_dict_verify pass; one row moved into another class on the
    device (an interior row: the compare inside the straight-line launch; a row next to the boundary: the read-back of the
    general-form rows) makes the NEXT assembly refuse A11's dictionary - and only that one -, the solve still equals the
    stored-value solve bit for bit and the host retires the dictionary at the end of it."""
    f = _ffi()
    import perphil_amd.fd as fdm

    N = 64      # 274 625 rows per block: the two-launch node kernel (asm_node_split_min 200 000), i.e. the fused check's path
    mesh = fdm.UnitCubeMesh(N, N, N, hexahedral=True)
    b = mesh.boundary_nodes()
    g1, g2 = o.exact_pressures(mesh.node_coordinates(b), P)
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_reduction=1e-1, inner_norm=1,
               mg_smooth=1, picard_rtol=1e-8)
    ref = None
    for fuse in (0, 1):
        ctx = gpu_ctx_factory()
        ctx.set_option("sell_dict_min_rows", 100000)
        ctx.set_option("sell_dict_fuse", fuse)
        ctx.mesh_build(3, f.CELL_HEX, N, N, N)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        ctx.assemble(P.k1, 3.0 * P.k2, P.beta, P.mu, monolithic=False)      # builds the dictionaries (+ the fused-check group)
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)            # other coefficients: same classes, new tables
        t = ctx.timers()
        assert t["dict_operators"] >= 3 and t["dict_status"] == 1, t
        x, info, hist = ctx.solve(cfg, hist_cap=32)
        assert info.converged and ctx.timers()["dict_operators"] >= 3
        if ref is None:
            ref = (x.copy(), hist.copy(), (info.iterations, info.inner_iterations))
            ctx.close()
            continue
        np.testing.assert_array_equal(x, ref[0])        # fused check or verify pass: the same operators, the same products
        np.testing.assert_array_equal(hist, ref[1])
        px = N + 1
        row = (px // 2) + px * ((px // 2) + px * (px // 2)) if which == "interior" else 1 + px * ((px // 2) + px * (px // 2))
        before = ctx.timers()["dict_operators"]
        ctx.set_option("sell_dict_corrupt_row", row)
        ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)            # the fused check meets the moved row
        x2, info2, hist2 = ctx.solve(cfg, hist_cap=32)
        assert info2.converged
        np.testing.assert_array_equal(x2, ref[0])       # the refused dictionary's products took the stored values
        t = ctx.timers()
        assert t["dict_status"] == -2 and t["dict_operators"] == before - 1, t
        ctx.close()


@pytest.mark.gpu
def test_large_results_come_back_in_pinned_arrays_that_are_safe_to_keep(gpu_ctx_factory):
    """Context.solution() / solve(fetch=True) hand large results out in page-locked arrays from a pool of three per context
    (pph_host_alloc): an array somebody still refers to - directly or through a view - is never written again, one that
    nobody keeps is reused, a fourth live result falls back to a pageable array; values equal the pageable path's."""
    f = _ffi()
    import perphil_amd.fd as fdm

    N = 80      # 2 x 81^3 = 1 062 882 entries: above the pinning threshold
    ctx = gpu_ctx_factory()
    ctx.mesh_build(3, f.CELL_HEX, N, N, N)
    hm = fdm.UnitCubeMesh(N, N, N, hexahedral=True)
    b = hm.boundary_nodes()
    g1, g2 = o.exact_pressures(hm.node_coordinates(b), P)
    ctx.set_dirichlet(0, b, g1)
    ctx.set_dirichlet(1, b, g2)
    ctx.assemble(P.k1, P.k2, P.beta, P.mu, monolithic=False)
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_reduction=1e-1, inner_norm=1,
               mg_smooth=1, picard_rtol=1e-8)
    x1, info, _ = ctx.solve(cfg)
    assert info.converged and 2 * ctx.n >= ctx._PIN_MIN
    ref = x1.copy()
    addr = lambda a: a.__array_interface__["data"][0]
    a1 = addr(x1)
    view = x1[5:50]                       # a view keeps the block busy
    del x1
    x2 = ctx.solution()
    assert addr(x2) != a1 and np.array_equal(x2, ref) and np.array_equal(view, ref[5:50])
    del view
    x3 = ctx.solution()                   # the first block is free again
    assert addr(x3) == a1 and np.array_equal(x3, ref)
    x4, x5 = ctx.solution(), ctx.solution()
    assert len({addr(x2), addr(x3), addr(x4), addr(x5)}) == 4 and np.array_equal(x5, ref)   # the fourth live one is pageable
    assert len(ctx._pinned) == 3
    ctx.close()
    assert np.array_equal(x2, ref)        # results outlive their context


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny", [(64, 64), (256, 128), (100, 60)])
def test_row_dictionary_on_quadrilateral_blocks(gpu_ctx_factory, nx, ny):
    """The 2D counterpart (k_spmv_sell<quad, mode, 2, sym, DICT>: 9-point rows, no walk kernel): products of the three blocks
    bit-identical to the stored-value kernel, a Picard solve with the same sweeps / iterations; 100 x 60 cells (spacing not exact
    in binary) keeps its stored values unless its distinct rows fit the table."""
    f = _ffi()
    import perphil_amd.fd as fdm

    mesh = fdm.UnitSquareMesh(nx, ny, quadrilateral=True)
    b = mesh.boundary_nodes()
    P2 = o.Params()
    g1, g2 = o.exact_pressures(mesh.node_coordinates(b), P2)
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_reduction=1e-1, inner_norm=1,
               mg_smooth=1, picard_rtol=1e-8)
    rng = np.random.default_rng(5)
    x = None
    res = {}
    for name in ("plain", "dict"):
        ctx = gpu_ctx_factory()
        ctx.set_option("sell_dict", 1 if name == "dict" else 0)
        ctx.set_option("sell_dict_min_rows", 1)
        ctx.mesh_build(2, f.CELL_QUAD, nx, ny, 0)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        ctx.assemble(P2.k1, P2.k2, P2.beta, P2.mu, monolithic=False)
        ctx.assemble(P2.k1, P2.k2, P2.beta, P2.mu, monolithic=False)
        t = ctx.timers()
        if name == "dict" and nx & (nx - 1) == 0 and ny & (ny - 1) == 0:
            assert t["dict_operators"] >= 3 and t["dict_status"] == 1 and 4 <= t["dict_classes"] <= 16, t
        if x is None:
            x = rng.uniform(-1, 1, ctx.n)
        ys = [ctx.spmv(w, x) for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12)]
        xs, info, hist = ctx.solve(cfg, hist_cap=64)
        assert info.converged
        res[name] = (ys, xs, (info.iterations, info.inner_iterations))
        ctx.close()
    for ya, yb in zip(res["plain"][0], res["dict"][0]):
        np.testing.assert_array_equal(ya, yb)
    assert res["plain"][2] == res["dict"][2]
    np.testing.assert_allclose(res["dict"][1], res["plain"][1], rtol=0, atol=1e-12 * np.abs(res["plain"][1]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,dims", [("tet", (32, 16, 16)), ("tet", (24, 20, 12)), ("tri", (64, 32))])
def test_row_dictionary_on_simplex_blocks(gpu_ctx_factory, kind, dims):
    """P1 blocks (15-point rows on Kuhn tetrahedra, 7-point rows on triangles; k_spmv_sell<kind, mode, 2, sym, DICT>): products of
    the three blocks bit-identical to the stored-value kernel, a Picard solve with the same sweeps / iterations."""
    f = _ffi()
    import perphil_amd.fd as fdm

    if kind == "tet":
        mesh = fdm.UnitCubeMesh(*dims)
        dim, ck, PP = 3, f.CELL_TET, P
    else:
        mesh = fdm.UnitSquareMesh(*dims)
        dim, ck, PP = 2, f.CELL_TRI, o.Params()
    b = mesh.boundary_nodes()
    g1, g2 = o.exact_pressures(mesh.node_coordinates(b), PP)
    cfg = _cfg(picard=1, inner_ksp_type=f.KSP_CG, inner_pc_type=f.PC_MG, inner_rtol=1e-10, inner_reduction=1e-1, inner_norm=1,
               mg_smooth=1, picard_rtol=1e-8)
    rng = np.random.default_rng(9)
    x = None
    res = {}
    exact = all(v & (v - 1) == 0 for v in dims)
    for name in ("plain", "dict"):
        ctx = gpu_ctx_factory()
        ctx.set_option("sell_dict", 1 if name == "dict" else 0)
        ctx.set_option("sell_dict_min_rows", 1)
        ctx.mesh_build(dim, ck, *(dims if dim == 3 else dims + (0,)))
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        ctx.assemble(PP.k1, PP.k2, PP.beta, PP.mu, monolithic=False)
        ctx.assemble(PP.k1, PP.k2, PP.beta, PP.mu, monolithic=False)
        t = ctx.timers()
        if name == "dict" and exact:
            assert t["dict_operators"] >= 3 and t["dict_status"] == 1 and 4 <= t["dict_classes"] <= 128, t
        if x is None:
            x = rng.uniform(-1, 1, ctx.n)
        ys = [ctx.spmv(w, x) for w in (f.MAT_A11, f.MAT_A22, f.MAT_A12)]
        xs, info, hist = ctx.solve(cfg, hist_cap=64)
        assert info.converged
        res[name] = (ys, xs, (info.iterations, info.inner_iterations))
        ctx.close()
    for ya, yb in zip(res["plain"][0], res["dict"][0]):
        np.testing.assert_array_equal(ya, yb)
    assert res["plain"][2] == res["dict"][2]
    np.testing.assert_allclose(res["dict"][1], res["plain"][1], rtol=0, atol=1e-12 * np.abs(res["plain"][1]).max())
