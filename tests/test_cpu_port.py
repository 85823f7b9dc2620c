"""The C/OpenMP restatement (oracle/dpp_cpu.c) against the reference's goldens and, entry for entry, against
the NumPy oracle.  It is the CPU baseline that bench.py times beside the HIP path."""
import numpy as np
import pytest

from oracle import dpp_cpu as cpu
from oracle import dpp_mg_oracle as mgo
from oracle import dpp_oracle as o

P = o.Params(k1=1.0, k2=0.01, beta=1.0, mu=1.0)
CASES = [(2, o.CELL_QUAD, 6, 4, 0), (2, o.CELL_TRI, 4, 6, 0), (3, o.CELL_HEX, 4, 2, 6), (3, o.CELL_TET, 4, 6, 2)]


def _system(dim, kind, nx, ny, nz, params=P):
    om = o.build_mesh(dim, kind, nx, ny, nz)
    S = cpu.CpuSystem(dim, kind, nx, ny, nz)
    b = o.boundary_nodes(om)
    g1, g2 = o.exact_pressures(om.coords[b], params)
    S.set_dirichlet(0, b, g1)
    S.set_dirichlet(1, b, g2)
    S.assemble(params.k1, params.k2, params.beta, params.mu)
    return om, S, b


@pytest.mark.parametrize("dim,kind,nx,ny,nz", CASES)
def test_mesh_matrices_rhs_match_numpy_oracle(dim, kind, nx, ny, nz):
    om, S, b = _system(dim, kind, nx, ny, nz)
    cells, coords = S.mesh()
    np.testing.assert_array_equal(cells, om.cells)
    np.testing.assert_array_equal(coords, om.coords)
    K, M = o.assemble_scalar(om)
    for which, ref in ((cpu.MAT_K, K), (cpu.MAT_M, M)):
        A = S.csr(which)
        np.testing.assert_array_equal(A.indptr, ref.indptr)
        np.testing.assert_array_equal(A.indices, ref.indices)
        np.testing.assert_allclose(A.data, ref.data, rtol=0, atol=1e-13 * np.abs(ref.data).max())
    osys = o.build_system(om, P)
    n = osys.n
    A = osys.A.tocsr()
    for which, ref in ((cpu.MAT_A11, A[:n, :n]), (cpu.MAT_A22, A[n:, n:]), (cpu.MAT_A12, A[:n, n:]), (cpu.MAT_A21, A[n:, :n])):
        d = (S.csr(which) - ref).tocoo()
        assert np.abs(d.data).max(initial=0.0) <= 1e-13 * np.abs(ref.data).max()
    r, u0 = S.rhs()
    np.testing.assert_allclose(r, osys.rhs, rtol=0, atol=1e-12 * np.abs(osys.rhs).max())
    np.testing.assert_array_equal(u0, osys.u0)


def test_G1_initial_residual_and_G2_slice(goldens):
    """Reference goldens: ||F(u0)|| of the 10x10 Q1 problem (ipynb :410) and the x = 0.5 slice of the
    solution (ipynb :317-322), here through the C port's Picard loop run to the fixed point."""
    om, S, b = _system(2, o.CELL_QUAD, 10, 10, 0)
    r, _ = S.rhs()
    assert np.linalg.norm(r) == pytest.approx(goldens["G1_initial_residual_10x10"], rel=5e-13)
    S.mg_setup()
    x, sweeps, inner, res = S.picard(pc=cpu.PC_MG, inner_rtol=1e-13, reduction=0.0, smooth=2, rtol=1e-13, max_it=500)
    assert sweeps > 0
    n = S.n
    g = goldens["G2_slice_x05_monolithic_10x10"]
    _, s1 = o.slice_along_x(om, x[:n], 0.5)
    _, s2 = o.slice_along_x(om, x[n:], 0.5)
    np.testing.assert_allclose(s1, g["p1"], rtol=1e-8)
    np.testing.assert_allclose(s2, g["p2"], rtol=1e-8)


@pytest.mark.parametrize("dim,kind,nx,ny,nz", [(2, o.CELL_QUAD, 8, 8, 0), (2, o.CELL_TRI, 8, 4, 0), (3, o.CELL_HEX, 8, 4, 8),
                                               (3, o.CELL_TET, 4, 4, 8)])
def test_vcycle_and_mg_pcg_match_mg_oracle(dim, kind, nx, ny, nz):
    om, S, b = _system(dim, kind, nx, ny, nz)
    nlev = S.mg_setup()
    a, bb, c = P.abc
    mask = np.zeros(S.n, bool)
    mask[b] = True
    rng = np.random.default_rng(3)
    r = rng.standard_normal(S.n)
    r[mask] = 0.0
    for which, cK in ((0, a), (1, c)):
        L = mgo.build_hierarchy(dim, kind, nx, ny, nz, cK, bb, mask)
        assert len(L) == nlev
        for smooth in (1, 2):
            z = S.vcycle(which, r, smooth)
            ref = mgo.vcycle(L, r, smooth)
            np.testing.assert_allclose(z, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
        A = S.csr(cpu.MAT_A11 if which == 0 else cpu.MAT_A22)
        res = o.pcg(A, r, lambda v: mgo.vcycle(L, v, 2), rtol=1e-10)
        x, its, rn = S.pcg(which, cpu.PC_MG, r, rtol=1e-10, smooth=2)
        assert its == res.its
        np.testing.assert_allclose(x, res.x, rtol=0, atol=1e-9 * np.abs(res.x).max())
        resj = o.pcg(A, r, o.jacobi_apply(A), rtol=1e-8)
        xj, itsj, _ = S.pcg(which, cpu.PC_JACOBI, r, rtol=1e-8)
        assert itsj == resj.its
        # unpreconditioned-norm test (KSP_NORM_UNPRECONDITIONED), cold and warm-started with a reduction target
        resu = o.pcg(A, r, lambda v: mgo.vcycle(L, v, 1), rtol=1e-9, norm="unpreconditioned")
        xu, itsu, rnu = S.pcg(which, cpu.PC_MG, r, rtol=1e-9, smooth=1, norm=1)
        assert itsu == resu.its and rnu == pytest.approx(resu.resnorm, rel=1e-6)
        x0 = 0.9 * resu.x
        resw = o.pcg(A, r, lambda v: mgo.vcycle(L, v, 1), rtol=1e-12, x0=x0, reduction=0.1, norm="unpreconditioned")
        xw, itsw, _ = S.pcg(which, cpu.PC_MG, r, x0=x0, rtol=1e-12, reduction=0.1, smooth=1, norm=1)
        assert itsw == resw.its
        np.testing.assert_allclose(xw, resw.x, rtol=0, atol=1e-9 * np.abs(resw.x).max())


@pytest.mark.parametrize("norm,red", [(0, 1e-2), (1, 1e-1)])
def test_picard_matches_direct_solution_3d(norm, red):
    om, S, b = _system(3, o.CELL_HEX, 8, 8, 8)
    S.mg_setup()
    x, sweeps, inner, res = S.picard(inner_norm=norm, reduction=red)   # (1, 0.1): the bench's settings, V(1,1)
    assert 0 < sweeps <= 12
    osys = o.build_system(om, P)
    ud = o.solve_direct(osys)
    assert np.abs(x - ud).max() <= 1e-7 * np.abs(ud).max()
    assert res <= 1e-8 * np.linalg.norm(osys.rhs)


def test_thread_count_does_not_change_the_answer():
    om, S, b = _system(3, o.CELL_HEX, 8, 8, 8)
    S.mg_setup()
    t = cpu.num_threads()
    cpu.set_threads(1)
    x1, s1, i1, _ = S.picard()
    cpu.set_threads(min(max(t, 2), 4))
    x2, s2, i2, _ = S.picard()
    cpu.set_threads(t)
    assert (s1, i1) == (s2, i2)
    np.testing.assert_allclose(x1, x2, rtol=0, atol=1e-9 * np.abs(x1).max())
