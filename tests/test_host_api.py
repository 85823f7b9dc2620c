"""Host-side mirror of the reference interface (no GPU needed): behavioural tests ported from the
reference's own unit tests onto the build's objects."""
import numpy as np
import pytest

import perphil_amd as pa
from perphil_amd import fd, solver_parameters as sp
from perphil_amd.solver import translate_options
from perphil_amd import _ffi


def _mixed_space(nx=2, ny=2, quad=True):
    mesh = pa.create_mesh(nx, ny, quadrilateral=quad)
    _, V = pa.create_function_spaces(mesh)
    return fd.MixedFunctionSpace((V, V))


def test_parameters_auto_constant_coercion_and_defaults():
    # reference models/dpp/_tests/test_parameters.py:10-17
    params = pa.DPPParameters(k1=2.0, k2=None, beta=3.0, mu=4.0)
    for c in (params.k1, params.k2, params.beta, params.mu):
        assert isinstance(c, fd.Constant)
    assert float(params.k2) == pytest.approx(float(params.k1) / params.scale_contrast)


def test_eta_computed_property():
    # reference test_parameters.py:20-23
    params = pa.DPPParameters(k1=1.0, k2=0.01, beta=1.0, mu=1.0)
    assert hasattr(params.eta, "ufl_shape") and params.eta.ufl_shape == ()
    assert float(params.eta) == pytest.approx(np.sqrt(1.01 / 0.01))


def test_mesh_and_spaces():
    # reference mesh/_tests/test_mesh.py, forms/_tests/test_spaces.py, test_perf_to_dict_regression.yml (dofs 18, cells 4)
    W = _mixed_space()
    assert W.num_sub_spaces() == 2 and W.dim() == 18 and W.mesh().num_cells() == 4
    assert pa.create_mesh(2, 2, quadrilateral=False).num_cells() == 8
    m3 = fd.UnitCubeMesh(4, 4, 4)
    V = fd.FunctionSpace(m3, "CG", 1)
    assert (V * V).dim() == 250 and m3.num_cells() == 384  # petsc_perf_breakdown_3d.csv row nx=4
    assert fd.UnitCubeMesh(4, 4, 4, hexahedral=True).num_cells() == 64


def test_dpp_form_guards_and_structure():
    # reference forms/_tests/test_dpp.py:23-40 and test_dpp_form_structure_regression.yml
    mesh = pa.create_mesh(2, 2)
    _, V = pa.create_function_spaces(mesh)
    with pytest.raises(ValueError):
        pa.dpp_form(V, pa.DPPParameters())
    a, L = pa.dpp_form(fd.MixedFunctionSpace((V, V)), pa.DPPParameters())
    assert a.rank == 2 and a.num_integrals == 4 and L.rank == 1
    F, fields = pa.dpp_splitted_form(fd.MixedFunctionSpace((V, V)), pa.DPPParameters())
    assert isinstance(fields, fd.Function) and F.rank == 1


def test_solve_dpp_raises_on_non_mixed_space():
    # reference solvers/_tests/test_solver.py:37-42 — the guard fires before any device work
    mesh = pa.create_mesh(2, 2)
    _, V = pa.create_function_spaces(mesh)
    with pytest.raises(ValueError):
        pa.solve_dpp(V, pa.DPPParameters(), bcs=[])
    with pytest.raises(ValueError):
        pa.solve_dpp_nonlinear(V, pa.DPPParameters(), bcs=[])


def test_solver_parameter_dicts():
    # reference solvers/_tests/test_solver_parameters.py
    assert sp.LINEAR_SOLVER_PARAMS["pc_type"] == "lu" and sp.LINEAR_SOLVER_PARAMS["ksp_type"] == "preonly"
    assert sp.GMRES_PARAMS["ksp_rtol"] == 1e-8 and sp.GMRES_PARAMS["ksp_atol"] == 1e-12
    assert sp.PLAIN_GMRES_PARAMS["pc_type"] == "none" and sp.GMRES_JACOBI_PARAMS["pc_type"] == "jacobi"
    assert sp.FIELDSPLIT_LU_PARAMS["pc_fieldsplit_type"] == "multiplicative"
    assert sp.FIELDSPLIT_LU_PARAMS["fieldsplit_0"] is sp.LINEAR_SOLVER_PARAMS
    assert sp.PICARD_LU_SOLVER_PARAMS["snes_type"] == "ngs" and sp.PICARD_LU_SOLVER_PARAMS["snes_rtol"] == 1e-8


def test_option_translation():
    cfg, info = translate_options(sp.PLAIN_GMRES_PARAMS)
    assert (cfg.ksp_type, cfg.pc_type, cfg.restart) == (_ffi.KSP_GMRES, _ffi.PC_NONE, 30)
    assert (cfg.rtol, cfg.atol, cfg.max_it) == (1e-8, 1e-12, 50000) and not cfg.picard
    cfg, info = translate_options(sp.LINEAR_SOLVER_PARAMS)
    assert info["direct_equivalent"] and cfg.pc_type == _ffi.PC_FIELDSPLIT and cfg.rtol <= 1e-12
    cfg, info = translate_options({})  # Firedrake's default for LinearVariationalSolver is a direct solve
    assert info["direct_equivalent"]
    cfg, _ = translate_options({**sp.GMRES_PARAMS, **sp.FIELDSPLIT_LU_PARAMS})
    assert cfg.pc_type == _ffi.PC_FIELDSPLIT and cfg.inner_pc_type == _ffi.PC_MG and cfg.inner_rtol == 1e-12
    cfg, _ = translate_options({**sp.GMRES_PARAMS, **sp.FIELDSPLIT_GMRES_PARAMS})
    assert cfg.inner_pc_type == _ffi.PC_NONE and cfg.inner_ksp_type == _ffi.KSP_GMRES and cfg.inner_rtol == 1e-8
    cfg, _ = translate_options({**sp.GMRES_PARAMS, **sp.FIELDSPLIT_GMRES_ILU_PARAMS})
    assert cfg.inner_pc_type == _ffi.PC_ILU and cfg.inner_ksp_type == _ffi.KSP_GMRES
    cfg, _ = translate_options(sp.PICARD_LU_SOLVER_PARAMS, nonlinear=True)
    assert cfg.picard == 1 and cfg.picard_rtol == 1e-8 and cfg.picard_atol == 1e-12
    # flattened fieldsplit keys (iterative_bench.make_fieldsplit_params_with)
    cfg, _ = translate_options({**sp.GMRES_PARAMS, **sp.FIELDSPLIT_LU_PARAMS, "fieldsplit_0_pc_type": "jacobi",
                                "fieldsplit_1_pc_type": "jacobi", "fieldsplit_0_ksp_type": "cg", "fieldsplit_1_ksp_type": "cg"})
    assert cfg.inner_pc_type == _ffi.PC_JACOBI
    # inexact Picard preset (the benchmark's algorithm): sub-solver norm type and reduction target
    cfg, _ = translate_options(sp.PICARD_MG_INEXACT_SOLVER_PARAMS, nonlinear=True)
    assert (cfg.picard, cfg.inner_pc_type, cfg.inner_norm, cfg.mg_smooth) == (1, _ffi.PC_MG, 1, 1)
    assert cfg.inner_reduction == 0.1 and cfg.inner_rtol == 1e-10
    cfg, _ = translate_options(sp.PICARD_MG_SOLVER_PARAMS, nonlinear=True)
    assert cfg.inner_norm == 0 and cfg.inner_reduction == 0.0
    cfg, _ = translate_options(sp.PICARD_MG_FIXED_SOLVER_PARAMS, nonlinear=True)     # ksp_norm_type none + ksp_max_it 1
    assert (cfg.inner_norm, cfg.inner_max_it, cfg.mg_smooth, cfg.inner_ksp_type) == (2, 1, 1, _ffi.KSP_CG)
    with pytest.raises(NotImplementedError):
        translate_options({**sp.PICARD_MG_SOLVER_PARAMS, "fieldsplit_0_ksp_norm_type": "none",
                           "fieldsplit_1_ksp_norm_type": "none"}, nonlinear=True)       # no ksp_max_it given
    with pytest.raises(NotImplementedError):
        translate_options({**sp.PICARD_MG_SOLVER_PARAMS, "fieldsplit_0_ksp_norm_type": "natural"}, nonlinear=True)
    cfg, info = translate_options(sp.GMRES_ILU_PARAMS)          # the stated preconditioner, no substitution
    assert cfg.pc_type == _ffi.PC_ILU and cfg.ksp_type == _ffi.KSP_GMRES
    with pytest.raises(NotImplementedError):
        translate_options({**sp.GMRES_ILU_PARAMS, "pc_factor_levels": 1})
    # Firedrake's defaults: ksp_rtol 1e-7 unless given; a user-set pc_type without ksp_type leaves the Krylov
    # method to PETSc (gmres); neither set = direct solve (parity of these two defaults is unpinned)
    cfg, info = translate_options({"pc_type": "jacobi"})
    assert cfg.ksp_type == _ffi.KSP_GMRES and cfg.pc_type == _ffi.PC_JACOBI and cfg.rtol == 1e-7
    cfg, info = translate_options(sp.FIELDSPLIT_LU_PARAMS)
    assert cfg.ksp_type == _ffi.KSP_GMRES and cfg.pc_type == _ffi.PC_FIELDSPLIT and not info["direct_equivalent"]
    cfg, info = translate_options({"ksp_type": "cg", "pc_type": "jacobi"})
    assert cfg.rtol == 1e-7
    from perphil_amd import iterative_bench as ib
    cfg, _ = translate_options(ib.make_fieldsplit_params_with("ilu"))
    assert (cfg.ksp_type, cfg.inner_ksp_type, cfg.inner_pc_type) == (_ffi.KSP_GMRES, _ffi.KSP_PREONLY, _ffi.PC_ILU)
    cfg, _ = translate_options(ib.make_fieldsplit_params_with("lu"))
    assert cfg.inner_pc_type == _ffi.PC_MG and cfg.inner_rtol == 1e-12
    with pytest.raises(NotImplementedError):
        translate_options({"ksp_type": "bcgs", "pc_type": "none"})
    with pytest.raises(NotImplementedError):
        translate_options({**sp.GMRES_PARAMS, **sp.FIELDSPLIT_LU_PARAMS, "pc_fieldsplit_type": "schur"})


def test_boundary_nodes_and_manufactured_data_match_oracle():
    from oracle import dpp_oracle as o

    for mesh, om in ((pa.create_mesh(5, 3), o.build_mesh(2, o.CELL_QUAD, 5, 3)),
                     (fd.UnitCubeMesh(3, 4, 2, hexahedral=True), o.build_mesh(3, o.CELL_HEX, 3, 4, 2))):
        np.testing.assert_array_equal(mesh.boundary_nodes(), o.boundary_nodes(om))
        np.testing.assert_array_equal(mesh.node_coordinates(), om.coords)
        params = pa.DPPParameters(k1=1.0, k2=0.01)
        ex = pa.exact_expressions if mesh.dim == 2 else pa.exact_expressions_3d
        _, p1, _, p2 = ex(mesh, params)
        e1, e2 = o.exact_pressures(om.coords, o.Params(k1=1.0, k2=0.01))
        np.testing.assert_array_equal(p1(om.coords), e1)
        np.testing.assert_array_equal(p2(om.coords), e2)
    W = _mixed_space(4, 4)
    bc = fd.DirichletBC(W.sub(1), fd.Constant(2.5), "on_boundary")
    nodes, vals = bc.nodes_and_values()
    assert bc.field == 1 and len(nodes) == 16 and np.all(vals == 2.5)
    with pytest.raises(NotImplementedError):
        fd.DirichletBC(W.sub(0), 0.0, 1)


def test_function_views_and_split():
    W = _mixed_space(2, 2)
    w = fd.Function(W, np.arange(18, dtype=float))
    p1, p2 = w.split()
    assert p1.vector()[0] == 0 and p2.vector()[0] == 9
    p2.vector()[0] = -1.0  # views
    assert w.vector()[9] == -1.0
    assert w.sub(0).at((0.5, 0.5)) == 4.0


def test_bench_gpus_n_launches_itself(monkeypatch):
    # `python bench.py --gpus N` without a launcher starts torch.distributed.run as a child (one rank per GPU,
    # rendezvous on 127.0.0.1) with the same arguments and returns its exit code; nothing touches a GPU before that
    import subprocess
    import sys

    import bench

    seen = {}

    class _Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_convergence_2d_host_logic(goldens):
    """perphil_amd.convergence_2d (mirror of reference experiments/convergence_2d.py): solver lists, row schema, and the
    observed-order fit checked against the reference's own convergence.csv -> convergence_eoc.csv pair (no GPU:
    the rows are the committed ones)."""
    from perphil_amd import convergence_2d as c2

    specs = c2._default_solvers([1e-8, 1e-10])                        # reference convergence_2d.py:118-133
    assert [s.name for s in specs] == ["mumps", "gmres_rtol=1e-08", "fs-lu_gmres_rtol=1e-08", "gmres_rtol=1e-10",
                                       "fs-lu_gmres_rtol=1e-10"]
    assert specs[1].params["ksp_rtol"] == 1e-8 and specs[1].params["pc_type"] == "none"
    assert specs[4].params["ksp_type"] == "gmres" and specs[4].params["ksp_atol"] == 1e-12
    assert specs[4].params["pc_type"] == "fieldsplit" and "ksp_rtol" not in sp.FIELDSPLIT_LU_PARAMS   # presets untouched
    assert c2.ROW_FIELDS == goldens["convergence_csv_columns"]
    assert [s.name for s in c2.approach_solvers()] == sorted({r["solver"] for r in goldens["G10_convergence_2d"]})
    got = {(r["solver"], r["err"]): r["slope"] for r in c2.observed_orders(goldens["G10_convergence_2d"])}
    assert len(got) == 20
    for r in goldens["G10_convergence_2d_eoc"]:
        assert got[(r["solver"], r["err"])] == pytest.approx(r["slope"], rel=1e-12)
    with pytest.raises(NotImplementedError):
        c2.run_one(4, specs[0], quad=True, degree=2, params=pa.DPPParameters())
