"""BASELINE config 5 on one GPU: P1 Kuhn tets, k1/k2 = 1e4, GMRES + multiplicative field-split with multigrid-CG
block solves (and the Picard variant), timing + residual check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from perphil_amd import _ffi, fd, DPPParameters, exact_expressions_3d
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
k1, k2, beta, mu = 1.0, 1e-4, 1.0, 1.0
mesh = fd.UnitCubeMesh(N, N, N)
b = mesh.boundary_nodes(); X = mesh.node_coordinates(b)
_, p1, _, p2 = exact_expressions_3d(mesh, DPPParameters(k1=k1, k2=k2, beta=beta, mu=mu))
ctx = _ffi.Context(0)
t0 = time.perf_counter(); ctx.mesh_build(3, _ffi.CELL_TET, N, N, N); ctx.synchronize(); t1 = time.perf_counter()
print(f"tets N={N}: cells {ctx.ncell} dofs {2*ctx.n} nnz_block {ctx.nnzb} mesh {1e3*(t1-t0):.1f} ms", flush=True)
ctx.set_dirichlet(0, b, p1(X)); ctx.set_dirichlet(1, b, p2(X))
for name, picard, mono in (("picard", 1, False), ("gmres_fieldsplit", 0, True)):
    cfg = _ffi.SolverCfg()
    cfg.ksp_type, cfg.pc_type, cfg.restart, cfg.max_it, cfg.rtol, cfg.atol = _ffi.KSP_GMRES, _ffi.PC_FIELDSPLIT, 30, 200, 1e-8, 1e-12
    cfg.inner_ksp_type, cfg.inner_pc_type, cfg.inner_max_it, cfg.inner_rtol, cfg.inner_atol = _ffi.KSP_CG, _ffi.PC_MG, 500, 1e-10, 1e-300
    cfg.picard, cfg.picard_rtol, cfg.picard_atol, cfg.picard_max_it, cfg.mg_smooth = picard, 1e-8, 1e-12, 100, 1
    cfg.inner_reduction = 1e-1 if picard else 0.0
    cfg.inner_norm = 1 if picard else 0
    for rep in range(2):
        ctx.set_option("invalidate_KM", 1)
        t0 = time.perf_counter()
        ctx.assemble(k1, k2, beta, mu, monolithic=mono)
        _, info, _ = ctx.solve(cfg, fetch=False)
        ctx.synchronize(); t1 = time.perf_counter()
    tm = ctx.timers()
    print(f"{name}: {1e3*(t1-t0):.1f} ms (assemble {tm['assemble_ms']+tm['bc_blocks_ms']:.1f}, solve {tm['solve_ms']:.1f}) "
          f"its {info.iterations} inner {info.inner_iterations} res {info.resnorm:.3e} rhs {info.rhs_norm:.3e} "
          f"-> {2*ctx.n/(t1-t0)/1e6:.1f} MDoF/s", flush=True)
