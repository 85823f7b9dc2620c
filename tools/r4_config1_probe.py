"""Round 4: where the 16 x 16 plumbing case (BASELINE config 1, LINEAR_SOLVER_PARAMS through solve_dpp) spends its time."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import perphil_amd as pa
from perphil_amd import fd, solver_parameters as spar
warnings.simplefilter("ignore")
params = pa.DPPParameters(k1=1.0, k2=1e-2, beta=1.0, mu=1.0)
mesh = pa.create_mesh(16, 16, quadrilateral=True)
_, V = pa.create_function_spaces(mesh)
W = fd.MixedFunctionSpace((V, V))
_, p1, _, p2 = pa.exact_expressions(mesh, params)
bcs = [fd.DirichletBC(W.sub(0), p1, "on_boundary"), fd.DirichletBC(W.sub(1), p2, "on_boundary")]
for name in ("LINEAR_SOLVER_PARAMS", "FIELDSPLIT_LU_PARAMS", "GMRES_JACOBI_PARAMS", "PICARD_MG_SOLVER_PARAMS"):
    opts = getattr(spar, name)
    solve = pa.solve_dpp_nonlinear if name.startswith("PICARD") else pa.solve_dpp
    opts = {**spar.GMRES_PARAMS, **opts} if name == "FIELDSPLIT_LU_PARAMS" else opts
    solve(W, params, bcs, solver_parameters=opts)
    t0 = time.perf_counter()
    for _ in range(5):
        sol = solve(W, params, bcs, solver_parameters=opts)
    ms = 1e3 * (time.perf_counter() - t0) / 5
    t = sol.info["timers"]
    print(f"{name}: {ms:.2f} ms per call; its {sol.info['iterations']} inner {sol.info['inner_iterations']} assemble {t['assemble_ms']:.2f} bc {t['bc_blocks_ms']:.2f} solve {t['solve_ms']:.2f}", flush=True)
