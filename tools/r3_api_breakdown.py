"""Where the wall time of a second solve_dpp_nonlinear() call goes at 256^3 (bench.py: config.api.api_wall_ms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import perphil_amd as pa
from perphil_amd import fd, solver_parameters as spar
from perphil_amd.manufactured_solutions import exact_expressions_3d

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh = fd.UnitCubeMesh(N, N, N, hexahedral=True)
V = fd.FunctionSpace(mesh, "CG", 1)
W = V * V
params = pa.DPPParameters(k1=1.0, k2=1e-2, beta=1.0, mu=1.0)
_u1, p1e, _u2, p2e = exact_expressions_3d(mesh, params)
bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
for rep in range(3):
    t = time.perf_counter()
    sol = pa.solve_dpp_nonlinear(W, params, bcs, solver_parameters=spar.PICARD_MG_INEXACT_SOLVER_PARAMS)
    print(f"call {rep}: {1e3 * (time.perf_counter() - t):.2f} ms, timers {sol.info['timers']['assemble_ms']:.2f} + {sol.info['timers']['solve_ms']:.2f}", flush=True)
ctx = mesh.context()
for rep in range(3):
    t = time.perf_counter(); x = ctx.solution(); print(f"ctx.solution(): {1e3 * (time.perf_counter() - t):.2f} ms", flush=True)
t = time.perf_counter(); y = np.empty_like(x); y[:] = 0.0; print(f"np.empty + first touch of 272 MB: {1e3 * (time.perf_counter() - t):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
sol = pa.solve_dpp_nonlinear(W, params, bcs, solver_parameters=spar.PICARD_MG_INEXACT_SOLVER_PARAMS)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
