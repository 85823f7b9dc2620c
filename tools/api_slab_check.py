#!/usr/bin/env python3
"""Multi-rank parity check THROUGH THE PUBLIC API (launched by torch.distributed.run): every rank calls
solve_dpp / solve_dpp_nonlinear on the reference's objects - fd.UnitCubeMesh (any nx, ny, nz; hexahedra or
Kuhn tetrahedra), V * V, DirichletBC with the manufactured pressures, one of the option dictionaries - exactly as
a single-process caller does (reference src/perphil/solvers/solver.py:30-76, experiments/petsc_profiling_3d.py:31-86);
under the initialised process group the mesh is this rank's cell slab.  Every rank then solves the same problem on a
COMM_SELF mesh (one context, whole cube) and compares: solution to 1e-15 of its largest entry where the iterates are
the same (Picard, CG), to the Krylov tolerance for GMRES (the Gram-Schmidt sums run in slab order), equal outer
iteration counts.  Exit code 0 = parity on every rank."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, nargs=3, default=[16, 16, 16], metavar=("NX", "NY", "NZ"))
ap.add_argument("--kind", default="hex", choices=["hex", "tet", "quad"])
ap.add_argument("--backend", default="gloo")
ap.add_argument("--params", default="PICARD_MG_SOLVER_PARAMS",
                help="name of an option dictionary of perphil_amd.solver_parameters")
ap.add_argument("--nonlinear", action="store_true", help="solve_dpp_nonlinear instead of solve_dpp")
ap.add_argument("--contrast", type=float, default=1e2, help="k1 / k2")
ap.add_argument("--constant-bc", action="store_true", help="Constant / nodal-array Dirichlet data instead of the manufactured pressures")
ap.add_argument("--inject-rccl-failure", action="store_true",
                help="ask for the RCCL transport and make its start-up fail: the run must continue on the callbacks, labelled")
ap.add_argument("--tol", type=float, default=0.0, help="solution tolerance (0: 1e-13 for Picard / CG dictionaries, 2e-7 for GMRES)")
args = ap.parse_args()

import perphil_amd as pa  # noqa: E402
from perphil_amd import fd, solver_parameters as spar  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
dist.init_process_group(backend=args.backend)

nx, ny, nz = args.cells
params = pa.DPPParameters(k1=1.0, k2=1.0 / args.contrast, beta=1.0, mu=1.0)
opts = getattr(spar, args.params)
solve = pa.solve_dpp_nonlinear if args.nonlinear else pa.solve_dpp


def problem(comm):
    if args.kind == "quad":
        mesh = fd.UnitSquareMesh(nx, ny, quadrilateral=True, comm=comm)
        ex = pa.exact_expressions
    else:
        mesh = fd.UnitCubeMesh(nx, ny, nz, hexahedral=(args.kind == "hex"), comm=comm)
        ex = pa.exact_expressions_3d
    if comm == fd.COMM_WORLD and args.inject_rccl_failure:
        mesh.distribute(transport="rccl", inject_rccl_failure=True)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    _, p1, _, p2 = ex(mesh, params)
    if args.constant_bc:
        # a Constant on field 0 and a GLOBAL nodal array on field 1 (what a caller holds who knows nothing of slabs)
        X = mesh.node_coordinates()
        bcs = [fd.DirichletBC(W.sub(0), fd.Constant(2.5), "on_boundary"),
               fd.DirichletBC(W.sub(1), 1.0 + X[:, 0] + 2.0 * X[:, 1] * X[:, -1], "on_boundary")]
    else:
        bcs = [fd.DirichletBC(W.sub(0), p1, "on_boundary"), fd.DirichletBC(W.sub(1), p2, "on_boundary")]
    return mesh, W, bcs


mesh, W, bcs = problem(fd.COMM_WORLD)
sol = solve(W, params, bcs, solver_parameters=opts)
ok = True
if args.kind != "quad":
    ok = ok and mesh.distributed and sol.info["distributed"]["world"] == world
    want = "torch-" + args.backend if (args.backend != "nccl" or args.inject_rccl_failure) else "rccl"
    got = sol.info["distributed"]["transport"]
    if got != want or (args.inject_rccl_failure and not sol.info["distributed"]["rccl_error"]):
        print(f"rank {rank}: transport {got} (rccl_error {sol.info['distributed']['rccl_error']!r}), expected {want}", flush=True)
        ok = False
    s = mesh.slab
    own = sol.solution.owned()
    ok = ok and own.shape == (2 * len(s.owned_planes) * s.plane,) and sol.solution.vector().shape == (2 * s.n_local,)
    # a second call on the same objects re-uses the slab context and its boundary data
    sol2 = solve(W, params, bcs, solver_parameters=opts)
    ok = ok and np.array_equal(sol2.solution.vector(), sol.solution.vector()) and sol2.iteration_number == sol.iteration_number
else:
    ok = ok and (not mesh.distributed) and "replicated" in sol.info
full = sol.gather().solution.vector()

mesh1, W1, bcs1 = problem(fd.COMM_SELF)
ref = solve(W1, params, bcs1, solver_parameters=opts)
x1 = ref.solution.vector()
err = float(np.abs(full - x1).max() / np.abs(x1).max())
gmres = opts.get("ksp_type", "") == "gmres" and not args.nonlinear
tol = args.tol if args.tol > 0 else (2e-7 if gmres else 1e-13)
its_ok = abs(sol.iteration_number - ref.iteration_number) <= (1 if gmres else 0)
ok = ok and err <= tol and its_ok and full.shape == x1.shape
print(f"rank {rank}/{world} {args.kind} {nx}x{ny}x{nz} {args.params}{' nonlinear' if args.nonlinear else ''}: iterations "
      f"{sol.iteration_number} vs {ref.iteration_number}, inner {sol.info['inner_iterations']} vs {ref.info['inner_iterations']}, "
      f"residual {sol.residual_error:.3e} vs {ref.residual_error:.3e}, max rel diff {err:.3e} (tol {tol:g}), "
      f"transport {sol.info.get('distributed', {}).get('transport', 'replicated')}: {'ok' if ok else 'MISMATCH'}", flush=True)
flag = torch.tensor([1.0 if ok else 0.0])
if args.backend == "nccl":
    flag = flag.cuda()
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print(f"world={world} api parity: {'ok' if flag.item() == 1.0 else 'FAILED'}", flush=True)
dist.barrier()
mesh.context().close()
dist.destroy_process_group()
sys.exit(0 if flag.item() == 1.0 else 1)
