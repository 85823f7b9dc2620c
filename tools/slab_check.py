#!/usr/bin/env python3
"""Multi-rank parity check of the slab-decomposed HIP path (launched by torch.distributed.run):
every rank solves its slab; rank 0 also solves the whole cube on one context and compares solution,
Picard sweeps and inner CG iteration totals.  Exit code 0 = parity."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=16)
ap.add_argument("--backend", default="gloo")
ap.add_argument("--kind", default="hex")
ap.add_argument("--inner-pc", default="mg")
ap.add_argument("--solver", default="picard", choices=["picard", "gmres_fs", "cg_block2", "gmres_jacobi"])
ap.add_argument("--inexact", action="store_true",
                help="the benchmark's Picard settings: V(1,1), block solves to a tenfold drop of the unpreconditioned residual")
ap.add_argument("--device-scalars", action="store_true",
                help="run the device-scalar CG branch (what the RCCL transport executes) over the callback transport")
ap.add_argument("--halo-overlap", action="store_true",
                help="products of large levels are split into interior and boundary rows, the halo exchange overlapping "
                     "the interior rows: must equal the same split without overlap bit for bit (and the single context as usual)")
ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE", help="extra pph_set_option settings on the slab contexts")
ap.add_argument("--fail-halo-after", type=int, default=-1,
                help="every rank's halo callback fails from this call on: the solve must return a COMM error, not a result")
args = ap.parse_args()

from perphil_amd import _ffi  # noqa: E402  (before torch: the library binds the system HIP runtime first)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
from perphil_amd.distributed import SlabSolver  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
ndev = torch.cuda.device_count()
device = local_rank % max(ndev, 1)
torch.cuda.set_device(device)
dist.init_process_group(backend=args.backend)
kind = _ffi.CELL_HEX if args.kind == "hex" else _ffi.CELL_TET
pc = _ffi.PC_MG if args.inner_pc == "mg" else _ffi.PC_JACOBI
k1, k2, beta, mu = 1.0, 1e-2, 1.0, 1.0
if args.inexact:
    solver = SlabSolver(args.cells, world, rank, device, k1, k2, beta, mu, kind=kind, inner_pc=pc, smooth=1,
                        inner_reduction=0.1, inner_norm=1)
else:
    solver = SlabSolver(args.cells, world, rank, device, k1, k2, beta, mu, kind=kind, inner_pc=pc)
if args.device_scalars:
    solver.ctx.set_option("device_scalars", 1)
for kv in args.set:
    solver.ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
if args.fail_halo_after >= 0:
    solver.comm.fail_halo_after = args.fail_halo_after
    try:
        solver.step()
    except RuntimeError as e:
        ok = "-5" in str(e) or "failed" in str(e)
        print(f"rank {rank}: solve refused after the injected halo failure: {e}", flush=True)
    else:
        ok = False
        print(f"rank {rank}: solve returned a result although its halo exchanges failed", flush=True)
    # a later solve on the same context must keep failing (sticky status) without touching the transport
    try:
        solver.step()
        ok = False
    except RuntimeError:
        pass
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"world={world} injected halo failure: {'COMM error on every rank' if flag.item() == 1.0 else 'NOT reported'}; max rel diff n/a", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)
serial_split = None
if args.halo_overlap:
    # the three-launch split without overlap first, then the overlapped run that the comparison below uses
    solver.ctx.set_option("halo_overlap_min_rows", 1000)
    solver.ctx.set_option("halo_overlap", 2)
    info_s = solver.step()
    serial_split = (solver.gather_solution().copy(), info_s.iterations, info_s.inner_iterations)
    solver.ctx.set_option("halo_overlap", 1)
mono = args.solver != "picard"
if mono:
    solver.cfg.picard = 0
    solver.cfg.inner_rtol = 1e-12
    solver.cfg.ksp_type = _ffi.KSP_CG if args.solver == "cg_block2" else _ffi.KSP_GMRES
    solver.cfg.pc_type = {"gmres_fs": _ffi.PC_FIELDSPLIT, "cg_block2": _ffi.PC_BLOCK2, "gmres_jacobi": _ffi.PC_JACOBI}[args.solver]
    solver.monolithic = True
info = solver.step()
full = solver.gather_solution()
ok = True
cap = [float(kv.split("=")[1]) for kv in args.set if kv.split("=")[0] == "part_cap"]
if cap:
    # the three launches of a split product share one reduction slot's partial-sum area: never more than part_cap of them
    got = solver.ctx.timers()["max_split_partials"]
    print(f"rank {rank}: split products wrote at most {got} partial sums (cap {int(cap[0])})", flush=True)
    ok = ok and 0 < got <= cap[0]
sym = solver.ctx.timers()["symmetric_storage"]
if not mono and not sym:
    print(f"rank {rank}: the slab operators are not in symmetric stencil-ELL storage", flush=True)
    ok = False
if serial_split is not None:
    same = np.array_equal(full, serial_split[0]) and (info.iterations, info.inner_iterations) == serial_split[1:]
    nsplit = solver.ctx.timers()["split_products"]
    print(f"rank {rank}: overlapped halo run {'==' if same else '!='} serial split (bitwise), sweeps {info.iterations}, inner {info.inner_iterations}, "
          f"{nsplit} products split", flush=True)
    ok = ok and same and nsplit > 0
if rank == 0:
    ctx = _ffi.Context(device)
    ctx.mesh_build(3, kind, args.cells, args.cells, args.cells)
    b, g1, g2 = bench.mms_boundary(args.cells, k1, k2, beta, mu) if kind == _ffi.CELL_HEX else (None, None, None)
    if b is None:
        from perphil_amd import fd, DPPParameters, exact_expressions_3d
        mesh = fd.UnitCubeMesh(args.cells, args.cells, args.cells)
        b = mesh.boundary_nodes()
        _, p1, _, p2 = exact_expressions_3d(mesh, DPPParameters(k1=k1, k2=k2, beta=beta, mu=mu))
        X = mesh.node_coordinates(b)
        g1, g2 = p1(X), p2(X)
    ctx.set_dirichlet(0, b, g1)
    ctx.set_dirichlet(1, b, g2)
    ctx.assemble(k1, k2, beta, mu, monolithic=mono)
    x1, info1, _ = ctx.solve(solver.cfg)
    err = np.abs(full - x1).max() / np.abs(x1).max()
    print(f"world={world} n={args.cells} kind={args.kind} solver={args.solver} pc={args.inner_pc}: sweeps {info.iterations} vs {info1.iterations}, "
          f"inner its {info.inner_iterations} vs {info1.inner_iterations}, residual {info.resnorm:.3e} vs {info1.resnorm:.3e}, "
          f"max rel diff {err:.3e}, halo calls {solver.comm.halo_calls}, allreduce calls {solver.comm.allreduce_calls}, "
          f"symmetric storage {sym}, row dictionaries on {solver.ctx.timers()['dict_operators']} slab operators", flush=True)
    ok = ok and (err < (1e-9 if not mono else 1e-7) and abs(info.iterations - info1.iterations) <= (0 if not mono else 2) and abs(info.inner_iterations - info1.inner_iterations) <= max(1, info1.inner_iterations // 50)
          and info.converged == 1)
flag = torch.tensor([1.0 if ok else 0.0])
if args.backend == "nccl":
    flag = flag.cuda()
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if flag.item() == 1.0 else 1)
