#!/usr/bin/env python3
"""
Headline benchmark of the DPP hot path (BASELINE.json): DoF/s over assemble + solve on the 3D
UnitCube Q1 two-pressure problem, solved by block Picard (fixed-stress) sweeps whose block solves
are multigrid-preconditioned CG on the CSR blocks.

One "step" = one pass of the hot path on inputs resident in HBM: integrate K and M (cell kernels +
scatter-add), eliminate Dirichlet dofs / form the DPP blocks and the lifted right-hand side, run the
Picard solve to snes_rtol 1e-8.  The mesh connectivity / CSR pattern and the boundary data are built
before the timed region, like the mesh construction that precedes the reference's timing window
(reference src/perphil/experiments/petsc_profiling_3d.py:57 vs :82-86).

    python bench.py --gpus 1 --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (CSR SpMV kernel,
HIP events per launch on the solver's stream, algorithmic bytes 12 nnz + 20 nrows) and
`cpu_baseline` (the NumPy/SciPy oracle of the same algorithm on a bounded sample, 1 core).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def picard_cfg(_ffi, inner_rtol=1e-10, smooth=2, reduction=0.0, inner_norm=0):
    cfg = _ffi.SolverCfg()
    cfg.ksp_type, cfg.pc_type, cfg.restart, cfg.max_it = _ffi.KSP_GMRES, _ffi.PC_FIELDSPLIT, 30, 50000
    cfg.rtol, cfg.atol = 1e-8, 1e-12
    cfg.inner_ksp_type, cfg.inner_pc_type, cfg.inner_max_it = _ffi.KSP_CG, _ffi.PC_MG, 50000
    cfg.inner_rtol, cfg.inner_atol = inner_rtol, 1e-300
    cfg.picard, cfg.picard_rtol, cfg.picard_atol, cfg.picard_max_it = 1, 1e-8, 1e-12, 100
    cfg.mg_smooth = smooth
    cfg.inner_reduction = reduction
    cfg.inner_norm = inner_norm
    return cfg


def mms_boundary(n_cells, k1, k2, beta, mu):
    """Boundary node ids and manufactured Dirichlet values of the global unit cube."""
    from perphil_amd import fd
    from perphil_amd.parameters import DPPParameters
    from perphil_amd.manufactured_solutions import exact_expressions_3d

    mesh = fd.UnitCubeMesh(n_cells, n_cells, n_cells, hexahedral=True)
    b = mesh.boundary_nodes()
    X = mesh.node_coordinates(b)
    _, p1, _, p2 = exact_expressions_3d(mesh, DPPParameters(k1=k1, k2=k2, beta=beta, mu=mu))
    return b, p1(X), p2(X)


def _mms_boundary_np(N, k1, k2, beta, mu):
    from oracle import dpp_oracle as o

    px = N + 1
    idx = np.arange(px ** 3)
    i, j, k = idx % px, (idx // px) % px, idx // (px * px)
    on = (i == 0) | (i == N) | (j == 0) | (j == N) | (k == 0) | (k == N)
    X = np.stack([i[on] / N, j[on] / N, k[on] / N], 1)
    g1, g2 = o.exact_pressures(X, o.Params(k1=k1, k2=k2, beta=beta, mu=mu))
    return idx[on], g1, g2


def _cpu_port_run(N, threads, k1, k2, beta, mu, smooth, reduction, inner_rtol, inner_norm):
    """One step (assemble + multigrid setup + inexact Picard) of the C/OpenMP restatement; seconds, sweeps, its."""
    from oracle import dpp_cpu as cpu

    cpu.set_threads(threads)
    S = cpu.CpuSystem(3, 2, N, N, N)          # mesh + pattern: outside the timed region, as on the GPU
    b, g1, g2 = _mms_boundary_np(N, k1, k2, beta, mu)
    S.set_dirichlet(0, b, g1)
    S.set_dirichlet(1, b, g2)
    # two steps, the second one timed: like the GPU's warm-up step, the first pays for allocation and first touch
    for _ in range(2):
        t0 = time.perf_counter()
        S.assemble(k1, k2, beta, mu)
        S.mg_setup()
        _, sweeps, inner, _ = S.picard(pc=cpu.PC_MG, inner_rtol=inner_rtol, reduction=reduction, smooth=smooth, rtol=1e-8,
                                       atol=1e-12, max_it=100, inner_norm=inner_norm)
        t = time.perf_counter() - t0
    spmv_gbs = (12.0 * S.nnz + 20.0 * S.n) / S.spmv_seconds(cpu.MAT_A11, 10) / 1e9
    dofs = 2 * S.n
    scipy_gbs = None
    if threads == 1:
        # independent single-thread SpMV sanity point (SciPy CSR, same fine-level scalar block)
        A = S.csr(cpu.MAT_A11)
        xr = np.random.default_rng(20260313).uniform(-1, 1, S.n)
        ts = time.perf_counter()
        for _ in range(10):
            A @ xr
        scipy_gbs = 10 * (12.0 * A.nnz + 20.0 * S.n) / (time.perf_counter() - ts) / 1e9
    S.close()
    return t, sweeps, inner, dofs, spmv_gbs, scipy_gbs


def host_cores(cap=16):
    """Threads the CPU baseline may use: the affinity mask clipped by the cgroup CPU quota (a GPU box exposes
    every core of the host in the mask but grants a share of them) and by `cap`."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                       # cgroup v2
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = int(fq.read()), int(fp.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def cpu_baseline(sample_n, k1, k2, beta, mu, smooth=1, reduction=1e-1, inner_rtol=1e-10, threads=0, inner_norm=1):
    """The CPU port (oracle/dpp_cpu.c: C99 + OpenMP restatement of the same algorithm - assembly, multigrid
    setup, inexact Picard with multigrid-CG block solves) timed on the host: all available cores on a
    `sample_n`^3 cube (bounded sample of the 256^3 workload) and one core on 64^3; returns DoF/s."""
    cores = threads if threads > 0 else host_cores()
    small = min(64, sample_n)
    t1, sw1, in1, dofs1, gbs1, scipy_gbs = _cpu_port_run(small, 1, k1, k2, beta, mu, smooth, reduction, inner_rtol, inner_norm)
    t, sw, inner, dofs, gbs, _ = _cpu_port_run(small, cores, k1, k2, beta, mu, smooth, reduction, inner_rtol, inner_norm)
    # largest cube up to sample_n whose predicted time stays within a minute and whose ~2.3 KB per node fit in memory
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 16 << 30
    pick = small
    cand = sample_n
    while cand > small:
        if t * (cand / small) ** 3 < 60.0 and 1.5 * 2300.0 * (cand + 1) ** 3 < avail:
            pick = cand
            break
        cand //= 2
    if pick > small:
        t, sw, inner, dofs, gbs, _ = _cpu_port_run(pick, cores, k1, k2, beta, mu, smooth, reduction, inner_rtol, inner_norm)
    sample_n = pick
    return {"value": dofs / t, "unit": "DoF/s", "cores": cores, "kind": "port",
            "single_core_value": dofs1 / t1, "spmv_gbs": round(gbs, 1), "single_core_spmv_gbs": round(gbs1, 1),
            "scipy_spmv_gbs": round(scipy_gbs, 1),
            "sample": f"{sample_n}^3 Q1 unit cube ({dofs} DoF), same algorithm as the GPU step (assemble + multigrid setup + "
                      f"inexact Picard: {sw} sweeps, {inner} CG iterations, V({smooth},{smooth}), reduction {reduction:g} of the "
                      f"{'unpreconditioned' if inner_norm else 'preconditioned'} residual) in "
                      f"C/OpenMP on {cores} threads, {t:.1f} s (second step, buffers warm); single_core_value: {small}^3 on 1 thread, "
                      f"{t1:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=256, help="cells per direction of the unit cube")
    ap.add_argument("--cpu-sample-n", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: cgroup share, at most 16)")
    ap.add_argument("--inner-rtol", type=float, default=1e-10)
    ap.add_argument("--smooth", type=int, default=1)
    ap.add_argument("--inner-reduction", type=float, default=1e-1)
    ap.add_argument("--inner-norm", type=int, default=1, choices=(0, 1),
                    help="norm tested by the block solves: 0 preconditioned, 1 unpreconditioned")
    ap.add_argument("--asm-kernel", type=int, default=2)
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="extra pph_set_option settings for A/B runs (e.g. spmv_kernel=8)")
    ap.add_argument("--skip-fine-bench", action="store_true", help="omit the isolated fine-level SpMV loop (PMC passes)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch  # device selection, barrier, synchronize: plumbing only
    from perphil_amd import _ffi

    dist = None
    device = local_rank
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        # "nccl" is RCCL on ROCm.  PERPHIL_DIST_BACKEND=gloo rehearses the multi-rank path on a box with
        # fewer GPUs than ranks (ranks then share devices; communication is staged through the host).
        backend = os.environ.get("PERPHIL_DIST_BACKEND", "nccl")
        device = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
    N = args.cells
    k1, k2, beta, mu = 1.0, 1e-2, 1.0, 1.0

    if world > 1:
        from perphil_amd.distributed import SlabSolver

        solver = SlabSolver(N, world, rank, device, k1, k2, beta, mu, inner_rtol=args.inner_rtol, smooth=args.smooth,
                            inner_reduction=args.inner_reduction, inner_norm=args.inner_norm)
        solver.ctx.set_option("asm_kernel", args.asm_kernel)
        for kv in args.set:
            solver.ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        dofs_global = solver.global_dofs
        step = solver.step
        ctx = solver.ctx
    else:
        ctx = _ffi.Context(device)
        ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
        b, g1, g2 = mms_boundary(N, k1, k2, beta, mu)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        cfg = picard_cfg(_ffi, args.inner_rtol, args.smooth, args.inner_reduction, args.inner_norm)
        ctx.set_option("asm_kernel", args.asm_kernel)
        for kv in args.set:
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        dofs_global = 2 * ctx.n
        last = {}

        def step():
            ctx.set_option("invalidate_KM", 1)          # integrate K and M again: assembly is part of the step
            ctx.assemble(k1, k2, beta, mu, monolithic=False)
            _, info, _ = ctx.solve(cfg, fetch=False)
            last["info"] = info
            return info

    def fence():
        torch.cuda.synchronize()
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = ctx.timers()

    # ---- roofline of the dominant kernel: one more (untimed) step with an event pair around every
    # SpMV launch on the solver's stream; bytes are algorithmic (12 nnz + 20 nrows per launch) --------
    ctx.set_option("time_spmv", 1)
    step()
    ctx.synchronize()
    tr = ctx.timers()
    ctx.set_option("time_spmv", 0)
    launches = tr["spmv_launches"] + tr["spmv_dot_launches"]
    ms = tr["spmv_ms"] + tr["spmv_dot_ms"]
    byts = tr["spmv_bytes"] + tr["spmv_dot_bytes"]
    achieved = (byts / 1e9) / (ms / 1e3) if ms > 0 else 0.0
    # fine-level scalar-block SpMV alone (the inner loop the 50 % target is stated on)
    fine_bytes = 12.0 * ctx.nnzb + 20.0 * ctx.n
    fine_ms = float("nan") if args.skip_fine_bench else ctx.spmv_bench(_ffi.MAT_A11, 200)
    # HBM traffic of the same kernel mix from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KB),
    # measured offline with tools/pmc_summarize.py and committed under profiles/ (cannot be sampled in-process)
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_spmv_bench256.json")
    defaults = (args.inner_norm == 1 and args.inner_reduction == 1e-1 and args.smooth == 1 and args.inner_rtol == 1e-10
                and args.asm_kernel == 2 and not args.set)
    if os.path.exists(pmc_file) and N == 256 and world == 1 and defaults:
        try:
            with open(pmc_file) as f:
                traffic = json.load(f).get("traffic_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "kernel": "k_spmv_wide<8,*,2> (aligned-wide CSR-vector SpMV, all multigrid levels of one step)",
        "launches_per_step": int(launches), "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
        "algorithmic_bytes_per_launch": round(byts / max(launches, 1), 0),
        "fine_level": {"avg_launch_ms": round(fine_ms, 4) if fine_ms == fine_ms else None, "algorithmic_bytes": fine_bytes,
                       "achieved": round(fine_bytes / 1e9 / (fine_ms / 1e3), 1) if fine_ms == fine_ms else None,
                       "frac": round(fine_bytes / 1e9 / (fine_ms / 1e3) / HBM_PEAK_GBS, 4) if fine_ms == fine_ms else None},
    }
    if tr["spmv_fine_launches"] and tr["spmv_fine_ms"] > 0:
        # the fine-level launches of the instrumented step itself (cold x, operators alternating), beside the isolated loop
        fa = tr["spmv_fine_bytes"] / 1e9 / (tr["spmv_fine_ms"] / 1e3)
        roofline["fine_level_in_solver"] = {"launches_per_step": tr["spmv_fine_launches"],
                                            "avg_launch_ms": round(tr["spmv_fine_ms"] / tr["spmv_fine_launches"], 4),
                                            "achieved": round(fa, 1), "frac": round(fa / HBM_PEAK_GBS, 4)}

    out = {
        "metric": "DoF/s (assemble+solve), 3D UnitCube Q1 DPP, Picard-split",
        "value": dofs_global * args.steps / elapsed,
        "unit": "DoF/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"3D UnitCube {N}^3 Q1, two-pressure DPP (k1=1, k2=1e-2, beta=mu=1), manufactured Dirichlet data, "
                        f"assemble + block Picard (fixed-stress) to snes_rtol 1e-8 (true residual), warm-started block solves = "
                        f"CG + geometric multigrid (Chebyshev-Jacobi V({args.smooth},{args.smooth})) on CSR blocks, each to a "
                        f"reduction of the {'unpreconditioned' if args.inner_norm else 'preconditioned'} residual by "
                        f"{args.inner_reduction:g} (or rtol {args.inner_rtol:g})",
            "cells": N ** 3, "dofs": int(dofs_global), "parallelism": f"slab{world}" if world > 1 else "single",
            "picard_sweeps": int(info.iterations), "inner_cg_iterations": int(info.inner_iterations),
            "picard_ms_per_sweep": round(tm["solve_ms"] / max(int(info.iterations), 1), 3),
            "assemble_ms": round(tm["assemble_ms"] + tm["bc_blocks_ms"], 3), "solve_ms": round(tm["solve_ms"], 3),
            "final_residual": float(info.resnorm), "rhs_norm": float(info.rhs_norm),
            "halo_exchanges_per_step": int(tm.get("halo_exchanges", 0)),
        },
        "roofline": roofline,
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, k1, k2, beta, mu, args.smooth, args.inner_reduction,
                                           args.inner_rtol, args.cpu_threads, args.inner_norm)
    elif rank == 0:
        out["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        ctx.close()  # destroys the library's RCCL communicator before the process group goes away
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
