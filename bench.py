#!/usr/bin/env python3
"""
Headline benchmark of the DPP hot path (BASELINE.json): DoF/s over assemble + solve on the 3D
UnitCube Q1 two-pressure problem, solved by block Picard (fixed-stress) sweeps whose block solves
are multigrid-preconditioned CG.

One "step" = one pass of the hot path on inputs resident in HBM: integrate the element matrices and
form the Dirichlet-eliminated DPP blocks + lifted right-hand side (fused assembly), refresh the
multigrid operators, run the Picard solve to snes_rtol 1e-8.  NOT in the timed step: the mesh
connectivity / sparsity pattern, the upload of the boundary data and the allocation of the operator
and multigrid buffers (the reference's timed `solve_dpp` call allocates its matrix inside the window,
reference src/perphil/solvers/solver.py:65-69 under experiments/petsc_profiling_3d.py:82-86; here
that one-off cost is reported beside the step as `config.setup_ms` and `config.cold_step_ms`).

    python bench.py --gpus 1 --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (the SpMV kernel of the
block solves on the stencil-ELL operator format, HIP events per launch on the solver's stream, algorithmic
bytes 8 S nrows + 16 nrows; the CSR kernel's figures - 12 nnz + 20 nrows - beside it under `roofline.csr`)
and `cpu_baseline` (oracle/dpp_cpu.c, the C/OpenMP restatement of the same step, on the host's cores).
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


def picard_cfg(_ffi, inner_rtol=1e-10, smooth=2, reduction=0.0, inner_norm=0, inner_max_it=50000):
    cfg = _ffi.SolverCfg()
    cfg.ksp_type, cfg.pc_type, cfg.restart, cfg.max_it = _ffi.KSP_GMRES, _ffi.PC_FIELDSPLIT, 30, 50000
    cfg.rtol, cfg.atol = 1e-8, 1e-12
    cfg.inner_ksp_type, cfg.inner_pc_type, cfg.inner_max_it = _ffi.KSP_CG, _ffi.PC_MG, inner_max_it
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


def _cpu_port_run(N, threads, k1, k2, beta, mu, smooth, reduction, inner_rtol, inner_norm, keep=None, kind=2, steps=2,
                  rtol=1e-8):
    """One step (assemble + multigrid setup + inexact Picard) of the C/OpenMP restatement; seconds, sweeps, its."""
    from oracle import dpp_cpu as cpu

    cpu.set_threads(threads)
    S = cpu.CpuSystem(3, kind, N, N, N)       # mesh + pattern: outside the timed region, as on the GPU
    b, g1, g2 = _mms_boundary_np(N, k1, k2, beta, mu)
    S.set_dirichlet(0, b, g1)
    S.set_dirichlet(1, b, g2)
    # two steps, the second one timed: like the GPU's warm-up step, the first pays for allocation and first touch
    for _ in range(steps):
        t0 = time.perf_counter()
        S.assemble(k1, k2, beta, mu)
        S.mg_setup()
        x, sweeps, inner, res = S.picard(pc=cpu.PC_MG, inner_rtol=inner_rtol, reduction=reduction, smooth=smooth, rtol=rtol,
                                         atol=1e-12, max_it=100, inner_norm=inner_norm)
        t = time.perf_counter() - t0
    if keep is not None:      # what main() compares the GPU step with (parity_vs_port)
        keep.update({"cells": N, "sweeps": int(sweeps), "inner": int(inner), "final_residual": float(res), "x": x})
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


def cpu_baseline(sample_n, k1, k2, beta, mu, smooth=1, reduction=1e-1, inner_rtol=1e-10, threads=0, inner_norm=1, keep=None):
    """The CPU port (oracle/dpp_cpu.c: C99 + OpenMP restatement of the same algorithm - assembly, multigrid
    setup, inexact Picard with multigrid-CG block solves) timed on the host: all available cores on a
    `sample_n`^3 cube (bounded sample of the 256^3 workload) and one core on 64^3; returns DoF/s."""
    cores = threads if threads > 0 else host_cores()
    small = min(64, sample_n)
    t1, sw1, in1, dofs1, gbs1, scipy_gbs = _cpu_port_run(small, 1, k1, k2, beta, mu, smooth, reduction, inner_rtol, inner_norm)
    t, sw, inner, dofs, gbs, _ = _cpu_port_run(small, cores, k1, k2, beta, mu, smooth, reduction, inner_rtol, inner_norm, keep)
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
        t, sw, inner, dofs, gbs, _ = _cpu_port_run(pick, cores, k1, k2, beta, mu, smooth, reduction, inner_rtol, inner_norm, keep)
    sample_n = pick
    return {"value": dofs / t, "unit": "DoF/s", "cores": cores, "kind": "port",
            "single_core_value": dofs1 / t1, "spmv_gbs": round(gbs, 1), "single_core_spmv_gbs": round(gbs1, 1),
            "scipy_spmv_gbs": round(scipy_gbs, 1),
            "sample": f"{sample_n}^3 Q1 unit cube ({dofs} DoF), same algorithm as the GPU step (assemble + multigrid setup + "
                      f"inexact Picard: {sw} sweeps, {inner} CG iterations, V({smooth},{smooth}), reduction {reduction:g} of the "
                      f"{'unpreconditioned' if inner_norm else 'preconditioned'} residual) in "
                      f"C/OpenMP on {cores} threads, {t:.1f} s (second step, buffers warm); single_core_value: {small}^3 on 1 thread, "
                      f"{t1:.1f} s"}


def self_launch(nproc):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N bench.py <same arguments>` as a child process (one rank per GPU, rendezvous on 127.0.0.1 at a
    free port), pass its output through and return its exit code."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def parity_vs_port(ctx_solution, info, keep):
    """GPU step against the C/OpenMP port on the same problem (reference
    src/perphil/experiments/petsc_profiling_3d.py:147-150 reads iterations / residual off one more solve the same
    way): sweeps and inner iterations equal, final residual to 1e-3 relative, the solution to 1e-9 of its largest
    entry over ALL entries, its 2-norm to 1e-9, and 8 probe nodes per field listed for the record."""
    xg, xp = ctx_solution, keep["x"]
    n = xp.size // 2
    px = round(n ** (1.0 / 3.0))
    q = [px // 4, (3 * px) // 4]
    probes = [i + px * (j + px * k) for k in q for j in q for i in q]
    idx = np.array(probes + [n + v for v in probes])
    scale = float(np.abs(xp).max())
    dmax = float(np.abs(xg - xp).max()) / scale
    ng, npn = float(np.linalg.norm(xg)), float(np.linalg.norm(xp))
    res_rel = abs(info.resnorm - keep["final_residual"]) / max(abs(keep["final_residual"]), 1e-300)
    ok = (int(info.iterations) == keep["sweeps"] and int(info.inner_iterations) == keep["inner"] and res_rel <= 1e-3
          and dmax <= 1e-9 and abs(ng - npn) <= 1e-9 * npn)
    return {"ok": bool(ok), "cells": int(keep["cells"]),
            "sweeps": [int(info.iterations), keep["sweeps"]], "inner_cg_iterations": [int(info.inner_iterations), keep["inner"]],
            "final_residual": [float(info.resnorm), keep["final_residual"]], "final_residual_rel_diff": res_rel,
            "solution_norm2": [ng, npn], "solution_max_abs_diff_over_max_abs": dmax,
            "probe_nodes": [int(v) for v in idx], "probe_gpu": [float(v) for v in xg[idx]],
            "probe_port": [float(v) for v in xp[idx]],
            "tolerances": "sweeps / iterations equal, residual 1e-3 rel, solution 1e-9 of max |x| (all entries), norm 1e-9 rel; "
                          "each pair is [gpu, port]"}


def api_wall(N, k1, k2, beta, mu):
    """What a caller of the public API pays: solve_dpp_nonlinear(W, params, bcs, PICARD_MG_INEXACT_SOLVER_PARAMS) on the
    N^3 Q1 cube through the reference's objects (mesh, spaces, DirichletBC with the manufactured pressures), protocol of
    reference src/perphil/experiments/petsc_profiling_3d.py:73-86: one warm-up call, then one timed call (wall clock,
    boundary-data application, assembly, solve and the copy of the solution to the host included)."""
    import perphil_amd as pa
    from perphil_amd import fd, solver_parameters as spar
    from perphil_amd.manufactured_solutions import exact_expressions_3d

    t0 = time.perf_counter()
    mesh = fd.UnitCubeMesh(N, N, N, hexahedral=True)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    params = pa.DPPParameters(k1=k1, k2=k2, beta=beta, mu=mu)
    _u1, p1e, _u2, p2e = exact_expressions_3d(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    sol = pa.solve_dpp_nonlinear(W, params, bcs, solver_parameters=spar.PICARD_MG_INEXACT_SOLVER_PARAMS)
    first = 1e3 * (time.perf_counter() - t0)
    t1 = time.perf_counter()
    sol = pa.solve_dpp_nonlinear(W, params, bcs, solver_parameters=spar.PICARD_MG_INEXACT_SOLVER_PARAMS)
    wall = 1e3 * (time.perf_counter() - t1)
    out = {"api_wall_ms": round(wall, 2), "api_first_call_ms": round(first, 2), "sweeps": int(sol.iteration_number),
           "inner_cg_iterations": int(sol.info["inner_iterations"]),
           "what": "second solve_dpp_nonlinear(W, params, bcs, PICARD_MG_INEXACT_SOLVER_PARAMS) call on the same objects: "
                   "Dirichlet data, assembly, Picard solve, solution copied to the host; first call = objects, mesh, "
                   "allocations and that solve"}
    mesh.context().close()
    return out


def bench_configs(_ffi, device, with_cpu, cores):
    """Driver-timed numbers for the BASELINE.json configurations that fit one GPU besides the headline (VERDICT r3 item
    6): config 1 (2D 16^2 Q1, LINEAR_SOLVER_PARAMS, through the public API - plumbing), 2 (64^3 Q1 monolithic CSR CG + 2x2
    block Jacobi), 3 (128^3 Q1 Picard-split, the headline's algorithm), 5 (256^3 P1 Kuhn tets, k1/k2 = 1e4, GMRES + field
    split with multigrid-CG block solves).  Each entry: workload, ms per assemble + solve step (one warm-up step, then the
    mean of `reps`), iterations, and parity against the oracle / the C port solving the same problem (rank 0, 1 GPU)."""
    import perphil_amd as pa
    from perphil_amd import fd, solver_parameters as spar
    from perphil_amd.manufactured_solutions import exact_expressions

    out = []

    def timed(ctx, step, reps):
        step()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            info = step()
        ctx.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps, info

    # ---- config 1: plumbing case through the public API -------------------------------------------------------------
    params = pa.DPPParameters(k1=1.0, k2=1e-2, beta=1.0, mu=1.0)
    mesh = pa.create_mesh(16, 16, quadrilateral=True)
    _, V = pa.create_function_spaces(mesh)
    W = fd.MixedFunctionSpace((V, V))
    _, p1, _, p2 = exact_expressions(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1, "on_boundary"), fd.DirichletBC(W.sub(1), p2, "on_boundary")]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(2):
            pa.solve_dpp(W, params, bcs, solver_parameters=spar.LINEAR_SOLVER_PARAMS)
        # a call is ~4 ms of host-driven launches: timed call by call, median reported (right after the 16-thread CPU leg of
        # this process a single scheduling stall - the cgroup's CPU quota - otherwise lands in a mean over five calls:
        # measured 18.8 ms against 3.6 ms without that leg)
        per_call = []
        for _ in range(15):
            t0 = time.perf_counter()
            sol = pa.solve_dpp(W, params, bcs, solver_parameters=spar.LINEAR_SOLVER_PARAMS)
            per_call.append(1e3 * (time.perf_counter() - t0))
        ms = float(np.median(per_call))
    e = {"config": 1, "workload": "2D UnitSquare 16x16 Q1, monolithic linear DPP, manufactured BCs, LINEAR_SOLVER_PARAMS "
                                  "through solve_dpp (direct-equivalent solve; wall clock of a call, median of 15, result on the host)",
         "dofs": int(W.dim()), "ms": round(ms, 3), "ms_mean": round(float(np.mean(per_call)), 3), "ms_max": round(max(per_call), 3),
         "calls": len(per_call), "iterations": int(sol.iteration_number), "parity": None}
    if with_cpu:
        from oracle import dpp_oracle as o

        om = o.build_mesh(2, o.CELL_QUAD, 16, 16, 0)
        ud = o.solve_direct(o.build_system(om, o.Params(k1=1.0, k2=1e-2, beta=1.0, mu=1.0)))
        err = float(np.abs(sol.solution.vector() - ud).max() / np.abs(ud).max())
        e.update(parity=bool(err <= 1e-9), parity_vs="NumPy oracle, sparse direct solve", max_rel_diff=err)
    mesh.context().close()
    out.append(e)

    # ---- configs 2, 3, 5 through the C ABI (results stay on the device, like the headline) ---------------------------
    def cube(N, kind, k2):
        ctx = _ffi.Context(device)
        ctx.mesh_build(3, kind, N, N, N)
        m = fd.UnitCubeMesh(N, N, N, hexahedral=(kind == _ffi.CELL_HEX), comm=fd.COMM_SELF)
        b = m.boundary_nodes()
        X = m.node_coordinates(b)
        _, q1, _, q2 = pa.exact_expressions_3d(m, pa.DPPParameters(k1=1.0, k2=k2, beta=1.0, mu=1.0))
        ctx.set_dirichlet(0, b, q1(X))
        ctx.set_dirichlet(1, b, q2(X))
        return ctx

    def stepper(ctx, cfg, k2, mono):
        def step():
            ctx.set_option("invalidate_KM", 1)
            ctx.assemble(1.0, k2, 1.0, 1.0, monolithic=mono)
            _, info, _ = ctx.solve(cfg, fetch=False)
            return info
        return step

    # config 2
    ctx = cube(64, _ffi.CELL_HEX, 1e-2)
    cfg = _ffi.SolverCfg()
    cfg.ksp_type, cfg.pc_type, cfg.restart, cfg.max_it, cfg.rtol, cfg.atol = _ffi.KSP_CG, _ffi.PC_BLOCK2, 30, 50000, 1e-8, 1e-12
    cfg.inner_ksp_type, cfg.inner_pc_type, cfg.inner_max_it, cfg.inner_rtol, cfg.inner_atol = _ffi.KSP_CG, _ffi.PC_MG, 500, 1e-10, 1e-300
    cfg.picard, cfg.mg_smooth = 0, 2
    ms, info = timed(ctx, stepper(ctx, cfg, 1e-2, True), 3)
    e = {"config": 2, "workload": "3D UnitCube 64^3 Q1, monolithic linear DPP on the field-major CSR (aij), CG + 2x2 node-block "
                                  "Jacobi to ksp_rtol 1e-8, assemble + solve",
         "dofs": int(2 * ctx.n), "ms": round(ms, 3), "iterations": int(info.iterations), "converged": bool(info.converged),
         "dofs_per_s": round(2 * ctx.n / ms * 1e3, 0), "parity": None}
    if with_cpu:
        keep = {}
        _cpu_port_run(64, cores, 1.0, 1e-2, 1.0, 1.0, 2, 0.0, 1e-12, 0, keep, steps=1, rtol=1e-11)
        x = ctx.solution()
        err = float(np.abs(x - keep["x"]).max() / np.abs(keep["x"]).max())
        e.update(parity=bool(info.converged and err <= 1e-6), parity_vs="C/OpenMP port, block Picard run to 1e-11 (same fixed point)",
                 max_rel_diff=err)
    ctx.close()
    out.append(e)

    # config 3
    ctx = cube(128, _ffi.CELL_HEX, 1e-2)
    cfg = picard_cfg(_ffi, 1e-10, 1, 1e-1, 1)
    ms, info = timed(ctx, stepper(ctx, cfg, 1e-2, False), 3)
    e = {"config": 3, "workload": "3D UnitCube 128^3 Q1, Picard-split (fixed-stress) solve, the headline's algorithm (inexact sweeps, "
                                  "CG + multigrid V(1,1) block solves), assemble + solve",
         "dofs": int(2 * ctx.n), "ms": round(ms, 3), "iterations": int(info.iterations), "inner_cg_iterations": int(info.inner_iterations),
         "converged": bool(info.converged), "dofs_per_s": round(2 * ctx.n / ms * 1e3, 0), "parity": None}
    if with_cpu:
        keep = {}
        _cpu_port_run(128, cores, 1.0, 1e-2, 1.0, 1.0, 1, 1e-1, 1e-10, 1, keep, steps=1)
        pv = parity_vs_port(ctx.solution(), info, keep)
        e.update(parity=bool(pv["ok"]), parity_vs="C/OpenMP port, same algorithm: sweeps / iterations equal, solution 1e-9",
                 max_rel_diff=pv["solution_max_abs_diff_over_max_abs"], sweeps=pv["sweeps"], inner=pv["inner_cg_iterations"])
    ctx.close()
    out.append(e)

    # config 5
    ctx = cube(256, _ffi.CELL_TET, 1e-4)
    cfg = config5_cfg(_ffi)
    ms, info = timed(ctx, stepper(ctx, cfg, 1e-4, True), 2)
    e = {"config": 5, "workload": "3D UnitCube 256^3 P1 Kuhn tetrahedra (100.7 M cells), k1/k2 = 1e4, GMRES(30) + multiplicative "
                                  "field-split to ksp_rtol 1e-8, block solves CG + multigrid V(1,1) to 1e-10, assemble + solve",
         "dofs": int(2 * ctx.n), "ms": round(ms, 3), "iterations": int(info.iterations), "inner_cg_iterations": int(info.inner_iterations),
         "converged": bool(info.converged), "dofs_per_s": round(2 * ctx.n / ms * 1e3, 0), "parity": None}
    if with_cpu:
        keep = {}
        tp = _cpu_port_run(256, cores, 1.0, 1e-4, 1.0, 1.0, 1, 1e-1, 1e-10, 1, keep, kind=3, steps=1, rtol=1e-10)[0]
        x = ctx.solution()
        err = float(np.abs(x - keep["x"]).max() / np.abs(keep["x"]).max())
        e.update(parity=bool(info.converged and int(info.iterations) == 4 and err <= 1e-6),
                 parity_vs="C/OpenMP port on the same tetrahedral problem, block Picard to 1e-10 (same fixed point); 4 outer "
                           "iterations = the reference's count at every size (petsc_perf_breakdown_3d.csv, Scale-Splitting GMRES)",
                 max_rel_diff=err, port_seconds=round(tp, 2))
    ctx.close()
    out.append(e)
    return out


def config5_cfg(_ffi):
    """BASELINE config 5: GMRES(30) + multiplicative field-split (reference src/perphil/solvers/parameters.py:30-37 with
    GMRES_PARAMS, experiments/petsc_profiling_3d.py:31 for the tetrahedra); block LU -> CG + multigrid to 1e-10."""
    cfg = _ffi.SolverCfg()
    cfg.ksp_type, cfg.pc_type, cfg.restart, cfg.max_it, cfg.rtol, cfg.atol = _ffi.KSP_GMRES, _ffi.PC_FIELDSPLIT, 30, 200, 1e-8, 1e-12
    cfg.inner_ksp_type, cfg.inner_pc_type, cfg.inner_max_it, cfg.inner_rtol, cfg.inner_atol = _ffi.KSP_CG, _ffi.PC_MG, 500, 1e-10, 1e-300
    cfg.picard, cfg.picard_rtol, cfg.picard_atol, cfg.picard_max_it, cfg.mg_smooth = 0, 1e-8, 1e-12, 100, 1
    cfg.inner_reduction, cfg.inner_norm = 0.0, 0
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=256, help="cells per direction of the unit cube")
    ap.add_argument("--cpu-sample-n", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api-wall", action="store_true", help="omit the public-API wall-clock measurement (config.api)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: cgroup share, at most 16)")
    ap.add_argument("--inner-rtol", type=float, default=1e-10)
    ap.add_argument("--smooth", type=int, default=1)
    ap.add_argument("--inner-reduction", type=float, default=1e-1)
    ap.add_argument("--inner-norm", type=int, default=1, choices=(0, 1, 2),
                    help="norm tested by the block solves: 0 preconditioned, 1 unpreconditioned, 2 none (exactly "
                         "--inner-max-it iterations per block solve: launch-only sweeps)")
    ap.add_argument("--inner-max-it", type=int, default=50000, help="iteration limit of a block solve (A/B experiments)")
    ap.add_argument("--asm-kernel", type=int, default=2)
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="extra pph_set_option settings for A/B runs (e.g. sell_dict=0)")
    ap.add_argument("--skip-fine-bench", action="store_true", help="omit the isolated fine-level SpMV loop (PMC passes)")
    ap.add_argument("--skip-csr", action="store_true", help="omit the extra untimed step on the CSR operator format")
    ap.add_argument("--halo-overlap", type=int, default=1, choices=(0, 1, 2),
                    help="N > 1: products of large levels as interior + boundary launches with the halo exchange of the "
                         "operand on a second stream behind the interior rows (1, default: bit-identical to 2 in the gloo "
                         "tests), the same launches serially (2), or exchange, then one launch (0)")
    ap.add_argument("--strict", action="store_true",
                    help="N > 1: exit non-zero when the library's RCCL transport fails to start (default: continue on the "
                         "torch.distributed callbacks - still RCCL on the device with the nccl backend - and say so in the "
                         "line: config.transport 'torch-nccl', config.rccl_native_error)")
    ap.add_argument("--allow-fallback", action="store_true", help="(the default since round 4; kept for old command lines)")
    ap.add_argument("--config", type=int, default=4, choices=(4, 5),
                    help="BASELINE.json configuration: 4 (default; = 3 at --cells 128) Q1 hexahedra, Picard-split - the headline "
                         "metric; 5: P1 Kuhn tetrahedra, k1/k2 = 1e4, GMRES + field-split - on --gpus N slabs either way")
    ap.add_argument("--no-configs", action="store_true",
                    help="omit the timed runs of the other single-GPU BASELINE configurations (1, 2, 3, 5; 'configs' in the line)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started like the 1-GPU run (`python bench.py --gpus N`): launch the ranks as a CHILD process - nothing in this
        # process has imported torch or touched a GPU yet - and relay rank 0's JSON line and the exit code
        raise SystemExit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch  # device selection, barrier, synchronize: plumbing only
    from perphil_amd import _ffi

    dist = None
    rehearsal = None
    device = local_rank
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        # "nccl" is RCCL on ROCm.  PERPHIL_DIST_BACKEND=gloo rehearses the multi-rank path on a box with
        # fewer GPUs than ranks (ranks then share devices; communication is staged through the host).
        backend = os.environ.get("PERPHIL_DIST_BACKEND", "nccl")
        if backend == "nccl" and torch.cuda.device_count() < world:
            # fewer GPUs than ranks: RCCL refuses two ranks on one device ("Duplicate GPU detected") - rehearse over gloo
            # instead of failing without a line; config.transport says "torch-gloo", config.rehearsal why
            backend = "gloo"
            rehearsal = f"{world} ranks on {torch.cuda.device_count()} GPU(s): gloo transport, ranks share devices - timings are not a scaling measurement"
        device = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
    N = args.cells
    c5 = args.config == 5
    k1, k2, beta, mu = 1.0, (1e-4 if c5 else 1e-2), 1.0, 1.0
    kind = _ffi.CELL_TET if c5 else _ffi.CELL_HEX
    rccl_error = None

    transport, ranks_seen = "none", 1
    # HIP runtime start-up (the first HIP call of the process) + context: a per-process cost, reported apart from the
    # problem's set-up
    t_ctx0 = time.perf_counter()
    _warm = _ffi.Context(device)
    _warm.synchronize()
    context_ms = 1e3 * (time.perf_counter() - t_ctx0)
    _warm.close()
    t_setup0 = time.perf_counter()
    if world > 1:
        from perphil_amd.distributed import SlabSolver

        # the reference's objects under the process group: fd.UnitCubeMesh = this rank's slab + its transport.  A failing
        # RCCL start-up continues on the torch.distributed callbacks and is labelled (--strict: exit non-zero instead)
        solver = SlabSolver(N, world, rank, device, k1, k2, beta, mu, inner_rtol=args.inner_rtol, smooth=args.smooth,
                            inner_reduction=args.inner_reduction, inner_norm=args.inner_norm, kind=kind,
                            strict=True if args.strict else None)
        transport, ranks_seen = solver.transport_label, solver.ranks_seen
        rccl_error = solver.transport_info.rccl_error
        if c5:
            solver.cfg = config5_cfg(_ffi)
            solver.monolithic = True
        solver.ctx.set_option("asm_kernel", args.asm_kernel)
        solver.ctx.set_option("halo_overlap", args.halo_overlap)
        for kv in args.set:
            solver.ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        dofs_global = solver.global_dofs
        step = solver.step
        ctx = solver.ctx
    else:
        ctx = _ffi.Context(device)
        ctx.mesh_build(3, kind, N, N, N)
        b, g1, g2 = mms_boundary(N, k1, k2, beta, mu)
        ctx.set_dirichlet(0, b, g1)
        ctx.set_dirichlet(1, b, g2)
        cfg = (config5_cfg(_ffi) if c5 else
               picard_cfg(_ffi, args.inner_rtol, args.smooth, args.inner_reduction, args.inner_norm, args.inner_max_it))
        ctx.set_option("asm_kernel", args.asm_kernel)
        for kv in args.set:
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        dofs_global = 2 * ctx.n
        last = {}

        def step():
            ctx.set_option("invalidate_KM", 1)          # integrate K and M again: assembly is part of the step
            ctx.assemble(k1, k2, beta, mu, monolithic=c5)
            _, info, _ = ctx.solve(cfg, fetch=False)
            last["info"] = info
            return info

    def fence():
        torch.cuda.synchronize()
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    ctx.synchronize()
    setup_ms = 1e3 * (time.perf_counter() - t_setup0)   # mesh + pattern + boundary data (host-side MMS evaluation included)
    cold_ms = None
    for w in range(args.warmup):
        tc = time.perf_counter()
        step()
        if w == 0:
            ctx.synchronize()
            cold_ms = 1e3 * (time.perf_counter() - tc)   # first step: allocates operators + multigrid hierarchy
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
    # the timed steps' solution, for the parity check against the port below (272 MB at 256^3; outside the timed region)
    x_gpu = ctx.solution() if (world == 1 and not args.no_cpu_baseline) else None
    info_timed = info

    # ---- roofline of the dominant kernel: one more (untimed) step with an event pair around every
    # SpMV launch on the solver's stream; bytes are algorithmic (stencil-ELL: 8 S nrows + 16 nrows per
    # launch; CSR: 12 nnz + 20 nrows) ------------------------------------------------------------------
    def instrumented_step():
        ctx.set_option("time_spmv", 1)
        step()
        ctx.synchronize()
        t = ctx.timers()
        ctx.set_option("time_spmv", 0)
        launches = t["spmv_launches"] + t["spmv_dot_launches"]
        ms = t["spmv_ms"] + t["spmv_dot_ms"]
        byts = t["spmv_bytes"] + t["spmv_dot_bytes"]
        return t, launches, ms, byts

    def fine_block(fmt_bytes, reps=200):
        if args.skip_fine_bench:
            return None
        ms1 = ctx.spmv_bench(_ffi.MAT_A11, reps)
        return {"avg_launch_ms": round(ms1, 4), "algorithmic_bytes": fmt_bytes,
                "achieved": round(fmt_bytes / 1e9 / (ms1 / 1e3), 1),
                "frac": round(fmt_bytes / 1e9 / (ms1 / 1e3) / HBM_PEAK_GBS, 4)}

    def in_solver(t):
        if not (t["spmv_fine_launches"] and t["spmv_fine_ms"] > 0):
            return None
        fa = t["spmv_fine_bytes"] / 1e9 / (t["spmv_fine_ms"] / 1e3)
        return {"launches_per_step": t["spmv_fine_launches"],
                "avg_launch_ms": round(t["spmv_fine_ms"] / t["spmv_fine_launches"], 4),
                "achieved": round(fa, 1), "frac": round(fa / HBM_PEAK_GBS, 4)}

    sell = not any(kv.split("=")[0] == "op_format" and float(kv.split("=")[1]) == 0 for kv in args.set)
    tr, launches, ms, byts = instrumented_step()
    achieved = (byts / 1e9) / (ms / 1e3) if ms > 0 else 0.0
    sym = bool(ctx.timers().get("symmetric_storage", False))   # what the last assembly actually stored
    SF = 15 if c5 else 27        # stencil: 27-point (Q1 hexahedra) / 15-point (P1 Kuhn tetrahedra)
    S = SF // 2 + 1 if sym else SF   # stored slots per row: diagonal + upper half of the stencil, or all of it
    extra_records = not args.skip_csr and not c5   # the stored-value / full-storage / CSR records belong to the headline config
    # row dictionaries (option sell_dict, default): the products of the fine blocks stream a 2-byte class per row
    # instead of the S stored values (pph_get_timers: operators on a dictionary, distinct rows of A11)
    dict_ops, dict_classes = int(tr.get("dict_operators", 0)), int(tr.get("dict_classes", 0))
    if dist is not None:
        # every rank must take the same extra (collective) steps below: the dictionaries count only if every slab has them
        t = torch.tensor([dict_ops], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        dict_ops = int(t.item())
    dicton = sell and dict_ops >= 3
    # (classes constant along z - every box with one Dirichlet set per face: the class words of four planes only are read)
    zconst = dicton and bool(tr.get("dict_zconst", False))
    cls_bytes = 2.0 * 4.0 / (N + 1) if zconst else 2.0
    sell_bytes = (cls_bytes if dicton else 8.0 * S) * ctx.n + 16.0 * ctx.n
    csr_bytes = 12.0 * ctx.nnzb + 20.0 * ctx.n
    # HBM traffic of the same kernel mix from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KB), taken
    # offline with tools/pmc_summarize.py and committed under profiles/ (PMC cannot be sampled in-process).  The
    # file is stamped with the kernel and the launch count it was taken on: anything else reports null.
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "r04_pmc_spmv_dict_bench256.json" if dicton else "r03_pmc_spmv_bench256.json")
    if os.path.exists(pmc_file) and N == 256 and world == 1 and sell and not c5:
        try:
            with open(pmc_file) as f:
                pmc = json.load(f)
            if (int(pmc.get("launches_per_step", -1)) == int(launches) and pmc.get("kernel") in ("k_spmv_sell", "k_spmv")
                    and bool(pmc.get("symmetric", False)) == bool(sym) and bool(pmc.get("dictionary", False)) == dicton):
                traffic = pmc.get("traffic_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "kernel": ((f"k_spmv_sell<tet,mode,2,sym,DICT> (15-point stencil-ELL SpMV on row dictionaries: {dict_classes} distinct rows of "
                    "A11 in LDS, a 2-byte class per row instead of the 8 stored values; field-split block solves; the monolithic "
                    "GMRES products run k_spmv_wide on the field-major CSR and are not in this figure)") if (dicton and c5) else
                   (f"k_spmv_dict_walk<mode> (stencil-ELL SpMV on row dictionaries: {dict_classes} distinct rows of A11 in LDS, a "
                    "2-byte class per row instead of the 14 stored values" + (" - read on four planes only: the classes of an in-plane "
                    "position are constant on the interior planes" if zconst else "") + ", x window of three planes in registers; operators "
                    "below 1 M rows - the coarse multigrid levels - k_spmv_sell on their stored values; all multigrid "
                    "levels of one step)")
                   if dicton else
                   (f"k_spmv_sell<kind,mode,2,{'sym' if sym else 'full'}> (stencil-ELL SpMV, "
                    + ("symmetric storage: 14 of 27 slot arrays, every value serves two rows, " if sym else "")
                    + "8 B per stored entry, all multigrid levels of one step)")
                   if sell else "k_spmv_wide<8,*,2> (aligned-wide CSR-vector SpMV, all multigrid levels of one step)"),
        "format": (("row dictionary over " if dicton else "") + ("stencil-ELL (symmetric)" if sym else "stencil-ELL")) if sell else "CSR",
        "dictionary": {"operators": dict_ops, "classes_A11": dict_classes, "classes_constant_along_z": zconst} if sell else None,
        "launches_per_step": int(launches), "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
        "algorithmic_bytes_per_launch": round(byts / max(launches, 1), 0),
        "fine_level": fine_block(sell_bytes if sell else csr_bytes),
        "fine_level_in_solver": in_solver(tr),
    }
    if dicton and extra_records:
        # the same step with the products on the STORED values (symmetric stencil-ELL, the dominant kernel of rounds 2-3
        # and the path of every mesh whose rows do not repeat): untimed, for the record
        ctx.set_option("sell_dict", 0)
        step()
        tv, lv, msv, bv = instrumented_step()
        t0v = time.perf_counter()
        step()
        ctx.synchronize()
        stored_ms = 1e3 * (time.perf_counter() - t0v)
        av = (bv / 1e9) / (msv / 1e3) if msv > 0 else 0.0
        stored_bytes = 8.0 * S * ctx.n + 16.0 * ctx.n
        roofline["stored_values"] = {"kernel": "k_spmv_sell<kind,mode,2,sym> (14 stored slot arrays, every value serves two rows)",
                                     "achieved": round(av, 1), "frac": round(av / HBM_PEAK_GBS, 4), "launches_per_step": int(lv),
                                     "algorithmic_bytes_per_launch": round(bv / max(lv, 1), 0), "ms_per_step": round(stored_ms, 3),
                                     "fine_level": fine_block(stored_bytes), "fine_level_in_solver": in_solver(tv)}
        ctx.set_option("sell_dict", 1)
        step()                                   # (the dictionaries are rebuilt by the next assembly)
    if sell and sym and extra_records:
        # the same step on full (27-slot) stencil-ELL storage: untimed, for the record
        ctx.set_option("sell_sym", 0)
        step()
        tf, lf, msf, bf = instrumented_step()
        t0f = time.perf_counter()
        step()
        ctx.synchronize()
        full_ms = 1e3 * (time.perf_counter() - t0f)
        af = (bf / 1e9) / (msf / 1e3) if msf > 0 else 0.0
        full_bytes = 8.0 * SF * ctx.n + 16.0 * ctx.n
        roofline["full_storage"] = {"kernel": "k_spmv_sell<kind,mode,2,full> (all 27 slot arrays)", "achieved": round(af, 1),
                                    "frac": round(af / HBM_PEAK_GBS, 4), "launches_per_step": int(lf),
                                    "algorithmic_bytes_per_launch": round(bf / max(lf, 1), 0), "ms_per_step": round(full_ms, 3),
                                    "fine_level": fine_block(full_bytes), "fine_level_in_solver": in_solver(tf)}
        ctx.set_option("sell_sym", 1)
    if sell and extra_records:
        # the same step on the CSR operator format (the north-star's "CSR SpMV inner loop"): untimed, for the record
        ctx.set_option("op_format", 0)
        step()                                   # re-assembles into CSR arrays, rebuilds the level operators
        tc, lc, msc, bc = instrumented_step()
        t0c = time.perf_counter()
        step()
        ctx.synchronize()
        csr_step_ms = 1e3 * (time.perf_counter() - t0c)
        ac = (bc / 1e9) / (msc / 1e3) if msc > 0 else 0.0
        roofline["csr"] = {"kernel": "k_spmv_wide<8,*,2> (aligned-wide CSR-vector SpMV)", "achieved": round(ac, 1),
                           "frac": round(ac / HBM_PEAK_GBS, 4), "launches_per_step": int(lc),
                           "algorithmic_bytes_per_launch": round(bc / max(lc, 1), 0), "ms_per_step": round(csr_step_ms, 3),
                           "fine_level": fine_block(csr_bytes), "fine_level_in_solver": in_solver(tc)}
        ctx.set_option("op_format", 1)
    cs = ctx.comm_stats()
    # communication split of one more (untimed) step: every halo exchange / all-reduce bracketed by an event pair on the
    # stream it runs on (RCCL transport; callback transports: host clock inside the callback), so that a measured scaling
    # curve can be read: comm_ms (exchange, all-reduce) beside the step's wall time taken the same way
    comm = None
    if world > 1:
        ctx.set_option("time_comm", 1)
        fence()
        tq = time.perf_counter()
        step()
        ctx.synchronize()
        step_ms_timed = 1e3 * (time.perf_counter() - tq)
        ct = ctx.comm_times()
        ctx.set_option("time_comm", 0)
        vals = [ct["halo_ms"], ct["allreduce_ms"], step_ms_timed]
        tv = torch.tensor(vals, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        halo_ms, ar_ms, step_ms_timed = (float(v) for v in tv.tolist())
        overlapped = args.halo_overlap == 1 and transport == "rccl"
        comm = {"halo_ms": round(halo_ms, 3), "allreduce_ms": round(ar_ms, 3),
                "halo_exchanges_timed": ct["halo_timed"], "allreduces_timed": ct["allreduce_timed"],
                "step_ms_with_comm_timers": round(step_ms_timed, 3),
                "kernel_ms": round(step_ms_timed - ar_ms - (0.0 if overlapped else halo_ms), 3),
                "what": "max over ranks, one extra untimed step; rccl transport: device time between HIP events around each "
                        "grouped ncclSend/ncclRecv and each ncclAllReduce (an exchange overlapped with interior rows counts "
                        "its full duration on the communication stream and is NOT subtracted in kernel_ms); callback "
                        "transports: host time inside the callback; kernel_ms = the step's wall time minus the serial "
                        "communication = kernels + launch gaps + host round trips"}

    out = {
        "metric": ("DoF/s (assemble+solve), 3D UnitCube P1 (Kuhn tetrahedra) DPP, k1/k2 = 1e4, GMRES + field-split" if c5 else
                   "DoF/s (assemble+solve), 3D UnitCube Q1 DPP, Picard-split"),
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
            "workload": (f"BASELINE config 5: 3D UnitCube {N}^3 P1 Kuhn tetrahedra ({6 * N ** 3} cells), two-pressure DPP, k1/k2 = 1e4 "
                         f"(k1=1, k2=1e-4, beta=mu=1), manufactured Dirichlet data, assemble + GMRES(30) with the multiplicative "
                         f"field-split preconditioner to ksp_rtol 1e-8, block solves = CG + geometric multigrid V(1,1) to 1e-10 "
                         f"on stencil-ELL scalar blocks, monolithic products on the field-major CSR") if c5 else
                        f"3D UnitCube {N}^3 Q1, two-pressure DPP (k1=1, k2=1e-2, beta=mu=1), manufactured Dirichlet data, "
                        f"assemble + block Picard (fixed-stress) to snes_rtol 1e-8 (true residual), warm-started block solves = "
                        f"CG + geometric multigrid (Chebyshev-Jacobi V({args.smooth},{args.smooth})) on {'stencil-ELL' if sell else 'CSR'} "
                        f"scalar blocks, each "
                        + (f"of exactly {args.inner_max_it} CG iteration(s), no inner convergence test (launch-only sweeps)"
                           if args.inner_norm == 2 else
                           f"to a reduction of the {'unpreconditioned' if args.inner_norm else 'preconditioned'} residual by "
                           f"{args.inner_reduction:g} (or rtol {args.inner_rtol:g})"),
            "baseline_config": args.config if (c5 or N == 256) else (3 if N == 128 else None),
            "preallocation": "outside the timed step: HIP runtime start-up + first context of the process (context_ms), mesh "
                             "+ host-side boundary data + its upload (setup_ms; the CSR pattern is built only on demand) and, "
                             "inside the first (cold) step (cold_step_ms), the buffer / multigrid-hierarchy allocation and the "
                             "FIRST BUILD of the row dictionaries (hash build + table + first bitwise check + read-back of the "
                             "verdict per operator: dict_build_ms for dict_builds operators - a symbolic phase, paid again only "
                             "when the mesh or a Dirichlet SET changes; the per-step check of every row stays in the timed step)",
            "dict_build_ms": round(float(tm.get("dict_build_ms", 0.0)), 3), "dict_builds": int(tm.get("dict_builds", 0)),
            "setup_ms": round(setup_ms, 2), "cold_step_ms": None if cold_ms is None else round(cold_ms, 2),
            "context_ms": round(context_ms, 2),
            "operator_format": ((("row dictionary (%d distinct rows) over " % dict_classes) if dicton else "")
                                + ("stencil-ELL, symmetric storage" if sym else "stencil-ELL")) if sell else "CSR",
            "transport": transport, "ranks_seen": int(ranks_seen), "rccl_native_error": rccl_error, "rehearsal": rehearsal,
            "comm": comm,
            "allreduces_per_step": int(cs["allreduces"]),
            "halo_overlap": int(args.halo_overlap), "split_products_per_step": int(tm.get("split_products", 0)),
            "cells": (6 if c5 else 1) * N ** 3, "dofs": int(dofs_global), "parallelism": f"slab{world}" if world > 1 else "single",
            ("gmres_iterations" if c5 else "picard_sweeps"): int(info.iterations), "inner_cg_iterations": int(info.inner_iterations),
            ("ms_per_outer_iteration" if c5 else "picard_ms_per_sweep"): round(tm["solve_ms"] / max(int(info.iterations), 1), 3),
            "assemble_ms": round(tm["assemble_ms"] + tm["bc_blocks_ms"], 3), "solve_ms": round(tm["solve_ms"], 3),
            "final_residual": float(info.resnorm), "rhs_norm": float(info.rhs_norm),
            "halo_exchanges_per_step": int(cs["halo_exchanges"]),
        },
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_api_wall and not c5:
        out["config"]["api"] = api_wall(N, k1, k2, beta, mu)
    parity_failed = False
    if c5:
        if rank == 0:
            out["cpu_baseline"] = None
            out["parity_vs_port"] = None
    elif rank == 0 and not args.no_cpu_baseline and world == 1:
        keep = {}
        out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_n, N), k1, k2, beta, mu, args.smooth, args.inner_reduction,
                                           args.inner_rtol, args.cpu_threads, args.inner_norm, keep)
        if args.inner_norm == 2:
            out["parity_vs_port"] = None      # (the port's leg runs the reduction-based sweeps)
        else:
            if keep["cells"] != N:
                # the port's bounded sample is a smaller cube than the timed one: one GPU step at the port's size
                ctx.mesh_build(3, _ffi.CELL_HEX, keep["cells"], keep["cells"], keep["cells"])
                b, g1, g2 = mms_boundary(keep["cells"], k1, k2, beta, mu)
                ctx.set_dirichlet(0, b, g1)
                ctx.set_dirichlet(1, b, g2)
                info_timed = step()
                x_gpu = ctx.solution()
            pv = parity_vs_port(x_gpu, info_timed, keep)
            out["parity_vs_port"] = pv["ok"]
            out["parity"] = pv
            parity_failed = parity_failed or not pv["ok"]
    elif rank == 0:
        out["cpu_baseline"] = None
        out["parity_vs_port"] = None
    if rank == 0 and world == 1 and not args.no_configs and not c5 and N == 256:
        # the other BASELINE configurations that fit one GPU, each timed and checked (after the headline: own contexts)
        ctx.close()
        cfgs = bench_configs(_ffi, device, not args.no_cpu_baseline, args.cpu_threads if args.cpu_threads > 0 else host_cores())
        out["configs"] = cfgs
        parity_failed = parity_failed or any(c.get("parity") is False for c in cfgs)
    if dist is not None:
        dist.barrier()
        ctx.close()  # destroys the library's RCCL communicator before the process group goes away
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    if parity_failed:
        raise SystemExit("bench.py: the GPU step and the CPU port / oracle disagree (see \"parity\" / \"configs\" in the line above)")


if __name__ == "__main__":
    main()
