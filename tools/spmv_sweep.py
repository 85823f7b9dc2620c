#!/usr/bin/env python3
"""SpMV micro-benchmark: every kernel variant / lanes-per-row on the scalar block and the monolithic
CSR of the N^3 Q1 unit cube; prints ms per launch and GB/s of algorithmic bytes (12 nnz + 20 nrows)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from perphil_amd import _ffi  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, nargs="+", default=[128, 256])
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--mono", action="store_true")
args = ap.parse_args()
for N in args.cells:
    ctx = _ffi.Context(0)
    ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
    b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
    ctx.set_dirichlet(0, b, g1)
    ctx.set_dirichlet(1, b, g2)
    ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=args.mono)
    mats = [("A11", _ffi.MAT_A11, ctx.nnzb, ctx.n)]
    if args.mono:
        mats.append(("mono", _ffi.MAT_MONO, 4 * ctx.nnzb, 2 * ctx.n))
    for name, which, nnz, nrows in mats:
        byts = 12.0 * nnz + 20.0 * nrows
        for kern in (0, 1, 2, 3):
            ctx.set_option("spmv_kernel", kern)
            for lanes in ((4, 8, 16, 32) if kern != 2 else (0,)):
                ctx.set_option("spmv_lanes", lanes)
                ms = ctx.spmv_bench(which, args.reps)
                print(f"N={N} {name:5s} kernel={kern} lanes={lanes:2d}  {ms:8.4f} ms  {byts / 1e9 / (ms / 1e3):8.1f} GB/s  "
                      f"({byts / 1e9 / (ms / 1e3) / 8000 * 100:5.1f}% of 8 TB/s)", flush=True)
    ctx.close()
