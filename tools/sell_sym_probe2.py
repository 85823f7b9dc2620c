"""Symmetric stencil-ELL SpMV: resident-footprint experiment (few workgroups, long z-walks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from perphil_amd import _ffi
import bench
N = 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
for rpt in (2, 1):
    ctx.set_option("sell_rpt", rpt)
    for blocks in (256, 512, 1024, 2048, 4096, 8192):
        ctx.set_option("sell_blocks", blocks)
        for z in (4, 8, 16, 32, 64):
            ctx.set_option("sell_zwalk", z)
            ms = ctx.spmv_bench(_ffi.MAT_A11, 30)
            print(f"rpt {rpt} blocks {blocks} zwalk {z}: {ms:.4f} ms", flush=True)
