"""Symmetric stencil-ELL SpMV: z-walk length / balanced column walk, workgroups per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
ctx.set_option("sell_zwalk_min_chunks", 1)
for xmap in (0, 1):
    ctx.set_option("sell_xmap", xmap)
    for blocks in (256, 512):
        ctx.set_option("sell_blocks", blocks)
        for z in (2, 4, 8, 16, 1000):
            ctx.set_option("sell_zwalk", z)
            ms = min(ctx.spmv_bench(_ffi.MAT_A11, 30) for _ in range(3))
            print(f"N {N} xmap {xmap} blocks {blocks} zwalk {z}: {ms:.4f} ms", flush=True)
