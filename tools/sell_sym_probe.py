"""Stencil-ELL SpMV: symmetric storage and the z-walk chunk order (fine-level 256^3 block)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
x = np.random.default_rng(1).uniform(-1, 1, ctx.n)
for sym in (1, 0):
    ctx.set_option("sell_sym", sym)
    ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
    ctx.set_option("sell_zwalk", 0)
    y0 = ctx.spmv(_ffi.MAT_A11, x)
    for blocks in (1024, 2048, 4096):
        ctx.set_option("sell_blocks", blocks)
        for z in (0, 2, 4, 8, 16, 32):
            ctx.set_option("sell_zwalk", z)
            y1 = ctx.spmv(_ffi.MAT_A11, x)
            ms = ctx.spmv_bench(_ffi.MAT_A11, 50)
            print(f"sym {sym} blocks {blocks} zwalk {z}: {ms:.4f} ms, max diff {np.abs(y1 - y0).max():.2e}", flush=True)
