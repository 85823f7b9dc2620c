"""Stencil-ELL SpMV against the CSR kernel on the fine-level scalar block: correctness (same product) and time
for rows-per-thread / grid / XCD-group choices.  usage: sell_probe.py [N] [hex|tet]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = _ffi.CELL_TET if (len(sys.argv) > 2 and sys.argv[2] == "tet") else _ffi.CELL_HEX
ctx = _ffi.Context(0); ctx.mesh_build(3, kind, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
S = 27 if kind == _ffi.CELL_HEX else 15
csr_bytes = 12.0 * ctx.nnzb + 20.0 * ctx.n
sell_bytes = 8.0 * S * ctx.n + 16.0 * ctx.n
x = np.random.default_rng(1).uniform(-1, 1, ctx.n)
ctx.set_option("op_format", 0)
y0 = ctx.spmv(_ffi.MAT_A11, x)
ms = ctx.spmv_bench(_ffi.MAT_A11, 50)
print(f"N={N} CSR default: {ms:.4f} ms  {csr_bytes/1e9/(ms/1e3):.0f} GB/s of its {csr_bytes/1e9:.2f} GB", flush=True)
ctx.set_option("op_format", 1)
for rpt in (1, 2):
    ctx.set_option("sell_rpt", rpt)
    y1 = ctx.spmv(_ffi.MAT_A11, x)
    print(f"rpt {rpt}: max |sell - csr| = {np.abs(y1 - y0).max():.3e} (|y| max {np.abs(y0).max():.3e})", flush=True)
    for blocks in (512, 1024, 2048, 4096):
        ctx.set_option("sell_blocks", blocks)
        for group in (1, 2, 8, 32, 129, 1032):
            ctx.set_option("sell_group", group)
            ms = ctx.spmv_bench(_ffi.MAT_A11, 50)
            print(f"  sell rpt={rpt} blocks={blocks} group={group}: {ms:.4f} ms  {sell_bytes/1e9/(ms/1e3):.0f} GB/s of its {sell_bytes/1e9:.2f} GB", flush=True)
