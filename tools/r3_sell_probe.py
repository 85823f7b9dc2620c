"""Round 3: symmetric stencil-ELL product on the 256^3 block - rows per thread, grid, walk length, non-temporal hints.
usage: r3_sell_probe.py [reps]   (each configuration = 20 warm-up + reps launches; the order is printed, so a PMC pass
over the same script can be split by dispatch order)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
ctx.set_option("sell_zwalk_min_chunks", 1)
base = {"sell_rpt": 2, "sell_blocks": 256, "sell_zwalk": 4, "sell_xmap": 1, "sell_flags": 0}
cfgs = [{}, {"sell_rpt": 1}, {"sell_rpt": 1, "sell_blocks": 512}, {"sell_rpt": 1, "sell_zwalk": 8}, {"sell_rpt": 1, "sell_zwalk": 16},
        {"sell_zwalk": 8}, {"sell_flags": 1}, {"sell_flags": 3}, {"sell_blocks": 128}, {"sell_rpt": 1, "sell_blocks": 128},
        {"sell_blocks": 192}, {}]
for i, c in enumerate(cfgs):
    o = dict(base); o.update(c)
    for k, v in o.items():
        ctx.set_option(k, v)
    ms = min(ctx.spmv_bench(_ffi.MAT_A11, reps) for _ in range(1 if reps < 10 else 3))
    print(f"cfg {i} {c}: {ms:.4f} ms", flush=True)
