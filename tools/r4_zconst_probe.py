"""Round 4: dictionary product with / without class loads inside the interior planes (sell_dict_zconst), 256^3 block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
for zc in (1, 0, 1, 0):
    ctx.set_option("sell_dict_zconst", zc)
    print("zconst", zc, "walk ms", round(ctx.spmv_bench(_ffi.MAT_A11, 200), 4), flush=True)
