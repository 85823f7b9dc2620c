"""Round 3: a few launches of the cached and the LDS hand-over product on the 256^3 block, for a PMC pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
for lds in (0, 1):
    ctx.set_option("sell_lds", lds)
    print(lds, ctx.spmv_bench(_ffi.MAT_A11, 3), flush=True)
