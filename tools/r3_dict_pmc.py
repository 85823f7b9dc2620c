"""Workload for counter passes over the dictionary products: 12 launches each of the walk kernel and of k_spmv_sell<DICT>
(256^3 block, y = A x), then 12 of the stored-value kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
ctx.set_option("spmv_bench_mode", 2)   # timed per launch; 5 warm-up + reps launches
print("walk ms", ctx.spmv_bench(_ffi.MAT_A11, 7))
ctx.set_option("sell_dict_walk", 0)
print("chunk-order dict ms", ctx.spmv_bench(_ffi.MAT_A11, 7))
ctx.set_option("sell_dict", 0)
print("stored values ms", ctx.spmv_bench(_ffi.MAT_A11, 7))
