"""Workload for the SQ / cache counter passes: 10 fine-level stencil-ELL SpMVs and 10 CSR SpMVs of the 256^3 block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
print("sell (symmetric) ms", ctx.spmv_bench(_ffi.MAT_A11, 10))
ctx.set_option("sell_sym", 0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
print("sell (full) ms", ctx.spmv_bench(_ffi.MAT_A11, 10))
ctx.set_option("op_format", 0)
print("csr ms", ctx.spmv_bench(_ffi.MAT_A11, 10))
