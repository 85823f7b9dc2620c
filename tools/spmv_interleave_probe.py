"""Fine-level SpMV launch time under four protocols: back-to-back (the isolated loop), with a streaming axpy
between products, with a one-block kernel between products, alternating A11 / A22.
usage: python tools/spmv_interleave_probe.py [cells] [kernel ids...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from perphil_amd import _ffi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kernels = [int(a) for a in sys.argv[2:]] or [3]
ctx = _ffi.Context(0)
ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b = np.array([0], dtype=np.int64)
ctx.set_dirichlet(0, b, np.zeros(1)); ctx.set_dirichlet(1, b, np.zeros(1))
ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
byts = 12.0 * ctx.nnzb + 20.0 * ctx.n
for k in kernels:
    ctx.set_option("spmv_kernel", k)
    for mode, name in ((0, "back-to-back"), (1, "axpy between"), (2, "1-block kernel between"), (3, "alternate A11/A22")):
        ctx.set_option("spmv_bench_mode", mode)
        ms = ctx.spmv_bench(_ffi.MAT_A11, 60)
        print(f"kernel {k:2d} {name:24s} {ms:.4f} ms  {byts / ms / 1e6:7.0f} GB/s", flush=True)
ctx.set_option("spmv_bench_mode", 0)
