"""Back-to-back launch cost of the stencil-ELL product against the CSR product on a tiny block (is the ~6 us gap
the kernel trace shows next to every k_spmv_sell real?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from perphil_amd import _ffi
import bench
for N in (8, 16, 32):
    for fmt in (1, 0):
        ctx = _ffi.Context(0)
        ctx.set_option("op_format", fmt)
        ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
        b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
        ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
        ms = min(ctx.spmv_bench(_ffi.MAT_A11, 2000) for _ in range(3))
        print(f"N {N} format {'sell' if fmt else 'csr'}: {ms * 1e3:.2f} us per launch", flush=True)
