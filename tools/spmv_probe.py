import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
byts = 12.0 * ctx.nnzb + 20.0 * ctx.n
kerns = [int(a) for a in sys.argv[2:]] or [3, 10, 11]
for blocks in (512, 1024, 2048):
    ctx.set_option("spmv_blocks", blocks)
    for kern in kerns:
        ctx.set_option("spmv_kernel", kern)
        ms = ctx.spmv_bench(_ffi.MAT_A11, 30)
        print(f"N={N} blocks={blocks} kernel={kern} {ms:.4f} ms {byts/1e9/(ms/1e3):.1f} GB/s", flush=True)
