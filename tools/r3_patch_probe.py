"""Round 3: patch walk of the symmetric stencil-ELL product (wave-private LDS mirrors) vs the cached kernel, 256^3 block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
for patch, z, blocks in ((0, 16, 0), (1, 4, 0), (1, 8, 0), (1, 16, 0), (1, 32, 0), (1, 64, 0), (1, 16, 512), (1, 32, 512), (1, 16, 768), (0, 16, 0)):
    ctx.set_option("sell_patch", patch); ctx.set_option("sell_patch_z", z); ctx.set_option("sell_blocks", blocks)
    ms = min(ctx.spmv_bench(_ffi.MAT_A11, reps) for _ in range(1 if reps < 10 else 3))
    print(f"N {N} sell_patch {patch} z {z} blocks {blocks}: {ms:.4f} ms", flush=True)
