"""Round 3: node kernel vs tile kernel, fine-level assembly time at 64^3 / 128^3 / 256^3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
for N in (64, 128, 256):
    ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
    b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
    ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2)
    for node in (0, 1, 0, 1):
        ctx.set_option("asm_node", node)
        for _ in range(3):
            ctx.set_option("invalidate_KM", 1)
            ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
        print(f"N {N} asm_node {node}: assemble {ctx.timers()['assemble_ms']:.3f} ms", flush=True)
    ctx.close()
