"""Round 3: tile assembly at 256^3 - XCD-contiguous tile order (asm_node_xmap) on / off; 3 assemblies per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2)
for xmap in (0, 1, 0, 1):
    ctx.set_option("asm_node_xmap", xmap)
    for _ in range(reps):
        ctx.set_option("invalidate_KM", 1)
        ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
    print(f"xmap {xmap}: assemble {ctx.timers()['assemble_ms']:.3f} ms", flush=True)
