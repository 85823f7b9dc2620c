import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2)
for mode in (0, 1, 2, 0, 1, 2):
    ctx.set_option("asm_kernel", mode)
    for rep in range(3):
        ctx.set_option("invalidate_KM", 1)
        t0 = time.perf_counter()
        ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
        ctx.synchronize()
        t1 = time.perf_counter()
        tm = ctx.timers()
        print(f"mode={mode} rep={rep} wall={1e3*(t1-t0):.1f} ms  K/M={tm['assemble_ms']:.1f} ms blocks={tm['bc_blocks_ms']:.1f} ms", flush=True)
