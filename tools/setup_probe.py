import os, sys, time
sys.path.insert(0, os.getcwd())
from perphil_amd import _ffi
import bench
N = 256
ctx = _ffi.Context(0)
for rep in range(2):
    t0 = time.perf_counter(); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N); ctx.synchronize() if hasattr(ctx, "synchronize") else None; t1 = time.perf_counter()
    b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0); t2 = time.perf_counter()
    ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); t3 = time.perf_counter()
    print(f"mesh_build {1e3*(t1-t0):.1f} ms, mms_boundary (host) {1e3*(t2-t1):.1f} ms, set_dirichlet x2 {1e3*(t3-t2):.1f} ms, timers mesh_ms {ctx.timers()['mesh_ms']:.1f}", flush=True)
