"""Round 4: isolated timing of the fine-level multigrid transfer kernels (256^3 hexahedra): option transfer_bench."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
cfg = bench.picard_cfg(_ffi, 1e-10, 1, 1e-1, 1)
cfg.picard_max_it, cfg.inner_max_it = 1, 1   # (one sweep of one-iteration block solves: builds the hierarchy; probe builds have wrong transfers)
ctx.solve(cfg, fetch=False, raise_on_diverged=False)
for _ in range(3):
    ctx.set_option("transfer_bench", 50)
