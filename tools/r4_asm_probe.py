"""Round 4: timing probes of k_asm_node2 at N^3 (fine level only, dictionaries off): full kernel, each launch alone, without
the operator stores, without the coordinate loads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
ctx.set_option("sell_dict", 0)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2)
def run(label, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    ts = []
    for _ in range(reps):
        ctx.set_option("invalidate_KM", 1)
        ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
        ts.append(ctx.timers()["assemble_ms"])
    print(f"{label:55s} assemble_ms {min(ts):.3f}  (all: {' '.join('%.3f' % t for t in ts)})", flush=True)
    for k in opts:
        ctx.set_option(k, {"asm_node": 1, "asm_node_split_min": 200000}.get(k, 0))
run("tile kernel (asm_node 0)", asm_node=0)
run("k_asm_node2, both launches")
run("k_asm_node2, straight-line launch only", asm_node_probe=2)
run("k_asm_node2, predicated launch only", asm_node_probe=1)
run("both, one kernel (PATH 0)", asm_node_split_min=1e12)
run("both launches, XCD-contiguous block order", asm_node_xmap=1)
