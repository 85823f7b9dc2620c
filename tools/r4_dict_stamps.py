"""Round 4: in-kernel timeline of k_spmv_dict_walk<0> (diagnostic build with -DPPH_DW_STAMPS: s_memtime stamps of wave 0 of
every workgroup; ROCm 7.2 on this image has no thread-trace decoder).  Prints, over the workgroups of the LAST of a few
back-to-back launches: start skew, table -> LDS, range set-up + first requests, cycles per 4-step block of the straight-line
loop (median / p10 / p90), tail, and the share of a workgroup's life spent in the loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from perphil_amd import _ffi
import bench
N = 256
ctx = _ffi.Context(0); ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
for kv in sys.argv[1:]:
    ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
print("ms per launch (no stamps)", round(ctx.spmv_bench(_ffi.MAT_A11, 100), 4))
ctx.set_option("dw_stamps", 1)
ms = ctx.spmv_bench(_ffi.MAT_A11, 20)
r, _ = ctx.rhs()
ctx.set_option("dw_stamps", 0)
st = r.view(np.uint64)[: (r.size // 64) * 64].reshape(-1, 64)[:8192]
used = st[:, 0] != 0
st = st[used].astype(np.int64)
G = st.shape[0]
print(f"ms per launch (stamped) {ms:.4f}; workgroups stamped {G}")
t0 = st[:, 0]; start = t0 - t0.min()
nst = st[:, 61]          # index of the stamp taken after the straight-line loop
life = st[:, 62] - t0
tab = st[:, 1] - t0
setup = st[:, 2] - st[:, 1]
q = lambda a: (int(np.percentile(a, 10)), int(np.median(a)), int(np.percentile(a, 90)))
print("cycles (p10, median, p90) at the s_memtime clock (100 MHz real-time counter beside it: see ratio)")
print("  workgroup start after the first one:", q(start))
print("  table -> LDS + barrier:", q(tab))
print("  range set-up + first three planes requested:", q(setup))
blocks = []
first = []
for w in range(G):
    k = int(nst[w])
    s = st[w, 2:k + 1]
    d = np.diff(s)
    if len(d) > 1:
        first.append(d[0]); blocks.extend(d[1:-1].tolist() if len(d) > 2 else [])
print("  first 4-step block (waits for the first requests):", q(np.array(first)))
print("  later 4-step blocks of the straight-line loop:", q(np.array(blocks)), f"-> {np.median(blocks) / 4:.0f} cycles per step and wave")
tail = np.array([st[w, int(nst[w]) + 1] - st[w, int(nst[w])] for w in range(G)])
print("  single steps after the loop:", q(tail))
print("  workgroup lifetime:", q(life), " share in 4-step blocks:", round(float(np.sum([st[w, int(nst[w])] - st[w, 2] for w in range(G)]) / life.sum()), 3))
span = (st[:, 62].max() - t0.min())
print(f"  kernel span {span} cycles; sum of lifetimes / span = {life.sum() / span:.1f} workgroups resident on average (of {G})")
