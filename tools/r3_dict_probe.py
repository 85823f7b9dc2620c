"""Round 3: row-dictionary product (option sell_dict) on the 256^3 block against the stored-value kernel: per epilogue
mode, grid cap and walk.  usage: r3_dict_probe.py [N] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)


def make(dict_on):
    ctx = _ffi.Context(0)
    ctx.set_option("sell_dict", dict_on)
    ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
    ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2)
    t0 = time.perf_counter(); ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False); t1 = time.perf_counter()
    ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False); t2 = time.perf_counter()
    tm = ctx.timers()
    print(f"dict {dict_on}: first assembly {1e3 * (t1 - t0):.2f} ms wall, second {1e3 * (t2 - t1):.2f} ms wall / {tm['assemble_ms']:.3f} ms events; "
          f"operators {tm['dict_operators']} classes {tm['dict_classes']} status {tm['dict_status']}", flush=True)
    return ctx


for dict_on in (0, 1):
    ctx = make(dict_on)
    W0 = {"sell_dict_walk": 0, "sell_dict_zwalk": 0, "sell_dict_blocks": 2048}
    cfgs = [{}] if not dict_on else [{}, {"sell_dict_blocks": 512}, {"sell_dict_blocks": 2048}, {"sell_dict_blocks": 4096}, {"sell_flags": 4}, {"sell_flags": 4, "sell_dict_blocks": 2048}, W0]
    base = {"sell_dict_blocks": 1024, "sell_dict_zwalk": -1, "sell_dict_walk": 1, "sell_flags": 0}
    for c in cfgs:
        if dict_on:
            o = dict(base); o.update(c)
            for k, v in o.items():
                ctx.set_option(k, v)
        ms = [min(ctx.spmv_bench(w, reps) for _ in range(3)) for w in (_ffi.MAT_A11, _ffi.MAT_A12)]
        print(f"dict {dict_on} {c}: A11 {ms[0]:.4f} ms  A12 {ms[1]:.4f} ms", flush=True)
    ctx.close()
