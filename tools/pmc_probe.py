"""Workload for the rocprofv3 --pmc passes: a calibration stream of known size (read-only, 16 B per lane)
followed by fine-level scalar-block SpMVs of the 256^3 problem."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fn = _ffi.lib.pph_bw_probe
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
fn.restype = C.c_int
ctx = _ffi.Context(0)
ms = C.c_double()
fn(ctx._h, 4 << 30, 0, 1024, C.byref(ms))      # k_bw_read: 4 GiB read per launch
fn(ctx._h, 2 << 30, 1, 1024, C.byref(ms))      # k_bw_copy: 2 GiB read + 2 GiB written per launch
ctx.mesh_build(3, _ffi.CELL_HEX, N, N, N)
b, g1, g2 = bench.mms_boundary(N, 1.0, 1e-2, 1.0, 1.0)
ctx.set_dirichlet(0, b, g1); ctx.set_dirichlet(1, b, g2)
ctx.assemble(1.0, 1e-2, 1.0, 1.0, monolithic=False)
print("spmv ms", ctx.spmv_bench(_ffi.MAT_A11, 5), "alg bytes", 12.0 * ctx.nnzb + 20.0 * ctx.n)
