"""Read-only stream at few workgroups per CU (what concurrency does a 16 B/lane stream need?)."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from perphil_amd import _ffi
fn = _ffi.lib.pph_bw_probe
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
fn.restype = C.c_int
ctx = _ffi.Context(0)
for blocks in (256, 512, 1024, 2048):
    ms = C.c_double()
    fn(ctx._h, 2 << 30, 0, blocks, C.byref(ms))
    print(f"read 2 GiB blocks={blocks} (x256 threads): {ms.value:.3f} ms  {(2 << 30) / 1e9 / (ms.value / 1e3):.0f} GB/s", flush=True)
