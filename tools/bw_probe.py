"""HBM bandwidth calibration on this device: read-only and copy streams (16 B per lane)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perphil_amd import _ffi
fn = _ffi.lib.pph_bw_probe
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
fn.restype = C.c_int
ctx = _ffi.Context(0)
for bytes_ in (1 << 30, 4 << 30):
    for mode, name in ((0, "read"), (1, "copy"), (2, "mix 32B+16B")):
        for blocks in (1024, 2048, 4096, 8192):
            ms = C.c_double()
            st = fn(ctx._h, bytes_, mode, blocks, C.byref(ms))
            moved = bytes_ * (2 if mode == 1 else 1.5 if mode == 2 else 1)
            print(f"{name} {bytes_ >> 30} GiB blocks={blocks}: {ms.value:.3f} ms  {moved / 1e9 / (ms.value / 1e3):.0f} GB/s", flush=True)
