import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
if order == "ffi_first":
    from perphil_amd import _ffi
    import torch
else:
    import torch
    from perphil_amd import _ffi
print(order, "torch.cuda.device_count()", torch.cuda.device_count(), flush=True)
try:
    c = _ffi.Context(0); print(order, "Context ok", flush=True)
except Exception as e:
    print(order, "Context failed:", e, flush=True)
try:
    print(order, "torch cuda avail", torch.cuda.is_available(), torch.zeros(1, device="cuda").item(), flush=True)
except Exception as e:
    print(order, "torch cuda failed:", e, flush=True)
maps = open("/proc/self/maps").read()
print(order, sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l}), flush=True)
