"""Bucket the fine-level SpMV dispatches of a rocprofv3 --kernel-trace CSV by duration and by the kernel that
ran just before them (is the in-solver launch slower than the isolated loop, and after which producer?).
usage: python tools/spmv_trace_hist.py <dir with *_kernel_trace.csv> [min_us]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for fn in files:
    with open(fn) as f:
        rows += list(csv.DictReader(f))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = defaultdict(list)
by_variant = defaultdict(list)
gaps = []
for i, r in enumerate(rows):
    name = r["Kernel_Name"]
    if "k_spmv" not in name:
        continue
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if us < min_us:
        continue
    p = rows[i - 1]["Kernel_Name"].split("(")[0][-40:] if i else "-"
    prev[p].append(us)
    by_variant[name.split("(")[0][-40:]].append(us)
    if i:
        gaps.append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
print(f"{sum(len(v) for v in by_variant.values())} SpMV dispatches >= {min_us} us")
for k, v in sorted(by_variant.items()):
    print(f"  variant {k:42s} n={len(v):4d} mean={sum(v)/len(v):8.1f} min={min(v):8.1f} max={max(v):8.1f}")
for k, v in sorted(prev.items(), key=lambda kv: -len(kv[1])):
    print(f"  after   {k:42s} n={len(v):4d} mean={sum(v)/len(v):8.1f} min={min(v):8.1f} max={max(v):8.1f}")
if gaps:
    print(f"  idle gap before the dispatch: mean {sum(gaps)/len(gaps):.1f} us, max {max(gaps):.1f} us")
