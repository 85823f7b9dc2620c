"""Durations of the runtime's copy kernels in a rocprofv3 --kernel-trace CSV, with the kernels around them.
usage: python tools/copy_hist.py <dir> [skip_first_n]"""
import csv, glob, sys
from collections import Counter
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(sys.argv[2]) if len(sys.argv) > 2 else 0:]
tot = Counter(); cnt = Counter()
for i, r in enumerate(rows):
    if "copyBuffer" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]:
        continue
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    prev = rows[i - 1]["Kernel_Name"].split("(")[0][-26:] if i else "-"
    nxt = rows[i + 1]["Kernel_Name"].split("(")[0][-26:] if i + 1 < len(rows) else "-"
    key = (r["Kernel_Name"][:28], "big" if us > 30 else "small", prev, nxt)
    tot[key] += us; cnt[key] += 1
for k, v in tot.most_common(16):
    print(f"{v / 1e3:7.2f} ms  n={cnt[k]:4d}  {k[0]:28s} {k[1]:5s} after {k[2]:28s} before {k[3]}")
