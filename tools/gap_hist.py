"""Idle gaps between consecutive kernel dispatches of a rocprofv3 --kernel-trace CSV (launch- or sync-bound?).
usage: python tools/gap_hist.py <dir> [skip_first_n_kernels]"""
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e6
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
print(f"{len(rows)} kernels, busy {busy:.1f} ms of {span:.1f} ms span ({100 * busy / span:.1f} %)")
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
    g = [x for x in gaps if lo <= x < hi]
    print(f"  gaps {lo:>4}-{hi if hi < 1e9 else 'inf':>4} us: {len(g):6d}  sum {sum(g) / 1e3:8.2f} ms")
from collections import Counter
big, cnt = Counter(), Counter()
for (a, b), x in zip(zip(rows, rows[1:]), gaps):
    if 5 <= x < 200:      # inside the steps (longer gaps are host-side set-up between phases)
        key = (a["Kernel_Name"].split("(")[0][-28:], b["Kernel_Name"].split("(")[0][-28:])
        big[key] += x
        cnt[key] += 1
for k, v in big.most_common(14):
    print(f"  {v / 1e3:7.2f} ms in {cnt[k]:5d} gaps (avg {v / cnt[k]:5.1f} us)  after {k[0]:30s} before {k[1]}")
