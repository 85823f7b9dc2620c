"""Splits the per-dispatch counters of one kernel family (rocprofv3 --pmc ... --kernel-trace CSV) into groups of
<group> consecutive dispatches (= the configurations of a probe script, in its order) and prints the mean of every
counter over the LAST 3 dispatches of each group.   usage: r3_pmc_by_order.py <counter_collection.csv> <substr> <group>"""
import csv, sys
from collections import OrderedDict
f, substr, group = sys.argv[1], sys.argv[2], int(sys.argv[3])
disp = OrderedDict()
for r in csv.DictReader(open(f)):
    if substr not in r["Kernel_Name"]:
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
for g in range(0, len(ids), group):
    chunk = ids[g:g + group][-min(3, group):]
    names = sorted(disp[chunk[0]])
    means = {n: sum(disp[i].get(n, 0.0) for i in chunk) / len(chunk) for n in names}
    hit = means.get("TCC_HIT_sum", 0.0); miss = means.get("TCC_MISS_sum", 0.0)
    extra = f" hit_rate {hit / (hit + miss):.3f}" if hit + miss > 0 else ""
    rd = means.get("TCC_EA0_RDREQ_sum")
    if rd is not None:
        extra += f" rd_GB(128B/req) {rd * 128 / 1e9:.3f}"
    print(f"cfg {g // group}: " + " ".join(f"{n}={means[n]:.4g}" for n in names) + extra)
