"""Per-kernel, per-grid-size time of the LAST step in a rocprofv3 kernel trace of bench.py (tools/r3_trace.sh): which
multigrid level costs what.  usage: python tools/r3_trace_levels.py gpurun_out/trace3/kernel_trace.csv"""
import collections
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))

    def nm(r):
        m = re.match(r"(?:void )?(\w+)(<[^>]*>)?", r["Kernel_Name"])
        return m.group(1) + (m.group(2) or "")

    # a step starts with the fine-level assembly kernel (the largest k_asm_node grid)
    asm = [i for i, r in enumerate(rows) if nm(r).startswith("k_asm_node")]
    gmax = max(int(rows[i]["Grid_Size_X"]) for i in asm)
    starts = [i for i in asm if int(rows[i]["Grid_Size_X"]) == gmax]
    lo = starts[-1]
    step = rows[lo:]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for r in step:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        busy += d
        agg[(nm(r), int(r["Grid_Size_X"]))][0] += 1
        agg[(nm(r), int(r["Grid_Size_X"]))][1] += d
    print("last step: %d kernels, span %.2f ms, busy %.2f ms" % (len(step), (t1 - t0) / 1e6, busy / 1e3))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
        print(k[0][:40].ljust(40), str(k[1]).rjust(9), str(v[0]).rjust(5), "%9.1f us avg" % (v[1] / v[0]), "%8.3f ms" % (v[1] / 1e3))


if __name__ == "__main__":
    main()
