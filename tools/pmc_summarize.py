#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only) of one
bench.py run into profiles/<name>.json: HBM-side traffic per launch of the kernels whose name contains <substr>
= mean over their launches of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes.  The factor 2 is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request of a wide stream; calibrated on
k_bw_read / k_bw_copy in profiles/r01_pmc_calibration.txt).  The result is stamped with the kernel and the launch
count per step it was taken on: bench.py reports `roofline.traffic` from it only when both still match.

usage: pmc_summarize.py <fetch_dir> <write_dir> <out.json> <kernel substring> <steps in the run> [label] [sym] [dict]"""
import csv
import glob
import json
import sys

fetch_dir, write_dir, out, substr, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
label = sys.argv[6] if len(sys.argv) > 6 else substr
symmetric = (len(sys.argv) > 7 and sys.argv[7] == "sym")
dictionary = (len(sys.argv) > 8 and sys.argv[8] == "dict")


def load(d):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return per


F, W = load(fetch_dir), load(write_dir)
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over bench.py "
                 "--cells 256 --steps 1 --warmup 0 --no-cpu-baseline --skip-fine-bench --skip-csr (one timed + one instrumented step)",
       "unit_note": "counters in KB; FETCH_SIZE x2 (gfx950 correction)", "kernel": label, "kernels": {}}
tot_f = tot_w = n = 0
for name in F:
    if substr not in name:
        continue
    f, w = F[name], W.get(name, [])
    res["kernels"][name[:60]] = {"launches": len(f), "fetch_kb_mean": sum(f) / len(f), "write_kb_mean": sum(w) / max(len(w), 1),
                                 "fetch_kb_max": max(f), "write_kb_max": max(w) if w else 0.0}
    tot_f += sum(f); tot_w += sum(w); n += len(f)
res["symmetric"] = symmetric
res["dictionary"] = dictionary
res["launches_in_run"] = n
res["launches_per_step"] = n // max(steps, 1)
res["traffic_bytes_per_launch"] = (2.0 * tot_f + tot_w) * 1024.0 / max(n, 1)
res["traffic_bytes_per_step"] = (2.0 * tot_f + tot_w) * 1024.0 / max(steps, 1)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
