#!/usr/bin/env python3
"""Summarises the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one bench.py run into
profiles/<name>.json: per-launch HBM traffic of the SpMV kernel = mean over its launches of
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes.  The factor 2 is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request of a wide stream); it is
re-checked here on the calibration kernels k_bw_read / k_bw_copy when they are present."""
import csv
import glob
import json
import sys

fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]


def load(d):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    per = {}
    for r in csv.DictReader(open(f)):
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return per


F, W = load(fetch_dir), load(write_dir)
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --cells 256 "
                 "--steps 1 --warmup 0 --no-cpu-baseline --skip-fine-bench", "unit_note": "counters in KB; FETCH_SIZE x2 (gfx950)",
       "kernels": {}}
tot_f = tot_w = n = 0
for name in F:
    if "k_spmv_wide" not in name:
        continue
    f, w = F[name], W.get(name, [])
    res["kernels"][name[:48]] = {"launches": len(f), "fetch_kb_mean": sum(f) / len(f), "write_kb_mean": sum(w) / max(len(w), 1)}
    tot_f += sum(f); tot_w += sum(w); n += len(f)
res["spmv_launches"] = n
res["traffic_bytes_per_launch"] = (2.0 * tot_f + tot_w) * 1024.0 / n
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
