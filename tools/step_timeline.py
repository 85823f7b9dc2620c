"""Kernel timeline of the LAST timed step of a rocprofv3 --kernel-trace CSV of bench.py (from the last fine-level assembly on).
usage: python tools/step_timeline.py <dir> [min_us] [fine_assembly_ns] [step]   -> one line per kernel: start offset, duration,
gap before, name.  step = index of the fine-level assembly launch the step starts with (bench.py: warm-up steps, timed steps, then
the instrumented / full-storage / CSR steps, which run with time_spmv or other formats - pick a timed one)"""
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
big = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_asm_tile") or r["Kernel_Name"].startswith("void k_elem_rows")]
# the last step starts at the last assembly launch that follows a long pause
starts = [i for i in big if i == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 0]
thr = float(sys.argv[3] if len(sys.argv) > 3 else 2e6)
fine = [i for i in big if int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) > thr]   # fine-level assembly launches = step starts
which = int(sys.argv[4]) if len(sys.argv) > 4 else len(fine) - 1
first = fine[which]
last = fine[which + 1] if which + 1 < len(fine) else len(rows)
rows = rows[:last]
rows = rows[first:]
t0 = int(rows[0]["Start_Timestamp"])
prev = t0
tot = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = (e - s) / 1e3
    tot[name] = tot.get(name, 0.0) + d
    if d >= min_us:
        print(f"{(s - t0) / 1e3:10.1f} us  {d:8.1f} us  gap {(s - prev) / 1e3:7.1f}  grid {r.get('Grid_Size', '?'):>9s}  {name}")
    prev = e
print("---- totals of this step (ms)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{v / 1e3:8.3f}  {k}")
print(f"span {(prev - t0) / 1e6:.3f} ms, busy {sum(tot.values()) / 1e3:.3f} ms, {len(rows)} kernels")
