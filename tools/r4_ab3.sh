#!/bin/bash
# Round 4: A/B of library builds with bench options per build: "lib:opt=val,opt=val" ...; isolated product + whole steps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; : > gpurun_out/ab3.log
for rep in 1 2; do
for spec in "$@"; do
  lib=${spec%%:*}; opts=${spec#*:}; [ "$opts" = "$spec" ] && opts=""
  sets=""; for kv in ${opts//,/ }; do sets="$sets --set $kv"; done
  PERPHIL_HIP_LIB=$PWD/perphil_amd/$lib timeout -k 5 180 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api-wall --skip-csr --no-configs $sets > gpurun_out/ab3.json 2>> gpurun_out/ab3.log || { echo "$spec failed" >> gpurun_out/ab3.log; continue; }
  python - >> gpurun_out/ab3.log 2>&1 <<PY
import json
b=json.loads([l for l in open("gpurun_out/ab3.json") if l.startswith("{")][-1])
r=b["roofline"]
print("$spec", "ms_per_step", round(b["ms_per_step"],3), "isolated", r["fine_level"]["avg_launch_ms"], "in solver", r["fine_level_in_solver"]["avg_launch_ms"], "sweeps", b["config"]["picard_sweeps"], b["config"]["inner_cg_iterations"])
PY
done; done
grep -v amdgpu.ids gpurun_out/ab3.log
