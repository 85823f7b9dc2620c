#!/bin/bash
# Round 4: A/B of library builds on whole 256^3 steps (alternating, three rounds): ms per step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; : > gpurun_out/ab2.log
for rep in 1 2 3; do
for lib in "$@"; do
  PERPHIL_HIP_LIB=$PWD/perphil_amd/$lib timeout -k 5 180 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api-wall --skip-csr --skip-fine-bench --no-configs > gpurun_out/ab2_$lib.json 2>> gpurun_out/ab2.log || { echo "$lib failed" >> gpurun_out/ab2.log; continue; }
  python - >> gpurun_out/ab2.log 2>&1 <<PY
import json
b=json.loads([l for l in open("gpurun_out/ab2_$lib.json") if l.startswith("{")][-1])
print("$lib", "ms_per_step", round(b["ms_per_step"],3), "in solver", b["roofline"]["fine_level_in_solver"]["avg_launch_ms"], "solve", b["config"]["solve_ms"])
PY
done; done
grep -v amdgpu.ids gpurun_out/ab2.log
