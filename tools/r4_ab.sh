#!/bin/bash
# Round 4: A/B of builds of the library on one box (PERPHIL_HIP_LIB): isolated dictionary product + a short bench line each.
# Every step under its own timeout, everything appended to gpurun_out/ab.log (no pipes: a hung step must be seen).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/ab.log
for lib in "$@"; do
  echo "== $lib" >> gpurun_out/ab.log
  PERPHIL_HIP_LIB=$PWD/perphil_amd/$lib timeout -k 5 120 python tools/r4_dict_probe.py 256 >> gpurun_out/ab.log 2>&1 || { echo "probe failed / timed out" >> gpurun_out/ab.log; continue; }
  case $lib in *nolds*) continue;; esac   # (timing probes with wrong coefficients: isolated product only)
  PERPHIL_HIP_LIB=$PWD/perphil_amd/$lib timeout -k 5 180 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-csr --no-configs > gpurun_out/ab_$lib.json 2>> gpurun_out/ab.log || { echo "bench failed / timed out" >> gpurun_out/ab.log; continue; }
  python - >> gpurun_out/ab.log 2>&1 <<PY
import json
b=json.loads([l for l in open("gpurun_out/ab_$lib.json") if l.startswith("{")][-1])
r=b["roofline"]; c=b["config"]
print("ms_per_step", round(b["ms_per_step"],3), "frac", r["frac"], "fine", r["fine_level"]["avg_launch_ms"], "in solver", r["fine_level_in_solver"]["avg_launch_ms"], "sweeps", c["picard_sweeps"], c["inner_cg_iterations"], "res", c["final_residual"])
PY
done
grep -v amdgpu.ids gpurun_out/ab.log
