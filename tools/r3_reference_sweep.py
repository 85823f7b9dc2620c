#!/usr/bin/env python3
"""Runs the reference's published solver sweep on the HIP path and writes it in the reference's CSV schema, with the
reference's own figures beside it (ref_iterations, ref_residual, ref_time_total: hardware unstated - context only):
    profiles/r03_reference_sweep_2d.csv   rows of notebooks/results-conforming-2d/petsc_profiling/petsc_perf_breakdown.csv
    profiles/r03_reference_sweep_3d.csv   rows of notebooks/results-conforming-3d/petsc_profiling/petsc_perf_breakdown_3d.csv
Protocol of the reference (one warm-up solve, `repeats` timed solves with re-assembly, one more for iterations / residual;
repeats = 5 in 2D and 3 in 3D like the stored runs).  usage: r3_reference_sweep.py <outdir> [2d|3d|both]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pandas as pd  # noqa: E402

from perphil_amd.iterative_bench import Approach  # noqa: E402
from perphil_amd.profiling_3d import run_perf_once, run_perf_once_3d  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
which = sys.argv[2] if len(sys.argv) > 2 else "both"
G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))
os.makedirs(out, exist_ok=True)
for dim, key, name in ((2, "G7_G9_perf_2d_q1", "r03_reference_sweep_2d.csv"), (3, "G6_G9_perf_3d_tets", "r03_reference_sweep_3d.csv")):
    if which not in ("both", f"{dim}d"):
        continue
    rows = []
    for g in G[key]:
        ap, nx = Approach(g["approach"]), int(g["nx"])
        row = run_perf_once(nx, nx, ap, eager=True, repeats=5) if dim == 2 else run_perf_once_3d(nx, ap, eager=True, repeats=3)
        row.pop("metadata", None)
        row["ref_iterations"], row["ref_residual"], row["ref_time_total"] = g["iterations"], g["residual"], g["time_total"]
        rows.append(row)
        print(f"{dim}D nx={nx:4d} {ap.value:34s} its {row['iterations']:6d} (ref {g['iterations']:6d})  residual {row['residual']:.4e} "
              f"(ref {g['residual']:.4e})  time_total {row['time_total']:.4e} s (ref {g['time_total']:.4e} s)", flush=True)
    pd.DataFrame(rows).to_csv(os.path.join(out, name), index=False)
