#!/usr/bin/env python3
"""All seven rows of the reference's notebooks/results-conforming-3d/conditioning/conditioning_3d.csv through the public API
on the device-assembled matrix (the test suite stops at N = 12: the dense SVD of 9 826 dofs takes minutes of host time).
Writes <outdir>/r03_conditioning_3d.csv in the reference's columns + the reference's values.  usage: r3_conditioning_3d.py <outdir>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pandas as pd  # noqa: E402
import perphil_amd as pa  # noqa: E402
from perphil_amd import fd  # noqa: E402
from perphil_amd.iterative_bench import estimate_condition_numbers  # noqa: E402
from perphil_amd.manufactured_solutions import exact_expressions_3d  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))["G5_conditioning_3d_hex"]
rows = []
for g in G:
    N = int(g["N"])
    t0 = time.perf_counter()
    mesh = fd.UnitCubeMesh(N, N, N, hexahedral=True)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    params = pa.DPPParameters(k1=1.0, k2=1.0 / 1e2, beta=1.0, mu=1.0)
    _u1, p1e, _u2, p2e = exact_expressions_3d(mesh, params)
    bcs = [fd.DirichletBC(W.sub(0), p1e, "on_boundary"), fd.DirichletBC(W.sub(1), p2e, "on_boundary")]
    c = estimate_condition_numbers(W, params=params, bcs=bcs, use_sparse=True, num_of_factors=0)
    rows.append({"N": N, "h": 1.0 / N, "cond_monolithic": c["monolithic"], "cond_macro": c["macro"], "cond_micro": c["micro"],
                 "n_dofs": W.dim(), "n0": W.sub(0).dim(), "n1": W.sub(1).dim(), "ref_cond_monolithic": g["cond_monolithic"],
                 "ref_cond_macro": g["cond_macro"], "ref_cond_micro": g["cond_micro"]})
    rel = max(abs(c[k] / g[r] - 1.0) for k, r in (("monolithic", "cond_monolithic"), ("macro", "cond_macro"), ("micro", "cond_micro")))
    print(f"N {N:3d}: kappa {c['monolithic']:.10e} / {c['macro']:.10e} / {c['micro']:.10e}  max rel diff vs reference {rel:.2e}  ({time.perf_counter() - t0:.1f} s)", flush=True)
pd.DataFrame(rows).to_csv(os.path.join(out, "r03_conditioning_3d.csv"), index=False)
