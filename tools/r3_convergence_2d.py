"""The reference's 2D h-convergence study (notebooks/results-conforming-2d/convergence.csv, 30 rows, and
convergence_eoc.csv, 20 slopes) on the HIP path through perphil_amd.convergence_2d; writes
profiles/r03_convergence_2d.csv (every column with the reference's value beside it) and r03_convergence_2d_eoc.csv.
    gpurun -- 'python tools/r3_convergence_2d.py gpurun_out/r3conv'"""
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3conv"
    os.makedirs(out, exist_ok=True)
    import perphil_amd as pa
    from perphil_amd import convergence_2d as c2
    from perphil_amd.iterative_bench import Approach
    import test_reference_sweep as trs

    G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")))
    rows, bad = [], 0
    with open(os.path.join(out, "r03_convergence_2d.csv"), "w", newline="") as f:
        w = None
        for g in G["G10_convergence_2d"]:
            spec = c2.approach_solvers([Approach(g["solver"])])[0]
            t0 = time.perf_counter()
            row = c2.run_one(N=int(g["N"]), solver=spec, quad=True, degree=1, params=pa.DPPParameters())
            dt = time.perf_counter() - t0
            rows.append(row)
            ok = True
            try:
                trs.check_convergence_row(row, g)
            except AssertionError as e:
                ok, bad = False, bad + 1
                print("MISMATCH", g["solver"], g["N"], e, flush=True)
            rec = dict(row)
            for k in ("it", "res", "e1_L2", "e2_L2", "e1_H1s", "e2_H1s"):
                rec["ref_" + k] = g[k]
            rec["max_rel_err_diff"] = max(abs(row[k] / g[k] - 1) for k in c2.ERROR_FIELDS)
            rec["wall_s"] = round(dt, 4)
            rec["ok"] = int(ok)
            if w is None:
                w = csv.DictWriter(f, fieldnames=list(rec.keys()))
                w.writeheader()
            w.writerow(rec)
            f.flush()
            print(g["solver"], int(g["N"]), "it", row["it"], int(g["it"]), "errdiff %.2e" % rec["max_rel_err_diff"], "%.2fs" % dt, flush=True)
    ref = {(r["solver"], r["err"]): r["slope"] for r in G["G10_convergence_2d_eoc"]}
    eoc = c2.observed_orders(rows)
    for r in eoc:
        r["ref_slope"] = ref[(r["solver"], r["err"])]
        r["rel_diff"] = abs(r["slope"] / r["ref_slope"] - 1)
    c2.write_csv(eoc, os.path.join(out, "r03_convergence_2d_eoc.csv"))
    print("rows", len(rows), "mismatches", bad, "max slope rel diff %.2e" % max(r["rel_diff"] for r in eoc))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
