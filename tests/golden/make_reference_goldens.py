"""
Collects the reference's own committed study outputs for the DPP hot path into one small JSON
fixture (values only — no reference source text).  Run in the build container, where the reference
checkout is mounted at /root/reference; the GPU box never sees that path and reads the JSON.

Sources:
  G1/G8  notebooks/conforming-galerkin-fem-operator-splitting-2D-perphil.ipynb  stored stdout:
         "0 SNES Function norm", KSP residual histories (plain GMRES head/tail, field-split LU)
  G2/G11 same notebook, stored slice arrays at x = 0.5 (monolithic LU, Picard)
  G3     same notebook, stored condition numbers (monolithic / macro / micro, 10x10)
  G4     notebooks/results-conforming-2d/conditioning/conditioning.csv
  G5     notebooks/results-conforming-3d/conditioning/conditioning_3d.csv
  G6/G9  notebooks/results-conforming-3d/petsc_profiling/petsc_perf_breakdown_3d.csv
  G7/G9  notebooks/results-conforming-2d/petsc_profiling/petsc_perf_breakdown.csv
  G10    notebooks/results-conforming-2d/convergence.csv, convergence_eoc.csv (observed orders)
  G12    src/perphil/experiments/_tests/test_petsc_profiling/test_perf_to_dict_regression.yml,
         src/perphil/forms/_tests/test_dpp_regressions/test_dpp_form_structure_regression.yml
"""
import csv
import json
import os
import re

REF = "/root/reference"
NB = os.path.join(REF, "notebooks")


def _rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def _notebook_outputs():
    nb = json.load(open(os.path.join(NB, "conforming-galerkin-fem-operator-splitting-2D-perphil.ipynb")))
    outs = []
    for cell in nb["cells"]:
        if cell["cell_type"] != "code":
            continue
        text = ""
        for o in cell.get("outputs", []):
            if "text" in o:
                text += "".join(o["text"])
            elif "data" in o and "text/plain" in o["data"]:
                text += "".join(o["data"]["text/plain"])
        outs.append(("".join(cell["source"]), text))
    return outs


def _ksp_history(text):
    return [float(m.group(1)) for m in re.finditer(r"KSP Residual norm ([0-9.eE+-]+)", text)]


def _arrays(text):
    arrs = []
    for m in re.finditer(r"array\(\[(.*?)\]\)", text, flags=re.S):
        arrs.append([float(t) for t in m.group(1).replace("\n", " ").split(",")])
    return arrs


def main():
    g = {"_about": "values copied from the reference's committed study outputs; see make_reference_goldens.py"}
    outs = _notebook_outputs()
    hist = [(src, _ksp_history(txt), txt) for src, txt in outs if "KSP Residual norm" in txt]
    # first three monitored solves: plain GMRES, GMRES+ILU, field-split LU GMRES
    g["G1_initial_residual_10x10"] = float(re.search(r"0 SNES Function norm ([0-9.eE+-]+)", hist[0][2]).group(1))
    g["G8_gmres_history_10x10"] = hist[0][1]
    g["G8_fieldsplit_lu_history_10x10"] = hist[2][1]
    slices = [(src, _arrays(txt)) for src, txt in outs if "array([0. , 0.1" in txt or "array([0. ," in txt]
    mono = next(a for src, a in slices if "p1_mono_at_x_mid_point" in src and len(a) == 3)
    g["G2_slice_x05_monolithic_10x10"] = {"y": mono[0], "p1": mono[1], "p2": mono[2]}
    pic = [a for src, a in slices if "picard" in src.lower() and len(a) == 3]
    if pic:
        g["G11_slice_x05_picard_10x10"] = {"y": pic[0][0], "p1": pic[0][1], "p2": pic[0][2]}
    conds = {}
    for src, txt in outs:
        for name, key in (("Monolithic system", "monolithic"), ("Macro system", "macro"), ("Micro system", "micro")):
            m = re.search(name + r" Condition Number: ([0-9.eE+-]+)", txt)
            if m and key not in conds:
                conds[key] = float(m.group(1))
    g["G3_condition_numbers_10x10"] = conds
    g["G4_conditioning_2d"] = [
        {k: float(v) for k, v in r.items()} for r in _rows(os.path.join(NB, "results-conforming-2d/conditioning/conditioning.csv"))]
    g["G5_conditioning_3d_hex"] = [
        {k: float(v) for k, v in r.items()} for r in _rows(os.path.join(NB, "results-conforming-3d/conditioning/conditioning_3d.csv"))]

    def perf(path):
        out = []
        for r in _rows(path):
            out.append({"approach": r["approach"], "nx": int(r["nx"]), "dofs": int(r["dofs"]), "num_cells": int(r["num_cells"]),
                        "iterations": int(r["iterations"]), "residual": float(r["residual"]), "time_total": float(r["time_total"])})
        return out

    g["G6_G9_perf_3d_tets"] = perf(os.path.join(NB, "results-conforming-3d/petsc_profiling/petsc_perf_breakdown_3d.csv"))
    g["G7_G9_perf_2d_q1"] = perf(os.path.join(NB, "results-conforming-2d/petsc_profiling/petsc_perf_breakdown.csv"))
    g["G10_convergence_2d"] = [
        {k: (v if k == "solver" else float(v)) for k, v in r.items()} for r in _rows(os.path.join(NB, "results-conforming-2d/convergence.csv"))]
    g["G10_convergence_2d_eoc"] = [
        {"solver": r["solver"], "err": r["err"], "slope": float(r["slope"])} for r in _rows(os.path.join(NB, "results-conforming-2d/convergence_eoc.csv"))]
    with open(os.path.join(NB, "results-conforming-2d/convergence.csv")) as f:
        g["convergence_csv_columns"] = f.readline().strip().split(",")
    with open(os.path.join(NB, "results-conforming-3d/petsc_profiling/petsc_perf_breakdown_3d.csv")) as f:
        g["perf_csv_columns_3d"] = f.readline().strip().split(",")
    g["G12_structure"] = {"mesh_2x2_dofs": 18, "mesh_2x2_num_cells": 4, "form_integrals": 4, "form_rank": 2}
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "reference_goldens.json"), "w") as f:
        json.dump(g, f, indent=1)
    print("wrote reference_goldens.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in g.items()})


if __name__ == "__main__":
    main()
