"""
PETSc-style option dictionaries: the configuration surface of ``solve_dpp``.  The data are the same
key/value pairs as reference ``src/perphil/solvers/parameters.py:1-102`` (they are configuration, not
code); ``perphil_amd.solver.translate_options`` maps the supported subset onto ``pph_solver_cfg``.
"""
_MAX_ITERATION_NUMBER = 50000

LINEAR_SOLVER_PARAMS: dict = {
    "mat_type": "aij",
    "ksp_type": "preonly",
    "pc_type": "lu",
    "pc_factor_mat_solver_type": "mumps",
}

GMRES_PARAMS: dict = {
    "mat_type": "aij",
    "ksp_type": "gmres",
    "ksp_rtol": 1.0e-8,
    "ksp_atol": 1.0e-12,
    "ksp_max_it": _MAX_ITERATION_NUMBER,
}

PLAIN_GMRES_PARAMS: dict = {"pc_type": "none", **GMRES_PARAMS}
GMRES_JACOBI_PARAMS: dict = {"pc_type": "jacobi", **GMRES_PARAMS}
GMRES_ILU_PARAMS: dict = {"pc_type": "ilu", "pc_factor_levels": 0, **GMRES_PARAMS}

_FIELDSPLIT_BASE = {
    "pc_type": "fieldsplit",
    "pc_fieldsplit_type": "multiplicative",
    "pc_fieldsplit_0_fields": "0",
    "pc_fieldsplit_1_fields": "1",
}
FIELDSPLIT_LU_PARAMS: dict = {**_FIELDSPLIT_BASE, "fieldsplit_0": LINEAR_SOLVER_PARAMS, "fieldsplit_1": LINEAR_SOLVER_PARAMS}
FIELDSPLIT_GMRES_PARAMS: dict = {**_FIELDSPLIT_BASE, "fieldsplit_0": PLAIN_GMRES_PARAMS, "fieldsplit_1": PLAIN_GMRES_PARAMS}
FIELDSPLIT_GMRES_ILU_PARAMS: dict = {**_FIELDSPLIT_BASE, "fieldsplit_0": GMRES_ILU_PARAMS, "fieldsplit_1": GMRES_ILU_PARAMS}

RICHARDSON_SOLVER_PARAMS: dict = {
    "snes_type": "nrichardson",
    "snes_max_it": _MAX_ITERATION_NUMBER,
    "snes_linesearch_type": "basic",
    "snes_linesearch_damping": 0.5,
    "snes_rtol": 1e-5,
    "snes_atol": 1e-12,
    **FIELDSPLIT_LU_PARAMS,
}

_PICARD_BASE = {"snes_type": "ngs", "snes_max_it": _MAX_ITERATION_NUMBER, "snes_rtol": 1e-8, "snes_atol": 1e-12}
PICARD_LU_SOLVER_PARAMS = {**_PICARD_BASE, **FIELDSPLIT_LU_PARAMS}
PICARD_GMRES_SOLVER_PARAMS = {**_PICARD_BASE, **FIELDSPLIT_GMRES_PARAMS}
PICARD_GMRES_ILU_SOLVER_PARAMS = {**_PICARD_BASE, **FIELDSPLIT_GMRES_ILU_PARAMS}

KSP_PREONLY_PARAMS: dict = {"snes_type": "ksponly", "ksp_monitor": None, **FIELDSPLIT_LU_PARAMS}

# ---- MI355X-path presets (no reference counterpart) --------------------------------------------
# CG on the SPD monolithic system with the 2x2 node-block Jacobi preconditioner
CG_BLOCK_JACOBI_PARAMS: dict = {"ksp_type": "cg", "pc_type": "pph_block2", "ksp_rtol": 1.0e-8, "ksp_atol": 1.0e-12,
                                "ksp_max_it": _MAX_ITERATION_NUMBER}
CG_JACOBI_PARAMS: dict = {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1.0e-8, "ksp_atol": 1.0e-12,
                          "ksp_max_it": _MAX_ITERATION_NUMBER}
# field-split GMRES whose block solves are multigrid-preconditioned CG
FIELDSPLIT_MG_PARAMS: dict = {
    **GMRES_PARAMS, **_FIELDSPLIT_BASE,
    "fieldsplit_0": {"ksp_type": "cg", "pc_type": "mg", "ksp_rtol": 1e-10},
    "fieldsplit_1": {"ksp_type": "cg", "pc_type": "mg", "ksp_rtol": 1e-10},
}
# block Picard (fixed-stress) sweeps with multigrid-preconditioned CG block solves
PICARD_MG_SOLVER_PARAMS: dict = {
    **_PICARD_BASE, **_FIELDSPLIT_BASE,
    "fieldsplit_0": {"ksp_type": "cg", "pc_type": "mg", "ksp_rtol": 1e-10},
    "fieldsplit_1": {"ksp_type": "cg", "pc_type": "mg", "ksp_rtol": 1e-10},
}
# the benchmark's algorithm (bench.py): inexact sweeps - every block solve stops once its unpreconditioned residual
# has dropped tenfold from its own start (warm starts make the sequence converge to the exact fixed point),
# V(1,1) cycles; the outer criterion (snes_rtol on the true residual) is unchanged
_INEXACT_BLOCK = {"ksp_type": "cg", "pc_type": "mg", "ksp_rtol": 1e-10, "ksp_norm_type": "unpreconditioned",
                  "pph_reduction": 0.1}
PICARD_MG_INEXACT_SOLVER_PARAMS: dict = {
    **_PICARD_BASE, **_FIELDSPLIT_BASE, "pph_mg_smooth": 1,
    "fieldsplit_0": dict(_INEXACT_BLOCK), "fieldsplit_1": dict(_INEXACT_BLOCK),
}
# the launch-only variant: block solves of exactly one multigrid-preconditioned CG iteration, no inner convergence
# test (PETSc: ksp_norm_type none + ksp_max_it 1) - a sweep holds no host decision and is replayed from a hipGraph;
# fastest where kernels are short (<= 128^3), 8 instead of 6 sweeps at the same number of CG iterations
_FIXED_BLOCK = {"ksp_type": "cg", "pc_type": "mg", "ksp_norm_type": "none", "ksp_max_it": 1}
PICARD_MG_FIXED_SOLVER_PARAMS: dict = {
    **_PICARD_BASE, **_FIELDSPLIT_BASE, "pph_mg_smooth": 1,
    "fieldsplit_0": dict(_FIXED_BLOCK), "fieldsplit_1": dict(_FIXED_BLOCK),
}
PICARD_JACOBI_SOLVER_PARAMS: dict = {
    **_PICARD_BASE, **_FIELDSPLIT_BASE,
    "fieldsplit_0": {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10},
    "fieldsplit_1": {"ksp_type": "cg", "pc_type": "jacobi", "ksp_rtol": 1e-10},
}
