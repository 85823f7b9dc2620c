"""
ctypes binding of ``libperphil_hip.so`` (C ABI declared in ``include/perphil_hip.h``).

This is the only crossing between the Python host layer and the device code: Python -> ctypes ->
C ABI -> HIP.  There is no CPU fallback: if the shared library is missing or cannot be loaded the
import of this module raises, and every call that fails inside the library raises
(``ValueError`` for PPH_ERR_INVALID, ``MemoryError`` for PPH_ERR_NOMEM, ``RuntimeError``
otherwise), mirroring how PETSc errors surface as exceptions from the reference's
``solver.solve()`` (reference ``src/perphil/solvers/solver.py:71``).
"""
from __future__ import annotations

import sys
import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PERPHIL_HIP_LIB") or os.path.join(_HERE, "libperphil_hip.so")   # (the override: A/B builds of tools/)

PPH_OK, PPH_ERR_INVALID, PPH_ERR_HIP, PPH_ERR_NOMEM, PPH_ERR_DIVERGED, PPH_ERR_COMM = 0, -1, -2, -3, -4, -5
CELL_QUAD, CELL_TRI, CELL_HEX, CELL_TET = 0, 1, 2, 3
KSP_PREONLY, KSP_CG, KSP_GMRES = 0, 1, 2
PC_NONE, PC_JACOBI, PC_BLOCK2, PC_FIELDSPLIT, PC_MG, PC_ILU = 0, 1, 2, 3, 4, 5
MAT_MONO, MAT_K, MAT_M, MAT_A11, MAT_A22, MAT_A12, MAT_A21 = 0, 1, 2, 3, 4, 5, 6

# every symbol include/perphil_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "pph_ctx_create", "pph_ctx_destroy", "pph_last_error", "pph_ctx_synchronize",
    "pph_mesh_build", "pph_mesh_sizes", "pph_get_dofmap", "pph_get_coords",
    "pph_set_dirichlet", "pph_assemble_dpp",
    "pph_solve", "pph_solve_device", "pph_get_solution", "pph_host_alloc", "pph_host_free",
    "pph_csr_sizes", "pph_get_csr", "pph_get_rhs", "pph_spmv", "pph_spmv_bench",
    "pph_get_timers", "pph_set_option", "pph_comm_set_callbacks",
    "pph_rccl_available", "pph_rccl_unique_id", "pph_comm_init_rccl", "pph_comm_selftest", "pph_comm_selftest2",
    "pph_comm_stats", "pph_comm_times", "pph_error_norms_mms", "pph_quadrature_points", "pph_error_norms_sampled", "pph_bw_probe",
    "pph_darcy_velocity",
]

HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)


class _PinnedBlock:
    """Owner of one pph_host_alloc block; freed when the last NumPy array over it is gone."""

    def __init__(self, ptr: int):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                lib.pph_host_free(C.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:
            pass


def _pinned_array(n: int):
    """float64[n] over page-locked memory, or None when the allocation fails."""
    out = C.c_void_p()
    if lib.pph_host_alloc(C.c_size_t(8 * n), C.byref(out)) != 0 or not out.value:
        return None
    cbuf = (C.c_double * n).from_address(out.value)
    cbuf._owner = _PinnedBlock(out.value)     # the ctypes array is the base of every NumPy view: it keeps the block alive
    return np.frombuffer(cbuf, dtype=np.float64, count=n)


class SolverCfg(C.Structure):
    """``pph_solver_cfg``"""
    _fields_ = [
        ("ksp_type", C.c_int32), ("pc_type", C.c_int32), ("restart", C.c_int32), ("max_it", C.c_int32),
        ("rtol", C.c_double), ("atol", C.c_double),
        ("inner_ksp_type", C.c_int32), ("inner_pc_type", C.c_int32), ("inner_max_it", C.c_int32),
        ("picard", C.c_int32),
        ("inner_rtol", C.c_double), ("inner_atol", C.c_double),
        ("picard_rtol", C.c_double), ("picard_atol", C.c_double),
        ("picard_max_it", C.c_int32), ("mg_smooth", C.c_int32),
        ("inner_reduction", C.c_double),
        ("inner_norm", C.c_int32), ("inner_exact", C.c_int32),
    ]


class SolveInfo(C.Structure):
    """``pph_solve_info``"""
    _fields_ = [
        ("iterations", C.c_int32), ("inner_iterations", C.c_int32), ("converged", C.c_int32),
        ("inner_failed", C.c_int32), ("resnorm", C.c_double), ("rhs_norm", C.c_double),
    ]


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so / libhsa-runtime64.so
    (same SONAMEs as /opt/rocm's, loaded by path): if this library binds the system copy first and torch
    is imported later, two runtimes coexist and the second one finds no GPU.  So when torch is installed
    and not yet imported, its copy is loaded first (RTLD_GLOBAL) and libperphil_hip.so resolves its
    libamdhip64.so.7 dependency to it.  PERPHIL_HIP_RUNTIME=system skips this."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("PERPHIL_HIP_RUNTIME", "") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def _load() -> C.CDLL:
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C perphil_amd/csrc`).  perphil_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    p = C.c_void_p
    i32p, i64p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
    sig = {
        "pph_ctx_create": ([C.c_int, C.POINTER(p)], C.c_int),
        "pph_ctx_destroy": ([p], C.c_int),
        "pph_last_error": ([p], C.c_char_p),
        "pph_ctx_synchronize": ([p], C.c_int),
        "pph_mesh_build": ([p] + [C.c_int] * 9, C.c_int),
        "pph_mesh_sizes": ([p, i64p, i64p, i32p, i64p], C.c_int),
        "pph_get_dofmap": ([p, C.c_void_p], C.c_int),
        "pph_get_coords": ([p, C.c_void_p], C.c_int),
        "pph_set_dirichlet": ([p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64], C.c_int),
        "pph_assemble_dpp": ([p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int], C.c_int),
        "pph_solve": ([p, C.POINTER(SolverCfg), C.c_void_p, C.POINTER(SolveInfo), C.c_void_p, C.c_int], C.c_int),
        "pph_solve_device": ([p, C.POINTER(SolverCfg), C.POINTER(SolveInfo), C.c_void_p, C.c_int], C.c_int),
        "pph_get_solution": ([p, C.c_void_p], C.c_int),
        "pph_host_alloc": ([C.c_size_t, C.POINTER(C.c_void_p)], C.c_int),
        "pph_host_free": ([C.c_void_p], C.c_int),
        "pph_csr_sizes": ([p, C.c_int, i64p, i64p], C.c_int),
        "pph_get_csr": ([p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
        "pph_get_rhs": ([p, C.c_void_p, C.c_void_p], C.c_int),
        "pph_spmv": ([p, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
        "pph_spmv_bench": ([p, C.c_int, C.c_int, f64p], C.c_int),
        "pph_get_timers": ([p, C.c_void_p, C.c_int], C.c_int),
        "pph_set_option": ([p, C.c_char_p, C.c_double], C.c_int),
        "pph_comm_set_callbacks": ([p, C.c_int, C.c_int, HALO_FN, ALLREDUCE_FN, C.c_void_p], C.c_int),
        "pph_rccl_unique_id": ([C.c_char_p, C.c_void_p], C.c_int),
        "pph_comm_init_rccl": ([p, C.c_int, C.c_int, C.c_void_p, C.c_char_p], C.c_int),
        "pph_comm_selftest": ([p], C.c_int),
        "pph_comm_selftest2": ([p, C.POINTER(C.c_int)], C.c_int),
        "pph_comm_stats": ([p, i64p, i64p, C.POINTER(C.c_int)], C.c_int),
        "pph_comm_times": ([p, f64p], C.c_int),
        "pph_rccl_available": ([C.c_char_p], C.c_int),
        "pph_bw_probe": ([p, C.c_int64, C.c_int, C.c_int, f64p], C.c_int),
        "pph_error_norms_mms": ([p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, f64p,
                                 f64p], C.c_int),
        "pph_darcy_velocity": ([p, C.c_void_p, C.c_double, C.c_void_p], C.c_int),
        "pph_quadrature_points": ([p, C.c_int, C.c_int64, C.c_int64, C.c_void_p], C.c_int),
        "pph_error_norms_sampled": ([p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, f64p, f64p], C.c_int),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


lib = _load()


class ConvergenceError(RuntimeError):
    """Krylov / Picard iteration did not converge (Firedrake raises ConvergenceError likewise)."""


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """Owner of one ``pph_ctx`` (one GPU, one stream).  Not thread-safe; one per device."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        st = lib.pph_ctx_create(int(device), C.byref(self._h))
        if st != PPH_OK:
            msg = (lib.pph_last_error(None) or b"").decode()
            raise RuntimeError(f"pph_ctx_create(device={device}) failed ({st}): {msg}")
        self.device = int(device)
        self.n = 0
        self.dim = 0
        self._bc_state = {}

    # -- plumbing -----------------------------------------------------------------------------
    def _check(self, st: int, allow_diverged: bool = False) -> int:
        if st == PPH_OK or (allow_diverged and st == PPH_ERR_DIVERGED):
            return st
        msg = (lib.pph_last_error(self._h) or b"").decode()
        if st == PPH_ERR_INVALID:
            raise ValueError(msg)
        if st == PPH_ERR_NOMEM:
            raise MemoryError(msg)
        if st == PPH_ERR_DIVERGED:
            raise ConvergenceError(msg)
        raise RuntimeError(f"libperphil_hip error {st}: {msg}")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            lib.pph_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self) -> None:
        self._check(lib.pph_ctx_synchronize(self._h))

    def set_option(self, name: str, value: float) -> None:
        self._check(lib.pph_set_option(self._h, name.encode(), float(value)))

    # -- mesh ---------------------------------------------------------------------------------
    def mesh_build(self, dim: int, kind: int, nx: int, ny: int, nz: int = 0, z_begin: int = 0,
                   z_count: Optional[int] = None, ghost_lo: bool = False, ghost_hi: bool = False) -> None:
        if z_count is None:
            z_count = nz
        self._check(lib.pph_mesh_build(self._h, dim, kind, nx, ny, nz, z_begin, z_count, int(ghost_lo), int(ghost_hi)))
        n, nc, m, nnz = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int64()
        self._check(lib.pph_mesh_sizes(self._h, C.byref(n), C.byref(nc), C.byref(m), C.byref(nnz)))
        self.n, self.ncell, self.m, self.nnzb, self.dim = n.value, nc.value, m.value, nnz.value, dim
        self._bc_state = {}

    def dofmap(self) -> np.ndarray:
        out = np.empty((self.ncell, self.m), dtype=np.int32)
        self._check(lib.pph_get_dofmap(self._h, _ptr(out)))
        return out

    def coords(self) -> np.ndarray:
        out = np.empty((self.n, self.dim), dtype=np.float64)
        self._check(lib.pph_get_coords(self._h, _ptr(out)))
        return out

    # -- system -------------------------------------------------------------------------------
    def set_dirichlet(self, field: int, nodes: np.ndarray, vals: np.ndarray) -> None:
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        if nodes.shape != vals.shape:
            raise ValueError("nodes and vals must have the same shape")
        self._check(lib.pph_set_dirichlet(self._h, int(field), _ptr(nodes), _ptr(vals), nodes.size))
        self._bc_state[int(field)] = (nodes.copy(), vals.copy())

    def same_dirichlet(self, field: int, nodes: np.ndarray, vals: np.ndarray) -> bool:
        """True when exactly this set (nodes and values) is what the device holds for `field` (callers skip the
        upload then: pph_set_dirichlet drops the assembled system and makes the multigrid levels re-derive their masks)."""
        old = self._bc_state.get(int(field))
        return (old is not None and np.array_equal(old[0], np.asarray(nodes, dtype=np.int64))
                and np.array_equal(old[1], np.asarray(vals, dtype=np.float64)))

    def assemble(self, k1: float, k2: float, beta: float, mu: float, monolithic: bool = True) -> None:
        self._check(lib.pph_assemble_dpp(self._h, float(k1), float(k2), float(beta), float(mu), int(monolithic)))

    # -- result vectors ---------------------------------------------------------------------
    # Large results land in page-locked host arrays: the device-to-host copy then runs at PCIe speed and no fresh pages
    # are touched (256^3: 5 - 6 ms instead of 15 - 27 for the 272 MB solution).  The arrays are ordinary NumPy arrays whose
    # memory belongs to a _PinnedBlock that frees it when the last array / view of it dies; a pool of up to three blocks
    # per context hands out a block again once nobody but the pool refers to its array (a loop that overwrites `sol` each
    # time alternates between two blocks); a caller that keeps more than three results alive gets pageable arrays.
    _PIN_MIN = 1 << 20      # entries; smaller results use pageable arrays
    _PIN_POOL = 3

    @staticmethod
    def _idle_refcount() -> int:
        # what sys.getrefcount reports for an object only a list refers to, counted the way _result_array counts (list entry,
        # loop variable, the call's argument): measured on this interpreter instead of assumed (3 on CPython 3.10)
        for o in [object()]:
            return sys.getrefcount(o)
        return 3

    def _result_array(self, n: int) -> np.ndarray:
        if n < self._PIN_MIN:
            return np.empty(n, dtype=np.float64)
        pool = self.__dict__.setdefault("_pinned", [])
        idle = self.__dict__.setdefault("_pinned_idle", self._idle_refcount())
        for arr in pool:
            if arr.shape[0] == n and sys.getrefcount(arr) <= idle:    # nobody but the pool refers to it (views count: their base is arr)
                return arr
        pool[:] = [a for a in pool if a.shape[0] == n]
        if len(pool) >= self._PIN_POOL:
            return np.empty(n, dtype=np.float64)
        arr = _pinned_array(n)
        if arr is None:
            return np.empty(n, dtype=np.float64)
        pool.append(arr)
        if len(pool) == 1:
            # the usual caller overwrites its previous result with the next: that needs two blocks - pin both now (17 ms
            # each at 272 MB) rather than inside the second, typically timed, call
            second = _pinned_array(n)
            if second is not None:
                pool.append(second)
        return arr

    def solve(self, cfg: SolverCfg, fetch: bool = True, hist_cap: int = 0, raise_on_diverged: bool = True):
        info = SolveInfo()
        hist = np.zeros(max(hist_cap, 1), dtype=np.float64)
        x = self._result_array(2 * self.n) if fetch else None
        if fetch:
            st = lib.pph_solve(self._h, C.byref(cfg), _ptr(x), C.byref(info), _ptr(hist), int(hist_cap))
        else:
            st = lib.pph_solve_device(self._h, C.byref(cfg), C.byref(info), _ptr(hist), int(hist_cap))
        self._check(st, allow_diverged=not raise_on_diverged)
        nh = min(hist_cap, info.iterations + 1)
        return x, info, hist[:nh].copy()

    def solution(self) -> np.ndarray:
        x = self._result_array(2 * self.n)
        self._check(lib.pph_get_solution(self._h, _ptr(x)))
        return x

    # -- export -------------------------------------------------------------------------------
    def csr(self, which: int):
        import scipy.sparse as sp

        nrows, nnz = C.c_int64(), C.c_int64()
        self._check(lib.pph_csr_sizes(self._h, which, C.byref(nrows), C.byref(nnz)))
        rowptr = np.empty(nrows.value + 1, dtype=np.int64)
        col = np.empty(nnz.value, dtype=np.int32)
        val = np.empty(nnz.value, dtype=np.float64)
        self._check(lib.pph_get_csr(self._h, which, _ptr(rowptr), _ptr(col), _ptr(val)))
        return sp.csr_matrix((val, col, rowptr), shape=(nrows.value, nrows.value))

    def rhs(self):
        r = np.empty(2 * self.n, dtype=np.float64)
        u0 = np.empty(2 * self.n, dtype=np.float64)
        self._check(lib.pph_get_rhs(self._h, _ptr(r), _ptr(u0)))
        return r, u0

    def spmv(self, which: int, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._check(lib.pph_spmv(self._h, which, _ptr(x), _ptr(y)))
        return y

    def spmv_bench(self, which: int, reps: int) -> float:
        ms = C.c_double()
        self._check(lib.pph_spmv_bench(self._h, which, int(reps), C.byref(ms)))
        return ms.value

    def error_norms_mms(self, field: int, nodal: np.ndarray, k1: float, k2: float, beta: float, mu: float, nq: int = 6):
        """(L2 error, H1-seminorm error) of a nodal CG-1 field against the manufactured pressure `field`."""
        nodal = np.ascontiguousarray(nodal, dtype=np.float64)
        if nodal.shape != (self.n,):
            raise ValueError("nodal array must have one value per mesh vertex")
        l2, h1 = C.c_double(), C.c_double()
        self._check(lib.pph_error_norms_mms(self._h, int(field), _ptr(nodal), float(k1), float(k2), float(beta), float(mu),
                                            int(nq), C.byref(l2), C.byref(h1)))
        return l2.value, h1.value

    def quadrature_points(self, nq: int, cell_begin: int, cell_count: int) -> np.ndarray:
        """Physical coordinates [cell_count * nq**dim, dim] of the Gauss points of a cell range (pph_quadrature_points)."""
        npts = nq ** self.dim
        out = np.empty((cell_count * npts, self.dim), dtype=np.float64)
        self._check(lib.pph_quadrature_points(self._h, int(nq), int(cell_begin), int(cell_count), _ptr(out)))
        return out

    def error_norms_sampled(self, nodal: np.ndarray, exact, grad=None, nq: int = 6, chunk_cells: int = 1 << 16):
        """(L2 error, H1-seminorm error) of a nodal CG-1 field against `exact`, a callable of point arrays [m, dim] -> [m]
        (None: the zero field), with `grad` its gradient callable [m, dim] -> [m, dim] (None: central differences of
        `exact` with step 1e-6).  The cells are processed in chunks: points out, samples in."""
        nodal = np.ascontiguousarray(nodal, dtype=np.float64)
        if nodal.shape != (self.n,):
            raise ValueError("nodal array must have one value per mesh vertex")
        l2, h1 = 0.0, 0.0
        a, b = C.c_double(), C.c_double()
        if exact is None:
            chunk_cells = self.ncell      # no host samples: one call over all cells
        first = True
        for c0 in range(0, self.ncell, chunk_cells):
            cnt = min(chunk_cells, self.ncell - c0)
            se = sg = None
            if exact is not None:
                X = self.quadrature_points(nq, c0, cnt)
                se = np.ascontiguousarray(exact(X), dtype=np.float64).reshape(-1)
                if grad is not None:
                    sg = np.ascontiguousarray(grad(X), dtype=np.float64).reshape(-1, self.dim)
                else:
                    h = 1e-6
                    sg = np.empty_like(X)
                    for d in range(self.dim):
                        E = np.zeros(self.dim)
                        E[d] = h
                        sg[:, d] = (np.asarray(exact(X + E), dtype=np.float64).reshape(-1)
                                    - np.asarray(exact(X - E), dtype=np.float64).reshape(-1)) / (2 * h)
                    sg = np.ascontiguousarray(sg)
            # the nodal field travels with the first chunk only (NULL afterwards = the field of the previous call)
            self._check(lib.pph_error_norms_sampled(self._h, _ptr(nodal) if first else None, int(nq), int(c0), int(cnt),
                                                    _ptr(se), _ptr(sg), C.byref(a), C.byref(b)))
            first = False
            l2 += a.value
            h1 += b.value
        return float(np.sqrt(l2)), float(np.sqrt(h1))

    def darcy_velocity(self, nodal: np.ndarray, conductivity: float) -> np.ndarray:
        """L2 projection of -conductivity * grad(p_h) onto CG-1 vectors; returns [n, dim]."""
        nodal = np.ascontiguousarray(nodal, dtype=np.float64)
        if nodal.shape != (self.n,):
            raise ValueError("nodal array must have one value per mesh vertex")
        out = np.empty((self.n, self.dim), dtype=np.float64)
        self._check(lib.pph_darcy_velocity(self._h, _ptr(nodal), float(conductivity), _ptr(out)))
        return out

    def comm_stats(self) -> dict:
        h, a, st = C.c_int64(), C.c_int64(), C.c_int()
        self._check(lib.pph_comm_stats(self._h, C.byref(h), C.byref(a), C.byref(st)))
        return {"halo_exchanges": h.value, "allreduces": a.value, "status": st.value}

    def comm_times(self) -> dict:
        """Summed durations of the last solve's exchanges / reductions (option time_comm)."""
        t = (C.c_double * 4)()
        self._check(lib.pph_comm_times(self._h, t))
        return {"halo_ms": t[0], "allreduce_ms": t[1], "halo_timed": int(t[2]), "allreduce_timed": int(t[3])}

    def timers(self) -> dict:
        t = np.zeros(23, dtype=np.float64)
        self._check(lib.pph_get_timers(self._h, _ptr(t), 23))
        return {"mesh_ms": t[0], "assemble_ms": t[1], "bc_blocks_ms": t[2], "solve_ms": t[3],
                "spmv_ms": t[4], "spmv_launches": int(t[5]), "spmv_bytes": t[6],
                "spmv_dot_ms": t[7], "spmv_dot_launches": int(t[8]), "spmv_dot_bytes": t[9],
                "halo_exchanges": int(t[10]),
                "spmv_fine_ms": t[11], "spmv_fine_launches": int(t[12]), "spmv_fine_bytes": t[13],
                "split_products": int(t[14]), "symmetric_storage": bool(t[15]),
                "max_split_partials": int(t[16]),
                "dict_operators": int(t[17]), "dict_classes": int(t[18]), "dict_status": int(t[19]),
                "dict_build_ms": t[20], "dict_builds": int(t[21]), "dict_zconst": bool(t[22])}
