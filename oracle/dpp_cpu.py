"""
ORACLE — test infrastructure only.  NOT part of the product path.

ctypes loader of ``oracle/libdpp_cpu.so`` (C99 + OpenMP restatement in ``oracle/dpp_cpu.c``; built by
``make -C oracle`` / ``__graft_entry__.build()``).  Used by tests/ and by bench.py's ``cpu_baseline`` leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libdpp_cpu.so")

MAT_K, MAT_M, MAT_A11, MAT_A22, MAT_A12, MAT_A21 = 1, 2, 3, 4, 5, 6
PC_NONE, PC_JACOBI, PC_MG = 0, 1, 4


def build() -> str:
    subprocess.run(["make", "-C", _HERE, "--no-print-directory"], check=True, stdout=subprocess.DEVNULL)
    return _LIB


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    p, i64, i32, f64 = C.c_void_p, C.c_int64, C.c_int, C.c_double
    lib.dppc_create.restype = p
    lib.dppc_create.argtypes = [i32] * 5
    lib.dppc_destroy.argtypes = [p]
    lib.dppc_destroy.restype = None
    lib.dppc_sizes.argtypes = [p, p, p, p, p]
    lib.dppc_sizes.restype = None
    lib.dppc_get_mesh.argtypes = [p, p, p]
    lib.dppc_get_mesh.restype = None
    lib.dppc_set_dirichlet.argtypes = [p, i32, p, p, i64]
    lib.dppc_assemble.argtypes = [p, f64, f64, f64, f64]
    lib.dppc_mg_setup.argtypes = [p, i32]
    lib.dppc_get_csr.argtypes = [p, i32, p, p, p]
    lib.dppc_get_rhs.argtypes = [p, p, p]
    lib.dppc_get_rhs.restype = None
    lib.dppc_spmv.argtypes = [p, i32, p, p]
    lib.dppc_spmv_bench.argtypes = [p, i32, i32]
    lib.dppc_spmv_bench.restype = f64
    lib.dppc_vcycle.argtypes = [p, i32, p, p, i32]
    lib.dppc_vcycle.restype = None
    lib.dppc_pcg.argtypes = [p, i32, i32, p, p, i32, f64, f64, i32, f64, i32, i32, p]
    lib.dppc_picard.argtypes = [p, i32, f64, f64, i32, f64, i32, i32, f64, f64, i32, p, p, p]
    lib.dppc_num_threads.restype = i32
    lib.dppc_set_threads.argtypes = [i32]
    lib.dppc_set_threads.restype = None
    return lib


lib = _load()
# a container may expose every core of its host while granting a small share: never start more than 16 threads
# unless the caller asks for them (bench.py reads the cgroup quota and calls set_threads)
lib.dppc_set_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class CpuSystem:
    """One DPP problem on the unit square / cube held by the C restatement."""

    def __init__(self, dim: int, kind: int, nx: int, ny: int, nz: int = 0):
        self._h = lib.dppc_create(dim, kind, nx, ny, nz)
        if not self._h:
            raise ValueError("invalid mesh arguments")
        n, nc, nnz, m = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        lib.dppc_sizes(self._h, C.byref(n), C.byref(nc), C.byref(nnz), C.byref(m))
        self.dim, self.n, self.ncell, self.nnz, self.m = dim, n.value, nc.value, nnz.value, m.value

    def close(self):
        if self._h:
            lib.dppc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def mesh(self):
        cells = np.empty((self.ncell, self.m), np.int32)
        coords = np.empty((self.n, self.dim), np.float64)
        lib.dppc_get_mesh(self._h, _ptr(cells), _ptr(coords))
        return cells, coords

    def set_dirichlet(self, field: int, nodes, vals):
        nodes = np.ascontiguousarray(nodes, np.int64)
        vals = np.ascontiguousarray(vals, np.float64)
        if lib.dppc_set_dirichlet(self._h, field, _ptr(nodes), _ptr(vals), nodes.size):
            raise ValueError("bad Dirichlet data")

    def assemble(self, k1, k2, beta, mu):
        if lib.dppc_assemble(self._h, k1, k2, beta, mu):
            raise MemoryError("assembly failed")

    def mg_setup(self, min_cells: int = 2) -> int:
        nlev = lib.dppc_mg_setup(self._h, min_cells)
        if nlev < 0:
            raise RuntimeError("mg_setup before assemble")
        return nlev

    def csr(self, which: int) -> sp.csr_matrix:
        rp = np.empty(self.n + 1, np.int64)
        col = np.empty(self.nnz, np.int32)
        val = np.empty(self.nnz, np.float64)
        if lib.dppc_get_csr(self._h, which, _ptr(rp), _ptr(col), _ptr(val)):
            raise ValueError("unknown matrix id")
        return sp.csr_matrix((val, col, rp), shape=(self.n, self.n))

    def rhs(self):
        r, u0 = np.empty(2 * self.n), np.empty(2 * self.n)
        lib.dppc_get_rhs(self._h, _ptr(r), _ptr(u0))
        return r, u0

    def spmv(self, which: int, x):
        x = np.ascontiguousarray(x, np.float64)
        y = np.empty_like(x)
        lib.dppc_spmv(self._h, which, _ptr(x), _ptr(y))
        return y

    def spmv_seconds(self, which: int, reps: int) -> float:
        return lib.dppc_spmv_bench(self._h, which, reps)

    def vcycle(self, which: int, r, smooth: int):
        r = np.ascontiguousarray(r, np.float64)
        z = np.empty_like(r)
        lib.dppc_vcycle(self._h, which, _ptr(r), _ptr(z), smooth)
        return z

    def pcg(self, which, pc, b, x0=None, rtol=1e-8, atol=1e-12, max_it=50000, reduction=0.0, smooth=2, norm=0):
        b = np.ascontiguousarray(b, np.float64)
        x = np.zeros_like(b) if x0 is None else np.array(x0, np.float64)
        res = C.c_double()
        its = lib.dppc_pcg(self._h, which, pc, _ptr(b), _ptr(x), int(x0 is not None), rtol, atol, max_it, reduction, smooth,
                           norm, C.byref(res))
        return x, its, res.value

    def picard(self, pc=PC_MG, inner_rtol=1e-10, inner_atol=1e-300, inner_max_it=50000, reduction=1e-2, smooth=1,
               rtol=1e-8, atol=1e-12, max_it=100, inner_norm=0):
        x = np.empty(2 * self.n)
        inner, res = C.c_int(), C.c_double()
        sweeps = lib.dppc_picard(self._h, pc, inner_rtol, inner_atol, inner_max_it, reduction, smooth, inner_norm, rtol, atol,
                                 max_it, _ptr(x), C.byref(inner), C.byref(res))
        return x, sweeps, inner.value, res.value


def num_threads() -> int:
    return lib.dppc_num_threads()


def set_threads(t: int) -> None:
    lib.dppc_set_threads(int(t))
