"""
Multi-GPU driver of the DPP hot path: one process per GPU, one cell slab per process
(perphil_amd/partition.py), halo planes and scalar all-reduces through ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the 8-GPU node; "gloo" in the tests, where all ranks may share one
GPU).  The library calls back into this module for communication (include/perphil_hip.h,
``pph_comm_set_callbacks``); the numerical kernels and the solver loops are the single-GPU ones.

The reference never runs in parallel (every recorded run is one process); what this
replaces is the halo VecScatter / VecDot all-reduce PETSc would do under mpiexec.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _ffi
from .partition import Slab, make_slab

_HALO_FN = _ffi.HALO_FN
_ALLREDUCE_FN = _ffi.ALLREDUCE_FN


def _loaded_rccl_path() -> bytes:
    """Path of the librccl the process already has (torch's bundled copy), else empty (library default)."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split()[-1].encode()
    except OSError:
        pass
    return b""


def _agree(ok: bool, group=None) -> bool:
    """Logical AND of a per-rank flag over the torch.distributed group (any backend)."""
    import torch
    import torch.distributed as dist

    flag = torch.tensor([1.0 if ok else 0.0])
    if dist.get_backend(group) == "nccl":
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() == 1.0)


def init_rccl(ctx: _ffi.Context, group=None) -> int:
    """Create the context's RCCL communicator and run the library's self-test; returns the world size the
    self-test's all-reduce observed.  Collective over the torch.distributed group, and written so that a failure
    on ONE rank can never leave the others waiting inside an RCCL call:

    1. every rank checks that librccl loads (``pph_rccl_available``) and the ranks agree on that (torch-level MIN)
       BEFORE anyone enters ``ncclCommInitRank``;
    2. rank 0 draws the unique id; a failed draw travels as an all-zero id in the same broadcast;
    3. every rank joins the communicator and the ranks agree that all of them did; only then they run the
       straight-line self-test (all three phases are always issued, failures are reported afterwards) and agree on
       its verdict.

    Raises RuntimeError on every rank when any rank failed."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    path = _loaded_rccl_path()
    loadable = _ffi.lib.pph_rccl_available(path) == _ffi.PPH_OK
    if not _agree(loadable, group):
        raise RuntimeError("librccl is not loadable on every rank" + ("" if loadable else f" (rank {rank}: missing)"))
    ident = np.zeros(128, dtype=np.uint8)
    if rank == 0:
        if _ffi.lib.pph_rccl_unique_id(path, ident.ctypes.data_as(C.c_void_p)) != _ffi.PPH_OK:
            ident[:] = 0
    t = torch.from_numpy(ident)
    if dist.get_backend(group) == "nccl":
        d = t.cuda()
        dist.broadcast(d, 0, group=group)
        ident = d.cpu().numpy()
    else:
        dist.broadcast(t, 0, group=group)
    ident = np.ascontiguousarray(ident)
    if not ident.any():   # identical on every rank: all of them leave here together
        raise RuntimeError("rank 0 could not draw an RCCL unique id")
    st = _ffi.lib.pph_comm_init_rccl(ctx._h, rank, world, ident.ctypes.data_as(C.c_void_p), path)
    # the self-test talks to the neighbours: nobody enters it unless EVERY rank joined the communicator (a rank that
    # skipped it alone would leave its peers waiting inside ncclSend/ncclRecv for ever)
    if not _agree(st == _ffi.PPH_OK, group):
        msg = (_ffi.lib.pph_last_error(ctx._h) or b"").decode() if st != _ffi.PPH_OK else "ok"
        raise RuntimeError(f"ncclCommInitRank failed on at least one rank (rank {rank}: {msg})")
    seen = C.c_int(0)
    st = _ffi.lib.pph_comm_selftest2(ctx._h, C.byref(seen))
    msg = (_ffi.lib.pph_last_error(ctx._h) or b"").decode() if st != _ffi.PPH_OK else ""
    if not _agree(st == _ffi.PPH_OK and seen.value == world, group):
        raise RuntimeError(f"RCCL transport failed its self-test on at least one rank (rank {rank}: {msg or 'ok'})")
    return int(seen.value)


class _DevView:
    """Zero-copy view of `count` doubles at a raw device address (CUDA array interface v3)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3}


class Communicator:
    """Halo exchange + all-reduce over a torch.distributed process group."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.on_device = self.backend == "nccl"
        self.halo_calls = 0
        self.allreduce_calls = 0
        self.fail_halo_after = -1      # tests: the halo callback reports a failure from this call on (every rank alike)
        self._halo_c = _HALO_FN(self._halo)
        self._allreduce_c = _ALLREDUCE_FN(self._allreduce)

    def _dev(self, ptr: int, off: int, count: int):
        return self.torch.as_tensor(_DevView(ptr + 8 * off, count), device="cuda")

    def _halo(self, _user, vec, plane, send_lo, recv_lo, send_hi, recv_hi) -> int:
        if 0 <= self.fail_halo_after <= self.halo_calls:
            return -1
        try:
            torch, dist = self.torch, self.dist
            ops, copies = [], []
            for send, recv, peer in ((send_lo, recv_lo, self.rank - 1), (send_hi, recv_hi, self.rank + 1)):
                if send < 0:
                    continue
                s, r = self._dev(vec, send, plane), self._dev(vec, recv, plane)
                if self.on_device:
                    ops += [dist.P2POp(dist.isend, s, peer, self.group), dist.P2POp(dist.irecv, r, peer, self.group)]
                else:
                    sh, rh = s.cpu(), torch.empty(plane, dtype=torch.float64)
                    ops += [dist.P2POp(dist.isend, sh, peer, self.group), dist.P2POp(dist.irecv, rh, peer, self.group)]
                    copies.append((r, rh))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for r, rh in copies:
                r.copy_(rh)
            torch.cuda.synchronize()
            self.halo_calls += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print(f"[perphil_amd.distributed] halo exchange failed on rank {self.rank}: {e!r}", flush=True)
            return -1

    def _allreduce(self, _user, vals, count) -> int:
        try:
            torch, dist = self.torch, self.dist
            a = np.ctypeslib.as_array(C.cast(vals, C.POINTER(C.c_double)), shape=(count,))
            t = torch.from_numpy(a)
            if self.on_device:
                d = t.cuda()
                dist.all_reduce(d, group=self.group)
                a[:] = d.cpu().numpy()
            else:
                dist.all_reduce(t, group=self.group)
            self.allreduce_calls += 1
            return 0
        except Exception as e:
            print(f"[perphil_amd.distributed] all-reduce failed on rank {self.rank}: {e!r}", flush=True)
            return -1

    def attach(self, ctx: _ffi.Context) -> None:
        ctx._check(_ffi.lib.pph_comm_set_callbacks(ctx._h, self.rank, self.world, self._halo_c, self._allreduce_c, None))
        ctx._comm = self  # keep the callbacks alive as long as the context


def default_device() -> int:
    """Device of this rank: torch's current device when torch already selected one, else LOCAL_RANK modulo the device
    count (ranks of a gloo rehearsal share devices)."""
    import os
    import torch

    ndev = max(torch.cuda.device_count(), 1)
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"]) % ndev
    return 0


def process_group():
    """(rank, world) of the default torch.distributed group when one is initialised with more than one rank, else
    None.  torch is not imported for the question: a process that never imported it is not distributed."""
    import sys

    if "torch" not in sys.modules:
        return None
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return None
    return dist.get_rank(), dist.get_world_size()


class Transport:
    """What carries a slab context's halo planes and reductions, and how it got there."""

    def __init__(self):
        self.kind = "none"            # "rccl" | "torch" | "none"
        self.backend = ""             # torch.distributed backend of the group
        self.ranks_seen = 1
        self.rccl_error = None        # message of a failed RCCL start-up when the callbacks took over
        self.comm: Optional[Communicator] = None

    @property
    def label(self) -> str:
        """ "rccl" (library-issued ncclSend/Recv/AllReduce on the solver stream), "torch-nccl" / "torch-gloo" (callbacks
        into torch.distributed, one host synchronisation per exchange), "none" (single rank)."""
        if self.kind == "rccl":
            return "rccl"
        return f"torch-{self.backend}" if self.kind == "torch" else "none"


def strict_rccl() -> bool:
    import os

    return os.environ.get("PERPHIL_STRICT_RCCL", "0") not in ("", "0")


def attach_transport(ctx: _ffi.Context, group=None, transport: str = "auto", strict: Optional[bool] = None,
                     inject_rccl_failure: bool = False) -> Transport:
    """Give a slab context its communication.  "auto": the library's own RCCL transport with the nccl backend, the
    torch.distributed callbacks otherwise (gloo rehearsals).  When the RCCL transport fails to start (library missing,
    ncclCommInitRank or the neighbour self-test failing on ANY rank) all ranks move to the callback transport
    together - still RCCL on the device with the nccl backend, but one host synchronisation per exchange - and the
    Transport says so (`label` "torch-nccl", `rccl_error` the message); `strict` (default: PERPHIL_STRICT_RCCL=1)
    raises instead.  `inject_rccl_failure` (tests): behave as if the RCCL start-up had failed."""
    import torch.distributed as dist

    t = Transport()
    t.backend = dist.get_backend(group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if strict is None:
        strict = strict_rccl()
    if transport == "auto":
        transport = "rccl" if t.backend == "nccl" else "torch"
    t.kind = transport
    if transport == "rccl":
        try:
            if inject_rccl_failure:
                raise RuntimeError("injected RCCL start-up failure (test)")
            t.ranks_seen = init_rccl(ctx, group)   # raises on EVERY rank when any rank failed
        except RuntimeError as e:
            if strict:
                raise RuntimeError(f"RCCL transport unavailable ({e}); without PERPHIL_STRICT_RCCL / --strict the run "
                                   f"continues on the torch.distributed callbacks and says so") from e
            if rank == 0:
                print(f"[perphil_amd.distributed] {e}: continuing on the torch.distributed callbacks on all ranks",
                      flush=True)
            t.kind, t.rccl_error = "torch", str(e)
    if t.kind == "torch":
        t.comm = Communicator(group)
        t.comm.attach(ctx)
        one = np.ones(1)   # the callback transport's own check: one all-reduce of 1 over the group
        t.comm._allreduce(None, one.ctypes.data_as(C.c_void_p), 1)
        t.ranks_seen = int(one[0])
    if t.ranks_seen != world:
        raise RuntimeError(f"transport {t.label} sees {t.ranks_seen} ranks of {world}")
    return t


class SlabSolver:
    """The benchmark's driver of a slab-decomposed unit cube: the reference's objects (fd.UnitCubeMesh under an
    initialised torch.distributed group -> this rank's slab context with its transport, DirichletBC with the
    manufactured pressures -> the slab's boundary nodes) and a solver configuration stepped WITHOUT fetching the
    solution (bench.py times assemble + solve with the result left on the device; callers of the public API use
    solve_dpp / solve_dpp_nonlinear on the same mesh objects and get the same context)."""

    def __init__(self, n_cells: int, world: int, rank: int, device: int, k1: float, k2: float, beta: float, mu: float,
                 inner_rtol: float = 1e-10, smooth: int = 2, kind: int = _ffi.CELL_HEX, group=None,
                 inner_pc: int = _ffi.PC_MG, transport: str = "auto", inner_reduction: float = 0.0,
                 inner_norm: int = 0, strict: Optional[bool] = None, inject_rccl_failure: bool = False):
        from .manufactured_solutions import exact_expressions_3d
        from .parameters import DPPParameters
        from .solver import _apply_bcs
        from . import fd

        self.params = (k1, k2, beta, mu)
        self.mesh = fd.UnitCubeMesh(n_cells, n_cells, n_cells, hexahedral=(kind == _ffi.CELL_HEX))
        self.mesh.distribute(group=group, device=device, transport=transport, strict=strict,
                             inject_rccl_failure=inject_rccl_failure)
        self.slab: Slab = self.mesh.slab
        assert self.slab is not None and (self.slab.world, self.slab.rank) == (world, rank)
        self.ctx = self.mesh.context()
        self.transport_info: Transport = self.mesh.transport
        self.comm = self.transport_info.comm
        self.ranks_seen = self.transport_info.ranks_seen
        V = fd.FunctionSpace(self.mesh, "CG", 1)
        self.W = V * V
        _, p1, _, p2 = exact_expressions_3d(self.mesh, DPPParameters(k1=k1, k2=k2, beta=beta, mu=mu))
        self.bcs = [fd.DirichletBC(self.W.sub(0), p1, "on_boundary"), fd.DirichletBC(self.W.sub(1), p2, "on_boundary")]
        _apply_bcs(self.ctx, self.W, self.bcs)
        self.global_dofs = self.W.dim()
        cfg = _ffi.SolverCfg()
        cfg.ksp_type, cfg.pc_type, cfg.restart, cfg.max_it = _ffi.KSP_GMRES, _ffi.PC_FIELDSPLIT, 30, 50000
        cfg.rtol, cfg.atol = 1e-8, 1e-12
        cfg.inner_ksp_type, cfg.inner_pc_type, cfg.inner_max_it = _ffi.KSP_CG, inner_pc, 50000
        cfg.inner_rtol, cfg.inner_atol = inner_rtol, 1e-300
        cfg.picard, cfg.picard_rtol, cfg.picard_atol, cfg.picard_max_it = 1, 1e-8, 1e-12, 100
        cfg.mg_smooth = smooth
        cfg.inner_reduction = inner_reduction
        cfg.inner_norm = inner_norm
        self.cfg = cfg
        self.info = None
        self.monolithic = False  # True: also materialise the monolithic CSR (monolithic Krylov solves)

    def step(self):
        k1, k2, beta, mu = self.params
        self.ctx.set_option("invalidate_KM", 1)
        self.ctx.assemble(k1, k2, beta, mu, monolithic=self.monolithic)
        _, self.info, _ = self.ctx.solve(self.cfg, fetch=False)
        return self.info

    @property
    def transport_label(self) -> str:
        return self.transport_info.label

    def gather_solution(self) -> Optional[np.ndarray]:
        """Global field-major solution on every rank (tests / small runs only)."""
        return gather_field_major(self.slab, self.ctx.solution())


def gather_field_major(slab: Slab, x_local: np.ndarray, group=None) -> np.ndarray:
    """Global field-major vector from every rank's local one (owned entries; ghost planes dropped): zero-padded
    all-reduce over the group - small runs, tests and post-processing, not the hot path."""
    import torch
    import torch.distributed as dist

    s = slab
    nl, ng = s.n_local, s.plane * (s.nz + 1)
    nf = x_local.size // nl
    full = np.zeros(nf * ng)
    for f in range(nf):
        full[f * ng:(f + 1) * ng][s.owned_global] = x_local[f * nl:(f + 1) * nl][s.owned_local]
    t = torch.from_numpy(full)
    if dist.get_backend(group) == "nccl":
        d = t.cuda()
        dist.all_reduce(d, group=group)
        full = d.cpu().numpy()
    else:
        dist.all_reduce(t, group=group)
    return full
