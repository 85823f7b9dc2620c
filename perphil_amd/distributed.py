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


class SlabSolver:
    """The DPP unit-cube problem on `world` slabs: assemble + block-Picard solve with multigrid-CG
    block solves, manufactured Dirichlet data; same algorithm and iteration counts as one GPU."""

    def __init__(self, n_cells: int, world: int, rank: int, device: int, k1: float, k2: float, beta: float, mu: float,
                 inner_rtol: float = 1e-10, smooth: int = 2, kind: int = _ffi.CELL_HEX, group=None,
                 inner_pc: int = _ffi.PC_MG, transport: str = "auto", inner_reduction: float = 0.0,
                 inner_norm: int = 0, allow_fallback: bool = False):
        from .manufactured_solutions import exact_expressions_3d
        from .parameters import DPPParameters
        from . import fd

        self.params = (k1, k2, beta, mu)
        self.slab: Slab = make_slab(n_cells, n_cells, n_cells, world, rank)
        self.comm = Communicator(group)
        assert self.comm.world == world and self.comm.rank == rank
        self.ctx = _ffi.Context(device)
        s = self.slab
        self.ctx.mesh_build(3, kind, s.nx, s.ny, s.nz, s.z_begin, s.z_count, s.ghost_lo, s.ghost_hi)
        # transport: "rccl" = ncclSend/Recv/AllReduce issued by the library on its own stream (default with
        # the nccl backend), "torch" = callbacks into torch.distributed (needed for gloo rehearsals)
        if transport == "auto":
            transport = "rccl" if self.comm.backend == "nccl" else "torch"
        self.transport_requested = transport
        self.transport = transport
        self.ranks_seen = world if world == 1 else 0
        if transport == "rccl":
            try:
                self.ranks_seen = init_rccl(self.ctx, group)   # raises on EVERY rank when any rank failed
            except RuntimeError as e:
                # no silent downgrade: the callback transport synchronises the host on every exchange, so a scaling
                # figure taken on it must say so (bench.py records config.transport) and has to be asked for
                if not allow_fallback:
                    raise RuntimeError(f"RCCL transport unavailable ({e}); pass allow_fallback=True (bench.py "
                                       f"--allow-fallback) to run on the torch.distributed callbacks instead") from e
                if rank == 0:
                    print(f"[perphil_amd.distributed] {e}: falling back to torch.distributed callbacks on all ranks",
                          flush=True)
                self.transport = "torch"
        if self.transport == "torch":
            self.comm.attach(self.ctx)
            if world > 1:
                # the callback transport's own check: one all-reduce of 1 over the group
                one = np.ones(1)
                self.comm._allreduce(None, one.ctypes.data_as(C.c_void_p), 1)
                self.ranks_seen = int(one[0])
        mesh = fd.UnitCubeMesh(n_cells, n_cells, n_cells, hexahedral=(kind == _ffi.CELL_HEX))
        loc, glob = s.boundary_local()
        X = mesh.node_coordinates(glob)
        _, p1, _, p2 = exact_expressions_3d(mesh, DPPParameters(k1=k1, k2=k2, beta=beta, mu=mu))
        self.ctx.set_dirichlet(0, loc, p1(X))
        self.ctx.set_dirichlet(1, loc, p2(X))
        self.global_dofs = 2 * (n_cells + 1) ** 3
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
        """What carries the halo planes and reductions: "rccl" (library-issued ncclSend/Recv/AllReduce on the solver
        stream), "torch-nccl" / "torch-gloo" (callbacks into torch.distributed, one host synchronisation per
        exchange)."""
        return "rccl" if self.transport == "rccl" else f"torch-{self.comm.backend}"

    def gather_solution(self) -> Optional[np.ndarray]:
        """Global field-major solution on rank 0 (tests / small runs only)."""
        torch, dist = self.comm.torch, self.comm.dist
        s = self.slab
        x = self.ctx.solution()
        nl, ng = s.n_local, s.plane * (s.nz + 1)
        full = np.zeros(2 * ng)
        for f in (0, 1):
            full[f * ng:(f + 1) * ng][s.owned_global] = x[f * nl:(f + 1) * nl][s.owned_local]
        t = torch.from_numpy(full)
        if self.comm.on_device:
            d = t.cuda()
            dist.all_reduce(d, group=self.comm.group)
            full = d.cpu().numpy()
        else:
            dist.all_reduce(t, group=self.comm.group)
        return full
