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


def init_rccl(ctx: _ffi.Context, group=None) -> None:
    """Create the context's RCCL communicator: rank 0 draws the unique id, torch.distributed broadcasts
    it (any backend), every rank joins; then the library's self-test runs on the context stream."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    path = _loaded_rccl_path()
    ident = np.zeros(128, dtype=np.uint8)
    if rank == 0:
        # a failure here must still reach the broadcast below, or the other ranks would wait for it forever
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
    if not ident.any():
        raise RuntimeError("rank 0 could not draw an RCCL unique id (librccl not loadable?)")
    ctx._check(_ffi.lib.pph_comm_init_rccl(ctx._h, rank, world, ident.ctypes.data_as(C.c_void_p), path))
    ctx._check(_ffi.lib.pph_comm_selftest(ctx._h))


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
        self._halo_c = _HALO_FN(self._halo)
        self._allreduce_c = _ALLREDUCE_FN(self._allreduce)

    def _dev(self, ptr: int, off: int, count: int):
        return self.torch.as_tensor(_DevView(ptr + 8 * off, count), device="cuda")

    def _halo(self, _user, vec, plane, send_lo, recv_lo, send_hi, recv_hi) -> int:
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
                 inner_norm: int = 0):
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
        self.transport = transport
        if transport == "rccl":
            ok = 1
            try:
                init_rccl(self.ctx, group)
            except Exception as e:
                ok = 0
                if world > 1:
                    print(f"[perphil_amd.distributed] RCCL transport unavailable on rank {rank} ({e!r})", flush=True)
            # every rank must end up on the same transport: one failed self-test sends all of them to the callbacks
            if world > 1:
                import torch
                import torch.distributed as dist

                flag = torch.tensor([float(ok)])
                if self.comm.backend == "nccl":
                    flag = flag.cuda()
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = int(flag.item())
            if not ok:
                if world > 1 and rank == 0:
                    print("[perphil_amd.distributed] falling back to torch.distributed callbacks on all ranks", flush=True)
                self.transport = "torch"
        if self.transport == "torch":
            self.comm.attach(self.ctx)
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
