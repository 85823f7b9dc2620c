"""N > 1 host logic on CPU: the slab partition (perphil_amd/partition.py) drives a world_size-2 (and 4)
gloo run of the ORACLE arithmetic — local boxes with ghost planes, rows of owned nodes assembled from
local cells only, one ghost plane per neighbour exchanged before every SpMV, dots over owned entries +
all-reduce — and must reproduce the serial oracle.  The HIP path uses the same partition and the same
exchange pattern (tools/slab_check.py checks it on the GPU)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp

    from oracle import dpp_oracle as o
    from perphil_amd.partition import make_slab

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    P = o.Params(k1=1.0, k2=0.01)
    slab = make_slab(n, n, n, world, rank)
    # local box mesh = the slab's cell layers of the global mesh (same element matrices: uniform grid)
    om = o.build_mesh(3, o.CELL_HEX, n, n, slab.z_count)
    om.coords[:, 2] = (om.coords[:, 2] * slab.z_count + slab.z_begin) / n
    K, M = o.assemble_scalar(om)
    a, b, _c = P.abc
    A = (a * K + b * M).tocsr()
    loc, glob = slab.boundary_local()
    mask = np.zeros(slab.n_local, bool)
    mask[loc] = True
    ghost = np.ones(slab.n_local, bool)
    ghost[slab.owned_local] = False
    # Dirichlet elimination, then empty ghost rows
    A = o.eliminate_dirichlet(A, np.nonzero(mask)[0]).tolil()
    A[np.nonzero(ghost)[0], :] = 0.0
    A = A.tocsr()
    # global reference pieces on every rank (small)
    gm = o.build_mesh(3, o.CELL_HEX, n, n, n)
    gs = o.build_system(gm, P)
    Ag = gs.A[: gs.n, : gs.n].tocsr()
    bg = gs.rhs[: gs.n]
    pl = slab.plane
    own_l, own_g = slab.owned_local, slab.owned_global

    def halo(v):
        reqs = []
        if slab.ghost_lo:
            s = torch.from_numpy(v[pl:2 * pl].copy()); r = torch.empty(pl, dtype=torch.float64)
            reqs += [(dist.isend(s, rank - 1), None), (dist.irecv(r, rank - 1), (r, slice(0, pl)))]
        if slab.ghost_hi:
            nl = slab.n_local
            s = torch.from_numpy(v[nl - 2 * pl:nl - pl].copy()); r = torch.empty(pl, dtype=torch.float64)
            reqs += [(dist.isend(s, rank + 1), None), (dist.irecv(r, rank + 1), (r, slice(nl - pl, nl)))]
        for w, tgt in reqs:
            w.wait()
            if tgt is not None:
                v[tgt[1]] = tgt[0].numpy()

    def gdot(x, y):
        t = torch.tensor([float(np.dot(x[own_l], y[own_l]))], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    def spmv(x):
        halo(x)
        return A @ x

    # Jacobi-CG on the slab-decomposed macro block; compare with the serial oracle run
    bl = np.zeros(slab.n_local)
    bl[own_l] = bg[own_g]
    dinv = np.where(A.diagonal() != 0, 1.0 / np.where(A.diagonal() != 0, A.diagonal(), 1.0), 1.0)
    x = np.zeros(slab.n_local); r = bl.copy(); z = dinv * r; p = z.copy()
    rz = gdot(r, z); tol = 1e-10 * np.sqrt(gdot(z, z)); its = 0
    while its < 2000:
        q = spmv(p)
        alpha = rz / gdot(p, q)
        x += alpha * p; r -= alpha * q; z = dinv * r; its += 1
        if np.sqrt(gdot(z, z)) <= tol:
            break
        rz_new = gdot(r, z); p = z + (rz_new / rz) * p; rz = rz_new
    ref = o.pcg(Ag, bg, o.jacobi_apply(Ag), rtol=1e-10)
    full = np.zeros(gs.n)
    full[own_g] = x[own_l]
    t = torch.from_numpy(full)
    dist.all_reduce(t)
    err = float(np.abs(full - ref.x).max() / np.abs(ref.x).max())
    # the slab-assembled owned rows equal the global rows (column ids shifted by the box offset)
    off = slab.z_begin * pl
    rows_ok = True
    for lr in range(own_l.start, own_l.stop, max(1, (own_l.stop - own_l.start) // 40)):
        gl = Ag.getrow(lr + off)
        lo = A.getrow(lr)
        gd = dict(zip(gl.indices.tolist(), gl.data.tolist()))
        ld = {c + off: v for c, v in zip(lo.indices.tolist(), lo.data.tolist()) if v != 0.0}
        gd = {c: v for c, v in gd.items() if v != 0.0}
        rows_ok &= gd.keys() == ld.keys() and all(abs(gd[c] - ld[c]) <= 1e-13 * max(1.0, abs(gd[c])) for c in gd)
    if rank == 0:
        out.put((its, ref.its, err, rows_ok))
    else:
        out.put((None, None, None, rows_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 8), (4, 8)])
def test_slab_partition_reproduces_serial_oracle(world, n):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[3] for r in res)
    its, ref_its, err, _ = next(r for r in res if r[0] is not None)
    assert abs(its - ref_its) <= 1 and err < 1e-9


def test_partition_invariants():
    from perphil_amd.partition import make_slab

    for nz, world in ((8, 2), (16, 4), (256, 8), (10, 4), (250, 8), (37, 3)):   # uneven splits included
        planes = []
        for r in range(world):
            s = make_slab(4, 6, nz, world, r)
            assert s.ghost_lo == (r > 0) and s.ghost_hi == (r < world - 1)
            assert len(s.owned_planes) >= 2
            assert s.local_planes == len(s.owned_planes) + s.ghost_lo + s.ghost_hi
            assert (s.owned_local.stop - s.owned_local.start) == len(s.owned_planes) * s.plane
            planes += list(s.owned_planes)
            loc, glob = s.boundary_local()
            assert np.all(glob - loc == s.z_begin * s.plane)
        assert planes == list(range(nz + 1))   # every node plane owned exactly once
    with pytest.raises(ValueError):
        make_slab(4, 4, 7, 4, 0)     # fewer than 2 cell layers per rank


def _api_worker(rank, world, port, out):
    """Host side of the rank-aware public API (no GPU): under an initialised group fd.UnitCubeMesh is this rank's slab,
    DirichletBC data are evaluated on the slab's boundary nodes, Functions hold local vectors and gather() restores the
    global one; 2D and COMM_SELF meshes stay whole."""
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    import torch.distributed as dist

    import perphil_amd as pa
    from perphil_amd import fd
    from perphil_amd.partition import make_slab

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ok = True
    nx, ny, nz = 5, 4, 9
    mesh = fd.UnitCubeMesh(nx, ny, nz, hexahedral=True)
    s = make_slab(nx, ny, nz, world, rank)
    ok &= mesh.distributed and mesh.slab == s and mesh.num_local_vertices() == s.n_local
    ok &= mesh.num_vertices() == (nx + 1) * (ny + 1) * (nz + 1)
    V = fd.FunctionSpace(mesh, "CG", 1)
    W = V * V
    ok &= W.dim() == 2 * mesh.num_vertices() and W.local_dim() == 2 * s.n_local
    params = pa.DPPParameters()
    _, p1, _, p2 = pa.exact_expressions_3d(mesh, params)
    whole = fd.UnitCubeMesh(nx, ny, nz, hexahedral=True, comm=fd.COMM_SELF)
    ok &= (not whole.distributed) and whole.num_local_vertices() == whole.num_vertices()
    gb = whole.boundary_nodes()
    gv = p1(whole.node_coordinates(gb))
    nodes, vals = fd.DirichletBC(W.sub(0), p1, "on_boundary").nodes_and_values()
    gl = mesh.local_to_global(nodes)
    lookup = dict(zip(gb.tolist(), gv.tolist()))
    ok &= all(g in lookup for g in gl.tolist()) and np.array_equal(vals, np.array([lookup[g] for g in gl.tolist()]))
    # every global boundary node inside this rank's box is among the local ones (ghost planes included)
    lo, hi = s.z_begin * s.plane, (s.z_begin + s.local_planes) * s.plane
    ok &= np.array_equal(np.sort(gl), gb[(gb >= lo) & (gb < hi)])
    # a global nodal array and a Constant as data
    arr = np.arange(mesh.num_vertices(), dtype=np.float64)
    n2, v2 = fd.DirichletBC(W.sub(1), arr, "on_boundary").nodes_and_values()
    ok &= np.array_equal(v2, arr[mesh.local_to_global(n2)])
    _, v3 = fd.DirichletBC(W.sub(1), fd.Constant(3.0), "on_boundary").nodes_and_values()
    ok &= bool(np.all(v3 == 3.0))
    # a Function interpolated locally gathers to the globally interpolated one
    f = fd.Function(W)
    f.sub(0).interpolate(p1)
    f.sub(1).interpolate(lambda X: X[:, 0] + 10.0 * X[:, 2])
    g = f.gather()
    Xg = whole.node_coordinates()
    ok &= g.function_space().mesh() is mesh.serial_twin() and not g.function_space().mesh().distributed
    ok &= np.array_equal(g.sub(0).vector(), p1(Xg)) and np.array_equal(g.sub(1).vector(), Xg[:, 0] + 10.0 * Xg[:, 2])
    ok &= f.owned().shape == (2 * len(s.owned_planes) * s.plane,)
    ok &= abs(f.sub(1).at((0.2, 0.5, 1.0)) - (0.2 + 10.0)) < 1e-14
    # the cache of evaluated boundary data follows the datum's parameters
    bc = fd.DirichletBC(W.sub(0), p1, "on_boundary")
    a = bc.nodes_and_values()[1]
    ok &= bc.nodes_and_values()[1] is a
    _, q1, _, _ = pa.exact_expressions_3d(mesh, pa.DPPParameters(k1=2.0))
    bc.value = q1
    ok &= not np.array_equal(bc.nodes_and_values()[1], a)
    # 2D meshes and meshes with too few layers are replicated
    m2 = fd.UnitSquareMesh(8, 8, quadrilateral=True)
    ok &= (not m2.distributed) and m2.replicated_because is not None
    m3 = fd.UnitCubeMesh(4, 4, 2 * world - 1)
    ok &= (not m3.distributed) and "fewer than 2" in m3.replicated_because
    out.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_public_api_objects_on_a_distributed_mesh(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_api_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)
