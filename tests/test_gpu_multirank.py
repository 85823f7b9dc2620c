"""Slab-decomposed (multi-rank) HIP path against the single-context run, on ONE GPU: the ranks share
the device and communicate through gloo (tools/slab_check.py).  Same solver code as the 8-GPU RCCL run;
only the transport differs."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,cells,kind,pc,solver", [(2, 16, "hex", "mg", "picard"), (4, 16, "hex", "mg", "picard"),
                                                         (2, 16, "tet", "mg", "picard"), (2, 8, "hex", "jacobi", "picard"),
                                                         (2, 16, "tet", "mg", "gmres_fs"), (2, 8, "hex", "mg", "cg_block2"),
                                                         (4, 8, "hex", "mg", "gmres_jacobi"),
                                                         (4, 32, "hex", "mg", "picard-inexact"), (2, 32, "tet", "mg", "picard-inexact"),
                                                         # slabs of unequal thickness (cells not a multiple of the ranks)
                                                         (4, 18, "hex", "mg", "picard-inexact"), (3, 20, "tet", "mg", "picard"),
                                                         # the device-scalar CG branch of the RCCL transport, over the callbacks
                                                         (2, 32, "hex", "mg", "picard-inexact-devscal"),
                                                         (4, 32, "hex", "mg", "picard-inexact-devscal"),
                                                         (2, 16, "tet", "jacobi", "picard-devscal"),
                                                         # halo exchange overlapped with the interior rows of the products
                                                         (2, 32, "hex", "mg", "picard-inexact-overlap"),
                                                         (4, 32, "hex", "mg", "picard-inexact-overlap"),
                                                         (2, 32, "hex", "mg", "picard-inexact-overlap-devscal"),
                                                         (3, 24, "tet", "mg", "picard-overlap")])
def test_slab_runs_match_single_context(world, cells, kind, pc, solver):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tools", "slab_check.py"), "--cells", str(cells), "--backend", "gloo", "--kind", kind,
           "--inner-pc", pc, "--solver", solver.split("-")[0]] + (["--inexact"] if "inexact" in solver else []) + (
               ["--device-scalars"] if solver.endswith("devscal") else []) + (["--halo-overlap"] if "overlap" in solver else [])
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    line = [l for l in r.stdout.splitlines() if l.startswith("world=")]
    assert r.returncode == 0, (line, r.stdout[-2000:], r.stderr[-2000:])
    assert line and "max rel diff" in line[0]


def _run_api_check(world, extra, timeout=280):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tools", "api_slab_check.py"), "--backend", "gloo"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    line = [l for l in r.stdout.splitlines() if l.startswith("world=")]
    assert r.returncode == 0 and line and "api parity: ok" in line[0], (line, r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("world,cells,kind,params,nonlinear", [
    # the slabs BEHIND solve_dpp / solve_dpp_nonlinear - hexahedra and tetrahedra, three and more option
    # dictionaries, boxes with nx != ny != nz and slabs of unequal thickness
    (2, (16, 16, 16), "hex", "PICARD_MG_SOLVER_PARAMS", True),
    (4, (16, 16, 16), "hex", "PICARD_MG_INEXACT_SOLVER_PARAMS", True),
    (2, (16, 16, 16), "tet", "PICARD_MG_SOLVER_PARAMS", True),
    (4, (12, 10, 18), "tet", "PICARD_MG_INEXACT_SOLVER_PARAMS", True),
    (2, (16, 16, 16), "hex", "FIELDSPLIT_MG_PARAMS", False),
    (4, (16, 16, 16), "tet", "FIELDSPLIT_MG_PARAMS", False),
    (2, (12, 16, 20), "hex", "FIELDSPLIT_LU_PARAMS", False),       # the reference's dictionary: GMRES + field-split, LU blocks
    (2, (8, 8, 8), "hex", "CG_BLOCK_JACOBI_PARAMS", False),
    (4, (8, 8, 8), "tet", "GMRES_JACOBI_PARAMS", False),
    (2, (8, 8, 8), "hex", "LINEAR_SOLVER_PARAMS", False),          # preonly + lu: the direct-equivalent solve
    (3, (10, 9, 20), "hex", "PICARD_JACOBI_SOLVER_PARAMS", True),
    (2, (16, 16, 0), "quad", "FIELDSPLIT_LU_PARAMS", False),       # 2D: replicated on every rank
])
def test_public_api_on_slabs_matches_single_context(world, cells, kind, params, nonlinear):
    """solve_dpp / solve_dpp_nonlinear called on every rank of a process group exactly as a single process calls them
    (reference src/perphil/solvers/solver.py:30-76): equal to the COMM_SELF solve of the same objects."""
    extra = ["--cells"] + [str(c) for c in cells] + ["--kind", kind, "--params", params] + (["--nonlinear"] if nonlinear else [])
    _run_api_check(world, extra)


def test_public_api_on_slabs_with_plain_boundary_data():
    """Constant and global nodal-array Dirichlet data (no manufactured solution) through the slab path."""
    _run_api_check(2, ["--cells", "10", "12", "16", "--kind", "hex", "--params", "PICARD_MG_SOLVER_PARAMS", "--nonlinear",
                       "--constant-bc"])


def test_rccl_start_up_failure_continues_on_labelled_callbacks():
    """a failing RCCL start-up (injected) moves every rank to the torch.distributed callbacks, the
    result is right and says which transport carried it; PERPHIL_STRICT_RCCL=1 keeps the old behaviour (raise)."""
    out = _run_api_check(2, ["--cells", "16", "16", "16", "--params", "PICARD_MG_INEXACT_SOLVER_PARAMS", "--nonlinear",
                             "--inject-rccl-failure"])
    assert "continuing on the torch.distributed callbacks" in out
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "api_slab_check.py"), "--backend", "gloo",
           "--params", "PICARD_MG_INEXACT_SOLVER_PARAMS", "--nonlinear", "--inject-rccl-failure"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, PERPHIL_STRICT_RCCL="1"), capture_output=True, text=True, timeout=280)
    assert r.returncode != 0 and "RCCL transport unavailable" in (r.stdout + r.stderr)


def test_split_products_stay_inside_one_partial_sum_area():
    """a product launched as interior + two boundary row ranges lays its per-workgroup partial sums
    back to back in ONE reduction slot's area; more of them than the area holds would alias the next slot (mode 7 keeps
    r.Ap and Ap.Ap there) and give CG a wrong alpha silently.  The grids are capped so that the three launches together
    stay within `part_cap` (= the area, 4096; lowered to 32 here so that the interior launch of a 32^3 slab hits the
    cap): device-scalar branch + merged all-reduce + split products, equal to the single context as always."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "slab_check.py"), "--cells", "32", "--backend",
           "gloo", "--inexact", "--device-scalars", "--halo-overlap", "--set", "part_cap=32"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "partial sums (cap 32)" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("cells,overlap", [(32, True), (64, True), (64, False)])
def test_slab_operators_on_row_dictionaries(cells, overlap):
    """The row dictionaries (option sell_dict) on slab operators - ghost planes, symmetric storage, products split into
    interior + boundary row ranges (k_spmv_sell<DICT> on chunk ranges) and whole-slab products (k_spmv_dict_walk from
    64^3 on): same sweeps / iterations / solution as the single context, dictionaries in use on every rank's blocks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "slab_check.py"), "--cells", str(cells), "--backend",
           "gloo", "--inexact", "--device-scalars", "--set", "sell_dict_min_rows=1"] + (["--halo-overlap"] if overlap else [])
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    line = [l for l in r.stdout.splitlines() if l.startswith("world=")]
    assert r.returncode == 0 and line, (r.stdout[-2000:], r.stderr[-2000:])
    assert int(line[0].split("row dictionaries on ")[1].split()[0]) >= 3, line[0]


def test_failed_halo_exchange_is_reported_not_computed_through():
    """A halo callback that fails (on every rank, at the same exchange) must surface as a COMM error from the solve -
    not as a result computed on stale ghost planes - and the context must keep refusing afterwards (sticky status)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "slab_check.py"), "--cells", "16", "--backend",
           "gloo", "--inexact", "--fail-halo-after", "40"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "COMM error on every rank" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_rccl_plumbing_single_rank():
    """A 1-rank RCCL communicator exercises dlopen, ncclCommInitRank, grouped ncclSend/ncclRecv and
    ncclAllReduce on the context stream (the 8-GPU run uses the same calls with more ranks)."""
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "from perphil_amd import _ffi\n"
        "from perphil_amd.distributed import init_rccl\n"
        "dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d', rank=0, world_size=1)\n"
        "ctx = _ffi.Context(0); init_rccl(ctx); print('rccl selftest ok'); dist.destroy_process_group()\n"
    ) % (ROOT, _free_port())
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and "rccl selftest ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("config", [4, 5])
def test_bench_gpus_2_starts_its_own_ranks(config):
    """`python bench.py --gpus 2` exactly as the driver starts the 1-GPU run (no launcher): bench.py spawns
    torch.distributed.run itself; gloo rehearsal, both ranks on the one GPU.  Both BASELINE multi-GPU configurations have a
    launcher (4: hexahedra, Picard-split; 5: Kuhn tetrahedra, k1/k2 = 1e4, GMRES + field-split) and the line carries what
    a measured curve needs to be read: transport, communication counts and event-timed communication beside the step."""
    import json

    env = dict(os.environ, PERPHIL_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "16", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--skip-fine-bench", "--skip-csr", "--config", str(config)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[0])
    c = out["config"]
    assert out["n_gpus"] == 2 and c["transport"] == "torch-gloo" and c["ranks_seen"] == 2 and c["rccl_native_error"] is None
    assert c["halo_exchanges_per_step"] > 0 and c["allreduces_per_step"] > 0 and out["value"] > 0
    assert c["halo_overlap"] == 1
    cm = c["comm"]
    assert cm["halo_ms"] > 0 and cm["allreduce_ms"] > 0 and cm["halo_exchanges_timed"] == c["halo_exchanges_per_step"]
    assert 0 < cm["kernel_ms"] < cm["step_ms_with_comm_timers"]
    if config == 5:
        assert "tetrahedra" in c["workload"] and c["gmres_iterations"] == 4 and c["cells"] == 6 * 16 ** 3
    else:
        assert c["picard_sweeps"] > 0
