/*
 * perphil_hip.h — C ABI of the MI355X-native DPP hot path (libperphil_hip.so).
 *
 * The reference (ThermoPhase-FCSRG/perphil) exposes no FFI: its boundary for this path is the
 * Python call  solve_dpp(W, model_params, bcs, solver_parameters, options_prefix) -> Solution
 * (reference src/perphil/solvers/solver.py:30-76) whose arithmetic runs inside Firedrake/PETSc.
 * This header declares what a ctypes binding of that call needs; each entry point cites the
 * reference interface (or the third-party work triggered from it) that it replaces.
 * perphil_amd/_ffi.py is the binding; INTEGRATION.md shows the stub a perphil maintainer adds.
 *
 * Conventions
 *   - every function returns 0 on success, a negative pph_status on failure; no exceptions and
 *     no aborts cross this boundary; pph_last_error() returns a message for the last failure.
 *   - the caller owns every host buffer it passes (borrowed for the duration of the call only);
 *     the library owns all device memory behind the opaque handle; nothing returned by pointer
 *     outlives pph_ctx_destroy().
 *   - one host thread drives one context (not re-entrant); independent contexts are independent.
 *   - numbering: node (i,j,k) -> i + (nx+1)*(j + (ny+1)*k) inside the LOCAL box of the context;
 *     monolithic dof = field*n + node (field-major; reference
 *     src/perphil/experiments/iterative_bench.py:323-324).
 *   - all floating point is IEEE fp64; indices are int32 (columns, cell->dof map) / int64 (row
 *     pointers, counts).
 */
#ifndef PERPHIL_HIP_H
#define PERPHIL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pph_ctx pph_ctx;

typedef enum {
  PPH_OK = 0,
  PPH_ERR_INVALID = -1,   /* bad argument / call order          -> Python ValueError   */
  PPH_ERR_HIP = -2,       /* HIP runtime failure                -> Python RuntimeError */
  PPH_ERR_NOMEM = -3,     /* device or host allocation failed   -> Python MemoryError  */
  PPH_ERR_DIVERGED = -4,  /* Krylov/Picard hit max_it or broke down (result still written) */
  PPH_ERR_COMM = -5       /* RCCL failure                                              */
} pph_status;

/* cell kinds of the structured unit square / cube
 * (reference src/perphil/mesh/builtin.py:20 -> fd.UnitSquareMesh quads / "left"-diagonal
 *  triangles; src/perphil/experiments/petsc_profiling_3d.py:31 -> fd.UnitCubeMesh, 6 Kuhn tets
 *  per cube; notebooks/condition-number-study-3d.py:66 -> hexahedral=True) */
enum { PPH_CELL_QUAD = 0, PPH_CELL_TRI = 1, PPH_CELL_HEX = 2, PPH_CELL_TET = 3 };

/* Krylov / preconditioner / outer-loop kinds: the supported subset of the PETSc option
 * dictionaries in reference src/perphil/solvers/parameters.py:4-102 */
enum { PPH_KSP_PREONLY = 0, PPH_KSP_CG = 1, PPH_KSP_GMRES = 2 };
enum {
  PPH_PC_NONE = 0,
  PPH_PC_JACOBI = 1,      /* pc_type jacobi                                              */
  PPH_PC_BLOCK2 = 2,      /* 2x2 node-block Jacobi (both pressures of one node coupled)    */
  PPH_PC_FIELDSPLIT = 3,  /* pc_type fieldsplit, pc_fieldsplit_type multiplicative         */
  PPH_PC_MG = 4,          /* geometric multigrid V-cycle (scalar blocks; inside fieldsplit/Picard) */
  PPH_PC_ILU = 5          /* pc_type ilu, pc_factor_levels 0: ILU(0) in the natural row order, level-scheduled
                           * (monolithic system, or the scalar blocks inside fieldsplit / Picard); single context */
};

typedef struct {
  int32_t ksp_type;      /* PPH_KSP_*  (ksp_type)                                        */
  int32_t pc_type;       /* PPH_PC_*   (pc_type)                                         */
  int32_t restart;       /* GMRES restart, PETSc default 30                              */
  int32_t max_it;        /* ksp_max_it                                                   */
  double rtol;           /* ksp_rtol                                                     */
  double atol;           /* ksp_atol                                                     */
  /* block solves of the field-split PC / Picard sweeps (fieldsplit_0_/fieldsplit_1_ options;
   * the reference's LU block solves become inner Krylov solves run to inner_rtol)          */
  int32_t inner_ksp_type;   /* PPH_KSP_PREONLY | PPH_KSP_CG | PPH_KSP_GMRES (restart 30, zero guess per solve) */
  int32_t inner_pc_type;    /* PPH_PC_NONE | PPH_PC_JACOBI | PPH_PC_MG | PPH_PC_ILU */
  int32_t inner_max_it;
  int32_t picard;        /* 0: Krylov on the monolithic system; 1: block Picard (fixed-stress)
                          * outer loop per reference src/perphil/forms/dpp.py:196-203       */
  double inner_rtol;
  double inner_atol;
  double picard_rtol;    /* snes_rtol */
  double picard_atol;    /* snes_atol */
  int32_t picard_max_it; /* snes_max_it */
  int32_t mg_smooth;     /* smoothing steps per level side for PPH_PC_MG (default 2)     */
  double inner_reduction; /* > 0: block solves stop once their preconditioned residual has dropped by this
                          * factor from its value at the start of the solve (inexact Picard sweeps; warm
                          * starts make the sequence converge to the exact fixed point), or at inner_rtol,
                          * whichever comes first.  0: inner_rtol only.                       */
  int32_t inner_norm;    /* norm the block solves test (ksp_norm_type of the fieldsplit_i_ solvers): 0 =
                          * preconditioned ||P^-1 r|| (PETSc's default for left-preconditioned CG), 1 =
                          * unpreconditioned ||r||_2 (KSP_NORM_UNPRECONDITIONED): the recurrence residual is
                          * known before the preconditioner runs, so a block solve of k iterations costs k
                          * preconditioner applications instead of k + 1; 2 = none (KSP_NORM_NONE): no convergence
                          * test, every block solve runs exactly inner_max_it CG iterations - a Picard sweep then
                          * holds no host decision and is enqueued (replayed from a hipGraph) in one go              */
  int32_t inner_exact;   /* 1: the block solves stand for the reference's LU blocks (fieldsplit_i_pc_type lu): where a block has at
                          * most 4096 rows (the plumbing configurations) it is solved to inner_rtol inside ONE workgroup, on chip
                          * (Jacobi-CG, no host round trip), instead of by the host-driven Krylov loop; larger blocks: as inner_*
                          * say.  0: always as inner_* say                                                              */
} pph_solver_cfg;

typedef struct {
  int32_t iterations;        /* outer Krylov its (ksp.getIterationNumber(), solver.py:73) or Picard sweeps */
  int32_t inner_iterations;  /* total inner Krylov iterations                            */
  int32_t converged;         /* 1 / 0                                                    */
  int32_t inner_failed;      /* 1: a block solve of the field-split PC / a Picard sweep, or the coarsest multigrid
                              * solve, hit its iteration limit or broke down (the outer result may still converge) */
  double resnorm;            /* final (preconditioned) residual norm (ksp.getResidualNorm(), solver.py:74) */
  double rhs_norm;           /* ||F(u0)||_2, PETSc's "0 SNES Function norm"               */
} pph_solve_info;

/* ---- context ------------------------------------------------------------------------- */
/* replaces: process-wide PETSc/Firedrake initialisation behind `import firedrake` */
int pph_ctx_create(int device, pph_ctx** out);
int pph_ctx_destroy(pph_ctx* ctx);
/* message of the last failure on ctx (ctx may be NULL: last failure of pph_ctx_create) */
const char* pph_last_error(const pph_ctx* ctx);
/* blocks until all device work of the context has finished */
int pph_ctx_synchronize(pph_ctx* ctx);

/* ---- mesh + cell->dof map --------------------------------------------------------------
 * replaces: create_mesh() (reference src/perphil/mesh/builtin.py:4-20), fd.UnitCubeMesh call
 * sites (petsc_profiling_3d.py:31), create_function_spaces()/MixedFunctionSpace (CG-1 dof map,
 * src/perphil/forms/spaces.py:34-35).
 * The context's LOCAL box holds cell layers [z_cell_begin, z_cell_begin+z_cell_count) of the
 * global nx*ny*nz mesh (2D: nz = 0, z_cell_begin = 0, z_cell_count = 0).  ghost_lo / ghost_hi
 * mark the lowest / highest local node plane as a ghost plane owned by the neighbouring slab
 * (multi-GPU cell-slab decomposition); single GPU: whole mesh, no ghosts. */
int pph_mesh_build(pph_ctx* ctx, int dim, int cell_kind, int nx, int ny, int nz,
                   int z_cell_begin, int z_cell_count, int ghost_lo, int ghost_hi);
int pph_mesh_sizes(const pph_ctx* ctx, int64_t* n_nodes, int64_t* n_cells, int32_t* nodes_per_cell,
                   int64_t* nnz_block);
int pph_get_dofmap(pph_ctx* ctx, int32_t* cells_host /* [n_cells][nodes_per_cell] */);
int pph_get_coords(pph_ctx* ctx, double* coords_host /* [n_nodes][dim] */);

/* ---- Dirichlet data ---------------------------------------------------------------------
 * replaces: fd.DirichletBC(W.sub(field), expr, "on_boundary") as consumed at
 * reference src/perphil/solvers/solver.py:66.  `nodes` are local node ids, `vals` the boundary
 * values (evaluated by the caller, e.g. from the manufactured solution,
 * src/perphil/utils/manufactured_solutions.py:39-51,87-88).  Replaces earlier data of `field`. */
int pph_set_dirichlet(pph_ctx* ctx, int field, const int64_t* nodes, const double* vals, int64_t count);

/* ---- assembly -----------------------------------------------------------------------------
 * replaces: dpp_form() (reference src/perphil/forms/dpp.py:95-132) + the TSFC element kernel /
 * PyOP2 cell loop / MatSetValues / BC elimination that solver.solve() triggers
 * (src/perphil/solvers/solver.py:71 -> SNESJacobianEval), and dpp_delayed_form() (dpp.py:135-205).
 * Builds on device: scalar CSR pattern, K and M by cell-local integration + scatter-add, then the
 * Dirichlet-eliminated blocks A11 = (k1/mu)K + (beta/mu)M, A22 = (k2/mu)K + (beta/mu)M,
 * A12 = A21^T = -(beta/mu)M and the lifted right-hand side.  `monolithic != 0` additionally
 * materialises the 2n x 2n field-major CSR. */
int pph_assemble_dpp(pph_ctx* ctx, double k1, double k2, double beta, double mu, int monolithic);

/* ---- solve --------------------------------------------------------------------------------
 * replaces: LinearVariationalSolver.solve() -> KSPSolve (reference src/perphil/solvers/solver.py:67-71);
 * with cfg->picard the block Picard loop stated by dpp_delayed_form (dpp.py:196-203).
 * x_host (len 2n, caller-owned) receives the full solution u = u0 + du, field-major.
 * hist (optional, capacity hist_cap) receives residual norms per outer iteration, entry 0 = initial. */
int pph_solve(pph_ctx* ctx, const pph_solver_cfg* cfg, double* x_host, pph_solve_info* info,
              double* hist, int hist_cap);
/* same solve, result left on the device (timing without the PCIe copy); fetch with pph_get_solution */
int pph_solve_device(pph_ctx* ctx, const pph_solver_cfg* cfg, pph_solve_info* info, double* hist, int hist_cap);
int pph_get_solution(pph_ctx* ctx, double* x_host /* len 2n */);
/* Page-locked host memory for the x_host / result arrays above (no reference counterpart: PETSc's Vec lives in host memory
 * already).  A copy into pageable memory is staged by the HIP runtime and pays the first touch of a fresh array's pages; the
 * Python host side keeps a small pool of pinned result buffers (perphil_amd/_ffi.py: Context.solution).  Freed with
 * pph_host_free; both are independent of any context. */
int pph_host_alloc(size_t bytes, void** out);
int pph_host_free(void* p);

/* ---- export for parity checks -----------------------------------------------------------
 * replaces: get_matrix_data_from_form() -> petsc_matrix.getValuesCSR()
 * (reference src/perphil/solvers/conditioning.py:66-102).
 * which: 0 monolithic (needs monolithic assembly), 1 K, 2 M, 3 A11, 4 A22, 5 A12, 6 A21 */
int pph_csr_sizes(const pph_ctx* ctx, int which, int64_t* nrows, int64_t* nnz);
int pph_get_csr(pph_ctx* ctx, int which, int64_t* rowptr, int32_t* col, double* val);
int pph_get_rhs(pph_ctx* ctx, double* rhs_host /* len 2n */, double* u0_host /* len 2n, may be NULL */);
/* y = A x with the selected matrix (host vectors); replaces PETSc MatMult for tests */
int pph_spmv(pph_ctx* ctx, int which, const double* x_host, double* y_host);
/* `reps` back-to-back device SpMVs on resident vectors, average kernel ms via HIP events */
int pph_spmv_bench(pph_ctx* ctx, int which, int reps, double* avg_ms);

/* HBM bandwidth calibration on this device (tools/bw_probe.py, tools/pmc_probe.py): `bytes` streamed with
 * 16 B per lane by `blocks` workgroups, mode 0 read-only, mode 1 copy; average ms per launch. */
int pph_bw_probe(pph_ctx* ctx, int64_t bytes, int mode, int blocks, double* ms_out);

/* ---- error norms (post-processing) -----------------------------------------------------------------
 * replaces: l2_error() / h1_seminorm_error() (reference src/perphil/utils/postprocessing.py:89-124) for
 * the manufactured pressures of src/perphil/utils/manufactured_solutions.py:39-51 (2D), :87-88 (3D).
 * `nodal_host`: the n nodal values of p1_h (field 0) or p2_h (field 1); nq-point Gauss rule per
 * direction (1..8) on quadrilateral / hexahedral cells, collapsed onto the simplex (Duffy transform) on
 * triangles / tetrahedra. */
int pph_error_norms_mms(pph_ctx* ctx, int field, const double* nodal_host, double k1, double k2, double beta,
                        double mu, int nq, double* l2_out, double* h1s_out);
/* The same two norms against ANY exact field, sampled by the caller at the quadrature points
 * replaces: l2_error / h1_seminorm_error for an arbitrary UFL expression (reference src/perphil/utils/postprocessing.py:89-124:
 * sqrt(assemble((p_h - p)^2 dx)), sqrt(assemble(|grad p_h - grad p|^2 dx))) and l2_errors_against_reference
 * (src/perphil/experiments/iterative_bench.py:340-362).  pph_quadrature_points writes the physical coordinates of the
 * nq^dim points of the cells [cell_begin, cell_begin + cell_count) to xq_host[((cell - cell_begin) npts + q) dim + d]
 * (same rule and point order as pph_error_norms_mms); the caller evaluates its field (and gradient) there and
 * pph_error_norms_sampled returns the SQUARED partial norms over those cells: exact_q_host[(cell - cell_begin) npts + q],
 * grad_q_host[((cell - cell_begin) npts + q) dim + d]; either may be NULL (zero field / zero gradient), so the norms of
 * a finite-element function itself come from the same call.  Chunking the cell range bounds the host arrays; nodal_host may
 * be NULL from the second chunk on: the nodal field uploaded by the previous call on this context is used again. */
int pph_quadrature_points(pph_ctx* ctx, int nq, int64_t cell_begin, int64_t cell_count, double* xq_host);
int pph_error_norms_sampled(pph_ctx* ctx, const double* nodal_host, int nq, int64_t cell_begin, int64_t cell_count,
                            const double* exact_q_host, const double* grad_q_host, double* l2sq_out, double* h1sq_out);

/* Darcy velocity u = -conductivity * grad(p_h), L2-projected onto the CG-1 vector space of the mesh
 * replaces: calculate_darcy_velocity_from_pressure() (reference src/perphil/utils/postprocessing.py:34-63,
 * fd.project(-k grad p, VectorFunctionSpace(mesh, "CG", 1))).  p_host: n nodal pressures; u_host: [n][dim]
 * (node-major, like a Firedrake vector Function's dat).  Mass-matrix solves run to rtol 1e-13. */
int pph_darcy_velocity(pph_ctx* ctx, const double* p_host, double conductivity, double* u_host);

/* ---- multi-GPU communication hooks -------------------------------------------------------------
 * replaces: PETSc's implicit VecScatter halo exchange and VecDot all-reduce under mpiexec (never run in
 * the reference).  One context per rank holds one cell slab (pph_mesh_build with
 * z_cell_begin/z_cell_count/ghost_lo/ghost_hi).  The library calls
 *   halo(user, vec, plane_elems, send_lo, recv_lo, send_hi, recv_hi): offsets (in elements, -1 = no
 *        neighbour) into the DEVICE vector `vec`: send [send_lo, +plane) to rank-1 and receive its
 *        top owned plane into [recv_lo, +plane); likewise send_hi/recv_hi with rank+1.  The context
 *        stream is idle when the callback runs; the callback returns when the data have arrived.
 *   allreduce(user, vals, count): sum `count` HOST doubles over all ranks, in place.
 * Both return 0 on success.  perphil_amd/distributed.py implements them over torch.distributed. */
typedef int (*pph_halo_fn)(void* user, double* vec, int64_t plane_elems, int64_t send_lo, int64_t recv_lo,
                           int64_t send_hi, int64_t recv_hi);
typedef int (*pph_allreduce_fn)(void* user, double* vals, int64_t count);
int pph_comm_set_callbacks(pph_ctx* ctx, int rank, int world, pph_halo_fn halo, pph_allreduce_fn allreduce,
                           void* user);
/* RCCL transport (default on the multi-GPU node): the library issues grouped ncclSend/ncclRecv of the
 * neighbour planes and ncclAllReduce of the reduction scalars on its own stream.  librccl is resolved with
 * dlopen (`libpath` may be NULL/empty: default search, then /opt/rocm/lib).  Rank 0 draws the 128-byte
 * unique id, the launcher broadcasts it, every rank calls pph_comm_init_rccl; pph_comm_selftest verifies
 * a send/recv to self and an all-reduce on the context stream. */
int pph_rccl_available(const char* libpath);   /* 0 when librccl loads with every entry point used here: every rank
                                                * checks (and the launcher agrees on the result) BEFORE any rank enters
                                                * ncclCommInitRank */
int pph_rccl_unique_id(const char* libpath, uint8_t* id128);
int pph_comm_init_rccl(pph_ctx* ctx, int rank, int world, const uint8_t* id128, const char* libpath);
/* straight-line self-test: every rank issues all three phases (to self, both slab neighbours, all-reduce) whatever
 * fails on the way, and reports afterwards; ranks_seen (optional) = world size observed by the all-reduce */
int pph_comm_selftest(pph_ctx* ctx);
int pph_comm_selftest2(pph_ctx* ctx, int* ranks_seen);
/* halo exchanges and all-reduces of the last solve, and the sticky communication status: after the first failed
 * exchange / reduction on a context every solve returns PPH_ERR_COMM until the transport is set up again */
int pph_comm_stats(pph_ctx* ctx, int64_t* halo_exchanges, int64_t* allreduces, int* status);
/* with option "time_comm" on: summed durations (ms) of the last solve's halo exchanges (out4[0]) and all-reduces (out4[1])
 * and how many of each were timed (out4[2], out4[3]).  RCCL transport: device time between two HIP events on the issuing
 * stream (an exchange overlapped with interior rows counts its full duration on the communication stream); callback
 * transport: host time inside the callback.  Lets a measured scaling curve be split into communication and the rest. */
int pph_comm_times(pph_ctx* ctx, double* out4);

/* ---- stats ---------------------------------------------------------------------------------
 * replaces: PETSc -log_view event times scraped by reference src/perphil/experiments/petsc_profiling.py:302-447.
 * out[0] mesh+pattern ms, [1] integration ms (fused assembly: the whole assembly), [2] elimination/blocks ms of
 * the two-step path (0 when fused), [3] last solve ms;
 * SpMV accounting of the last solve per kernel variant v (0: plain, 1: fused with the p.Ap dot):
 * out[4+3v] sum of per-launch durations in ms (0 unless option "time_spmv" is on), out[5+3v] launches,
 * out[6+3v] algorithmic bytes (CSR launches: 12 nnz + 20 nrows; stencil-ELL launches: (8 S + 16 + e) nrows with S the
 * STORED slots per row - 14 of 27 for hexahedra in symmetric storage - and e = 8 / 16 / 32..48 for the residual,
 * Jacobi-update and Picard-bookkeeping epilogues' extra vectors); out[10] halo exchanges of the last solve;
 * out[11..13] the same three figures (ms, launches, bytes; both variants together) for the launches on
 * fine-level operators only (rows >= nodes of the mesh), i.e. without the coarser multigrid levels;
 * out[14] products of the last solve launched as interior + boundary rows (option "halo_overlap", slabs only);
 * out[15] 1 when the stencil-ELL blocks of the last assembly use symmetric storage;
 * out[16] the most partial sums a split product has written into one reduction slot so far (<= option "part_cap");
 * out[17] stencil-ELL operators (fine blocks + multigrid levels) whose products run on a row dictionary (option
 * "sell_dict": distinct rows stored once + a 2-byte class per row; their launches count (2 + 16 + e) nrows bytes),
 * out[18] distinct rows of A11 found by the last build (0: not tried), out[19] its status on the device (1 in use,
 * 0 not built, -1 more distinct rows than the cap, -2 a row failed the bitwise check: plain storage is used);
 * out[20] ms spent so far in FIRST builds of row dictionaries (hash build + table + first bitwise check + the read-back of
 * the verdict; once per operator, mesh and Dirichlet-set pair - a symbolic-phase cost like the pattern of a CSR matrix),
 * out[21] how many such builds; out[22] 1 when the products of A11 take the classes of the interior planes from the plane below
 * (option "sell_dict_zconst": the class words of four planes only are read - launches count (16 + e) nrows bytes + those). */
int pph_get_timers(pph_ctx* ctx, double* out, int n);
/* tuning / profiling switches (no reference counterpart; defaults in brackets):
 *   "op_format" [1]      operator format of the scalar blocks inside block solves / Picard sweeps: 1 stencil-ELL
 *                        (values only, val[slot][row]; written directly by the fused assembly), 0 CSR.  pph_get_csr /
 *                        pph_spmv export CSR either way (converted on demand).
 *   "sell_rpt" [2], "sell_blocks" [4096; z-walk: one per CU], "sell_group" [1]   stencil-ELL SpMV: rows per thread, grid cap, XCD chunk group
 *   "sell_sym" [1]       symmetric blocks store the diagonal and the upper stencil slots only; "sell_sym_slabs" [1]: also
 *                        on slabs, where ghost rows then keep their entries towards owned columns (the mirrors of the
 *                        owned rows' lower entries; operators converted from CSR values stay in full storage there)
 *   "sell_zwalk" [4], "sell_zwalk_min_chunks" [5500], "sell_xmap" [1]   symmetric product on large 3D levels: a workgroup
 *                        walks this many node planes at one in-plane position (>= 1000: a balanced share of a whole z
 *                        column); the in-plane positions of one XCD's workgroups are consecutive
 *   "spmv_lanes" [0 = automatic, 4..64 lanes per CSR row], "spmv_blocks" [0 = 1024 workgroups], "spmv_bench_mode" [0]
 *   "time_spmv" [0]      1: bracket every SpMV launch of a solve with a HIP event pair on the context stream
 *   "asm_kernel" [2]     multilinear assembly: 0 cell-centred scatter-add (atomics), 1 node-centred gather, 2 fused kernels
 *   "asm_tile" [1]       fused multilinear assembly: 1 single-pass tile kernel on levels of at least
 *                        "asm_tile_min_nodes" [30000] nodes (2: on every level), 0 two-pass (element rows + gather)
 *   "asm_affine" [1]     tile kernel: a cell whose parallel edges are equal vectors gets its constant geometry factor once
 *                        per cell; 0: Jacobian at every Gauss point of every cell (the general pass)
 *   "asm_fused" [1]      the node-centred pass writes the eliminated blocks, lifted right-hand side and smoother
 *                        diagonal directly; 0: K and M first, then separate elimination kernels
 *   "asm_keep_km" [0]    1: the fused pass also stores K and M (otherwise they are integrated on demand)
 *   "invalidate_KM"      drop the integrated K and M so that the next assemble integrates again
 *   "mg_fused" [1]       V(1,1) on stencil-ELL levels: fused smoother / transfer kernels + on-chip tail; 0 general cycle
 *   "mg_tail_rows" [5000] levels with at most min(this, 1024) rows join the single-workgroup tail of the cycle
 *   "coarse_max_it" [500] iteration limit of the coarsest-level Jacobi-CG (reported through inner_failed)
 *   "mg_fp32" [0], "mg_replicate_below" [40000 nodes], "coarse_on_device" [1]   multigrid: fp32 copies of the V-cycle
 *                        operators (CSR only), replication threshold of coarse levels on slabs, coarsest solve on the device
 *   "mg_replicate_rows_per_rank" [40000], "mg_replicate_cap" [1e6]   slabs: a level is also replicated when its share per
 *                        rank is at most this many nodes (its kernels are shorter than one halo exchange) and the whole
 *                        level has at most mg_replicate_cap nodes; 0 restores the global threshold alone
 *   "time_comm" [0]      1: every halo exchange / all-reduce of a solve is timed (pph_comm_times)
 *   "fetch_spin" [1]     reduction results reach the host through a mapped mirror the host polls; 0: D2H copy + sync
 *   "merge_allreduce" [1] slabs, CG block solves on stencil-ELL operators: the product also sums r.Ap and Ap.Ap, and
 *                        { p.Ap, r.Ap, Ap.Ap, r.r of the previous update } travel in ONE all-reduce; the host forms the next
 *                        r.r by one step of the recurrence (two scalar all-reduces per iteration instead of three)
 *   "use_graphs" [1]     launch sequences replayed from captured hipGraphs: ILU(0) sweeps, the launch-only Picard sweeps of
 *                        inner_norm 2, and CG iteration bodies on systems of up to "graph_cg_max_rows" [0] rows (2: always)
 *   "device_scalars" [0] 1: the device-scalar CG branch also over the callback transport (tests)
 *   "halo_overlap" [1]   slabs: products on levels of at least "halo_overlap_min_rows" [200000] rows are launched as
 *                        interior rows + boundary rows; 1: the exchange of the operand's ghost planes runs on a second
 *                        stream while the interior rows are computed, 2: the same launches, exchange first (the two
 *                        give bit-identical results).  The three launches share one reduction slot's partial-sum area:
 *                        their grids are capped so that together they write at most "part_cap" [4096, the area's size;
 *                        tests lower it] partial sums
 *   "sell_dict" [1]      row dictionaries for the symmetric stencil-ELL operators of at least "sell_dict_min_rows" [1e6]
 *                        rows: after an assembly the distinct rows (all 27 / 15 / 9 / 7 coefficients a product would load for a row,
 *                        bit patterns; all four cell kinds) are stored once and every row gets a 2-byte class; every assembly checks every row
 *                        against its class bit for bit; products then read the class instead of the stored values and are
 *                        bit-identical.  More than "sell_dict_cap" [256] distinct rows (graded meshes, node spacing not
 *                        exact in binary) or a failed check: stored values as before.  "sell_dict_walk" [1]: whole-operator
 *                        products on hexahedra keep the x window in registers (k_spmv_dict_walk), "sell_dict_blocks"
 *                        [1024] their grid, "sell_dict_zconst" [1]: where the class of an in-plane position is the same on all
 *                        interior planes (verified on the class array at the build) a step takes its classes from the step
 *                        below instead of loading them; 0 takes effect at once, 1 at the next assembly
 *   "fold_finals" [1]    single context: the final reductions of p.Ap and of the multigrid cycle's r.z are summed inside the
 *                        kernels that consume them instead of by launches of their own (same sums, same order)
 *   "transfer_bench"     diagnostic: times `value` launches of the fine-level interpolation / restriction kernels of an existing
 *                        hexahedral hierarchy and prints the result on stderr (tools/r4_transfer_probe.py) */
int pph_set_option(pph_ctx* ctx, const char* name, double value);

#ifdef __cplusplus
}
#endif
#endif /* PERPHIL_HIP_H */
