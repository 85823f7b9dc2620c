// C-ABI entry points of libperphil_hip.so (see include/perphil_hip.h for the contract and the
// reference interfaces each one replaces).
#include "pph_internal.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

static thread_local std::string g_last_error;

void pph_set_error(pph_ctx* ctx, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  g_last_error = buf;
}

template <typename T>
int DevBuf<T>::alloc(pph_ctx* ctx, size_t count) {
  if (count == 0) count = 1;
  if (p && n == count) return PPH_OK;  // same size: keep the allocation (hipMalloc/hipFree of multi-GB buffers is slow)
  release();
  void* q = nullptr;
  // 64 bytes of slack: the aligned-wide SpMV reads whole 16-byte groups around a row's range
  hipError_t e = hipMalloc(&q, count * sizeof(T) + 64);
  if (e != hipSuccess) {
    pph_set_error(ctx, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
    (void)hipGetLastError();
    return (e == hipErrorOutOfMemory) ? PPH_ERR_NOMEM : PPH_ERR_HIP;
  }
  p = (T*)q;
  n = count;
  return PPH_OK;
}

template <typename T>
void DevBuf<T>::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  n = 0;
}

template struct DevBuf<double>;
template struct DevBuf<float>;
template struct DevBuf<int32_t>;
template struct DevBuf<int64_t>;
template struct DevBuf<uint8_t>;
template struct DevBuf<uint16_t>;
template struct DevBuf<uint32_t>;
template struct DevBuf<unsigned long long>;

// Dirichlet data of one field: the ghost-plane flag (bit 1) of the mask survives, the constrained flag (bit 0) and the
// boundary values are rebuilt from the (node, value) list (byte stores of different lanes never share a byte)
__global__ __launch_bounds__(256) void k_dirichlet_clear(uint8_t* __restrict__ mask, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    mask[i] &= (uint8_t)2;
}
__global__ __launch_bounds__(256) void k_dirichlet_set(uint8_t* __restrict__ mask, double* __restrict__ g,
                                                       const int64_t* __restrict__ nodes, const double* __restrict__ vals,
                                                       int64_t count) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t nd = nodes[i];
    mask[nd] |= (uint8_t)1;
    g[nd] = vals[i];
  }
}

extern "C" {

const char* pph_last_error(const pph_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int pph_ctx_create(int device, pph_ctx** out) {
  if (!out) return PPH_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    pph_set_error(nullptr, "no HIP device available (%s)", hipGetErrorString(e));
    return PPH_ERR_HIP;
  }
  if (device < 0 || device >= count) {
    pph_set_error(nullptr, "device %d out of range [0,%d)", device, count);
    return PPH_ERR_INVALID;
  }
  pph_ctx* ctx = new (std::nothrow) pph_ctx();
  if (!ctx) return PPH_ERR_NOMEM;
  ctx->device = device;
  auto fail = [&](const char* what, hipError_t err) {
    pph_set_error(nullptr, "%s failed: %s", what, hipGetErrorString(err));
    delete ctx;
    return PPH_ERR_HIP;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->num_cus = cus;
  }
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
  if ((e = hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
  if ((e = hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreateWithFlags(&ctx->ev_h, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_asm0)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_asm1)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreateWithFlags(&ctx->ev_lam, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_lam), 64 * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
    return fail("hipHostMalloc", e);
  // host mirror of the reduction results: pinned, mapped and coherent, so that a one-wave kernel can publish results
  // straight into it and the host can poll a sequence word instead of paying a stream synchronisation (la_fetch)
  if ((e = hipHostMalloc((void**)&ctx->h_scal, sizeof(double) * (PPH_MAX_SCAL + 8),
                         hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
    return fail("hipHostMalloc", e);
  memset(ctx->h_scal, 0, sizeof(double) * (PPH_MAX_SCAL + 8));
  if ((e = hipHostGetDevicePointer((void**)&ctx->h_scal_dev, ctx->h_scal, 0)) != hipSuccess)
    return fail("hipHostGetDevicePointer", e);
  ctx->h_seq = reinterpret_cast<unsigned long long*>(ctx->h_scal + PPH_MAX_SCAL);
  ctx->dict_alarm = reinterpret_cast<int*>(ctx->h_scal + PPH_MAX_SCAL + 4);   // (the mirror has 8 spare words behind the slots)
  ctx->h_seq_dev = reinterpret_cast<unsigned long long*>(ctx->h_scal_dev + PPH_MAX_SCAL);
  ctx->dict_alarm_dev = reinterpret_cast<int*>(ctx->h_scal_dev + PPH_MAX_SCAL + 4);
  // reduction results + partial sums (32 slots x 4096 workgroups)
  if (ctx->scal.alloc(ctx, (size_t)PPH_MAX_SCAL + 32 * 4096) < 0) {
    g_last_error = ctx->err;
    delete ctx;
    return PPH_ERR_NOMEM;
  }
  if (ctx->pub_ctr.alloc(ctx, 1) < 0 || hipMemset(ctx->pub_ctr.p, 0, sizeof(unsigned long long)) != hipSuccess) {
    g_last_error = ctx->err;
    delete ctx;
    return PPH_ERR_NOMEM;
  }
  *out = ctx;
  return PPH_OK;
}

// the assembled system is stale (parameters / boundary data changed): buffers are kept for re-use
static void release_system(pph_ctx* ctx) {
  for (auto& I : ctx->ilu) I.valid = false;   // factors of a stale matrix (the level structure survives)
  ctx->asm_ok = false;
  ctx->ell_ok = false;
  ctx->csr_ok = false;
  ctx->mono_ok = false;
  ctx->mg_ok = false;
}

// the mesh goes away: free everything that was sized by it
static void free_system(pph_ctx* ctx) {
  ctx->A11.release(); ctx->A22.release(); ctx->A12.release(); ctx->A21.release();
  ctx->E11.release(); ctx->E22.release(); ctx->E12.release(); ctx->E21.release();
  ctx->D11.release(); ctx->D22.release(); ctx->D12.release(); ctx->DG.release();
  ctx->rhs.release(); ctx->u0.release(); ctx->sol.release();
  ctx->mrowptr.release(); ctx->mcol.release(); ctx->mval.release();
  mg_release(ctx);
  for (auto& I : ctx->ilu) ilu_release(I);
  release_system(ctx);
}

int pph_ctx_destroy(pph_ctx* ctx) {
  if (!ctx) return PPH_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_system(ctx);
  ctx->mesh.release_all();
  for (int f = 0; f < 2; ++f) { ctx->bcmask[f].release(); ctx->g[f].release(); }
  ctx->rownear.release();
  ctx->sell_tmp.release();
  ctx->post_u.release();
  ctx->dinv0[0].release(); ctx->dinv0[1].release(); ctx->lam0.release();
  comm_release(ctx);
  for (auto& w : ctx->work) w.release();
  for (auto& p : ctx->ev_pool) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
  ctx->scal.release();
  ctx->pub_ctr.release();
  la_release_graphs(ctx);
  if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev_asm0) (void)hipEventDestroy(ctx->ev_asm0);
  if (ctx->ev_asm1) (void)hipEventDestroy(ctx->ev_asm1);
  if (ctx->ev_lam) (void)hipEventDestroy(ctx->ev_lam);
  if (ctx->h_lam) (void)hipHostFree(ctx->h_lam);
  ctx->mg_lam.release();
  if (ctx->ev_x) (void)hipEventDestroy(ctx->ev_x);
  if (ctx->ev_h) (void)hipEventDestroy(ctx->ev_h);
  if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return PPH_OK;
}

int pph_ctx_synchronize(pph_ctx* ctx) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PPH_OK;
}

// ---- mesh ---------------------------------------------------------------------------------------
__global__ void k_fill_u8(uint8_t* __restrict__ p, uint8_t v, int64_t begin, int64_t end) {
  for (int64_t i = begin + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < end;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] |= v;
}

int pph_mesh_build(pph_ctx* ctx, int dim, int cell_kind, int nx, int ny, int nz, int z_cell_begin, int z_cell_count,
                   int ghost_lo, int ghost_hi) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  PPH_REQUIRE(ctx, dim == 2 || dim == 3, "dim must be 2 or 3, got %d", dim);
  PPH_REQUIRE(ctx, nx >= 1 && ny >= 1, "nx, ny must be >= 1");
  if (dim == 2) {
    PPH_REQUIRE(ctx, cell_kind == PPH_CELL_QUAD || cell_kind == PPH_CELL_TRI, "2D cell kinds: quad (0), tri (1)");
    PPH_REQUIRE(ctx, nz == 0 && z_cell_begin == 0 && z_cell_count == 0 && !ghost_lo && !ghost_hi,
                "2D meshes take nz = z_cell_begin = z_cell_count = 0 and no ghost planes");
  } else {
    PPH_REQUIRE(ctx, cell_kind == PPH_CELL_HEX || cell_kind == PPH_CELL_TET, "3D cell kinds: hex (2), tet (3)");
    PPH_REQUIRE(ctx, nz >= 1 && z_cell_count >= 1 && z_cell_begin >= 0 && z_cell_begin + z_cell_count <= nz,
                "slab [%d,%d) outside [0,%d)", z_cell_begin, z_cell_begin + z_cell_count, nz);
    PPH_REQUIRE(ctx, !(ghost_lo && z_cell_begin == 0) && !(ghost_hi && z_cell_begin + z_cell_count == nz),
                "a ghost plane cannot lie on the domain boundary");
  }
  free_system(ctx);
  ctx->mesh.release_all();
  ctx->mesh.km_valid = false;
  ctx->mesh_ok = false;
  ctx->post_u_valid = false;
  PPH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  MeshData& m = ctx->mesh;
  m.dim = dim; m.kind = cell_kind; m.nx = nx; m.ny = ny; m.nz = nz; m.z0 = z_cell_begin; m.nzl = z_cell_count;
  ctx->ghost_lo = ghost_lo ? 1 : 0;
  ctx->ghost_hi = ghost_hi ? 1 : 0;
  m.glo = ctx->ghost_lo;
  m.ghi = ctx->ghost_hi;
  PPH_TRY(pph_launch_mesh(ctx, m));
  ctx->n = m.n;
  ctx->nnzb = m.nnzb;
  for (int f = 0; f < 2; ++f) {
    PPH_TRY(ctx->bcmask[f].alloc(ctx, (size_t)m.n));
    PPH_TRY(ctx->g[f].alloc(ctx, (size_t)m.n));
    PPH_HIP(ctx, hipMemsetAsync(ctx->bcmask[f].p, 0, (size_t)m.n, ctx->stream));
    ctx->bc_dirty = true;
    ctx->bc_epoch++;
    PPH_HIP(ctx, hipMemsetAsync(ctx->g[f].p, 0, sizeof(double) * (size_t)m.n, ctx->stream));
    const int64_t plane = (int64_t)m.px * m.py;
    if (ctx->ghost_lo)
      hipLaunchKernelGGL(k_fill_u8, dim3(64), dim3(256), 0, ctx->stream, ctx->bcmask[f].p, (uint8_t)2, (int64_t)0, plane);
    if (ctx->ghost_hi)
      hipLaunchKernelGGL(k_fill_u8, dim3(64), dim3(256), 0, ctx->stream, ctx->bcmask[f].p, (uint8_t)2, m.n - plane, m.n);
  }
  PPH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  PPH_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  PPH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->t_mesh = ms;
  PPH_HIP(ctx, hipGetLastError());
  ctx->mesh_ok = true;
  return PPH_OK;
}

int pph_mesh_sizes(const pph_ctx* ctx, int64_t* n_nodes, int64_t* n_cells, int32_t* nodes_per_cell,
                   int64_t* nnz_block) {
  if (!ctx || !ctx->mesh_ok) return PPH_ERR_INVALID;
  if (n_nodes) *n_nodes = ctx->mesh.n;
  if (n_cells) *n_cells = ctx->mesh.ncell;
  if (nodes_per_cell) *nodes_per_cell = ctx->mesh.m;
  if (nnz_block) *nnz_block = ctx->mesh.nnzb;
  return PPH_OK;
}

int pph_get_dofmap(pph_ctx* ctx, int32_t* cells_host) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, ctx->mesh_ok && cells_host, "pph_get_dofmap: no mesh or NULL buffer");
  PPH_HIP(ctx, hipMemcpyAsync(cells_host, ctx->mesh.cells.p, sizeof(int32_t) * (size_t)ctx->mesh.ncell * ctx->mesh.m,
                              hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PPH_OK;
}

int pph_get_coords(pph_ctx* ctx, double* coords_host) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, ctx->mesh_ok && coords_host, "pph_get_coords: no mesh or NULL buffer");
  const MeshData& m = ctx->mesh;
  const size_t n = (size_t)m.n;
  std::vector<double> tmp(n);
  const double* src[3] = {m.cx.p, m.cy.p, m.cz.p};
  for (int d = 0; d < m.dim; ++d) {
    PPH_HIP(ctx, hipMemcpyAsync(tmp.data(), src[d], sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i) coords_host[i * m.dim + d] = tmp[i];
  }
  return PPH_OK;
}

// ---- Dirichlet data -----------------------------------------------------------------------------
int pph_set_dirichlet(pph_ctx* ctx, int field, const int64_t* nodes, const double* vals, int64_t count) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  PPH_REQUIRE(ctx, ctx->mesh_ok, "pph_set_dirichlet before pph_mesh_build");
  PPH_REQUIRE(ctx, field == 0 || field == 1, "field must be 0 or 1, got %d", field);
  PPH_REQUIRE(ctx, count >= 0 && (count == 0 || (nodes && vals)), "NULL nodes/vals with count %lld", (long long)count);
  const int64_t n = ctx->n;
  for (int64_t i = 0; i < count; ++i)
    PPH_REQUIRE(ctx, nodes[i] >= 0 && nodes[i] < n, "Dirichlet node %lld outside [0,%lld)", (long long)nodes[i],
                (long long)n);
  release_system(ctx);  // any assembled system is stale now
  // only the (node, value) list travels to the device; mask and boundary-value vector are rebuilt there (a node listed
  // more than once must carry the same value in all its entries)
  DevBuf<int64_t> dn;
  DevBuf<double> dv;
  if (count > 0) {
    PPH_TRY(dn.alloc(ctx, (size_t)count));
    PPH_TRY(dv.alloc(ctx, (size_t)count));
    PPH_HIP(ctx, hipMemcpyAsync(dn.p, nodes, sizeof(int64_t) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    PPH_HIP(ctx, hipMemcpyAsync(dv.p, vals, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  }
  PPH_HIP(ctx, hipMemsetAsync(ctx->g[field].p, 0, sizeof(double) * (size_t)n, ctx->stream));
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_dirichlet_clear, dim3(grid < 1 ? 1 : grid), dim3(256), 0, ctx->stream, ctx->bcmask[field].p, n);
  if (count > 0) {
    const int g2 = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_dirichlet_set, dim3(g2), dim3(256), 0, ctx->stream, ctx->bcmask[field].p, ctx->g[field].p, dn.p,
                       dv.p, count);
  }
  ctx->bc_dirty = true;
  ctx->bc_epoch++;
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the caller's arrays and the staging buffers are free again)
  PPH_HIP(ctx, hipGetLastError());
  return PPH_OK;
}

// ---- assembly -----------------------------------------------------------------------------------
int pph_assemble_dpp(pph_ctx* ctx, double k1, double k2, double beta, double mu, int monolithic) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  PPH_REQUIRE(ctx, ctx->mesh_ok, "pph_assemble_dpp before pph_mesh_build");
  PPH_REQUIRE(ctx, k1 > 0 && k2 > 0 && mu > 0 && beta >= 0, "need k1, k2, mu > 0 and beta >= 0");
  release_system(ctx);
  ctx->a = k1 / mu; ctx->b = beta / mu; ctx->c = k2 / mu;
  float ms = 0.f;
  if (!ctx->mesh.km_valid && pph_can_fuse_assembly(ctx)) {
    // integration needed anyway: element rows, then one node-centred pass straight to the eliminated blocks
    // (no wait for the kernels: the solve's first launches queue up behind them; the time is read when somebody asks,
    // pph_get_timers - a failing kernel surfaces at the solve's first fetch)
    PPH_HIP(ctx, hipEventRecord(ctx->ev_asm0, ctx->stream));
    PPH_TRY(pph_launch_assemble_fused(ctx, monolithic));
    PPH_HIP(ctx, hipEventRecord(ctx->ev_asm1, ctx->stream));
    ctx->asm_time_pending = true;
    ctx->t_bc = 0.0;
    PPH_HIP(ctx, hipGetLastError());
    ctx->asm_ok = true;
    return PPH_OK;
  }
  // K and M depend only on the mesh: integrate once per mesh
  if (!ctx->mesh.km_valid) {
    PPH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    PPH_TRY(pph_launch_assemble_KM(ctx, ctx->mesh));
    ctx->mesh.km_valid = true;
    PPH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    PPH_HIP(ctx, hipEventSynchronize(ctx->ev1));
    PPH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->t_asm = ms;
  }
  PPH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  PPH_TRY(pph_launch_blocks(ctx, monolithic));
  PPH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  PPH_HIP(ctx, hipEventSynchronize(ctx->ev1));
  PPH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->t_bc = ms;
  PPH_HIP(ctx, hipGetLastError());
  ctx->asm_ok = true;
  return PPH_OK;
}

// ---- export -------------------------------------------------------------------------------------
// products (pph_spmv, pph_spmv_bench) on an assembled block run on its stencil-ELL copy when that is the operator format:
// no CSR pattern, no CSR values needed (a 512^3 block has more entries than the pattern's int32 positions can address)
static bool select_sell_only(pph_ctx* ctx, int which, Csr* A) {
  if (!(ctx->mesh_ok && ctx->asm_ok && ctx->op_format == 1 && ctx->ell_ok && which >= 3 && which <= 6)) return false;
  const MeshData& m = ctx->mesh;
  A->rowptr = nullptr; A->col = nullptr; A->val = nullptr; A->nrows = m.n; A->nnz = m.nnzb; A->max_row = m.max_row;
  A->lanes = pph_pick_lanes(ctx, A->nnz, A->nrows);
  A->ell = (which == 3) ? ctx->S11 : (which == 4) ? ctx->S22 : (which == 5) ? ctx->S12 : ctx->S21;
  return true;
}

static int select_csr(pph_ctx* ctx, int which, Csr* A) {
  PPH_REQUIRE(ctx, ctx->mesh_ok, "no mesh");
  PPH_TRY(pph_ensure_pattern(ctx, ctx->mesh));   // (exports are CSR)
  const MeshData& m = ctx->mesh;
  A->rowptr = m.rowptr.p; A->col = m.col.p; A->nrows = m.n; A->nnz = m.nnzb;
  A->lanes = pph_pick_lanes(ctx, A->nnz, A->nrows);
  A->max_row = m.max_row;
  switch (which) {
    case 0:
      PPH_REQUIRE(ctx, ctx->mono_ok, "monolithic CSR not assembled (pph_assemble_dpp(..., monolithic=1))");
      A->rowptr = ctx->mrowptr.p; A->col = ctx->mcol.p; A->val = ctx->mval.p; A->nrows = 2 * m.n; A->nnz = 4 * m.nnzb;
      A->lanes = pph_pick_lanes(ctx, A->nnz, A->nrows);
      A->max_row = 2 * m.max_row;
      return PPH_OK;
    case 1:
    case 2:
      if (!m.km_valid) {   // not kept by the fused assembly (or never assembled): integrate now
        PPH_TRY(pph_launch_assemble_KM(ctx, ctx->mesh));
        ctx->mesh.km_valid = true;
      }
      A->val = (which == 1) ? m.K.p : m.M.p;
      return PPH_OK;
    case 3: case 4: case 5: case 6:
      PPH_REQUIRE(ctx, ctx->asm_ok, "blocks not assembled");
      PPH_TRY(pph_ensure_csr_blocks(ctx));   // the fused assembly keeps the blocks in stencil-ELL form only
      A->val = (which == 3) ? ctx->A11.p : (which == 4) ? ctx->A22.p : (which == 5) ? ctx->A12.p : ctx->A21p();
      return PPH_OK;
    default: pph_set_error(ctx, "unknown matrix selector %d", which); return PPH_ERR_INVALID;
  }
}

// scalar-block selectors of pph_spmv / pph_spmv_bench run on a stencil-ELL copy when op_format is 1
static int attach_sell(pph_ctx* ctx, int which, Csr* A) {
  if (ctx->op_format != 1 || which == 0) return PPH_OK;
  if (which >= 3 && ctx->ell_ok) {
    A->ell = (which == 3) ? ctx->S11 : (which == 4) ? ctx->S22 : (which == 5) ? ctx->S12 : ctx->S21;
    return PPH_OK;
  }
  // K, M and the blocks of the two-step path: symmetric as well (K, M before elimination; A12 only with one Dirichlet set)
  const int sym = pph_sell_sym_from_csr(ctx) && (which != 5 && which != 6 ? 1 : (ctx->a21_alias ? 1 : 0));
  return sell_from_csr(ctx, ctx->mesh, A->val, ctx->sell_tmp, &A->ell, sym);
}

int pph_csr_sizes(const pph_ctx* cctx, int which, int64_t* nrows, int64_t* nnz) {
  pph_ctx* ctx = const_cast<pph_ctx*>(cctx);
  if (!ctx) return PPH_ERR_INVALID;
  Csr A;
  PPH_TRY(select_csr(ctx, which, &A));
  if (nrows) *nrows = A.nrows;
  if (nnz) *nnz = A.nnz;
  return PPH_OK;
}

int pph_get_csr(pph_ctx* ctx, int which, int64_t* rowptr, int32_t* col, double* val) {
  if (!ctx) return PPH_ERR_INVALID;
  Csr A;
  PPH_TRY(select_csr(ctx, which, &A));
  if (rowptr) PPH_HIP(ctx, hipMemcpyAsync(rowptr, A.rowptr, sizeof(int64_t) * (size_t)(A.nrows + 1), hipMemcpyDeviceToHost, ctx->stream));
  if (col) PPH_HIP(ctx, hipMemcpyAsync(col, A.col, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToHost, ctx->stream));
  if (val) PPH_HIP(ctx, hipMemcpyAsync(val, A.val, sizeof(double) * (size_t)A.nnz, hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PPH_OK;
}

int pph_get_rhs(pph_ctx* ctx, double* rhs_host, double* u0_host) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, ctx->asm_ok, "pph_get_rhs before pph_assemble_dpp");
  const size_t bytes = sizeof(double) * 2 * (size_t)ctx->n;
  if (rhs_host) PPH_HIP(ctx, hipMemcpyAsync(rhs_host, ctx->rhs.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (u0_host) PPH_HIP(ctx, hipMemcpyAsync(u0_host, ctx->u0.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PPH_OK;
}

int pph_spmv(pph_ctx* ctx, int which, const double* x_host, double* y_host) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, x_host && y_host, "NULL vector");
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  Csr A;
  if (!select_sell_only(ctx, which, &A)) {
    PPH_TRY(select_csr(ctx, which, &A));
    PPH_TRY(attach_sell(ctx, which, &A));
  }
  DevBuf<double> x, y;
  PPH_TRY(x.alloc(ctx, (size_t)A.nrows));
  PPH_TRY(y.alloc(ctx, (size_t)A.nrows));
  PPH_HIP(ctx, hipMemcpyAsync(x.p, x_host, sizeof(double) * (size_t)A.nrows, hipMemcpyHostToDevice, ctx->stream));
  la_spmv(ctx, A, x.p, y.p);
  PPH_HIP(ctx, hipMemcpyAsync(y_host, y.p, sizeof(double) * (size_t)A.nrows, hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PPH_HIP(ctx, hipGetLastError());
  x.release();
  y.release();
  return PPH_OK;
}

__global__ void k_fill_pattern(double* __restrict__ x, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    // cheap deterministic pseudo-random values in (-1, 1)
    unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    x[i] = (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
}

int pph_spmv_bench(pph_ctx* ctx, int which, int reps, double* avg_ms) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, reps >= 1 && avg_ms, "reps must be >= 1 and avg_ms non-NULL");
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  Csr A;
  if (!select_sell_only(ctx, which, &A)) {
    PPH_TRY(select_csr(ctx, which, &A));
    A.lanes = pph_pick_lanes(ctx, A.nnz, A.nrows);
    PPH_TRY(attach_sell(ctx, which, &A));
  }
  DevBuf<double> x, y;
  PPH_TRY(x.alloc(ctx, (size_t)A.nrows));
  PPH_TRY(y.alloc(ctx, (size_t)A.nrows));
  hipLaunchKernelGGL(k_fill_pattern, dim3(2048), dim3(256), 0, ctx->stream, x.p, A.nrows);
  if (ctx->spmv_bench_mode != 0) {
    // interleaved protocols (what does a launch cost when something else ran in between?): every SpMV is timed
    // with its own event pair; 1: a streaming axpy on two other vectors between products, 2: a one-block
    // kernel between products, 3: products alternate between this matrix and A22 (or A11)
    DevBuf<double> u, v;
    PPH_TRY(u.alloc(ctx, (size_t)A.nrows));
    PPH_TRY(v.alloc(ctx, (size_t)A.nrows));
    la_set(ctx, u.p, 1.0, A.nrows);
    la_set(ctx, v.p, 2.0, A.nrows);
    Csr B = A;
    if (ctx->spmv_bench_mode == 3) {
      PPH_TRY(select_csr(ctx, which == 4 ? 3 : 4, &B));
      B.lanes = A.lanes;
    }
    const bool was = ctx->time_spmv;
    for (int i = 0; i < 5; ++i) la_spmv(ctx, (i & 1) ? B : A, x.p, y.p);
    la_reset_spmv_stats(ctx);
    ctx->time_spmv = true;
    for (int i = 0; i < reps; ++i) {
      la_spmv(ctx, (ctx->spmv_bench_mode == 3 && (i & 1)) ? B : A, x.p, y.p);
      if (ctx->spmv_bench_mode == 1) la_axpy(ctx, u.p, 0.5, v.p, A.nrows);
      if (ctx->spmv_bench_mode == 2) la_set(ctx, u.p, 1.0, 64);
    }
    la_harvest_spmv_times(ctx);
    ctx->time_spmv = was;
    *avg_ms = ctx->t_spmv[0] / reps;
    la_reset_spmv_stats(ctx);
    u.release(); v.release(); x.release(); y.release();
    return PPH_OK;
  }
  for (int i = 0; i < 20; ++i) la_spmv(ctx, A, x.p, y.p);  // warm-up (SURVEY.md §8d protocol: 20 + 200)
  PPH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  for (int i = 0; i < reps; ++i) la_spmv(ctx, A, x.p, y.p);
  PPH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  PPH_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  PPH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *avg_ms = (double)ms / reps;
  PPH_HIP(ctx, hipGetLastError());
  x.release();
  y.release();
  return PPH_OK;
}

int pph_set_option(pph_ctx* ctx, const char* name, double value) {
  if (!ctx || !name) return PPH_ERR_INVALID;
  la_release_graphs(ctx);   // captured iteration bodies were recorded under the old settings
  if (!strcmp(name, "spmv_lanes")) {
    const int v = (int)value;
    PPH_REQUIRE(ctx, v == 0 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64, "spmv_lanes must be 0,4,8,16,32,64");
    ctx->spmv_lanes_override = v;
    return PPH_OK;
  }
  if (!strcmp(name, "mg_replicate_below")) {
    ctx->mg_replicate_below = (int64_t)value;
    mg_release(ctx);
    return PPH_OK;
  }
  if (!strcmp(name, "mg_replicate_rows_per_rank")) {
    ctx->mg_replicate_rows_per_rank = (int64_t)value;
    mg_release(ctx);
    return PPH_OK;
  }
  if (!strcmp(name, "mg_replicate_cap")) {
    ctx->mg_replicate_cap = (int64_t)value;
    mg_release(ctx);
    return PPH_OK;
  }
  if (!strcmp(name, "mg_fp32")) {
    ctx->mg_fp32 = value != 0.0 ? 1 : 0;
    ctx->mg_ok = false;
    return PPH_OK;
  }
  if (!strcmp(name, "asm_kernel")) {
    const int v = (int)value;
    PPH_REQUIRE(ctx, v >= 0 && v <= 2, "asm_kernel: 0 cell-centred scatter-add (atomics), 1 node-centred gather, 2 two-pass gather");
    ctx->asm_kernel = v;
    ctx->mesh.km_valid = false;
    for (auto& L : ctx->mg) L.mesh.km_valid = false;
    ctx->mg_ok = false;
    return PPH_OK;
  }
  if (!strcmp(name, "spmv_blocks")) { ctx->spmv_blocks = (int)value; return PPH_OK; }
  if (!strcmp(name, "use_graphs")) { ctx->use_graphs = (value == 2.0) ? 2 : (value != 0.0 ? 1 : 0); return PPH_OK; }
  if (!strcmp(name, "fetch_spin")) { ctx->fetch_spin = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "device_scalars")) { ctx->device_scalars = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "op_format")) {
    PPH_REQUIRE(ctx, value == 0.0 || value == 1.0, "op_format: 0 CSR, 1 stencil-ELL");
    ctx->op_format = (int)value;
    return PPH_OK;
  }
  if (!strcmp(name, "sell_sym")) {   // takes effect at the next assembly
    ctx->sell_sym = value != 0.0 ? 1 : 0;
    release_system(ctx);
    return PPH_OK;
  }
  if (!strcmp(name, "sell_sym_slabs")) {   // takes effect at the next assembly
    ctx->sell_sym_slabs = value != 0.0 ? 1 : 0;
    release_system(ctx);
    return PPH_OK;
  }
  if (!strcmp(name, "sell_zwalk_min_chunks")) { ctx->sell_zwalk_min_chunks = (int64_t)value; return PPH_OK; }
  if (!strcmp(name, "graph_cg_max_rows")) { ctx->graph_cg_max_rows = (int64_t)value; la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "merge_allreduce")) { ctx->merge_allreduce = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "halo_overlap")) { ctx->halo_overlap = (value == 2.0) ? 2 : (value != 0.0 ? 1 : 0); la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "halo_overlap_min_rows")) { ctx->halo_overlap_min_rows = (int64_t)value; la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "sell_xmap")) { ctx->sell_xmap = value != 0; la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "sell_zwalk")) { ctx->sell_zwalk = value > 0 ? (int)value : 0; return PPH_OK; }
  if (!strcmp(name, "sell_rpt")) { ctx->sell_rpt = (int)value; return PPH_OK; }
  if (!strcmp(name, "sell_blocks")) { ctx->sell_blocks = (int)value; return PPH_OK; }
  if (!strcmp(name, "part_cap")) {   // tests: partial sums one (split) product may write in total
    PPH_REQUIRE(ctx, value >= 32 && value <= PPH_PART_STRIDE && ((int)value % 32) == 0, "part_cap must be a multiple of 32 in [32, %d]", PPH_PART_STRIDE);
    ctx->part_cap = (int)value; la_release_graphs(ctx); return PPH_OK;
  }
  if (!strcmp(name, "sell_flags")) { ctx->sell_flags = (int)value; la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "sell_group")) { ctx->sell_group = (int)value; return PPH_OK; }
  // row dictionaries of the stencil-ELL blocks (pph_internal.h: struct SellDict); takes effect at the next assembly
  if (!strcmp(name, "sell_dict")) {
    ctx->sell_dict = value != 0;
    if (!ctx->sell_dict) {   // off: at once (the plain storage is always there); on: built by the next assembly
      ctx->S11.dict = ctx->S22.dict = ctx->S12.dict = ctx->S21.dict = nullptr;
      for (auto& L : ctx->mg) L.ell[0].dict = L.ell[1].dict = nullptr;
    }
    la_release_graphs(ctx);
    return PPH_OK;
  }
  if (!strcmp(name, "sell_dict_min_rows")) { ctx->sell_dict_min_rows = (int64_t)value; return PPH_OK; }
  if (!strcmp(name, "sell_dict_blocks")) { ctx->sell_dict_blocks = (int)value; la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "sell_dict_zwalk")) { ctx->sell_dict_zwalk = (int)value; la_release_graphs(ctx); return PPH_OK; }
#ifdef PPH_DW_STAMPS
  if (!strcmp(name, "dw_stamps")) return sell_dw_stamps(ctx, value != 0.0);
#endif
  if (!strcmp(name, "sell_dict_fuse")) { ctx->dict_fuse = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "sell_dict_corrupt_row")) {
    // tests: move one row of A11 into another class ON THE DEVICE - the next assembly's fused check must notice that the
    // entries it stores for that row are not its class's and refuse the dictionary
    PPH_REQUIRE(ctx, ctx->D11.on && ctx->D11.cls.p && value >= 0 && (int64_t)value < ctx->n && ctx->D11.ncls > 1, "no dictionary to corrupt");
    uint16_t c = 0;
    PPH_HIP(ctx, hipMemcpy(&c, ctx->D11.cls.p + (int64_t)value, sizeof(c), hipMemcpyDeviceToHost));
    // (another class whose STORED half differs: classes that differ in their mirrored lower half only meet the row through the
    // class adjacencies recorded at build time, which assume what holds outside this test - class arrays do not change)
    const int S = sell_slots(ctx->S11.kind), C0 = S / 2;
    std::vector<double> tab((size_t)ctx->D11.ncls * S);
    PPH_HIP(ctx, hipMemcpy(tab.data(), ctx->D11.tab.p, sizeof(double) * tab.size(), hipMemcpyDeviceToHost));
    int pick = -1;
    for (int k = 1; k < ctx->D11.ncls && pick < 0; ++k) {
      const int cc = (c + k) % ctx->D11.ncls;
      if (memcmp(&tab[(size_t)cc * S + C0], &tab[(size_t)c * S + C0], sizeof(double) * (size_t)(S - C0)) != 0) pick = cc;
    }
    PPH_REQUIRE(ctx, pick >= 0, "no class with another stored half");
    c = (uint16_t)pick;
    PPH_HIP(ctx, hipMemcpy(ctx->D11.cls.p + (int64_t)value, &c, sizeof(c), hipMemcpyHostToDevice));
    ctx->D11.zconst = false;   // (what k_dict_zconst established no longer holds for this class array)
    la_release_graphs(ctx);
    return PPH_OK;
  }
  if (!strcmp(name, "sell_dict_cap")) { ctx->sell_dict_cap = (int)value; return PPH_OK; }
  if (!strcmp(name, "sell_dict_walk")) { ctx->sell_dict_walk = value != 0; la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "fold_finals")) { ctx->fold_finals = value != 0.0 ? 1 : 0; la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "transfer_bench")) {
    // diagnostic (tools/r4_transfer_probe.py): times the fine-level transfer kernels, result on stderr
    double t[2] = {0.0, 0.0};
    PPH_TRY(mg_transfer_bench(ctx, 0, value > 0 ? (int)value : 50, t));
    fprintf(stderr, "transfer_bench: interpolation %.4f ms, restriction %.4f ms per launch\n", t[0], t[1]);
    return PPH_OK;
  }
  if (!strcmp(name, "sell_dict_zconst")) { ctx->sell_dict_zconst = value != 0; la_release_graphs(ctx); return PPH_OK; }
  if (!strcmp(name, "sell_dict_poison")) {
    // tests: mark the dictionaries of the fine blocks as failed ON THE DEVICE only, as a failed re-assembly check would -
    // the products launched for them must then take the stored values (the plain path inside the dictionary kernel)
    static const int bad = -2;
    for (SellDict* D : {&ctx->D11, &ctx->D22, &ctx->D12})
      if (D->state.p) PPH_HIP(ctx, hipMemcpyAsync(D->state.p + 1, &bad, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (value == 2.0 && ctx->dict_alarm) *ctx->dict_alarm = 1;   // ... and raise the word the check kernels raise (sell_dict_poll)
    return PPH_OK;
  }
  if (!strcmp(name, "asm_tile")) { ctx->asm_tile = (value == 2.0) ? 2 : (value != 0.0 ? 1 : 0); return PPH_OK; }
  if (!strcmp(name, "asm_node_xmap")) { ctx->asm_node_xmap = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "asm_node")) { ctx->asm_node = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "asm_uniform")) { ctx->asm_uniform = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "asm_node_probe")) { ctx->asm_node_probe = (int)value; return PPH_OK; }
  if (!strcmp(name, "asm_node_split_min")) { ctx->asm_node_split_min = (int64_t)value; return PPH_OK; }
  if (!strcmp(name, "asm_tile_xmap")) { ctx->asm_tile_xmap = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "asm_affine")) { ctx->asm_affine = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "asm_tile_probe")) { ctx->asm_tile_probe = (int)value; return PPH_OK; }
  if (!strcmp(name, "asm_tile_min_nodes")) { ctx->asm_tile_min_nodes = (int64_t)value; return PPH_OK; }
  if (!strcmp(name, "asm_fused")) { ctx->asm_fused = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "asm_ring")) { ctx->asm_ring = value > 0.0 ? (int)value : 0; return PPH_OK; }
  if (!strcmp(name, "asm_keep_km")) { ctx->asm_keep_km = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "coarse_max_it")) { ctx->coarse_max_it = value >= 1 ? (int)value : 1; return PPH_OK; }
  if (!strcmp(name, "mg_fused")) { ctx->mg_fused = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "mg_tail_rows")) { ctx->mg_tail_rows = (int64_t)value; return PPH_OK; }
  if (!strcmp(name, "coarse_on_device")) { ctx->coarse_on_device = value != 0.0 ? 1 : 0; return PPH_OK; }
  if (!strcmp(name, "spmv_bench_mode")) { ctx->spmv_bench_mode = (int)value; return PPH_OK; }
  if (!strcmp(name, "time_spmv")) { ctx->time_spmv = value != 0.0; return PPH_OK; }
  if (!strcmp(name, "time_comm")) { ctx->time_comm = value != 0.0; return PPH_OK; }
  if (!strcmp(name, "invalidate_KM")) {
    // forget the integrated K and M (all multigrid levels) so that the next assemble + solve integrates
    // again (benchmarks); the buffers are kept
    ctx->mesh.km_valid = false;
    for (auto& L : ctx->mg) L.mesh.km_valid = false;
    ctx->mg_ok = false;
    return PPH_OK;
  }
  pph_set_error(ctx, "unknown option '%s'", name);
  return PPH_ERR_INVALID;
}

int pph_comm_set_callbacks(pph_ctx* ctx, int rank, int world, pph_halo_fn halo, pph_allreduce_fn allreduce,
                           void* user) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "rank %d outside world %d", rank, world);
  PPH_REQUIRE(ctx, world == 1 || (halo && allreduce), "multi-rank contexts need both callbacks");
  comm_release(ctx);
  ctx->rank = rank;
  ctx->world = world;
  ctx->halo_cb = halo;
  ctx->allreduce_cb = allreduce;
  ctx->comm_user = user;
  ctx->comm_status = PPH_OK;
  release_system(ctx);   // operators assembled for another decomposition (storage format, ghost rows) are stale
  mg_release(ctx);
  ctx->mg_ok = false;
  return PPH_OK;
}

// page-locked host memory for result vectors (perphil_amd/_ffi.py keeps a small pool of them): a device-to-host copy into
// pageable memory is staged by the runtime and pays the first touch of every page of a fresh array - 15 - 27 ms for the 272 MB
// solution of a 256^3 problem against 5 - 6 ms into pinned memory
int pph_host_alloc(size_t bytes, void** out) {
  if (!out) return PPH_ERR_INVALID;
  *out = nullptr;
  void* q = nullptr;
  const hipError_t e = hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    g_last_error = std::string("hipHostMalloc failed: ") + hipGetErrorString(e);
    return (e == hipErrorOutOfMemory) ? PPH_ERR_NOMEM : PPH_ERR_HIP;
  }
  *out = q;
  return PPH_OK;
}

int pph_host_free(void* p) {
  if (!p) return PPH_OK;
  const hipError_t e = hipHostFree(p);
  if (e != hipSuccess) { (void)hipGetLastError(); return PPH_ERR_HIP; }
  return PPH_OK;
}

int pph_get_timers(pph_ctx* ctx, double* out, int n) {
  if (!ctx || !out) return PPH_ERR_INVALID;
  la_harvest_spmv_times(ctx);
  // row dictionaries: operators using one now (fine blocks and multigrid levels), classes of A11's, device status of A11's
  int dn = 0;
  if (ctx->asm_time_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(ctx->ev_asm1) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev_asm0, ctx->ev_asm1) == hipSuccess) ctx->t_asm = ms;
    ctx->asm_time_pending = false;
  }
  for (const Sell* E : {&ctx->S11, &ctx->S22, &ctx->S12}) dn += (E->dict && E->dict->on) ? 1 : 0;
  const bool zc11 = ctx->S11.dict && ctx->S11.dict->on && sell_stream_bytes(ctx, ctx->S11) < 2.0;   // (classes constant along z, used)
  for (size_t l = 1; ctx->mg_ok && l < ctx->mg.size(); ++l)   // (levels: only while the hierarchy matches the assembled system)
    for (int f = 0; f < 2; ++f) dn += (ctx->mg[l].ell[f].dict && ctx->mg[l].ell[f].dict->on) ? 1 : 0;
  int dst[2] = {0, 0};
  if (ctx->D11.on && ctx->D11.state.p) {
    (void)hipMemcpyAsync(dst, ctx->D11.state.p, sizeof(dst), hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
  }
  const double v[23] = {ctx->t_mesh, ctx->t_asm, ctx->t_bc, ctx->t_solve,
                        ctx->t_spmv[0], (double)ctx->n_spmv[0], ctx->spmv_bytes[0],
                        ctx->t_spmv[1], (double)ctx->n_spmv[1], ctx->spmv_bytes[1], (double)ctx->n_halo,
                        ctx->t_spmv_fine, (double)ctx->n_spmv_fine, ctx->spmv_bytes_fine, (double)ctx->n_split,
                        (ctx->ell_ok && ctx->S11.sym) ? 1.0 : 0.0, (double)ctx->max_split_partials,
                        (double)dn, (double)(ctx->D11.tried ? ctx->D11.ncls : 0), (double)(ctx->D11.on ? dst[1] : ctx->D11.status),
                        ctx->t_dict_build, (double)ctx->n_dict_build, zc11 ? 1.0 : 0.0};
  for (int i = 0; i < n && i < 23; ++i) out[i] = v[i];
  return PPH_OK;
}

}  // extern "C"

// ---- bandwidth calibration (no reference counterpart; used by tools/ and DESIGN.md) -------------------

__global__ __launch_bounds__(256) void k_bw_read(const double2* __restrict__ a, int64_t n2, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 1.2345e300) out[0] = s;  // keeps the loads alive
}
__global__ __launch_bounds__(256) void k_bw_copy(const double2* __restrict__ a, double2* __restrict__ b, int64_t n2) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
    b[i] = a[i];
}

// mode 2: the SpMV's stream mix without any row logic: per lane and step 32 B of `a` (two 16-byte loads at a
// 32-byte lane stride, like the values) and 16 B of `b` (like the columns)
__global__ __launch_bounds__(256) void k_bw_mix(const double2* __restrict__ a, const int4* __restrict__ b, int64_t n4,
                                                double* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 v0 = a[2 * i], v1 = a[2 * i + 1];
    const int4 c = b[i];
    s += v0.x + v0.y + v1.x + v1.y + (double)(c.x + c.y + c.z + c.w);
  }
  if (s == 1.2345e300) out[0] = s;
}

extern "C" int pph_bw_probe(pph_ctx* ctx, int64_t bytes, int mode, int blocks, double* ms_out) {
  if (!ctx || !ms_out || bytes < 4096) return PPH_ERR_INVALID;
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf<double> a, b;
  const size_t n = (size_t)(bytes / 8);
  PPH_TRY(a.alloc(ctx, n));
  if (mode == 1) PPH_TRY(b.alloc(ctx, n));
  if (mode == 2) PPH_TRY(b.alloc(ctx, n / 2 + 8));
  PPH_HIP(ctx, hipMemsetAsync(a.p, 0, n * 8, ctx->stream));
  const int reps = 10;
  for (int it = 0; it < 2 + reps; ++it) {
    if (it == 2) PPH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (mode == 2)
      hipLaunchKernelGGL(k_bw_mix, dim3(blocks), dim3(256), 0, ctx->stream, (const double2*)a.p, (const int4*)b.p, (int64_t)(n / 4), ctx->scal.p);
    else if (mode == 0)
      hipLaunchKernelGGL(k_bw_read, dim3(blocks), dim3(256), 0, ctx->stream, (const double2*)a.p, (int64_t)(n / 2), ctx->scal.p);
    else
      hipLaunchKernelGGL(k_bw_copy, dim3(blocks), dim3(256), 0, ctx->stream, (const double2*)a.p, (double2*)b.p, (int64_t)(n / 2));
  }
  PPH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  PPH_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  PPH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *ms_out = ms / reps;
  a.release();
  b.release();
  return PPH_OK;
}
