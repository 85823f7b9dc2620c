// Structured unit-square / unit-cube mesh, cell->dof map and closed-form CSR sparsity pattern.
//
// Replaces what Firedrake's UnitSquareMesh / UnitCubeMesh + DMPlex + FunctionSpace("CG",1) do for the
// reference (src/perphil/mesh/builtin.py:20, src/perphil/forms/spaces.py:34-35,
// src/perphil/experiments/petsc_profiling_3d.py:31).  Numbering is lexicographic inside the local
// box: node (i,j,k) -> i + px*(j + py*k); see include/perphil_hip.h.
#include "pph_internal.h"

// ------------------------------------------------------------------------------------------------
// coordinates + cell->dof map
// ------------------------------------------------------------------------------------------------
__global__ void k_coords(double* __restrict__ cx, double* __restrict__ cy, double* __restrict__ cz,
                         int px, int py, int64_t n, int nx, int ny, int nz, int z0, int dim) {
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < n;
       id += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(id % px);
    int64_t t = id / px;
    int j = (int)(t % py);
    int k = (int)(t / py);
    cx[id] = (double)i / (double)nx;
    cy[id] = (double)j / (double)ny;
    if (dim == 3) cz[id] = (double)(z0 + k) / (double)nz;
  }
}

// one thread per square / cube; writes 1 (quad, hex), 2 (tri) or 6 (tet) cells
__global__ void k_dofmap(int32_t* __restrict__ cells, int kind, int nx, int ny, int nzl, int px, int py) {
  int64_t nbox = (int64_t)nx * ny * (nzl > 0 ? nzl : 1);
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < nbox;
       b += (int64_t)gridDim.x * blockDim.x) {
    int ci = (int)(b % nx);
    int64_t t = b / nx;
    int cj = (int)(t % ny);
    int ck = (int)(t / ny);
    int32_t v0 = (int32_t)(ci + (int64_t)px * (cj + (int64_t)py * ck));
    int32_t dx = 1, dy = px, dz = px * py;
    if (kind == PPH_CELL_QUAD) {
      int32_t* c = cells + 4 * b;
      c[0] = v0; c[1] = v0 + dx; c[2] = v0 + dy; c[3] = v0 + dx + dy;
    } else if (kind == PPH_CELL_TRI) {
      // "left" diagonal: joins (i+1,j) and (i,j+1)
      int32_t* c = cells + 6 * b;
      c[0] = v0; c[1] = v0 + dx; c[2] = v0 + dy;
      c[3] = v0 + dx; c[4] = v0 + dx + dy; c[5] = v0 + dy;
    } else if (kind == PPH_CELL_HEX) {
      int32_t* c = cells + 8 * b;
      c[0] = v0;      c[1] = v0 + dx;      c[2] = v0 + dy;      c[3] = v0 + dx + dy;
      c[4] = v0 + dz; c[5] = v0 + dx + dz; c[6] = v0 + dy + dz; c[7] = v0 + dx + dy + dz;
    } else {
      // six Kuhn tets sharing the diagonal v0-v7
      int32_t v[8] = {v0, v0 + dx, v0 + dy, v0 + dx + dy, v0 + dz, v0 + dx + dz, v0 + dy + dz, v0 + dx + dy + dz};
      const int T[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
      int32_t* c = cells + 24 * b;
#pragma unroll
      for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int a = 0; a < 4; ++a) c[4 * s + a] = v[T[s][a]];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// sparsity pattern: fixed stencil per cell kind, listed in ascending (dz,dy,dx) order so that the
// columns of a row come out sorted
// ------------------------------------------------------------------------------------------------
__global__ void k_row_count(int32_t* __restrict__ cnt, Stencil st, int px, int py, int pz, int64_t n) {
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < n;
       id += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(id % px);
    int64_t t = id / px;
    int j = (int)(t % py);
    int k = (int)(t / py);
    int c = 0;
    for (int s = 0; s < st.count; ++s) {
      int ii = i + st.d[s][0], jj = j + st.d[s][1], kk = k + st.d[s][2];
      c += (ii >= 0 && ii < px && jj >= 0 && jj < py && kk >= 0 && kk < pz) ? 1 : 0;
    }
    cnt[id] = c;
  }
}

__global__ void k_row_fill(int32_t* __restrict__ col, const int64_t* __restrict__ rowptr, Stencil st, int px,
                           int py, int pz, int64_t n) {
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < n;
       id += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(id % px);
    int64_t t = id / px;
    int j = (int)(t % py);
    int k = (int)(t / py);
    int64_t o = rowptr[id];
    for (int s = 0; s < st.count; ++s) {
      int ii = i + st.d[s][0], jj = j + st.d[s][1], kk = k + st.d[s][2];
      if (ii >= 0 && ii < px && jj >= 0 && jj < py && kk >= 0 && kk < pz)
        col[o++] = (int32_t)(ii + (int64_t)px * (jj + (int64_t)py * kk));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// exclusive scan int32 counts -> int64 row pointers (three passes; SCAN_TILE items per block)
// ------------------------------------------------------------------------------------------------
#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ inline int64_t block_exclusive_scan(int64_t v, int64_t* lds, int64_t* total) {
  // inclusive scan inside the wave, then across the 4 waves
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    int64_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int64_t woff = 0, tot = 0;
  for (int q = 0; q < SCAN_THREADS / 64; ++q) {
    int64_t s = lds[q];
    if (q < w) woff += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return woff + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(const int32_t* __restrict__ cnt,
                                                             int64_t* __restrict__ out,
                                                             int64_t* __restrict__ tile_sums, int64_t n) {
  __shared__ int64_t lds[SCAN_THREADS / 64];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int64_t loc[SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int q = 0; q < SCAN_ITEMS; ++q) {
    int64_t id = base + q;
    int64_t c = (id < n) ? (int64_t)cnt[id] : 0;
    loc[q] = s;
    s += c;
  }
  int64_t total;
  int64_t off = block_exclusive_scan(s, lds, &total);
#pragma unroll
  for (int q = 0; q < SCAN_ITEMS; ++q) {
    int64_t id = base + q;
    if (id < n) out[id] = off + loc[q];
  }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: exclusive scan of the tile sums in place; writes the grand total to *total_out
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(int64_t* __restrict__ tile_sums, int64_t ntiles,
                                                            int64_t* __restrict__ total_out) {
  __shared__ int64_t lds[SCAN_THREADS / 64];
  int64_t carry = 0;
  for (int64_t start = 0; start < ntiles; start += SCAN_THREADS) {
    int64_t id = start + threadIdx.x;
    int64_t v = (id < ntiles) ? tile_sums[id] : 0;
    int64_t total;
    int64_t ex = block_exclusive_scan(v, lds, &total);
    if (id < ntiles) tile_sums[id] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ void k_scan_add(int64_t* __restrict__ out, const int64_t* __restrict__ tile_sums, int64_t n,
                           const int64_t* __restrict__ total) {
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id <= n;
       id += (int64_t)gridDim.x * blockDim.x) {
    if (id < n) out[id] += tile_sums[id / SCAN_TILE];
    else out[n] = *total;
  }
}

static int grid_for(int64_t n, int threads = 256, int max_blocks = 2048) {
  int64_t b = ceil_div64(n, threads);
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

int pph_launch_pattern(pph_ctx* ctx, int dim, int kind, int px, int py, int pz, DevBuf<int64_t>& rowptr,
                       DevBuf<int32_t>& col, int64_t* nnz_out) {
  (void)dim;
  int64_t n = (int64_t)px * py * pz;
  Stencil st = make_stencil(kind);
  DevBuf<int32_t> cnt;
  DevBuf<int64_t> sums;
  int64_t ntiles = ceil_div64(n, SCAN_TILE);
  PPH_TRY(cnt.alloc(ctx, (size_t)n));
  PPH_TRY(sums.alloc(ctx, (size_t)ntiles + 1));
  PPH_TRY(rowptr.alloc(ctx, (size_t)n + 1));
  hipLaunchKernelGGL(k_row_count, dim3(grid_for(n)), dim3(256), 0, ctx->stream, cnt.p, st, px, py, pz, n);
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, ctx->stream, cnt.p, rowptr.p,
                     sums.p, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, sums.p, ntiles, sums.p + ntiles);
  hipLaunchKernelGGL(k_scan_add, dim3(grid_for(n + 1)), dim3(256), 0, ctx->stream, rowptr.p, sums.p, n,
                     sums.p + ntiles);
  int64_t nnz = 0;
  PPH_HIP(ctx, hipMemcpyAsync(&nnz, sums.p + ntiles, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PPH_REQUIRE(ctx, nnz > 0 && nnz < (int64_t)2147483647, "scalar block nnz %lld outside int32 column range",
              (long long)nnz);
  PPH_TRY(col.alloc(ctx, (size_t)nnz));
  hipLaunchKernelGGL(k_row_fill, dim3(grid_for(n)), dim3(256), 0, ctx->stream, col.p, rowptr.p, st, px, py, pz, n);
  PPH_HIP(ctx, hipGetLastError());
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  cnt.release();
  sums.release();
  *nnz_out = nnz;
  return PPH_OK;
}

// fills the derived sizes of `mesh` from (dim, kind, nx, ny, nz, z0, nzl) and builds coordinates,
// cell->dof map and the scalar CSR pattern on the device
int pph_launch_mesh(pph_ctx* ctx, MeshData& mesh) {
  mesh.m = (mesh.kind == PPH_CELL_QUAD) ? 4 : (mesh.kind == PPH_CELL_TRI) ? 3 : (mesh.kind == PPH_CELL_HEX) ? 8 : 4;
  mesh.max_row = (mesh.kind == PPH_CELL_QUAD) ? 9 : (mesh.kind == PPH_CELL_TRI) ? 7 : (mesh.kind == PPH_CELL_HEX) ? 27 : 15;
  mesh.px = mesh.nx + 1;
  mesh.py = mesh.ny + 1;
  mesh.pzl = (mesh.dim == 3) ? mesh.nzl + 1 : 1;
  mesh.n = (int64_t)mesh.px * mesh.py * mesh.pzl;
  const int64_t nbox = (int64_t)mesh.nx * mesh.ny * (mesh.dim == 3 ? mesh.nzl : 1);
  mesh.ncell = nbox * ((mesh.kind == PPH_CELL_TRI) ? 2 : (mesh.kind == PPH_CELL_TET) ? 6 : 1);
  PPH_REQUIRE(ctx, mesh.n < (int64_t)1073741823, "mesh has %lld nodes per field: beyond the int32 dof range",
              (long long)mesh.n);
  const int64_t n = mesh.n;
  PPH_TRY(mesh.cx.alloc(ctx, (size_t)n));
  PPH_TRY(mesh.cy.alloc(ctx, (size_t)n));
  PPH_TRY(mesh.cz.alloc(ctx, mesh.dim == 3 ? (size_t)n : 1));
  PPH_TRY(mesh.cells.alloc(ctx, (size_t)mesh.ncell * mesh.m));
  hipLaunchKernelGGL(k_coords, dim3(grid_for(n)), dim3(256), 0, ctx->stream, mesh.cx.p, mesh.cy.p, mesh.cz.p,
                     mesh.px, mesh.py, n, mesh.nx, mesh.ny, mesh.nz > 0 ? mesh.nz : 1, mesh.z0, mesh.dim);
  hipLaunchKernelGGL(k_dofmap, dim3(grid_for(nbox)), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.kind, mesh.nx,
                     mesh.ny, mesh.dim == 3 ? mesh.nzl : 0, mesh.px, mesh.py);
  PPH_HIP(ctx, hipGetLastError());
  // The CSR pattern (int64 row pointers + int32 columns: 1.8 GB at 256^3) is built when something asks for it
  // (pph_ensure_pattern: CSR exports, the monolithic system, ILU(0), the two-step path, op_format 0); the default path -
  // fused assembly into stencil-ELL operators, products and multigrid on them - never reads it.  Its size in closed
  // form: every stencil offset (dx,dy,dz) contributes the nodes whose neighbour lies inside the box.
  {
    const Stencil st = make_stencil(mesh.kind);
    int64_t nnz = 0;
    for (int q = 0; q < st.count; ++q) {
      const int64_t ax = mesh.px - (st.d[q][0] < 0 ? -st.d[q][0] : st.d[q][0]);
      const int64_t ay = mesh.py - (st.d[q][1] < 0 ? -st.d[q][1] : st.d[q][1]);
      const int64_t az = mesh.pzl - (st.d[q][2] < 0 ? -st.d[q][2] : st.d[q][2]);
      if (ax > 0 && ay > 0 && az > 0) nnz += ax * ay * az;
    }
    mesh.nnzb = nnz;
    mesh.pattern_ok = false;
    mesh.rowptr.release();
    mesh.col.release();
  }
  PPH_TRY(pph_mesh_check_affine(ctx, mesh));
  return PPH_OK;
}

// scalar CSR pattern of `mesh` on demand (see pph_launch_mesh)
int pph_ensure_pattern(pph_ctx* ctx, MeshData& mesh) {
  if (mesh.pattern_ok) return PPH_OK;
  PPH_REQUIRE(ctx, mesh.nnzb > 0 && mesh.nnzb < (int64_t)2147483647,
              "scalar block of %lld entries: beyond the int32 positions of the CSR pattern (the stencil-ELL path has no such limit)",
              (long long)mesh.nnzb);
  int64_t nnz = 0;
  PPH_TRY(pph_launch_pattern(ctx, mesh.dim, mesh.kind, mesh.px, mesh.py, mesh.pzl, mesh.rowptr, mesh.col, &nnz));
  PPH_REQUIRE(ctx, nnz == mesh.nnzb, "CSR pattern holds %lld entries, closed form says %lld", (long long)nnz, (long long)mesh.nnzb);
  mesh.pattern_ok = true;
  return PPH_OK;
}
