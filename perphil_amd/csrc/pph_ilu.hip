// ILU(0) preconditioner on the device: PETSc's pc_type ilu / pc_factor_levels 0 of GMRES_ILU_PARAMS and of the block
// solves of FIELDSPLIT_GMRES_ILU_PARAMS (reference src/perphil/solvers/parameters.py:27, :50-57), i.e. IKJ Gaussian
// elimination restricted to the CSR pattern in the natural row order, then z = U^-1 L^-1 r.
//
// Both the factorisation and the two triangular solves are sequential along the elimination order, but on the
// lexicographically numbered structured meshes of this path a row depends only on rows of SMALLER
// level(i,j,k) = i + 2 j + 4 k  (every lower neighbour (i+dx, j+dy, k+dz) of the 27-point stencil - hence of its
// 15 / 9 / 7-point sub-stencils - has dx + 2 dy + 4 dz <= -1; in the field-major monolithic system all rows of
// field 0 precede those of field 1).  Rows of one level are independent: one kernel launch per level, levels in
// ascending order for the factorisation and the L solve, descending for the U solve; the launch sequence of one
// preconditioner application is replayed from a captured hipGraph.  Same arithmetic, in the same order per row, as
// the sequential algorithm (oracle/dpp_oracle.py: ilu0): the factors are identical up to rounding, so iteration
// counts equal the oracle's (and the reference's where its numbering happens to give the same factors: 2D Q1).
#include "pph_internal.h"
#include <algorithm>

__global__ __launch_bounds__(256) void k_ilu_levels(int32_t* __restrict__ lev, int64_t nrows, int64_t n, int px, int py,
                                                    int dim, int fieldoff) {
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < nrows; row += (int64_t)gridDim.x * blockDim.x) {
    const int64_t node = row % n;
    const int f = (int)(row / n);
    const int i = (int)(node % px);
    const int64_t t = node / px;
    const int j = (int)(t % py), k = (int)(t / py);
    lev[row] = f * fieldoff + i + 2 * j + ((dim == 3) ? 4 * k : 0);
  }
}

__global__ __launch_bounds__(256) void k_ilu_diag(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  int64_t nrows, int64_t* __restrict__ diag) {
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < nrows; row += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = rowptr[row], hi = rowptr[row + 1] - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)col[mid] < row) lo = mid + 1; else hi = mid;
    }
    diag[row] = lo;   // the pattern of this path always holds the diagonal
  }
}

// rows perm[lo .. hi) of one level: eliminate with the (final) rows of lower levels
__global__ __launch_bounds__(128) void k_ilu_factor(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    double* lu, const int64_t* __restrict__ diag,
                                                    const int32_t* __restrict__ perm, int64_t lo, int64_t hi) {
  for (int64_t t = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < hi; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = perm[t];
    const int64_t rs = rowptr[r], re = rowptr[r + 1], rd = diag[r];
    for (int64_t kk = rs; kk < rd; ++kk) {
      const int64_t c = col[kk];
      const double dc = lu[diag[c]];
      if (dc == 0.0) continue;             // empty (ghost) row: nothing to eliminate with
      const double piv = lu[kk] / dc;
      lu[kk] = piv;
      if (piv == 0.0) continue;
      // row c beyond its diagonal against row r beyond kk: both sorted - merge
      int64_t jr = kk + 1;
      for (int64_t jc = diag[c] + 1; jc < rowptr[c + 1]; ++jc) {
        const int32_t cj = col[jc];
        while (jr < re && col[jr] < cj) ++jr;
        if (jr == re) break;
        if (col[jr] == cj) lu[jr] -= piv * lu[jc];
      }
    }
  }
}

// y = L^-1 b on the rows of one level (unit lower triangle)
__global__ __launch_bounds__(128) void k_ilu_lsolve(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const double* __restrict__ lu, const int64_t* __restrict__ diag,
                                                    const int32_t* __restrict__ perm, int64_t lo, int64_t hi,
                                                    const double* __restrict__ b, double* y) {
  // (grid-stride: a level wider than the capped grid is still swept completely; rows of one level are independent)
  for (int64_t t = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < hi; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = perm[t];
    double s = b[r];
    for (int64_t kk = rowptr[r]; kk < diag[r]; ++kk) s -= lu[kk] * y[col[kk]];
    y[r] = s;
  }
}

// x = U^-1 y on the rows of one level (x and y may be the same vector)
__global__ __launch_bounds__(128) void k_ilu_usolve(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const double* __restrict__ lu, const int64_t* __restrict__ diag,
                                                    const int32_t* __restrict__ perm, int64_t lo, int64_t hi, double* x) {
  for (int64_t t = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < hi; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = perm[t];
    double s = x[r];
    const int64_t d = diag[r];
    for (int64_t kk = d + 1; kk < rowptr[r + 1]; ++kk) s -= lu[kk] * x[col[kk]];
    const double dd = lu[d];
    x[r] = (dd != 0.0) ? s / dd : s;       // empty (ghost) rows: identity
  }
}

static inline int ilu_grid(int64_t n, int threads) {
  int64_t b = ceil_div64(n, threads);
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

void ilu_release(IluData& I) {
  I.lu.release(); I.diag.release(); I.perm.release(); I.y.release();
  I.levptr.clear();
  I.struct_ok = false;
  I.valid = false;
}

// level structure of the rows of A (pattern only): perm = rows sorted by level, levptr = offsets of the levels
static int ilu_structure(pph_ctx* ctx, IluData& I, const Csr& A) {
  const MeshData& m = ctx->mesh;
  const int64_t nrows = A.nrows;
  DevBuf<int32_t> lev;
  PPH_TRY(lev.alloc(ctx, (size_t)nrows));
  const int lmax = m.px + 2 * m.py + ((m.dim == 3) ? 4 * m.pzl : 0);
  hipLaunchKernelGGL(k_ilu_levels, dim3(ilu_grid(nrows, 256)), dim3(256), 0, ctx->stream, lev.p, nrows, m.n, m.px, m.py,
                     m.dim, lmax + 1);
  std::vector<int32_t> h((size_t)nrows);
  PPH_HIP(ctx, hipMemcpyAsync(h.data(), lev.p, sizeof(int32_t) * (size_t)nrows, hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  lev.release();
  const int nlev = (int)(nrows / m.n) * (lmax + 1);
  std::vector<int64_t> count((size_t)nlev + 1, 0);
  for (int64_t r = 0; r < nrows; ++r) count[(size_t)h[(size_t)r] + 1]++;
  for (int l = 0; l < nlev; ++l) count[(size_t)l + 1] += count[(size_t)l];
  std::vector<int32_t> perm((size_t)nrows);
  {
    std::vector<int64_t> next(count.begin(), count.end() - 1);
    for (int64_t r = 0; r < nrows; ++r) perm[(size_t)next[(size_t)h[(size_t)r]]++] = (int32_t)r;   // stable: rows ascending within a level
  }
  I.levptr.clear();
  for (int l = 0; l < nlev; ++l)
    if (count[(size_t)l + 1] > count[(size_t)l]) {
      if (I.levptr.empty()) I.levptr.push_back(count[(size_t)l]);
      I.levptr.push_back(count[(size_t)l + 1]);
    }
  PPH_TRY(I.perm.alloc(ctx, (size_t)nrows));
  PPH_HIP(ctx, hipMemcpy(I.perm.p, perm.data(), sizeof(int32_t) * (size_t)nrows, hipMemcpyHostToDevice));
  PPH_TRY(I.diag.alloc(ctx, (size_t)nrows));
  hipLaunchKernelGGL(k_ilu_diag, dim3(ilu_grid(nrows, 256)), dim3(256), 0, ctx->stream, A.rowptr, A.col, nrows, I.diag.p);
  PPH_TRY(I.lu.alloc(ctx, (size_t)A.nnz));
  PPH_TRY(I.y.alloc(ctx, (size_t)nrows));
  I.rowptr = A.rowptr; I.col = A.col; I.nrows = nrows; I.nnz = A.nnz;
  I.struct_ok = true;
  I.epoch++;
  la_release_graphs(ctx);
  return PPH_OK;
}

// factorises the CSR matrix A (values copied; A itself is left alone)
int ilu_factor(pph_ctx* ctx, IluData& I, const Csr& A) {
  PPH_REQUIRE(ctx, ctx->world == 1, "pc_type ilu is a sequential-elimination preconditioner: single context only");
  PPH_REQUIRE(ctx, A.val != nullptr, "ILU(0) needs the CSR values of the operator");
  if (!I.struct_ok || I.rowptr != A.rowptr || I.nrows != A.nrows) PPH_TRY(ilu_structure(ctx, I, A));
  PPH_HIP(ctx, hipMemcpyAsync(I.lu.p, A.val, sizeof(double) * (size_t)A.nnz, hipMemcpyDeviceToDevice, ctx->stream));
  const size_t nl = I.levptr.size() - 1;
  for (size_t l = 0; l < nl; ++l) {
    const int64_t lo = I.levptr[l], hi = I.levptr[l + 1];
    hipLaunchKernelGGL(k_ilu_factor, dim3(ilu_grid(hi - lo, 128)), dim3(128), 0, ctx->stream, I.rowptr, I.col, I.lu.p,
                       I.diag.p, I.perm.p, lo, hi);
  }
  PPH_HIP(ctx, hipGetLastError());
  I.valid = true;
  return PPH_OK;
}

// z = U^-1 L^-1 r
int ilu_apply(pph_ctx* ctx, IluData& I, const double* r, double* z) {
  PPH_REQUIRE(ctx, I.valid, "ILU(0) factors not available");
  const size_t nl = I.levptr.size() - 1;
  auto body = [&]() -> int {
    for (size_t l = 0; l < nl; ++l) {
      const int64_t lo = I.levptr[l], hi = I.levptr[l + 1];
      hipLaunchKernelGGL(k_ilu_lsolve, dim3(ilu_grid(hi - lo, 128)), dim3(128), 0, ctx->stream, I.rowptr, I.col, I.lu.p,
                         I.diag.p, I.perm.p, lo, hi, r, z);
    }
    for (size_t l = nl; l-- > 0;) {
      const int64_t lo = I.levptr[l], hi = I.levptr[l + 1];
      hipLaunchKernelGGL(k_ilu_usolve, dim3(ilu_grid(hi - lo, 128)), dim3(128), 0, ctx->stream, I.rowptr, I.col, I.lu.p,
                         I.diag.p, I.perm.p, lo, hi, z);
    }
    return PPH_OK;
  };
  // thousands of tiny launches per application: replayed from a graph (up to a size a graph still instantiates quickly)
  if (ctx->use_graphs && !ctx->time_spmv && nl <= 4096) {
    GraphKey gk;
    gk.p[0] = I.lu.p; gk.p[1] = r; gk.p[2] = z; gk.p[3] = I.perm.p;
    gk.n = I.nrows; gk.tag = 7000; gk.epoch = I.epoch;
    return la_run_graph(ctx, gk, body, false);
  }
  return body();
}
