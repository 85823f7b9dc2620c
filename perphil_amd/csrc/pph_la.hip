// K-SPMV / K-DOT / K-AXPY / K-MDOT / K-MAXPY / K-JAC: fp64 CSR SpMV and the BLAS-1 kernels of the
// Krylov loops.  Replaces PETSc MatMult (seqaij), VecDot/VecMDot/VecAXPY/VecMAXPY and PCApply
// (jacobi) that run inside KSPSolve behind reference src/perphil/solvers/solver.py:71.
//
// All kernels are HBM-bound streaming kernels: 64-wide wavefront shuffle reductions, grid-stride
// loops over at most 2048 workgroups, reductions finished by a second tiny kernel in a fixed order
// (deterministic results, no float atomics).
#include "pph_internal.h"

#define RED_BLOCKS 1024        // grid of the BLAS-1 reduction kernels
#define SPMV_MAX_BLOCKS 2048   // grid cap of the persistent SpMV kernel (multiple of 8 XCDs)
#define PART_STRIDE 2048       // partial sums per reduction slot
#define PART_SLOTS 32          // concurrent reduction slots (GMRES restart 30 + 2)

static inline double* partials(pph_ctx* ctx) { return ctx->scal.p + PPH_MAX_SCAL; }

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the 256-thread workgroup; result valid in thread 0
__device__ inline double block_sum(double v, double* lds) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  if (w == 0) {
    v = (lane < (int)(blockDim.x >> 6)) ? lds[lane] : 0.0;
    v = wave_sum(v);
  }
  return v;
}

// ------------------------------------------------------------------------------------------------
// CSR SpMV, G lanes per row, persistent workgroups, XCD-aware chunk order:
// workgroups with equal (blockIdx % 8) share an XCD (and its 4 MiB L2), so each XCD walks one
// contiguous eighth of the rows and the x entries its rows gather stay in that XCD's L2.
// ------------------------------------------------------------------------------------------------
template <int G, bool DOT>
__global__ __launch_bounds__(256) void k_spmv(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                              const double* __restrict__ val, const double* __restrict__ x,
                                              double* __restrict__ y, int64_t nrows, double* __restrict__ part) {
  constexpr int RPB = 256 / G;  // rows per workgroup per step
  __shared__ double lds[4];
  const int sub = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  const int64_t nchunks = (nrows + RPB - 1) / RPB;
  const int xcd = blockIdx.x & 7;
  const int bx = blockIdx.x >> 3;
  const int bpx = gridDim.x >> 3;  // launcher keeps gridDim.x a multiple of 8
  const int64_t cpx = (nchunks + 7) >> 3;
  const int64_t c_begin = (int64_t)xcd * cpx;
  const int64_t c_end = (c_begin + cpx < nchunks) ? c_begin + cpx : nchunks;
  double acc = 0.0;
  for (int64_t ch = c_begin + bx; ch < c_end; ch += bpx) {
    const int64_t row = ch * RPB + grp;
    double sum = 0.0;
    if (row < nrows) {
      const int64_t s = rowptr[row], e = rowptr[row + 1];
      for (int64_t k = s + sub; k < e; k += G) sum += val[k] * x[col[k]];
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, G);
    if (sub == 0 && row < nrows) {
      y[row] = sum;
      if (DOT) acc += sum * x[row];
    }
  }
  if (DOT) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

__global__ __launch_bounds__(256) void k_reduce_final(const double* __restrict__ part, int nblocks,
                                                      double* __restrict__ out) {
  __shared__ double lds[4];
  const double* p = part + (int64_t)blockIdx.x * PART_STRIDE;
  double v = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) v += p[i];
  v = block_sum(v, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

static int spmv_grid(int64_t nrows, int G) {
  int64_t nchunks = ceil_div64(nrows, 256 / G);
  int64_t g = nchunks < SPMV_MAX_BLOCKS ? nchunks : SPMV_MAX_BLOCKS;
  g = ((g + 7) / 8) * 8;
  return (int)g;
}

template <bool DOT>
static void spmv_dispatch(pph_ctx* ctx, const Csr& A, const double* x, double* y, double* part) {
  const int G = A.lanes;
  const int grid = spmv_grid(A.nrows, G);
  const int variant = DOT ? 1 : 0;
  pph_ctx::EvPair* ev = nullptr;
  if (ctx->time_spmv) {
    if (ctx->ev_used == ctx->ev_pool.size()) {
      pph_ctx::EvPair p;
      p.variant = 0;
      if (hipEventCreate(&p.e0) == hipSuccess && hipEventCreate(&p.e1) == hipSuccess) ctx->ev_pool.push_back(p);
    }
    if (ctx->ev_used < ctx->ev_pool.size()) {
      ev = &ctx->ev_pool[ctx->ev_used++];
      ev->variant = variant;
      (void)hipEventRecord(ev->e0, ctx->stream);
    }
  }
#define PPH_SPMV_CASE(GG)                                                                                    \
  case GG:                                                                                                   \
    hipLaunchKernelGGL((k_spmv<GG, DOT>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, y, \
                       A.nrows, part);                                                                       \
    break;
  switch (G) {
    PPH_SPMV_CASE(4)
    PPH_SPMV_CASE(8)
    PPH_SPMV_CASE(16)
    PPH_SPMV_CASE(32)
    PPH_SPMV_CASE(64)
    default:
      hipLaunchKernelGGL((k_spmv<8, DOT>), dim3(spmv_grid(A.nrows, 8)), dim3(256), 0, ctx->stream, A.rowptr, A.col,
                         A.val, x, y, A.nrows, part);
  }
#undef PPH_SPMV_CASE
  if (ev) (void)hipEventRecord(ev->e1, ctx->stream);
  ctx->n_spmv[variant]++;
  ctx->spmv_bytes[variant] += 12.0 * (double)A.nnz + 20.0 * (double)A.nrows;
}

void la_harvest_spmv_times(pph_ctx* ctx) {
  if (ctx->ev_used == 0) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_pool[i].e0, ctx->ev_pool[i].e1) == hipSuccess)
      ctx->t_spmv[ctx->ev_pool[i].variant] += ms;
  }
  ctx->ev_used = 0;
}

void la_reset_spmv_stats(pph_ctx* ctx) {
  la_harvest_spmv_times(ctx);
  for (int v = 0; v < 2; ++v) { ctx->t_spmv[v] = 0; ctx->spmv_bytes[v] = 0; ctx->n_spmv[v] = 0; }
}


void la_spmv(pph_ctx* ctx, const Csr& A, const double* x, double* y) {
  spmv_dispatch<false>(ctx, A, x, y, nullptr);
}

void la_spmv_dot(pph_ctx* ctx, const Csr& A, const double* x, double* y, int slot) {
  double* part = partials(ctx);
  spmv_dispatch<true>(ctx, A, x, y, part);
  const int grid = spmv_grid(A.nrows, A.lanes);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot);
}

// ------------------------------------------------------------------------------------------------
// BLAS-1
// ------------------------------------------------------------------------------------------------
static inline int ew_grid(int64_t n) {
  int64_t b = ceil_div64(n, 256 * 2);
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

#define EW_LOOP(i, n) \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void k_set(double* __restrict__ x, double v, int64_t n) { EW_LOOP(i, n) x[i] = v; }
__global__ void k_copy(double* __restrict__ d, const double* __restrict__ s, int64_t n) { EW_LOOP(i, n) d[i] = s[i]; }
__global__ void k_axpy(double* __restrict__ y, double a, const double* __restrict__ x, int64_t n) {
  EW_LOOP(i, n) y[i] += a * x[i];
}
__global__ void k_axpby(double* __restrict__ y, double a, const double* __restrict__ x, double b, int64_t n) {
  EW_LOOP(i, n) y[i] = a * x[i] + b * y[i];
}
__global__ void k_scale(double* __restrict__ y, double a, int64_t n) { EW_LOOP(i, n) y[i] *= a; }
__global__ void k_pmult(double* __restrict__ z, const double* __restrict__ d, const double* __restrict__ r,
                        int64_t n) {
  EW_LOOP(i, n) z[i] = d[i] * r[i];
}
__global__ void k_sub(double* __restrict__ z, const double* __restrict__ a, const double* __restrict__ b, int64_t n) {
  EW_LOOP(i, n) z[i] = a[i] - b[i];
}
__global__ void k_block2(double* __restrict__ z, const double* __restrict__ binv, const double* __restrict__ r,
                         int64_t n) {
  EW_LOOP(i, n) {
    const double r1 = r[i], r2 = r[n + i];
    z[i] = binv[i] * r1 + binv[n + i] * r2;
    z[n + i] = binv[2 * n + i] * r1 + binv[3 * n + i] * r2;
  }
}

void la_set(pph_ctx* ctx, double* x, double v, int64_t n) {
  hipLaunchKernelGGL(k_set, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, x, v, n);
}
void la_copy(pph_ctx* ctx, double* dst, const double* src, int64_t n) {
  hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream);
}
void la_axpy(pph_ctx* ctx, double* y, double alpha, const double* x, int64_t n) {
  hipLaunchKernelGGL(k_axpy, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, y, alpha, x, n);
}
void la_axpby(pph_ctx* ctx, double* y, double alpha, const double* x, double beta, int64_t n) {
  hipLaunchKernelGGL(k_axpby, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, y, alpha, x, beta, n);
}
void la_scale(pph_ctx* ctx, double* y, double alpha, int64_t n) {
  hipLaunchKernelGGL(k_scale, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, y, alpha, n);
}
void la_pointwise_mult(pph_ctx* ctx, double* z, const double* d, const double* r, int64_t n) {
  hipLaunchKernelGGL(k_pmult, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, z, d, r, n);
}
void la_sub(pph_ctx* ctx, double* z, const double* a, const double* b, int64_t n) {
  hipLaunchKernelGGL(k_sub, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, z, a, b, n);
}
void la_block2_apply(pph_ctx* ctx, double* z, const double* binv, const double* r, int64_t n) {
  hipLaunchKernelGGL(k_block2, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, z, binv, r, n);
}

// ---- reductions ---------------------------------------------------------------------------------
// k dots against one vector in a single pass over w: out[i] = V_i . w
template <int KB>
__global__ __launch_bounds__(256) void k_mdot(const double* __restrict__ V, int64_t ld, int k0,
                                              const double* __restrict__ w, int64_t n, double* __restrict__ part) {
  __shared__ double lds[4];
  double acc[KB];
#pragma unroll
  for (int q = 0; q < KB; ++q) acc[q] = 0.0;
  EW_LOOP(i, n) {
    const double wi = w[i];
#pragma unroll
    for (int q = 0; q < KB; ++q) acc[q] += V[(int64_t)(k0 + q) * ld + i] * wi;
  }
#pragma unroll
  for (int q = 0; q < KB; ++q) {
    const double s = block_sum(acc[q], lds);
    if (threadIdx.x == 0) part[(int64_t)(k0 + q) * PART_STRIDE + blockIdx.x] = s;
  }
}

void la_mdot(pph_ctx* ctx, const double* V, int64_t ld, int k, const double* w, int64_t n, int slot) {
  double* part = partials(ctx);
  int grid = ew_grid(n);
  if (grid > RED_BLOCKS) grid = RED_BLOCKS;
  int k0 = 0;
  while (k0 < k) {
    const int rem = k - k0;
    if (rem >= 4) {
      hipLaunchKernelGGL(k_mdot<4>, dim3(grid), dim3(256), 0, ctx->stream, V, ld, k0, w, n, part);
      k0 += 4;
    } else if (rem >= 2) {
      hipLaunchKernelGGL(k_mdot<2>, dim3(grid), dim3(256), 0, ctx->stream, V, ld, k0, w, n, part);
      k0 += 2;
    } else {
      hipLaunchKernelGGL(k_mdot<1>, dim3(grid), dim3(256), 0, ctx->stream, V, ld, k0, w, n, part);
      k0 += 1;
    }
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(k), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot);
}

void la_dot(pph_ctx* ctx, const double* x, const double* y, int64_t n, int slot) { la_mdot(ctx, x, 0, 1, y, n, slot); }

__global__ __launch_bounds__(256) void k_dot2(const double* __restrict__ x, const double* __restrict__ y,
                                              const double* __restrict__ z, int64_t n, double* __restrict__ part) {
  __shared__ double lds[4];
  double a = 0.0, b = 0.0;
  EW_LOOP(i, n) {
    a += x[i] * y[i];
    const double zi = z[i];
    b += zi * zi;
  }
  a = block_sum(a, lds);
  b = block_sum(b, lds);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = a;
    part[PART_STRIDE + blockIdx.x] = b;
  }
}

void la_dot2(pph_ctx* ctx, const double* x, const double* y, const double* z, int64_t n, int slot) {
  double* part = partials(ctx);
  int grid = ew_grid(n);
  if (grid > RED_BLOCKS) grid = RED_BLOCKS;
  hipLaunchKernelGGL(k_dot2, dim3(grid), dim3(256), 0, ctx->stream, x, y, z, n, part);
  hipLaunchKernelGGL(k_reduce_final, dim3(2), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot);
}

__global__ __launch_bounds__(256) void k_maxpy(double* __restrict__ w, const double* __restrict__ V, int64_t ld,
                                               int k, const double* __restrict__ h, double sign, int64_t n) {
  EW_LOOP(i, n) {
    double s = 0.0;
    for (int q = 0; q < k; ++q) s += h[q] * V[(int64_t)q * ld + i];
    w[i] += sign * s;
  }
}

static void maxpy_impl(pph_ctx* ctx, double* w, const double* V, int64_t ld, int k, const double* h, double sign,
                       int64_t n) {
  // coefficients travel through a device slot region at the end of the result area
  double* dh = ctx->scal.p + (PPH_MAX_SCAL - 64);
  hipMemcpyAsync(dh, h, sizeof(double) * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
  hipLaunchKernelGGL(k_maxpy, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, w, V, ld, k, dh, sign, n);
}
void la_maxpy_neg(pph_ctx* ctx, double* w, const double* V, int64_t ld, int k, const double* h, int64_t n) {
  maxpy_impl(ctx, w, V, ld, k, h, -1.0, n);
}
void la_maxpy(pph_ctx* ctx, double* x, const double* V, int64_t ld, int k, const double* y, int64_t n) {
  maxpy_impl(ctx, x, V, ld, k, y, 1.0, n);
}

// fused CG update: x += alpha p ; r -= alpha q ; z = dinv .* r ; partials of r.z and z.z
__global__ __launch_bounds__(256) void k_cg_update(double* __restrict__ x, double* __restrict__ r,
                                                   double* __restrict__ z, const double* __restrict__ p,
                                                   const double* __restrict__ q, const double* __restrict__ dinv,
                                                   double alpha, int64_t n, double* __restrict__ part) {
  __shared__ double lds[4];
  double a = 0.0, b = 0.0;
  EW_LOOP(i, n) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    const double zi = dinv ? dinv[i] * ri : ri;
    z[i] = zi;
    a += ri * zi;
    b += zi * zi;
  }
  a = block_sum(a, lds);
  b = block_sum(b, lds);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = a;
    part[PART_STRIDE + blockIdx.x] = b;
  }
}

void la_cg_update(pph_ctx* ctx, double* x, double* r, double* z, const double* p, const double* q,
                  const double* dinv, double alpha, int64_t n, int slot) {
  double* part = partials(ctx);
  int grid = ew_grid(n);
  if (grid > RED_BLOCKS) grid = RED_BLOCKS;
  hipLaunchKernelGGL(k_cg_update, dim3(grid), dim3(256), 0, ctx->stream, x, r, z, p, q, dinv, alpha, n, part);
  hipLaunchKernelGGL(k_reduce_final, dim3(2), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot);
}

__global__ void k_diag_inv(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                           const double* __restrict__ val, int64_t n, double* __restrict__ dinv) {
  EW_LOOP(row, n) {
    int64_t lo = rowptr[row], hi = rowptr[row + 1] - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)col[mid] < row) lo = mid + 1; else hi = mid;
    }
    const double d = val[lo];
    dinv[row] = (d != 0.0) ? 1.0 / d : 1.0;
  }
}

void la_extract_diag_inv(pph_ctx* ctx, const Csr& A, double* dinv) {
  hipLaunchKernelGGL(k_diag_inv, dim3(ew_grid(A.nrows)), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, A.nrows,
                     dinv);
}

int la_fetch(pph_ctx* ctx, int slot, int count) {
  PPH_HIP(ctx, hipMemcpyAsync(ctx->h_scal + slot, ctx->scal.p + slot, sizeof(double) * (size_t)count,
                              hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PPH_OK;
}
