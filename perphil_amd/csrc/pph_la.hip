// K-SPMV / K-DOT / K-AXPY / K-MDOT / K-MAXPY / K-JAC: fp64 CSR SpMV and the BLAS-1 kernels of the
// Krylov loops.  Replaces PETSc MatMult (seqaij), VecDot/VecMDot/VecAXPY/VecMAXPY and PCApply
// (jacobi) that run inside KSPSolve behind reference src/perphil/solvers/solver.py:71.
//
// All kernels are HBM-bound streaming kernels: 64-wide wavefront shuffle reductions, grid-stride
// loops over at most 2048 workgroups, reductions finished by a second tiny kernel in a fixed order
// (deterministic results, no float atomics).
#include "pph_internal.h"

#define RED_BLOCKS 1024        // grid of the BLAS-1 reduction kernels
#define SPMV_MAX_BLOCKS 2048   // upper limit of the persistent SpMV grid (multiple of 8 XCDs)
#define SPMV_DEF_BLOCKS 1024   // default grid: 4 workgroups per CU measured fastest (tools/spmv_probe.py)
#define PART_STRIDE PPH_PART_STRIDE   // partial sums per reduction slot (pph_internal.h)
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

// the sum k_reduce_final forms of n partial sums (same order: thread-strided, then block_sum), in EVERY thread of a 256-thread
// block: kernels that only need the scalar a final reduction would produce compute it themselves (one launch less)
__device__ inline double block_total_of_partials(const double* __restrict__ part, int n, double* lds) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) v += part[i];
  v = block_sum(v, lds);
  __syncthreads();
  if (threadIdx.x == 0) lds[0] = v;
  __syncthreads();
  v = lds[0];
  __syncthreads();
  return v;
}

// sum over groups of 8 consecutive lanes with DPP row shifts (no LDS crossbar): lane 0 of each group gets
// the group total
template <int CTRL>
__device__ inline double dpp_row_shl(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double group8_sum(double v) {
  v += dpp_row_shl<0x104>(v);  // row_shl:4
  v += dpp_row_shl<0x102>(v);  // row_shl:2
  v += dpp_row_shl<0x101>(v);  // row_shl:1
  return v;
}

// Aligned-wide CSR-vector variant: every lane reads 4 consecutive non-zeros per step with 16-byte loads
// (one dwordx4 of columns, two dwordx4 of values) from the row start rounded DOWN to a multiple of 4;
// entries outside [rowptr[row], rowptr[row+1]) are masked.  8-byte and 4-byte-per-lane streams reach a
// markedly lower share of the HBM rate than 16-byte-per-lane streams on gfx950.  Device buffers carry
// 64 bytes of slack, so the rounded reads stay inside the allocations.
template <int G, bool DOT, int MODE = 0, typename VT = double>
__global__ __launch_bounds__(256) void k_spmv_wide(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const VT* __restrict__ val, const double* __restrict__ x,
                                                   const double* __restrict__ bvec, double* __restrict__ y,
                                                   int64_t nrows, double* __restrict__ part) {
  constexpr int RPB = 256 / G;
  __shared__ double lds[4];
  const int sub = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  const int64_t nchunks = (nrows + RPB - 1) / RPB;
  const int xcd = blockIdx.x & 7;
  const int bx = blockIdx.x >> 3;
  const int bpx = gridDim.x >> 3;
  const int64_t cpx = (nchunks + 7) >> 3;
  // MODE 2 (experiment): plain grid-stride chunk order instead of one contiguous eighth per XCD
  constexpr bool LINEAR = (MODE >= 2);
  const int64_t c_begin = LINEAR ? 0 : (int64_t)xcd * cpx;
  const int64_t c_end = LINEAR ? nchunks : ((c_begin + cpx < nchunks) ? c_begin + cpx : nchunks);
  const int64_t c_first = LINEAR ? blockIdx.x : c_begin + bx;
  const int64_t c_step = LINEAR ? gridDim.x : bpx;
  double acc = 0.0;
  // row pointers are fetched one chunk ahead, so that a row's matrix loads do not wait for them
  int64_t s = 0, e = 0;
  if (c_first < c_end && c_first * RPB + grp < nrows) {
    s = rowptr[c_first * RPB + grp];
    e = rowptr[c_first * RPB + grp + 1];
  }
  for (int64_t ch = c_first; ch < c_end; ch += c_step) {
    const int64_t row = ch * RPB + grp;
    const int64_t nrow = (ch + c_step) * RPB + grp;
    int64_t ns = 0, ne = 0;
    if (ch + c_step < c_end && nrow < nrows) {
      ns = rowptr[nrow];
      ne = rowptr[nrow + 1];
    }
    double sum = 0.0;
    // operands of the epilogue (b[row] of the residual form, x[row] of the fused p.Ap) are requested together with
    // the row pointers: loading them after the reduction put one more memory latency on every row's critical path
    double bv = 0.0, xr = 0.0;
    if (row < nrows) {
      if (bvec) bv = bvec[row];
      if (DOT) xr = x[row];
      const int32_t safe = 0;  // column used by masked lanes (x[0] is always valid)
      for (int64_t base = (s & ~(int64_t)3) + 4 * sub; base < e; base += 4 * G) {
        int4 c;
        double2 v01, v23;
        if constexpr (sizeof(VT) == 4) {
          // values stored in fp32 (multigrid preconditioner operands): one 16-byte load carries 4 of them;
          // products and sums stay in fp64
          c = *reinterpret_cast<const int4*>(col + base);
          const float4 vf = *reinterpret_cast<const float4*>(val + base);
          v01 = make_double2((double)vf.x, (double)vf.y);
          v23 = make_double2((double)vf.z, (double)vf.w);
        } else if (MODE == 3) {  // non-temporal matrix stream: keep the L2 for x
          typedef int v4i __attribute__((ext_vector_type(4)));
          typedef double v2d __attribute__((ext_vector_type(2)));
          const v4i cc = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(col + base));
          const v2d a01 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(reinterpret_cast<const double*>(val) + base));
          const v2d a23 = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(reinterpret_cast<const double*>(val) + base + 2));
          c = make_int4(cc.x, cc.y, cc.z, cc.w);
          v01 = make_double2(a01.x, a01.y);
          v23 = make_double2(a23.x, a23.y);
        } else {
          c = *reinterpret_cast<const int4*>(col + base);
          v01 = *reinterpret_cast<const double2*>(reinterpret_cast<const double*>(val) + base);
          v23 = *reinterpret_cast<const double2*>(reinterpret_cast<const double*>(val) + base + 2);
        }
        const bool k0 = base >= s, k1 = base + 1 >= s && base + 1 < e, k2 = base + 2 >= s && base + 2 < e,
                   k3 = base + 3 < e && base + 3 >= s;
        double x0, x1, x2, x3;
        if (MODE == 1) {  // experiment: no gather (streams the matrix only); results are wrong on purpose
          x0 = x1 = x2 = x3 = (double)(c.x + c.y + c.z + c.w);
        } else if (MODE == 4) {  // experiment: same 4 gathers, but all addresses inside one 512-byte window of x
          x0 = x[c.x & 63]; x1 = x[c.y & 63]; x2 = x[c.z & 63]; x3 = x[c.w & 63];
        } else if (MODE == 5) {  // experiment: one gather instead of four
          x0 = x[k0 ? c.x : safe]; x1 = x0 + (double)c.y; x2 = x0 + (double)c.z; x3 = x0 + (double)c.w;
        } else {
          x0 = x[k0 ? c.x : safe]; x1 = x[k1 ? c.y : safe]; x2 = x[k2 ? c.z : safe]; x3 = x[k3 ? c.w : safe];
        }
        sum += (k0 ? v01.x : 0.0) * x0 + (k1 ? v01.y : 0.0) * x1 + (k2 ? v23.x : 0.0) * x2 + (k3 ? v23.y : 0.0) * x3;
      }
    }
    if constexpr (G == 8) {
      sum = group8_sum(sum);
    } else {
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, G);
    }
    if (sub == 0 && row < nrows) {
      y[row] = bvec ? bv - sum : sum;
      if (DOT) acc += sum * xr;
    }
    s = ns;
    e = ne;
  }
  if (DOT) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

__global__ __launch_bounds__(256) void k_reduce_final(const double* __restrict__ part, int nblocks,
                                                      double* __restrict__ out, const double* copy_src = nullptr,
                                                      double* copy_dst = nullptr) {
  __shared__ double lds[4];
  if (copy_dst && blockIdx.x == 0 && threadIdx.x == 0) *copy_dst = *copy_src;
  const double* p = part + (int64_t)blockIdx.x * PART_STRIDE;
  double v = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) v += p[i];
  v = block_sum(v, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}

// the same (one block) followed by the publication of `count` scalars from psrc - the result just written among them - to
// the host's mirror, as k_publish does it: one launch less per Krylov iteration (round 4; sum and order of k_reduce_final)
__global__ __launch_bounds__(256) void k_reduce_final_publish(const double* __restrict__ part, int nblocks, double* __restrict__ out,
                                                              const double* psrc, double* __restrict__ hdst, int count,
                                                              unsigned long long* __restrict__ hseq, unsigned long long* ctr) {
  __shared__ double lds[4];
  double v = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) v += part[i];
  v = block_sum(v, lds);
  if (threadIdx.x == 0) {
    out[0] = v;
    for (int i = 0; i < count; ++i) hdst[i] = (psrc + i == out) ? v : psrc[i];
    __threadfence_system();
    const unsigned long long seq = *ctr + 1;
    *ctr = seq;
    __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static int spmv_grid(const pph_ctx* ctx, int64_t nrows, int rows_per_block) {
  const int cap = (ctx->spmv_blocks >= 8 && ctx->spmv_blocks <= SPMV_MAX_BLOCKS) ? (ctx->spmv_blocks / 8) * 8 : SPMV_DEF_BLOCKS;
  int64_t nchunks = ceil_div64(nrows, rows_per_block);
  int64_t g = nchunks < cap ? nchunks : cap;
  g = ((g + 7) / 8) * 8;
  return (int)g;
}

// lanes per row of the reduction phase of the stream kernel: smallest T with 8 T >= max_row
static int stream_T(int max_row) {
  int T = 4;
  while (8 * T < max_row && T < 64) T *= 2;
  return T;
}


// Stencil-ELL product of a slab whose operand needs its ghost planes refreshed.  halo_overlap 0: exchange, then one
// launch.  1 / 2: three launches - the chunks whose rows read no ghost value ("interior": everything but the two node
// planes next to each ghost plane), then the boundary chunks below and above - and with 1 the exchange runs on the
// communication stream WHILE the interior rows are computed: x is complete at ev_x; the exchange waits for ev_x
// only, the boundary launches wait for the exchange (ev_h).  The rows' results do not depend on the split; the
// partial sums of the dot-product modes are laid out launch after launch, so modes 1 and 2 are bit-identical.
// Returns the number of partial sums written.
static int sell_product(pph_ctx* ctx, const Csr& A, int mode, const double* x, const double* b, const double* dinv,
                        const double* w, double* y, double* part, int64_t dlo, int64_t dhi, double* aux, double* z0,
                        bool need_halo) {
  const MeshData* g = A.geom;
  const bool ghosts = g && ctx->world > 1 && (g->glo || g->ghi) && need_halo;
  const bool one_field = g && A.nrows == g->n;
  if (ghosts && one_field && ctx->halo_overlap > 0 && A.nrows >= ctx->halo_overlap_min_rows && ctx->comm_status == PPH_OK) {
    const int64_t CH = 256 * ((ctx->sell_rpt == 1) ? 1 : 2);
    const int64_t nchunks = ceil_div64(A.nrows, CH);
    const int64_t pl = g->plane();
    const int64_t reach = pl + g->px + 2;   // rows beyond its own a chunk reads x from
    const int64_t c_lo = g->glo ? ceil_div64(pl + reach, CH) : 0;                                  // first interior chunk
    const int64_t c_hi = g->ghi ? (A.nrows - pl - reach) / CH : nchunks;                           // one past the last
    if (c_lo < c_hi) {
      const bool overlap = ctx->halo_overlap == 1;
      double* xv = const_cast<double*>(x);
      int total = 0;
      ctx->n_split++;
      if (overlap) {
        (void)hipEventRecord(ctx->ev_x, ctx->stream);
        if (ctx->nccl_comm) {
          (void)hipStreamWaitEvent(ctx->comm_stream, ctx->ev_x, 0);
          (void)la_halo(ctx, *g, xv, ctx->comm_stream);
          (void)hipEventRecord(ctx->ev_h, ctx->comm_stream);
        }
      } else {
        (void)la_halo(ctx, *g, xv);
      }
      // The partial sums of the dot-product modes are laid out launch after launch in ONE slot area of PPH_PART_STRIDE
      // entries (mode 7 keeps two more areas at that stride): the three grids together must stay within it, or the
      // tail of one sum would alias the next slot.  The boundary launches get up to a quarter each, the interior the rest.
      int cap_i = 0, cap_lo = 0, cap_hi = 0;
      if (mode == 2 || mode >= 4) {
        const int pc = ctx->part_cap;
        const int64_t q = (pc / 4) & ~7;
        auto grid8 = [](int64_t chunks, int64_t cap) { const int64_t g = chunks < cap ? chunks : cap; return (int)(((g + 7) / 8) * 8); };
        cap_lo = (int)q; cap_hi = (int)q;
        cap_i = pc - (c_lo > 0 ? grid8(c_lo, q) : 0) - (nchunks - c_hi > 0 ? grid8(nchunks - c_hi, q) : 0);
      }
      total += sell_spmv(ctx, A.ell, A.nrows, mode, x, b, dinv, w, y, part, dlo, dhi, aux, z0, c_lo, c_hi, cap_i);
      if (overlap) {
        if (ctx->nccl_comm) (void)hipStreamWaitEvent(ctx->stream, ctx->ev_h, 0);
        else (void)la_halo(ctx, *g, xv, nullptr, ctx->ev_x);   // callback transport: the host exchanges while the interior rows run
      }
      total += sell_spmv(ctx, A.ell, A.nrows, mode, x, b, dinv, w, y, part + total, dlo, dhi, aux, z0, 0, c_lo, cap_lo);
      total += sell_spmv(ctx, A.ell, A.nrows, mode, x, b, dinv, w, y, part + total, dlo, dhi, aux, z0, c_hi, nchunks, cap_hi);
      if ((mode == 2 || mode >= 4) && total > ctx->part_cap && ctx->comm_status == PPH_OK) {
        // cannot happen with the caps above; if it ever does the sums are wrong: refuse the solve instead of computing on
        ctx->comm_status = PPH_ERR_INVALID;
        ctx->comm_error = "split product wrote more partial sums than one reduction slot holds";
      }
      if ((mode == 2 || mode >= 4) && total > ctx->max_split_partials) ctx->max_split_partials = total;   // (the other modes write no partial sums)
      return total;
    }
  }
  if (g && need_halo) {  // ghost planes of x <- owners (slabs only); field-major mixed vectors carry two fields
    (void)la_halo(ctx, *g, const_cast<double*>(x));
    if (A.nrows == 2 * g->n) (void)la_halo(ctx, *g, const_cast<double*>(x) + g->n);
  }
  return sell_spmv(ctx, A.ell, A.nrows, mode, x, b, dinv, w, y, part, dlo, dhi, aux, z0);
}

// returns the grid used (number of partial sums written when DOT)
// jdinv != null (stencil-ELL operators only): y = x + jw * jdinv .* (bvec - A x), one damped-Jacobi sweep out of place
template <bool DOT>
static int spmv_dispatch(pph_ctx* ctx, const Csr& A, const double* x, const double* bvec, double* y, double* part,
                         const double* jdinv = nullptr, const double* jw = nullptr, bool jdot = false, int64_t dlo = 0,
                         int64_t dhi = 0, bool x_ghosts_valid = false) {
  const int variant = DOT ? 1 : 0;
  if (A.geom && !x_ghosts_valid && !A.ell.val) {  // ghost planes of x <- owners (slabs only); field-major mixed vectors carry two fields
    (void)la_halo(ctx, *A.geom, const_cast<double*>(x));
    if (A.nrows == 2 * A.geom->n) (void)la_halo(ctx, *A.geom, const_cast<double*>(x) + A.geom->n);
  }
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
      ev->fine = A.nrows >= ctx->mesh.n;
      (void)hipEventRecord(ev->e0, ctx->stream);
    }
  }
  int grid = 8;
  double bytes_per_nnz = 12.0;
  if (A.ell.val) {
    // stencil-ELL copy of the operator (pph_sell.hip): 8 B per stored entry, no index arrays
    grid = sell_product(ctx, A, jdinv ? (jdot ? 4 : 3) : (DOT ? (bvec ? 7 : 2) : (bvec ? 1 : 0)), x, bvec, jdinv, jw, y,
                        part ? part : partials(ctx), dlo, dhi, nullptr, nullptr, !x_ghosts_valid);
    if (ev) (void)hipEventRecord(ev->e1, ctx->stream);
    // algorithmic bytes: every stored value once, x read once, y written once, plus the epilogue's vectors
    // (b for the residual form; b and 1 / a_ii for the Jacobi update)
    const double extra = jdinv ? 16.0 : (bvec ? 8.0 : 0.0);
    const double bytes = (sell_stream_bytes(ctx, A.ell) + 16.0 + extra) * (double)A.nrows;
    ctx->n_spmv[variant]++;
    ctx->spmv_bytes[variant] += bytes;
    if (A.nrows >= ctx->mesh.n) { ctx->n_spmv_fine++; ctx->spmv_bytes_fine += bytes; }
    return grid;
  }
  if (A.val32) {
    // fp32-valued operator (multigrid levels): aligned-wide kernel, 8 lanes per row
    const int G = (A.max_row > 0 && A.max_row + 3 <= 16) ? 4 : 8;
    grid = spmv_grid(ctx, A.nrows, 256 / G);
    bytes_per_nnz = 8.0;
    if (G == 4)
      hipLaunchKernelGGL((k_spmv_wide<4, DOT, 2, float>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val32, x,
                         bvec, y, A.nrows, part);
    else
      hipLaunchKernelGGL((k_spmv_wide<8, DOT, 2, float>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val32, x,
                         bvec, y, A.nrows, part);
    if (ev) (void)hipEventRecord(ev->e1, ctx->stream);
    ctx->n_spmv[variant]++;
    ctx->spmv_bytes[variant] += bytes_per_nnz * (double)A.nnz + 20.0 * (double)A.nrows;
    if (A.nrows >= ctx->mesh.n) { ctx->n_spmv_fine++; ctx->spmv_bytes_fine += bytes_per_nnz * (double)A.nnz + 20.0 * (double)A.nrows; }
    return grid;
  }
  {
    // aligned-wide CSR-vector kernel.  Lanes per row: one 4-wide step covers 4 G entries; rows of up to 29 entries fit G = 8
    int G = ctx->spmv_lanes_override > 0 ? ctx->spmv_lanes_override : (A.max_row > 0 && A.max_row + 3 <= 16 ? 4 : 8);
    if (G != 4 && G != 8 && G != 16 && G != 32 && G != 64) G = 8;
    grid = spmv_grid(ctx, A.nrows, 256 / G);
#define PPH_WIDE_CASE(GG)                                                                                       \
  case GG:                                                                                                      \
    hipLaunchKernelGGL((k_spmv_wide<GG, DOT, 2>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, \
                       bvec, y, A.nrows, part);                                                                 \
    break;
    switch (G) {
      PPH_WIDE_CASE(4)
      PPH_WIDE_CASE(8)
      PPH_WIDE_CASE(16)
      PPH_WIDE_CASE(32)
      PPH_WIDE_CASE(64)
    }
#undef PPH_WIDE_CASE
  }
  if (ev) (void)hipEventRecord(ev->e1, ctx->stream);
  ctx->n_spmv[variant]++;
  ctx->spmv_bytes[variant] += 12.0 * (double)A.nnz + 20.0 * (double)A.nrows;
  if (A.nrows >= ctx->mesh.n) { ctx->n_spmv_fine++; ctx->spmv_bytes_fine += 12.0 * (double)A.nnz + 20.0 * (double)A.nrows; }
  return grid;
}

void la_harvest_spmv_times(pph_ctx* ctx) {
  if (ctx->ev_used == 0) return;
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_pool[i].e0, ctx->ev_pool[i].e1) == hipSuccess) {
      const int v = ctx->ev_pool[i].variant;
      if (v >= 2) { ctx->t_comm[v - 2] += ms; ctx->n_comm_timed[v - 2]++; continue; }   // halo exchange / all-reduce (pph_comm.hip)
      ctx->t_spmv[v] += ms;
      if (ctx->ev_pool[i].fine) ctx->t_spmv_fine += ms;
    }
  }
  ctx->ev_used = 0;
}

// launches a pending final reduction the way its producer would have (nobody folded it into a consumer)
void la_flush_final(pph_ctx* ctx) {
  if (!ctx->pend.valid) return;
  const pph_ctx::PendFinal f = ctx->pend;
  ctx->pend.valid = false;
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, f.part, f.n, ctx->scal.p + f.slot,
                     f.copy_src >= 0 ? (const double*)(ctx->scal.p + f.copy_src) : (const double*)nullptr,
                     f.copy_src >= 0 ? ctx->scal.p + f.copy_dst : (double*)nullptr);
}
static inline bool la_can_fold(const pph_ctx* ctx) { return ctx->fold_finals && ctx->world == 1; }

void la_reset_spmv_stats(pph_ctx* ctx) {
  la_harvest_spmv_times(ctx);
  for (int v = 0; v < 2; ++v) { ctx->t_spmv[v] = 0; ctx->spmv_bytes[v] = 0; ctx->n_spmv[v] = 0; }
  ctx->t_spmv_fine = 0; ctx->spmv_bytes_fine = 0; ctx->n_spmv_fine = 0;
  ctx->t_comm[0] = ctx->t_comm[1] = 0; ctx->n_comm_timed[0] = ctx->n_comm_timed[1] = 0;
}


void la_spmv(pph_ctx* ctx, const Csr& A, const double* x, double* y) {
  spmv_dispatch<false>(ctx, A, x, nullptr, y, nullptr);
}

// y = b - A x
void la_spmv_resid(pph_ctx* ctx, const Csr& A, const double* x, const double* b, double* y) {
  spmv_dispatch<false>(ctx, A, x, b, y, nullptr);
}

// y = x + w * dinv .* (b - A x): one damped-Jacobi (one-step Chebyshev) sweep, out of place (y != x); A must carry a
// stencil-ELL copy
void la_spmv_jacobi(pph_ctx* ctx, const Csr& A, const double* x, const double* b, const double* dinv, const double* w,
                    double* y, int dot_slot, int64_t dlo, int64_t dhi, bool x_ghosts_valid) {
  const int grid = spmv_dispatch<false>(ctx, A, x, b, y, dot_slot >= 0 ? partials(ctx) : nullptr, dinv, w, dot_slot >= 0,
                                        dlo, dhi, x_ghosts_valid);
  if (dot_slot >= 0) {
    if (ctx->defer_next_final && la_can_fold(ctx)) {
      ctx->defer_next_final = false;
      ctx->pend.part = partials(ctx); ctx->pend.n = grid; ctx->pend.slot = dot_slot; ctx->pend.copy_src = ctx->pend.copy_dst = -1;
      ctx->pend.valid = true;
      return;
    }
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, partials(ctx), grid, ctx->scal.p + dot_slot);
  }
}

void la_spmv_shift(pph_ctx* ctx, const Csr& A, const double* x, const double* b, double* R, double* told, double* tmp,
                   int slot, int64_t dlo, int64_t dhi, double* z0, const double* dinv0, const double* w0) {
  if (!A.ell.val) {   // (callers offer z0 only for stencil-ELL operators)
    if (b) la_spmv_resid(ctx, A, x, b, tmp); else la_spmv(ctx, A, x, tmp);
    la_shift(ctx, R, told, tmp, b ? 1.0 : -1.0, A.nrows);
    la_dot(ctx, R + dlo, R + dlo, dhi - dlo, slot);
    return;
  }
  double* part = partials(ctx);
  pph_ctx::EvPair* ev = nullptr;
  if (ctx->time_spmv) {   // (as spmv_dispatch: the instrumented step of bench.py times every product)
    if (ctx->ev_used == ctx->ev_pool.size()) {
      pph_ctx::EvPair p;
      p.variant = 0;
      if (hipEventCreate(&p.e0) == hipSuccess && hipEventCreate(&p.e1) == hipSuccess) ctx->ev_pool.push_back(p);
    }
    if (ctx->ev_used < ctx->ev_pool.size()) {
      ev = &ctx->ev_pool[ctx->ev_used++];
      ev->variant = 0;
      ev->fine = A.nrows >= ctx->mesh.n;
      (void)hipEventRecord(ev->e0, ctx->stream);
    }
  }
  const int grid = sell_product(ctx, A, b ? 5 : 6, x, b, z0 ? dinv0 : nullptr, z0 ? w0 : nullptr, told, part, dlo, dhi, R, z0, true);
  if (ev) (void)hipEventRecord(ev->e1, ctx->stream);
  const double bytes = (sell_stream_bytes(ctx, A.ell) + 16.0 + (b ? 8.0 : 0.0) + 24.0 + (z0 ? 16.0 : 0.0)) * (double)A.nrows;
  ctx->n_spmv[0]++;
  ctx->spmv_bytes[0] += bytes;
  if (A.nrows >= ctx->mesh.n) { ctx->n_spmv_fine++; ctx->spmv_bytes_fine += bytes; }
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot);
}

// y = A x with x.y, r.y and y.y -> scal[slot .. slot + 2] (stencil-ELL operators only): what the host needs to form
// the NEXT residual norm of a CG iteration, r.r - 2 alpha r.Ap + alpha^2 Ap.Ap, while the update kernel still runs
void la_spmv_dot3(pph_ctx* ctx, const Csr& A, const double* x, const double* r, double* y, int slot, int copy_src,
                  int copy_dst) {
  double* part = partials(ctx);
  const Seg sg = pph_owned_seg(A.geom, A.nrows);
  const int grid = spmv_dispatch<true>(ctx, A, x, r, y, part, nullptr, nullptr, false, sg.off1, sg.off1 + sg.len1);
  hipLaunchKernelGGL(k_reduce_final, dim3(3), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot,
                     copy_src >= 0 ? (const double*)(ctx->scal.p + copy_src) : (const double*)nullptr,
                     copy_src >= 0 ? ctx->scal.p + copy_dst : (double*)nullptr);
}

void la_spmv_dot(pph_ctx* ctx, const Csr& A, const double* x, double* y, int slot, int copy_src, int copy_dst, bool defer) {
  la_flush_final(ctx);
  double* part = partials(ctx);
  // the sum runs over the owned rows (slabs: ghost rows are empty in CSR / full stencil-ELL storage, but not in symmetric storage)
  const Seg sg = pph_owned_seg(A.geom, A.nrows);
  const bool two = sg.len2 > 0;   // field-major mixed vector: two owned segments - only operators with empty ghost rows get here
  const int grid = spmv_dispatch<true>(ctx, A, x, nullptr, y, part, nullptr, nullptr, false, two ? 0 : sg.off1,
                                       two ? A.nrows : sg.off1 + sg.len1);
  if (defer && la_can_fold(ctx)) {
    ctx->pend.part = part; ctx->pend.n = grid; ctx->pend.slot = slot; ctx->pend.copy_src = copy_src; ctx->pend.copy_dst = copy_dst;
    ctx->pend.valid = true;
    return;
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot,
                     copy_src >= 0 ? (const double*)(ctx->scal.p + copy_src) : (const double*)nullptr,
                     copy_src >= 0 ? ctx->scal.p + copy_dst : (double*)nullptr);
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
  // (a kernel of our own: the runtime's device-to-device copy leaves 35 - 40 us of idle time around each call -
  // profiles/r04_c_gaps256.txt, "after __amd_rocclr_copyBuffer")
  if (n > 0) hipLaunchKernelGGL(k_copy, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, dst, src, n);
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
// R += sign * (tnew - told) ; told = tnew   (residual bookkeeping of the Picard sweeps: one pass instead of
// two axpys and a copy)
__global__ void k_shift(double* __restrict__ R, double* __restrict__ told, const double* __restrict__ tnew, double sign,
                        int64_t n) {
  EW_LOOP(i, n) {
    const double tn = tnew[i];
    R[i] += sign * (tn - told[i]);
    told[i] = tn;
  }
}

void la_shift(pph_ctx* ctx, double* R, double* told, const double* tnew, double sign, int64_t n) {
  hipLaunchKernelGGL(k_shift, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, R, told, tnew, sign, n);
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
                                              const double* __restrict__ w, Seg sg, double* __restrict__ part) {
  __shared__ double lds[4];
  double acc[KB];
#pragma unroll
  for (int q = 0; q < KB; ++q) acc[q] = 0.0;
  const int64_t n = sg.len1 + sg.len2;
  EW_LOOP(ii, n) {
    const int64_t i = ii < sg.len1 ? sg.off1 + ii : sg.off2 + (ii - sg.len1);
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
  Seg sg;
  sg.off1 = 0; sg.len1 = n; sg.off2 = 0; sg.len2 = 0;
  la_mdot_seg(ctx, V, ld, k, w, sg, slot);
}

void la_mdot_seg(pph_ctx* ctx, const double* V, int64_t ld, int k, const double* w, Seg sg, int slot) {
  double* part = partials(ctx);
  const int64_t n = sg.len1 + sg.len2;
  int grid = ew_grid(n);
  if (grid > RED_BLOCKS) grid = RED_BLOCKS;
  int k0 = 0;
  while (k0 < k) {
    const int rem = k - k0;
    if (rem >= 4) {
      hipLaunchKernelGGL(k_mdot<4>, dim3(grid), dim3(256), 0, ctx->stream, V, ld, k0, w, sg, part);
      k0 += 4;
    } else if (rem >= 2) {
      hipLaunchKernelGGL(k_mdot<2>, dim3(grid), dim3(256), 0, ctx->stream, V, ld, k0, w, sg, part);
      k0 += 2;
    } else {
      hipLaunchKernelGGL(k_mdot<1>, dim3(grid), dim3(256), 0, ctx->stream, V, ld, k0, w, sg, part);
      k0 += 1;
    }
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(k), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot);
}

void la_dot(pph_ctx* ctx, const double* x, const double* y, int64_t n, int slot) { la_mdot(ctx, x, 0, 1, y, n, slot); }

__global__ __launch_bounds__(256) void k_dot2(const double* __restrict__ x, const double* __restrict__ y,
                                              const double* __restrict__ z, Seg sg, double* __restrict__ part) {
  __shared__ double lds[4];
  double a = 0.0, b = 0.0;
  const int64_t n = sg.len1 + sg.len2;
  EW_LOOP(ii, n) {
    const int64_t i = ii < sg.len1 ? sg.off1 + ii : sg.off2 + (ii - sg.len1);
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
  Seg sg;
  sg.off1 = 0; sg.len1 = n; sg.off2 = 0; sg.len2 = 0;
  la_dot2_seg(ctx, x, y, z, sg, slot);
}

void la_dot2_seg(pph_ctx* ctx, const double* x, const double* y, const double* z, Seg sg, int slot) {
  double* part = partials(ctx);
  const int64_t n = sg.len1 + sg.len2;
  int grid = ew_grid(n);
  if (grid > RED_BLOCKS) grid = RED_BLOCKS;
  hipLaunchKernelGGL(k_dot2, dim3(grid), dim3(256), 0, ctx->stream, x, y, z, sg, part);
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
  (void)hipMemcpyAsync(dh, h, sizeof(double) * (size_t)k, hipMemcpyHostToDevice, ctx->stream);
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
                                                   double alpha, int64_t n, Seg sg, double* __restrict__ part) {
  __shared__ double lds[4];
  double a = 0.0, b = 0.0;
  EW_LOOP(i, n) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    const double zi = dinv ? dinv[i] * ri : ri;
    if (z) z[i] = zi;   // z == null: only the updates and r.r are wanted (unpreconditioned-norm CG)
    if ((i >= sg.off1 && i < sg.off1 + sg.len1) || (i >= sg.off2 && i < sg.off2 + sg.len2)) {  // owned entries only
      a += ri * zi;
      b += zi * zi;
    }
  }
  a = block_sum(a, lds);
  b = block_sum(b, lds);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = a;
    part[PART_STRIDE + blockIdx.x] = b;
  }
}

void la_cg_update(pph_ctx* ctx, double* x, double* r, double* z, const double* p, const double* q,
                  const double* dinv, double alpha, int64_t n, int slot, Seg sg) {
  double* part = partials(ctx);
  int grid = ew_grid(n);
  if (grid > RED_BLOCKS) grid = RED_BLOCKS;
  hipLaunchKernelGGL(k_cg_update, dim3(grid), dim3(256), 0, ctx->stream, x, r, z, p, q, dinv, alpha, n, sg, part);
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

// ---- reduction scalars that stay on the device ---------------------------------------------------
// A Krylov iteration needs alpha = r.z / p.Ap and beta = r.z_new / r.z only inside the next vector kernels; when
// those read them from ctx->scal, the host has to see one number per iteration (the residual norm for the
// convergence test) instead of three.  Possible on a single context and with the RCCL transport (sums are
// all-reduced on the stream); the callback transport reduces on the host and keeps the host-scalar path.
// (option "device_scalars" = 1 runs the same branch over the callback transport - the all-reduce then stages through
// the host - so that the multi-rank tests on one GPU cover the code the RCCL run executes)
bool la_device_scalars(const pph_ctx* ctx) {
  return ctx->world == 1 || ctx->comm_suspended || ctx->nccl_comm != nullptr || ctx->device_scalars;
}

int la_reduce_device(pph_ctx* ctx, int slot, int count) {
  if (ctx->world > 1 && !ctx->comm_suspended) return comm_allreduce_device(ctx, ctx->scal.p + slot, count);
  return PPH_OK;
}

// Reduction results -> host without a stream synchronisation: a one-wave kernel copies the scalars into the pinned,
// mapped, coherent mirror and then stores a sequence number behind it; the host polls that word.  A D2H copy plus
// hipStreamSynchronize left the GPU idle for about 26 us per fetch (wake-up of the waiting thread + the next launch);
// polling sees the values about 2 us after the kernel wrote them.  Every Krylov iteration fetches once.
__global__ __launch_bounds__(64) void k_publish(const double* __restrict__ src, double* __restrict__ hdst, int count,
                                                unsigned long long* __restrict__ hseq, unsigned long long* ctr) {
  for (int i = threadIdx.x; i < count; i += 64) hdst[i] = src[i];
  __threadfence_system();   // the wave's stores have reached the host before the sequence word follows
  if (threadIdx.x == 0) {
    // the sequence number is counted on the device (one publication at a time on the stream), so that a publication
    // replayed from a captured graph needs no new kernel argument
    const unsigned long long seq = *ctr + 1;
    *ctr = seq;
    __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

void la_publish(pph_ctx* ctx, int slot, int count) {
  la_flush_final(ctx);
  ++ctx->pub_seq;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, ctx->scal.p + slot, ctx->h_scal_dev + slot, count,
                     ctx->h_seq_dev, ctx->pub_ctr.p);
}

int la_wait_published(pph_ctx* ctx) {
  const unsigned long long seq = ctx->pub_seq;
  volatile unsigned long long* p = ctx->h_seq;
  for (unsigned long spins = 1;; ++spins) {
    if (*p >= seq) break;
    if ((spins & 0x3FFFF) == 0) {
      // every ~millisecond: is the stream still alive?  (a failed kernel would otherwise be polled for ever)
      const hipError_t e = hipStreamQuery(ctx->stream);
      if (e == hipSuccess) {
        if (*p >= seq) break;
        pph_set_error(ctx, "reduction results were not published (sequence %llu, expected %llu)", *p, seq);
        return PPH_ERR_HIP;
      }
      if (e != hipErrorNotReady) {
        pph_set_error(ctx, "stream failed while waiting for reduction results: %s", hipGetErrorString(e));
        return PPH_ERR_HIP;
      }
    }
    __builtin_ia32_pause();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return PPH_OK;
}

static int fetch_to_host(pph_ctx* ctx, int slot, int count) {
  la_flush_final(ctx);
  if (!ctx->fetch_spin) {
    PPH_HIP(ctx, hipMemcpyAsync(ctx->h_scal + slot, ctx->scal.p + slot, sizeof(double) * (size_t)count,
                                hipMemcpyDeviceToHost, ctx->stream));
    PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PPH_OK;
  }
  la_publish(ctx, slot, count);
  return la_wait_published(ctx);
}

// ---- captured iteration bodies ------------------------------------------------------------------------
void la_release_graphs(pph_ctx* ctx) {
  for (auto& g : ctx->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  ctx->graphs.clear();
}

int la_run_graph(pph_ctx* ctx, const GraphKey& key, const std::function<int()>& body, bool publishes) {
  GraphEntry* hit = nullptr;
  for (auto& g : ctx->graphs)
    if (g.key == key) { hit = &g; break; }
  if (!hit) {
    // capture: the launches of `body` are recorded, not executed
    const unsigned long long seq0 = ctx->pub_seq;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      ctx->use_graphs = 0;
      return body();
    }
    const int st = body();
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    ctx->pub_seq = seq0;   // nothing was published yet
    hipGraphExec_t exec = nullptr;
    if (st < 0 || e != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (graph) (void)hipGraphDestroy(graph);
      ctx->use_graphs = 0;   // this runtime / this body cannot be captured: eager from now on
      if (st < 0) return st;
      return body();
    }
    (void)hipGraphDestroy(graph);
    if (ctx->graphs.size() >= 16) {   // drop the least recently used entry
      size_t lru = 0;
      for (size_t i = 1; i < ctx->graphs.size(); ++i)
        if (ctx->graphs[i].used < ctx->graphs[lru].used) lru = i;
      (void)hipGraphExecDestroy(ctx->graphs[lru].exec);
      ctx->graphs.erase(ctx->graphs.begin() + (long)lru);
    }
    GraphEntry ge;
    ge.key = key; ge.exec = exec;
    ctx->graphs.push_back(ge);
    hit = &ctx->graphs.back();
    ctx->n_graph_capture++;
  }
  hit->used = ++ctx->graph_clock;
  // what the body would have counted on the host: one publication per replay
  if (publishes) ++ctx->pub_seq;
  PPH_HIP(ctx, hipGraphLaunch(hit->exec, ctx->stream));
  ctx->n_graph_launch++;
  return PPH_OK;
}

// host copy of already reduced scalars
int la_fetch_raw(pph_ctx* ctx, int slot, int count) {
  PPH_TRY(fetch_to_host(ctx, slot, count));
  if (ctx->comm_status != PPH_OK) { ctx->err = ctx->comm_error; return ctx->comm_status; }
  return PPH_OK;
}

// x += alpha p ; r -= alpha q ; partials of r.r, with alpha = *num / *den read on the device
// (z0 != null: also z0 = dinv0 .* r * w0, the multigrid cycle's pre-smoothing of the NEW residual from a zero guess -
// the first kernel of the next preconditioner application, which then starts at its residual product)
__global__ __launch_bounds__(256) void k_cg_update_dev(double* __restrict__ x, double* __restrict__ r,
                                                       const double* __restrict__ p, const double* __restrict__ q,
                                                       const double* num, const double* __restrict__ den,
                                                       int64_t n, Seg sg, double* __restrict__ part,
                                                       double* __restrict__ z0, const double* __restrict__ dinv0,
                                                       const double* __restrict__ w0p, double* bad,
                                                       const double* __restrict__ den_part, int den_n, double* den_out,
                                                       const double* rot_src, double* rot_dst) {
  __shared__ double lds[4];
  // den_part: the denominator's final reduction was left to this kernel - every block sums the partial sums (k_reduce_final's
  // order), block 0 leaves the result (and the rotated r.z) where the host's publication and the next kernels read them
  double dn;
  if (den_part) {
    dn = block_total_of_partials(den_part, den_n, lds);
    if (rot_src) num = rot_src;             // r.z (current) := r.z (new): read from the source, written once below
    if (blockIdx.x == 0 && threadIdx.x == 0) { *den_out = dn; if (rot_dst) *rot_dst = *rot_src; }
  } else {
    dn = *den;
  }
  // p.Ap = 0 or not a number (a zero block residual, a breakdown): no update instead of NaNs in x and r, and a count in
  // *bad for the host - the launch-only sweeps (cg_solve_fixed) see no scalar of this solve otherwise
  const bool okd = dn > 0.0 || dn < 0.0;
  const double alpha = okd ? *num / dn : 0.0;
  if (bad && !okd && blockIdx.x == 0 && threadIdx.x == 0) *bad += 1.0;
  const double w0 = z0 ? *w0p : 0.0;
  double a = 0.0;
  EW_LOOP(i, n) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    if (z0) z0[i] = dinv0[i] * ri * w0;
    if ((i >= sg.off1 && i < sg.off1 + sg.len1) || (i >= sg.off2 && i < sg.off2 + sg.len2)) a += ri * ri;
  }
  a = block_sum(a, lds);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

bool la_cg_update_dev(pph_ctx* ctx, double* x, double* r, const double* p, const double* q, int slot_num, int slot_den,
                      int64_t n, int slot_out, Seg sg, double* z0, const double* dinv0, const double* w0, int slot_bad,
                      int pub_slot, int pub_count) {
  double* part = partials(ctx);
  int grid = ew_grid(n);
  if (grid > RED_BLOCKS) grid = RED_BLOCKS;
  // a pending final reduction of the denominator is folded into this kernel (its own partial sums then go to another area)
  const double* den_part = nullptr; int den_n = 0; double* den_out = nullptr; const double* rot_src = nullptr; double* rot_dst = nullptr;
  if (ctx->pend.valid && ctx->pend.slot == slot_den && (ctx->pend.copy_src < 0 || ctx->pend.copy_dst == slot_num)) {
    den_part = ctx->pend.part; den_n = ctx->pend.n; den_out = ctx->scal.p + slot_den;
    if (ctx->pend.copy_src >= 0) { rot_src = ctx->scal.p + ctx->pend.copy_src; rot_dst = ctx->scal.p + ctx->pend.copy_dst; }
    ctx->pend.valid = false;
    part = partials(ctx) + 8 * PART_STRIDE;
  } else {
    la_flush_final(ctx);
  }
  hipLaunchKernelGGL(k_cg_update_dev, dim3(grid), dim3(256), 0, ctx->stream, x, r, p, q, ctx->scal.p + slot_num,
                     ctx->scal.p + slot_den, n, sg, part, z0, dinv0, w0, slot_bad >= 0 ? ctx->scal.p + slot_bad : (double*)nullptr,
                     den_part, den_n, den_out, rot_src, rot_dst);
  if (pub_count > 0 && ctx->fetch_spin && (ctx->world == 1 || ctx->comm_suspended)) {
    // the caller publishes right after this update (la_publish semantics): reduce and publish in one launch
    ++ctx->pub_seq;
    hipLaunchKernelGGL(k_reduce_final_publish, dim3(1), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot_out,
                       ctx->scal.p + pub_slot, ctx->h_scal_dev + pub_slot, pub_count, ctx->h_seq_dev, ctx->pub_ctr.p);
    return true;
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot_out);
  return false;
}

// p = z + beta p with beta = *num / *den read on the device
__global__ __launch_bounds__(256) void k_p_update_dev(double* __restrict__ p, const double* __restrict__ z, const double* __restrict__ num,
                               const double* __restrict__ den, int64_t n, const double* __restrict__ num_part, int num_n,
                               double* num_out) {
  __shared__ double lds[4];
  const double dn = *den;
  // num_part: the numerator's final reduction (r.z of the cycle's last kernel) was left to this kernel
  double nm;
  if (num_part) {
    nm = block_total_of_partials(num_part, num_n, lds);
    if (blockIdx.x == 0 && threadIdx.x == 0) *num_out = nm;
  } else {
    nm = *num;
  }
  const double beta = (dn > 0.0 || dn < 0.0) ? nm / dn : 0.0;   // (r.z = 0: restart the direction, see k_cg_update_dev)
  EW_LOOP(i, n) p[i] = z[i] + beta * p[i];
}

void la_p_update_dev(pph_ctx* ctx, double* p, const double* z, int slot_num, int slot_den, int64_t n) {
  const double* num_part = nullptr; int num_n = 0; double* num_out = nullptr;
  if (ctx->pend.valid && ctx->pend.slot == slot_num && ctx->pend.copy_src < 0) {
    num_part = ctx->pend.part; num_n = ctx->pend.n; num_out = ctx->scal.p + slot_num;
    ctx->pend.valid = false;
  } else {
    la_flush_final(ctx);
  }
  hipLaunchKernelGGL(k_p_update_dev, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, p, z, ctx->scal.p + slot_num,
                     ctx->scal.p + slot_den, n, num_part, num_n, num_out);
}

int la_fetch(pph_ctx* ctx, int slot, int count) {
  const bool reduce = ctx->world > 1 && !ctx->comm_suspended;
  // RCCL: sum the partial results on the device, on the stream, before they travel to the host
  if (reduce && ctx->nccl_comm) PPH_TRY(comm_allreduce_device(ctx, ctx->scal.p + slot, count));
  PPH_TRY(fetch_to_host(ctx, slot, count));
  if (reduce && !ctx->nccl_comm) PPH_TRY(comm_allreduce_host(ctx, ctx->h_scal + slot, (int64_t)count));
  // a halo exchange or vector all-reduce that failed since the last fetch (their callers cannot return a status)
  if (ctx->comm_status != PPH_OK) { ctx->err = ctx->comm_error; return ctx->comm_status; }
  return PPH_OK;
}
