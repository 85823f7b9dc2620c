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

// ------------------------------------------------------------------------------------------------
// CSR SpMV, G lanes per row, persistent workgroups, XCD-aware chunk order:
// workgroups with equal (blockIdx % 8) share an XCD (and its 4 MiB L2), so each XCD walks one
// contiguous eighth of the rows and the x entries its rows gather stay in that XCD's L2.
// ------------------------------------------------------------------------------------------------
template <int G, bool DOT, int U>
__global__ __launch_bounds__(256) void k_spmv(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                              const double* __restrict__ val, const double* __restrict__ x,
                                              const double* __restrict__ bvec, double* __restrict__ y, int64_t nrows,
                                              double* __restrict__ part) {
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
      if (U == 1) {
        for (int64_t k = s + sub; k < e; k += G) sum += val[k] * x[col[k]];
      } else {
        // issue U column + U value loads per lane before the first gather (more bytes in flight per wave)
        for (int64_t k0 = s + sub; k0 < e; k0 += (int64_t)U * G) {
          int32_t c[U];
          double v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int64_t k = k0 + (int64_t)u * G;
            const bool ok = k < e;
            c[u] = ok ? col[k] : 0;
            v[u] = ok ? val[k] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) sum += v[u] * x[c[u]];
        }
      }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, G);
    if (sub == 0 && row < nrows) {
      y[row] = bvec ? bvec[row] - sum : sum;
      if (DOT) acc += sum * x[row];
    }
  }
  if (DOT) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
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

// Software-pipelined aligned-wide variant.  A row's critical path in k_spmv_wide is three dependent memory
// latencies (row pointers -> matrix entries -> x gather) that only occupancy hides.  Here every lane group
// keeps two stages in flight: the row pointers of chunk i+2 and the first 4 G matrix entries of chunk i+1 are
// requested before chunk i's gather starts, so the gather of one row overlaps the matrix stream of the next
// and the only exposed latency per row is the gather itself.  Rows longer than one 4 G-entry step finish
// with unpipelined steps (none for the 27- / 15-entry stencil rows of this path).
template <int G, bool DOT, typename VT = double>
__global__ __launch_bounds__(256) void k_spmv_pipe(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const VT* __restrict__ val, const double* __restrict__ x,
                                                   const double* __restrict__ bvec, double* __restrict__ y,
                                                   int64_t nrows, double* __restrict__ part) {
  constexpr int RPB = 256 / G;
  __shared__ double lds[4];
  const int sub = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  const int64_t nchunks = (nrows + RPB - 1) / RPB;
  const int64_t step = gridDim.x;
  auto load_ptr = [&](int64_t chunk, int64_t& ps, int64_t& pe) {
    const int64_t r = chunk * RPB + grp;
    ps = pe = 0;
    if (chunk < nchunks && r < nrows) {
      ps = rowptr[r];
      pe = rowptr[r + 1];
    }
  };
  auto load_mat = [&](int64_t base, int64_t pe, int4& c, double2& a01, double2& a23) {
    c = make_int4(0, 0, 0, 0);
    a01 = a23 = make_double2(0.0, 0.0);
    if (base < pe) {
      c = *reinterpret_cast<const int4*>(col + base);
      if constexpr (sizeof(VT) == 4) {
        const float4 vf = *reinterpret_cast<const float4*>(val + base);
        a01 = make_double2((double)vf.x, (double)vf.y);
        a23 = make_double2((double)vf.z, (double)vf.w);
      } else {
        a01 = *reinterpret_cast<const double2*>(reinterpret_cast<const double*>(val) + base);
        a23 = *reinterpret_cast<const double2*>(reinterpret_cast<const double*>(val) + base + 2);
      }
    }
  };
  auto consume = [&](int64_t base, int64_t s, int64_t e, const int4& c, const double2& a01, const double2& a23) -> double {
    const bool k0 = base >= s && base < e, k1 = base + 1 >= s && base + 1 < e, k2 = base + 2 >= s && base + 2 < e,
               k3 = base + 3 >= s && base + 3 < e;
    const double x0 = x[k0 ? c.x : 0], x1 = x[k1 ? c.y : 0], x2 = x[k2 ? c.z : 0], x3 = x[k3 ? c.w : 0];
    return (k0 ? a01.x : 0.0) * x0 + (k1 ? a01.y : 0.0) * x1 + (k2 ? a23.x : 0.0) * x2 + (k3 ? a23.y : 0.0) * x3;
  };
  double acc = 0.0;
  int64_t ch = blockIdx.x;
  int64_t s, e, ns, ne;
  int4 c;
  double2 v01, v23;
  load_ptr(ch, s, e);
  load_mat((s & ~(int64_t)3) + 4 * sub, e, c, v01, v23);
  load_ptr(ch + step, ns, ne);
  for (; ch < nchunks; ch += step) {
    const int64_t row = ch * RPB + grp;
    int4 nc;
    double2 nv01, nv23;
    load_mat((ns & ~(int64_t)3) + 4 * sub, ne, nc, nv01, nv23);   // chunk i+1: matrix entries
    int64_t nns, nne;
    load_ptr(ch + 2 * step, nns, nne);                            // chunk i+2: row pointers
    double bv = 0.0, xr = 0.0;
    if (row < nrows) {
      if (bvec) bv = bvec[row];
      if (DOT) xr = x[row];
    }
    int64_t base = (s & ~(int64_t)3) + 4 * sub;
    double sum = 0.0;
    if (base < e) sum = consume(base, s, e, c, v01, v23);
    for (base += 4 * G; base < e; base += 4 * G) {                // long rows: remaining steps, unpipelined
      int4 c2;
      double2 a01, a23;
      load_mat(base, e, c2, a01, a23);
      sum += consume(base, s, e, c2, a01, a23);
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
    s = ns; e = ne; c = nc; v01 = nv01; v23 = nv23; ns = nns; ne = nne;
  }
  if (DOT) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

// Row-block streaming variant (one wavefront per workgroup): R consecutive rows own a contiguous range of
// non-zeros; the wave streams that range with fully contiguous 16-byte loads (columns straight to registers
// for the x gather, values through LDS because their 2-per-lane layout differs from the columns' 4-per-lane
// layout), parks the products in LDS and lets two lanes per row sum them.  No masked over-read beyond the
// 4-alignment of the block start, no per-row pointer dependency in front of the loads.
// Needs R * max_row + 3 <= 1024.
template <int R, bool DOT>
__global__ __launch_bounds__(64) void k_spmv_block(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const double* __restrict__ val, const double* __restrict__ x,
                                                   const double* __restrict__ bvec, double* __restrict__ y,
                                                   int64_t nrows, double* __restrict__ part) {
  constexpr int SLOTS = 1024;
  constexpr int LPR = 64 / R;  // lanes per row in the summation phase (R = 32: 2, R = 16: 4)
  __shared__ __attribute__((aligned(16))) double sv[SLOTS + 8];
  __shared__ __attribute__((aligned(16))) double sp[SLOTS + 8];
  const int lane = threadIdx.x;
  const int64_t nchunks = (nrows + R - 1) / R;
  double acc = 0.0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t r0 = ch * R;
    const int64_t rend = (r0 + R < nrows) ? r0 + R : nrows;
    const int64_t s = rowptr[r0], e = rowptr[rend];
    const int64_t s4 = s & ~(int64_t)3;
    const int span = (int)(e - s4);
    const int lead = (int)(s - s4);
    // my row's range (summation phase), requested early
    const int rl = lane / LPR, hl = lane % LPR;
    const int64_t myrow = r0 + rl;
    int b = 0, en = 0;
    if (myrow < rend) {
      b = (int)(rowptr[myrow] - s4);
      en = (int)(rowptr[myrow + 1] - s4);
    }
    int4 c[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = 4 * (lane + 64 * t);
      c[t] = (k < span) ? *reinterpret_cast<const int4*>(col + s4 + k) : make_int4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = 2 * (lane + 64 * u);
      if (k < span) *reinterpret_cast<double2*>(sv + k) = *reinterpret_cast<const double2*>(val + s4 + k);
    }
    double xs[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = 4 * (lane + 64 * t);
      const bool v0 = k >= lead && k < span, v1 = k + 1 >= lead && k + 1 < span, v2 = k + 2 >= lead && k + 2 < span,
                 v3 = k + 3 >= lead && k + 3 < span;
      xs[t][0] = v0 ? x[c[t].x] : 0.0;
      xs[t][1] = v1 ? x[c[t].y] : 0.0;
      xs[t][2] = v2 ? x[c[t].z] : 0.0;
      xs[t][3] = v3 ? x[c[t].w] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = 4 * (lane + 64 * t);
      if (k < span) {
        const double2 a01 = *reinterpret_cast<const double2*>(sv + k);
        const double2 a23 = *reinterpret_cast<const double2*>(sv + k + 2);
        *reinterpret_cast<double2*>(sp + k) = make_double2(a01.x * xs[t][0], a01.y * xs[t][1]);
        *reinterpret_cast<double2*>(sp + k + 2) = make_double2(a23.x * xs[t][2], a23.y * xs[t][3]);
      }
    }
    __syncthreads();
    double sum = 0.0;
    for (int i = b + hl; i < en; i += LPR) sum += sp[i];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, LPR);
    if (hl == 0 && myrow < rend) {
      y[myrow] = bvec ? bvec[myrow] - sum : sum;
      if (DOT) acc += sum * x[myrow];
    }
    __syncthreads();
  }
  if (DOT) {
    acc = wave_sum(acc);
    if (lane == 0) part[blockIdx.x] = acc;
  }
}

// Variant of the aligned-wide kernel in which every load INSTRUCTION of a lane group is contiguous: the group's
// 32 slots are covered by two column loads (8 B per lane) and two value loads (16 B per lane) over slots
// [2 sub, 2 sub + 1] and [16 + 2 sub, 17 + 2 sub] instead of one 16-byte column load and two interleaved value
// loads per lane (which touch every 128-byte line of the values twice).
template <bool DOT>
__global__ __launch_bounds__(256) void k_spmv_wide2(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const double* __restrict__ val, const double* __restrict__ x,
                                                    const double* __restrict__ bvec, double* __restrict__ y,
                                                    int64_t nrows, double* __restrict__ part) {
  constexpr int G = 8, RPB = 256 / G;
  __shared__ double lds[4];
  const int sub = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  const int64_t nchunks = (nrows + RPB - 1) / RPB;
  double acc = 0.0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t row = ch * RPB + grp;
    double sum = 0.0;
    if (row < nrows) {
      const int64_t s = rowptr[row], e = rowptr[row + 1];
      for (int64_t g0 = (s & ~(int64_t)3); g0 < e; g0 += 32) {
        const int64_t ba = g0 + 2 * sub, bb = g0 + 16 + 2 * sub;
        const bool la = ba < e, lb = bb < e;
        int2 ca = make_int2(0, 0), cb = make_int2(0, 0);
        double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
        if (la) { ca = *reinterpret_cast<const int2*>(col + ba); va = *reinterpret_cast<const double2*>(val + ba); }
        if (lb) { cb = *reinterpret_cast<const int2*>(col + bb); vb = *reinterpret_cast<const double2*>(val + bb); }
        const bool k0 = ba >= s && ba < e, k1 = ba + 1 >= s && ba + 1 < e, k2 = bb >= s && bb < e, k3 = bb + 1 >= s && bb + 1 < e;
        const double x0 = x[k0 ? ca.x : 0], x1 = x[k1 ? ca.y : 0], x2 = x[k2 ? cb.x : 0], x3 = x[k3 ? cb.y : 0];
        sum += (k0 ? va.x : 0.0) * x0 + (k1 ? va.y : 0.0) * x1 + (k2 ? vb.x : 0.0) * x2 + (k3 ? vb.y : 0.0) * x3;
      }
    }
    sum = group8_sum(sum);
    if (sub == 0 && row < nrows) {
      y[row] = bvec ? bvec[row] - sum : sum;
      if (DOT) acc += sum * x[row];
    }
  }
  if (DOT) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

// Experiment: rows padded to a multiple of 4 entries with explicit zeros (column = the row itself) and
// row starts aligned to 4: no masks at all in the inner loop, every 16-byte load is aligned inside its row.
template <int G, bool DOT>
__global__ __launch_bounds__(256) void k_spmv_padded(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     const double* __restrict__ val, const double* __restrict__ x,
                                                     const double* __restrict__ bvec, double* __restrict__ y,
                                                     int64_t nrows, double* __restrict__ part) {
  constexpr int RPB = 256 / G;
  __shared__ double lds[4];
  const int sub = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  const int64_t nchunks = (nrows + RPB - 1) / RPB;
  double acc = 0.0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t row = ch * RPB + grp;
    double sum = 0.0;
    if (row < nrows) {
      const int64_t s = rowptr[row], e = rowptr[row + 1];
      for (int64_t base = s + 4 * sub; base < e; base += 4 * G) {
        const int4 c = *reinterpret_cast<const int4*>(col + base);
        const double2 v01 = *reinterpret_cast<const double2*>(val + base);
        const double2 v23 = *reinterpret_cast<const double2*>(val + base + 2);
        sum += v01.x * x[c.x] + v01.y * x[c.y] + v23.x * x[c.z] + v23.y * x[c.w];
      }
    }
    if constexpr (G == 8) sum = group8_sum(sum);
    else {
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, G);
    }
    if (sub == 0 && row < nrows) {
      y[row] = bvec ? bvec[row] - sum : sum;
      if (DOT) acc += sum * x[row];
    }
  }
  if (DOT) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

__global__ void k_pad_fill(const int64_t* __restrict__ rp, const int32_t* __restrict__ col, const double* __restrict__ val,
                           const int64_t* __restrict__ rpp, int32_t* __restrict__ colp, double* __restrict__ valp,
                           int64_t nrows) {
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < nrows; row += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = rp[row], e = rp[row + 1], sp = rpp[row], ep = rpp[row + 1];
    for (int64_t k = 0; k < ep - sp; ++k) {
      const bool real = s + k < e;
      colp[sp + k] = real ? col[s + k] : (int32_t)row;
      valp[sp + k] = real ? val[s + k] : 0.0;
    }
  }
}

// Aligned-wide CSR-vector kernel with UR rows in flight per lane group: the dependent chain
// rowptr -> (columns, values) -> x gather is latency-bound when a wave carries one row per group
// (about 3 KB in flight), so every group walks UR row-chunks at once: all row pointers are requested
// first, then all 16-byte column/value loads, then all gathers.
template <int G, bool DOT, int UR>
__global__ __launch_bounds__(256) void k_spmv_wide_u(const int64_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col,
                                                     const double* __restrict__ val, const double* __restrict__ x,
                                                     const double* __restrict__ bvec, double* __restrict__ y,
                                                     int64_t nrows, double* __restrict__ part) {
  constexpr int RPB = 256 / G;
  __shared__ double lds[4];
  const int sub = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  const int64_t nchunks = (nrows + RPB - 1) / RPB;
  double acc = 0.0;
  for (int64_t ch0 = blockIdx.x; ch0 < nchunks; ch0 += (int64_t)UR * gridDim.x) {
    int64_t row[UR], s[UR], e[UR], base[UR];
    double sum[UR], bv[UR], xr[UR];   // epilogue operands requested with the row pointers (see k_spmv_wide)
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t ch = ch0 + (int64_t)u * gridDim.x;
      row[u] = ch * RPB + grp;
      const bool ok = ch < nchunks && row[u] < nrows;
      if (!ok) row[u] = -1;
      s[u] = ok ? rowptr[row[u]] : 0;
      e[u] = ok ? rowptr[row[u] + 1] : 0;
      bv[u] = (ok && bvec) ? bvec[row[u]] : 0.0;
      xr[u] = (ok && DOT) ? x[row[u]] : 0.0;
      sum[u] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) base[u] = (s[u] & ~(int64_t)3) + 4 * sub;
    bool more = false;
#pragma unroll
    for (int u = 0; u < UR; ++u) more = more || (base[u] < e[u]);
    while (more) {
      int4 c[UR];
      double2 va[UR], vb[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const bool act = base[u] < e[u];
        const int64_t b = act ? base[u] : 0;
        c[u] = *reinterpret_cast<const int4*>(col + b);
        va[u] = *reinterpret_cast<const double2*>(val + b);
        vb[u] = *reinterpret_cast<const double2*>(val + b + 2);
      }
      double xg[UR][4];
      bool kk[UR][4];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int64_t b = base[u];
        kk[u][0] = b >= s[u] && b < e[u];
        kk[u][1] = b + 1 >= s[u] && b + 1 < e[u];
        kk[u][2] = b + 2 >= s[u] && b + 2 < e[u];
        kk[u][3] = b + 3 >= s[u] && b + 3 < e[u];
        xg[u][0] = x[kk[u][0] ? c[u].x : 0];
        xg[u][1] = x[kk[u][1] ? c[u].y : 0];
        xg[u][2] = x[kk[u][2] ? c[u].z : 0];
        xg[u][3] = x[kk[u][3] ? c[u].w : 0];
      }
      more = false;
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        sum[u] += (kk[u][0] ? va[u].x : 0.0) * xg[u][0] + (kk[u][1] ? va[u].y : 0.0) * xg[u][1] +
                  (kk[u][2] ? vb[u].x : 0.0) * xg[u][2] + (kk[u][3] ? vb[u].y : 0.0) * xg[u][3];
        base[u] += 4 * G;
        more = more || (base[u] < e[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      double t = sum[u];
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) t += __shfl_down(t, o, G);
      if (sub == 0 && row[u] >= 0) {
        y[row[u]] = bvec ? bv[u] - t : t;
        if (DOT) acc += t * xr[u];
      }
    }
  }
  if (DOT) {
    acc = block_sum(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
  }
}

// LDS-transposed CSR variant ("stream in by non-zero, consume by row"): a 64-lane workgroup (one wavefront)
// takes RB = 64/T consecutive rows, copies their contiguous column/value ranges into LDS with 16-byte
// coalesced loads, then every lane walks ONE row (T = 1; T lanes share a row for long rows) out of LDS.
// Consecutive lanes hold consecutive rows, so for a stencil-like matrix the x gathers of one wave
// instruction fall on a few consecutive cache lines instead of one line per lane group, and y is written
// coalesced without any cross-lane reduction.  Needs max_row <= 32 T (at most SPMV_LDS_N entries per step).
#define SPMV_LDS_N 2048
template <int T, bool DOT>
__global__ __launch_bounds__(64) void k_spmv_lds(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                 const double* __restrict__ val, const double* __restrict__ x,
                                                 const double* __restrict__ bvec, double* __restrict__ y,
                                                 int64_t nrows, double* __restrict__ part) {
  constexpr int RB = 64 / T;
  __shared__ __attribute__((aligned(16))) int32_t sc[SPMV_LDS_N + 8];
  __shared__ __attribute__((aligned(16))) double sv[SPMV_LDS_N + 8];
  const int lane = threadIdx.x;
  const int64_t nchunks = (nrows + RB - 1) / RB;
  double acc = 0.0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t R0 = ch * RB;
    const int64_t Rend = (R0 + RB < nrows) ? R0 + RB : nrows;
    const int64_t s = rowptr[R0];
    const int64_t s4 = s & ~(int64_t)3;
    const int span = (int)(rowptr[Rend] - s4);
    for (int i = lane * 4; i < span; i += 256) {
      const int4 c = *reinterpret_cast<const int4*>(col + s4 + i);
      const double2 v01 = *reinterpret_cast<const double2*>(val + s4 + i);
      const double2 v23 = *reinterpret_cast<const double2*>(val + s4 + i + 2);
      *reinterpret_cast<int4*>(sc + i) = c;
      *reinterpret_cast<double2*>(sv + i) = v01;
      *reinterpret_cast<double2*>(sv + i + 2) = v23;
    }
    __syncthreads();
    const int rl = lane / T, j = lane % T;
    const int64_t row = R0 + rl;
    double sum = 0.0;
    if (row < Rend) {
      const int b = (int)(rowptr[row] - s4), e = (int)(rowptr[row + 1] - s4);
      const int per = (e - b + T - 1) / T;
      const int b2 = b + j * per;
      const int e2 = (b2 + per < e) ? b2 + per : e;
      for (int i = b2; i < e2; i += 8) {
        double xv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = i + u < e2;
          const int32_t c = ok ? sc[i + u] : 0;
          vv[u] = ok ? sv[i + u] : 0.0;
          xv[u] = x[c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += vv[u] * xv[u];
      }
    }
#pragma unroll
    for (int o = T / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, T);
    if (j == 0 && row < Rend) {
      y[row] = bvec ? bvec[row] - sum : sum;
      if (DOT) acc += sum * x[row];
    }
    __syncthreads();
  }
  if (DOT) {
    acc = wave_sum(acc);
    if (lane == 0) part[blockIdx.x] = acc;
  }
}

// CSR-stream variant: a workgroup takes RB = 256/T consecutive rows, streams their (contiguous) column
// and value ranges with fully coalesced loads, parks the products in LDS and lets T lanes per row sum
// them.  Requires max row length <= 8 T (launcher picks T), i.e. at most SPMV_LB products per chunk.
#define SPMV_LB 2048
template <int T, bool DOT>
__global__ __launch_bounds__(256) void k_spmv_stream(const int64_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col,
                                                     const double* __restrict__ val, const double* __restrict__ x,
                                                     const double* __restrict__ bvec, double* __restrict__ y,
                                                     int64_t nrows, double* __restrict__ part) {
  constexpr int RB = 256 / T;
  constexpr int U = SPMV_LB / 256;
  __shared__ double prod[SPMV_LB];
  __shared__ int64_t rp[RB + 1];
  __shared__ double lds[4];
  const int tid = threadIdx.x;
  const int64_t nchunks = (nrows + RB - 1) / RB;
  const int xcd = blockIdx.x & 7;
  const int bx = blockIdx.x >> 3;
  const int bpx = gridDim.x >> 3;
  const int64_t cpx = (nchunks + 7) >> 3;
  const int64_t c_begin = (int64_t)xcd * cpx;
  const int64_t c_end = (c_begin + cpx < nchunks) ? c_begin + cpx : nchunks;
  double acc = 0.0;
  for (int64_t ch = c_begin + bx; ch < c_end; ch += bpx) {
    const int64_t R0 = ch * RB;
    if (tid <= RB) {
      const int64_t r = R0 + tid;
      rp[tid] = rowptr[r < nrows ? r : nrows];
    }
    __syncthreads();
    const int64_t s = rp[0];
    const int cnt = (int)(rp[RB] - s);
    int32_t c[U];
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = tid + u * 256;
      const bool ok = k < cnt;
      c[u] = ok ? col[s + k] : 0;
      v[u] = ok ? val[s + k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = tid + u * 256;
      if (k < cnt) prod[k] = v[u] * x[c[u]];
    }
    __syncthreads();
    const int rl = tid / T, j = tid % T;
    const int b = (int)(rp[rl] - s), e = (int)(rp[rl + 1] - s);
    double sum = 0.0;
    for (int i = b + j; i < e; i += T) sum += prod[i];
#pragma unroll
    for (int o = T / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, T);
    const int64_t row = R0 + rl;
    if (j == 0 && row < nrows) {
      y[row] = bvec ? bvec[row] - sum : sum;
      if (DOT) acc += sum * x[row];
    }
    __syncthreads();
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

// returns the grid used (number of partial sums written when DOT)
// jdinv != null (stencil-ELL operators only): y = x + jw * jdinv .* (bvec - A x), one damped-Jacobi sweep out of place
template <bool DOT>
static int spmv_dispatch(pph_ctx* ctx, const Csr& A, const double* x, const double* bvec, double* y, double* part,
                         const double* jdinv = nullptr, const double* jw = nullptr, bool jdot = false, int64_t dlo = 0,
                         int64_t dhi = 0, bool x_ghosts_valid = false) {
  const int variant = DOT ? 1 : 0;
  if (A.geom && !x_ghosts_valid) {  // ghost planes of x <- owners (slabs only); field-major mixed vectors carry two fields
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
    grid = sell_spmv(ctx, A.ell, A.nrows, jdinv ? (jdot ? 4 : 3) : (DOT ? 2 : (bvec ? 1 : 0)), x, bvec, jdinv, jw, y,
                     part ? part : partials(ctx), dlo, dhi);
    if (ev) (void)hipEventRecord(ev->e1, ctx->stream);
    const double bytes = 8.0 * sell_slots(A.ell.kind) * (double)A.nrows + 16.0 * (double)A.nrows;
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
  const bool can_stream = A.max_row > 0 && A.max_row <= 8 * 64;
  if (ctx->spmv_kernel == 2 && can_stream) {
    const int T = stream_T(A.max_row);
    grid = spmv_grid(ctx, A.nrows, 256 / T);
#define PPH_STREAM_CASE(TT)                                                                                       \
  case TT:                                                                                                        \
    hipLaunchKernelGGL((k_spmv_stream<TT, DOT>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, \
                       bvec, y, A.nrows, part);                                                                   \
    break;
    switch (T) {
      PPH_STREAM_CASE(4)
      PPH_STREAM_CASE(8)
      PPH_STREAM_CASE(16)
      PPH_STREAM_CASE(32)
      default:
        hipLaunchKernelGGL((k_spmv_stream<64, DOT>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x,
                           bvec, y, A.nrows, part);
    }
#undef PPH_STREAM_CASE
  } else if (ctx->spmv_kernel == 4 && A.max_row > 0 && A.max_row <= 32 * 4) {
    const int T = A.max_row <= 32 ? 1 : (A.max_row <= 64 ? 2 : 4);
    const int64_t nchunks = ceil_div64(A.nrows, 64 / T);
    grid = (int)(nchunks < SPMV_MAX_BLOCKS ? nchunks : SPMV_MAX_BLOCKS);
    if (T == 1)
      hipLaunchKernelGGL((k_spmv_lds<1, DOT>), dim3(grid), dim3(64), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
    else if (T == 2)
      hipLaunchKernelGGL((k_spmv_lds<2, DOT>), dim3(grid), dim3(64), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
    else
      hipLaunchKernelGGL((k_spmv_lds<4, DOT>), dim3(grid), dim3(64), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
  } else if (ctx->spmv_kernel >= 5 && ctx->spmv_kernel <= 7) {
    grid = spmv_grid(ctx, A.nrows, 256 / 8);
    if (ctx->spmv_kernel == 5)
      hipLaunchKernelGGL((k_spmv_wide_u<8, DOT, 2>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
    else if (ctx->spmv_kernel == 6)
      hipLaunchKernelGGL((k_spmv_wide_u<8, DOT, 4>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
    else
      hipLaunchKernelGGL((k_spmv_wide_u<4, DOT, 4>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
  } else if (ctx->spmv_kernel == 14 && A.max_row > 0 && A.max_row * 16 + 3 <= 1024) {
    const bool r32 = A.max_row * 32 + 3 <= 1024;
    const int64_t nchunks = ceil_div64(A.nrows, r32 ? 32 : 16);
    const int64_t cap = (int64_t)(ctx->spmv_blocks >= 8 ? ctx->spmv_blocks : SPMV_DEF_BLOCKS) * 4;  // one wave per workgroup
    grid = (int)(nchunks < cap ? nchunks : cap);
    if (r32)
      hipLaunchKernelGGL((k_spmv_block<32, DOT>), dim3(grid), dim3(64), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
    else
      hipLaunchKernelGGL((k_spmv_block<16, DOT>), dim3(grid), dim3(64), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
  } else if (ctx->spmv_kernel == 17) {
    if (A.max_row > 0 && A.max_row + 3 <= 16) {
      grid = spmv_grid(ctx, A.nrows, 256 / 4);
      hipLaunchKernelGGL((k_spmv_pipe<4, DOT>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
    } else {
      grid = spmv_grid(ctx, A.nrows, 256 / 8);
      hipLaunchKernelGGL((k_spmv_pipe<8, DOT>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
    }
  } else if (ctx->spmv_kernel == 15) {
    grid = spmv_grid(ctx, A.nrows, 256 / 8);
    hipLaunchKernelGGL((k_spmv_wide2<DOT>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
  } else if (ctx->spmv_kernel == 9) {
    grid = spmv_grid(ctx, A.nrows, 256 / 8);
    hipLaunchKernelGGL((k_spmv_wide<8, DOT, 3>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
  } else if (ctx->spmv_kernel == 10) {
    grid = spmv_grid(ctx, A.nrows, 256 / 8);
    hipLaunchKernelGGL((k_spmv_wide<8, DOT, 1>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
  } else if (ctx->spmv_kernel == 12) {
    grid = spmv_grid(ctx, A.nrows, 256 / 8);
    hipLaunchKernelGGL((k_spmv_wide<8, DOT, 4>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
  } else if (ctx->spmv_kernel == 13) {
    grid = spmv_grid(ctx, A.nrows, 256 / 8);
    hipLaunchKernelGGL((k_spmv_wide<8, DOT, 5>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, y, A.nrows, part);
  } else if (ctx->spmv_kernel == 3 || ctx->spmv_kernel == 8 || ctx->spmv_kernel == 11) {
    // lanes per row: one 4-wide step covers 4 G entries; rows of up to 29 entries fit G = 8
    int G = ctx->spmv_lanes_override > 0 ? ctx->spmv_lanes_override : (A.max_row > 0 && A.max_row + 3 <= 16 ? 4 : 8);
    if (G != 4 && G != 8 && G != 16 && G != 32 && G != 64) G = 8;
    grid = spmv_grid(ctx, A.nrows, 256 / G);
#define PPH_WIDE_CASE(GG)                                                                                       \
  case GG:                                                                                                      \
    if (ctx->spmv_kernel == 8)                                                                                  \
      hipLaunchKernelGGL((k_spmv_wide<GG, DOT, 0>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, \
                         bvec, y, A.nrows, part);                                                               \
    else                                                                                                        \
      hipLaunchKernelGGL((k_spmv_wide<GG, DOT, 2>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, \
                         bvec, y, A.nrows, part);                                                               \
    break;
    switch (G) {
      PPH_WIDE_CASE(4)
      PPH_WIDE_CASE(8)
      PPH_WIDE_CASE(16)
      PPH_WIDE_CASE(32)
      PPH_WIDE_CASE(64)
    }
#undef PPH_WIDE_CASE
  } else {
    int G = A.lanes;
    if (G != 4 && G != 8 && G != 16 && G != 32 && G != 64) G = 8;
    grid = spmv_grid(ctx, A.nrows, 256 / G);
#define PPH_SPMV_CASE(GG)                                                                                          \
  case GG:                                                                                                         \
    if (ctx->spmv_kernel == 0)                                                                                     \
      hipLaunchKernelGGL((k_spmv<GG, DOT, 1>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, \
                         y, A.nrows, part);                                                                        \
    else                                                                                                           \
      hipLaunchKernelGGL((k_spmv<GG, DOT, 4>), dim3(grid), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, x, bvec, \
                         y, A.nrows, part);                                                                        \
    break;
    switch (G) {
      PPH_SPMV_CASE(4)
      PPH_SPMV_CASE(8)
      PPH_SPMV_CASE(16)
      PPH_SPMV_CASE(32)
      PPH_SPMV_CASE(64)
    }
#undef PPH_SPMV_CASE
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
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_pool[i].e0, ctx->ev_pool[i].e1) == hipSuccess) {
      ctx->t_spmv[ctx->ev_pool[i].variant] += ms;
      if (ctx->ev_pool[i].fine) ctx->t_spmv_fine += ms;
    }
  }
  ctx->ev_used = 0;
}

void la_reset_spmv_stats(pph_ctx* ctx) {
  la_harvest_spmv_times(ctx);
  for (int v = 0; v < 2; ++v) { ctx->t_spmv[v] = 0; ctx->spmv_bytes[v] = 0; ctx->n_spmv[v] = 0; }
  ctx->t_spmv_fine = 0; ctx->spmv_bytes_fine = 0; ctx->n_spmv_fine = 0;
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
  if (dot_slot >= 0)
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, partials(ctx), grid, ctx->scal.p + dot_slot);
}

void la_spmv_dot(pph_ctx* ctx, const Csr& A, const double* x, double* y, int slot, int copy_src, int copy_dst) {
  double* part = partials(ctx);
  const int grid = spmv_dispatch<true>(ctx, A, x, nullptr, y, part);
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
  (void)hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream);
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
                                                       const double* __restrict__ num, const double* __restrict__ den,
                                                       int64_t n, Seg sg, double* __restrict__ part,
                                                       double* __restrict__ z0, const double* __restrict__ dinv0,
                                                       const double* __restrict__ w0p) {
  __shared__ double lds[4];
  const double alpha = *num / *den;
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

void la_cg_update_dev(pph_ctx* ctx, double* x, double* r, const double* p, const double* q, int slot_num, int slot_den,
                      int64_t n, int slot_out, Seg sg, double* z0, const double* dinv0, const double* w0) {
  double* part = partials(ctx);
  int grid = ew_grid(n);
  if (grid > RED_BLOCKS) grid = RED_BLOCKS;
  hipLaunchKernelGGL(k_cg_update_dev, dim3(grid), dim3(256), 0, ctx->stream, x, r, p, q, ctx->scal.p + slot_num,
                     ctx->scal.p + slot_den, n, sg, part, z0, dinv0, w0);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, ctx->stream, part, grid, ctx->scal.p + slot_out);
}

// p = z + beta p with beta = *num / *den read on the device
__global__ void k_p_update_dev(double* __restrict__ p, const double* __restrict__ z, const double* __restrict__ num,
                               const double* __restrict__ den, int64_t n) {
  const double beta = *num / *den;
  EW_LOOP(i, n) p[i] = z[i] + beta * p[i];
}

void la_p_update_dev(pph_ctx* ctx, double* p, const double* z, int slot_num, int slot_den, int64_t n) {
  hipLaunchKernelGGL(k_p_update_dev, dim3(ew_grid(n)), dim3(256), 0, ctx->stream, p, z, ctx->scal.p + slot_num,
                     ctx->scal.p + slot_den, n);
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

// builds the padded copy of A (host-side row-pointer scan: experiment only) and times `reps` launches
int la_padded_experiment(pph_ctx* ctx, const Csr& A, int reps, double* avg_ms) {
  std::vector<int64_t> rp((size_t)A.nrows + 1), rpp((size_t)A.nrows + 1);
  PPH_HIP(ctx, hipMemcpy(rp.data(), A.rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost));
  rpp[0] = 0;
  for (int64_t r = 0; r < A.nrows; ++r) rpp[(size_t)r + 1] = rpp[(size_t)r] + ((rp[(size_t)r + 1] - rp[(size_t)r] + 3) / 4) * 4;
  const int64_t nnzp = rpp[(size_t)A.nrows];
  DevBuf<int64_t> drpp;
  DevBuf<int32_t> colp;
  DevBuf<double> valp, x, y;
  PPH_TRY(drpp.alloc(ctx, rpp.size()));
  PPH_TRY(colp.alloc(ctx, (size_t)nnzp));
  PPH_TRY(valp.alloc(ctx, (size_t)nnzp));
  PPH_TRY(x.alloc(ctx, (size_t)A.nrows));
  PPH_TRY(y.alloc(ctx, (size_t)A.nrows));
  PPH_HIP(ctx, hipMemcpy(drpp.p, rpp.data(), sizeof(int64_t) * rpp.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_pad_fill, dim3(2048), dim3(256), 0, ctx->stream, A.rowptr, A.col, A.val, drpp.p, colp.p, valp.p, A.nrows);
  hipLaunchKernelGGL(k_set, dim3(2048), dim3(256), 0, ctx->stream, x.p, 1.0, A.nrows);
  const int grid = spmv_grid(ctx, A.nrows, 32);
  for (int i = 0; i < 5 + reps; ++i) {
    if (i == 5) PPH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL((k_spmv_padded<8, false>), dim3(grid), dim3(256), 0, ctx->stream, drpp.p, colp.p, valp.p, x.p,
                       (const double*)nullptr, y.p, A.nrows, (double*)nullptr);
  }
  PPH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  PPH_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  PPH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *avg_ms = ms / reps;
  drpp.release(); colp.release(); valp.release(); x.release(); y.release();
  return PPH_OK;
}

