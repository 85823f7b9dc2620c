// Geometric multigrid V-cycle for the scalar blocks A11 = (k1/mu)K + (beta/mu)M and
// A22 = (k2/mu)K + (beta/mu)M: the GPU stand-in for the LU block solves of the reference's
// field-split / Picard configurations (reference src/perphil/solvers/parameters.py:30-37, :79-85).
//
// Hierarchy: nx/2^l structured meshes, operators re-discretised on every level with the same
// K-ASM kernels (equal to the Galerkin product for the nested CG-1 spaces), Dirichlet masks injected.
// Smoother: Chebyshev-Jacobi on [lam/4, lam], lam = max row sum of |D^-1 A|.  Transfers: the
// interpolation of the CG-1 basis (multilinear for Q1, edge-midpoint averaging for the Kuhn /
// left-diagonal P1 triangulations) and its transpose.  The cycle is symmetric, so it is a valid CG
// preconditioner.  oracle/dpp_mg_oracle.py restates it for the parity tests.
#include "pph_internal.h"
#include <cmath>
#include <cstring>

#define MG_CHEB_LOWER 0.25

struct TStencil {
  int count;
  int8_t d[27][3];
  double w[27];
};

static TStencil make_transfer_stencil(int kind) {
  TStencil s;
  s.count = 0;
  auto push = [&](int dx, int dy, int dz) {
    const int nzc = (dx != 0) + (dy != 0) + (dz != 0);
    double w;
    if (kind == PPH_CELL_QUAD || kind == PPH_CELL_HEX) w = std::ldexp(1.0, -nzc);
    else w = (nzc == 0) ? 1.0 : 0.5;
    s.d[s.count][0] = (int8_t)dx; s.d[s.count][1] = (int8_t)dy; s.d[s.count][2] = (int8_t)dz;
    s.w[s.count] = w;
    s.count++;
  };
  if (kind == PPH_CELL_QUAD) {
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) push(dx, dy, 0);
  } else if (kind == PPH_CELL_TRI) {
    push(0, -1, 0); push(1, -1, 0); push(-1, 0, 0); push(0, 0, 0); push(1, 0, 0); push(-1, 1, 0); push(0, 1, 0);
  } else if (kind == PPH_CELL_HEX) {
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) push(dx, dy, dz);
  } else {
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      const bool pos = dx >= 0 && dy >= 0 && dz >= 0, neg = dx <= 0 && dy <= 0 && dz <= 0;
      if (pos || neg) push(dx, dy, dz);
    }
  }
  return s;
}

#define NODE_LOOP(id, n) \
  for (int64_t id = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; id < (n); id += (int64_t)gridDim.x * blockDim.x)

// Level geometry for the transfers: local node dims and the GLOBAL index of local node plane 0, so that
// slabs (local boxes with ghost planes) and replicated coarse levels use the same kernels:
// global fine plane = gzf + kf, global coarse plane = gzc + kc, and fine = 2 * coarse + d.
struct TGeom {
  int pxc, pyc, pzc, gzc;
  int pxf, pyf, pzf, gzf;
  int own_lo_f, own_hi_f;  // owned global fine planes [lo, hi): a coarse plane K is assigned to this rank iff 2K is owned
};

// coarse mask <- fine mask at the coinciding node (0 where the fine node is not in the local box or the
// coarse plane is not assigned to this rank; ghost planes are completed by a halo exchange afterwards)
__global__ void k_inject_mask(double* __restrict__ mc, const uint8_t* __restrict__ mf, TGeom g) {
  const int64_t nc = (int64_t)g.pxc * g.pyc * g.pzc;
  NODE_LOOP(id, nc) {
    const int I = (int)(id % g.pxc);
    const int64_t t = id / g.pxc;
    const int J = (int)(t % g.pyc), K = (int)(t / g.pyc) + g.gzc;
    const int kf = 2 * K - g.gzf;
    double v = 0.0;
    if (2 * K >= g.own_lo_f && 2 * K < g.own_hi_f && kf >= 0 && kf < g.pzf)
      v = (double)(mf[2 * I + (int64_t)g.pxf * (2 * J + (int64_t)g.pyf * kf)] & 1);
    mc[id] = v;
  }
}

__global__ void k_to_float(float* __restrict__ o, const double* __restrict__ v, int64_t n) {
  NODE_LOOP(id, n) o[id] = (float)v[id];
}

__global__ void k_mask_from_double(uint8_t* __restrict__ m, const double* __restrict__ v, int64_t n, int64_t plane,
                                   int glo, int ghi) {
  NODE_LOOP(id, n) {
    uint8_t b = (v[id] != 0.0) ? 1 : 0;
    if ((glo && id < plane) || (ghi && id >= n - plane)) b |= 2;
    m[id] = b;
  }
}

// b_c[C] = sum_d w_d r_f[2C + d]   (0 on constrained coarse dofs and on coarse planes of other ranks)
// (xc != null: also the coarse level's pre-smoothing from a zero guess, x_c = dinv_c .* b_c * wc, in the same pass)
__global__ void k_restrict(double* __restrict__ bc, const double* __restrict__ rf, TStencil st,
                           const uint8_t* __restrict__ mc, const uint8_t* __restrict__ mf, TGeom g,
                           double* __restrict__ xc, const double* __restrict__ dinvc, const double* __restrict__ wcp) {
  const int64_t nc = (int64_t)g.pxc * g.pyc * g.pzc;
  NODE_LOOP(id, nc) {
    const int I = (int)(id % g.pxc);
    const int64_t t = id / g.pxc;
    const int J = (int)(t % g.pyc), K = (int)(t / g.pyc) + g.gzc;
    double s = 0.0;
    if (mc[id] == 0 && 2 * K >= g.own_lo_f && 2 * K < g.own_hi_f) {
      for (int q = 0; q < st.count; ++q) {
        const int i = 2 * I + st.d[q][0], j = 2 * J + st.d[q][1], k = 2 * K + st.d[q][2] - g.gzf;
        if (i >= 0 && i < g.pxf && j >= 0 && j < g.pyf && k >= 0 && k < g.pzf) {
          const int64_t f = i + (int64_t)g.pxf * (j + (int64_t)g.pyf * k);
          if ((mf[f] & 1) == 0) s += st.w[q] * rf[f];  // ghost fine entries hold the owner's residual
        }
      }
    }
    bc[id] = s;
    if (xc) xc[id] = dinvc[id] * s * (*wcp);
  }
}

// Q1 (quad / hex) restriction with the 3^DIM stencil unrolled and every load unconditional (out-of-range and
// constrained neighbours get weight 0 and a clamped address): all loads of a coarse node are in flight together,
// where the generic loop above waits for each conditional load before issuing the next.
// fast[C] = 1 where the restriction of coarse node C needs neither masks nor range checks: C unconstrained and
// assigned to this rank, its 3^DIM fine neighbours all inside the local box and unconstrained (every node away from
// the boundary and from Dirichlet sets).  Depends on the masks only: rebuilt with them.
template <int DIM>
__global__ __launch_bounds__(256) void k_restrict_flags(uint8_t* __restrict__ fast, const uint8_t* __restrict__ mc,
                                                        const uint8_t* __restrict__ mf, TGeom g) {
  const int64_t nc = (int64_t)g.pxc * g.pyc * g.pzc;
  NODE_LOOP(id, nc) {
    const int I = (int)(id % g.pxc);
    const int64_t t = id / g.pxc;
    const int J = (int)(t % g.pyc), K = (int)(t / g.pyc) + g.gzc;
    bool ok = mc[id] == 0 && 2 * K >= g.own_lo_f && 2 * K < g.own_hi_f;
    constexpr int NZ = (DIM == 3) ? 3 : 1;
    for (int dz = 0; dz < NZ && ok; ++dz)
      for (int dy = 0; dy < 3 && ok; ++dy)
        for (int dx = 0; dx < 3 && ok; ++dx) {
          const int i = 2 * I + dx - 1, j = 2 * J + dy - 1, k = (DIM == 3) ? 2 * K + dz - 1 - g.gzf : 0;
          ok = i >= 0 && i < g.pxf && j >= 0 && j < g.pyf && k >= 0 && k < g.pzf;
          if (ok) ok = (mf[i + (int64_t)g.pxf * (j + (int64_t)g.pyf * k)] & 1) == 0;
        }
    fast[id] = ok ? 1 : 0;
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void k_restrict_q1(double* __restrict__ bc, const double* __restrict__ rf,
                                                     const uint8_t* __restrict__ mc, const uint8_t* __restrict__ mf,
                                                     TGeom g, double* __restrict__ xc, const double* __restrict__ dinvc,
                                                     const double* __restrict__ wcp, const uint8_t* __restrict__ fast) {
  const int64_t nc = (int64_t)g.pxc * g.pyc * g.pzc;
  NODE_LOOP(id, nc) {
    int I, J, K;
    if (nc < (int64_t)1 << 31) {   // (32-bit index arithmetic where the level allows it)
      const uint32_t i32 = (uint32_t)id, t32 = i32 / (uint32_t)g.pxc, k32 = t32 / (uint32_t)g.pyc;
      I = (int)(i32 - t32 * (uint32_t)g.pxc); J = (int)(t32 - k32 * (uint32_t)g.pyc); K = (int)k32 + g.gzc;
    } else {
      I = (int)(id % g.pxc);
      const int64_t t = id / g.pxc;
      J = (int)(t % g.pyc); K = (int)(t / g.pyc) + g.gzc;
    }
    double s = 0.0;
    if (fast && fast[id]) {
      // interior: 3^DIM unconditional loads, no masks (same weights, same order of accumulation as below)
      constexpr int NZ = (DIM == 3) ? 3 : 1;
      const int64_t f0 = (2 * I - 1) + (int64_t)g.pxf * ((2 * J - 1) + (int64_t)g.pyf * ((DIM == 3) ? 2 * K - 1 - g.gzf : 0));
      double v[NZ][3][3];
#pragma unroll
      for (int dz = 0; dz < NZ; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) v[dz][dy][dx] = rf[f0 + dx + (int64_t)g.pxf * (dy + (int64_t)g.pyf * dz)];
#pragma unroll
      for (int dz = 0; dz < NZ; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int nzc = (dx != 1) + (dy != 1) + ((DIM == 3) ? (dz != 1) : 0);
            const double w = (nzc == 0) ? 1.0 : (nzc == 1) ? 0.5 : (nzc == 2) ? 0.25 : 0.125;
            s += w * v[dz][dy][dx];
          }
    } else if (mc[id] == 0 && 2 * K >= g.own_lo_f && 2 * K < g.own_hi_f) {
      constexpr int NZ = (DIM == 3) ? 3 : 1;
      double v[NZ][3][3];
      uint8_t m[NZ][3][3];
      bool in[NZ][3][3];
#pragma unroll
      for (int dz = 0; dz < NZ; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int i = 2 * I + dx - 1, j = 2 * J + dy - 1, k = (DIM == 3) ? 2 * K + dz - 1 - g.gzf : 0;
            const bool ok = i >= 0 && i < g.pxf && j >= 0 && j < g.pyf && k >= 0 && k < g.pzf;
            const int64_t f = ok ? i + (int64_t)g.pxf * (j + (int64_t)g.pyf * k) : 0;
            in[dz][dy][dx] = ok;
            v[dz][dy][dx] = rf[f];
            m[dz][dy][dx] = mf[f];
          }
#pragma unroll
      for (int dz = 0; dz < NZ; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int nzc = (dx != 1) + (dy != 1) + ((DIM == 3) ? (dz != 1) : 0);
            const double w = (nzc == 0) ? 1.0 : (nzc == 1) ? 0.5 : (nzc == 2) ? 0.25 : 0.125;
            // same order of accumulation as the generic kernel's stencil (dz, dy, dx ascending)
            if (in[dz][dy][dx] && (m[dz][dy][dx] & 1) == 0) s += w * v[dz][dy][dx];
          }
    }
    bc[id] = s;
    if (xc) xc[id] = dinvc[id] * s * (*wcp);
  }
}

// x_f[f] += (P x_c)[f]   (constrained fine dofs untouched).  Closed form of the transpose of k_restrict:
// a fine node f = 2c + o (o = parity vector of the GLOBAL index) interpolates
//   Q1 (TK 0)         : the 2^|o| coarse nodes c + s, s <= o component-wise, weight 2^-|o|
//   Kuhn P1 (TK 1)    : 1/2 (x_c[c] + x_c[c + o])                         (o != 0)
//   left-diag P1 (TK 2): o = (1,1): 1/2 (x_c[c + x] + x_c[c + y]), else as Kuhn
// out of place: xout = xf + P xc (constrained fine dofs: xout = xf); the post-smoothing sweep then reads xout and
// writes xf (la_spmv_jacobi)
template <int TK>
__global__ __launch_bounds__(256) void k_prolong_to(double* __restrict__ xout, const double* __restrict__ xf,
                                                    const double* __restrict__ xc, const uint8_t* __restrict__ mf,
                                                    TGeom g) {
  const int64_t nf = (int64_t)g.pxf * g.pyf * g.pzf;
  NODE_LOOP(id, nf) {
    const double x0 = xf[id];
    // ghost rows (mask bit 2) are interpolated like owned ones: with the ghost planes of xf and xc refreshed, the
    // ghost planes of xout equal the owners' values and the post-smoothing product needs no exchange of its own
    if ((mf[id] & 1) != 0) { xout[id] = x0; continue; }
    const int i = (int)(id % g.pxf);
    const int64_t t = id / g.pxf;
    const int j = (int)(t % g.pyf), kg = (int)(t / g.pyf) + g.gzf;
    const int ox = i & 1, oy = j & 1, oz = kg & 1;
    const int64_t sx = 1, sy = g.pxc, sz = (int64_t)g.pxc * g.pyc;
    const int64_t c = (i >> 1) + sy * (j >> 1) + sz * ((kg >> 1) - g.gzc);
    double s;
    if (TK == 0) {
      const int64_t ex = ox ? sx : 0, ey = oy ? sy : 0, ez = oz ? sz : 0;
      const double c000 = xc[c], c100 = xc[c + ex], c010 = xc[c + ey], c110 = xc[c + ey + ex];
      const double c001 = xc[c + ez], c101 = xc[c + ez + ex], c011 = xc[c + ez + ey], c111 = xc[c + ez + ey + ex];
      const double v00 = 0.5 * (c000 + c100), v10 = 0.5 * (c010 + c110), v01 = 0.5 * (c001 + c101),
                   v11 = 0.5 * (c011 + c111);
      if (oy && oz) s = 0.25 * (v00 + v10 + v01 + v11);
      else if (oy | oz) s = 0.5 * (v00 + (oy ? v10 : v01));
      else s = v00;
    } else {
      const int64_t o = ox * sx + oy * sy + oz * sz;
      if (TK == 2 && ox && oy) s = 0.5 * (xc[c + sx] + xc[c + sy]);
      else s = (o == 0) ? xc[c] : 0.5 * (xc[c] + xc[c + o]);
    }
    xout[id] = x0 + s;
  }
}

// Q1 interpolation, one thread per PAIR of fine nodes (2m, 2m + 1) of a grid line: the eight coarse values the odd
// node interpolates contain the four of the even node - 8 loads for two nodes instead of 16; per node the
// arithmetic of k_prolong_to<0> (an even node's 0.5 (c + c) is c exactly)
__global__ __launch_bounds__(256) void k_prolong_to_q1(double* __restrict__ xout, const double* __restrict__ xf,
                                                       const double* __restrict__ xc, const uint8_t* __restrict__ mf,
                                                       TGeom g) {
  const int hx = (g.pxf + 1) >> 1;
  const int64_t npairs = (int64_t)hx * g.pyf * g.pzf;
  NODE_LOOP(pid, npairs) {
    // (32-bit index arithmetic where the level allows it: a 64-bit division is ~100 instructions on this hardware)
    int m, j, kl;
    if (npairs < (int64_t)1 << 31) {
      const uint32_t p32 = (uint32_t)pid, t32 = p32 / (uint32_t)hx;
      m = (int)(p32 - t32 * (uint32_t)hx); kl = (int)(t32 / (uint32_t)g.pyf); j = (int)(t32 - (uint32_t)kl * (uint32_t)g.pyf);
    } else {
      m = (int)(pid % hx);
      const int64_t t = pid / hx;
      j = (int)(t % g.pyf); kl = (int)(t / g.pyf);
    }
    const int kg = kl + g.gzf;
    const int64_t id0 = 2 * m + (int64_t)g.pxf * (j + (int64_t)g.pyf * kl);
    const bool has1 = 2 * m + 1 < g.pxf;
    const int oy = j & 1, oz = kg & 1;
    const int64_t sy = g.pxc, sz = (int64_t)g.pxc * g.pyc;
    const int64_t c = m + sy * (j >> 1) + sz * ((kg >> 1) - g.gzc);
    const int64_t ex = has1 ? 1 : 0, ey = oy ? sy : 0, ez = oz ? sz : 0;
    double x0, x1 = 0.0;
    if (has1) sell_ld2(xf + id0, x0, x1); else x0 = xf[id0];   // (16-byte accesses per pair: pph_internal.h)
    const uint8_t m0 = mf[id0], m1 = has1 ? mf[id0 + 1] : 1;
    // (round 4, measured with option transfer_bench: taking the right column from the next lane by a DPP wave shift - four
    // loads instead of eight - leaves the kernel at 0.096 ms; a probe that also drops the odd node's arithmetic reaches 0.075)
    const double a00 = xc[c], b00 = xc[c + ex], a10 = xc[c + ey], b10 = xc[c + ey + ex];
    const double a01 = xc[c + ez], b01 = xc[c + ez + ex], a11 = xc[c + ez + ey], b11 = xc[c + ez + ey + ex];
    double s0, s1;
    {
      const double v00 = 0.5 * (a00 + a00), v10 = 0.5 * (a10 + a10), v01 = 0.5 * (a01 + a01), v11 = 0.5 * (a11 + a11);
      if (oy && oz) s0 = 0.25 * (v00 + v10 + v01 + v11);
      else if (oy | oz) s0 = 0.5 * (v00 + (oy ? v10 : v01));
      else s0 = v00;
    }
    {
      const double v00 = 0.5 * (a00 + b00), v10 = 0.5 * (a10 + b10), v01 = 0.5 * (a01 + b01), v11 = 0.5 * (a11 + b11);
      if (oy && oz) s1 = 0.25 * (v00 + v10 + v01 + v11);
      else if (oy | oz) s1 = 0.5 * (v00 + (oy ? v10 : v01));
      else s1 = v00;
    }
    const double r0 = ((m0 & 1) != 0) ? x0 : x0 + s0;
    if (has1) sell_st2(xout + id0, r0, ((m1 & 1) != 0) ? x1 : x1 + s1);
    else xout[id0] = r0;
  }
}

template <int TK>
__global__ __launch_bounds__(256) void k_prolong_add(double* __restrict__ xf, const double* __restrict__ xc,
                                                     const uint8_t* __restrict__ mf, TGeom g) {
  const int64_t nf = (int64_t)g.pxf * g.pyf * g.pzf;
  NODE_LOOP(id, nf) {
    if (mf[id] != 0) continue;
    const int i = (int)(id % g.pxf);
    const int64_t t = id / g.pxf;
    const int j = (int)(t % g.pyf), kg = (int)(t / g.pyf) + g.gzf;
    const int ox = i & 1, oy = j & 1, oz = kg & 1;
    const int64_t sx = 1, sy = g.pxc, sz = (int64_t)g.pxc * g.pyc;
    const int64_t c = (i >> 1) + sy * (j >> 1) + sz * ((kg >> 1) - g.gzc);
    double s;
    if (TK == 0) {
      // the 8 candidate coarse values are requested unconditionally (even parities repeat an address), then combined
      // exactly as before: no load waits behind a branch
      const int64_t ex = ox ? sx : 0, ey = oy ? sy : 0, ez = oz ? sz : 0;
      const double c000 = xc[c], c100 = xc[c + ex], c010 = xc[c + ey], c110 = xc[c + ey + ex];
      const double c001 = xc[c + ez], c101 = xc[c + ez + ex], c011 = xc[c + ez + ey], c111 = xc[c + ez + ey + ex];
      const double v00 = 0.5 * (c000 + c100), v10 = 0.5 * (c010 + c110), v01 = 0.5 * (c001 + c101),
                   v11 = 0.5 * (c011 + c111);
      if (oy && oz) s = 0.25 * (v00 + v10 + v01 + v11);
      else if (oy | oz) s = 0.5 * (v00 + (oy ? v10 : v01));
      else s = v00;
    } else {
      const int64_t o = ox * sx + oy * sy + oz * sz;
      if (TK == 2 && ox && oy) s = 0.5 * (xc[c + sx] + xc[c + sy]);
      else s = (o == 0) ? xc[c] : 0.5 * (xc[c] + xc[c + o]);
    }
    xf[id] += s;
  }
}

// one pass over an operator for the smoother's two ingredients: dinv_i = 1 / a_ii (1 for an empty diagonal) and the
// bound max_i sum_j |a_ij| / |a_ii| of the spectrum of D^-1 A (integer atomic max on the bit pattern of the
// non-negative double).  8 lanes per row, 16-byte aligned loads from the row
// start rounded down to a multiple of 4 (the layout of the SpMV kernel; buffers carry slack for the over-read).
__global__ __launch_bounds__(256) void k_diag_lam(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  const double* __restrict__ val, int64_t n, double* __restrict__ dinv,
                                                  unsigned long long* __restrict__ out) {
  const int sub = threadIdx.x & 7;
  double best = 0.0;
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3; row < n;
       row += ((int64_t)gridDim.x * blockDim.x) >> 3) {
    const int64_t s0 = rowptr[row], e0 = rowptr[row + 1];
    double s = 0.0, d = 0.0;
    for (int64_t base = (s0 & ~(int64_t)3) + 4 * sub; base < e0; base += 32) {
      const int4 c = *reinterpret_cast<const int4*>(col + base);
      const double2 v01 = *reinterpret_cast<const double2*>(val + base), v23 = *reinterpret_cast<const double2*>(val + base + 2);
      const int32_t cj[4] = {c.x, c.y, c.z, c.w};
      const double vv[4] = {v01.x, v01.y, v23.x, v23.y};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool in = base + t >= s0 && base + t < e0;
        s += in ? fabs(vv[t]) : 0.0;
        d += (in && (int64_t)cj[t] == row) ? vv[t] : 0.0;
      }
    }
    s += __shfl_down(s, 4, 8); d += __shfl_down(d, 4, 8);
    s += __shfl_down(s, 2, 8); d += __shfl_down(d, 2, 8);
    s += __shfl_down(s, 1, 8); d += __shfl_down(d, 1, 8);
    if (sub == 0) {
      const double di = (d != 0.0) ? 1.0 / d : 1.0;
      dinv[row] = di;
      s *= fabs(di);
      best = s > best ? s : best;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_down(best, o, 64);
    best = t > best ? t : best;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(best));
}

// d = dinv .* r / theta ; x = d (zero guess) or x += d
// (wp != null: 1 / theta is read from device memory instead of the argument)
__global__ void k_cheb_init(double* __restrict__ x, double* __restrict__ d, const double* __restrict__ r,
                            const double* __restrict__ dinv, double inv_theta, int zero_guess, int write_d, int64_t n,
                            const double* __restrict__ wp = nullptr) {
  if (wp) inv_theta = *wp;
  NODE_LOOP(i, n) {
    const double di = dinv[i] * r[i] * inv_theta;
    if (write_d) d[i] = di;  // only the multi-step recurrence needs the direction
    x[i] = zero_guess ? di : x[i] + di;
  }
}

// r -= t ; d = c1 d + c2 dinv .* r ; x += d      (t = A d_old)
__global__ void k_cheb_step(double* __restrict__ x, double* __restrict__ d, double* __restrict__ r,
                            const double* __restrict__ t, const double* __restrict__ dinv, double c1, double c2,
                            int64_t n) {
  NODE_LOOP(i, n) {
    const double ri = r[i] - t[i];
    r[i] = ri;
    const double di = c1 * d[i] + c2 * dinv[i] * ri;
    d[i] = di;
    x[i] += di;
  }
}

// smoother weight 1 / (0.5 (hi + lo)), lo = MG_CHEB_LOWER hi, of every level and field from the bounds' bit patterns - the
// arithmetic of cheb_w below, every operation rounded by itself (no contraction: the host's value bit for bit)
__global__ void k_mg_weights(const unsigned long long* __restrict__ bits, double* __restrict__ w, int count) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const double hi = __longlong_as_double((long long)bits[i]);
    const double lo = __dmul_rn(MG_CHEB_LOWER, hi);
    w[i] = __ddiv_rn(1.0, __dmul_rn(0.5, __dadd_rn(hi, lo)));
  }
}

static inline double cheb_w(const MgLevel& L, int which) {
  const double hi = L.lam[which], lo = MG_CHEB_LOWER * hi;
  return 1.0 / (0.5 * (hi + lo));
}
// its device copy (refreshed by mg_setup)
static inline const double* cheb_wp(const pph_ctx* ctx, int l, int which) { return ctx->mg_w.p + 2 * l + which; }

static inline int mg_grid(int64_t n) {
  // (round 4, option transfer_bench: without the cap the ISOLATED fine-level interpolation runs 0.074 instead of 0.099 ms - one
  // pair per thread, no loop - but the 256^3 step does not move (20.6 - 20.8 ms at 2 048, 65 536 and 10^6 blocks): inside
  // the cycle the kernel waits for operands the previous kernel just wrote, not for its own index arithmetic)
  int64_t b = ceil_div64(n, 256);
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

// lowp: the fp32 copy of the level operator (smoother / residual SpMVs inside the V-cycle)
static Csr level_csr(const pph_ctx* ctx, const MgLevel& L, int which, bool lowp = false) {
  Csr A;
  A.rowptr = L.rowptr; A.col = L.col; A.val = L.val[which]; A.nrows = L.n; A.nnz = L.nnz;
  if (lowp && ctx->mg_fp32 && L.val32[which].p) A.val32 = L.val32[which].p;
  else A.ell = L.ell[which];
  A.max_row = ctx->mesh.max_row;
  A.geom = (ctx->world > 1 && !L.replicated) ? L.geom : nullptr;
  A.lanes = pph_pick_lanes(ctx, A.nnz, A.nrows);
  return A;
}

static TGeom tgeom(const MgLevel& F, const MgLevel& C) {
  TGeom g;
  g.pxc = C.px; g.pyc = C.py; g.pzc = C.pz; g.gzc = C.gz0;
  g.pxf = F.px; g.pyf = F.py; g.pzf = F.pz; g.gzf = F.gz0;
  g.own_lo_f = F.own_lo; g.own_hi_f = F.own_hi;
  return g;
}

// isolated timing of the fine-level transfer kernels (hexahedra; tools/r4_transfer_probe.py): reps launches each of the
// interpolation t = x + P x_c (levels 0 <- 1) and of the restriction b_c = R r (+ the coarse pre-smoothing), ms per launch
int mg_transfer_bench(pph_ctx* ctx, int which, int reps, double* out2) {
  PPH_REQUIRE(ctx, ctx->mg_ok && ctx->mg.size() >= 2 && ctx->mesh.kind == PPH_CELL_HEX, "transfer bench: needs a hexahedral hierarchy");
  MgLevel& L = ctx->mg[0];
  MgLevel& C = ctx->mg[1];
  const TGeom tg = tgeom(L, C);
  float ms = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < reps + 5; ++i) {
      if (i == 5) PPH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
      if (pass == 0)
        hipLaunchKernelGGL(k_prolong_to_q1, dim3(mg_grid((L.n + 1) / 2 + L.py * L.pz)), dim3(256), 0, ctx->stream, L.t.p, L.d.p, C.x.p,
                           L.maskp[which], tg);
      else
        hipLaunchKernelGGL(k_restrict_q1<3>, dim3(mg_grid(C.n)), dim3(256), 0, ctx->stream, C.b.p, L.r.p, C.maskp[which],
                           L.maskp[which], tg, C.x.p, C.dinv[which].p, cheb_wp(ctx, 1, which), C.rfast[which].p);
    }
    PPH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    PPH_HIP(ctx, hipEventSynchronize(ctx->ev1));
    PPH_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    out2[pass] = (double)ms / reps;
  }
  return PPH_OK;
}

void mg_release(pph_ctx* ctx) {
  for (size_t l = 0; l < ctx->mg.size(); ++l) {
    MgLevel& L = ctx->mg[l];
    if (l > 0) L.mesh.release_all();
    for (int f = 0; f < 2; ++f) {
      L.own_val[f].release(); L.own_ell[f].release(); L.dict[f].release(); L.dgroup.release(); L.dinv[f].release(); L.mask[f].release(); L.val32[f].release();
      L.ell[f] = Sell();
    }
    L.x.release(); L.b.release(); L.r.release(); L.d.release(); L.t.release(); L.w.release();
  }
  ctx->mg.clear();
  ctx->mg_w.release();
  for (int f = 0; f < 2; ++f) { ctx->mg_tail_pack[f].release(); ctx->mg_tail_lt[f] = -1; }
  la_release_graphs(ctx);   // captured iteration bodies hold the hierarchy's pointers
  ctx->mg_epoch++;
  ctx->mg_ok = false;
  ctx->mg_struct_ok = false;
}

// minimum of one integer per rank (collective); identity on a single rank
static int comm_min_int(pph_ctx* ctx, int v, int* out) {
  *out = v;
  if (ctx->world <= 1) return PPH_OK;
  std::vector<double> buf((size_t)ctx->world, 0.0);
  buf[(size_t)ctx->rank] = (double)v;
  PPH_TRY(comm_allreduce_host(ctx, buf.data(), (int64_t)ctx->world));
  double m = buf[0];
  for (double b : buf) m = b < m ? b : m;
  *out = (int)m;
  return PPH_OK;
}

static int comm_max_double(pph_ctx* ctx, double v, double* out) {
  *out = v;
  if (ctx->world <= 1) return PPH_OK;
  std::vector<double> buf((size_t)ctx->world, 0.0);
  buf[(size_t)ctx->rank] = v;
  PPH_TRY(comm_allreduce_host(ctx, buf.data(), (int64_t)ctx->world));
  double m = buf[0];
  for (double b : buf) m = b > m ? b : m;
  *out = m;
  return PPH_OK;
}

// Hierarchy.  Single GPU: levels nx/2^l while every direction stays even with >= 2 cells.  Slabs: the
// same global hierarchy; level l is DISTRIBUTED (each rank a sub-slab with ghost planes) while the owned
// cell range of every rank is divisible by 2^l and keeps >= 2 layers, and REPLICATED (whole coarse mesh on
// every rank, right-hand side summed by one all-reduce) below that, so the cycle equals the 1-GPU one.
static int mg_tail_pack(pph_ctx* ctx, int which);

int mg_setup(pph_ctx* ctx) {
  if (ctx->mg_ok) return PPH_OK;
  const bool build = !ctx->mg_struct_ok;  // level meshes / patterns / buffers survive re-assembly
  if (build) mg_release(ctx);
  const MeshData& fm = ctx->mesh;
  const bool dist = ctx->world > 1;
  PPH_REQUIRE(ctx, !dist || fm.dim == 3, "slab decomposition needs a 3D mesh");
  int nlev = 1;
  {
    int nx = fm.nx, ny = fm.ny, nz = (fm.dim == 3) ? fm.nz : 0;
    while (nx % 2 == 0 && ny % 2 == 0 && (fm.dim == 2 || nz % 2 == 0) && nx / 2 >= 2 && ny / 2 >= 2 &&
           (fm.dim == 2 || nz / 2 >= 2)) {
      nx /= 2; ny /= 2; nz /= 2; nlev++;
    }
  }
  // owned cell layers [c0, c1) of this rank (the local box has one extra layer below when glo)
  const int c0 = fm.z0 + fm.glo, c1 = fm.z0 + fm.nzl;
  int ndist = nlev;  // levels [0, ndist) are distributed
  if (dist) {
    int l = 0;
    while (l + 1 < nlev && (c0 % (2 << l)) == 0 && (c1 % (2 << l)) == 0 && ((c1 - c0) >> (l + 1)) >= 2) ++l;
    // small levels are latency-bound: below `mg_replicate_below` global nodes a level is replicated (one
    // all-reduce of its right-hand side per cycle) instead of distributed (five halo exchanges per cycle) - and so is
    // a level whose share PER RANK is at most `mg_replicate_rows_per_rank` nodes (its slab kernels are shorter than one
    // exchange: 256^3 on 8 ranks, level 2 = 65^3 = 34 k nodes per rank) while the whole level stays small enough to be
    // cheap on every rank (`mg_replicate_cap` global nodes)
    int lr = l + 1;
    for (int q = 1; q <= l; ++q) {
      const int64_t gn = (int64_t)((fm.nx >> q) + 1) * ((fm.ny >> q) + 1) * ((fm.nz >> q) + 1);
      const bool small_share = gn / ctx->world <= ctx->mg_replicate_rows_per_rank && gn <= ctx->mg_replicate_cap;
      if (gn <= ctx->mg_replicate_below || small_share) { lr = q; break; }
    }
    PPH_TRY(comm_min_int(ctx, lr, &ndist));
  }
  if (build) ctx->mg.resize(nlev);
  PPH_REQUIRE(ctx, (int)ctx->mg.size() == nlev, "multigrid hierarchy out of date");
  DevBuf<unsigned long long>& lamdev = ctx->mg_lam;
  DevBuf<double> mtmp;
  PPH_REQUIRE(ctx, nlev <= 32, "more than 32 multigrid levels");
  if (lamdev.n < (size_t)(2 * nlev)) PPH_TRY(lamdev.alloc(ctx, (size_t)64));   // spectral bounds of all levels (kept across assemblies)
  la_set(ctx, reinterpret_cast<double*>(lamdev.p), 0.0, 2 * nlev);
  const double coefK[2] = {ctx->a, ctx->c};
  // operator format of the levels: stencil-ELL (default) or CSR; the fp32 option keeps CSR values
  const bool use_ell = ctx->op_format == 1 && !ctx->mg_fp32;
  if (!use_ell || !ctx->ell_ok) PPH_TRY(pph_ensure_csr_blocks(ctx));
  for (int l = 0; l < nlev; ++l) {
    MgLevel& L = ctx->mg[l];
    bool level_fused = false;   // dinv / spectral bound of this level already produced by the fused pass
    L.replicated = dist && l >= ndist;
    if (l == 0) {
      L.rowptr = fm.rowptr.p; L.col = fm.col.p;
      L.val[0] = ctx->csr_ok ? ctx->A11.p : nullptr; L.val[1] = ctx->csr_ok ? ctx->A22.p : nullptr;
      L.ell[0] = (use_ell && ctx->ell_ok) ? ctx->S11 : Sell();
      L.ell[1] = (use_ell && ctx->ell_ok) ? ctx->S22 : Sell();
      L.n = fm.n; L.nnz = fm.nnzb; L.px = fm.px; L.py = fm.py; L.pz = fm.pzl;
      L.maskp[0] = ctx->bcmask[0].p; L.maskp[1] = ctx->bcmask[1].p;
      L.geom = &ctx->mesh;
    } else {
      const MgLevel& F = ctx->mg[l - 1];
      MeshData& m = L.mesh;
      m.dim = fm.dim; m.kind = fm.kind;
      m.nx = fm.nx >> l; m.ny = fm.ny >> l; m.nz = (fm.dim == 3) ? (fm.nz >> l) : 0;
      if (fm.dim == 2) { m.z0 = 0; m.nzl = 0; m.glo = m.ghi = 0; }
      else if (L.replicated || !dist) { m.z0 = 0; m.nzl = m.nz; m.glo = m.ghi = 0; }
      else { m.glo = fm.glo; m.ghi = fm.ghi; m.z0 = (c0 >> l) - m.glo; m.nzl = (c1 >> l) - m.z0; }
      if (build) PPH_TRY(pph_launch_mesh(ctx, m));
      // level operators straight from the element rows (fused pass) or from K, M of this level (two-step path)
      const bool fuse_lv = pph_can_fuse_assembly(ctx);
      if (!(use_ell && fuse_lv)) PPH_TRY(pph_ensure_pattern(ctx, m));   // CSR level operators: the pattern of this level
      if (!fuse_lv && !m.km_valid) {
        PPH_TRY(pph_launch_assemble_KM(ctx, m));
        m.km_valid = true;
      }
      L.rowptr = m.rowptr.p; L.col = m.col.p; L.n = m.n; L.nnz = m.nnzb; L.px = m.px; L.py = m.py; L.pz = m.pzl;
      L.geom = &L.mesh;
      L.gz0 = m.z0;
      L.own_lo = m.z0 + m.glo; L.own_hi = m.z0 + m.pzl - m.ghi;
      const TGeom tg = tgeom(F, L);
      // injected masks depend on the Dirichlet sets only: rebuilt when those (or the hierarchy) change
      const bool masks_stale = build || L.bc_epoch != ctx->bc_epoch;
      if (masks_stale) {
        PPH_TRY(mtmp.alloc(ctx, (size_t)L.n));
        for (int f = 0; f < 2; ++f) {
          PPH_TRY(L.mask[f].alloc(ctx, (size_t)L.n));
          hipLaunchKernelGGL(k_inject_mask, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, mtmp.p, F.maskp[f], tg);
          if (L.replicated && !F.replicated) PPH_TRY(la_allreduce_vec(ctx, mtmp.p, L.n));
          else if (dist && !L.replicated) PPH_TRY(la_halo(ctx, m, mtmp.p));
          hipLaunchKernelGGL(k_mask_from_double, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, L.mask[f].p, mtmp.p, L.n,
                             m.plane(), m.glo, m.ghi);
          // restriction fast path of the multilinear transfers
          if (fm.kind == PPH_CELL_HEX || fm.kind == PPH_CELL_QUAD) {
            PPH_TRY(L.rfast[f].alloc(ctx, (size_t)L.n));
            if (fm.kind == PPH_CELL_HEX)
              hipLaunchKernelGGL(k_restrict_flags<3>, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, L.rfast[f].p, L.mask[f].p,
                                 F.maskp[f], tg);
            else
              hipLaunchKernelGGL(k_restrict_flags<2>, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, L.rfast[f].p, L.mask[f].p,
                                 F.maskp[f], tg);
          }
        }
      }
      const bool ell_only = use_ell && fuse_lv;   // the fused pass writes the stencil-ELL arrays directly
      for (int f = 0; f < 2; ++f) {
        L.maskp[f] = L.mask[f].p;
        L.ell[f] = Sell();
        if (ell_only) {
          PPH_TRY(sell_alloc(ctx, m, L.own_ell[f], &L.ell[f], pph_sell_sym(ctx)));
          L.val[f] = nullptr;
        } else {
          PPH_TRY(L.own_val[f].alloc(ctx, (size_t)L.nnz));
          L.val[f] = L.own_val[f].p;
        }
      }
      if (fuse_lv) {
        if (masks_stale) {
          PPH_TRY(L.rownear.alloc(ctx, (size_t)L.n));
          pph_launch_row_near(ctx, m, L.maskp[0], L.maskp[1], L.rownear.p);
        }
        for (int f = 0; f < 2; ++f) PPH_TRY(L.dinv[f].alloc(ctx, (size_t)L.n));
        PPH_TRY(pph_launch_level_operators(ctx, m, L.maskp[0], L.maskp[1], L.rownear.p, ctx->a21_alias ? 1 : 0, coefK[0],
                                           coefK[1], ctx->b, ell_only ? L.own_ell[0].p : L.own_val[0].p,
                                           ell_only ? L.own_ell[1].p : L.own_val[1].p, L.dinv[0].p, L.dinv[1].p,
                                           lamdev.p + 2 * l, ell_only ? L.ell[0].ld : 0, ell_only ? L.ell[0].sym : 0,
                                           ell_only ? &L.dgroup : nullptr, ell_only ? L.dict : nullptr, ell_only ? L.ell : nullptr));
        level_fused = true;
        if (ell_only) {
          const int b0 = ctx->n_dict_build;
          for (int f = 0; f < 2; ++f) PPH_TRY(sell_dict_update(ctx, &L.ell[f], L.dict[f], L.n));
          if (ctx->n_dict_build != b0) {     // dictionaries (re)built: their fused-check group follows
            SellDict* ds[2] = {L.dict[0].on ? &L.dict[0] : nullptr, L.dict[1].on ? &L.dict[1] : nullptr};
            L.dgroup.release();
            if (ds[0] && ds[1]) PPH_TRY(dict_group_build(ctx, L.dgroup, ds, 2, L.ell[0], L.n));
          }
        }
      } else {
        for (int f = 0; f < 2; ++f) {
          pph_launch_scalar_block(ctx, m, L.maskp[f], coefK[f], ctx->b, L.own_val[f].p);
          if (use_ell) PPH_TRY(sell_from_csr(ctx, m, L.own_val[f].p, L.own_ell[f], &L.ell[f], pph_sell_sym_from_csr(ctx)));
        }
      }
      L.bc_epoch = ctx->bc_epoch;
    }
    if (l == 0) {
      L.gz0 = fm.z0;
      L.own_lo = fm.z0 + fm.glo; L.own_hi = fm.z0 + fm.pzl - fm.ghi;
      if (fm.dim == 2) { L.gz0 = 0; L.own_lo = 0; L.own_hi = 1; }
    } else if (fm.dim == 2) {
      L.gz0 = 0; L.own_lo = 0; L.own_hi = 1;
    }
    for (int f = 0; f < 2; ++f) {
      PPH_TRY(L.dinv[f].alloc(ctx, (size_t)L.n));
      if (ctx->mg_fp32) {
        PPH_TRY(L.val32[f].alloc(ctx, (size_t)L.nnz));
        hipLaunchKernelGGL(k_to_float, dim3(mg_grid(L.nnz)), dim3(256), 0, ctx->stream, L.val32[f].p, L.val[f], L.nnz);
      }
    }
    if (l == 0 && ctx->diag0_valid) {
      // the fused assembly already produced the fine-level diagonal inverses and bounds
      for (int f = 0; f < 2; ++f)
        la_copy(ctx, L.dinv[f].p, ctx->dinv0[f].p, L.n);
      la_copy(ctx, reinterpret_cast<double*>(lamdev.p + 2 * l), reinterpret_cast<const double*>(ctx->lam0.p), 2);
    } else if (!level_fused)
    for (int f = 0; f < 2; ++f)
      hipLaunchKernelGGL(k_diag_lam, dim3(mg_grid(L.n * 8)), dim3(256), 0, ctx->stream, L.rowptr, L.col, L.val[f], L.n,
                         L.dinv[f].p, lamdev.p + 2 * l + f);
    if (l > 0) { PPH_TRY(L.x.alloc(ctx, (size_t)L.n)); PPH_TRY(L.b.alloc(ctx, (size_t)L.n)); }
    PPH_TRY(L.r.alloc(ctx, (size_t)L.n));
    PPH_TRY(L.d.alloc(ctx, (size_t)L.n));
    PPH_TRY(L.t.alloc(ctx, (size_t)L.n));
    if (l == nlev - 1) PPH_TRY(L.w.alloc(ctx, (size_t)L.n));
  }
  // Smoother weights of the fused cycle, read by its kernels from device memory - computed ON the device from the bounds
  // (round 4): the host neither waits for the assembly's kernels here nor uploads anything.  The bounds travel to pinned
  // host memory behind an event; MgLevel::lam is filled from there when host code needs it (mg_lam_host).  Slabs take the
  // maximum over the ranks on the host and upload the weights, as before.
  PPH_TRY(ctx->mg_w.alloc(ctx, (size_t)(2 * nlev)));
  PPH_HIP(ctx, hipMemcpyAsync(ctx->h_lam, lamdev.p, sizeof(unsigned long long) * (size_t)(2 * nlev), hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipEventRecord(ctx->ev_lam, ctx->stream));
  ctx->mg_lam_pending = true;
  if (dist) {
    PPH_TRY(mg_lam_host(ctx));
    ctx->mg_w_host.resize((size_t)(2 * nlev));
    for (int l = 0; l < nlev; ++l)
      for (int f = 0; f < 2; ++f) ctx->mg_w_host[(size_t)(2 * l + f)] = cheb_w(ctx->mg[l], f);
    PPH_HIP(ctx, hipMemcpyAsync(ctx->mg_w.p, ctx->mg_w_host.data(), sizeof(double) * ctx->mg_w_host.size(),
                                hipMemcpyHostToDevice, ctx->stream));
  } else {
    hipLaunchKernelGGL(k_mg_weights, dim3(1), dim3(64), 0, ctx->stream, lamdev.p, ctx->mg_w.p, 2 * nlev);
  }
  mtmp.release();
  PPH_HIP(ctx, hipGetLastError());
  if (build) ctx->mg_epoch++;
  ctx->mg_struct_ok = true;
  ctx->mg_ok = true;
  for (int f = 0; f < 2; ++f) PPH_TRY(mg_tail_pack(ctx, f));
  return PPH_OK;
}

int mg_lam_host(pph_ctx* ctx) {
  if (!ctx->mg_lam_pending) return PPH_OK;
  PPH_HIP(ctx, hipEventSynchronize(ctx->ev_lam));
  ctx->mg_lam_pending = false;
  const bool dist = ctx->world > 1;
  const int nlev = (int)ctx->mg.size();
  for (int l = 0; l < nlev; ++l) {
    MgLevel& L = ctx->mg[l];
    for (int f = 0; f < 2; ++f) {
      double v;
      memcpy(&v, &ctx->h_lam[(size_t)(2 * l + f)], sizeof(double));
      if (dist && !L.replicated) PPH_TRY(comm_max_double(ctx, v, &v));
      L.lam[f] = v;
      PPH_REQUIRE(ctx, v > 0.0 && v == v, "multigrid level %d: bad spectral bound %g", l, v);
    }
  }
  return PPH_OK;
}

// `steps` Chebyshev-Jacobi steps on A x = b.  zero_guess: x is overwritten, no initial SpMV.
static void chebyshev(pph_ctx* ctx, MgLevel& L, int which, const double* b, double* x, int steps, bool zero_guess) {
  const Csr A = level_csr(ctx, L, which, true);
  if (steps == 1 && ctx->world == 1) {
    // one step = a weighted Jacobi sweep: the weight is read from the device array (no host copy of the bound needed)
    double* r1 = L.r.p;
    const double* r01 = b;
    if (!zero_guess) { la_spmv_resid(ctx, A, x, b, r1); r01 = r1; }
    hipLaunchKernelGGL(k_cheb_init, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, x, L.d.p, r01, L.dinv[which].p, 0.0,
                       zero_guess ? 1 : 0, 0, L.n, cheb_wp(ctx, (int)(&L - ctx->mg.data()), which));
    return;
  }
  if (mg_lam_host(ctx) != PPH_OK) return;
  const double hi = L.lam[which], lo = MG_CHEB_LOWER * hi;
  const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo);
  const double sigma = theta / delta;
  double rho = 1.0 / sigma;
  double* r = L.r.p;
  const int grid = mg_grid(L.n);
  const double* r0 = r;  // residual the first step starts from
  if (zero_guess) {
    if (steps > 1) la_copy(ctx, r, b, L.n);  // the recurrence updates r in place
    else r0 = b;                              // single step: read b directly
  } else {
    la_spmv_resid(ctx, A, x, b, r);
  }
  hipLaunchKernelGGL(k_cheb_init, dim3(grid), dim3(256), 0, ctx->stream, x, L.d.p, r0, L.dinv[which].p, 1.0 / theta,
                     zero_guess ? 1 : 0, steps > 1 ? 1 : 0, L.n);
  for (int s = 1; s < steps; ++s) {
    la_spmv(ctx, A, L.d.p, L.t.p);
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    hipLaunchKernelGGL(k_cheb_step, dim3(grid), dim3(256), 0, ctx->stream, x, L.d.p, r, L.t.p, L.dinv[which].p,
                       rho_new * rho, 2.0 * rho_new / delta, L.n);
    rho = rho_new;
  }
}

// Coarsest-level solve without the host: one workgroup runs the whole Jacobi-preconditioned CG (same recurrence,
// same test on ||D^-1 r|| <= rtol ||D^-1 b||, at most max_it iterations as pph_cg_jacobi) on a system of a few
// thousand rows at most.  The host-driven CG costs two stream synchronisations per iteration, which is what a
// V-cycle on a small or distributed problem mostly waits for.  Block reductions are fixed-order (deterministic).
__device__ __forceinline__ double coarse_block_sum(double v, double* lds) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();                       // lds reuse across calls
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += lds[w];
  return t;
}

__global__ __launch_bounds__(1024) void k_coarse_cg(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const double* __restrict__ val, const double* __restrict__ dinv,
                                                    const double* __restrict__ b, double* __restrict__ x,
                                                    double* __restrict__ r, double* __restrict__ p, double* __restrict__ q,
                                                    int n, double rtol, int max_it) {
  __shared__ double lds[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  double zz = 0.0, rz = 0.0;
  for (int i = tid; i < n; i += nt) {
    const double ri = b[i], zi = dinv[i] * ri;
    x[i] = 0.0; r[i] = ri; p[i] = zi;
    zz += zi * zi; rz += ri * zi;
  }
  zz = coarse_block_sum(zz, lds);
  rz = coarse_block_sum(rz, lds);
  const double tol = rtol * sqrt(zz);
  if (!(sqrt(zz) > tol)) return;         // zero (or NaN) right-hand side: x = 0
  for (int it = 0; it < max_it; ++it) {
    __syncthreads();                     // p complete
    double pq = 0.0;
    for (int i = tid; i < n; i += nt) {
      double s = 0.0;
      for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * p[col[k]];
      q[i] = s;
      pq += p[i] * s;
    }
    pq = coarse_block_sum(pq, lds);
    if (!(pq > 0.0)) return;
    const double alpha = rz / pq;
    double zz2 = 0.0, rz2 = 0.0;
    for (int i = tid; i < n; i += nt) {
      x[i] += alpha * p[i];
      const double ri = r[i] - alpha * q[i], zi = dinv[i] * ri;
      r[i] = ri;
      zz2 += zi * zi; rz2 += ri * zi;
    }
    zz2 = coarse_block_sum(zz2, lds);
    rz2 = coarse_block_sum(rz2, lds);
    if (sqrt(zz2) <= tol) return;
    const double beta = rz2 / rz;
    __syncthreads();                     // every thread has read p[col] of this iteration's product
    for (int i = tid; i < n; i += nt) p[i] = dinv[i] * r[i] + beta * p[i];
    rz = rz2;
  }
}

// the same solve on a stencil-ELL operator (no CSR values exist on a level the fused pass wrote in that form)
__global__ __launch_bounds__(1024) void k_coarse_cg_sell(const double* __restrict__ val, int64_t ld, int sym, Stencil st,
                                                         int px, int py, int pz, const double* __restrict__ dinv,
                                                         const double* __restrict__ b, double* __restrict__ x,
                                                         double* __restrict__ r, double* __restrict__ p,
                                                         double* __restrict__ q, int n, double rtol, int max_it) {
  __shared__ double lds[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  double zz = 0.0, rz = 0.0;
  for (int i = tid; i < n; i += nt) {
    const double ri = b[i], zi = dinv[i] * ri;
    x[i] = 0.0; r[i] = ri; p[i] = zi;
    zz += zi * zi; rz += ri * zi;
  }
  zz = coarse_block_sum(zz, lds);
  rz = coarse_block_sum(rz, lds);
  const double tol = rtol * sqrt(zz);
  if (!(sqrt(zz) > tol)) return;
  for (int it = 0; it < max_it; ++it) {
    __syncthreads();
    double pq = 0.0;
    for (int i = tid; i < n; i += nt) {
      const int ii = i % px, t = i / px, jj = t % py, kk = t / py;
      double s = 0.0;
      for (int sl = 0; sl < st.count; ++sl) {
        const int a = ii + st.d[sl][0], bb = jj + st.d[sl][1], c = kk + st.d[sl][2];
        if (a >= 0 && a < px && bb >= 0 && bb < py && c >= 0 && c < pz) {
          const int jn = a + px * (bb + py * c);
          const int c0 = st.count / 2;
          // symmetric storage: lower entry (i, jn) = upper entry (jn, i), mirror slot, stored at row jn
          const double v = !sym ? val[(int64_t)sl * ld + i]
                                : (sl >= c0 ? val[(int64_t)(sl - c0) * ld + i] : val[(int64_t)(st.count - 1 - sl - c0) * ld + jn]);
          s += v * p[jn];
        }
      }
      q[i] = s;
      pq += p[i] * s;
    }
    pq = coarse_block_sum(pq, lds);
    if (!(pq > 0.0)) return;
    const double alpha = rz / pq;
    double zz2 = 0.0, rz2 = 0.0;
    for (int i = tid; i < n; i += nt) {
      x[i] += alpha * p[i];
      const double ri = r[i] - alpha * q[i], zi = dinv[i] * ri;
      r[i] = ri;
      zz2 += zi * zi; rz2 += ri * zi;
    }
    zz2 = coarse_block_sum(zz2, lds);
    rz2 = coarse_block_sum(rz2, lds);
    if (sqrt(zz2) <= tol) return;
    const double beta = rz2 / rz;
    __syncthreads();
    for (int i = tid; i < n; i += nt) p[i] = dinv[i] * r[i] + beta * p[i];
    rz = rz2;
  }
}

void mg_onchip_cg(pph_ctx* ctx, const Sell& E, const double* dinv, const double* b, double* x, double* r, double* p, double* q,
                  int64_t n, double rtol, int max_it) {
  hipLaunchKernelGGL(k_coarse_cg_sell, dim3(1), dim3(n <= 256 ? 256 : 1024), 0, ctx->stream, E.val, E.ld, E.sym, make_stencil(E.kind),
                     E.px, E.py, E.pz, dinv, b, x, r, p, q, (int)n, rtol, max_it);
}

// ------------------------------------------------------------------------------------------------
// Tail of the V(1,1) cycle in ONE workgroup: the coarsest levels (from the first one with at most 1024 rows: 9^3
// and coarser in 3D) are swept inside a single launch, entirely on-chip - every vector of every tail level lives
// in LDS, the operator row of the largest tail level in the registers of the thread that owns the row, the smaller
// operators in LDS - with workgroup barriers between the phases instead of a kernel launch per operation and
// a global-memory round trip per phase.  Pre-smoothing, residual and restriction on the way down, the coarsest
// Jacobi-CG, interpolation and post-smoothing on the way up: the arithmetic of the kernel-per-operation cycle
// (same stencil orders, same smoother).  A 256^3 cycle has 3 such levels = 16 of its launches; a 64^3 step spends
// a third of its time in them.
// ------------------------------------------------------------------------------------------------
#define MG_TAIL_MAX 4
#define MG_TAIL_ROWS 1024          // rows of the largest tail level = threads of the workgroup
#define MG_TAIL_MATPOOL 6144       // doubles of LDS for the operators of tail levels 1..
// The operators, inverse diagonals and masks of the tail levels are the same for every cycle of a solve: mg_setup
// packs them once per assembly (k_tail_pack) into one buffer in the order the kernel consumes them - level 0 as full
// rows, slot-major (symmetric storage resolved: thread i reads its S entries coalesced), the operators of levels
// 1.. as the LDS image (one flat copy), then the inverse diagonals, then the masks as bytes.
struct TailLevel {
  int px, py, pz, n;
  const double* w;        // 1 / theta of the one-step Chebyshev smoother (device)
};
struct TailArgs {
  TailLevel L[MG_TAIL_MAX];
  const double* pack;
  const double* b0;       // right-hand side of tail level 0 (global)
  double* x0;             // its solution (global)
};
struct TailSrc {          // source of the pack: the level's stencil-ELL operator
  const double* A;
  int sym;
  int64_t ld;
  const double* dinv;
  const uint8_t* mask;
  int px, py, pz, n;
};
struct TailPackArgs {
  int nl;
  TailSrc L[MG_TAIL_MAX];
};

// doubles of the pack in front of the masks / bytes of the masks (each level padded to 8)
__host__ __device__ inline int tail_pack_doubles(const int* n, int nl, int S) {
  int d = 0;
  for (int l = 0; l < nl; ++l) d += (S + 1) * n[l];
  return d;
}
__host__ __device__ inline int tail_pack_maskbytes(const int* n, int nl) {
  int b = 0;
  for (int l = 0; l < nl; ++l) b += (n[l] + 7) & ~7;
  return b;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_tail_pack(TailPackArgs pa, double* __restrict__ pack) {
  using ST = SellSt<KIND>;
  constexpr int S = ST::S;
  const int nl = pa.nl;
  int ns[MG_TAIL_MAX];
  for (int l = 0; l < MG_TAIL_MAX; ++l) ns[l] = l < nl ? pa.L[l].n : 0;
  const int gt = blockIdx.x * blockDim.x + threadIdx.x, gn = gridDim.x * blockDim.x;
  // operators: level 0 at 0, its inverse diagonal behind it, then levels 1.., then their inverse diagonals
  int offA = 0, offD = 0;
  for (int l = 1; l < nl; ++l) offD += S * ns[l];
  offD += (S + 1) * ns[0];
  for (int l = 0; l < nl; ++l) {
    const TailSrc F = pa.L[l];
    const int base = (l == 0) ? 0 : offA;
    for (int e = gt; e < S * F.n; e += gn) {
      const int slot = e / F.n, row = e % F.n;
      double v;
      if (!F.sym) v = F.A[(int64_t)slot * F.ld + row];
      else {
        constexpr int C0 = S / 2;
        if (slot >= C0) v = F.A[(int64_t)(slot - C0) * F.ld + row];
        else {
          // lower entry = the mirror slot of the row it points to
          int sl = 0, off = 0;
          for (int q = 0; q < ST::NL; ++q) {
            const int mask = ST::mask(q);
            for (int d = 0; d < 3; ++d) {
              if (!((mask >> d) & 1)) continue;
              if (sl == slot) off = (d - 1) + ST::dy(q) * F.px + ST::dz(q) * F.px * F.py;
              ++sl;
            }
          }
          const int rr = row + off;
          v = (rr >= 0 && rr < F.n) ? F.A[(int64_t)(S - 1 - slot - C0) * F.ld + rr] : 0.0;
        }
      }
      pack[base + e] = v;
    }
    if (l == 0) {
      for (int i = gt; i < F.n; i += gn) pack[S * F.n + i] = F.dinv[i];
      offA = (S + 1) * F.n;
    } else {
      for (int i = gt; i < F.n; i += gn) pack[offD + i] = F.dinv[i];
      offA += S * F.n;
      offD += F.n;
    }
  }
  uint8_t* mb = reinterpret_cast<uint8_t*>(pack + tail_pack_doubles(ns, nl, S));
  for (int l = 0; l < nl; ++l) {
    const TailSrc F = pa.L[l];
    const int padded = (F.n + 7) & ~7;
    for (int i = gt; i < padded; i += gn) mb[i] = i < F.n ? F.mask[i] : 0;
    mb += padded;
  }
}

// row i of A v with the operator row in a[] (stride astride between slots: 1 for registers, n for an LDS operator
// stored slot-major) and v in LDS; neighbours outside the box carry a zero entry, their index is clamped
template <int KIND>
__device__ __forceinline__ double tail_row(const double* a, int astride, const double* v, int i, int px, int pxy, int n) {
  using ST = SellSt<KIND>;
  double s = 0.0;
  int slot = 0;
#pragma unroll
  for (int l = 0; l < ST::NL; ++l) {
    const int mask = ST::mask(l);
    if (mask == 0) continue;
    const int L = i + ST::dy(l) * px + ST::dz(l) * pxy;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (!((mask >> d) & 1)) continue;
      int idx = L + d - 1;
      idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
      s += a[slot * astride] * v[idx];
      ++slot;
    }
  }
  return s;
}

// weight of the transfer tap (dx, dy, dz) of make_transfer_stencil(kind); 0: not a tap
__host__ __device__ constexpr double tail_tap_weight(int kind, int dx, int dy, int dz) {
  const int nzc = (dx != 0) + (dy != 0) + (dz != 0);
  if (kind == PPH_CELL_QUAD || kind == PPH_CELL_HEX) return nzc == 0 ? 1.0 : (nzc == 1 ? 0.5 : (nzc == 2 ? 0.25 : 0.125));
  if (kind == PPH_CELL_TRI) return (dx * dy > 0) ? 0.0 : (nzc == 0 ? 1.0 : 0.5);
  const bool pos = dx >= 0 && dy >= 0 && dz >= 0, neg = dx <= 0 && dy <= 0 && dz <= 0;
  return (pos || neg) ? (nzc == 0 ? 1.0 : 0.5) : 0.0;
}

// sum over the first wave only (coarsest level of at most 64 rows: the other waves hold zeros - the result equals
// coarse_block_sum's, which adds their zeros in order)
__device__ __forceinline__ double tail_wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return __shfl(v, 0, 64);
}
#define TAIL_WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <int KIND, int NL>
__global__ __launch_bounds__(1024) void k_mg_tail(TailArgs ta, TStencil ts, int tk, double cg_rtol, int cg_max_it) {
  using ST = SellSt<KIND>;
  constexpr int S = ST::S;
  extern __shared__ double sm[];
  __shared__ double red[16];
  const int tid = threadIdx.x;
  // LDS layout: per level x, b, t (t = residual on the way down, interpolated iterate on the way up); the coarsest
  // level also p, q of its CG; then the operators of levels 1.. (slot-major), then the masks as bytes
  double *X[NL], *B[NL], *T[NL], *M[NL];
  uint8_t* MK[NL];
  int ns[NL];
  int off = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    ns[l] = ta.L[l].n;
    X[l] = sm + off; B[l] = X[l] + ns[l]; T[l] = B[l] + ns[l];
    off += 3 * ns[l];
  }
  double* P = sm + off; double* Q = P + ns[NL - 1];
  off += 2 * ns[NL - 1];
  double* const Mbase = sm + off;
  int mat_tot = 0;
  M[0] = nullptr;
#pragma unroll
  for (int l = 1; l < NL; ++l) { M[l] = Mbase + mat_tot; mat_tot += S * ns[l]; }
  uint8_t* const MKbase = reinterpret_cast<uint8_t*>(Mbase + mat_tot);
  int mk_tot = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) { MK[l] = MKbase + mk_tot; mk_tot += (ns[l] + 7) & ~7; }
  // ---- load from the pack: operator row of level 0 into registers, the other operators and the masks into LDS
  const double* pk = ta.pack;
  const int n0 = ns[0];
  double a0[S];
  double dv[NL], wl[NL];
  {
    const bool in = tid < n0;
#pragma unroll
    for (int s = 0; s < S; ++s) a0[s] = in ? pk[s * n0 + tid] : 0.0;
    dv[0] = in ? pk[S * n0 + tid] : 0.0;
    const double* mats = pk + (S + 1) * n0;
    // (batches of eight loads in flight: one memory round trip for the usual sizes instead of one per element)
    for (int e0 = tid; e0 < mat_tot; e0 += 8 * (int)blockDim.x) {
      double tmp[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = e0 + k * (int)blockDim.x;
        tmp[k] = e < mat_tot ? mats[e] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = e0 + k * (int)blockDim.x;
        if (e < mat_tot) Mbase[e] = tmp[k];
      }
    }
    const double* dl = mats + mat_tot;
#pragma unroll
    for (int l = 1; l < NL; ++l) { dv[l] = (tid < ns[l]) ? dl[tid] : 0.0; dl += ns[l]; }
    const unsigned long long* mw = reinterpret_cast<const unsigned long long*>(dl);
    unsigned long long* mkw = reinterpret_cast<unsigned long long*>(MKbase);
    for (int e = tid; e < (mk_tot >> 3); e += blockDim.x) mkw[e] = mw[e];
#pragma unroll
    for (int l = 0; l < NL; ++l) wl[l] = *ta.L[l].w;
    if (in) {
      const double b = ta.b0[tid];
      B[0][tid] = b;
      X[0][tid] = dv[0] * b * wl[0];                               // pre-smoothing from a zero guess
    }
  }
  __syncthreads();
  // ---- downward leg
#pragma unroll
  for (int l = 0; l + 1 < NL; ++l) {
    const TailLevel F = ta.L[l];
    const TailLevel C = ta.L[l + 1];
    const int n = F.n, pxy = F.px * F.py;
    if (tid < n) {                                                // residual
      const double ax = (l == 0) ? tail_row<KIND>(a0, 1, X[l], tid, F.px, pxy, n)
                                 : tail_row<KIND>(M[l] + tid, n, X[l], tid, F.px, pxy, n);
      T[l][tid] = B[l][tid] - ax;
    }
    __syncthreads();
    if (tid < C.n) {                                              // restriction (order of k_restrict)
      const int I = tid % C.px, t = tid / C.px, J = t % C.py, K = t / C.py;
      double s = 0.0;
      if (MK[l + 1][tid] == 0) {
        // the taps of make_transfer_stencil(KIND), same order, as compile-time constants
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
          for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
              constexpr bool dim3 = (KIND == PPH_CELL_HEX || KIND == PPH_CELL_TET);
              if (!dim3 && dz != 0) continue;
              const double w = tail_tap_weight(KIND, dx, dy, dz);
              if (w == 0.0) continue;
              const int i = 2 * I + dx, j = 2 * J + dy, k = 2 * K + dz;
              const bool ok = i >= 0 && i < F.px && j >= 0 && j < F.py && k >= 0 && k < F.pz;
              const int f = ok ? i + F.px * (j + F.py * k) : 0;
              const double tv = T[l][f];
              const bool use = ok && (MK[l][f] & 1) == 0;
              if (use) s += w * tv;
            }
      }
      B[l + 1][tid] = s;
      if (l + 2 < NL) X[l + 1][tid] = dv[l + 1] * s * wl[l + 1];   // the coarse level's pre-smoothing rides along
    }
    __syncthreads();
  }
  // ---- coarsest level: Jacobi-preconditioned CG (the algorithm of k_coarse_cg), vectors in LDS
  {
    constexpr int lc = NL - 1;
    const TailLevel C = ta.L[lc];
    const int n = C.n, pxy = C.px * C.py;
    const bool in = tid < n;
    double* x = X[lc]; double* r = T[lc]; const double* b = B[lc];
    if (n <= 64) {
      // one wave: no workgroup barriers (LDS operations of a wave complete in order)
      if (tid < 64) {
        double ri = 0.0, zi = 0.0;
        if (in) { ri = b[tid]; zi = dv[lc] * ri; x[tid] = 0.0; r[tid] = ri; P[tid] = zi; }
        const double zz = tail_wave_sum(zi * zi);
        double rz = tail_wave_sum(ri * zi);
        const double tol = cg_rtol * sqrt(zz);
        bool go = sqrt(zz) > tol;
        double rr = ri, pi = zi;
        for (int it = 0; go && it < cg_max_it; ++it) {
          TAIL_WAVE_FENCE();
          double s = 0.0;
          if (in) s = (lc == 0) ? tail_row<KIND>(a0, 1, P, tid, C.px, pxy, n) : tail_row<KIND>(M[lc] + tid, n, P, tid, C.px, pxy, n);
          const double pq = tail_wave_sum(pi * s);
          if (!(pq > 0.0)) break;
          const double alpha = rz / pq;
          double r2 = 0.0, z2 = 0.0;
          if (in) {
            x[tid] += alpha * pi;
            r2 = rr - alpha * s;
            z2 = dv[lc] * r2;
            rr = r2;
          }
          const double zz2 = tail_wave_sum(z2 * z2);
          const double rz2 = tail_wave_sum(r2 * z2);
          if (sqrt(zz2) <= tol) break;
          const double beta = rz2 / rz;
          TAIL_WAVE_FENCE();   // every lane has read P of this iteration's product
          pi = z2 + beta * pi;
          if (in) P[tid] = pi;
          rz = rz2;
        }
      }
    } else {
      double ri = 0.0, zi = 0.0;
      if (in) { ri = b[tid]; zi = dv[lc] * ri; x[tid] = 0.0; r[tid] = ri; P[tid] = zi; }
      double zz = coarse_block_sum(zi * zi, red);
      double rz = coarse_block_sum(ri * zi, red);
      const double tol = cg_rtol * sqrt(zz);
      bool go = sqrt(zz) > tol;
      for (int it = 0; go && it < cg_max_it; ++it) {
        __syncthreads();
        double s = 0.0, pi = 0.0;
        if (in) {
          s = (lc == 0) ? tail_row<KIND>(a0, 1, P, tid, C.px, pxy, n) : tail_row<KIND>(M[lc] + tid, n, P, tid, C.px, pxy, n);
          pi = P[tid];
          Q[tid] = s;
        }
        const double pq = coarse_block_sum(pi * s, red);
        if (!(pq > 0.0)) break;
        const double alpha = rz / pq;
        double r2 = 0.0, z2 = 0.0;
        if (in) {
          x[tid] += alpha * pi;
          r2 = r[tid] - alpha * s;
          z2 = dv[lc] * r2;
          r[tid] = r2;
        }
        const double zz2 = coarse_block_sum(z2 * z2, red);
        const double rz2 = coarse_block_sum(r2 * z2, red);
        if (sqrt(zz2) <= tol) break;
        const double beta = rz2 / rz;
        __syncthreads();   // every thread has read P of this iteration's product
        if (in) P[tid] = z2 + beta * pi;
        rz = rz2;
      }
    }
  }
  __syncthreads();
  // ---- upward leg
#pragma unroll
  for (int l = NL - 2; l >= 0; --l) {
    const TailLevel F = ta.L[l];
    const TailLevel C = ta.L[l + 1];
    const int n = F.n, pxy = F.px * F.py;
    const bool in = tid < n;
    if (in) {                                                     // t = x + P x_c (k_prolong_to)
      const double x0 = X[l][tid];
      double s = 0.0;
      if (MK[l][tid] == 0) {
        const int i = tid % F.px, t = tid / F.px, j = t % F.py, kg = t / F.py;
        const int ox = i & 1, oy = j & 1, oz = kg & 1;
        const int sx = 1, sy = C.px, sz = C.px * C.py;
        const int c = (i >> 1) + sy * (j >> 1) + sz * (kg >> 1);
        const double* xc = X[l + 1];
        if (tk == 0) {
          const int ex = ox ? sx : 0, ey = oy ? sy : 0, ez = oz ? sz : 0;
          const double c000 = xc[c], c100 = xc[c + ex], c010 = xc[c + ey], c110 = xc[c + ey + ex];
          const double c001 = xc[c + ez], c101 = xc[c + ez + ex], c011 = xc[c + ez + ey], c111 = xc[c + ez + ey + ex];
          const double v00 = 0.5 * (c000 + c100), v10 = 0.5 * (c010 + c110), v01 = 0.5 * (c001 + c101),
                       v11 = 0.5 * (c011 + c111);
          if (oy && oz) s = 0.25 * (v00 + v10 + v01 + v11);
          else if (oy | oz) s = 0.5 * (v00 + (oy ? v10 : v01));
          else s = v00;
        } else {
          const int o = ox * sx + oy * sy + oz * sz;
          if (tk == 2 && ox && oy) s = 0.5 * (xc[c + sx] + xc[c + sy]);
          else s = (o == 0) ? xc[c] : 0.5 * (xc[c] + xc[c + o]);
        }
      }
      T[l][tid] = x0 + s;
    }
    __syncthreads();
    if (in) {                                                     // post-smoothing (la_spmv_jacobi)
      const double at = (l == 0) ? tail_row<KIND>(a0, 1, T[l], tid, F.px, pxy, n)
                                 : tail_row<KIND>(M[l] + tid, n, T[l], tid, F.px, pxy, n);
      const double xn = T[l][tid] + dv[l] * (B[l][tid] - at) * wl[l];
      if (l == 0) ta.x0[tid] = xn;
      else X[l][tid] = xn;
    }
    if (l > 0) __syncthreads();
  }
  if (NL == 1 && tid < ns[0]) ta.x0[tid] = X[0][tid];
}

// LDS bytes of the tail kernel for the given level sizes; 0 when the levels do not fit its limits
static size_t mg_tail_lds(const int* n, int nl, int S) {
  if (nl < 1 || nl > MG_TAIL_MAX || n[0] > MG_TAIL_ROWS) return 0;
  size_t d = 0, mat = 0, mk = 0;
  for (int l = 0; l < nl; ++l) { d += 3 * (size_t)n[l]; mk += ((size_t)n[l] + 7) & ~(size_t)7; if (l > 0) mat += (size_t)S * n[l]; }
  d += 2 * (size_t)n[nl - 1];
  if (mat > MG_TAIL_MATPOOL) return 0;
  return (d + mat) * sizeof(double) + mk;
}



// first level of the tail (nlev: no tail): from there on every level holds a stencil-ELL operator, is not
// distributed and the levels fit the kernel's limits (rows of the first one, LDS of the others)
static int mg_tail_begin(const pph_ctx* ctx, int which) {
  const std::vector<MgLevel>& mg = ctx->mg;
  const int nlev = (int)mg.size();
  if (!ctx->coarse_on_device) return nlev;
  const int64_t cap = ctx->mg_tail_rows < MG_TAIL_ROWS ? ctx->mg_tail_rows : MG_TAIL_ROWS;
  int lt = nlev;
  for (int l = nlev - 1; l >= 1; --l) {
    const MgLevel& L = mg[l];
    if (L.n > cap || !L.ell[which].val || (ctx->world > 1 && !L.replicated)) break;
    int n[MG_TAIL_MAX];
    const int nl = nlev - l;
    if (nl > MG_TAIL_MAX) break;
    for (int q = 0; q < nl; ++q) n[q] = (int)mg[l + q].n;
    if (mg_tail_lds(n, nl, sell_slots(ctx->mesh.kind)) == 0) break;
    lt = l;
  }
  return lt;
}

// packs the tail levels of block `which` for k_mg_tail (after every assembly of the level operators)
static int mg_tail_pack(pph_ctx* ctx, int which) {
  std::vector<MgLevel>& mg = ctx->mg;
  const int nlev = (int)mg.size();
  ctx->mg_tail_lt[which] = -1;
  if (!ctx->mg_fused || nlev < 2) return PPH_OK;
  for (const MgLevel& L : mg)
    if (!L.ell[which].val) return PPH_OK;
  const int lt = mg_tail_begin(ctx, which);
  if (lt >= nlev) return PPH_OK;
  const int kind = ctx->mesh.kind;
  const int S = sell_slots(kind);
  TailPackArgs pa;
  int ns[MG_TAIL_MAX];
  pa.nl = nlev - lt;
  for (int q = 0; q < pa.nl; ++q) {
    MgLevel& L = mg[lt + q];
    TailSrc& T = pa.L[q];
    T.A = L.ell[which].val; T.sym = L.ell[which].sym; T.ld = L.ell[which].ld; T.dinv = L.dinv[which].p; T.mask = L.maskp[which];
    T.px = L.px; T.py = L.py; T.pz = L.pz; T.n = (int)L.n;
    ns[q] = (int)L.n;
  }
  const size_t want = (size_t)tail_pack_doubles(ns, pa.nl, S) + (size_t)tail_pack_maskbytes(ns, pa.nl) / 8;
  if (ctx->mg_tail_pack[which].n != want) PPH_TRY(ctx->mg_tail_pack[which].alloc(ctx, want));
  double* pack = ctx->mg_tail_pack[which].p;
#define PPH_PACK_GO(KK) hipLaunchKernelGGL(k_tail_pack<KK>, dim3(16), dim3(256), 0, ctx->stream, pa, pack)
  switch (kind) {
    case PPH_CELL_QUAD: PPH_PACK_GO(PPH_CELL_QUAD); break;
    case PPH_CELL_TRI: PPH_PACK_GO(PPH_CELL_TRI); break;
    case PPH_CELL_HEX: PPH_PACK_GO(PPH_CELL_HEX); break;
    default: PPH_PACK_GO(PPH_CELL_TET); break;
  }
#undef PPH_PACK_GO
  ctx->mg_tail_lt[which] = lt;
  return PPH_OK;
}

static bool mg_can_fuse(const pph_ctx* ctx, int which, int nsmooth) {
  if (!ctx->mg_fused || nsmooth != 1 || ctx->mg.size() < 2) return false;
  for (const MgLevel& L : ctx->mg)
    if (!L.ell[which].val) return false;
  return true;
}

// V(1,1) on stencil-ELL levels: per level two SpMV launches (residual; post-smoothing sweep with the Jacobi update in
// its epilogue) and two transfer launches (restriction fused with the coarse level's pre-smoothing; interpolation),
// then the tail kernel.  Same operations as the general cycle below.
static void mg_vcycle_fused(pph_ctx* ctx, int which, const double* rin, double* zout) {
  std::vector<MgLevel>& mg = ctx->mg;
  const int nlev = (int)mg.size();
  const bool dist = ctx->world > 1;
  const int kind = ctx->mesh.kind;
  const TStencil st = make_transfer_stencil(kind);
  const int lt = mg_tail_begin(ctx, which);
  const int top = (lt < nlev) ? lt : nlev - 1;   // levels [0, top) are swept by full-chip kernels
  // requests of the calling Krylov loop (see pph_internal.h): zout already pre-smoothed; rin . zout wanted
  const bool x0_ready = ctx->mg_x0_ready;
  const int dot_slot = ctx->mg_dot_slot;
  ctx->mg_x0_ready = false;
  for (int l = 0; l < top; ++l) {
    MgLevel& L = mg[l];
    MgLevel& C = mg[l + 1];
    const double* b = (l == 0) ? rin : L.b.p;
    double* x = (l == 0) ? zout : L.x.p;
    ctx->comm_suspended = L.replicated;
    if (l == 0 && !x0_ready)   // coarser levels: done by the restriction that produced their right-hand side
      hipLaunchKernelGGL(k_cheb_init, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, x, L.d.p, b, L.dinv[which].p, 0.0,
                         1, 0, L.n, cheb_wp(ctx, l, which));
    la_spmv_resid(ctx, level_csr(ctx, L, which), x, b, L.r.p);
    if (dist && !L.replicated) (void)la_halo(ctx, *L.geom, L.r.p);
    // the coarse level's pre-smoothing rides along unless the tail kernel (or the coarsest solve) does it itself
    const bool init_c = (l + 1 < top);
    double* xc = init_c ? C.x.p : nullptr;
    const double* dc = init_c ? C.dinv[which].p : nullptr;
    const double* wc = init_c ? cheb_wp(ctx, l + 1, which) : nullptr;
    const bool sum_c = dist && C.replicated && !L.replicated;   // partial right-hand sides are summed first
    if (sum_c) { xc = nullptr; }
    if (kind == PPH_CELL_HEX)
      hipLaunchKernelGGL(k_restrict_q1<3>, dim3(mg_grid(C.n)), dim3(256), 0, ctx->stream, C.b.p, L.r.p, C.maskp[which],
                         L.maskp[which], tgeom(L, C), xc, dc, wc, C.rfast[which].p);
    else if (kind == PPH_CELL_QUAD)
      hipLaunchKernelGGL(k_restrict_q1<2>, dim3(mg_grid(C.n)), dim3(256), 0, ctx->stream, C.b.p, L.r.p, C.maskp[which],
                         L.maskp[which], tgeom(L, C), xc, dc, wc, C.rfast[which].p);
    else
      hipLaunchKernelGGL(k_restrict, dim3(mg_grid(C.n)), dim3(256), 0, ctx->stream, C.b.p, L.r.p, st, C.maskp[which],
                         L.maskp[which], tgeom(L, C), xc, dc, wc);
    if (sum_c) {
      (void)la_allreduce_vec(ctx, C.b.p, C.n);
      if (init_c)
        hipLaunchKernelGGL(k_cheb_init, dim3(mg_grid(C.n)), dim3(256), 0, ctx->stream, C.x.p, C.d.p, C.b.p,
                           C.dinv[which].p, 0.0, 1, 0, C.n, cheb_wp(ctx, l + 1, which));
    }
  }
  if (lt < nlev) {
    if (ctx->mg_tail_lt[which] != lt) (void)mg_tail_pack(ctx, which);   // (an option moved the tail after the set-up)
    TailArgs ta;
    int ns[MG_TAIL_MAX];
    const int nl = nlev - lt;
    for (int q = 0; q < MG_TAIL_MAX; ++q) {
      TailLevel& T = ta.L[q];
      if (q < nl) {
        MgLevel& L = mg[lt + q];
        T.px = L.px; T.py = L.py; T.pz = L.pz; T.n = (int)L.n;
        T.w = cheb_wp(ctx, lt + q, which);
        ns[q] = (int)L.n;
      } else {
        T.px = T.py = T.pz = 1; T.n = 0; T.w = nullptr; ns[q] = 0;
      }
    }
    ta.pack = ctx->mg_tail_pack[which].p;
    ta.b0 = mg[lt].b.p;
    ta.x0 = mg[lt].x.p;
    const int tk = (kind == PPH_CELL_QUAD || kind == PPH_CELL_HEX) ? 0 : (kind == PPH_CELL_TET ? 1 : 2);
    int threads = (int)((mg[lt].n + 63) / 64) * 64;
    if (threads < 64) threads = 64;
    const size_t lds = mg_tail_lds(ns, nl, sell_slots(kind));
#define PPH_TAIL_GO(KK, NN) \
  hipLaunchKernelGGL((k_mg_tail<KK, NN>), dim3(1), dim3(threads), lds, ctx->stream, ta, st, tk, 1e-12, ctx->coarse_max_it)
#define PPH_TAIL_NL(KK)                                                                       \
  switch (nl) {                                                                               \
    case 1: PPH_TAIL_GO(KK, 1); break;                                                        \
    case 2: PPH_TAIL_GO(KK, 2); break;                                                        \
    case 3: PPH_TAIL_GO(KK, 3); break;                                                        \
    default: PPH_TAIL_GO(KK, 4); break;                                                       \
  }
    switch (kind) {
      case PPH_CELL_QUAD: PPH_TAIL_NL(PPH_CELL_QUAD); break;
      case PPH_CELL_TRI: PPH_TAIL_NL(PPH_CELL_TRI); break;
      case PPH_CELL_HEX: PPH_TAIL_NL(PPH_CELL_HEX); break;
      default: PPH_TAIL_NL(PPH_CELL_TET); break;
    }
#undef PPH_TAIL_NL
#undef PPH_TAIL_GO
  } else {
    // no tail (coarsest level too large or distributed): the host-driven / single-level solve of the general cycle
    MgLevel& C = mg[nlev - 1];
    int its = 0;
    ctx->comm_suspended = C.replicated;
    if ((!dist || C.replicated) && C.n <= 4096 && ctx->coarse_on_device)
      hipLaunchKernelGGL(k_coarse_cg_sell, dim3(1), dim3(C.n <= 256 ? 256 : 1024), 0, ctx->stream, C.ell[which].val,
                         C.ell[which].ld, C.ell[which].sym, make_stencil(kind), C.px, C.py, C.pz, C.dinv[which].p, C.b.p, C.x.p, C.r.p,
                         C.d.p, C.t.p, (int)C.n, 1e-12, ctx->coarse_max_it);
    else
      pph_cg_jacobi(ctx, level_csr(ctx, C, which), C.b.p, C.x.p, C.dinv[which].p, 1e-12, 0.0, ctx->coarse_max_it, C.r.p, C.d.p, C.t.p,
                    C.w.p, &its);
  }
  for (int l = top - 1; l >= 0; --l) {
    MgLevel& L = mg[l];
    MgLevel& C = mg[l + 1];
    const double* b = (l == 0) ? rin : L.b.p;
    double* x = (l == 0) ? zout : L.x.p;
    ctx->comm_suspended = L.replicated;
    if (dist && !C.replicated) (void)la_halo(ctx, *C.geom, C.x.p);
    const TGeom tg = tgeom(L, C);
    if (kind == PPH_CELL_QUAD || kind == PPH_CELL_HEX)
      hipLaunchKernelGGL(k_prolong_to_q1, dim3(mg_grid((L.n + 1) / 2 + L.py * L.pz)), dim3(256), 0, ctx->stream, L.t.p, x, C.x.p, L.maskp[which], tg);
    else if (kind == PPH_CELL_TET)
      hipLaunchKernelGGL(k_prolong_to<1>, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, L.t.p, x, C.x.p, L.maskp[which], tg);
    else
      hipLaunchKernelGGL(k_prolong_to<2>, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, L.t.p, x, C.x.p, L.maskp[which], tg);
    // ghost planes of L.t: x's were refreshed by the residual product of the downward leg, the coarse correction's
    // just above, and k_prolong_to interpolates ghost rows as well - one exchange per level and cycle less
    if (l == 0 && dot_slot >= 0) {
      la_spmv_jacobi(ctx, level_csr(ctx, L, which), L.t.p, b, L.dinv[which].p, cheb_wp(ctx, l, which), x, dot_slot,
                     ctx->mg_dot_seg.off1, ctx->mg_dot_seg.off1 + ctx->mg_dot_seg.len1, true);
      ctx->mg_dot_slot = -2;   // delivered
    } else {
      la_spmv_jacobi(ctx, level_csr(ctx, L, which), L.t.p, b, L.dinv[which].p, cheb_wp(ctx, l, which), x, -1, 0, 0, true);
    }
  }
  ctx->comm_suspended = false;
}

// the pre-smoothing the fused cycle starts with on the fine level: z0 = dinv .* r * w (false: the cycle is not the
// fused one, nothing to offer)
bool mg_pre_smoother(pph_ctx* ctx, int which, int nsmooth, const double** dinv, const double** w, bool* launch_only) {
  if (!ctx->mg_ok || !mg_can_fuse(ctx, which, nsmooth)) return false;
  *dinv = ctx->mg[0].dinv[which].p;
  *w = cheb_wp(ctx, 0, which);
  // the cycle consists of kernel launches only (capturable into a graph) unless its coarsest solve is host-driven
  const MgLevel& C = ctx->mg.back();
  const int nlev = (int)ctx->mg.size();
  *launch_only = mg_tail_begin(ctx, which) < nlev ||
                 ((ctx->world == 1 || C.replicated) && C.n <= 4096 && ctx->coarse_on_device);
  return true;
}

void mg_vcycle(pph_ctx* ctx, int which, const double* rin, double* zout, int nsmooth) {
  if (mg_can_fuse(ctx, which, nsmooth)) { mg_vcycle_fused(ctx, which, rin, zout); return; }
  ctx->mg_x0_ready = false;   // the general cycle offers neither by-product (mg_dot_slot stays undelivered)
  std::vector<MgLevel>& mg = ctx->mg;
  const int nlev = (int)mg.size();
  const TStencil st = make_transfer_stencil(ctx->mesh.kind);
  const bool dist = ctx->world > 1;
  if (nlev == 1) {
    // mesh cannot be coarsened: polynomial (Chebyshev) preconditioner only
    chebyshev(ctx, mg[0], which, rin, zout, nsmooth > 2 ? nsmooth : 2, true);
    return;
  }
  // downward leg
  for (int l = 0; l < nlev - 1; ++l) {
    MgLevel& L = mg[l];
    MgLevel& C = mg[l + 1];
    const double* b = (l == 0) ? rin : L.b.p;
    double* x = (l == 0) ? zout : L.x.p;
    ctx->comm_suspended = L.replicated;
    chebyshev(ctx, L, which, b, x, nsmooth, true);
    la_spmv_resid(ctx, level_csr(ctx, L, which, true), x, b, L.r.p);
    if (dist && !L.replicated) (void)la_halo(ctx, *L.geom, L.r.p);  // restriction reads one fine plane beyond the owned ones
    if (ctx->mesh.kind == PPH_CELL_HEX)
      hipLaunchKernelGGL(k_restrict_q1<3>, dim3(mg_grid(C.n)), dim3(256), 0, ctx->stream, C.b.p, L.r.p, C.maskp[which],
                         L.maskp[which], tgeom(L, C), (double*)nullptr, (const double*)nullptr, (const double*)nullptr,
                         C.rfast[which].p);
    else if (ctx->mesh.kind == PPH_CELL_QUAD)
      hipLaunchKernelGGL(k_restrict_q1<2>, dim3(mg_grid(C.n)), dim3(256), 0, ctx->stream, C.b.p, L.r.p, C.maskp[which],
                         L.maskp[which], tgeom(L, C), (double*)nullptr, (const double*)nullptr, (const double*)nullptr,
                         C.rfast[which].p);
    else
      hipLaunchKernelGGL(k_restrict, dim3(mg_grid(C.n)), dim3(256), 0, ctx->stream, C.b.p, L.r.p, st, C.maskp[which],
                         L.maskp[which], tgeom(L, C), (double*)nullptr, (const double*)nullptr, (const double*)nullptr);
    if (dist && C.replicated && !L.replicated) (void)la_allreduce_vec(ctx, C.b.p, C.n);
  }
  // coarsest level: Jacobi-CG to 1e-12 (a handful of unknowns)
  {
    MgLevel& C = mg[nlev - 1];
    int its = 0;
    ctx->comm_suspended = C.replicated;
    if ((!dist || C.replicated) && C.n <= 4096 && ctx->coarse_on_device && C.ell[which].val)
      hipLaunchKernelGGL(k_coarse_cg_sell, dim3(1), dim3(C.n <= 256 ? 256 : 1024), 0, ctx->stream, C.ell[which].val,
                         C.ell[which].ld, C.ell[which].sym, make_stencil(ctx->mesh.kind), C.px, C.py, C.pz, C.dinv[which].p, C.b.p, C.x.p,
                         C.r.p, C.d.p, C.t.p, (int)C.n, 1e-12, ctx->coarse_max_it);
    else if ((!dist || C.replicated) && C.n <= 4096 && ctx->coarse_on_device)
      hipLaunchKernelGGL(k_coarse_cg, dim3(1), dim3(C.n <= 256 ? 256 : 1024), 0, ctx->stream, C.rowptr, C.col, C.val[which],
                         C.dinv[which].p, C.b.p, C.x.p, C.r.p, C.d.p, C.t.p, (int)C.n, 1e-12, ctx->coarse_max_it);
    else
      pph_cg_jacobi(ctx, level_csr(ctx, C, which), C.b.p, C.x.p, C.dinv[which].p, 1e-12, 0.0, ctx->coarse_max_it, C.r.p, C.d.p, C.t.p,
                    C.w.p, &its);
  }
  // upward leg
  for (int l = nlev - 2; l >= 0; --l) {
    MgLevel& L = mg[l];
    MgLevel& C = mg[l + 1];
    const double* b = (l == 0) ? rin : L.b.p;
    double* x = (l == 0) ? zout : L.x.p;
    ctx->comm_suspended = L.replicated;
    if (dist && !C.replicated) (void)la_halo(ctx, *C.geom, C.x.p);  // interpolation reads the coarse ghost plane
    const TGeom tg = tgeom(L, C);
    const int kind = ctx->mesh.kind;
    if (kind == PPH_CELL_QUAD || kind == PPH_CELL_HEX)
      hipLaunchKernelGGL(k_prolong_add<0>, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, x, C.x.p, L.maskp[which], tg);
    else if (kind == PPH_CELL_TET)
      hipLaunchKernelGGL(k_prolong_add<1>, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, x, C.x.p, L.maskp[which], tg);
    else
      hipLaunchKernelGGL(k_prolong_add<2>, dim3(mg_grid(L.n)), dim3(256), 0, ctx->stream, x, C.x.p, L.maskp[which], tg);
    chebyshev(ctx, L, which, b, x, nsmooth, false);
  }
  ctx->comm_suspended = false;
}
