// Error norms of a CG-1 field against the manufactured DPP pressures, on the device.
//
// Replaces l2_error / h1_seminorm_error (reference src/perphil/utils/postprocessing.py:89-124, which
// assemble ||p_h - p||^2 and |p_h - p|_1^2 with Firedrake) for the exact solutions of
// src/perphil/utils/manufactured_solutions.py:39-51 (2D) and :87-88 (3D) — SURVEY.md §8f rank 2.
// One thread per cell, nq-point Gauss rule per direction on the isoparametric map (multilinear cells) or collapsed
// onto the simplex (P1 cells); the exact pressure and its gradient are evaluated in closed form at every
// quadrature point.
#include "pph_internal.h"
#include <cmath>

struct GaussRule {
  int nq;
  double x[8], w[8];
};

struct MmsPar {
  double mu_over_pi, coef_e, eta;  // p = (mu/pi) e^{pi x} S(y,z) + coef_e E(y,z);  coef_e = -mu/(beta k1) or +mu/(beta k2)
  double mu;
};

// SRC 0: the manufactured pressure in closed form (MmsPar).  SRC 1: the exact field sampled by the caller at the
// quadrature points of the cells [c0, c1) - se[(cell - c0) npts + q] and, when sg != null, its gradient
// sg[((cell - c0) npts + q) DIM + d]; se == null: the exact field is zero (norms of the finite-element function itself).
// SRC 2: no sums - the physical coordinates of those quadrature points to xout[((cell - c0) npts + q) DIM + d].
struct ErrSamples { const double* se; const double* sg; double* xout; int64_t c0, c1; };

template <int DIM, int SRC = 0>
__global__ __launch_bounds__(256) void k_error_norms(const int32_t* __restrict__ cells, const double* __restrict__ cx,
                                                     const double* __restrict__ cy, const double* __restrict__ cz,
                                                     const double* __restrict__ u, GaussRule g, MmsPar p,
                                                     int64_t ncell, double* __restrict__ part, ErrSamples es = ErrSamples()) {
  constexpr int NB = 1 << DIM;
  __shared__ double lds[4];
  const double PI = 3.14159265358979323846;
  double l2 = 0.0, h1 = 0.0;
  const int64_t cbeg = SRC ? es.c0 : 0, cend = SRC ? es.c1 : ncell;
  for (int64_t cell = cbeg + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; cell < cend;
       cell += (int64_t)gridDim.x * blockDim.x) {
    double X[NB][DIM], U[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int32_t nd = cells[cell * NB + b];
      X[b][0] = cx[nd];
      X[b][1] = cy[nd];
      if constexpr (DIM == 3) X[b][2] = cz[nd];
      U[b] = u[nd];
    }
    const int nq = g.nq;
    const int npts = (DIM == 2) ? nq * nq : nq * nq * nq;
    for (int q = 0; q < npts; ++q) {
      int qi[3] = {q % nq, (q / nq) % nq, q / (nq * nq)};
      double xi[DIM], w = 1.0;
#pragma unroll
      for (int e = 0; e < DIM; ++e) { xi[e] = g.x[qi[e]]; w *= g.w[qi[e]]; }
      double N[NB], dN[NB][DIM];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        double nv = 1.0;
#pragma unroll
        for (int e = 0; e < DIM; ++e) {
          const double s = ((b >> e) & 1) ? 1.0 : -1.0;
          nv *= 0.5 * (1.0 + s * xi[e]);
        }
        N[b] = nv;
#pragma unroll
        for (int e = 0; e < DIM; ++e) {
          double d = 1.0;
#pragma unroll
          for (int f = 0; f < DIM; ++f) {
            const double s = ((b >> f) & 1) ? 1.0 : -1.0;
            d *= (f == e) ? 0.5 * s : 0.5 * (1.0 + s * xi[f]);
          }
          dN[b][e] = d;
        }
      }
      double J[DIM][DIM], xq[DIM], uh = 0.0, gu_ref[DIM];
#pragma unroll
      for (int e = 0; e < DIM; ++e) {
        gu_ref[e] = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) J[e][d] = 0.0;
      }
#pragma unroll
      for (int d = 0; d < DIM; ++d) xq[d] = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        uh += N[b] * U[b];
#pragma unroll
        for (int d = 0; d < DIM; ++d) xq[d] += N[b] * X[b][d];
#pragma unroll
        for (int e = 0; e < DIM; ++e) {
          gu_ref[e] += dN[b][e] * U[b];
#pragma unroll
          for (int d = 0; d < DIM; ++d) J[e][d] += dN[b][e] * X[b][d];
        }
      }
      double det, I[DIM][DIM];
      if constexpr (DIM == 2) {
        det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        const double r = 1.0 / det;
        I[0][0] = J[1][1] * r;  I[0][1] = -J[0][1] * r;
        I[1][0] = -J[1][0] * r; I[1][1] = J[0][0] * r;
      } else {
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
        const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
        const double r = 1.0 / det;
        I[0][0] = c00 * r;
        I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
        I[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
        I[1][0] = c01 * r;
        I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
        I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
        I[2][0] = c02 * r;
        I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
        I[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
      }
      // exact pressure and gradient
      double pe = 0.0, ge[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) ge[d] = 0.0;
      if constexpr (SRC == 2) {
        const int64_t o = ((cell - es.c0) * npts + q) * DIM;
#pragma unroll
        for (int d = 0; d < DIM; ++d) es.xout[o + d] = xq[d];
        continue;
      } else if constexpr (SRC == 1) {
        const int64_t o = (cell - es.c0) * npts + q;
        if (es.se) pe = es.se[o];
        if (es.sg) {
#pragma unroll
          for (int d = 0; d < DIM; ++d) ge[d] = es.sg[o * DIM + d];
        }
      } else {
        const double ex = exp(PI * xq[0]);
        double S = sin(PI * xq[1]), E = exp(p.eta * xq[1]);
        if constexpr (DIM == 3) {
          const double Sz = sin(PI * xq[2]), Ez = exp(p.eta * xq[2]);
          pe = p.mu_over_pi * ex * (S + Sz) + p.coef_e * (E + Ez);
          ge[0] = p.mu * ex * (S + Sz);
          ge[1] = p.mu * ex * cos(PI * xq[1]) + p.coef_e * p.eta * E;
          ge[2] = p.mu * ex * cos(PI * xq[2]) + p.coef_e * p.eta * Ez;
        } else {
          pe = p.mu_over_pi * ex * S + p.coef_e * E;
          ge[0] = p.mu * ex * S;
          ge[1] = p.mu * ex * cos(PI * xq[1]) + p.coef_e * p.eta * E;
        }
      }
      const double wd = w * fabs(det);
      const double du = uh - pe;
      l2 += wd * du * du;
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        double gh = 0.0;
#pragma unroll
        for (int e = 0; e < DIM; ++e) gh += I[d][e] * gu_ref[e];
        const double dg = gh - ge[d];
        h1 += wd * dg * dg;
      }
    }
  }
  // block sums -> partials [0][block], [1][block]
  auto bsum = [&](double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[wv] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += lds[q];
    return t;
  };
  if constexpr (SRC == 2) return;   // (coordinates only: `part` is null)
  const double a = bsum(l2);
  const double b = bsum(h1);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = a;
    part[2048 + blockIdx.x] = b;
  }
}

// P1 simplices (left-diagonal triangles, Kuhn tetrahedra): constant gradient of p_h per cell; quadrature by the
// collapsed (Duffy) tensor Gauss rule: lambda_1 = u, lambda_2 = v (1 - u), lambda_3 = w (1 - u)(1 - v) on [0,1]^d.
template <int DIM, int SRC = 0>
__global__ __launch_bounds__(256) void k_error_norms_simplex(const int32_t* __restrict__ cells, const double* __restrict__ cx,
                                                             const double* __restrict__ cy, const double* __restrict__ cz,
                                                             const double* __restrict__ u, GaussRule g, MmsPar p,
                                                             int64_t ncell, double* __restrict__ part, ErrSamples es = ErrSamples()) {
  constexpr int NB = DIM + 1;
  __shared__ double lds[4];
  const double PI = 3.14159265358979323846;
  double l2 = 0.0, h1 = 0.0;
  const int64_t cbeg = SRC ? es.c0 : 0, cend = SRC ? es.c1 : ncell;
  for (int64_t cell = cbeg + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; cell < cend;
       cell += (int64_t)gridDim.x * blockDim.x) {
    double X[NB][DIM], U[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int32_t nd = cells[cell * NB + b];
      X[b][0] = cx[nd];
      X[b][1] = cy[nd];
      if constexpr (DIM == 3) X[b][2] = cz[nd];
      U[b] = u[nd];
    }
    double E[DIM][DIM], dU[DIM], gh[DIM], det;
#pragma unroll
    for (int r = 0; r < DIM; ++r) {
      dU[r] = U[r + 1] - U[0];
#pragma unroll
      for (int d = 0; d < DIM; ++d) E[r][d] = X[r + 1][d] - X[0][d];
    }
    if constexpr (DIM == 2) {
      det = E[0][0] * E[1][1] - E[0][1] * E[1][0];
      gh[0] = (E[1][1] * dU[0] - E[0][1] * dU[1]) / det;
      gh[1] = (-E[1][0] * dU[0] + E[0][0] * dU[1]) / det;
    } else {
      const double c00 = E[1][1] * E[2][2] - E[1][2] * E[2][1];
      const double c01 = E[1][2] * E[2][0] - E[1][0] * E[2][2];
      const double c02 = E[1][0] * E[2][1] - E[1][1] * E[2][0];
      det = E[0][0] * c00 + E[0][1] * c01 + E[0][2] * c02;
      const double r = 1.0 / det;
      gh[0] = (c00 * dU[0] + (E[0][2] * E[2][1] - E[0][1] * E[2][2]) * dU[1] + (E[0][1] * E[1][2] - E[0][2] * E[1][1]) * dU[2]) * r;
      gh[1] = (c01 * dU[0] + (E[0][0] * E[2][2] - E[0][2] * E[2][0]) * dU[1] + (E[0][2] * E[1][0] - E[0][0] * E[1][2]) * dU[2]) * r;
      gh[2] = (c02 * dU[0] + (E[0][1] * E[2][0] - E[0][0] * E[2][1]) * dU[1] + (E[0][0] * E[1][1] - E[0][1] * E[1][0]) * dU[2]) * r;
    }
    const int nq = g.nq;
    const int npts = (DIM == 2) ? nq * nq : nq * nq * nq;
    for (int q = 0; q < npts; ++q) {
      const int qi[3] = {q % nq, (q / nq) % nq, q / (nq * nq)};
      const double uu = 0.5 * (g.x[qi[0]] + 1.0), vv = 0.5 * (g.x[qi[1]] + 1.0);
      double lam[DIM], w = 0.25 * g.w[qi[0]] * g.w[qi[1]] * (1.0 - uu);
      lam[0] = uu;
      lam[1] = vv * (1.0 - uu);
      if constexpr (DIM == 3) {
        const double ww = 0.5 * (g.x[qi[2]] + 1.0);
        lam[2] = ww * (1.0 - uu) * (1.0 - vv);
        w *= 0.5 * g.w[qi[2]] * (1.0 - uu) * (1.0 - vv);
      }
      double xq[DIM], uh = U[0];
#pragma unroll
      for (int d = 0; d < DIM; ++d) xq[d] = X[0][d];
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
        uh += lam[r] * dU[r];
#pragma unroll
        for (int d = 0; d < DIM; ++d) xq[d] += lam[r] * E[r][d];
      }
      double pe = 0.0, ge[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) ge[d] = 0.0;
      if constexpr (SRC == 2) {
        const int64_t o = ((cell - es.c0) * npts + q) * DIM;
#pragma unroll
        for (int d = 0; d < DIM; ++d) es.xout[o + d] = xq[d];
        continue;
      } else if constexpr (SRC == 1) {
        const int64_t o = (cell - es.c0) * npts + q;
        if (es.se) pe = es.se[o];
        if (es.sg) {
#pragma unroll
          for (int d = 0; d < DIM; ++d) ge[d] = es.sg[o * DIM + d];
        }
      } else {
        const double ex = exp(PI * xq[0]);
        const double S = sin(PI * xq[1]), Ey = exp(p.eta * xq[1]);
        if constexpr (DIM == 3) {
          const double Sz = sin(PI * xq[2]), Ez = exp(p.eta * xq[2]);
          pe = p.mu_over_pi * ex * (S + Sz) + p.coef_e * (Ey + Ez);
          ge[0] = p.mu * ex * (S + Sz);
          ge[1] = p.mu * ex * cos(PI * xq[1]) + p.coef_e * p.eta * Ey;
          ge[2] = p.mu * ex * cos(PI * xq[2]) + p.coef_e * p.eta * Ez;
        } else {
          pe = p.mu_over_pi * ex * S + p.coef_e * Ey;
          ge[0] = p.mu * ex * S;
          ge[1] = p.mu * ex * cos(PI * xq[1]) + p.coef_e * p.eta * Ey;
        }
      }
      const double wd = w * fabs(det);
      const double du = uh - pe;
      l2 += wd * du * du;
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        const double dg = gh[d] - ge[d];
        h1 += wd * dg * dg;
      }
    }
  }
  auto bsum = [&](double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[wv] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += lds[q];
    return t;
  };
  if constexpr (SRC == 2) return;   // (coordinates only: `part` is null)
  const double a = bsum(l2);
  const double b = bsum(h1);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = a;
    part[2048 + blockIdx.x] = b;
  }
}

__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
  __shared__ double lds[256];
  const double* p = part + (int64_t)blockIdx.x * 2048;
  double v = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) v += p[i];
  lds[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
}

static void gauss_legendre(int n, double* x, double* w) {
  // Newton iteration on P_n; n <= 8
  for (int i = 0; i < n; ++i) {
    double z = std::cos(3.14159265358979323846 * (i + 0.75) / (n + 0.5));
    double pp = 0.0;
    for (int it = 0; it < 100; ++it) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 0; j < n; ++j) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      const double z1 = z;
      z = z1 - p1 / pp;
      if (std::fabs(z - z1) < 1e-15) break;
    }
    x[i] = -z;
    w[i] = 2.0 / ((1.0 - z * z) * pp * pp);
  }
}

extern "C" int pph_error_norms_mms(pph_ctx* ctx, int field, const double* nodal_host, double k1, double k2,
                                   double beta, double mu, int nq, double* l2_out, double* h1s_out) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, ctx->mesh_ok, "pph_error_norms_mms before pph_mesh_build");
  PPH_REQUIRE(ctx, ctx->world == 1, "error norms are implemented for single-context meshes");
  const MeshData& m = ctx->mesh;
  PPH_REQUIRE(ctx, field == 0 || field == 1, "field must be 0 or 1");
  PPH_REQUIRE(ctx, nq >= 1 && nq <= 8 && nodal_host && l2_out && h1s_out, "bad arguments");
  PPH_REQUIRE(ctx, k1 > 0 && k2 > 0 && mu > 0 && beta > 0, "need positive parameters");
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  DevBuf<double> u, part;
  PPH_TRY(u.alloc(ctx, (size_t)m.n));
  PPH_TRY(part.alloc(ctx, 2 * 2048 + 2));
  PPH_HIP(ctx, hipMemcpyAsync(u.p, nodal_host, sizeof(double) * (size_t)m.n, hipMemcpyHostToDevice, ctx->stream));
  GaussRule g;
  g.nq = nq;
  gauss_legendre(nq, g.x, g.w);
  MmsPar p;
  p.mu = mu;
  p.mu_over_pi = mu / 3.14159265358979323846;
  p.eta = std::sqrt(beta * (k1 + k2) / (k1 * k2));
  p.coef_e = (field == 0) ? -mu / (beta * k1) : mu / (beta * k2);
  int64_t nb = ceil_div64(m.ncell, 256);
  const int grid = (int)(nb < 2048 ? nb : 2048);
  if (m.kind == PPH_CELL_QUAD)
    hipLaunchKernelGGL(k_error_norms<2>, dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p, u.p, g,
                       p, m.ncell, part.p);
  else if (m.kind == PPH_CELL_TRI)
    hipLaunchKernelGGL(k_error_norms_simplex<2>, dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p,
                       u.p, g, p, m.ncell, part.p);
  else if (m.kind == PPH_CELL_TET)
    hipLaunchKernelGGL(k_error_norms_simplex<3>, dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p,
                       u.p, g, p, m.ncell, part.p);
  else
    hipLaunchKernelGGL(k_error_norms<3>, dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p, u.p, g,
                       p, m.ncell, part.p);
  hipLaunchKernelGGL(k_sum_partials, dim3(2), dim3(256), 0, ctx->stream, part.p, grid, part.p + 4096);
  double r[2];
  PPH_HIP(ctx, hipMemcpyAsync(r, part.p + 4096, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PPH_HIP(ctx, hipGetLastError());
  *l2_out = std::sqrt(r[0]);
  *h1s_out = std::sqrt(r[1]);
  u.release();
  part.release();
  return PPH_OK;
}

// launches the kernel of the mesh's cell kind in source mode SRC
template <int SRC>
static void err_launch(pph_ctx* ctx, const MeshData& m, const double* u, const GaussRule& g, double* part, ErrSamples es, int grid) {
  MmsPar p = MmsPar();
  if (m.kind == PPH_CELL_QUAD)
    hipLaunchKernelGGL((k_error_norms<2, SRC>), dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p, u, g, p, m.ncell, part, es);
  else if (m.kind == PPH_CELL_TRI)
    hipLaunchKernelGGL((k_error_norms_simplex<2, SRC>), dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p, u, g, p, m.ncell, part, es);
  else if (m.kind == PPH_CELL_TET)
    hipLaunchKernelGGL((k_error_norms_simplex<3, SRC>), dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p, u, g, p, m.ncell, part, es);
  else
    hipLaunchKernelGGL((k_error_norms<3, SRC>), dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p, u, g, p, m.ncell, part, es);
}

extern "C" int pph_quadrature_points(pph_ctx* ctx, int nq, int64_t cell_begin, int64_t cell_count, double* xq_host) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, ctx->mesh_ok, "pph_quadrature_points before pph_mesh_build");
  const MeshData& m = ctx->mesh;
  PPH_REQUIRE(ctx, nq >= 1 && nq <= 8 && xq_host, "bad arguments");
  PPH_REQUIRE(ctx, cell_begin >= 0 && cell_count >= 1 && cell_begin + cell_count <= m.ncell, "cell range outside the mesh");
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t npts = (m.dim == 2) ? nq * nq : nq * nq * nq;
  DevBuf<double> xo;
  PPH_TRY(xo.alloc(ctx, (size_t)(cell_count * npts * m.dim)));
  GaussRule g;
  g.nq = nq;
  gauss_legendre(nq, g.x, g.w);
  ErrSamples es{nullptr, nullptr, xo.p, cell_begin, cell_begin + cell_count};
  const int64_t nb = ceil_div64(cell_count, 256);
  err_launch<2>(ctx, m, m.cx.p /* unused nodal field: any valid array of n values */, g, nullptr, es, (int)(nb < 2048 ? nb : 2048));
  PPH_HIP(ctx, hipMemcpyAsync(xq_host, xo.p, sizeof(double) * (size_t)(cell_count * npts * m.dim), hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PPH_HIP(ctx, hipGetLastError());
  xo.release();
  return PPH_OK;
}

extern "C" int pph_error_norms_sampled(pph_ctx* ctx, const double* nodal_host, int nq, int64_t cell_begin, int64_t cell_count,
                                       const double* exact_q_host, const double* grad_q_host, double* l2sq_out, double* h1sq_out) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, ctx->mesh_ok, "pph_error_norms_sampled before pph_mesh_build");
  PPH_REQUIRE(ctx, ctx->world == 1, "error norms are implemented for single-context meshes");
  const MeshData& m = ctx->mesh;
  PPH_REQUIRE(ctx, nq >= 1 && nq <= 8 && l2sq_out && h1sq_out, "bad arguments");
  PPH_REQUIRE(ctx, cell_begin >= 0 && cell_count >= 1 && cell_begin + cell_count <= m.ncell, "cell range outside the mesh");
  // nodal_host == NULL: the field uploaded by the previous call on this context (a caller that walks the cells in chunks
  // ships the nodal vector once, not once per chunk)
  PPH_REQUIRE(ctx, nodal_host || (ctx->post_u.p && ctx->post_u.n == (size_t)m.n && ctx->post_u_valid),
              "pph_error_norms_sampled: no nodal field (NULL means: the one of the previous call)");
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t npts = (m.dim == 2) ? nq * nq : nq * nq * nq;
  DevBuf<double> part, se, sg;
  DevBuf<double>& u = ctx->post_u;
  PPH_TRY(part.alloc(ctx, 2 * 2048 + 2));
  if (nodal_host) {
    ctx->post_u_valid = false;
    PPH_TRY(u.alloc(ctx, (size_t)m.n));
    PPH_HIP(ctx, hipMemcpyAsync(u.p, nodal_host, sizeof(double) * (size_t)m.n, hipMemcpyHostToDevice, ctx->stream));
    ctx->post_u_valid = true;
  }
  if (exact_q_host) {
    PPH_TRY(se.alloc(ctx, (size_t)(cell_count * npts)));
    PPH_HIP(ctx, hipMemcpyAsync(se.p, exact_q_host, sizeof(double) * (size_t)(cell_count * npts), hipMemcpyHostToDevice, ctx->stream));
  }
  if (grad_q_host) {
    PPH_TRY(sg.alloc(ctx, (size_t)(cell_count * npts * m.dim)));
    PPH_HIP(ctx, hipMemcpyAsync(sg.p, grad_q_host, sizeof(double) * (size_t)(cell_count * npts * m.dim), hipMemcpyHostToDevice, ctx->stream));
  }
  GaussRule g;
  g.nq = nq;
  gauss_legendre(nq, g.x, g.w);
  ErrSamples es{exact_q_host ? se.p : nullptr, grad_q_host ? sg.p : nullptr, nullptr, cell_begin, cell_begin + cell_count};
  const int64_t nb = ceil_div64(cell_count, 256);
  const int grid = (int)(nb < 2048 ? nb : 2048);
  err_launch<1>(ctx, m, u.p, g, part.p, es, grid);
  hipLaunchKernelGGL(k_sum_partials, dim3(2), dim3(256), 0, ctx->stream, part.p, grid, part.p + 4096);
  double r[2];
  PPH_HIP(ctx, hipMemcpyAsync(r, part.p + 4096, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PPH_HIP(ctx, hipGetLastError());
  *l2sq_out = r[0];
  *h1sq_out = r[1];
  part.release(); se.release(); sg.release();
  return PPH_OK;
}

// ------------------------------------------------------------------------------------------------
// Darcy velocity u = -k grad(p_h), L2-projected onto the CG-1 vector space: M u_d = b_d with
// b_d[a] = int -k (d p_h / d x_d) phi_a  (reference src/perphil/utils/postprocessing.py:34-63, fd.project).
// Node-centred right-hand side (one thread per node walks its incident cells; deterministic), then one
// Jacobi-CG solve with the mass matrix per component.
// ------------------------------------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(256) void k_darcy_rhs_multilinear(const int32_t* __restrict__ cells,
                                                               const double* __restrict__ cx, const double* __restrict__ cy,
                                                               const double* __restrict__ cz, const double* __restrict__ p,
                                                               double kcond, int nx, int ny, int nzl, int px, int py,
                                                               int64_t n, double* __restrict__ b /* [DIM][n] */) {
  constexpr int NB = 1 << DIM;
  const double gp = 0.57735026918962576451;
  for (int64_t node = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; node < n;
       node += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(node % px);
    const int64_t t = node / px;
    const int j = (int)(t % py), k = (int)(t / py);
    double acc[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] = 0.0;
    for (int c = 0; c < NB; ++c) {
      const int ci = i - (c & 1), cj = j - ((c >> 1) & 1), ck = (DIM == 3) ? k - ((c >> 2) & 1) : 0;
      if (ci < 0 || ci >= nx || cj < 0 || cj >= ny || (DIM == 3 && (ck < 0 || ck >= nzl))) continue;
      const int64_t cell = ci + (int64_t)nx * (cj + (int64_t)ny * ck);
      double X[NB][DIM], P[NB];
      int a = -1;
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const int32_t nd = cells[cell * NB + bb];
        a = (nd == (int32_t)node) ? bb : a;
        X[bb][0] = cx[nd];
        X[bb][1] = cy[nd];
        if constexpr (DIM == 3) X[bb][2] = cz[nd];
        P[bb] = p[nd];
      }
      if (a < 0) continue;
      for (int q = 0; q < NB; ++q) {
        double xi[DIM];
#pragma unroll
        for (int e = 0; e < DIM; ++e) xi[e] = ((q >> e) & 1) ? gp : -gp;
        double J[DIM][DIM], gref[DIM], Na = 1.0;
#pragma unroll
        for (int e = 0; e < DIM; ++e) {
          gref[e] = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; ++d) J[e][d] = 0.0;
        }
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          double nv = 1.0, dn[DIM];
#pragma unroll
          for (int e = 0; e < DIM; ++e) {
            const double s = ((bb >> e) & 1) ? 1.0 : -1.0;
            nv *= 0.5 * (1.0 + s * xi[e]);
            double dd = 1.0;
#pragma unroll
            for (int f = 0; f < DIM; ++f) {
              const double sf = ((bb >> f) & 1) ? 1.0 : -1.0;
              dd *= (f == e) ? 0.5 * sf : 0.5 * (1.0 + sf * xi[f]);
            }
            dn[e] = dd;
          }
          Na = (bb == a) ? nv : Na;
#pragma unroll
          for (int e = 0; e < DIM; ++e) {
            gref[e] += dn[e] * P[bb];
#pragma unroll
            for (int d = 0; d < DIM; ++d) J[e][d] += dn[e] * X[bb][d];
          }
        }
        double det, I[DIM][DIM];
        if constexpr (DIM == 2) {
          det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
          const double r = 1.0 / det;
          I[0][0] = J[1][1] * r;  I[0][1] = -J[0][1] * r;
          I[1][0] = -J[1][0] * r; I[1][1] = J[0][0] * r;
        } else {
          const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
          const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
          const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
          det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
          const double r = 1.0 / det;
          I[0][0] = c00 * r;
          I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
          I[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
          I[1][0] = c01 * r;
          I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
          I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
          I[2][0] = c02 * r;
          I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
          I[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
        }
        const double w = fabs(det) * Na * (-kcond);
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double g = 0.0;
#pragma unroll
          for (int e = 0; e < DIM; ++e) g += I[d][e] * gref[e];
          acc[d] += w * g;
        }
      }
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d) b[(int64_t)d * n + node] = acc[d];
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void k_darcy_rhs_simplex(const int32_t* __restrict__ cells, const double* __restrict__ cx,
                                                           const double* __restrict__ cy, const double* __restrict__ cz,
                                                           const double* __restrict__ p, double kcond, int nx, int ny,
                                                           int nzl, int px, int py, int64_t n, double* __restrict__ b) {
  constexpr int NB = DIM + 1;
  constexpr int NSUB = (DIM == 2) ? 2 : 6;
  for (int64_t node = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; node < n;
       node += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(node % px);
    const int64_t t = node / px;
    const int j = (int)(t % py), k = (int)(t / py);
    double acc[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] = 0.0;
    for (int c = 0; c < (1 << DIM); ++c) {
      const int bi = i - (c & 1), bj = j - ((c >> 1) & 1), bk = (DIM == 3) ? k - ((c >> 2) & 1) : 0;
      if (bi < 0 || bi >= nx || bj < 0 || bj >= ny || (DIM == 3 && (bk < 0 || bk >= nzl))) continue;
      const int64_t box = bi + (int64_t)nx * (bj + (int64_t)ny * bk);
      for (int sub = 0; sub < NSUB; ++sub) {
        const int64_t cell = box * NSUB + sub;
        double X[NB][DIM], P[NB];
        bool has = false;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          const int32_t nd = cells[cell * NB + bb];
          has = has || (nd == (int32_t)node);
          X[bb][0] = cx[nd];
          X[bb][1] = cy[nd];
          if constexpr (DIM == 3) X[bb][2] = cz[nd];
          P[bb] = p[nd];
        }
        if (!has) continue;
        double E[DIM][DIM], dP[DIM];
#pragma unroll
        for (int r = 0; r < DIM; ++r) {
          dP[r] = P[r + 1] - P[0];
#pragma unroll
          for (int d = 0; d < DIM; ++d) E[r][d] = X[r + 1][d] - X[0][d];
        }
        // grad p = E^-1 dP  (E rows = edge vectors): solve with the adjugate
        double det, g[DIM];
        if constexpr (DIM == 2) {
          det = E[0][0] * E[1][1] - E[0][1] * E[1][0];
          g[0] = (E[1][1] * dP[0] - E[0][1] * dP[1]) / det;
          g[1] = (-E[1][0] * dP[0] + E[0][0] * dP[1]) / det;
        } else {
          const double c00 = E[1][1] * E[2][2] - E[1][2] * E[2][1];
          const double c01 = E[1][2] * E[2][0] - E[1][0] * E[2][2];
          const double c02 = E[1][0] * E[2][1] - E[1][1] * E[2][0];
          det = E[0][0] * c00 + E[0][1] * c01 + E[0][2] * c02;
          const double r = 1.0 / det;
          g[0] = (c00 * dP[0] + (E[0][2] * E[2][1] - E[0][1] * E[2][2]) * dP[1] + (E[0][1] * E[1][2] - E[0][2] * E[1][1]) * dP[2]) * r;
          g[1] = (c01 * dP[0] + (E[0][0] * E[2][2] - E[0][2] * E[2][0]) * dP[1] + (E[0][2] * E[1][0] - E[0][0] * E[1][2]) * dP[2]) * r;
          g[2] = (c02 * dP[0] + (E[0][1] * E[2][0] - E[0][0] * E[2][1]) * dP[1] + (E[0][0] * E[1][1] - E[0][1] * E[1][0]) * dP[2]) * r;
        }
        const double w = -kcond * fabs(det) / (DIM == 2 ? 2.0 : 6.0) / (double)(DIM + 1);
#pragma unroll
        for (int d = 0; d < DIM; ++d) acc[d] += w * g[d];
      }
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d) b[(int64_t)d * n + node] = acc[d];
  }
}

extern "C" int pph_darcy_velocity(pph_ctx* ctx, const double* p_host, double conductivity, double* u_host) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, ctx->mesh_ok && p_host && u_host, "pph_darcy_velocity: no mesh or NULL buffer");
  PPH_REQUIRE(ctx, ctx->world == 1, "Darcy velocity projection is implemented for single-context meshes");
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  MeshData& m = ctx->mesh;
  if (!m.km_valid) {
    PPH_TRY(pph_launch_assemble_KM(ctx, m));
    m.km_valid = true;
  }
  const int64_t n = m.n;
  const int dim = m.dim;
  DevBuf<double> p, b, u, dinv, w1, w2, w3, w4;
  PPH_TRY(p.alloc(ctx, (size_t)n));
  PPH_TRY(b.alloc(ctx, (size_t)n * dim));
  PPH_TRY(u.alloc(ctx, (size_t)n));
  PPH_TRY(dinv.alloc(ctx, (size_t)n));
  PPH_TRY(w1.alloc(ctx, (size_t)n)); PPH_TRY(w2.alloc(ctx, (size_t)n));
  PPH_TRY(w3.alloc(ctx, (size_t)n)); PPH_TRY(w4.alloc(ctx, (size_t)n));
  PPH_HIP(ctx, hipMemcpyAsync(p.p, p_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  int64_t nb = ceil_div64(n, 256);
  const int grid = (int)(nb < 8192 ? nb : 8192);
  const int nzl = (dim == 3) ? m.nzl : 0;
  if (m.kind == PPH_CELL_QUAD)
    hipLaunchKernelGGL(k_darcy_rhs_multilinear<2>, dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p,
                       p.p, conductivity, m.nx, m.ny, nzl, m.px, m.py, n, b.p);
  else if (m.kind == PPH_CELL_HEX)
    hipLaunchKernelGGL(k_darcy_rhs_multilinear<3>, dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p,
                       p.p, conductivity, m.nx, m.ny, nzl, m.px, m.py, n, b.p);
  else if (m.kind == PPH_CELL_TRI)
    hipLaunchKernelGGL(k_darcy_rhs_simplex<2>, dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p, p.p,
                       conductivity, m.nx, m.ny, nzl, m.px, m.py, n, b.p);
  else
    hipLaunchKernelGGL(k_darcy_rhs_simplex<3>, dim3(grid), dim3(256), 0, ctx->stream, m.cells.p, m.cx.p, m.cy.p, m.cz.p, p.p,
                       conductivity, m.nx, m.ny, nzl, m.px, m.py, n, b.p);
  PPH_HIP(ctx, hipGetLastError());
  Csr M;
  M.rowptr = m.rowptr.p; M.col = m.col.p; M.val = m.M.p; M.nrows = n; M.nnz = m.nnzb; M.max_row = m.max_row;
  M.lanes = pph_pick_lanes(ctx, M.nnz, M.nrows);
  la_extract_diag_inv(ctx, M, dinv.p);
  std::vector<double> tmp((size_t)n);
  for (int d = 0; d < dim; ++d) {
    int its = 0;
    PPH_TRY(pph_cg_jacobi(ctx, M, b.p + (size_t)d * n, u.p, dinv.p, 1e-13, 0.0, 1000, w1.p, w2.p, w3.p, w4.p, &its));
    PPH_HIP(ctx, hipMemcpyAsync(tmp.data(), u.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < n; ++i) u_host[(size_t)i * dim + d] = tmp[(size_t)i];
  }
  p.release(); b.release(); u.release(); dinv.release(); w1.release(); w2.release(); w3.release(); w4.release();
  return PPH_OK;
}
