// K-ASM / K-BC: cell-local stiffness + mass integration, wavefront-segmented scatter-add into CSR,
// Dirichlet elimination, DPP block formation and the lifted right-hand side.
//
// Replaces, for reference src/perphil/forms/dpp.py:57-58,89-90,129-130 (dpp_form) and :196-203
// (dpp_delayed_form), the TSFC-generated element kernel + PyOP2 cell loop + PETSc MatSetValues +
// Firedrake BC application that run inside solver.solve() (src/perphil/solvers/solver.py:71).
//
// Layout: one lane per (cell, local row a).  A 256-thread workgroup takes 256/NB cells per batch;
// their cell->dof entries are read coalesced, nodal coordinates and the reference basis tables
// (values + gradients at the 2^d Gauss points) are staged in LDS, each lane integrates row a of K_e
// and M_e, and the rows are scattered into the scalar CSR arrays with fp64 atomics.  Lanes of
// neighbouring cells in one wavefront that target the same CSR slot are merged by shuffle first
// (segments found from the cell->dof map itself, so the merge is also correct on any other numbering).
#include "pph_internal.h"
#include <hip/hip_runtime.h>
#include <type_traits>

__device__ inline int64_t find_slot(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                    int32_t row, int32_t c) {
  int64_t lo = rowptr[row], hi = rowptr[row + 1] - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (col[mid] < c) lo = mid + 1; else hi = mid;
  }
  return lo;  // col[lo] == c by construction of the pattern
}

__device__ inline void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// ------------------------------------------------------------------------------------------------
// multilinear cells (Q1 quad: DIM 2, Q1 hex: DIM 3), 2-point Gauss per direction
// ------------------------------------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(256) void k_asm_multilinear(const int32_t* __restrict__ cells,
                                                          const double* __restrict__ cx,
                                                          const double* __restrict__ cy,
                                                          const double* __restrict__ cz,
                                                          const int64_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col, double* __restrict__ K,
                                                          double* __restrict__ M, int64_t ncell) {
  constexpr int NB = 1 << DIM;     // nodes per cell == Gauss points
  constexpr int CPB = 256 / NB;    // cells per workgroup batch
  constexpr int CPW = 64 / NB;     // cells per wavefront
  __shared__ double sN[NB][NB];           // [q][b]   basis values
  __shared__ double sdN[NB][NB][DIM];     // [q][b][e] reference gradients
  __shared__ double sX[CPB][NB][DIM];     // nodal coordinates of the batch
  __shared__ int32_t sC[CPB + 1][NB];     // cell->dof entries of the batch (+1 row: sentinel)

  const int tid = threadIdx.x;
  const int lc = tid / NB;  // cell inside the batch
  const int a = tid % NB;   // local row
  if (tid < NB * NB) {
    const int q = tid / NB, b = tid % NB;
    const double gp = 0.57735026918962576451;  // 1/sqrt(3)
    double xi[DIM], s[DIM];
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      xi[e] = ((q >> e) & 1) ? gp : -gp;
      s[e] = ((b >> e) & 1) ? 1.0 : -1.0;
    }
    double nv = 1.0;
#pragma unroll
    for (int e = 0; e < DIM; ++e) nv *= 0.5 * (1.0 + s[e] * xi[e]);
    sN[q][b] = nv;
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      double d = 1.0;
#pragma unroll
      for (int f = 0; f < DIM; ++f) d *= (f == e) ? 0.5 * s[f] : 0.5 * (1.0 + s[f] * xi[f]);
      sdN[q][b][e] = d;
    }
  }
  if (tid < NB) sC[CPB][tid] = -1;

  const int64_t nbatch = (ncell + CPB - 1) / CPB;
  for (int64_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    __syncthreads();  // previous batch fully consumed (also orders the table writes on the first pass)
    const int64_t cell = batch * CPB + lc;
    const bool valid = cell < ncell;
    int32_t node = -1;
    if (valid) {
      node = cells[cell * NB + a];
      sX[lc][a][0] = cx[node];
      sX[lc][a][1] = cy[node];
      if constexpr (DIM == 3) sX[lc][a][2] = cz[node];
    }
    sC[lc][a] = node;
    __syncthreads();

    double Kr[NB], Mr[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { Kr[b] = 0.0; Mr[b] = 0.0; }

    if (valid) {
#pragma unroll 1
      for (int q = 0; q < NB; ++q) {
        double J[DIM][DIM];  // J[e][d] = d x_d / d xi_e
#pragma unroll
        for (int e = 0; e < DIM; ++e)
#pragma unroll
          for (int d = 0; d < DIM; ++d) J[e][d] = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < DIM; ++e)
#pragma unroll
            for (int d = 0; d < DIM; ++d) J[e][d] += sdN[q][b][e] * sX[lc][b][d];
        double det, I[DIM][DIM];  // I[d][e] = d xi_e / d x_d
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
          // inverse of J (as a matrix indexed [e][d]) is Jinv[d][e]
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
        const double wdet = fabs(det);  // Gauss weights are 1
        double Ga[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double g = 0.0;
#pragma unroll
          for (int e = 0; e < DIM; ++e) g += I[d][e] * sdN[q][a][e];
          Ga[d] = g;
        }
        const double Na = sN[q][a];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          double dotg = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; ++d) {
            double g = 0.0;
#pragma unroll
            for (int e = 0; e < DIM; ++e) g += I[d][e] * sdN[q][b][e];
            dotg += Ga[d] * g;
          }
          Kr[b] += wdet * dotg;
          Mr[b] += wdet * Na * sN[q][b];
        }
      }
    }

    // ---- wavefront-segmented scatter-add -------------------------------------------------------
    // Entry (a,b) of cell c and entry (a-1,b-1) of cell c+1 hit the same CSR slot when the two cells
    // share the corresponding nodes (x-neighbours in the lexicographic cell order).  The left lane
    // (a odd) takes the right lane's value by shuffle and issues one atomic for both.
    const int lane = tid & 63;
    const int cw = lane / NB;  // cell inside the wavefront
    const bool has_right = (cw + 1 < CPW);
    const bool has_left = (cw > 0);
    const int32_t row = valid ? sC[lc][a] : -1;
    // does my row coincide with the partner's row?
    const bool a_odd = (a & 1) != 0;
    const int32_t right_row = (a_odd && has_right) ? sC[lc + 1][a - 1] : -2;   // sentinel row holds -1
    const int32_t left_row = (!a_odd && has_left && lc > 0) ? sC[lc - 1][a + 1] : -2;
    const bool row_recv = valid && a_odd && right_row == row;
    const bool row_give = valid && !a_odd && left_row == row;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      double kv = Kr[b], mv = Mr[b];
      bool emit = valid;
      if (b & 1) {
        // candidate receiver: partner lane = lane + NB - 1 holds entry (a-1, b-1) of the next cell
        const double kp = __shfl(Kr[b - 1], lane + NB - 1, 64);
        const double mp = __shfl(Mr[b - 1], lane + NB - 1, 64);
        if (row_recv && sC[lc + 1][b - 1] == sC[lc][b]) { kv += kp; mv += mp; }
      } else {
        // candidate giver: my (a, b) with a even, b even is merged into the previous cell's (a+1, b+1)
        if (row_give && sC[lc - 1][b + 1] == sC[lc][b]) emit = false;
      }
      if (emit) {
        const int64_t slot = find_slot(rowptr, col, row, sC[lc][b]);
        atomic_add_f64(K + slot, kv);
        atomic_add_f64(M + slot, mv);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// simplices (P1 triangle: DIM 2, P1 tetrahedron: DIM 3): constant gradients, closed-form mass
// one lane per (cell, local row); 4 lanes per cell (triangles leave the 4th lane idle)
// ------------------------------------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(256) void k_asm_simplex(const int32_t* __restrict__ cells,
                                                     const double* __restrict__ cx, const double* __restrict__ cy,
                                                     const double* __restrict__ cz,
                                                     const int64_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, double* __restrict__ K,
                                                     double* __restrict__ M, int64_t ncell) {
  constexpr int NB = DIM + 1;
  const int64_t total = ncell * 4;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = t >> 2;
    const int a = (int)(t & 3);
    if (a >= NB) continue;
    int32_t nd[NB];
    double X[NB][DIM];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      nd[b] = cells[cell * NB + b];
      X[b][0] = cx[nd[b]];
      X[b][1] = cy[nd[b]];
      if constexpr (DIM == 3) X[b][2] = cz[nd[b]];
    }
    double E[DIM][DIM];  // E[r][d] = X[r+1][d] - X[0][d]
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int d = 0; d < DIM; ++d) E[r][d] = X[r + 1][d] - X[0][d];
    double det, G[NB][DIM];  // G[b][d] = d lambda_b / d x_d ; columns of E^-1 for b >= 1
    if constexpr (DIM == 2) {
      det = E[0][0] * E[1][1] - E[0][1] * E[1][0];
      const double r = 1.0 / det;
      // E^-1 = 1/det [[E11, -E01], [-E10, E00]] ; G[b][d] = Einv[d][b-1]
      G[1][0] = E[1][1] * r;  G[1][1] = -E[1][0] * r;
      G[2][0] = -E[0][1] * r; G[2][1] = E[0][0] * r;
    } else {
      const double c00 = E[1][1] * E[2][2] - E[1][2] * E[2][1];
      const double c01 = E[1][2] * E[2][0] - E[1][0] * E[2][2];
      const double c02 = E[1][0] * E[2][1] - E[1][1] * E[2][0];
      det = E[0][0] * c00 + E[0][1] * c01 + E[0][2] * c02;
      const double r = 1.0 / det;
      double Inv[3][3];  // Inv = E^-1, Inv[d][r]
      Inv[0][0] = c00 * r;
      Inv[0][1] = (E[0][2] * E[2][1] - E[0][1] * E[2][2]) * r;
      Inv[0][2] = (E[0][1] * E[1][2] - E[0][2] * E[1][1]) * r;
      Inv[1][0] = c01 * r;
      Inv[1][1] = (E[0][0] * E[2][2] - E[0][2] * E[2][0]) * r;
      Inv[1][2] = (E[0][2] * E[1][0] - E[0][0] * E[1][2]) * r;
      Inv[2][0] = c02 * r;
      Inv[2][1] = (E[0][1] * E[2][0] - E[0][0] * E[2][1]) * r;
      Inv[2][2] = (E[0][0] * E[1][1] - E[0][1] * E[1][0]) * r;
#pragma unroll
      for (int b = 1; b < NB; ++b)
#pragma unroll
        for (int d = 0; d < DIM; ++d) G[b][d] = Inv[d][b - 1];
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double s = 0.0;
#pragma unroll
      for (int b = 1; b < NB; ++b) s += G[b][d];
      G[0][d] = -s;
    }
    const double vol = fabs(det) / (DIM == 2 ? 2.0 : 6.0);
    const double mfac = vol / (double)((DIM + 1) * (DIM + 2));
    double Ga[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double g = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) g = (b == a) ? G[b][d] : g;
      Ga[d] = g;
    }
    int32_t row = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) row = (b == a) ? nd[b] : row;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      double dotg = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) dotg += Ga[d] * G[b][d];
      const int64_t slot = find_slot(rowptr, col, row, nd[b]);
      atomic_add_f64(K + slot, vol * dotg);
      atomic_add_f64(M + slot, (b == a) ? 2.0 * mfac : mfac);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Node-centred (gather) assembly for multilinear cells: deterministic, no atomics, every CSR entry is
// written exactly once.  One lane per (node, incident cell): the 2^d lanes of a node integrate the row of
// K_e / M_e that belongs to the node in each of its (up to) 2^d incident cells — the same quadrature as
// the scatter kernel — and park the rows in LDS; then the lanes of the node walk its CSR row and sum, in a
// fixed order, the parked entries whose column matches.  Incident cells are enumerated from the box
// structure, the local row index and the column matching come from the cell->dof map.
// ------------------------------------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(256) void k_asm_gather(const int32_t* __restrict__ cells, const double* __restrict__ cx,
                                                    const double* __restrict__ cy, const double* __restrict__ cz,
                                                    const int64_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col, double* __restrict__ K,
                                                    double* __restrict__ M, int nx, int ny, int nzl, int px, int py,
                                                    int64_t n) {
  constexpr int NB = 1 << DIM;   // nodes per cell == incident cells per node == Gauss points
  constexpr int NPB = 256 / NB;  // nodes per workgroup batch
  __shared__ double sN[NB][NB];
  __shared__ double sdN[NB][NB][DIM];
  __shared__ double sK[NPB][NB][NB];    // [node][incident cell][local column]
  __shared__ double sM[NPB][NB][NB];
  __shared__ int32_t sC[NPB][NB][NB];   // node ids of the incident cells (-1: cell absent)
  const int tid = threadIdx.x;
  if (tid < NB * NB) {
    const int q = tid / NB, b = tid % NB;
    const double gp = 0.57735026918962576451;
    double xi[DIM], sg[DIM];
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      xi[e] = ((q >> e) & 1) ? gp : -gp;
      sg[e] = ((b >> e) & 1) ? 1.0 : -1.0;
    }
    double nv = 1.0;
#pragma unroll
    for (int e = 0; e < DIM; ++e) nv *= 0.5 * (1.0 + sg[e] * xi[e]);
    sN[q][b] = nv;
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      double d = 1.0;
#pragma unroll
      for (int f = 0; f < DIM; ++f) d *= (f == e) ? 0.5 * sg[f] : 0.5 * (1.0 + sg[f] * xi[f]);
      sdN[q][b][e] = d;
    }
  }
  const int ln = tid / NB;  // node inside the batch
  const int c = tid % NB;   // incident-cell slot: the node is local vertex c of that cell
  const int64_t nbatch = (n + NPB - 1) / NPB;
  for (int64_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    __syncthreads();
    const int64_t node = batch * NPB + ln;
    bool valid = node < n;
    int64_t cell = -1;
    if (valid) {
      const int i = (int)(node % px);
      const int64_t t = node / px;
      const int j = (int)(t % py), k = (int)(t / py);
      const int ci = i - (c & 1), cj = j - ((c >> 1) & 1), ck = (DIM == 3) ? k - ((c >> 2) & 1) : 0;
      const bool inb = ci >= 0 && ci < nx && cj >= 0 && cj < ny && (DIM == 2 || (ck >= 0 && ck < nzl));
      if (inb) cell = ci + (int64_t)nx * (cj + (int64_t)ny * ck);
    }
    double Kr[NB], Mr[NB];
    int32_t nd[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { Kr[b] = 0.0; Mr[b] = 0.0; nd[b] = -1; }
    if (cell >= 0) {
      double X[NB][DIM];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        nd[b] = cells[cell * NB + b];
        X[b][0] = cx[nd[b]];
        X[b][1] = cy[nd[b]];
        if constexpr (DIM == 3) X[b][2] = cz[nd[b]];
      }
#pragma unroll 1
      for (int q = 0; q < NB; ++q) {
        double J[DIM][DIM];
#pragma unroll
        for (int e = 0; e < DIM; ++e)
#pragma unroll
          for (int d = 0; d < DIM; ++d) J[e][d] = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < DIM; ++e)
#pragma unroll
            for (int d = 0; d < DIM; ++d) J[e][d] += sdN[q][b][e] * X[b][d];
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
        const double wdet = fabs(det);
        double Ga[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double g = 0.0;
#pragma unroll
          for (int e = 0; e < DIM; ++e) g += I[d][e] * sdN[q][c][e];
          Ga[d] = g;
        }
        const double Na = sN[q][c];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          double dotg = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; ++d) {
            double g = 0.0;
#pragma unroll
            for (int e = 0; e < DIM; ++e) g += I[d][e] * sdN[q][b][e];
            dotg += Ga[d] * g;
          }
          Kr[b] += wdet * dotg;
          Mr[b] += wdet * Na * sN[q][b];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      sK[ln][c][b] = Kr[b];
      sM[ln][c][b] = Mr[b];
      sC[ln][c][b] = nd[b];
    }
    __syncthreads();
    if (valid) {
      const int64_t s = rowptr[node], e = rowptr[node + 1];
      for (int64_t k = s + c; k < e; k += NB) {
        const int32_t j = col[k];
        double kv = 0.0, mv = 0.0;
#pragma unroll
        for (int cc = 0; cc < NB; ++cc)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const bool hit = sC[ln][cc][b] == j;
            kv += hit ? sK[ln][cc][b] : 0.0;
            mv += hit ? sM[ln][cc][b] : 0.0;
          }
        K[k] = kv;
        M[k] = mv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Two-pass deterministic assembly for multilinear cells (default).
// Pass 1 (cell-centred): lane (cell, q) evaluates the Jacobian, its inverse and the physical basis
// gradients at ONE Gauss point and shares them through LDS; lane (cell, a) then contracts them into row a
// of K_e and M_e and streams the row (16 doubles) to an element-matrix buffer.  The Jacobian work is done
// once per (cell, Gauss point) instead of once per (cell, row, Gauss point).
// Pass 2 (node-centred): the 2^d lanes of a node fetch the rows that belong to it from its incident cells
// and sum matching columns into the CSR row in a fixed order: no atomics, bitwise reproducible, every
// CSR entry written once.
// ------------------------------------------------------------------------------------------------
// Element-row buffer addressing.  The buffer is either the whole mesh ([cell][a][K row | M row]) or a RING of a few
// cell layers: the assembly then alternates element pass (cell layers [begin, end)) and node-centred pass (the node
// planes those layers complete), so that the rows are still in the memory-side cache (256 MB) when they are read
// back and never travel to HBM and back (34 GB per 256^3 assembly otherwise).
struct ERing {
  int64_t begin, end;      // cells (element pass) or nodes (gather pass) handled by this launch
  int64_t layer_cells;     // cells per layer
  int ring;                // layers in the buffer (>= number of layers: plain addressing)
  __host__ __device__ int64_t slot(int64_t cell) const {
    return ((cell / layer_cells) % ring) * layer_cells + cell % layer_cells;
  }
};

template <int DIM>
__global__ __launch_bounds__(256) void k_elem_rows(const int32_t* __restrict__ cells, const double* __restrict__ cx,
                                                   const double* __restrict__ cy, const double* __restrict__ cz,
                                                   double* __restrict__ erows, int64_t ncell, ERing er) {
  constexpr int NB = 1 << DIM;
  constexpr int CPB = 256 / NB;
  __shared__ double sN[NB][NB];
  __shared__ double sdN[NB][NB][DIM];
  // per-cell strides are padded off the bank period: the 8 cells of a wavefront read the same (q, b, d)
  // of DIFFERENT cells in the contraction, and an unpadded stride (192 doubles = a multiple of all 64 banks)
  // serialises those reads 8-fold
  constexpr int SXC = NB * DIM + 1;        // sX cell stride
  constexpr int SGQ = NB * DIM + 1;        // sG Gauss-point stride
  constexpr int SGC = NB * SGQ + 2;        // sG cell stride
  constexpr int SWC = NB + 1;              // sW cell stride
  constexpr int SOR = 2 * NB + 1;          // output staging: row stride (2 NB values + 1 pad), reuses sGf
  static_assert(CPB * NB * SOR <= CPB * SGC, "output staging must fit in the gradient buffer");
  __shared__ double sXf[CPB * SXC];
  __shared__ double sGf[CPB * SGC];        // [cell][q][b][d] physical gradients
  __shared__ double sWf[CPB * SWC];        // [cell][q] |det J| (Gauss weights are 1)
#define sX(c_, b_, d_) sXf[(c_) * SXC + (b_) * DIM + (d_)]
#define sG(c_, q_, b_, d_) sGf[(c_) * SGC + (q_) * SGQ + (b_) * DIM + (d_)]
#define sW(c_, q_) sWf[(c_) * SWC + (q_)]
  const int tid = threadIdx.x;
  const int lc = tid / NB, a = tid % NB;
  if (tid < NB * NB) {
    const int q = tid / NB, b = tid % NB;
    const double gp = 0.57735026918962576451;
    double xi[DIM], sg[DIM];
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      xi[e] = ((q >> e) & 1) ? gp : -gp;
      sg[e] = ((b >> e) & 1) ? 1.0 : -1.0;
    }
    double nv = 1.0;
#pragma unroll
    for (int e = 0; e < DIM; ++e) nv *= 0.5 * (1.0 + sg[e] * xi[e]);
    sN[q][b] = nv;
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      double d = 1.0;
#pragma unroll
      for (int f = 0; f < DIM; ++f) d *= (f == e) ? 0.5 * sg[f] : 0.5 * (1.0 + sg[f] * xi[f]);
      sdN[q][b][e] = d;
    }
  }
  const int64_t nbatch = (er.end - er.begin + CPB - 1) / CPB;
  // vertex coordinates of the NEXT batch (two dependent loads: cell->dof entry, then the coordinates) are requested
  // before the current batch is integrated
  double pxyz[3] = {0.0, 0.0, 0.0};
  auto fetch = [&](int64_t bt) {
    const int64_t cl = er.begin + bt * CPB + lc;
    if (bt < nbatch && cl < er.end) {
      const int32_t node = cells[cl * NB + a];
      pxyz[0] = cx[node];
      pxyz[1] = cy[node];
      if constexpr (DIM == 3) pxyz[2] = cz[node];
    }
  };
  fetch(blockIdx.x);
  for (int64_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    __syncthreads();
    const int64_t cell = er.begin + batch * CPB + lc;
    const bool valid = cell < er.end;
    if (valid) {
      sX(lc, a, 0) = pxyz[0];
      sX(lc, a, 1) = pxyz[1];
      if constexpr (DIM == 3) sX(lc, a, 2) = pxyz[2];
    }
    fetch(batch + gridDim.x);
    __syncthreads();
    if (valid) {
      const int q = a;  // this lane's Gauss point
      double J[DIM][DIM];
#pragma unroll
      for (int e = 0; e < DIM; ++e)
#pragma unroll
        for (int d = 0; d < DIM; ++d) J[e][d] = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < DIM; ++e)
#pragma unroll
          for (int d = 0; d < DIM; ++d) J[e][d] += sdN[q][b][e] * sX(lc, b, d);
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
      sW(lc, q) = fabs(det);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double g = 0.0;
#pragma unroll
          for (int e = 0; e < DIM; ++e) g += I[d][e] * sdN[q][b][e];
          sG(lc, q, b, d) = g;
        }
    }
    __syncthreads();
    // Contraction K_e = sum_{q,d} (w_q G[q][a][d]) G[q][b][d], M_e = sum_q (w_q N_q[a]) N_q[b] on the matrix cores:
    // v_mfma_f64_16x16x4_f64 tiles of 16/NB cells (block-diagonal part used), A/B operand of lane l = row/column
    // l & 15, k = 4 step + (l >> 4); result register r of lane l = D[(l >> 4) + 4 r][l & 15].  One LDS read per
    // operand pair instead of one per multiply-add: the scalar version of this loop was LDS-bandwidth bound.
    typedef double v4d __attribute__((ext_vector_type(4)));
    constexpr int CT = 16 / NB;                 // cells per tile
    constexpr int TPW = (64 / NB) / CT;         // tiles per wavefront (4)
    const int lane = tid & 63, wcell0 = (tid >> 6) * (64 / NB);
    const int ti = lane & 15, tk = lane >> 4;   // operand row/column and k offset of this lane
    v4d accK[TPW], accM[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int oc = wcell0 + t * CT + ti / NB, on = ti % NB;   // operand cell (in the batch) and vertex
      v4d ck = {0.0, 0.0, 0.0, 0.0}, cm = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int st = 0; st < NB * DIM / 4; ++st) {
        const int kk = 4 * st + tk, q = kk / DIM, d = kk % DIM;
        const double g = sG(oc, q, on, d);
        ck = __builtin_amdgcn_mfma_f64_16x16x4f64(g * sW(oc, q), g, ck, 0, 0, 0);
      }
#pragma unroll
      for (int st = 0; st < NB / 4; ++st) {
        const int q = 4 * st + tk;
        const double nv = sN[q][on];
        cm = __builtin_amdgcn_mfma_f64_16x16x4f64(nv * sW(oc, q), nv, cm, 0, 0, 0);
      }
      accK[t] = ck;
      accM[t] = cm;
    }
    // park the rows in LDS (the gradients are dead after this barrier): the workgroup then streams its CPB x NB
    // rows - one contiguous range of the element-row buffer - with coalesced 16-byte stores
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int cj = ti / NB, b = ti % NB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tk + 4 * r;
        if (i / NB == cj) {
          double* so = sGf + ((wcell0 + t * CT + cj) * NB + i % NB) * SOR;
          so[b] = accK[t][r];
          so[NB + b] = accM[t][r];
        }
      }
    }
    __syncthreads();
    {
      const int64_t cell0 = er.begin + batch * CPB;    // a launch never straddles a ring wrap (whole layers)
      const int64_t left = er.end - cell0;
      const int rows = (int)(left < CPB ? left : CPB) * NB;
      double* out = erows + er.slot(cell0) * NB * (2 * NB);
      for (int e2 = tid; e2 < rows * NB; e2 += 256) {     // one double2 per thread and pass
        const int row = e2 / NB, colp = 2 * (e2 % NB);
        const double* si = sGf + row * SOR + colp;
        *reinterpret_cast<double2*>(out + (int64_t)row * (2 * NB) + colp) = make_double2(si[0], si[1]);
      }
    }
  }
}
#undef sX
#undef sG
#undef sW

// Operands of the fused epilogue of k_gather_rows<DIM, true>: at the point where a CSR entry (K_ij, M_ij) of a
// complete row is known, the Dirichlet-eliminated DPP blocks, the lifted right-hand side and the smoother's
// diagonal / spectral bound of both diagonal blocks are formed from it directly, instead of writing K and M and
// streaming them again through k_lift_rhs, k_blocks and k_diag_lam.
struct FuseArgs {
  const uint8_t *m1, *m2, *near;
  const double *g1, *g2;
  double a, b, c;
  double *A11, *A22, *A12, *A21;   // A21 null: aliased to A12 (same Dirichlet set on both fields)
  double *rhs, *u0;                // [2n]
  double *dinv1, *dinv2;           // [n] each
  unsigned long long* lam;         // [2] max_i sum_j |a_ij| / |a_ii| as the bit pattern of a non-negative double
  int keep_km;                     // also store K and M
  int same;                        // both fields carry the same Dirichlet set (one mask gather per entry)
  // ld > 0: A11 .. A21 are stencil-ELL arrays (pph_sell.hip): entry (row, column row + (dx,dy,dz)) is stored at
  // [slot_of[(dz+1)*9 + (dy+1)*3 + (dx+1)] * ld + row]; ld == 0: CSR value arrays addressed by the pattern position
  // With symmetric storage (Sell::sym) only the diagonal and the upper slots are stored: slot_of is then the STORED
  // slot (s - S/2) or -1 for a lower slot, whose entry is not written.  The diagonal blocks (slot_of) and the coupling
  // blocks (slot_of_c: symmetric only when both fields carry the same Dirichlet set) have their own tables.
  int symg;                        // symmetric storage on a slab: ghost rows keep their entries towards owned columns
  int64_t ld;
  int8_t slot_of[27];
  int8_t slot_of_c[27];
  // k_asm_node2 on a uniform box (MeshData::uniform): the canonical edge lengths the cells are integrated on
  double hcan[3] = {0, 0, 0};
  // k_asm_node2, listed mode: row i of the (mini) output is the row of node list[i]
  const uint32_t* list = nullptr;
  // k_asm_node2, check mode (row dictionaries, pph_sell.hip "check fused into the assembly"): class arrays, tables and status
  // words of the operators this launch writes (0: A11, 1: A22, 2: A12; null: no dictionary), the alarm word of the context
  const uint16_t* dcls[3] = {nullptr, nullptr, nullptr};
  const double* dtab[3] = {nullptr, nullptr, nullptr};
  int* dstate[3] = {nullptr, nullptr, nullptr};
  int dn[3] = {0, 0, 0};
  int* alarm = nullptr;
  // host side only: the group and dictionaries behind the check, and the views they describe
  DictGroup* G = nullptr;
  SellDict* dicts[3] = {nullptr, nullptr, nullptr};
  const Sell* views[3] = {nullptr, nullptr, nullptr};
};

// Eliminated entries of the fused epilogues.  rm / cm: mask bytes of the row and of the column dof (bit 0 constrained,
// bit 1 ghost plane of a slab).  Ghost rows belong to the neighbouring slab and are empty here - except, with
// symmetric storage on slabs (`symg`), their entries towards OWNED columns: an owned row reads its lower entry
// (r, g) as the upper entry stored with row g, which the local cells between the two planes determine completely.
__device__ __forceinline__ double fuse_elim_diag(double v, uint8_t rm, uint8_t cm, bool diag, int symg) {
  if (rm & 2) return (symg && !(cm & 2) && !(rm & 1) && !(cm & 1)) ? v : 0.0;
  if (rm & 1) return diag ? 1.0 : 0.0;
  return (cm & 1) ? 0.0 : v;
}
__device__ __forceinline__ double fuse_elim_coupling(double v, uint8_t rm, uint8_t cm_other, int symg) {
  if (rm & 2) return (symg && !(cm_other & 2) && !(rm & 1) && !(cm_other & 1)) ? v : 0.0;
  return ((rm & 1) || (cm_other & 1)) ? 0.0 : v;
}

// max over the workgroup of the two spectral-bound candidates, then ONE atomic pair per workgroup (and none for zeros):
// atomics to one word are served one after the other (~50 ns each) - with a pair per WAVE, the 65 k atomics of a 4096 x 512
// launch cost 0.6 ms at the end of a small level's assembly
__device__ __forceinline__ void fuse_lam_max(double best1, double best2, unsigned long long* lam) {
  __shared__ double slam[2][16];
  for (int o = 32; o > 0; o >>= 1) {
    const double t1 = __shfl_down(best1, o, 64), t2 = __shfl_down(best2, o, 64);
    best1 = t1 > best1 ? t1 : best1;
    best2 = t2 > best2 ? t2 : best2;
  }
  if ((threadIdx.x & 63) == 0) { slam[0][threadIdx.x >> 6] = best1; slam[1][threadIdx.x >> 6] = best2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double m1 = 0.0, m2 = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { m1 = slam[0][w] > m1 ? slam[0][w] : m1; m2 = slam[1][w] > m2 ? slam[1][w] : m2; }
    if (m1 > 0.0) atomicMax(lam, (unsigned long long)__double_as_longlong(m1));
    if (m2 > 0.0) atomicMax(lam + 1, (unsigned long long)__double_as_longlong(m2));
  }
}

static void fuse_set_format(FuseArgs& fa, int kind, int64_t ld, int sym = 0, int sym_c = 0) {
  fa.ld = ld;
  for (int q = 0; q < 27; ++q) { fa.slot_of[q] = -1; fa.slot_of_c[q] = -1; }
  const Stencil st = make_stencil(kind);
  const int c0 = st.count / 2;
  for (int s = 0; s < st.count; ++s) {
    const int q = (st.d[s][2] + 1) * 9 + (st.d[s][1] + 1) * 3 + (st.d[s][0] + 1);
    fa.slot_of[q] = (int8_t)(sym ? (s >= c0 ? s - c0 : -1) : s);
    fa.slot_of_c[q] = (int8_t)(sym_c ? (s >= c0 ? s - c0 : -1) : s);
  }
}

// output position of the entry (node, column j): the CSR position k or the stencil-ELL address; -1: not stored
// (lower half of a symmetric operator).  coupling: the table of the coupling blocks.
__device__ __forceinline__ int64_t fuse_out_index(const FuseArgs& fa, int64_t k, int64_t node, int32_t j, int px, int py,
                                                  bool coupling = false) {
  if (fa.ld == 0) return k;
  const int i0 = (int)(node % px);
  const int64_t t0 = node / px;
  const int j0 = (int)(t0 % py), k0 = (int)(t0 / py);
  const int i1 = (int)(j % px);
  const int64_t t1 = (int64_t)j / px;
  const int j1 = (int)(t1 % py), k1 = (int)(t1 / py);
  const int q = (k1 - k0 + 1) * 9 + (j1 - j0 + 1) * 3 + (i1 - i0 + 1);
  const int s = coupling ? fa.slot_of_c[q] : fa.slot_of[q];
  return s < 0 ? -1 : (int64_t)s * fa.ld + node;
}
// A12 / A21 null: coupling blocks not wanted; rhs null: no lifting (multigrid coarse levels: g1, g2, u0 unused)

template <int DIM, bool FUSED = false, bool CLOSED = false>
__global__ __launch_bounds__(256) void k_gather_rows(const int32_t* __restrict__ cells,
                                                     const double* __restrict__ erows,
                                                     const int64_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, double* __restrict__ K,
                                                     double* __restrict__ M, int nx, int ny, int nzl, int px, int py,
                                                     int64_t n, FuseArgs fa, ERing er) {
  constexpr int NB = 1 << DIM;
  double best1 = 0.0, best2 = 0.0;
  constexpr int NPB = 256 / NB;
  __shared__ double sK[NPB][NB][NB + 1];  // +1: rows of different incident cells land in different banks
  __shared__ double sM[NPB][NB][NB + 1];
  // CLOSED (at least two cells per direction): columns are matched by corner arithmetic and only a validity flag
  // per incident cell is kept; otherwise the cell->dof entries are parked for a search (tiny meshes)
  __shared__ int32_t sC[CLOSED ? 1 : NPB][NB][NB];
  __shared__ uint8_t sOK[NPB][NB];
  const int tid = threadIdx.x;
  const int ln = tid / NB, c = tid % NB;
  const int64_t nbatch = (er.end - er.begin + NPB - 1) / NPB;
  // the element row and the cell->dof entries of the NEXT batch are requested before the current batch is matched
  // and stored (register double buffer): the two phases of a batch are separated by barriers, and with a dozen
  // waves per CU the loads of one phase were not overlapping the stores of the other
  double2 pk[NB / 2], pm[NB / 2];
  int32_t pc[NB];
  int64_t pcell = -1;
  bool pok = false;
  auto fetch = [&](int64_t bt) {
    pcell = -1;
    pok = false;
    const int64_t nd = er.begin + bt * NPB + ln;
    if (bt < nbatch && nd < er.end) {
      const int i = (int)(nd % px);
      const int64_t t = nd / px;
      const int j = (int)(t % py), k = (int)(t / py);
      const int ci = i - (c & 1), cj = j - ((c >> 1) & 1), ck = (DIM == 3) ? k - ((c >> 2) & 1) : 0;
      const bool inb = ci >= 0 && ci < nx && cj >= 0 && cj < ny && (DIM == 2 || (ck >= 0 && ck < nzl));
      if (inb) pcell = ci + (int64_t)nx * (cj + (int64_t)ny * ck);
    }
    if (pcell >= 0) {
      // the node is local vertex c of this cell (checked against the cell->dof map)
      const double* in = erows + (er.slot(pcell) * NB + c) * (2 * NB);
      const int32_t* cn = cells + pcell * NB;
#pragma unroll
      for (int b = 0; b < NB; b += 2) {
        pk[b / 2] = *reinterpret_cast<const double2*>(in + b);
        pm[b / 2] = *reinterpret_cast<const double2*>(in + NB + b);
      }
      if constexpr (CLOSED) {
        pok = cn[c] == (int32_t)nd;
      } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) pc[b] = cn[b];
        pok = pc[c] == (int32_t)nd;
      }
    }
  };
  fetch(blockIdx.x);
  for (int64_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    __syncthreads();
    const int64_t node = er.begin + batch * NPB + ln;
    const bool valid = node < er.end;
    if (pcell >= 0) {
#pragma unroll
      for (int b = 0; b < NB; b += 2) {
        sK[ln][c][b] = pk[b / 2].x; sK[ln][c][b + 1] = pk[b / 2].y;
        sM[ln][c][b] = pm[b / 2].x; sM[ln][c][b + 1] = pm[b / 2].y;
      }
      sOK[ln][c] = pok ? 1 : 0;
      if constexpr (!CLOSED) {
#pragma unroll
        for (int b = 0; b < NB; ++b) sC[ln][c][b] = pok ? pc[b] : -1;
      }
    } else {
      sOK[ln][c] = 0;
      if constexpr (!CLOSED) {
#pragma unroll
        for (int b = 0; b < NB; ++b) sC[ln][c][b] = -1;
      }
    }
    fetch(batch + gridDim.x);
    __syncthreads();
    if (valid) {
      const int64_t s = rowptr[node], e = rowptr[node + 1];
      // fused epilogue state of this lane's share of the row
      bool near = false;
      uint8_t r1 = 0, r2 = 0;
      double s11 = 0.0, s22 = 0.0, d11 = 0.0, d22 = 0.0, lK1 = 0.0, lK2 = 0.0, lM = 0.0;
      if constexpr (FUSED) {
        near = fa.near[node] != 0;
        if (near) { r1 = fa.m1[node]; r2 = fa.m2[node]; }
      }
      for (int64_t k = s + c; k < e; k += NB) {
        const int32_t j = col[k];
        double kv = 0.0, mv = 0.0;
        if constexpr (CLOSED) {
          // column = node + (dx, dy, dz); incident cell cc = (cx,cy,cz) holds the node at local corner cc and the
          // column at corner b = cc + d: per direction d=0 -> corners (0,0),(1,1); d=+1 -> (0,1); d=-1 -> (1,0)
          const int64_t pxy = (int64_t)px * py;
          int64_t d = (int64_t)j - node;
          int dz = 0;
          if (DIM == 3) { dz = (d > pxy / 2) ? 1 : (d < -(pxy / 2)) ? -1 : 0; d -= dz * pxy; }
          const int dy = (d > px / 2) ? 1 : (d < -(px / 2)) ? -1 : 0;
          const int dx = (int)(d - (int64_t)dy * px);
          const int dd[3] = {dx, dy, dz};
#pragma unroll 1
          for (int m = 0; m < NB; ++m) {   // m enumerates the candidate corner bits of the incident cell
            int cc = 0, b = 0;
            bool okc = true;
#pragma unroll
            for (int a = 0; a < DIM; ++a) {
              const int cb = (m >> a) & 1;          // corner bit of the node in the cell
              const int bb = cb + dd[a];            // corner bit of the column node
              okc = okc && (bb == 0 || bb == 1);
              cc |= cb << a;
              b |= (bb & 1) << a;
            }
            if (okc && sOK[ln][cc]) {
              kv += sK[ln][cc][b];
              mv += sM[ln][cc][b];
            }
          }
        } else {
#pragma unroll 1
          for (int cc = 0; cc < NB; ++cc)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const bool hit = sC[ln][cc][b] == j;
              kv += hit ? sK[ln][cc][b] : 0.0;
              mv += hit ? sM[ln][cc][b] : 0.0;
            }
        }
        if constexpr (!FUSED) {
          K[k] = kv;
          M[k] = mv;
        } else {
          if (fa.keep_km) { K[k] = kv; M[k] = mv; }
          const bool diag = (j == (int32_t)node);
          double o11 = fa.a * kv + fa.b * mv, o22 = fa.c * kv + fa.b * mv, o12 = -fa.b * mv, o21 = o12;
          if (near) {
            // same arithmetic as k_lift_rhs / k_blocks: ghost rows empty, Dirichlet rows identity, constrained columns zero
            const uint8_t cm1 = fa.m1[j], cm2 = fa.same ? cm1 : fa.m2[j];
            if (fa.rhs) {
              const double v1 = fa.g1[j], v2 = fa.g2[j];
              lK1 += kv * v1; lK2 += kv * v2; lM += mv * (v1 - v2);
            }
            o11 = fuse_elim_diag(o11, r1, cm1, diag, fa.symg);
            o22 = fuse_elim_diag(o22, r2, cm2, diag, fa.symg);
            o12 = fuse_elim_coupling(-fa.b * mv, r1, cm2, fa.symg);
            o21 = fuse_elim_coupling(-fa.b * mv, r2, cm1, fa.symg);
          }
          const int64_t ko = fuse_out_index(fa, k, node, j, px, py), kc = fuse_out_index(fa, k, node, j, px, py, true);
          if (ko >= 0) { fa.A11[ko] = o11; fa.A22[ko] = o22; }
          if (kc >= 0) {
            if (fa.A12) fa.A12[kc] = o12;
            if (fa.A21) fa.A21[kc] = o21;
          }
          s11 += fabs(o11); s22 += fabs(o22);
          if (diag) { d11 = o11; d22 = o22; }
        }
      }
      if constexpr (FUSED) {
        // row sums over the NB lanes of the node
#pragma unroll
        for (int o = NB / 2; o > 0; o >>= 1) {
          s11 += __shfl_down(s11, o, NB); s22 += __shfl_down(s22, o, NB);
          d11 += __shfl_down(d11, o, NB); d22 += __shfl_down(d22, o, NB);
          lK1 += __shfl_down(lK1, o, NB); lK2 += __shfl_down(lK2, o, NB); lM += __shfl_down(lM, o, NB);
        }
        if (c == 0) {
          const double i1 = (d11 != 0.0) ? 1.0 / d11 : 1.0, i2 = (d22 != 0.0) ? 1.0 / d22 : 1.0;
          fa.dinv1[node] = i1;
          fa.dinv2[node] = i2;
          const double q1 = s11 * fabs(i1), q2 = s22 * fabs(i2);
          if (!(r1 & 2)) best1 = q1 > best1 ? q1 : best1;   // (ghost rows: not rows of this rank's operator)
          if (!(r2 & 2)) best2 = q2 > best2 ? q2 : best2;
          if (fa.rhs) {
            fa.rhs[node] = (!near || r1 != 0) ? 0.0 : -(fa.a * lK1 + fa.b * lM);
            fa.rhs[n + node] = (!near || r2 != 0) ? 0.0 : -(fa.c * lK2 - fa.b * lM);
            fa.u0[node] = near ? fa.g1[node] : 0.0;
            fa.u0[n + node] = near ? fa.g2[node] : 0.0;
          }
        }
      }
    }
  }
  if constexpr (FUSED) {
    fuse_lam_max(best1, best2, fa.lam);
  }
}

// ------------------------------------------------------------------------------------------------
// Node-centred assembly for simplices (P1 triangles: 2 per square, P1 tetrahedra: 6 Kuhn tets per cube):
// one thread per node walks the cells of its (up to) 2^d incident boxes, keeps those that contain the node
// (membership and local index from the cell->dof map), evaluates the constant gradients and adds the node's
// row of K_e / M_e into the CSR row held in registers; fixed order, no atomics, every entry written once.
// ------------------------------------------------------------------------------------------------
template <int DIM, bool FUSED = false>
__global__ __launch_bounds__(256) void k_asm_simplex_gather(const int32_t* __restrict__ cells,
                                                            const double* __restrict__ cx, const double* __restrict__ cy,
                                                            const double* __restrict__ cz,
                                                            const int64_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col, double* __restrict__ K,
                                                            double* __restrict__ M, int nx, int ny, int nzl, int px, int py,
                                                            int64_t n, FuseArgs fa) {
  constexpr int NB = DIM + 1;
  double best1 = 0.0, best2 = 0.0;
  constexpr int NSUB = (DIM == 2) ? 2 : 6;
  constexpr int MAXROW = (DIM == 2) ? 7 : 15;
  for (int64_t node = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; node < n;
       node += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(node % px);
    const int64_t t = node / px;
    const int j = (int)(t % py), k = (int)(t / py);
    const int64_t s = rowptr[node];
    const int len = (int)(rowptr[node + 1] - s);
    int32_t cl[MAXROW];
    double aK[MAXROW], aM[MAXROW];
#pragma unroll
    for (int q = 0; q < MAXROW; ++q) {
      cl[q] = (q < len) ? col[s + q] : -1;
      aK[q] = 0.0;
      aM[q] = 0.0;
    }
    for (int c = 0; c < (1 << DIM); ++c) {
      const int bi = i - (c & 1), bj = j - ((c >> 1) & 1), bk = (DIM == 3) ? k - ((c >> 2) & 1) : 0;
      if (bi < 0 || bi >= nx || bj < 0 || bj >= ny || (DIM == 3 && (bk < 0 || bk >= nzl))) continue;
      const int64_t box = bi + (int64_t)nx * (bj + (int64_t)ny * bk);
      for (int sub = 0; sub < NSUB; ++sub) {
        const int64_t cell = box * NSUB + sub;
        int32_t nd[NB];
        int a = -1;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          nd[b] = cells[cell * NB + b];
          a = (nd[b] == (int32_t)node) ? b : a;
        }
        if (a < 0) continue;
        double X[NB][DIM];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          X[b][0] = cx[nd[b]];
          X[b][1] = cy[nd[b]];
          if constexpr (DIM == 3) X[b][2] = cz[nd[b]];
        }
        double E[DIM][DIM];
#pragma unroll
        for (int r = 0; r < DIM; ++r)
#pragma unroll
          for (int d = 0; d < DIM; ++d) E[r][d] = X[r + 1][d] - X[0][d];
        double det, G[NB][DIM];
        if constexpr (DIM == 2) {
          det = E[0][0] * E[1][1] - E[0][1] * E[1][0];
          const double r = 1.0 / det;
          G[1][0] = E[1][1] * r;  G[1][1] = -E[1][0] * r;
          G[2][0] = -E[0][1] * r; G[2][1] = E[0][0] * r;
        } else {
          const double c00 = E[1][1] * E[2][2] - E[1][2] * E[2][1];
          const double c01 = E[1][2] * E[2][0] - E[1][0] * E[2][2];
          const double c02 = E[1][0] * E[2][1] - E[1][1] * E[2][0];
          det = E[0][0] * c00 + E[0][1] * c01 + E[0][2] * c02;
          const double r = 1.0 / det;
          G[1][0] = c00 * r; G[1][1] = c01 * r; G[1][2] = c02 * r;
          G[2][0] = (E[0][2] * E[2][1] - E[0][1] * E[2][2]) * r;
          G[2][1] = (E[0][0] * E[2][2] - E[0][2] * E[2][0]) * r;
          G[2][2] = (E[0][1] * E[2][0] - E[0][0] * E[2][1]) * r;
          G[3][0] = (E[0][1] * E[1][2] - E[0][2] * E[1][1]) * r;
          G[3][1] = (E[0][2] * E[1][0] - E[0][0] * E[1][2]) * r;
          G[3][2] = (E[0][0] * E[1][1] - E[0][1] * E[1][0]) * r;
        }
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double sg = 0.0;
#pragma unroll
          for (int b = 1; b < NB; ++b) sg += G[b][d];
          G[0][d] = -sg;
        }
        const double vol = fabs(det) / (DIM == 2 ? 2.0 : 6.0);
        const double mfac = vol / (double)((DIM + 1) * (DIM + 2));
        double Ga[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double g = 0.0;
#pragma unroll
          for (int b = 0; b < NB; ++b) g = (b == a) ? G[b][d] : g;
          Ga[d] = g;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          double dotg = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; ++d) dotg += Ga[d] * G[b][d];
          const double kv = vol * dotg, mv = (b == a) ? 2.0 * mfac : mfac;
#pragma unroll
          for (int q = 0; q < MAXROW; ++q) {
            const bool hit = cl[q] == nd[b];
            aK[q] += hit ? kv : 0.0;
            aM[q] += hit ? mv : 0.0;
          }
        }
      }
    }
    if constexpr (!FUSED) {
#pragma unroll
      for (int q = 0; q < MAXROW; ++q)
        if (q < len) {
          K[s + q] = aK[q];
          M[s + q] = aM[q];
        }
    } else {
      // fused epilogue (see FuseArgs): the row is complete in registers
      const bool near = fa.near[node] != 0;
      uint8_t r1 = 0, r2 = 0;
      if (near) { r1 = fa.m1[node]; r2 = fa.m2[node]; }
      double s11 = 0.0, s22 = 0.0, d11 = 0.0, d22 = 0.0, lK1 = 0.0, lK2 = 0.0, lM = 0.0;
#pragma unroll
      for (int q = 0; q < MAXROW; ++q)
        if (q < len) {
          const double kv = aK[q], mv = aM[q];
          const int32_t jc = cl[q];
          if (fa.keep_km) { K[s + q] = kv; M[s + q] = mv; }
          const bool diag = (jc == (int32_t)node);
          double o11 = fa.a * kv + fa.b * mv, o22 = fa.c * kv + fa.b * mv, o12 = -fa.b * mv, o21 = o12;
          if (near) {
            const uint8_t cm1 = fa.m1[jc], cm2 = fa.same ? cm1 : fa.m2[jc];
            if (fa.rhs) {
              const double v1 = fa.g1[jc], v2 = fa.g2[jc];
              lK1 += kv * v1; lK2 += kv * v2; lM += mv * (v1 - v2);
            }
            o11 = fuse_elim_diag(o11, r1, cm1, diag, fa.symg);
            o22 = fuse_elim_diag(o22, r2, cm2, diag, fa.symg);
            o12 = fuse_elim_coupling(-fa.b * mv, r1, cm2, fa.symg);
            o21 = fuse_elim_coupling(-fa.b * mv, r2, cm1, fa.symg);
          }
          const int64_t ko = fuse_out_index(fa, s + q, node, jc, px, py), kc = fuse_out_index(fa, s + q, node, jc, px, py, true);
          if (ko >= 0) { fa.A11[ko] = o11; fa.A22[ko] = o22; }
          if (kc >= 0) {
            if (fa.A12) fa.A12[kc] = o12;
            if (fa.A21) fa.A21[kc] = o21;
          }
          s11 += fabs(o11); s22 += fabs(o22);
          if (diag) { d11 = o11; d22 = o22; }
        }
      const double i1 = (d11 != 0.0) ? 1.0 / d11 : 1.0, i2 = (d22 != 0.0) ? 1.0 / d22 : 1.0;
      fa.dinv1[node] = i1;
      fa.dinv2[node] = i2;
      const double q1 = s11 * fabs(i1), q2 = s22 * fabs(i2);
      if (!(r1 & 2)) best1 = q1 > best1 ? q1 : best1;   // (ghost rows: not rows of this rank's operator)
      if (!(r2 & 2)) best2 = q2 > best2 ? q2 : best2;
      if (fa.rhs) {
        fa.rhs[node] = (!near || r1 != 0) ? 0.0 : -(fa.a * lK1 + fa.b * lM);
        fa.rhs[n + node] = (!near || r2 != 0) ? 0.0 : -(fa.c * lK2 - fa.b * lM);
        fa.u0[node] = near ? fa.g1[node] : 0.0;
        fa.u0[n + node] = near ? fa.g2[node] : 0.0;
      }
    }
  }
  if constexpr (FUSED) {
    fuse_lam_max(best1, best2, fa.lam);
  }
}

int pph_launch_assemble_KM(pph_ctx* ctx_, MeshData& mesh) {
  PPH_TRY(pph_ensure_pattern(ctx_, mesh));
  pph_ctx* ctx = ctx_;
  PPH_TRY(mesh.K.alloc(ctx, (size_t)mesh.nnzb));
  PPH_TRY(mesh.M.alloc(ctx, (size_t)mesh.nnzb));
  const int64_t ncell = mesh.ncell;
  const bool multilinear = (mesh.kind == PPH_CELL_QUAD || mesh.kind == PPH_CELL_HEX);
  if (multilinear && ctx->asm_kernel == 2) {
    // two-pass: element rows to a buffer (Jacobians shared through LDS), then node-centred gather
    PPH_TRY(mesh.erows.alloc(ctx, (size_t)ncell * mesh.m * 2 * mesh.m));
    const int cpb = 256 / mesh.m;
    int64_t nb1 = ceil_div64(ncell, cpb), nb2 = ceil_div64(mesh.n, cpb);
    int g1 = (int)(nb1 < 256 * 16 ? nb1 : 256 * 16), g2 = (int)(nb2 < 256 * 16 ? nb2 : 256 * 16);
    const ERing ec{0, ncell, ncell, 1}, en{0, mesh.n, ncell, 1};   // whole mesh in one launch, plain addressing
    if (mesh.kind == PPH_CELL_QUAD) {
      hipLaunchKernelGGL(k_elem_rows<2>, dim3(g1), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p, mesh.cy.p,
                         mesh.cz.p, mesh.erows.p, ncell, ec);
      if (mesh.px >= 3 && mesh.py >= 3)
        hipLaunchKernelGGL((k_gather_rows<2, false, true>), dim3(g2), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.erows.p,
                           mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, mesh.nx, mesh.ny, 0, mesh.px, mesh.py, mesh.n, FuseArgs{}, en);
      else
        hipLaunchKernelGGL((k_gather_rows<2, false, false>), dim3(g2), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.erows.p,
                           mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, mesh.nx, mesh.ny, 0, mesh.px, mesh.py, mesh.n, FuseArgs{}, en);
    } else {
      hipLaunchKernelGGL(k_elem_rows<3>, dim3(g1), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p, mesh.cy.p,
                         mesh.cz.p, mesh.erows.p, ncell, ec);
      if (mesh.px >= 3 && mesh.py >= 3 && mesh.nzl >= 2)
        hipLaunchKernelGGL((k_gather_rows<3, false, true>), dim3(g2), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.erows.p,
                           mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, mesh.nx, mesh.ny, mesh.nzl, mesh.px, mesh.py,
                           mesh.n, FuseArgs{}, en);
      else
        hipLaunchKernelGGL((k_gather_rows<3, false, false>), dim3(g2), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.erows.p,
                           mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, mesh.nx, mesh.ny, mesh.nzl, mesh.px, mesh.py,
                           mesh.n, FuseArgs{}, en);
    }
    PPH_HIP(ctx, hipGetLastError());
    return PPH_OK;
  }
  if (multilinear && ctx->asm_kernel == 1) {
    // node-centred gather: writes every entry once, no memset, no atomics
    const int npb = 256 / mesh.m;
    int64_t nbatch = ceil_div64(mesh.n, npb);
    int grid = (int)(nbatch < 256 * 16 ? nbatch : 256 * 16);
    if (mesh.kind == PPH_CELL_QUAD)
      hipLaunchKernelGGL(k_asm_gather<2>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p, mesh.cy.p,
                         mesh.cz.p, mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, mesh.nx, mesh.ny, 0, mesh.px, mesh.py,
                         mesh.n);
    else
      hipLaunchKernelGGL(k_asm_gather<3>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p, mesh.cy.p,
                         mesh.cz.p, mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, mesh.nx, mesh.ny, mesh.nzl, mesh.px,
                         mesh.py, mesh.n);
    PPH_HIP(ctx, hipGetLastError());
    return PPH_OK;
  }
  if (!multilinear && ctx->asm_kernel != 0) {
    // simplices: node-centred gather (deterministic)
    int64_t nb = ceil_div64(mesh.n, 256);
    int grid = (int)(nb < 256 * 32 ? nb : 256 * 32);
    if (mesh.kind == PPH_CELL_TRI)
      hipLaunchKernelGGL(k_asm_simplex_gather<2>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p,
                         mesh.cy.p, mesh.cz.p, mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, mesh.nx, mesh.ny, 0, mesh.px,
                         mesh.py, mesh.n, FuseArgs{});
    else
      hipLaunchKernelGGL(k_asm_simplex_gather<3>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p,
                         mesh.cy.p, mesh.cz.p, mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, mesh.nx, mesh.ny, mesh.nzl,
                         mesh.px, mesh.py, mesh.n, FuseArgs{});
    PPH_HIP(ctx, hipGetLastError());
    return PPH_OK;
  }
  PPH_HIP(ctx, hipMemsetAsync(mesh.K.p, 0, sizeof(double) * mesh.nnzb, ctx->stream));
  PPH_HIP(ctx, hipMemsetAsync(mesh.M.p, 0, sizeof(double) * mesh.nnzb, ctx->stream));
  if (multilinear) {
    const int cpb = 256 / mesh.m;
    int64_t nbatch = ceil_div64(ncell, cpb);
    int grid = (int)(nbatch < 256 * 16 ? nbatch : 256 * 16);
    if (mesh.kind == PPH_CELL_QUAD)
      hipLaunchKernelGGL(k_asm_multilinear<2>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p,
                         mesh.cy.p, mesh.cz.p, mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, ncell);
    else
      hipLaunchKernelGGL(k_asm_multilinear<3>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p,
                         mesh.cy.p, mesh.cz.p, mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, ncell);
  } else {
    int64_t nthreads = ncell * 4;
    int64_t nb = ceil_div64(nthreads, 256);
    int grid = (int)(nb < 256 * 16 ? nb : 256 * 16);
    if (mesh.kind == PPH_CELL_TRI)
      hipLaunchKernelGGL(k_asm_simplex<2>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p,
                         mesh.cy.p, mesh.cz.p, mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, ncell);
    else
      hipLaunchKernelGGL(k_asm_simplex<3>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p,
                         mesh.cy.p, mesh.cz.p, mesh.rowptr.p, mesh.col.p, mesh.K.p, mesh.M.p, ncell);
  }
  PPH_HIP(ctx, hipGetLastError());
  return PPH_OK;
}

// ------------------------------------------------------------------------------------------------
// K-BC: lifted right-hand side and Dirichlet-eliminated blocks (8 lanes per row)
// mask bits: 1 = Dirichlet dof, 2 = ghost dof (row owned by the neighbouring slab)
// ------------------------------------------------------------------------------------------------
#define BC_LANES 8

// rownear[row] = 1 when the row is constrained / ghost in either field or one of its columns is constrained:
// only those rows need the masks in k_blocks and have a non-zero lifting term.  Depends on the pattern and the
// Dirichlet sets only, so it is rebuilt when they change, not per assembly.
__global__ __launch_bounds__(256) void k_row_near(Stencil st, int px, int py, int pz, const uint8_t* __restrict__ m1,
                                                  const uint8_t* __restrict__ m2, int64_t n, uint8_t* __restrict__ rownear) {
  // (walks the stencil of the row's node: the columns of its CSR row, without the CSR pattern)
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(row % px);
    const int64_t t = row / px;
    const int j = (int)(t % py), k = (int)(t / py);
    int f = (m1[row] | m2[row]) != 0;
    for (int q = 0; q < st.count; ++q) {
      const int ii = i + st.d[q][0], jj = j + st.d[q][1], kk = k + st.d[q][2];
      if (ii < 0 || ii >= px || jj < 0 || jj >= py || kk < 0 || kk >= pz) continue;
      const int64_t c = ii + (int64_t)px * (jj + (int64_t)py * kk);
      f |= ((m1[c] | m2[c]) & 1) != 0;
    }
    rownear[row] = (uint8_t)f;
  }
}

__global__ __launch_bounds__(256) void k_lift_rhs(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  const double* __restrict__ K, const double* __restrict__ M,
                                                  const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2,
                                                  const double* __restrict__ g1, const double* __restrict__ g2,
                                                  double a, double b, double c, int64_t n,
                                                  const uint8_t* __restrict__ rownear,
                                                  double* __restrict__ rhs, double* __restrict__ u0) {
  const int sub = threadIdx.x % BC_LANES;
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / BC_LANES; row < n;
       row += ((int64_t)gridDim.x * blockDim.x) / BC_LANES) {
    if (!rownear[row]) {   // no constrained column: u0 vanishes on the whole row, nothing to lift
      if (sub == 0) { rhs[row] = 0.0; rhs[n + row] = 0.0; u0[row] = 0.0; u0[n + row] = 0.0; }
      continue;
    }
    double sK1 = 0, sK2 = 0, sM = 0;
    for (int64_t k = rowptr[row] + sub; k < rowptr[row + 1]; k += BC_LANES) {
      const int32_t j = col[k];
      const double v1 = g1[j], v2 = g2[j];
      const double kk = K[k], mm = M[k];
      sK1 += kk * v1; sK2 += kk * v2; sM += mm * (v1 - v2);
    }
#pragma unroll
    for (int o = BC_LANES / 2; o > 0; o >>= 1) {
      sK1 += __shfl_down(sK1, o, BC_LANES);
      sK2 += __shfl_down(sK2, o, BC_LANES);
      sM += __shfl_down(sM, o, BC_LANES);
    }
    if (sub == 0) {
      rhs[row] = (m1[row] != 0) ? 0.0 : -(a * sK1 + b * sM);
      rhs[n + row] = (m2[row] != 0) ? 0.0 : -(c * sK2 - b * sM);
      u0[row] = g1[row];
      u0[n + row] = g2[row];
    }
  }
}

// 8 lanes per row, every lane owns 4 consecutive entries of the row range rounded down to a multiple of 4 (the
// layout of the SpMV kernel): 16-byte aligned loads of K, M, col and 16-byte aligned stores of the blocks;
// entries of neighbouring rows that share a 16/32-byte unit are written by neighbouring lane groups of the same
// wave instruction.  A21 == A12 when both fields carry the same Dirichlet set: then A21 is not written (null).
__global__ __launch_bounds__(256) void k_blocks(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                const double* __restrict__ K, const double* __restrict__ M,
                                                const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2,
                                                double a, double b, double c, int64_t n,
                                                const uint8_t* __restrict__ rownear, double* __restrict__ A11,
                                                double* __restrict__ A22, double* __restrict__ A12,
                                                double* __restrict__ A21) {
  const int sub = threadIdx.x % BC_LANES;
  const bool same = (A21 == nullptr);   // both fields carry the same Dirichlet set: one mask gather per entry
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / BC_LANES; row < n;
       row += ((int64_t)gridDim.x * blockDim.x) / BC_LANES) {
    const int64_t s = rowptr[row], e = rowptr[row + 1];
    if (!rownear[row]) {
      // interior row without a constrained column (almost all rows): pure streaming, no column or mask reads
      for (int64_t base = (s & ~(int64_t)3) + 4 * sub; base < e; base += 4 * BC_LANES) {
        const double2 k01 = *reinterpret_cast<const double2*>(K + base), k23 = *reinterpret_cast<const double2*>(K + base + 2);
        const double2 q01 = *reinterpret_cast<const double2*>(M + base), q23 = *reinterpret_cast<const double2*>(M + base + 2);
        const double kk4[4] = {k01.x, k01.y, k23.x, k23.y};
        const double mm4[4] = {q01.x, q01.y, q23.x, q23.y};
        if (base >= s && base + 3 < e) {
          *reinterpret_cast<double2*>(A11 + base) = make_double2(a * kk4[0] + b * mm4[0], a * kk4[1] + b * mm4[1]);
          *reinterpret_cast<double2*>(A11 + base + 2) = make_double2(a * kk4[2] + b * mm4[2], a * kk4[3] + b * mm4[3]);
          *reinterpret_cast<double2*>(A22 + base) = make_double2(c * kk4[0] + b * mm4[0], c * kk4[1] + b * mm4[1]);
          *reinterpret_cast<double2*>(A22 + base + 2) = make_double2(c * kk4[2] + b * mm4[2], c * kk4[3] + b * mm4[3]);
          *reinterpret_cast<double2*>(A12 + base) = make_double2(-b * mm4[0], -b * mm4[1]);
          *reinterpret_cast<double2*>(A12 + base + 2) = make_double2(-b * mm4[2], -b * mm4[3]);
          if (A21) {
            *reinterpret_cast<double2*>(A21 + base) = make_double2(-b * mm4[0], -b * mm4[1]);
            *reinterpret_cast<double2*>(A21 + base + 2) = make_double2(-b * mm4[2], -b * mm4[3]);
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (base + t >= s && base + t < e) {
              A11[base + t] = a * kk4[t] + b * mm4[t];
              A22[base + t] = c * kk4[t] + b * mm4[t];
              A12[base + t] = -b * mm4[t];
              if (A21) A21[base + t] = -b * mm4[t];
            }
        }
      }
      continue;
    }
    const uint8_t r1 = m1[row], r2 = m2[row];
    for (int64_t base = (s & ~(int64_t)3) + 4 * sub; base < e; base += 4 * BC_LANES) {
      const int4 cj = *reinterpret_cast<const int4*>(col + base);
      const double2 k01 = *reinterpret_cast<const double2*>(K + base), k23 = *reinterpret_cast<const double2*>(K + base + 2);
      const double2 q01 = *reinterpret_cast<const double2*>(M + base), q23 = *reinterpret_cast<const double2*>(M + base + 2);
      const int32_t jj[4] = {cj.x, cj.y, cj.z, cj.w};
      const double kk4[4] = {k01.x, k01.y, k23.x, k23.y};
      const double mm4[4] = {q01.x, q01.y, q23.x, q23.y};
      double o11[4], o22[4], o12[4], o21[4];
      bool in[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        in[t] = base + t >= s && base + t < e;
        const int32_t j = in[t] ? jj[t] : 0;
        const double kk = kk4[t], mm = mm4[t];
        const bool diag = (j == (int32_t)row);
        const bool c1 = (m1[j] & 1) != 0, c2 = same ? c1 : (m2[j] & 1) != 0;
        // ghost rows (bit 1) belong to the neighbouring slab: empty here; Dirichlet rows: identity
        o11[t] = (r1 & 2) ? 0.0 : (r1 & 1) ? (diag ? 1.0 : 0.0) : (c1 ? 0.0 : a * kk + b * mm);
        o22[t] = (r2 & 2) ? 0.0 : (r2 & 1) ? (diag ? 1.0 : 0.0) : (c2 ? 0.0 : c * kk + b * mm);
        o12[t] = (r1 != 0 || c2) ? 0.0 : -b * mm;
        o21[t] = (r2 != 0 || c1) ? 0.0 : -b * mm;
      }
      if (in[0] && in[1] && in[2] && in[3]) {
        *reinterpret_cast<double2*>(A11 + base) = make_double2(o11[0], o11[1]);
        *reinterpret_cast<double2*>(A11 + base + 2) = make_double2(o11[2], o11[3]);
        *reinterpret_cast<double2*>(A22 + base) = make_double2(o22[0], o22[1]);
        *reinterpret_cast<double2*>(A22 + base + 2) = make_double2(o22[2], o22[3]);
        *reinterpret_cast<double2*>(A12 + base) = make_double2(o12[0], o12[1]);
        *reinterpret_cast<double2*>(A12 + base + 2) = make_double2(o12[2], o12[3]);
        if (A21) {
          *reinterpret_cast<double2*>(A21 + base) = make_double2(o21[0], o21[1]);
          *reinterpret_cast<double2*>(A21 + base + 2) = make_double2(o21[2], o21[3]);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (in[t]) {
            A11[base + t] = o11[t];
            A22[base + t] = o22[t];
            A12[base + t] = o12[t];
            if (A21) A21[base + t] = o21[t];
          }
      }
    }
  }
}

__global__ void k_mask_diff(const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2, int64_t n,
                            int* __restrict__ out) {
  int d = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d |= (m1[i] != m2[i]) ? 1 : 0;
  if (d) atomicOr(out, 1);
}

// monolithic field-major CSR: row i < n = [A11 row | A12 row], row n+i = [A21 row | A22 row]
__global__ __launch_bounds__(256) void k_mono_rowptr(const int64_t* __restrict__ rowptr, int64_t n, int64_t nnzb,
                                                     int64_t* __restrict__ mrowptr) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i <= n;
       i += (int64_t)gridDim.x * blockDim.x) {
    mrowptr[i] = 2 * rowptr[i];
    mrowptr[n + i] = 2 * nnzb + 2 * rowptr[i];
  }
}

__global__ __launch_bounds__(256) void k_mono_fill(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const double* __restrict__ A11, const double* __restrict__ A22,
                                                   const double* __restrict__ A12, const double* __restrict__ A21,
                                                   int64_t n, int64_t nnzb, int32_t* __restrict__ mcol,
                                                   double* __restrict__ mval) {
  const int sub = threadIdx.x % BC_LANES;
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / BC_LANES; row < n;
       row += ((int64_t)gridDim.x * blockDim.x) / BC_LANES) {
    const int64_t s = rowptr[row], e = rowptr[row + 1], len = e - s;
    const int64_t o1 = 2 * s, o2 = 2 * nnzb + 2 * s;
    for (int64_t k = s + sub; k < e; k += BC_LANES) {
      const int32_t j = col[k];
      const int64_t q = k - s;
      mcol[o1 + q] = j;            mval[o1 + q] = A11[k];
      mcol[o1 + len + q] = j + (int32_t)n; mval[o1 + len + q] = A12[k];
      mcol[o2 + q] = j;            mval[o2 + q] = A21[k];
      mcol[o2 + len + q] = j + (int32_t)n; mval[o2 + len + q] = A22[k];
    }
  }
}

// buffers of the eliminated system and everything that depends on the Dirichlet sets only
static int blocks_alloc_csr(pph_ctx* ctx) {
  const int64_t nnzb = ctx->nnzb;
  PPH_TRY(ctx->A11.alloc(ctx, (size_t)nnzb));
  PPH_TRY(ctx->A22.alloc(ctx, (size_t)nnzb));
  PPH_TRY(ctx->A12.alloc(ctx, (size_t)nnzb));
  if (ctx->a21_alias) ctx->A21.release();
  else PPH_TRY(ctx->A21.alloc(ctx, (size_t)nnzb));
  return PPH_OK;
}

static int blocks_alloc_sell(pph_ctx* ctx) {
  // diagonal blocks: symmetric after the symmetric elimination; coupling blocks: A21 = A12^T, and A12 itself is
  // symmetric only when both fields carry the same Dirichlet set (then A21 is not stored at all)
  const int sym = pph_sell_sym(ctx), sym_c = (sym && ctx->a21_alias) ? 1 : 0;
  PPH_TRY(sell_alloc(ctx, ctx->mesh, ctx->E11, &ctx->S11, sym));
  PPH_TRY(sell_alloc(ctx, ctx->mesh, ctx->E22, &ctx->S22, sym));
  PPH_TRY(sell_alloc(ctx, ctx->mesh, ctx->E12, &ctx->S12, sym_c));
  if (ctx->a21_alias) { ctx->E21.release(); ctx->S21 = ctx->S12; }
  else PPH_TRY(sell_alloc(ctx, ctx->mesh, ctx->E21, &ctx->S21, 0));
  return PPH_OK;
}

static int blocks_prepare(pph_ctx* ctx, bool want_csr) {
  const int64_t n = ctx->n;
  // same Dirichlet set on both fields (every reference configuration): A21 == A12, stored once
  if (ctx->bc_dirty) {
    int* flag = reinterpret_cast<int*>(ctx->scal.p + (PPH_MAX_SCAL - 200));
    int h = 0;
    PPH_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_mask_diff, dim3(1024), dim3(256), 0, ctx->stream, ctx->bcmask[0].p, ctx->bcmask[1].p, n, flag);
    PPH_HIP(ctx, hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->a21_alias = (h == 0);
    PPH_TRY(ctx->rownear.alloc(ctx, (size_t)n));
    pph_launch_row_near(ctx, ctx->mesh, ctx->bcmask[0].p, ctx->bcmask[1].p, ctx->rownear.p);
    ctx->bc_dirty = false;
  }
  if (want_csr) { PPH_TRY(pph_ensure_pattern(ctx, ctx->mesh)); PPH_TRY(blocks_alloc_csr(ctx)); }
  PPH_TRY(ctx->rhs.alloc(ctx, (size_t)(2 * n)));
  PPH_TRY(ctx->u0.alloc(ctx, (size_t)(2 * n)));
  PPH_TRY(ctx->sol.alloc(ctx, (size_t)(2 * n)));
  return PPH_OK;
}

static int blocks_mono(pph_ctx* ctx, int monolithic) {
  const int64_t n = ctx->n, nnzb = ctx->nnzb;
  int64_t nb = ceil_div64(n * BC_LANES, 256);
  int grid = (int)(nb < 256 * 16 ? nb : 256 * 16);
  ctx->mono_ok = false;
  if (monolithic) {
    PPH_TRY(pph_ensure_csr_blocks(ctx));
    PPH_REQUIRE(ctx, 4 * nnzb < (int64_t)2147483647 * 4 && 2 * n < (int64_t)2147483647,
                "monolithic system too large for int32 columns");
    PPH_TRY(ctx->mrowptr.alloc(ctx, (size_t)(2 * n + 1)));
    PPH_TRY(ctx->mcol.alloc(ctx, (size_t)(4 * nnzb)));
    PPH_TRY(ctx->mval.alloc(ctx, (size_t)(4 * nnzb)));
    int g2 = (int)(ceil_div64(n + 1, 256) < 4096 ? ceil_div64(n + 1, 256) : 4096);
    hipLaunchKernelGGL(k_mono_rowptr, dim3(g2), dim3(256), 0, ctx->stream, ctx->mesh.rowptr.p, n, nnzb, ctx->mrowptr.p);
    hipLaunchKernelGGL(k_mono_fill, dim3(grid), dim3(256), 0, ctx->stream, ctx->mesh.rowptr.p, ctx->mesh.col.p, ctx->A11.p,
                       ctx->A22.p, ctx->A12.p, ctx->a21_alias ? ctx->A12.p : ctx->A21.p, n, nnzb, ctx->mcol.p, ctx->mval.p);
    PPH_HIP(ctx, hipGetLastError());
    ctx->mono_ok = true;
  }
  return PPH_OK;
}

// CSR values of the blocks from their stencil-ELL copies (export, monolithic CSR, Jacobi / 2x2-block preconditioners)
int pph_ensure_csr_blocks(pph_ctx* ctx) {
  if (ctx->csr_ok) return PPH_OK;
  PPH_REQUIRE(ctx, ctx->ell_ok, "blocks not assembled");
  PPH_TRY(pph_ensure_pattern(ctx, ctx->mesh));
  PPH_TRY(blocks_alloc_csr(ctx));
  PPH_TRY(sell_to_csr(ctx, ctx->mesh, ctx->S11, ctx->A11.p));
  PPH_TRY(sell_to_csr(ctx, ctx->mesh, ctx->S22, ctx->A22.p));
  PPH_TRY(sell_to_csr(ctx, ctx->mesh, ctx->S12, ctx->A12.p));
  if (!ctx->a21_alias) PPH_TRY(sell_to_csr(ctx, ctx->mesh, ctx->S21, ctx->A21.p));
  ctx->csr_ok = true;
  return PPH_OK;
}

int pph_launch_blocks(pph_ctx* ctx, int monolithic) {
  const int64_t n = ctx->n;
  PPH_TRY(blocks_prepare(ctx, true));
  int64_t nb = ceil_div64(n * BC_LANES, 256);
  int grid = (int)(nb < 256 * 16 ? nb : 256 * 16);
  hipLaunchKernelGGL(k_lift_rhs, dim3(grid), dim3(256), 0, ctx->stream, ctx->mesh.rowptr.p, ctx->mesh.col.p, ctx->mesh.K.p,
                     ctx->mesh.M.p, ctx->bcmask[0].p, ctx->bcmask[1].p, ctx->g[0].p, ctx->g[1].p, ctx->a, ctx->b, ctx->c,
                     n, ctx->rownear.p, ctx->rhs.p, ctx->u0.p);
  hipLaunchKernelGGL(k_blocks, dim3(grid), dim3(256), 0, ctx->stream, ctx->mesh.rowptr.p, ctx->mesh.col.p, ctx->mesh.K.p, ctx->mesh.M.p,
                     ctx->bcmask[0].p, ctx->bcmask[1].p, ctx->a, ctx->b, ctx->c, n, ctx->rownear.p, ctx->A11.p, ctx->A22.p,
                     ctx->A12.p, ctx->a21_alias ? nullptr : ctx->A21.p);
  PPH_HIP(ctx, hipGetLastError());
  ctx->diag0_valid = false;
  ctx->csr_ok = true;
  ctx->ell_ok = false;
  if (ctx->op_format == 1) {
    // the block solves run on stencil-ELL copies (CSR rows of ghost nodes are empty: symmetric storage only without slabs)
    const int sym = pph_sell_sym_from_csr(ctx), sym_c = (sym && ctx->a21_alias) ? 1 : 0;
    PPH_TRY(sell_from_csr(ctx, ctx->mesh, ctx->A11.p, ctx->E11, &ctx->S11, sym));
    PPH_TRY(sell_from_csr(ctx, ctx->mesh, ctx->A22.p, ctx->E22, &ctx->S22, sym));
    PPH_TRY(sell_from_csr(ctx, ctx->mesh, ctx->A12.p, ctx->E12, &ctx->S12, sym_c));
    if (ctx->a21_alias) { ctx->E21.release(); ctx->S21 = ctx->S12; }
    else PPH_TRY(sell_from_csr(ctx, ctx->mesh, ctx->A21.p, ctx->E21, &ctx->S21, 0));
    ctx->ell_ok = true;
  }
  return blocks_mono(ctx, monolithic);
}

// ------------------------------------------------------------------------------------------------
// Single-pass fused assembly for multilinear cells ("tile" kernel, default): no element-row buffer at all.
//
// A workgroup owns a tile of TX x TY x TZ nodes (8 x 4 x 2 hexahedral, 16 x 8 quadrilateral nodes).
//   A. geometry factors of the (TX+1)(TY+1)(TZ+1) cells touching the tile, from vertex coordinates staged in LDS:
//      D = |det J| J^-1 J^-T (symmetric DIM x DIM) + the weight |det J| (7 doubles instead of the 8 x 8 x 3 physical
//      gradients).  A cell whose parallel edges are equal vectors has a constant Jacobian: one lane per CELL stores
//      its factor once; any other cell takes the general pass, lane (cell, Gauss point q);
//   B. lane = node, wave = corner c (kept in a scalar register): row c of K_e and M_e of the node's incident cell c,
//      K_e[a][b] = sum_q dN_q[a]^T D_q dN_q[b],  M_e[a][b] = sum_q |det J_q| N_q[a] N_q[b]
//      - for a constant factor through the reference-element tables TileRef (6 + 1 multiply-adds per entry) -
//      rows -> LDS as [corner][column][node] (over the factors, which are dead by then; conflict-free);
//   C. lane = node, wave = slot group g: the slots g, g + 2^d, ... of the node's stencil row; per slot the entries of
//      the incident cells' rows that belong to its column are summed in a fixed order (the candidate list of a slot
//      is wave-uniform) and the fused epilogue of k_gather_rows applied (Dirichlet elimination, DPP blocks, lifting,
//      1 / a_ii, spectral bound) before the only global stores; the groups' row sums meet in LDS and the group that
//      holds the diagonal slot finishes the row.
// Every element row is formed exactly once; the Jacobian work is repeated for the
// cells on tile faces (2.1 evaluations per cell on average instead of 1), which is what removing the 17 GB
// element-row round trip of the two-pass kernels costs.  Deterministic, no atomics.  Vertices are addressed in
// closed form (vertex v of cell (ci,cj,ck) = node (ci + v&1, cj + (v>>1)&1, ck + (v>>2)&1), the cell->dof map of
// k_dofmap); the two-pass kernels (asm_tile 0) read the map itself.
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr double tile_fac(int bbit, int qbit) {   // 0.5 (1 + s_b xi_q), xi = +-1/sqrt(3)
  return (bbit == qbit) ? 0.78867513459481288225 : 0.21132486540518711775;
}
template <int DIM>
__host__ __device__ constexpr double tile_N(int q, int b) {
  double v = 1.0;
  for (int f = 0; f < DIM; ++f) v *= tile_fac((b >> f) & 1, (q >> f) & 1);
  return v;
}
template <int DIM>
__host__ __device__ constexpr double tile_dN(int q, int b, int e) {
  double v = ((b >> e) & 1) ? 0.5 : -0.5;
  for (int f = 0; f < DIM; ++f)
    if (f != e) v *= tile_fac((b >> f) & 1, (q >> f) & 1);
  return v;
}

// Reference-element integrals for cells with a constant geometry factor D (equal parallel edges):
//   K_e[a][b] = sum_{e <= f} D[ef] G[a][b][ef],  G[a][b][ef] = sum_q dN_q[a][e] dN_q[b][f] (+ the transposed term, e != f)
//   M_e[a][b] = |det J| M[a][b],                 M[a][b]     = sum_q N_q[a] N_q[b]
// (ef in the packed order of tile_factor: 00 01 02 11 12 22 / 00 01 11)
template <int DIM>
struct TileRef {
  double G[1 << DIM][1 << DIM][DIM * (DIM + 1) / 2];
  double M[1 << DIM][1 << DIM];
  constexpr TileRef() : G(), M() {
    for (int a = 0; a < (1 << DIM); ++a)
      for (int b = 0; b < (1 << DIM); ++b) {
        int k = 0;
        for (int e = 0; e < DIM; ++e)
          for (int f = e; f < DIM; ++f) {
            double s = 0.0;
            for (int q = 0; q < (1 << DIM); ++q)
              s += tile_dN<DIM>(q, a, e) * tile_dN<DIM>(q, b, f) + ((e != f) ? tile_dN<DIM>(q, a, f) * tile_dN<DIM>(q, b, e) : 0.0);
            G[a][b][k++] = s;
          }
        double m = 0.0;
        for (int q = 0; q < (1 << DIM); ++q) m += tile_N<DIM>(q, a) * tile_N<DIM>(q, b);
        M[a][b] = m;
      }
  }
};
__device__ constexpr TileRef<2> kTileRef2 = TileRef<2>();
__device__ constexpr TileRef<3> kTileRef3 = TileRef<3>();
template <int DIM> __device__ __forceinline__ const TileRef<DIM>& tile_ref();
template <> __device__ __forceinline__ const TileRef<2>& tile_ref<2>() { return kTileRef2; }
template <> __device__ __forceinline__ const TileRef<3>& tile_ref<3>() { return kTileRef3; }

// geometry factor of one Gauss point from the Jacobian J[e][d] = sum_b dN[b][e] x_b[d]: packed symmetric
// D = |det J| J^-1 J^-T and the weight |det J| (the arithmetic of the general pass of k_asm_tile)
template <int DIM>
__device__ __forceinline__ void tile_factor(const double (&J)[DIM][DIM], double* out) {
  if constexpr (DIM == 2) {
    const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const double r = 1.0 / det;
    const double I00 = J[1][1] * r, I01 = -J[0][1] * r, I10 = -J[1][0] * r, I11 = J[0][0] * r;
    const double w = fabs(det);
    out[0] = w * (I00 * I00 + I10 * I10);
    out[1] = w * (I00 * I01 + I10 * I11);
    out[2] = w * (I01 * I01 + I11 * I11);
    out[3] = w;
  } else {
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double r = 1.0 / det;
    double I[3][3];
    I[0][0] = c00 * r;
    I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
    I[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
    I[1][0] = c01 * r;
    I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
    I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
    I[2][0] = c02 * r;
    I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
    I[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
    const double w = fabs(det);
    out[0] = w * (I[0][0] * I[0][0] + I[1][0] * I[1][0] + I[2][0] * I[2][0]);
    out[1] = w * (I[0][0] * I[0][1] + I[1][0] * I[1][1] + I[2][0] * I[2][1]);
    out[2] = w * (I[0][0] * I[0][2] + I[1][0] * I[1][2] + I[2][0] * I[2][2]);
    out[3] = w * (I[0][1] * I[0][1] + I[1][1] * I[1][1] + I[2][1] * I[2][1]);
    out[4] = w * (I[0][1] * I[0][2] + I[1][1] * I[1][2] + I[2][1] * I[2][2]);
    out[5] = w * (I[0][2] * I[0][2] + I[1][2] * I[1][2] + I[2][2] * I[2][2]);
    out[6] = w;
  }
}

template <int DIM> struct TileGeo;
template <> struct TileGeo<3> { static constexpr int TX = 8, TY = 4, TZ = 2; };
template <> struct TileGeo<2> { static constexpr int TX = 16, TY = 8, TZ = 1; };

template <int DIM>
__global__ __launch_bounds__(512, 4) void k_asm_tile(const double* __restrict__ cx, const double* __restrict__ cy,
                                                  const double* __restrict__ cz, const int64_t* __restrict__ rowptr,
                                                  double* __restrict__ K, double* __restrict__ M, int nx, int ny, int nzl,
                                                  int px, int py, int pz, int64_t n, FuseArgs fa, int probe, int xmap) {
  using TG = TileGeo<DIM>;
  constexpr int NB = 1 << DIM;
  constexpr int TX = TG::TX, TY = TG::TY, TZ = TG::TZ;
  constexpr int NT = TX * TY * TZ;                       // nodes of a tile (64 / 128)
  constexpr int CX = TX + 1, CY = TY + 1, CZ = (DIM == 3) ? TZ + 1 : 1;
  constexpr int NC = CX * CY * CZ;                       // cells touching the tile (135 / 153)
  constexpr int VX = TX + 2, VY = TY + 2, VZ = (DIM == 3) ? TZ + 2 : 1;
  constexpr int NV = VX * VY * VZ;                       // their vertices (240 / 180)
  constexpr int ND = DIM * (DIM + 1) / 2 + 1;            // doubles per (cell, q): packed symmetric D and |det J|
  constexpr int DSTR = NB * ND + 1;                      // cell stride of the factors (+1: off the bank period)
  static_assert(NT * NB == 512, "one lane per (node, corner)");
  // one LDS region, two lives: vertex coordinates + geometry factors (phases A, B), then the element rows K | M
  // (phase C) - 74 KB for hexahedra, so that two workgroups share a CU
  constexpr int NAB = NC * DSTR + NV * DIM;
  constexpr int NROW = NB * NB * NT;                     // entries of the element rows K (and M): [corner][column][node]
  constexpr int NUNI = (NAB > 2 * NROW) ? NAB : 2 * NROW;
  __shared__ double sU[NUNI];
  double (*const sXv)[DIM] = reinterpret_cast<double (*)[DIM]>(sU + NC * DSTR);
  __shared__ double sdN[NB][NB][DIM];
  __shared__ double sNq[NB][NB];
  __shared__ double sRed[2][NB][NT];                     // per-group partial row sums, combined by the diagonal's group
  __shared__ uint8_t sAff[NC];
  double* const sD = sU;
  double* const sK = sU;
  double* const sM = sU + NROW;
  const int tid = threadIdx.x;
  if (tid < NB * NB) {
    const int q = tid / NB, b = tid % NB;
    double nv = 1.0;
#pragma unroll
    for (int f = 0; f < DIM; ++f) nv *= tile_fac((b >> f) & 1, (q >> f) & 1);
    sNq[q][b] = nv;
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      double d = ((b >> e) & 1) ? 0.5 : -0.5;
#pragma unroll
      for (int f = 0; f < DIM; ++f)
        if (f != e) d *= tile_fac((b >> f) & 1, (q >> f) & 1);
      sdN[q][b][e] = d;
    }
  }
  // per stencil slot: the incident cells that hold both the row node (corner cc) and the column node (corner b) - per
  // direction d = 0 -> corner bit 0 or 1 for both, d = +1 -> (0, 1), d = -1 -> (1, 0): 8 / 4 / 2 / 1 cells for the
  // diagonal / a face / an edge / a corner neighbour, listed in ascending cc (the summation order of k_gather_rows) -
  // packed as 6-bit (cc | b << 3) fields, count in bits 48.., plus the flattened column offset of the slot
  constexpr int NSLOT = (DIM == 3) ? 27 : 9;
  __shared__ unsigned long long sCand[NSLOT];
  __shared__ long long sOff[NSLOT];
  if (tid < NSLOT) {
    const int dd[3] = {tid % 3 - 1, (tid / 3) % 3 - 1, (DIM == 3) ? tid / 9 - 1 : 0};
    unsigned long long pk = 0;
    int cnt = 0;
    for (int m = 0; m < NB; ++m) {
      int b = 0;
      bool okc = true;
      for (int a = 0; a < DIM; ++a) {
        const int bb = ((m >> a) & 1) + dd[a];
        okc = okc && (bb == 0 || bb == 1);
        b |= (bb & 1) << a;
      }
      if (okc) { pk |= (unsigned long long)(m | (b << 3)) << (6 * cnt); ++cnt; }
    }
    sCand[tid] = pk | ((unsigned long long)cnt << 48);
    sOff[tid] = (long long)dd[0] + (long long)dd[1] * px + (long long)dd[2] * (long long)px * py;
  }
  const int tiles_x = (px + TX - 1) / TX, tiles_y = (py + TY - 1) / TY, tiles_z = (DIM == 3) ? (pz + TZ - 1) / TZ : 1;
  const int ntiles = tiles_x * tiles_y * tiles_z;   // (launcher: fewer than 2^31 tiles; 32-bit, wave-uniform index arithmetic)
  const int64_t pxy = (int64_t)px * py;
  // this lane's node and corner inside the tile
  // lane mapping: corner / slot group c = tid / NT (uniform over a wave), node ln = tid % NT: consecutive lanes are
  // consecutive nodes in x - the LDS rows [corner][column][node] are read and written without bank conflicts, the
  // reference gradients of corner c and the candidate list of a slot are wave-uniform (scalar registers), and the
  // candidate loop of a slot runs exactly as long as that slot needs on every lane of the wave
  const int c = __builtin_amdgcn_readfirstlane(tid / NT), ln = tid % NT;   // (NT is a multiple of the wave size: uniform, kept in a scalar register)
  const int lx = ln % TX, ly = (ln / TX) % TY, lz = ln / (TX * TY);
  double best1 = 0.0, best2 = 0.0;
  static_assert(NV <= 512, "one vertex per lane");
  // vertices of a tile's cells: nodes (i0-1 .. i0+TX, j0-1 .. j0+TY, k0-1 .. k0+TZ), clamped into the box.  The
  // coordinates of the NEXT tile are requested while the current one is computed (register double buffer): the
  // load latency sat exposed in front of phase A with only two workgroups per CU to hide it
  double pv[3] = {0.0, 0.0, 0.0};
  auto fetch_vertex = [&](int tile) {
    if (tile >= ntiles || tid >= NV) return;
    const int tx = tile % tiles_x;
    const int tt = tile / tiles_x;
    const int ty = (int)(tt % tiles_y), tz = (int)(tt / tiles_y);
    const int vx = tid % VX, vy = (tid / VX) % VY, vz = tid / (VX * VY);
    int gi = tx * TX - 1 + vx, gj = ty * TY - 1 + vy, gk = (DIM == 3) ? tz * TZ - 1 + vz : 0;
    gi = gi < 0 ? 0 : (gi > px - 1 ? px - 1 : gi);
    gj = gj < 0 ? 0 : (gj > py - 1 ? py - 1 : gj);
    gk = gk < 0 ? 0 : (gk > pz - 1 ? pz - 1 : gk);
    const int64_t g = gi + (int64_t)px * gj + pxy * gk;
    pv[0] = cx[g];
    pv[1] = cy[g];
    if constexpr (DIM == 3) pv[2] = cz[g];
  };
  // Tile order.  A slot store of a tile is 8 runs of TX doubles (64 B, never aligned: px is odd on the usual 2^k cell
  // meshes): the two halves of a 128-B line belong to x-adjacent tiles.  Dealt round-robin (tile = blockIdx + k grid)
  // those tiles sit on different XCDs, whose L2s each write back a partial line - the memory side then reads the line
  // to merge it (PMC, 256^3: 4.4 GB fetched by a kernel with 0.5 GB of inputs, 1.5 x its stores written).  xmap: every
  // XCD (blockIdx % 8) takes one contiguous eighth of the tile sequence and its workgroups consecutive tiles of it, so
  // both halves of a line meet in ONE L2 within a few microseconds.
  const int xcd = blockIdx.x & 7, bxx = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;                                // tiles per XCD
  const int t_first = xmap ? xcd * tpx + bxx : (int)blockIdx.x;
  const int t_step = xmap ? bpx : (int)gridDim.x;
  const int t_end = xmap ? ((xcd + 1) * tpx < ntiles ? (xcd + 1) * tpx : ntiles) : ntiles;
  fetch_vertex(t_first < t_end ? t_first : ntiles);
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int tx = tile % tiles_x;
    const int tt = tile / tiles_x;
    const int ty = (int)(tt % tiles_y), tz = (int)(tt / tiles_y);
    const int i0 = tx * TX, j0 = ty * TY, k0 = tz * TZ;
    __syncthreads();   // previous tile's rows are consumed
    if (tid < NV) {
      sXv[tid][0] = pv[0];
      sXv[tid][1] = pv[1];
      if constexpr (DIM == 3) sXv[tid][2] = pv[2];
    }
    fetch_vertex(tile + t_step < t_end ? tile + t_step : ntiles);
    // operands of phase C that depend on the node only: requested now, used after three barriers
    uint8_t pnear = 0, pr1 = 0, pr2 = 0;
    {
      const int gi = i0 + lx, gj = j0 + ly, gk = (DIM == 3) ? k0 + lz : 0;
      if (gi < px && gj < py && gk < pz) {
        const int64_t nd = gi + (int64_t)px * gj + pxy * gk;
        pnear = fa.near[nd];
        pr1 = fa.m1[nd];
        pr2 = fa.m2[nd];
      }
    }
    __syncthreads();
    // ---- A: geometry factors.  A cell whose parallel edges are equal vectors (every cell of a structured box mesh:
    // its vertices share coordinate values exactly) has a constant Jacobian J = [e_x e_y e_z] / 2: one lane per CELL
    // computes the factor once and stores it for all Gauss points; the other cells take the general pass, lane (cell, q)
    int any_general = 0;
    for (int cl = tid; cl < NC; cl += 512) {
      const int ccx = cl % CX, ccy = (cl / CX) % CY, ccz = cl / (CX * CY);
      const int gi = i0 - 1 + ccx, gj = j0 - 1 + ccy, gk = (DIM == 3) ? k0 - 1 + ccz : 0;
      const bool incell = gi >= 0 && gi < nx && gj >= 0 && gj < ny && (DIM == 2 || (gk >= 0 && gk < nzl));
      uint8_t aff = 1;
      if (incell) {
        double X[NB][DIM];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int v = (ccx + (b & 1)) + VX * ((ccy + ((b >> 1) & 1)) + VY * ((DIM == 3) ? ccz + ((b >> 2) & 1) : 0));
#pragma unroll
          for (int d = 0; d < DIM; ++d) X[b][d] = sXv[v][d];
        }
        // edge vectors from vertex 0 along each reference direction e; affine iff every parallel edge equals it
        double E[DIM][DIM];   // E[e][d] = x_{2^e}[d] - x_0[d]
        bool ok = (probe != 6);   // (option asm_affine 0: every cell takes the general pass)
#pragma unroll
        for (int e = 0; e < DIM; ++e) {
#pragma unroll
          for (int d = 0; d < DIM; ++d) E[e][d] = X[1 << e][d] - X[0][d];
#pragma unroll
          for (int b = 1; b < NB; ++b) {
            if ((b >> e) & 1) continue;
#pragma unroll
            for (int d = 0; d < DIM; ++d) ok = ok && (X[b | (1 << e)][d] - X[b][d] == E[e][d]);
          }
        }
        if (ok) {
          // J[e][d] = dx_d / dxi_e = E[e][d] / 2  (the convention of the general pass: J[e][d] = sum_b dN[b][e] x_b[d])
          double J[DIM][DIM];
#pragma unroll
          for (int e = 0; e < DIM; ++e)
#pragma unroll
            for (int d = 0; d < DIM; ++d) J[e][d] = 0.5 * E[e][d];
          double o[ND];
          tile_factor<DIM>(J, o);
          double* out = sD + cl * DSTR;   // slot of Gauss point 0: phase B reads it for every q of such a cell
#pragma unroll
          for (int f = 0; f < ND; ++f) out[f] = o[f];
        } else {
          aff = 0;
          any_general = 1;
        }
      }
      sAff[cl] = aff;
    }
    any_general = __syncthreads_or(any_general);
    if (any_general) {
    for (int task = tid; task < NC * NB; task += 512) {
        const int cl = task / NB, q = task % NB;
        const int ccx = cl % CX, ccy = (cl / CX) % CY, ccz = cl / (CX * CY);
        const int gi = i0 - 1 + ccx, gj = j0 - 1 + ccy, gk = (DIM == 3) ? k0 - 1 + ccz : 0;
        const bool incell = gi >= 0 && gi < nx && gj >= 0 && gj < ny && (DIM == 2 || (gk >= 0 && gk < nzl));
        if (!incell || sAff[cl]) continue;
        double J[DIM][DIM];
#pragma unroll
        for (int e = 0; e < DIM; ++e)
#pragma unroll
          for (int d = 0; d < DIM; ++d) J[e][d] = 0.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int v = (ccx + (b & 1)) + VX * ((ccy + ((b >> 1) & 1)) + VY * ((DIM == 3) ? ccz + ((b >> 2) & 1) : 0));
#pragma unroll
          for (int e = 0; e < DIM; ++e)
#pragma unroll
            for (int d = 0; d < DIM; ++d) J[e][d] += sdN[q][b][e] * sXv[v][d];
        }
        double* out = sD + cl * DSTR + q * ND;
        if constexpr (DIM == 2) {
          const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
          const double r = 1.0 / det;
          // I = J^-1 in the convention of k_elem_rows: g[b][d] = sum_e I[d][e] dN[b][e]
          const double I00 = J[1][1] * r, I01 = -J[0][1] * r, I10 = -J[1][0] * r, I11 = J[0][0] * r;
          const double w = fabs(det);
          out[0] = w * (I00 * I00 + I10 * I10);
          out[1] = w * (I00 * I01 + I10 * I11);
          out[2] = w * (I01 * I01 + I11 * I11);
          out[3] = w;
        } else {
          const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
          const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
          const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
          const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
          const double r = 1.0 / det;
          double I[3][3];
          I[0][0] = c00 * r;
          I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
          I[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
          I[1][0] = c01 * r;
          I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
          I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
          I[2][0] = c02 * r;
          I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
          I[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
          const double w = fabs(det);
          // D[e][f] = w sum_d I[d][e] I[d][f], packed 00 01 02 11 12 22
          out[0] = w * (I[0][0] * I[0][0] + I[1][0] * I[1][0] + I[2][0] * I[2][0]);
          out[1] = w * (I[0][0] * I[0][1] + I[1][0] * I[1][1] + I[2][0] * I[2][1]);
          out[2] = w * (I[0][0] * I[0][2] + I[1][0] * I[1][2] + I[2][0] * I[2][2]);
          out[3] = w * (I[0][1] * I[0][1] + I[1][1] * I[1][1] + I[2][1] * I[2][1]);
          out[4] = w * (I[0][1] * I[0][2] + I[1][1] * I[1][2] + I[2][1] * I[2][2]);
          out[5] = w * (I[0][2] * I[0][2] + I[1][2] * I[1][2] + I[2][2] * I[2][2]);
          out[6] = w;
        }
      }
    }
    __syncthreads();
    if (probe == 1) continue;   // timing probe (asm_tile_probe): phase A only
    // ---- B: row c of the incident cell c of this lane's node
    const int gi = i0 + lx, gj = j0 + ly, gk = (DIM == 3) ? k0 + lz : 0;
    const bool innode = gi < px && gj < py && gk < pz;
    const int64_t node = gi + (int64_t)px * gj + pxy * gk;
    double Kr[NB], Mr[NB];
    bool rowok = false;
    {
      const int ci = gi - (c & 1), cj = gj - ((c >> 1) & 1), ck = (DIM == 3) ? gk - ((c >> 2) & 1) : 0;
      rowok = innode && ci >= 0 && ci < nx && cj >= 0 && cj < ny && (DIM == 2 || (ck >= 0 && ck < nzl));
#pragma unroll
      for (int b = 0; b < NB; ++b) { Kr[b] = 0.0; Mr[b] = 0.0; }
      if (rowok) {
        const int cl = (ci - (i0 - 1)) + CX * ((cj - (j0 - 1)) + CY * ((DIM == 3) ? ck - (k0 - 1) : 0));
        const double* dc = sD + cl * DSTR;
        if (sAff[cl]) {
          // constant factor: the sum over the Gauss points is a table of the reference element (row c: wave-uniform)
          const TileRef<DIM>& R = tile_ref<DIM>();
          double D[ND];
#pragma unroll
          for (int f = 0; f < ND; ++f) D[f] = dc[f];
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            double kb = 0.0;
#pragma unroll
            for (int f = 0; f < ND - 1; ++f) kb += D[f] * R.G[c][b][f];
            Kr[b] = kb;
            Mr[b] = D[ND - 1] * R.M[c][b];
          }
        } else
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const double* dq = dc + q * ND;
          double t[DIM];
          if constexpr (DIM == 2) {
            const double g0 = sdN[q][c][0], g1 = sdN[q][c][1];
            t[0] = g0 * dq[0] + g1 * dq[1];
            t[1] = g0 * dq[1] + g1 * dq[2];
          } else {
            const double g0 = sdN[q][c][0], g1 = sdN[q][c][1], g2 = sdN[q][c][2];
            t[0] = g0 * dq[0] + g1 * dq[1] + g2 * dq[2];
            t[1] = g0 * dq[1] + g1 * dq[3] + g2 * dq[4];
            t[2] = g0 * dq[2] + g1 * dq[4] + g2 * dq[5];
          }
          const double sw = dq[ND - 1] * sNq[q][c];
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            double kb = Kr[b];
#pragma unroll
            for (int e = 0; e < DIM; ++e) kb += t[e] * tile_dN<DIM>(q, b, e);
            Kr[b] = kb;
            Mr[b] += sw * tile_N<DIM>(q, b);
          }
        }
      }
    }
    __syncthreads();   // every lane has read its factors: the rows may overwrite them
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      sK[(c * NB + b) * NT + ln] = Kr[b];
      sM[(c * NB + b) * NT + ln] = Mr[b];
    }
    // (the barrier also tells every lane whether the tile holds a node next to a Dirichlet dof: lifting sums needed)
    const int tile_near = __syncthreads_or((innode && pnear != 0) ? 1 : 0);
    if (probe == 2) continue;   // timing probe: phases A and B
    // ---- C: stencil row of the node; group c takes the slots c, c + NB, ... of all nodes of the tile; fused epilogue
    const bool near = innode && pnear != 0;
    const uint8_t r1 = near ? pr1 : 0, r2 = near ? pr2 : 0;
    double s11 = 0.0, s22 = 0.0, d11 = 0.0, d22 = 0.0, lK1 = 0.0, lK2 = 0.0, lM = 0.0;
    if (innode) {
      int64_t rp = 0;
      if (fa.ld == 0 || fa.keep_km) rp = rowptr[node];
      for (int slot = c; slot < ((probe == 5) ? 0 : NSLOT); slot += NB) {   // (timing probe 5: no slot loop at all)
        const int dx = slot % 3 - 1, dy = (slot / 3) % 3 - 1, dz = (DIM == 3) ? slot / 9 - 1 : 0;
        const int ni = gi + dx, nj = gj + dy, nk = gk + dz;
        if (ni < 0 || ni >= px || nj < 0 || nj >= py || nk < 0 || nk >= pz) continue;   // no such neighbour: pad / absent
        double kv = 0.0, mv = 0.0;
        const unsigned long long pk = sCand[slot];
        const int ncand = (probe == 4) ? 0 : (int)(pk >> 48);   // (timing probe 4: no gather)
        // rows of cells that do not exist are zero rows (phase B stores them as such): no validity test; the candidate
        // count is uniform over the wave and one of 1 / 2 / 4 / 8 - straight-line loads, all in flight together, summed
        // in ascending candidate order
        auto gather = [&](auto NC_) {
          constexpr int NCAND = decltype(NC_)::value;
          double kk[NCAND], mm[NCAND];
#pragma unroll
          for (int q = 0; q < NCAND; ++q) {
            const int e = (int)(pk >> (6 * q)) & 63, cc = e & 7, b = e >> 3;
            kk[q] = sK[(cc * NB + b) * NT + ln];
            mm[q] = sM[(cc * NB + b) * NT + ln];
          }
#pragma unroll
          for (int q = 0; q < NCAND; ++q) { kv += kk[q]; mv += mm[q]; }
        };
        switch (ncand) {
          case 1: gather(std::integral_constant<int, 1>()); break;
          case 2: gather(std::integral_constant<int, 2>()); break;
          case 4: gather(std::integral_constant<int, 4>()); break;
          case 8: gather(std::integral_constant<int, 8>()); break;
          default: break;
        }
        const int32_t j = (int32_t)(node + sOff[slot]);
        // position of the entry in the CSR row = number of existing neighbours in the slots before this one
        int64_t kcsr = 0;
        if (fa.ld == 0 || fa.keep_km) {
          int rank = 0;
          for (int s2 = 0; s2 < slot; ++s2) {
            const int ex = gi + s2 % 3 - 1, ey = gj + (s2 / 3) % 3 - 1, ez = gk + ((DIM == 3) ? s2 / 9 - 1 : 0);
            rank += (ex >= 0 && ex < px && ey >= 0 && ey < py && ez >= 0 && ez < pz) ? 1 : 0;
          }
          kcsr = rp + rank;
        }
        if (fa.keep_km) { K[kcsr] = kv; M[kcsr] = mv; }
        const bool diag = (slot == NSLOT / 2);
        double o11 = fa.a * kv + fa.b * mv, o22 = fa.c * kv + fa.b * mv, o12 = -fa.b * mv, o21 = o12;
        if (near) {
          const uint8_t cm1 = fa.m1[j], cm2 = fa.same ? cm1 : fa.m2[j];
          if (fa.rhs) {
            const double v1 = fa.g1[j], v2 = fa.g2[j];
            lK1 += kv * v1; lK2 += kv * v2; lM += mv * (v1 - v2);
          }
          o11 = fuse_elim_diag(o11, r1, cm1, diag, fa.symg);
          o22 = fuse_elim_diag(o22, r2, cm2, diag, fa.symg);
          o12 = fuse_elim_coupling(-fa.b * mv, r1, cm2, fa.symg);
          o21 = fuse_elim_coupling(-fa.b * mv, r2, cm1, fa.symg);
        }
        const int sq = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
        const int so = fa.slot_of[sq], sc = fa.slot_of_c[sq];     // stored slots (-1: lower half of a symmetric operator)
        const int64_t ko = fa.ld ? (so < 0 ? -1 : (int64_t)so * fa.ld + node) : kcsr;
        const int64_t kc = fa.ld ? (sc < 0 ? -1 : (int64_t)sc * fa.ld + node) : kcsr;
        if (probe != 3) {   // (timing probe 3: everything but the operator stores)
          if (ko >= 0) { fa.A11[ko] = o11; fa.A22[ko] = o22; }
          if (kc >= 0) {
            if (fa.A12) fa.A12[kc] = o12;
            if (fa.A21) fa.A21[kc] = o21;
          }
        }
        s11 += fabs(o11); s22 += fabs(o22);
        if (diag) { d11 = o11; d22 = o22; }
      }
    }
    // the groups' partial sums of a node meet in LDS; the group that holds the diagonal slot combines them in the order
    // of a shuffle tree over the groups (l[i] += l[i + NB/2], ..., l[0] += l[1]) and finishes the row
    constexpr int GD = (NSLOT / 2) % NB;
    auto combine = [&](const double (*part)[NT]) -> double {
      double l[NB];
#pragma unroll
      for (int g = 0; g < NB; ++g) l[g] = part[g][ln];
#pragma unroll
      for (int o = NB / 2; o > 0; o >>= 1)
#pragma unroll
        for (int g = 0; g < o; ++g) l[g] += l[g + o];
      return l[0];
    };
    sRed[0][c][ln] = s11;
    sRed[1][c][ln] = s22;
    __syncthreads();
    double i1 = 1.0, i2 = 1.0;
    if (c == GD && innode) {
      const double t11 = combine(sRed[0]), t22 = combine(sRed[1]);
      i1 = (d11 != 0.0) ? 1.0 / d11 : 1.0;
      i2 = (d22 != 0.0) ? 1.0 / d22 : 1.0;
      fa.dinv1[node] = i1;
      fa.dinv2[node] = i2;
      const double q1 = t11 * fabs(i1), q2 = t22 * fabs(i2);
      if (!(r1 & 2)) best1 = q1 > best1 ? q1 : best1;   // (ghost rows: not rows of this rank's operator)
      if (!(r2 & 2)) best2 = q2 > best2 ? q2 : best2;
    }
    if (fa.rhs) {
      if (tile_near) {
        // lifting sums of the rows next to Dirichlet dofs: the same meeting point, one quantity pair at a time
        __syncthreads();
        sRed[0][c][ln] = lK1;
        sRed[1][c][ln] = lK2;
        __syncthreads();
        double tK1 = 0.0, tK2 = 0.0;
        if (c == GD && innode) { tK1 = combine(sRed[0]); tK2 = combine(sRed[1]); }
        __syncthreads();
        sRed[0][c][ln] = lM;
        __syncthreads();
        if (c == GD && innode) {
          const double tM = combine(sRed[0]);
          fa.rhs[node] = (!near || r1 != 0) ? 0.0 : -(fa.a * tK1 + fa.b * tM);
          fa.rhs[n + node] = (!near || r2 != 0) ? 0.0 : -(fa.c * tK2 - fa.b * tM);
          fa.u0[node] = near ? fa.g1[node] : 0.0;
          fa.u0[n + node] = near ? fa.g2[node] : 0.0;
        }
      } else if (c == GD && innode) {
        fa.rhs[node] = 0.0;
        fa.rhs[n + node] = 0.0;
        fa.u0[node] = 0.0;
        fa.u0[n + node] = 0.0;
      }
    }
  }
  fuse_lam_max(best1, best2, fa.lam);
}

// ------------------------------------------------------------------------------------------------
// Node kernel, round 4 (k_asm_node2; default): the same integrals, organised so that a node-wave needs ~3 x fewer
// instructions.  What the round-3 kernel spent its 10 k instructions per node-wave on (ISA): 1 400 fp64 operations, 1 370
// v_readlane / v_writelane (scalar registers spilled into vector lanes: the ~150 64-bit literals of the reference tables
// G[a][b][ef] + the saved exec masks of 35 predicated regions), 760 selects, ~1 500 exec-mask / branch instructions, 81
// coordinate loads with clamped indices.  Three changes:
//
//  1. Edges instead of neighbourhoods.  Every cell has equal parallel edges (MeshData::all_affine: the exact test at
//     mesh build), so the Jacobian of the cell with the node as corner a is made of the node's OWN edges
//     E(d, +) = x(node + e_d) - x(node) (a_d = 0) or E(d, -) = x(node) - x(node - e_d) (a_d = 1) - bitwise the edges the
//     tile kernel reads off the cell's corner 0.  7 vertices (21 loads) instead of 27 (81).
//  2. Sums over cells BEFORE the reference integrals.  With b = a + o the reference integrals of trilinear functions on
//     a constant-Jacobian cell factor into one-dimensional ones ( int n_i n_j = 2/3 | 1/3, int n_i' n_j' = +-1/2,
//     int n_i' n_j = s_i / 2 ):
//        G[a][b][ee] = sigma_e(o) / 2 * prod_{d != e} mm_d(o)                       sigma = +1 (o_d = 0) | -1,  mm = 2/3 | 1/3
//        G[a][b][ef] = s_e(a) s_f(a) (sigma_e + sigma_f) / 4 * mm_h(o)  (e < f)     s_d(a) = +1 (a_d = 1) | -1
//        M[a][b]     = prod_d mm_d(o)
//     i.e. a coefficient that depends on the SLOT o only, times a sign that depends on the cell only.  Hence
//        K(node, node + o) = sum_ef c_ef(o) * [ sum over the cells c that hold both nodes of  s_e s_f D^c_ef ]
//     and the inner sums over the 2 x 2 x 2 cells around the node are box sums along the axes with o_d = 0: 19
//     additions per component give all 27 slots.  9 distinct literals instead of ~150; 480 (factors) + 133 (box sums)
//     + 153 (combination) fp64 operations instead of 1 400; the cell loop has no corner loop and no predicated region
//     (a cell outside the box gets the weight |det J| = 0, its edge replaced by the opposite one).
//  3. A wave whose 64 nodes are all interior and far from constrained dofs (~70 % of the waves of a 256^3 level) runs a
//     straight-line body: no masks, no existence tests, right-hand side 0.  Storage variants (full / symmetric,
//     coupling block symmetric or not, with or without A21 / right-hand side) are template parameters, not tables in
//     scalar registers.
// Entries agree with the tile kernel's to 1e-15 of the largest entry (closed-form reference integrals instead of
// Gauss sums evaluated at compile time, other association of the sums); same sweeps / iterations
// (test_node_assembly_kernel_equals_tile_kernel).  Rows of a uniform box repeat bit for bit as before: every node of a
// class runs the same arithmetic on the same edge vectors (row dictionaries, pph_sell.hip).
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr double n2_mm(int od) { return od == 0 ? 0.66666666666666663 : 0.33333333333333331; }
template <int DIM> __host__ __device__ constexpr int n2_o(int q, int d) { return d == 0 ? q % 3 - 1 : (d == 1 ? (q / 3) % 3 - 1 : (DIM == 3 ? q / 9 - 1 : 0)); }
// coefficient of the packed component k = (e, f) of D (order of tile_factor: 3D 00 01 02 11 12 22, 2D 00 01 11) in slot q
template <int DIM>
__host__ __device__ constexpr double n2_coef(int q, int e, int f) {
  const int oe = n2_o<DIM>(q, e), of = n2_o<DIM>(q, f);
  if (e == f) {
    double v = (oe == 0) ? 0.5 : -0.5;
    for (int d = 0; d < DIM; ++d) if (d != e) v *= n2_mm(n2_o<DIM>(q, d));
    return v;
  }
  if ((oe == 0) != (of == 0)) return 0.0;
  double v = (oe == 0) ? 0.5 : -0.5;
  for (int d = 0; d < DIM; ++d) if (d != e && d != f) v *= n2_mm(n2_o<DIM>(q, d));
  return v;
}
template <int DIM>
__host__ __device__ constexpr double n2_cmass(int q) {
  double v = 1.0;
  for (int d = 0; d < DIM; ++d) v *= n2_mm(n2_o<DIM>(q, d));
  return v;
}
// box sums of the 2^DIM cell values Q[a] (a_d = 1: the cell lies on the low side of the node along d): out[q] = sum of
// Q[a] over the cells that hold both the node and node + o(q)  (o_d = +1: a_d = 0, o_d = -1: a_d = 1, o_d = 0: both)
template <int DIM>
__device__ __forceinline__ void n2_box(const double (&Q)[1 << DIM], double (&out)[DIM == 3 ? 27 : 9]) {
  if constexpr (DIM == 2) {
    double X[3][2];
#pragma unroll
    for (int ay = 0; ay < 2; ++ay) { X[0][ay] = Q[1 + 2 * ay]; X[2][ay] = Q[2 * ay]; X[1][ay] = Q[2 * ay] + Q[1 + 2 * ay]; }
#pragma unroll
    for (int ox = 0; ox < 3; ++ox) { out[ox] = X[ox][1]; out[6 + ox] = X[ox][0]; out[3 + ox] = X[ox][0] + X[ox][1]; }
  } else {
    double X[3][2][2];
#pragma unroll
    for (int az = 0; az < 2; ++az)
#pragma unroll
      for (int ay = 0; ay < 2; ++ay) {
        const double q0 = Q[2 * ay + 4 * az], q1 = Q[1 + 2 * ay + 4 * az];
        X[0][ay][az] = q1; X[2][ay][az] = q0; X[1][ay][az] = q0 + q1;
      }
    double Y[3][3][2];
#pragma unroll
    for (int az = 0; az < 2; ++az)
#pragma unroll
      for (int ox = 0; ox < 3; ++ox) { Y[ox][0][az] = X[ox][1][az]; Y[ox][2][az] = X[ox][0][az]; Y[ox][1][az] = X[ox][0][az] + X[ox][1][az]; }
#pragma unroll
    for (int oy = 0; oy < 3; ++oy)
#pragma unroll
      for (int ox = 0; ox < 3; ++ox) {
        out[ox + 3 * oy] = Y[ox][oy][1]; out[18 + ox + 3 * oy] = Y[ox][oy][0]; out[9 + ox + 3 * oy] = Y[ox][oy][0] + Y[ox][oy][1];
      }
  }
}

// Geometry factor for the node kernel: tile_factor's formulas with every rounding written out (no contraction left to the
// compiler).  The straight-line and the general body, the listed (mini) launch and the check mode are different
// instantiations of one source; where the compiler may choose which multiply to fuse into which add, two instantiations
// can round one row differently - harmless for the operator (1e-16), fatal for the row dictionaries, which need the rows
// of a class to repeat BIT FOR BIT (on 2^k cells per direction every product with h is exact and nothing showed; at 384^3
// A11 had 37 classes instead of 28 and the fused check refused them).
template <int DIM>
__device__ __forceinline__ void n2_factor(const double (&J)[DIM][DIM], double* out) {
#pragma clang fp contract(off)
  if constexpr (DIM == 2) {
    const double det = __builtin_fma(J[0][0], J[1][1], -(J[0][1] * J[1][0]));
    const double r = 1.0 / det;
    const double I00 = J[1][1] * r, I01 = -(J[0][1] * r), I10 = -(J[1][0] * r), I11 = J[0][0] * r;
    const double w = fabs(det);
    out[0] = w * __builtin_fma(I00, I00, I10 * I10);
    out[1] = w * __builtin_fma(I00, I01, I10 * I11);
    out[2] = w * __builtin_fma(I01, I01, I11 * I11);
    out[3] = w;
  } else {
    const double c00 = __builtin_fma(J[1][1], J[2][2], -(J[1][2] * J[2][1]));
    const double c01 = __builtin_fma(J[1][2], J[2][0], -(J[1][0] * J[2][2]));
    const double c02 = __builtin_fma(J[1][0], J[2][1], -(J[1][1] * J[2][0]));
    const double det = __builtin_fma(J[0][0], c00, __builtin_fma(J[0][1], c01, J[0][2] * c02));
    const double r = 1.0 / det;
    double I[3][3];
    I[0][0] = c00 * r;
    I[0][1] = __builtin_fma(J[0][2], J[2][1], -(J[0][1] * J[2][2])) * r;
    I[0][2] = __builtin_fma(J[0][1], J[1][2], -(J[0][2] * J[1][1])) * r;
    I[1][0] = c01 * r;
    I[1][1] = __builtin_fma(J[0][0], J[2][2], -(J[0][2] * J[2][0])) * r;
    I[1][2] = __builtin_fma(J[0][2], J[1][0], -(J[0][0] * J[1][2])) * r;
    I[2][0] = c02 * r;
    I[2][1] = __builtin_fma(J[0][1], J[2][0], -(J[0][0] * J[2][1])) * r;
    I[2][2] = __builtin_fma(J[0][0], J[1][1], -(J[0][1] * J[1][0])) * r;
    const double w = fabs(det);
    out[0] = w * __builtin_fma(I[0][0], I[0][0], __builtin_fma(I[1][0], I[1][0], I[2][0] * I[2][0]));
    out[1] = w * __builtin_fma(I[0][0], I[0][1], __builtin_fma(I[1][0], I[1][1], I[2][0] * I[2][1]));
    out[2] = w * __builtin_fma(I[0][0], I[0][2], __builtin_fma(I[1][0], I[1][2], I[2][0] * I[2][2]));
    out[3] = w * __builtin_fma(I[0][1], I[0][1], __builtin_fma(I[1][1], I[1][1], I[2][1] * I[2][1]));
    out[4] = w * __builtin_fma(I[0][1], I[0][2], __builtin_fma(I[1][1], I[1][2], I[2][1] * I[2][2]));
    out[5] = w * __builtin_fma(I[0][2], I[0][2], __builtin_fma(I[1][2], I[1][2], I[2][2] * I[2][2]));
    out[6] = w;
  }
}

// coordinates a node's row needs: the node and its two neighbours along every axis (indices clamped to the node itself
// where the box ends), and the `near` byte
template <int DIM>
struct N2Pre {
  double X0[DIM], XP[DIM][DIM], XM[DIM][DIM];   // XP[d] = x(node + e_d), XM[d] = x(node - e_d)
  uint8_t near;
};
template <int DIM, bool UNI>
__device__ __forceinline__ void n2_fetch(const double* __restrict__ cx, const double* __restrict__ cy, const double* __restrict__ cz,
                                         int px, int py, int pz, const uint8_t* __restrict__ nearp, uint32_t node, int gi, int gj,
                                         int gk, int probe, N2Pre<DIM>& P) {
  if (UNI) return;     // uniform box: the cells are integrated on the canonical edges, no coordinate is read
  const uint32_t stride[3] = {1u, (uint32_t)px, (uint32_t)px * (uint32_t)py};
  const int g[3] = {gi, gj, gk};
  const int pd[3] = {px, py, pz};
  const double* cc[3] = {cx, cy, cz};
  P.near = nearp[node];
#pragma unroll
  for (int c = 0; c < DIM; ++c) P.X0[c] = cc[c][node];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const bool hp = g[d] < pd[d] - 1, hm = g[d] > 0;
    const uint32_t np = hp ? node + stride[d] : node, nm = hm ? node - stride[d] : node;
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      P.XP[d][c] = cc[c][np];
      P.XM[d][c] = cc[c][nm];
    }
  }
}

// K and M of the row of one node, slot by slot.  FAST: every cell around the node exists (no selects).
// CELLMAJOR (default): the 2^d cells one after the other, each adding its 2^d x (components + weight) terms to the slots it
// touches - only kv, mv and ONE cell's factor are live (3 waves per SIMD instead of 2).  The slot coefficients are a
// leading literal per component class (2/9 | 1/3 | 8/27 in 3D) times 1, 1/2, 1/4, 1/8: the literal is multiplied into the
// cell's factor once and the powers of two are exact, so every term is one add / multiply-add with an inline constant
// and equals c(o) * D bit for bit.  !CELLMAJOR: box sums over the cells first (19 additions per component give all
// slots; fewer operations, 56 factor values live).
template <int DIM> __host__ __device__ constexpr double n2_lead(int e, int f) {   // largest |coefficient| of component (e, f): o = 0
  return n2_coef<DIM>((DIM == 3) ? 13 : 4, e, f);
}
template <int DIM, bool FAST, bool UNI, bool CELLMAJOR = true>
__device__ __forceinline__ void n2_row(const N2Pre<DIM>& P, const bool (&has)[DIM][2], double (&kv)[DIM == 3 ? 27 : 9],
                                       double (&mv)[DIM == 3 ? 27 : 9], const double (&hcan)[3]) {
#pragma clang fp contract(off)
  constexpr int NB = 1 << DIM;
  constexpr int NSLOT = (DIM == 3) ? 27 : 9;
  constexpr int NC = DIM * (DIM + 1) / 2;     // components of D
  // the node's edges: E[d][0] towards +d, E[d][1] from -d (a_d = 1); a missing one is replaced by the opposite one
  double E[DIM][2][DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d)
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      if (UNI) {
        // uniform box (every edge checked against the canonical one at mesh build): all cells around the node are the
        // same box, their factors one computation; rows repeat bit for bit whatever the rounding of i / nx (row dictionaries)
        E[d][0][c] = E[d][1][c] = (c == d) ? hcan[d] : 0.0;
      } else {
        const double ep = P.XP[d][c] - P.X0[c], em = P.X0[c] - P.XM[d][c];
        E[d][0][c] = (FAST || has[d][0]) ? ep : em;
        E[d][1][c] = (FAST || has[d][1]) ? em : ep;
      }
    }
  if constexpr (CELLMAJOR) {
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) { kv[q] = 0.0; mv[q] = 0.0; }
#pragma unroll
    for (int a = 0; a < NB; ++a) {
      double J[DIM][DIM];
      bool in = true;
#pragma unroll
      for (int e = 0; e < DIM; ++e) {
        const int ae = (a >> e) & 1;
        in = in && has[e][ae];
#pragma unroll
        for (int d = 0; d < DIM; ++d) J[e][d] = 0.5 * E[e][ae][d];
      }
      double D[NC + 1];
      n2_factor<DIM>(J, D);
      // leading coefficient (and the cell's sign s_e s_f, and 0 for a cell outside the box) folded into the factor
      double U[NC + 1];
      int k = 0;
#pragma unroll
      for (int e = 0; e < DIM; ++e)
#pragma unroll
        for (int f = e; f < DIM; ++f) {
          const bool neg = (e != f) && (((a >> e) & 1) != ((a >> f) & 1));     // s_e s_f = -1
          const double u = (neg ? -n2_lead<DIM>(e, f) : n2_lead<DIM>(e, f)) * D[k];
          U[k] = (FAST || in) ? u : 0.0;
          ++k;
        }
      {
        const double u = n2_cmass<DIM>((DIM == 3) ? 13 : 4) * D[NC];
        U[NC] = (FAST || in) ? u : 0.0;
      }
      // the 2^d slots this cell touches: o_d = 0 or towards the cell (a_d = 0: +1, a_d = 1: -1)
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        int q = 0, mul = 1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          const int od = ((t >> d) & 1) ? (((a >> d) & 1) ? -1 : 1) : 0;
          q += (od + 1) * mul;
          mul *= 3;
        }
        k = 0;
#pragma unroll
        for (int e = 0; e < DIM; ++e)
#pragma unroll
          for (int f = e; f < DIM; ++f) {
            constexpr double zero = 0.0;
            const double r = n2_coef<DIM>(q, e, f) / n2_lead<DIM>(e, f);     // +- 1, 1/2, 1/4 (exact) or 0
            if (r != zero) kv[q] = __builtin_fma(r, U[k], kv[q]);
            ++k;
          }
        mv[q] = __builtin_fma(n2_cmass<DIM>(q) / n2_cmass<DIM>((DIM == 3) ? 13 : 4), U[NC], mv[q]);
      }
    }
    return;
  }
  // geometry factors of the 2^d cells: Dc[k][a], k < NC the packed symmetric D (signed: s_e s_f), k = NC the weight
  double Dc[NC + 1][NB];
#pragma unroll
  for (int a = 0; a < NB; ++a) {
    double J[DIM][DIM];
    bool in = true;
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      const int ae = (a >> e) & 1;
      in = in && has[e][ae];
#pragma unroll
      for (int d = 0; d < DIM; ++d) J[e][d] = 0.5 * E[e][ae][d];
    }
    double D[NC + 1];
    n2_factor<DIM>(J, D);
    int k = 0;
#pragma unroll
    for (int e = 0; e < DIM; ++e)
#pragma unroll
      for (int f = e; f < DIM; ++f) {
        const bool neg = (e != f) && (((a >> e) & 1) != ((a >> f) & 1));     // s_e s_f = -1
        const double v = neg ? -D[k] : D[k];
        Dc[k][a] = (FAST || in) ? v : 0.0;
        ++k;
      }
    Dc[NC][a] = (FAST || in) ? D[NC] : 0.0;
  }
  double bs[NSLOT];
  int k = 0;
#pragma unroll
  for (int e = 0; e < DIM; ++e)
#pragma unroll
    for (int f = e; f < DIM; ++f) {
      n2_box<DIM>(Dc[k], bs);
#pragma unroll
      for (int q = 0; q < NSLOT; ++q) {
        constexpr double zero = 0.0;
        const double c = n2_coef<DIM>(q, e, f);
        if (k == 0) kv[q] = c * bs[q];
        else if (c != zero) kv[q] = __builtin_fma(c, bs[q], kv[q]);
      }
      ++k;
    }
  n2_box<DIM>(Dc[NC], bs);
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) mv[q] = n2_cmass<DIM>(q) * bs[q];
}

// Two output streams, ONE 16-byte store per lane: lanes 2i and 2i + 1 hold the entries of the consecutive rows 2i, 2i + 1 for
// stream A (va) and stream B (vb); they swap one value (DPP quad_perm [1,0,3,2]) so that the even lane writes rows (2i, 2i + 1)
// of stream A and the odd lane rows (2i, 2i + 1) of stream B.  A wave-instruction then covers two contiguous, aligned 512-byte
// runs instead of one, and a row's 42 operator entries leave in 21 store instructions instead of 42: the kernel's stores
// are issue-bound, not bandwidth-bound (timing probes).  pa / pb: this lane's stream base (already
// selected by parity), offp = 8 * (row & ~1).
__device__ __forceinline__ double n2_swap1(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0xB1, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0xB1, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ void n2_store_pair(char* base, uint32_t offp, bool odd, double va, double vb) {
  const double got = n2_swap1(odd ? va : vb);     // the even lane needs its partner's va, the odd lane its partner's vb
  pph_d2 v;
  v.x = odd ? got : va;
  v.y = odd ? vb : got;
  // non-temporal: the operator entries are not read again by this kernel and would evict the coordinate lines its
  // neighbours' rows need from the XCD's L2 (256^3 fine level: 2.50 -> 2.18 ms)
  __builtin_nontemporal_store(v, reinterpret_cast<pph_d2*>(base + offp));
}

// fused epilogue (Dirichlet elimination, A11 / A22 / A12 (/ A21), lifting, u0, 1 / a_ii, spectral bound) - no predicated
// region in either form: the general form loads the masks / boundary values of every neighbour (index clamped to the node
// itself where there is none; the entry is then an exact 0 and is stored as the pad's +0) and selects.  All predicates are
// 0 / 1 integers combined with & | ^ (C++'s && would come back as exec-mask regions with the loads sunk into them).
// SAME: both fields carry one Dirichlet set (m1 == m2): one predicate per entry position instead of four.
// MODE 1: every stored entry is compared with the stored half of its class's table row (LDS; fact A of the fused dictionary
// check); MODE 2 (listed): only the operator entries are produced
template <int DIM, bool FAST, bool SAME, bool SYM, bool SYMC, bool HAS12, bool HAS21, bool HASRHS, int MODE>
__device__ __forceinline__ void n2_epilogue(const double (&kv)[DIM == 3 ? 27 : 9], const double (&mv)[DIM == 3 ? 27 : 9],
                                            const unsigned (&hasb)[DIM][2], int px, int py, int64_t n, const FuseArgs& fa,
                                            uint32_t node, uint32_t nodeS, unsigned liveb, unsigned nearb, double& best1,
                                            double& best2, const double* __restrict__ stab, int c11, int c22, int c12) {
#pragma clang fp contract(off)
  // node: the row for loads (0 for a lane beyond n); nodeS: the lane's own row index for the operator stores (a lane beyond
  // n, always inside the leading dimension, writes the zeros of its padding row); liveb: 1 for a row of the mesh
  constexpr int NSLOT = (DIM == 3) ? 27 : 9;
  const uint32_t upxy = (uint32_t)px * (uint32_t)py;
  unsigned r1 = 0, r2 = 0;
  if (!FAST) {
    const unsigned nm = 0u - nearb;
    r1 = (unsigned)fa.m1[node] & nm;
    r2 = SAME ? r1 : ((unsigned)fa.m2[node] & nm);
  }
  const unsigned free1 = (r1 & 1u) ^ 1u, free2 = (r2 & 1u) ^ 1u, own1 = ((r1 >> 1) & 1u) ^ 1u, own2 = ((r2 >> 1) & 1u) ^ 1u;
  const unsigned symg = fa.symg ? 1u : 0u;
  double t11 = 0.0, t22 = 0.0, d11 = 0.0, d22 = 0.0, tK1 = 0.0, tK2 = 0.0, tM = 0.0;
  const uint32_t offp = (nodeS & ~1u) * 8u;                     // (launcher: n < 2^29)
  const bool odd = (nodeS & 1u) != 0;
  const int64_t ldb = fa.ld * 8;
  // this lane's stream bases: diagonal blocks - even lanes A11, odd lanes A22 of the same slot; coupling blocks - A12 and
  // A21 of the same slot when A21 is stored, else two consecutive slots of A12 (even lane the first)
  char* const pD = reinterpret_cast<char*>(odd ? fa.A22 : fa.A11);
  char* const pC = HAS21 ? reinterpret_cast<char*>(odd ? fa.A21 : fa.A12) : reinterpret_cast<char*>(fa.A12) + (odd ? ldb : 0);
  double held12 = 0.0;
  // check modes: the row's class per operator -> its table row (stored half: NSLOT / 2 + 1 entries per class).  MODE 3: all
  // 64 rows of the wave are of ONE class per operator (every interior wave of a uniform box) - the table row is wave-uniform
  // and comes through scalar loads from the global table (c11 .. c12 are then the wave's classes); MODE 1: per-lane rows from LDS.
  constexpr int SSC = NSLOT / 2 + 1;
  constexpr bool CHK = (MODE == 1 || MODE == 3);
  const bool chk12 = CHK && SYMC && HAS12 && fa.dcls[2] != nullptr;
  const double *tr11 = nullptr, *tr22 = nullptr, *tr12 = nullptr;
  unsigned long long acc11 = 0ull, acc22 = 0ull, acc12 = 0ull;
  if (MODE == 1) {
    tr11 = stab + c11 * SSC;
    tr22 = stab + fa.dn[0] * SSC + c22 * SSC;
    tr12 = stab + (fa.dn[0] + fa.dn[1]) * SSC + (chk12 ? c12 : 0) * SSC;
  } else if (MODE == 3) {
    tr11 = fa.dtab[0] + c11 * NSLOT + NSLOT / 2;
    tr22 = fa.dtab[1] + c22 * NSLOT + NSLOT / 2;
    tr12 = chk12 ? fa.dtab[2] + c12 * NSLOT + NSLOT / 2 : tr11;
  }
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) {
    const int ox = n2_o<DIM>(q, 0), oy = n2_o<DIM>(q, 1), oz = n2_o<DIM>(q, 2);
    const double kvs = kv[q], mvs = mv[q];
    const bool diag = (q == NSLOT / 2);
    const double bm = fa.b * mvs;
    double o11 = __builtin_fma(fa.a, kvs, bm), o22 = __builtin_fma(fa.c, kvs, bm), o12 = -bm, o21 = o12;
    if (!FAST) {
      unsigned ex = liveb;
      if (ox != 0) ex &= hasb[0][ox > 0 ? 0 : 1];
      if (oy != 0) ex &= hasb[1][oy > 0 ? 0 : 1];
      if (DIM == 3 && oz != 0) ex &= hasb[DIM - 1][oz > 0 ? 0 : 1];
      const uint32_t j = node + ((uint32_t)(ox + oy * px) + (uint32_t)oz * upxy) * ex;     // (the node itself where there is no neighbour)
      const unsigned um = 0u - (ex & nearb);
      const unsigned cm1 = (unsigned)fa.m1[j] & um;
      const unsigned cm2 = SAME ? cm1 : ((unsigned)fa.m2[j] & um);
      if (HASRHS) {
        const double v1 = fa.g1[j], v2 = fa.g2[j];
        tK1 = __builtin_fma(kvs, v1, tK1); tK2 = __builtin_fma(kvs, v2, tK2);
        tM = __builtin_fma(mvs, v1 - v2, tM);     // (kvs = mvs = 0 where there is no neighbour)
      }
      // fuse_elim_diag / fuse_elim_coupling: an entry survives in a free row towards a free column; the ghost row of a slab
      // keeps it only in symmetric storage and towards an owned column; a constrained (non-ghost) row keeps a unit diagonal
      const unsigned cf1 = (cm1 & 1u) ^ 1u, cf2 = (cm2 & 1u) ^ 1u, co1 = ((cm1 >> 1) & 1u) ^ 1u, co2 = ((cm2 >> 1) & 1u) ^ 1u;
      const unsigned k11 = ex & free1 & cf1 & (own1 | (symg & co1));
      const unsigned k22 = SAME ? k11 : (ex & free2 & cf2 & (own2 | (symg & co2)));
      const unsigned k12 = SAME ? k11 : (ex & free1 & cf2 & (own1 | (symg & co2)));
      const unsigned k21 = SAME ? k11 : (ex & free2 & cf1 & (own2 | (symg & co1)));
      const double u1 = (diag && ((free1 ^ 1u) & own1)) ? 1.0 : 0.0, u2 = (diag && ((free2 ^ 1u) & own2)) ? 1.0 : 0.0;
      o11 = k11 ? o11 : u1;
      o22 = k22 ? o22 : u2;
      o12 = k12 ? o12 : 0.0;
      o21 = k21 ? o21 : 0.0;
    }
    const int so = SYM ? q - NSLOT / 2 : q, sc = SYMC ? q - NSLOT / 2 : q;   // stored slot (< 0: lower half, not stored)
    constexpr int NSC = SYMC ? NSLOT / 2 + 1 : NSLOT;                         // stored coupling slots
    if (CHK && SYM && so >= 0) {
      acc11 |= (unsigned long long)(__double_as_longlong(o11) ^ __double_as_longlong(tr11[so]));
      acc22 |= (unsigned long long)(__double_as_longlong(o22) ^ __double_as_longlong(tr22[so]));
    }
    if (CHK && SYMC && HAS12 && sc >= 0) acc12 |= (unsigned long long)(__double_as_longlong(o12) ^ __double_as_longlong(tr12[sc]));
    if (so >= 0) n2_store_pair(pD + so * ldb, offp, odd, o11, o22);
    if (sc >= 0 && HAS12) {
      if (HAS21) {
        n2_store_pair(pC + sc * ldb, offp, odd, o12, o21);
      } else if ((sc & 1) == 0 && sc + 1 < NSC) {
        held12 = o12;                                                        // first of a pair of slots
      } else if ((sc & 1) == 1) {
        n2_store_pair(pC + (sc - 1) * ldb, offp, odd, held12, o12);
      } else {                                                               // odd slot count: the last one alone
        *reinterpret_cast<double*>(reinterpret_cast<char*>(fa.A12) + sc * ldb + nodeS * 8u) = o12;
      }
    }
    t11 += fabs(o11); t22 += fabs(o22);
    if (diag) { d11 = o11; d22 = o22; }
    // general form: keep the neighbour loads of later slots from being hoisted above this slot's stores (each of the 27
    // slots brings 2 - 4 loads: all of them in flight at once is 150 more live registers, i.e. scratch)
    if (!FAST) __builtin_amdgcn_sched_barrier(0);
  }
  if (CHK) {
    // a row that left its class refuses that operator's dictionary: products take the stored values (pph_sell.hip)
    const bool b11 = liveb && acc11 != 0ull, b22 = liveb && acc22 != 0ull, b12 = liveb && chk12 && acc12 != 0ull;
    if (b11 | b22 | b12) {
      if (b11) atomicExch(fa.dstate[0] + 1, -2);
      if (b22) atomicExch(fa.dstate[1] + 1, -2);
      if (b12) atomicExch(fa.dstate[2] + 1, -2);
      if (fa.alarm) __hip_atomic_store(fa.alarm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (MODE == 2) return;                       // listed mode: operator entries only
  const double i1 = (d11 != 0.0) ? 1.0 / d11 : 1.0, i2 = (d22 != 0.0) ? 1.0 / d22 : 1.0;
  const double q1 = t11 * fabs(i1), q2 = t22 * fabs(i2);
  best1 = (own1 && q1 > best1) ? q1 : best1;   // (ghost rows: not rows of this rank's operator)
  best2 = (own2 && q2 > best2) ? q2 : best2;
  if (!FAST && !liveb) return;                 // (the per-row vectors have n entries, not the leading dimension)
  fa.dinv1[node] = i1;
  fa.dinv2[node] = i2;
  if (HASRHS) {
    double o1 = 0.0, o2 = 0.0, u1 = 0.0, u2 = 0.0;
    if (!FAST) {
      const double btm = fa.b * tM;
      const double w1 = -__builtin_fma(fa.a, tK1, btm), w2 = -__builtin_fma(fa.c, tK2, -btm);
      const double h1 = fa.g1[node], h2 = fa.g2[node];
      o1 = (nearb & (r1 == 0u ? 1u : 0u)) ? w1 : 0.0;
      o2 = (nearb & (r2 == 0u ? 1u : 0u)) ? w2 : 0.0;
      u1 = nearb ? h1 : 0.0;
      u2 = nearb ? h2 : 0.0;
    }
    fa.rhs[node] = o1;
    fa.rhs[n + node] = o2;
    fa.u0[node] = u1;
    fa.u0[n + node] = u2;
  }
}

// PATH 0: both bodies in one kernel; 1: only the waves that take the straight-line body, 2: only the others (two
// launches with separate register allocations; measured against PATH 0)
template <int DIM, bool SYM, bool SYMC, bool HAS12, bool HAS21, bool HASRHS, bool SAME, int PATH, int MODE = 0, bool UNI = false>
__global__ __launch_bounds__(256, (PATH == 1 && MODE != 1) ? 3 : 2) void k_asm_node2(const double* __restrict__ cx, const double* __restrict__ cy,
                                                      const double* __restrict__ cz, int nx, int ny, int nzl, int px, int py,
                                                      int pz, int64_t n, FuseArgs fa, int xmap) {
  constexpr int NSLOT = (DIM == 3) ? 27 : 9;
  double best1 = 0.0, best2 = 0.0;
  // (the timing probes of DESIGN.md section 4.2 - no operator stores, synthetic coordinates - were runtime flags: every flag
  // test splits the straight-line code; removed after measuring)
  extern __shared__ double n2_stab[];
  if (MODE == 1) {
    // stored halves of the dictionaries' tables -> LDS: [A11 classes][A22 classes][A12 classes] x (NSLOT / 2 + 1)
    constexpr int SSC = NSLOT / 2 + 1, C0 = NSLOT / 2;
    int base = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (fa.dtab[d])
        for (int i = threadIdx.x; i < fa.dn[d] * SSC; i += 256) n2_stab[base + i] = fa.dtab[d][(i / SSC) * NSLOT + C0 + i % SSC];
      base += fa.dn[d] * SSC;
    }
    __syncthreads();
  }
  // Block order (blocks of 256 consecutive nodes): round-robin over the workgroups (xmap 0, default) or one contiguous eighth
  // per XCD (1).  Measured at 256^3 with the non-temporal stores: round-robin 2.17 ms, contiguous eighths 2.75 (eight write
  // fronts), and an XCD-striped order - XCD x takes, plane after plane, the blocks of the x-th in-plane stripe, persistent
  // grid, so that a node's y and z neighbours share an L2 - 3.44 (the general-form waves pile up in the boundary stripes
  // and planes of a static assignment); removed.
  const int64_t nblk = (n + 255) / 256;
  const int64_t bpx = xmap ? (int64_t)(gridDim.x >> 3) : (int64_t)gridDim.x;
  const int64_t cpx = xmap ? (nblk + 7) >> 3 : nblk;
  const int64_t blk0 = xmap ? (int64_t)(blockIdx.x & 7) * cpx : 0;
  // (A rotated loop - the NEXT block's coordinates requested before this block's stores - was also built and measured: the
  // wait for those loads at the loop head then waits for the stores issued after them as well (one counter, in issue order),
  // and 43 more live registers spill: 1.34 -> 1.91 ms.  The loads stay at the head of their own block.)
  for (int64_t c = xmap ? (blockIdx.x >> 3) : blockIdx.x; c < cpx; c += bpx) {
    const int64_t node64 = (blk0 + c) * 256 + threadIdx.x;
    if (node64 - (threadIdx.x & 63) >= n) continue;     // a wave without a row (wave-uniform)
    // a lane beyond n (last wave only; its index stays inside the leading dimension, a multiple of 64) goes through the general
    // form with every predicate 0 and writes the zeros of its padding row: the paired stores need both lanes of a pair
    // (listed mode: n = the rows of the mini operator, a multiple of 64, every one of them a node of the list)
    const bool live = node64 < n;
    const uint32_t nodeS = (uint32_t)node64;
    const uint32_t node = MODE == 2 ? fa.list[nodeS] : (live ? nodeS : 0u);
    const int gi = (int)(node % (uint32_t)px);
    const uint32_t tq = node / (uint32_t)px;
    const int gj = (int)(tq % (uint32_t)py), gk = (int)(tq / (uint32_t)py);
    const bool near = live && fa.near[node] != 0;
    const bool inner = live && !near && gi > 0 && gi < px - 1 && gj > 0 && gj < py - 1 && (DIM == 2 || (gk > 0 && gk < pz - 1));
    const bool fast = __all(inner);
    const bool do_fast = PATH != 2 && fast, do_gen = PATH != 1 && !fast;
    if (!do_fast && !do_gen) continue;
    N2Pre<DIM> P;
    n2_fetch<DIM, UNI>(cx, cy, cz, px, py, pz, fa.near, node, gi, gj, gk, 0, P);
    bool has[DIM][2];
    has[0][0] = gi < px - 1; has[0][1] = gi > 0;
    has[1][0] = gj < py - 1; has[1][1] = gj > 0;
    if constexpr (DIM == 3) { has[2][0] = gk < pz - 1; has[2][1] = gk > 0; }
    unsigned hasb[DIM][2];
#pragma unroll
    for (int d = 0; d < DIM; ++d) { hasb[d][0] = has[d][0] ? 1u : 0u; hasb[d][1] = has[d][1] ? 1u : 0u; }
    // check mode: the row's classes (a wave-uniform variant - every interior wave of a uniform box has ONE class per operator,
    // its table rows through scalar loads - was built beside the per-lane LDS rows: two epilogues in one kernel spill 0.6 - 1.3 KB
    // per lane; the LDS rows alone are cheap enough)
    int c11 = 0, c22 = 0, c12 = 0;
    if (MODE == 1) {
      const bool has12 = SYMC && HAS12 && fa.dcls[2] != nullptr;
      c11 = fa.dcls[0][node]; c22 = fa.dcls[1][node]; c12 = has12 ? (int)fa.dcls[2][node] : 0;
    }
    double kv[NSLOT], mv[NSLOT];
    if (do_fast) {
      n2_row<DIM, true, UNI>(P, has, kv, mv, fa.hcan);
      n2_epilogue<DIM, true, true, SYM, SYMC, HAS12, HAS21, HASRHS, MODE>(kv, mv, hasb, px, py, n, fa, node, nodeS, 1u, 0u, best1, best2,
                                                                         n2_stab, c11, c22, c12);
    } else {
      n2_row<DIM, false, UNI>(P, has, kv, mv, fa.hcan);
      const unsigned nearb = near ? 1u : 0u;
      n2_epilogue<DIM, false, SAME, SYM, SYMC, HAS12, HAS21, HASRHS, MODE>(kv, mv, hasb, px, py, n, fa, node, nodeS, live ? 1u : 0u, nearb,
                                                                            best1, best2, n2_stab, c11, c22, c12);
    }
  }
  if (MODE != 2) fuse_lam_max(best1, best2, fa.lam);
}

// Fact A of the fused dictionary check for the rows of the GENERAL-form waves: the straight-line launch compares what it
// stores for free (it waits on memory), the general-form launch has no registers to spare (the compare costs it 1 KB of scratch
// per lane: 1.0 -> 2.85 ms), so its rows - about a quarter of a uniform box - are read back once: stored half of every
// operator against the stored half of the row's class, bit for bit.  Same wave classification as k_asm_node2.
template <int DIM>
__global__ __launch_bounds__(256) void k_n2_check_general(int px, int py, int pz, int64_t n, FuseArgs fa) {
  constexpr int NSLOT = (DIM == 3) ? 27 : 9, SSC = NSLOT / 2 + 1, C0 = NSLOT / 2;
  extern __shared__ double n2_ctab[];
  int base = 0, off[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    off[d] = base;
    if (fa.dtab[d])
      for (int i = threadIdx.x; i < fa.dn[d] * SSC; i += 256) n2_ctab[base + i] = fa.dtab[d][(i / SSC) * NSLOT + C0 + i % SSC];
    base += fa.dn[d] * SSC;
  }
  __syncthreads();
  const double* vals[3] = {fa.A11, fa.A22, fa.A12};
  const int64_t nblk = (n + 255) / 256;
  for (int64_t c = blockIdx.x; c < nblk; c += gridDim.x) {
    const int64_t node64 = c * 256 + threadIdx.x;
    if (node64 - (threadIdx.x & 63) >= n) continue;
    const bool live = node64 < n;
    const uint32_t node = live ? (uint32_t)node64 : 0u;
    const int gi = (int)(node % (uint32_t)px);
    const uint32_t tq = node / (uint32_t)px;
    const int gj = (int)(tq % (uint32_t)py), gk = (int)(tq / (uint32_t)py);
    const bool near = live && fa.near[node] != 0;
    const bool inner = live && !near && gi > 0 && gi < px - 1 && gj > 0 && gj < py - 1 && (DIM == 2 || (gk > 0 && gk < pz - 1));
    if (__all(inner) || !live) continue;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (!fa.dtab[d]) continue;
      const double* row = n2_ctab + off[d] + (int)fa.dcls[d][node] * SSC;
      unsigned long long acc = 0ull;
#pragma unroll
      for (int ss = 0; ss < SSC; ++ss)
        acc |= (unsigned long long)(__double_as_longlong(vals[d][(int64_t)ss * fa.ld + node]) ^ __double_as_longlong(row[ss]));
      if (acc != 0ull) {
        atomicExch(fa.dstate[d] + 1, -2);
        if (fa.alarm) __hip_atomic_store(fa.alarm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// every cell of a multilinear mesh has equal parallel edges (exact test of the tile kernel's phase A): out[0] != 0 otherwise
template <int DIM>
__global__ __launch_bounds__(256) void k_affine_check(const double* __restrict__ cx, const double* __restrict__ cy,
                                                      const double* __restrict__ cz, int nx, int ny, int nzl, int px, int py,
                                                      int* __restrict__ out, double hx, double hy, double hz, double tol) {
  constexpr int NB = 1 << DIM;
  const int64_t ncell = (int64_t)nx * ny * (DIM == 3 ? nzl : 1);
  const int64_t pxy = (int64_t)px * py;
  bool bad = false, odd = false;
  const double hc[3] = {hx, hy, hz};
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < ncell; c += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(c % nx);
    const int64_t t = c / nx;
    const int cj = (int)(t % ny), ck = (int)(t / ny);
    double X[NB][DIM];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int64_t g = (ci + (b & 1)) + (int64_t)px * (cj + ((b >> 1) & 1)) + pxy * (ck + ((DIM == 3) ? (b >> 2) & 1 : 0));
      X[b][0] = cx[g];
      X[b][1] = cy[g];
      if constexpr (DIM == 3) X[b][2] = cz[g];
    }
#pragma unroll
    for (int e = 0; e < DIM; ++e)
#pragma unroll
      for (int b = 1; b < NB; ++b) {
        if ((b >> e) & 1) continue;
#pragma unroll
        for (int d = 0; d < DIM; ++d) bad = bad || !(X[b | (1 << e)][d] - X[b][d] == X[1 << e][d] - X[0][d]);
      }
    // uniform box: the edge along axis e is (hc[e] e_e) up to the rounding of the coordinates
#pragma unroll
    for (int e = 0; e < DIM; ++e)
#pragma unroll
      for (int d = 0; d < DIM; ++d) odd = odd || !(fabs((X[1 << e][d] - X[0][d]) - (d == e ? hc[e] : 0.0)) <= tol);
  }
  if (bad) atomicOr(out, 1);
  if (odd) atomicOr(out, 2);
}

int pph_mesh_check_affine(pph_ctx* ctx, MeshData& mesh) {
  mesh.all_affine = false;
  mesh.uniform = false;
  if (mesh.kind != PPH_CELL_QUAD && mesh.kind != PPH_CELL_HEX) return PPH_OK;
  // canonical edges of the unit square / cube this library builds (k_coords: node i at i / nx): the same doubles on every
  // rank of a slab decomposition and on every multigrid level; tolerance = one rounding of a coordinate of magnitude <= 1
  const double hx = 1.0 / (double)mesh.nx, hy = 1.0 / (double)mesh.ny, hz = mesh.dim == 3 ? 1.0 / (double)mesh.nz : 0.0;
  const double tol = 2.220446049250313e-16;
  DevBuf<int> flag;
  PPH_TRY(flag.alloc(ctx, 1));
  PPH_HIP(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
  const int64_t ncell = mesh.ncell;
  const int grid = (int)(ceil_div64(ncell, 256) < 4096 ? ceil_div64(ncell, 256) : 4096);
  if (mesh.dim == 2)
    hipLaunchKernelGGL(k_affine_check<2>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cx.p, mesh.cy.p, mesh.cz.p, mesh.nx,
                       mesh.ny, 0, mesh.px, mesh.py, flag.p, hx, hy, hz, tol);
  else
    hipLaunchKernelGGL(k_affine_check<3>, dim3(grid), dim3(256), 0, ctx->stream, mesh.cx.p, mesh.cy.p, mesh.cz.p, mesh.nx,
                       mesh.ny, mesh.nzl, mesh.px, mesh.py, flag.p, hx, hy, hz, tol);
  int h = 1;
  PPH_HIP(ctx, hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  flag.release();
  mesh.all_affine = (h & 1) == 0;
  mesh.uniform = h == 0;
  mesh.hcan[0] = hx; mesh.hcan[1] = hy; mesh.hcan[2] = hz;
  return PPH_OK;
}

// Fused assembly of the fine level for multilinear cells (two-pass kernels): element rows, then ONE node-centred
// pass that writes the eliminated blocks, the lifted right-hand side and the smoother's diagonal / spectral bound
// (and K, M as well when `asm_keep_km` is set, so that later assemblies with other coefficients reuse them).
bool pph_can_fuse_assembly(const pph_ctx* ctx) {
  const MeshData& m = ctx->mesh;
  if (!ctx->asm_fused) return false;
  if (m.kind == PPH_CELL_QUAD || m.kind == PPH_CELL_HEX) return ctx->asm_kernel == 2;
  return ctx->asm_kernel != 0;   // simplices: the node-centred gather kernel
}

// element rows (multilinear cells) + the fused node-centred pass on any level mesh
int pph_launch_fused_kernels(pph_ctx* ctx, MeshData& mesh, const FuseArgs& fa, double* Kp, double* Mp) {
  const bool multilinear = (mesh.kind == PPH_CELL_QUAD || mesh.kind == PPH_CELL_HEX);
  // CSR positions are needed for CSR output and for K / M; the simplex and two-pass gather kernels also walk the
  // pattern's columns (the node and tile kernels address neighbours in closed form)
  const bool closed_form = multilinear && ctx->asm_tile && !ctx->asm_ring &&
                           ((ctx->asm_node && ctx->asm_affine && mesh.all_affine && !ctx->asm_tile_probe && mesh.n < ((int64_t)1 << 29)) ||
                            ctx->asm_tile == 2 || mesh.n >= ctx->asm_tile_min_nodes);
  if (fa.ld == 0 || fa.keep_km || !closed_form) PPH_TRY(pph_ensure_pattern(ctx, mesh));
  if (!multilinear) {
    int64_t nbs = ceil_div64(mesh.n, 256);
    const int gs = (int)(nbs < 256 * 32 ? nbs : 256 * 32);
    if (mesh.kind == PPH_CELL_TRI)
      hipLaunchKernelGGL((k_asm_simplex_gather<2, true>), dim3(gs), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p,
                         mesh.cy.p, mesh.cz.p, mesh.rowptr.p, mesh.col.p, Kp, Mp, mesh.nx, mesh.ny, 0, mesh.px, mesh.py,
                         mesh.n, fa);
    else
      hipLaunchKernelGGL((k_asm_simplex_gather<3, true>), dim3(gs), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p,
                         mesh.cy.p, mesh.cz.p, mesh.rowptr.p, mesh.col.p, Kp, Mp, mesh.nx, mesh.ny, mesh.nzl, mesh.px,
                         mesh.py, mesh.n, fa);
    PPH_HIP(ctx, hipGetLastError());
    return PPH_OK;
  }
  // box meshes (every cell with equal parallel edges), stencil-ELL output: one thread per node, registers only
  if (ctx->asm_node && ctx->asm_tile && ctx->asm_affine && mesh.all_affine && fa.ld != 0 && !fa.keep_km && !ctx->asm_tile_probe && !ctx->asm_ring &&
      mesh.n < ((int64_t)1 << 29)) {
    const int64_t nb = ((ceil_div64(mesh.n, 256) + 7) / 8) * 8;
    const int grid = (int)(nb < 256 * 64 ? nb : 256 * 64);
    // k_asm_node2 (round 4): storage variants as template parameters; 32-bit node offsets (n < 2^29: 812^3 nodes)
    const int q0 = mesh.kind == PPH_CELL_QUAD ? 9 : 0;     // the first (lowest) stencil offset: not stored in symmetric storage
    const bool sym = fa.slot_of[q0] < 0, symc = fa.slot_of_c[q0] < 0;
    const bool h12 = fa.A12 != nullptr, h21 = fa.A21 != nullptr, hr = fa.rhs != nullptr;
    const bool samek = fa.same != 0;       // both fields carry one Dirichlet set (then A21 is not stored: aliased to A12)
    const int variant = (!h12 && !h21 && !hr) ? (sym ? 0 : 1) + (samek ? 0 : 6)             // multigrid level operators
                        : (h12 && hr && !h21 && samek && sym == symc) ? (sym ? 2 : 3)       // fine level, A21 aliased to A12
                        : (h12 && hr && h21 && !samek && !symc) ? (sym ? 4 : 5) : -1;       // fine level, A21 stored on its own
    PPH_REQUIRE(ctx, variant >= 0, "node assembly kernel: no variant for this combination of stored blocks (A12 %d A21 %d rhs %d same %d sym %d symc %d)",
                (int)h12, (int)h21, (int)hr, (int)samek, (int)sym, (int)symc);
    {
      const int pz = mesh.kind == PPH_CELL_QUAD ? 1 : mesh.pzl, nz = mesh.kind == PPH_CELL_QUAD ? 0 : mesh.nzl;
      const bool split = mesh.n >= ctx->asm_node_split_min;   // two launches: straight-line waves, then the others
      // Row dictionaries in use on the operators of this launch and unchanged in shape: their per-assembly check runs inside
      // the kernel (pph_sell.hip, "check fused into the assembly") - representative rows first (listed mode) -> tables +
      // class adjacencies -> the assembly proper compares what it stores.  Otherwise sell_dict_update checks afterwards.
      // uniform box: integrate on the canonical edges (no coordinate loads; rows repeat whatever the rounding of i / nx)
      const bool uni = ctx->asm_uniform && mesh.uniform;
      FuseArgs fc = fa;
      for (int d = 0; d < 3; ++d) fc.hcan[d] = mesh.hcan[d];
      bool fuse = ctx->dict_fuse && fa.G && fa.G->ok && split && sym && (variant == 0 || variant == 2 || variant == 4 || variant == 6);
      int nd = 0;
      size_t lds = 0;
      if (fuse) {
        const int want_nd = (variant == 2) ? 3 : 2;      // (A12 has a dictionary only in symmetric storage, i.e. with one Dirichlet set)
        for (int d = 0; d < want_nd && fuse; ++d) {
          const SellDict* D = fa.dicts[d];
          const Sell* E = fa.views[d];
          fuse = D && E && D->on && D->adj_ok && D->val == E->val && D->n == mesh.n && D->px == E->px && D->py == E->py &&
                 D->bc_epoch == ctx->bc_epoch && D->cap == ctx->sell_dict_cap && D->ncls <= PPH_DICT_FUSE_CAP &&
                 fa.G->cls_of[d] == (const void*)D->cls.p && fa.G->ncls[d] == D->ncls && ctx->sell_dict &&
                 mesh.n >= ctx->sell_dict_min_rows && ctx->sell_rpt != 1;
        }
        nd = want_nd;
      }
      if (fuse) {
        const int SS = sell_stored(mesh.kind, 1);
        DictGroup& G = *fa.G;
        FuseArgs fl = fc;
        fl.ld = G.ldm;
        fl.A11 = G.mini.p; fl.A22 = G.mini.p + (size_t)SS * G.ldm; fl.A12 = nd == 3 ? G.mini.p + (size_t)2 * SS * G.ldm : nullptr;
        fl.A21 = nullptr; fl.rhs = nullptr; fl.u0 = nullptr;
        fl.list = G.list.p;
        const int gl = (int)ceil_div64(G.ldm, 256);
#define PPH_N2L(DIMV, H12, HR, SM, UNIV)                                                                                            \
        hipLaunchKernelGGL((k_asm_node2<DIMV, true, true, H12, false, HR, SM, 0, 2, UNIV>), dim3(gl), dim3(256), 0, ctx->stream,       \
                           mesh.cx.p, mesh.cy.p, mesh.cz.p, mesh.nx, mesh.ny, nz, mesh.px, mesh.py, pz, G.ldm, fl, 0)
#define PPH_N2L_DIM(DIMV, UNIV)                                                                                     \
        do {                                                                                                        \
          if (nd == 3) PPH_N2L(DIMV, true, true, true, UNIV);                                                       \
          else if (samek) PPH_N2L(DIMV, false, false, true, UNIV);                                                  \
          else PPH_N2L(DIMV, false, false, false, UNIV);                                                            \
        } while (0)
        if (mesh.kind == PPH_CELL_QUAD) { if (uni) PPH_N2L_DIM(2, true); else PPH_N2L_DIM(2, false); }
        else { if (uni) PPH_N2L_DIM(3, true); else PPH_N2L_DIM(3, false); }
#undef PPH_N2L_DIM
#undef PPH_N2L
        PPH_TRY(dict_group_tables(ctx, G, fa.dicts, nd, *fa.views[0]));
        for (int d = 0; d < nd; ++d) {
          fc.dcls[d] = fa.dicts[d]->cls.p; fc.dtab[d] = fa.dicts[d]->tab.p; fc.dstate[d] = fa.dicts[d]->state.p; fc.dn[d] = fa.dicts[d]->ncls;
          lds += (size_t)fa.dicts[d]->ncls * SS * sizeof(double);
        }
        fc.alarm = ctx->dict_alarm_dev;
      }
#define PPH_N2P(DIMV, S, SC, H12, H21, HR, SM, PATHV, MODEV, UNIV)                                                                      \
      hipLaunchKernelGGL((k_asm_node2<DIMV, S, SC, H12, H21, HR, SM, PATHV, MODEV, UNIV>), dim3(grid), dim3(256), MODEV == 1 ? lds : 0,    \
                         ctx->stream, mesh.cx.p, mesh.cy.p, mesh.cz.p, mesh.nx, mesh.ny, nz, mesh.px, mesh.py, pz, mesh.n, fc,           \
                         ctx->asm_node_xmap)
#define PPH_N2(DIMV, S, SC, H12, H21, HR, SM, UNIV)                                                        \
      do {                                                                                                  \
        if (split) {                                                                                        \
          if (!(ctx->asm_node_probe & 1)) PPH_N2P(DIMV, S, SC, H12, H21, HR, SM, 1, 0, UNIV);               \
          if (!(ctx->asm_node_probe & 2)) PPH_N2P(DIMV, S, SC, H12, H21, HR, SM, 2, 0, UNIV);               \
        }                                                                                                   \
        else PPH_N2P(DIMV, S, SC, H12, H21, HR, SM, 0, 0, UNIV);                                            \
      } while (0)
      // (check mode exists for the symmetric-storage variants in two launches only: what a dictionary needs anyway)
#define PPH_N2C(DIMV, S, SC, H12, H21, HR, SM, UNIV)                                                       \
      do {                                                                                                  \
        PPH_N2P(DIMV, S, SC, H12, H21, HR, SM, 1, 1, UNIV);                                                 \
        PPH_N2P(DIMV, S, SC, H12, H21, HR, SM, 2, 0, UNIV);                                                 \
        hipLaunchKernelGGL(k_n2_check_general<DIMV>, dim3(grid), dim3(256), lds, ctx->stream, mesh.px, mesh.py, pz, mesh.n, fc); \
      } while (0)
#define PPH_N2_DIM(DIMV, UNIV)                                                  \
      switch (variant) {                                                         \
        case 0: if (fuse) PPH_N2C(DIMV, true, true, false, false, false, true, UNIV); else PPH_N2(DIMV, true, true, false, false, false, true, UNIV); break;      \
        case 1: PPH_N2(DIMV, false, false, false, false, false, true, UNIV); break;    \
        case 2: if (fuse) PPH_N2C(DIMV, true, true, true, false, true, true, UNIV); else PPH_N2(DIMV, true, true, true, false, true, true, UNIV); break;        \
        case 3: PPH_N2(DIMV, false, false, true, false, true, true, UNIV); break;      \
        case 4: if (fuse) PPH_N2C(DIMV, true, false, true, true, true, false, UNIV); else PPH_N2(DIMV, true, false, true, true, true, false, UNIV); break;       \
        case 5: PPH_N2(DIMV, false, false, true, true, true, false, UNIV); break;      \
        case 6: if (fuse) PPH_N2C(DIMV, true, true, false, false, false, false, UNIV); else PPH_N2(DIMV, true, true, false, false, false, false, UNIV); break;     \
        default: PPH_N2(DIMV, false, false, false, false, false, false, UNIV); break;  \
      }
      if (mesh.kind == PPH_CELL_QUAD) { if (uni) { PPH_N2_DIM(2, true) } else { PPH_N2_DIM(2, false) } }
      else { if (uni) { PPH_N2_DIM(3, true) } else { PPH_N2_DIM(3, false) } }
#undef PPH_N2_DIM
#undef PPH_N2C
#undef PPH_N2
#undef PPH_N2P
      PPH_HIP(ctx, hipGetLastError());
      return PPH_OK;
    }
  }
  // (small levels: the two-pass kernels, whose many small workgroups fill the chip where a few thousand tiles do not)
  if (ctx->asm_tile && !ctx->asm_ring && (ctx->asm_tile == 2 || mesh.n >= ctx->asm_tile_min_nodes)) {
    // single pass, no element-row buffer (k_asm_tile)
    const int tx = (mesh.dim == 3) ? 8 : 16, ty = (mesh.dim == 3) ? 4 : 8, tz = (mesh.dim == 3) ? 2 : 1;
    const int64_t ntiles = ceil_div64(mesh.px, tx) * ceil_div64(mesh.py, ty) * ceil_div64(mesh.pzl, tz);
    PPH_REQUIRE(ctx, ntiles < (int64_t)1 << 31, "tile assembly: too many tiles for 32-bit tile indices");
    int grid = (int)(ntiles < 256 * 16 ? ntiles : 256 * 16);
    const int xmap = (ctx->asm_tile_xmap && grid >= 64) ? 1 : 0;
    if (xmap) grid &= ~7;   // (the XCD-contiguous tile order wants a multiple of 8 workgroups)
    if (mesh.kind == PPH_CELL_QUAD)
      hipLaunchKernelGGL(k_asm_tile<2>, dim3(grid), dim3(512), 0, ctx->stream, mesh.cx.p, mesh.cy.p, mesh.cz.p, mesh.rowptr.p,
                         Kp, Mp, mesh.nx, mesh.ny, 0, mesh.px, mesh.py, 1, mesh.n, fa, ctx->asm_tile_probe ? ctx->asm_tile_probe : (ctx->asm_affine ? 0 : 6), xmap);
    else
      hipLaunchKernelGGL(k_asm_tile<3>, dim3(grid), dim3(512), 0, ctx->stream, mesh.cx.p, mesh.cy.p, mesh.cz.p, mesh.rowptr.p,
                         Kp, Mp, mesh.nx, mesh.ny, mesh.nzl, mesh.px, mesh.py, mesh.pzl, mesh.n, fa, ctx->asm_tile_probe ? ctx->asm_tile_probe : (ctx->asm_affine ? 0 : 6), xmap);
    PPH_HIP(ctx, hipGetLastError());
    return PPH_OK;
  }
  const int cpb = 256 / mesh.m;
  // Optional layered schedule (3D, `asm_ring` > 0): element pass on L cell layers, then the node planes they
  // complete, with a ring of L + 1 layers of element rows meant to stay in the memory-side cache.  Measured at 256^3:
  // 62 ms with one layer (64 K cells) per launch (514 short dependent launches), 35 / 24 / 19 / 18 ms with 4 / 8 / 16 /
  // 32 layers per launch, against 16.8 ms for the whole mesh in two launches - the default.
  const int64_t layer_cells = (mesh.dim == 3) ? (int64_t)mesh.nx * mesh.ny : mesh.ncell;
  const int64_t plane = (mesh.dim == 3) ? (int64_t)mesh.px * mesh.py : mesh.n;
  const int layers = (mesh.dim == 3) ? mesh.nzl : 1;
  int L = (int)ceil_div64((int64_t)ctx->asm_ring, layer_cells);   // asm_ring: cells per launch
  if (L < 1) L = 1;
  // (a workgroup batch of the element pass must not straddle two layers: their ring slots need not be adjacent)
  const bool ringed = ctx->asm_ring && mesh.dim == 3 && L + 1 < layers && layer_cells % cpb == 0;
  if (!ringed) L = layers;
  const int ring = ringed ? L + 1 : 1;
  const int64_t ring_cells = ringed ? (int64_t)ring * layer_cells : mesh.ncell;
  PPH_TRY(mesh.erows.alloc(ctx, (size_t)ring_cells * mesh.m * 2 * mesh.m));
  const int nplanes = (mesh.dim == 3) ? mesh.pzl : 1;
  for (int k0 = 0; k0 < nplanes; k0 += L) {
    // cells of layers [k0, k0 + L) (none left for the topmost node plane), then node planes [k0, k0 + L)
    const int k1 = (k0 + L < layers) ? k0 + L : layers;
    const int p1 = (k0 + L < nplanes) ? k0 + L : nplanes;
    ERing ec{(int64_t)k0 * layer_cells, (int64_t)k1 * layer_cells, layer_cells, ringed ? ring : 1};
    ERing en{(int64_t)k0 * plane, (int64_t)p1 * plane, layer_cells, ringed ? ring : 1};
    if (!ringed) { ec = ERing{0, mesh.ncell, mesh.ncell, 1}; en = ERing{0, mesh.n, mesh.ncell, 1}; }
    const int64_t nb1 = ceil_div64(ec.end - ec.begin, cpb), nb2 = ceil_div64(en.end - en.begin, cpb);
    const int g1 = (int)(nb1 < 256 * 16 ? nb1 : 256 * 16), g2 = (int)(nb2 < 256 * 16 ? nb2 : 256 * 16);
    if (mesh.kind == PPH_CELL_QUAD) {
      hipLaunchKernelGGL(k_elem_rows<2>, dim3(g1), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p, mesh.cy.p, mesh.cz.p,
                         mesh.erows.p, mesh.ncell, ec);
      if (mesh.px >= 3 && mesh.py >= 3)
        hipLaunchKernelGGL((k_gather_rows<2, true, true>), dim3(g2), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.erows.p,
                           mesh.rowptr.p, mesh.col.p, Kp, Mp, mesh.nx, mesh.ny, 0, mesh.px, mesh.py, mesh.n, fa, en);
      else
        hipLaunchKernelGGL((k_gather_rows<2, true, false>), dim3(g2), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.erows.p,
                           mesh.rowptr.p, mesh.col.p, Kp, Mp, mesh.nx, mesh.ny, 0, mesh.px, mesh.py, mesh.n, fa, en);
    } else {
      if (ec.end > ec.begin)
        hipLaunchKernelGGL(k_elem_rows<3>, dim3(g1), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.cx.p, mesh.cy.p,
                           mesh.cz.p, mesh.erows.p, mesh.ncell, ec);
      if (mesh.px >= 3 && mesh.py >= 3 && mesh.nzl >= 2)
        hipLaunchKernelGGL((k_gather_rows<3, true, true>), dim3(g2), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.erows.p,
                           mesh.rowptr.p, mesh.col.p, Kp, Mp, mesh.nx, mesh.ny, mesh.nzl, mesh.px, mesh.py, mesh.n, fa, en);
      else
        hipLaunchKernelGGL((k_gather_rows<3, true, false>), dim3(g2), dim3(256), 0, ctx->stream, mesh.cells.p, mesh.erows.p,
                           mesh.rowptr.p, mesh.col.p, Kp, Mp, mesh.nx, mesh.ny, mesh.nzl, mesh.px, mesh.py, mesh.n, fa, en);
    }
    if (!ringed) break;
  }
  PPH_HIP(ctx, hipGetLastError());
  return PPH_OK;
}

// Dirichlet-eliminated operators coefK[f] K + coefM M (f = 0, 1) of a multigrid level straight from the element
// rows, with the smoother's diagonal inverses and spectral bounds (no K/M, no coupling blocks, no lifting)
int pph_launch_level_operators(pph_ctx* ctx, MeshData& mesh, const uint8_t* m1, const uint8_t* m2, const uint8_t* near,
                               int same, double coefK1, double coefK2, double coefM, double* A1, double* A2,
                               double* dinv1, double* dinv2, unsigned long long* lam, int64_t ell_ld, int ell_sym,
                               DictGroup* group, SellDict* dicts, const Sell* views) {
  FuseArgs fa;
  if (group && dicts && views) {
    fa.G = group;
    for (int d = 0; d < 2; ++d) { fa.dicts[d] = &dicts[d]; fa.views[d] = &views[d]; }
  }
  fuse_set_format(fa, mesh.kind, ell_ld, ell_sym, ell_sym);
  fa.symg = (ell_sym && ctx->world > 1) ? 1 : 0;
  fa.m1 = m1; fa.m2 = m2; fa.near = near;
  fa.g1 = nullptr; fa.g2 = nullptr;
  fa.a = coefK1; fa.b = coefM; fa.c = coefK2;
  fa.A11 = A1; fa.A22 = A2; fa.A12 = nullptr; fa.A21 = nullptr;
  fa.rhs = nullptr; fa.u0 = nullptr;
  fa.dinv1 = dinv1; fa.dinv2 = dinv2;
  fa.lam = lam;
  fa.keep_km = 0;
  fa.same = same;
  return pph_launch_fused_kernels(ctx, mesh, fa, nullptr, nullptr);
}

void pph_launch_row_near(pph_ctx* ctx, const MeshData& mesh, const uint8_t* m1, const uint8_t* m2, uint8_t* out) {
  const int64_t nbn = ceil_div64(mesh.n, 256);
  hipLaunchKernelGGL(k_row_near, dim3((int)(nbn < 4096 ? (nbn < 1 ? 1 : nbn) : 4096)), dim3(256), 0, ctx->stream,
                     make_stencil(mesh.kind), mesh.px, mesh.py, mesh.pzl, m1, m2, mesh.n, out);
}

int pph_launch_assemble_fused(pph_ctx* ctx, int monolithic) {
  MeshData& mesh = ctx->mesh;
  const int64_t n = ctx->n;
  const bool ell = ctx->op_format == 1;
  PPH_TRY(blocks_prepare(ctx, !ell));
  if (ell) PPH_TRY(blocks_alloc_sell(ctx));
  if (ctx->asm_keep_km) {
    PPH_TRY(mesh.K.alloc(ctx, (size_t)mesh.nnzb));
    PPH_TRY(mesh.M.alloc(ctx, (size_t)mesh.nnzb));
  }
  PPH_TRY(ctx->dinv0[0].alloc(ctx, (size_t)n));
  PPH_TRY(ctx->dinv0[1].alloc(ctx, (size_t)n));
  PPH_TRY(ctx->lam0.alloc(ctx, 2));
  PPH_HIP(ctx, hipMemsetAsync(ctx->lam0.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
  FuseArgs fa;
  fa.m1 = ctx->bcmask[0].p; fa.m2 = ctx->bcmask[1].p; fa.near = ctx->rownear.p;
  fa.g1 = ctx->g[0].p; fa.g2 = ctx->g[1].p;
  fa.a = ctx->a; fa.b = ctx->b; fa.c = ctx->c;
  if (ell) {
    fa.A11 = ctx->E11.p; fa.A22 = ctx->E22.p; fa.A12 = ctx->E12.p; fa.A21 = ctx->a21_alias ? nullptr : ctx->E21.p;
  } else {
    fa.A11 = ctx->A11.p; fa.A22 = ctx->A22.p; fa.A12 = ctx->A12.p; fa.A21 = ctx->a21_alias ? nullptr : ctx->A21.p;
  }
  fuse_set_format(fa, mesh.kind, ell ? ctx->S11.ld : 0, ell ? ctx->S11.sym : 0, ell ? ctx->S12.sym : 0);
  fa.symg = (ell && ctx->S11.sym && ctx->world > 1) ? 1 : 0;
  fa.rhs = ctx->rhs.p; fa.u0 = ctx->u0.p;
  fa.dinv1 = ctx->dinv0[0].p; fa.dinv2 = ctx->dinv0[1].p;
  fa.lam = ctx->lam0.p;
  fa.keep_km = ctx->asm_keep_km;
  fa.same = ctx->a21_alias ? 1 : 0;
  if (ell) {
    fa.G = &ctx->DG;
    fa.dicts[0] = &ctx->D11; fa.dicts[1] = &ctx->D22; fa.dicts[2] = &ctx->D12;
    fa.views[0] = &ctx->S11; fa.views[1] = &ctx->S22; fa.views[2] = &ctx->S12;
  }
  PPH_TRY(pph_launch_fused_kernels(ctx, mesh, fa, ctx->asm_keep_km ? mesh.K.p : nullptr,
                                   ctx->asm_keep_km ? mesh.M.p : nullptr));
  PPH_HIP(ctx, hipGetLastError());
  if (ell) {
    // row dictionaries of the three stored blocks (sell_dict): built on the first assembly; on the next ones checked by the
    // assembly kernel itself (the group below) or, failing that, re-read and checked here
    const int b0 = ctx->n_dict_build;
    PPH_TRY(sell_dict_update(ctx, &ctx->S11, ctx->D11, n));
    PPH_TRY(sell_dict_update(ctx, &ctx->S22, ctx->D22, n));
    PPH_TRY(sell_dict_update(ctx, &ctx->S12, ctx->D12, n));
    if (ctx->a21_alias) ctx->S21 = ctx->S12;
    if (ctx->n_dict_build != b0) {     // dictionaries (re)built: their fused-check group follows
      SellDict* ds[3] = {ctx->D11.on ? &ctx->D11 : nullptr, ctx->D22.on ? &ctx->D22 : nullptr, ctx->D12.on ? &ctx->D12 : nullptr};
      ctx->DG.release();
      if (ds[0] && ds[1]) PPH_TRY(dict_group_build(ctx, ctx->DG, ds, ds[2] ? 3 : 2, ctx->S11, n));
    }
  }
  mesh.km_valid = ctx->asm_keep_km != 0;
  ctx->diag0_valid = true;
  ctx->ell_ok = ell;
  ctx->csr_ok = !ell;
  return blocks_mono(ctx, monolithic);
}

// one Dirichlet-eliminated scalar operator coefK*K + coefM*M on any level (multigrid coarse operators)
__global__ __launch_bounds__(256) void k_scalar_block(const int64_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col, const double* __restrict__ K,
                                                      const double* __restrict__ M, const uint8_t* __restrict__ mask,
                                                      double coefK, double coefM, int64_t n, double* __restrict__ out) {
  const int sub = threadIdx.x % BC_LANES;
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / BC_LANES; row < n;
       row += ((int64_t)gridDim.x * blockDim.x) / BC_LANES) {
    const uint8_t r = mask[row];
    for (int64_t k = rowptr[row] + sub; k < rowptr[row + 1]; k += BC_LANES) {
      const int32_t j = col[k];
      const bool diag = (j == (int32_t)row);
      out[k] = (r & 2) ? 0.0 : (r & 1) ? (diag ? 1.0 : 0.0) : (((mask[j] & 1) != 0) ? 0.0 : coefK * K[k] + coefM * M[k]);
    }
  }
}

void pph_launch_scalar_block(pph_ctx* ctx, const MeshData& mesh, const uint8_t* mask, double coefK, double coefM,
                             double* out) {
  int64_t nb = ceil_div64(mesh.n * BC_LANES, 256);
  int grid = (int)(nb < 256 * 16 ? nb : 256 * 16);
  hipLaunchKernelGGL(k_scalar_block, dim3(grid), dim3(256), 0, ctx->stream, mesh.rowptr.p, mesh.col.p, mesh.K.p,
                     mesh.M.p, mask, coefK, coefM, mesh.n, out);
}
