/*
 * ORACLE — test infrastructure only.  NOT part of the product path.
 *
 * C99 + OpenMP restatement of the double-porosity/permeability hot path on the host CPU, independent of
 * the NumPy oracle (oracle/dpp_oracle.py, oracle/dpp_mg_oracle.py) and of the HIP product
 * (perphil_amd/csrc).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * What it follows in the reference (ThermoPhase-FCSRG/perphil):
 *   operator     src/perphil/forms/dpp.py:27,57-58,89-90,129-130   A = [[aK+bM, -bM], [-bM, cK+bM]], L = 0
 *   coefficients src/perphil/models/dpp/parameters.py:26-52        a = k1/mu, b = beta/mu, c = k2/mu
 *   Picard split src/perphil/forms/dpp.py:196-203                  A11 du1 = b1 - A12 du2 ; A22 du2 = b2 - A21 du1
 *   BC handling  Firedrake semantics at src/perphil/solvers/solver.py:66 (rows+columns zeroed, unit diagonal,
 *                rhs = -(A u0) on free rows), tolerances src/perphil/solvers/parameters.py:15-17,74-75
 *   meshes       src/perphil/mesh/builtin.py:4-20 and fd.UnitCubeMesh call sites (lexicographic numbering,
 *                left-diagonal triangles, 6 Kuhn tetrahedra per cube sharing the diagonal v0-v7)
 * The scalar block solves use the geometric multigrid restated in oracle/dpp_mg_oracle.py (the reference
 * uses MUMPS LU there; see that file's header).  Pinned by the reference's goldens in
 * tests/test_cpu_port.py (G1 initial residual, G2/G11 solution slice) and entry-for-entry by the NumPy oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

enum { QUAD = 0, TRI = 1, HEX = 2, TET = 3 };
enum { MAT_K = 1, MAT_M = 2, MAT_A11 = 3, MAT_A22 = 4, MAT_A12 = 5, MAT_A21 = 6 };
enum { PC_NONE = 0, PC_JACOBI = 1, PC_MG = 4 };

typedef struct {
  int dim, kind, m, nx, ny, nz, px, py, pz, nsub;
  int64_t n, ncell, nnz;
  double* xyz;      /* [n][3] */
  int32_t* cells;   /* [ncell][m] */
  int64_t* rowptr;
  int32_t* col;
  double *K, *M;
  int nst;
  int st[27][3];
} cmesh;

typedef struct {
  cmesh* mesh;      /* level 0 aliases the system mesh */
  int owns_mesh;
  double* A;        /* eliminated operator values on the mesh pattern */
  double* dinv;
  unsigned char* mask;
  double lam;
  double *x, *b, *r, *d, *t;            /* work vectors of the cycle */
  double *cr, *cz, *cp, *cq;            /* coarsest-level CG */
} clevel;

typedef struct {
  int nlev;
  clevel* lv;
} chier;

typedef struct {
  cmesh mesh;
  unsigned char* mask[2];
  double* g;        /* [2n] boundary values (0 inside) */
  double *A11, *A22, *A12, *A21, *rhs;
  double a, b, c;
  int assembled;
  chier H[2];
  cmesh* coarse[32];   /* coarse meshes with their K and M, shared by the two hierarchies */
  int ncoarse;
} csys;

/* ------------------------------------------------------------------------------------------------ mesh */
static int make_stencil(int kind, int st[27][3]) {
  int c = 0;
  int zlo = (kind == HEX || kind == TET) ? -1 : 0, zhi = -zlo;
  for (int dz = zlo; dz <= zhi; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        int keep = 1;
        if (kind == TRI) keep = (dx == 0 || dy == 0 || dx == -dy);        /* edges along x, y and the (1,-1) diagonal */
        if (kind == TET) keep = (dx >= 0 && dy >= 0 && dz >= 0) || (dx <= 0 && dy <= 0 && dz <= 0);
        if (keep) { st[c][0] = dx; st[c][1] = dy; st[c][2] = dz; ++c; }
      }
  return c;
}

static void mesh_free(cmesh* m) {
  free(m->xyz); free(m->cells); free(m->rowptr); free(m->col); free(m->K); free(m->M);
  memset(m, 0, sizeof(*m));
}

static int mesh_build(cmesh* M, int dim, int kind, int nx, int ny, int nz) {
  memset(M, 0, sizeof(*M));
  if (!((dim == 2 && (kind == QUAD || kind == TRI)) || (dim == 3 && (kind == HEX || kind == TET)))) return -1;
  if (nx < 1 || ny < 1 || (dim == 3 && nz < 1)) return -1;
  if (dim == 2) nz = 0;
  M->dim = dim; M->kind = kind; M->nx = nx; M->ny = ny; M->nz = nz;
  M->px = nx + 1; M->py = ny + 1; M->pz = dim == 3 ? nz + 1 : 1;
  M->m = kind == QUAD ? 4 : kind == TRI ? 3 : kind == HEX ? 8 : 4;
  M->nsub = kind == TRI ? 2 : kind == TET ? 6 : 1;
  M->n = (int64_t)M->px * M->py * M->pz;
  const int64_t nbox = (int64_t)nx * ny * (dim == 3 ? nz : 1);
  M->ncell = nbox * M->nsub;
  M->xyz = (double*)malloc(sizeof(double) * 3 * (size_t)M->n);
  M->cells = (int32_t*)malloc(sizeof(int32_t) * (size_t)M->ncell * M->m);
  M->rowptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(M->n + 1));
  if (!M->xyz || !M->cells || !M->rowptr) return -2;
  const int px = M->px, py = M->py, pz = M->pz;
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < M->n; ++v) {
    const int i = (int)(v % px), j = (int)((v / px) % py), k = (int)(v / ((int64_t)px * py));
    M->xyz[3 * v + 0] = (double)i / nx;
    M->xyz[3 * v + 1] = (double)j / ny;
    M->xyz[3 * v + 2] = dim == 3 ? (double)k / nz : 0.0;
  }
  static const int tets[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
  static const int tris[2][3] = {{0, 1, 2}, {1, 3, 2}};
#pragma omp parallel for schedule(static)
  for (int64_t bx = 0; bx < nbox; ++bx) {
    const int i = (int)(bx % nx), j = (int)((bx / nx) % ny), k = (int)(bx / ((int64_t)nx * ny));
    int32_t v[8];
    const int64_t v0 = i + (int64_t)px * (j + (int64_t)py * k);
    for (int c = 0; c < 8; ++c)
      v[c] = (int32_t)(v0 + (c & 1) + ((c >> 1) & 1) * (int64_t)px + ((c >> 2) & 1) * (int64_t)px * py);
    int32_t* out = M->cells + (size_t)bx * M->nsub * M->m;
    if (kind == QUAD) for (int c = 0; c < 4; ++c) out[c] = v[c];
    else if (kind == HEX) for (int c = 0; c < 8; ++c) out[c] = v[c];
    else if (kind == TRI) for (int s = 0; s < 2; ++s) for (int c = 0; c < 3; ++c) out[3 * s + c] = v[tris[s][c]];
    else for (int s = 0; s < 6; ++s) for (int c = 0; c < 4; ++c) out[4 * s + c] = v[tets[s][c]];
  }
  M->nst = make_stencil(kind, M->st);
  /* sparsity pattern from the structured stencil, columns ascending */
  M->rowptr[0] = 0;
  for (int64_t v = 0; v < M->n; ++v) {
    const int i = (int)(v % px), j = (int)((v / px) % py), k = (int)(v / ((int64_t)px * py));
    int c = 0;
    for (int s = 0; s < M->nst; ++s) {
      const int ii = i + M->st[s][0], jj = j + M->st[s][1], kk = k + M->st[s][2];
      c += (ii >= 0 && ii < px && jj >= 0 && jj < py && kk >= 0 && kk < pz);
    }
    M->rowptr[v + 1] = M->rowptr[v] + c;
  }
  M->nnz = M->rowptr[M->n];
  M->col = (int32_t*)malloc(sizeof(int32_t) * (size_t)M->nnz);
  if (!M->col) return -2;
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < M->n; ++v) {
    const int i = (int)(v % px), j = (int)((v / px) % py), k = (int)(v / ((int64_t)px * py));
    int64_t p = M->rowptr[v];
    for (int s = 0; s < M->nst; ++s) {
      const int ii = i + M->st[s][0], jj = j + M->st[s][1], kk = k + M->st[s][2];
      if (ii >= 0 && ii < px && jj >= 0 && jj < py && kk >= 0 && kk < pz)
        M->col[p++] = (int32_t)(ii + (int64_t)px * (jj + (int64_t)py * kk));
    }
  }
  return 0;
}

/* --------------------------------------------------------------------------------- element integrals */
static double det_inv(int d, const double J[3][3], double I[3][3]) {
  if (d == 2) {
    const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    I[0][0] = J[1][1] / det;  I[0][1] = -J[0][1] / det;
    I[1][0] = -J[1][0] / det; I[1][1] = J[0][0] / det;
    return det;
  }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  I[0][0] = c00 / det;
  I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
  I[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
  I[1][0] = c01 / det;
  I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
  I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  I[2][0] = c02 / det;
  I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
  I[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  return det;
}

/* K_e = int grad phi_a . grad phi_b, M_e = int phi_a phi_b for one cell with vertex coordinates X[m][3] */
static void element_matrices(int dim, int kind, const double X[8][3], double* Ke, double* Me) {
  const int m = kind == QUAD ? 4 : kind == TRI ? 3 : kind == HEX ? 8 : 4;
  for (int t = 0; t < m * m; ++t) Ke[t] = Me[t] = 0.0;
  if (kind == QUAD || kind == HEX) {
    const double gp = 1.0 / sqrt(3.0);
    for (int q = 0; q < (1 << dim); ++q) {
      double xi[3], N[8], dN[8][3], J[3][3] = {{0}}, I[3][3], G[8][3];
      for (int e = 0; e < dim; ++e) xi[e] = ((q >> e) & 1) ? gp : -gp;
      for (int a = 0; a < m; ++a) {
        N[a] = 1.0;
        for (int e = 0; e < dim; ++e) dN[a][e] = 1.0;
        for (int c = 0; c < dim; ++c) {
          const double s = ((a >> c) & 1) ? 1.0 : -1.0;
          N[a] *= 0.5 * (1.0 + s * xi[c]);
          for (int e = 0; e < dim; ++e) dN[a][e] *= (e == c) ? 0.5 * s : 0.5 * (1.0 + s * xi[c]);
        }
      }
      for (int e = 0; e < dim; ++e)
        for (int d = 0; d < dim; ++d)
          for (int a = 0; a < m; ++a) J[e][d] += dN[a][e] * X[a][d];      /* J[e][d] = d x_d / d xi_e */
      const double det = det_inv(dim, J, I);
      /* physical gradient: grad_d = sum_e (J^-1)[d][e] dN_e with J^-1 the inverse of the matrix J[e][d] */
      for (int a = 0; a < m; ++a)
        for (int d = 0; d < dim; ++d) {
          double g = 0.0;
          for (int e = 0; e < dim; ++e) g += I[d][e] * dN[a][e];
          G[a][d] = g;
        }
      for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b) {
          double gg = 0.0;
          for (int d = 0; d < dim; ++d) gg += G[a][d] * G[b][d];
          Ke[a * m + b] += det * gg;
          Me[a * m + b] += det * N[a] * N[b];
        }
    }
    return;
  }
  /* simplices: constant gradients, closed-form mass matrix */
  double E[3][3] = {{0}}, I[3][3], G[4][3];
  for (int r = 0; r < dim; ++r)
    for (int d = 0; d < dim; ++d) E[r][d] = X[r + 1][d] - X[0][d];
  const double det = det_inv(dim, E, I);
  const double vol = fabs(det) / (dim == 2 ? 2.0 : 6.0);
  for (int d = 0; d < dim; ++d) {
    G[0][d] = 0.0;
    for (int r = 0; r < dim; ++r) { G[r + 1][d] = I[d][r]; G[0][d] -= I[d][r]; }   /* columns of E^-1 */
  }
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      double gg = 0.0;
      for (int d = 0; d < dim; ++d) gg += G[a][d] * G[b][d];
      Ke[a * m + b] = vol * gg;
      Me[a * m + b] = vol / ((dim + 1) * (dim + 2)) * (a == b ? 2.0 : 1.0);
    }
}

/* scalar K and M on the mesh pattern.  Plane sweep along the slowest direction: the element matrices of
 * cell layer k are integrated (in parallel) into one of two layer buffers, then every row of node plane k
 * gathers its entries from the incident cells of layers k-1 and k in a fixed order (deterministic, no
 * atomics, working set of two cell layers). */
static int assemble_KM(cmesh* M) {
  const int m = M->m, dim = M->dim, mm = m * m;
  if (!M->K) M->K = (double*)malloc(sizeof(double) * (size_t)M->nnz);
  if (!M->M) M->M = (double*)malloc(sizeof(double) * (size_t)M->nnz);
  const int px = M->px, py = M->py, nx = M->nx, ny = M->ny, nlay = dim == 3 ? M->nz : 1;
  const int64_t lay_boxes = (int64_t)nx * ny, lay_cells = lay_boxes * M->nsub, plane = (int64_t)px * py;
  double* buf[2];
  buf[0] = (double*)malloc(sizeof(double) * (size_t)lay_cells * mm * 2);
  buf[1] = (double*)malloc(sizeof(double) * (size_t)lay_cells * mm * 2);
  if (!M->K || !M->M || !buf[0] || !buf[1]) { free(buf[0]); free(buf[1]); return -2; }
  for (int k = 0; k < M->pz; ++k) {
    if (k < nlay) {
      double* B = buf[k & 1];
#pragma omp parallel for schedule(static)
      for (int64_t lc = 0; lc < lay_cells; ++lc) {
        const int64_t c = (int64_t)k * lay_cells + lc;
        double X[8][3];
        for (int a = 0; a < m; ++a)
          for (int d = 0; d < 3; ++d) X[a][d] = M->xyz[3 * (size_t)M->cells[c * m + a] + d];
        element_matrices(dim, M->kind, X, B + (size_t)lc * mm * 2, B + (size_t)lc * mm * 2 + mm);
      }
    }
#pragma omp parallel for schedule(static)
    for (int64_t pv = 0; pv < plane; ++pv) {
      const int64_t v = (int64_t)k * plane + pv;
      const int i = (int)(pv % px), j = (int)(pv / px);
      const int64_t r0 = M->rowptr[v], r1 = M->rowptr[v + 1];
      for (int64_t p = r0; p < r1; ++p) M->K[p] = M->M[p] = 0.0;
      for (int cb = 0; cb < (1 << dim); ++cb) {
        const int bi = i - (cb & 1), bj = j - ((cb >> 1) & 1), bk = dim == 3 ? k - ((cb >> 2) & 1) : 0;
        if (bi < 0 || bi >= nx || bj < 0 || bj >= ny || bk < 0 || bk >= nlay) continue;
        const int64_t lbox = bi + (int64_t)nx * bj;
        const double* B = buf[bk & 1];
        for (int s = 0; s < M->nsub; ++s) {
          const int64_t lc = lbox * M->nsub + s;
          const int32_t* cn = M->cells + (size_t)((int64_t)bk * lay_cells + lc) * m;
          int a = -1;
          for (int t = 0; t < m; ++t) if (cn[t] == (int32_t)v) a = t;
          if (a < 0) continue;
          const double* Ke = B + (size_t)lc * mm * 2 + a * m;
          const double* Me = Ke + mm;
          for (int b = 0; b < m; ++b) {
            int64_t p = r0;
            while (p < r1 && M->col[p] != cn[b]) ++p;
            M->K[p] += Ke[b];
            M->M[p] += Me[b];
          }
        }
      }
    }
  }
  free(buf[0]); free(buf[1]);
  return 0;
}

/* ---------------------------------------------------------------------------------- linear algebra */
static void spmv(const cmesh* M, const double* val, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < M->n; ++r) {
    double s = 0.0;
    for (int64_t p = M->rowptr[r]; p < M->rowptr[r + 1]; ++p) s += val[p] * x[M->col[p]];
    y[r] = s;
  }
}
/* y = b - A x */
static void resid(const cmesh* M, const double* val, const double* x, const double* b, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < M->n; ++r) {
    double s = 0.0;
    for (int64_t p = M->rowptr[r]; p < M->rowptr[r + 1]; ++p) s += val[p] * x[M->col[p]];
    y[r] = b[r] - s;
  }
}
static double dot(int64_t n, const double* x, const double* y) {
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int64_t i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}
static void axpy(int64_t n, double a, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] += a * x[i];
}
static void xpby(int64_t n, const double* x, double b, double* y) { /* y = x + b y */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] = x[i] + b * y[i];
}
static void vcopy(int64_t n, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] = x[i];
}
static void vzero(int64_t n, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) y[i] = 0.0;
}

/* eliminated block: val = cK*K + cM*M with rows in rmask and columns in cmask zeroed; unit diagonal on
 * rmask rows when `diag` (square blocks) */
static void make_block(const cmesh* M, double cK, double cM, const unsigned char* rmask, const unsigned char* cmask,
                       int diag, double* val) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < M->n; ++r)
    for (int64_t p = M->rowptr[r]; p < M->rowptr[r + 1]; ++p) {
      const int32_t c = M->col[p];
      double v = cK * M->K[p] + cM * M->M[p];
      if (rmask[r] || cmask[c]) v = (diag && c == r) ? 1.0 : 0.0;
      val[p] = v;
    }
}

/* ------------------------------------------------------------------------------------- multigrid */
static double transfer_weight(int kind, const int* d) {
  const int nzc = (d[0] != 0) + (d[1] != 0) + (d[2] != 0);
  if (kind == QUAD || kind == HEX) return ldexp(1.0, -nzc);
  return nzc == 0 ? 1.0 : 0.5;
}

static void restrict_to(const cmesh* F, const cmesh* C, const double* rf, double* rc) {
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < C->n; ++v) {
    const int i = (int)(v % C->px), j = (int)((v / C->px) % C->py), k = (int)(v / ((int64_t)C->px * C->py));
    double s = 0.0;
    for (int t = 0; t < F->nst; ++t) {
      const int fi = 2 * i + F->st[t][0], fj = 2 * j + F->st[t][1], fk = 2 * k + F->st[t][2];
      if (fi < 0 || fi >= F->px || fj < 0 || fj >= F->py || fk < 0 || fk >= F->pz) continue;
      s += transfer_weight(F->kind, F->st[t]) * rf[fi + (int64_t)F->px * (fj + (int64_t)F->py * fk)];
    }
    rc[v] = s;
  }
}

/* xf += P xc on unconstrained fine nodes */
static void prolong_add(const cmesh* F, const cmesh* C, const double* xc, const unsigned char* fmask, double* xf) {
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < F->n; ++v) {
    if (fmask[v]) continue;
    const int i = (int)(v % F->px), j = (int)((v / F->px) % F->py), k = (int)(v / ((int64_t)F->px * F->py));
    double s = 0.0;
    for (int t = 0; t < F->nst; ++t) {
      const int ci = i - F->st[t][0], cj = j - F->st[t][1], ck = k - F->st[t][2];
      if ((ci & 1) || (cj & 1) || (ck & 1) || ci < 0 || cj < 0 || ck < 0) continue;
      const int I = ci / 2, J = cj / 2, K = ck / 2;
      if (I >= C->px || J >= C->py || K >= C->pz) continue;
      s += transfer_weight(F->kind, F->st[t]) * xc[I + (int64_t)C->px * (J + (int64_t)C->py * K)];
    }
    xf[v] += s;
  }
}

static void hier_free(chier* H) {
  for (int l = 0; l < H->nlev; ++l) {
    clevel* L = &H->lv[l];
    free(L->A); free(L->dinv); free(L->mask);
    free(L->x); free(L->b); free(L->r); free(L->d); free(L->t);
    free(L->cr); free(L->cz); free(L->cp); free(L->cq);
  }
  free(H->lv);
  H->lv = NULL; H->nlev = 0;
}

static int hier_build(csys* S, chier* H, cmesh* fine, double cK, double cM, const unsigned char* mask_fine,
                      int min_cells) {
  hier_free(H);
  H->lv = (clevel*)calloc(32, sizeof(clevel));
  cmesh* M = fine;
  const unsigned char* mask = mask_fine;
  unsigned char* injected = NULL;   /* mask of the level being built when it is not the caller's */
  for (;;) {
    clevel* L = &H->lv[H->nlev];
    L->mesh = M; L->owns_mesh = 0;
    const int64_t n = M->n;
    L->mask = (unsigned char*)malloc((size_t)n);
    memcpy(L->mask, mask, (size_t)n);
    free(injected);
    injected = NULL;
    L->A = (double*)malloc(sizeof(double) * (size_t)M->nnz);
    L->dinv = (double*)malloc(sizeof(double) * (size_t)n);
    make_block(M, cK, cM, L->mask, L->mask, 1, L->A);
    double lam = 0.0;
#pragma omp parallel for schedule(static) reduction(max : lam)
    for (int64_t r = 0; r < n; ++r) {
      double s = 0.0, d = 1.0;
      for (int64_t p = M->rowptr[r]; p < M->rowptr[r + 1]; ++p) {
        s += fabs(L->A[p]);
        if (M->col[p] == r) d = L->A[p];
      }
      L->dinv[r] = 1.0 / d;
      if (s / d > lam) lam = s / d;
    }
    L->lam = lam;
    L->x = (double*)malloc(sizeof(double) * (size_t)n); L->b = (double*)malloc(sizeof(double) * (size_t)n);
    L->r = (double*)malloc(sizeof(double) * (size_t)n); L->d = (double*)malloc(sizeof(double) * (size_t)n);
    L->t = (double*)malloc(sizeof(double) * (size_t)n);
    H->nlev++;
    int mn = M->nx < M->ny ? M->nx : M->ny;
    if (M->dim == 3 && M->nz < mn) mn = M->nz;
    const int can = M->nx % 2 == 0 && M->ny % 2 == 0 && (M->dim == 2 || M->nz % 2 == 0) && mn / 2 >= min_cells;
    if (!can || H->nlev == 32) {
      L->cr = (double*)malloc(sizeof(double) * (size_t)n); L->cz = (double*)malloc(sizeof(double) * (size_t)n);
      L->cp = (double*)malloc(sizeof(double) * (size_t)n); L->cq = (double*)malloc(sizeof(double) * (size_t)n);
      break;
    }
    const int ci = H->nlev - 1;       /* index of the coarse mesh below level nlev-1 */
    if (ci >= S->ncoarse) {
      cmesh* N = (cmesh*)calloc(1, sizeof(cmesh));
      if (mesh_build(N, M->dim, M->kind, M->nx / 2, M->ny / 2, M->dim == 3 ? M->nz / 2 : 0)) return -2;
      if (assemble_KM(N)) return -2;
      S->coarse[S->ncoarse++] = N;
    }
    cmesh* C = S->coarse[ci];
    unsigned char* cm = (unsigned char*)malloc((size_t)C->n);   /* injected mask: coarse C <- fine 2C */
    for (int64_t v = 0; v < C->n; ++v) {
      const int i = (int)(v % C->px), j = (int)((v / C->px) % C->py), k = (int)(v / ((int64_t)C->px * C->py));
      cm[v] = L->mask[2 * i + (int64_t)M->px * (2 * j + (int64_t)M->py * 2 * k)];
    }
    M = C;
    mask = injected = cm;
  }
  return 0;
}

#define CHEB_LOWER 0.25
/* `steps` Chebyshev-Jacobi steps on A x = b; zero_guess: x is taken as 0 and overwritten */
static void chebyshev(clevel* L, const double* b, double* x, int steps, int zero_guess) {
  const cmesh* M = L->mesh;
  const int64_t n = M->n;
  const double lo = CHEB_LOWER * L->lam, hi = L->lam;
  const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
  double rho = 1.0 / sigma;
  double *r = L->r, *d = L->d;
  if (zero_guess) { vcopy(n, b, r); vzero(n, x); }
  else resid(M, L->A, x, b, r);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) { d[i] = L->dinv[i] * r[i] / theta; x[i] += d[i]; }
  for (int s = 1; s < steps; ++s) {
    spmv(M, L->A, d, L->t);
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      r[i] -= L->t[i];
      d[i] = c1 * d[i] + c2 * (L->dinv[i] * r[i]);
      x[i] += d[i];
    }
    rho = rho_new;
  }
}

/* Jacobi-PCG on the coarsest level: rtol 1e-12 on the preconditioned residual, at most 500 iterations */
static void coarse_solve(clevel* L, const double* b, double* x) {
  const cmesh* M = L->mesh;
  const int64_t n = M->n;
  double *r = L->cr, *z = L->cz, *p = L->cp, *q = L->cq;
  vzero(n, x); vcopy(n, b, r);
  for (int64_t i = 0; i < n; ++i) z[i] = L->dinv[i] * r[i];
  double res = sqrt(dot(n, z, z));
  const double tol = fmax(1e-12 * res, 1e-300);
  if (res <= tol) return;
  vcopy(n, z, p);
  double rz = dot(n, r, z);
  for (int it = 0; it < 500; ++it) {
    spmv(M, L->A, p, q);
    const double alpha = rz / dot(n, p, q);
    axpy(n, alpha, p, x);
    axpy(n, -alpha, q, r);
    for (int64_t i = 0; i < n; ++i) z[i] = L->dinv[i] * r[i];
    res = sqrt(dot(n, z, z));
    if (res <= tol) return;
    const double rzn = dot(n, r, z);
    xpby(n, z, rzn / rz, p);
    rz = rzn;
  }
}

static void vcycle(chier* H, int l, const double* b, double* x, int steps) {
  clevel* L = &H->lv[l];
  if (l == H->nlev - 1) { coarse_solve(L, b, x); return; }
  const cmesh* M = L->mesh;
  const int64_t n = M->n;
  clevel* C = &H->lv[l + 1];
  chebyshev(L, b, x, steps, 1);
  resid(M, L->A, x, b, L->t);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) if (L->mask[i]) L->t[i] = 0.0;
  restrict_to(M, C->mesh, L->t, C->b);
  for (int64_t i = 0; i < C->mesh->n; ++i) if (C->mask[i]) C->b[i] = 0.0;
  vcycle(H, l + 1, C->b, C->x, steps);
  prolong_add(M, C->mesh, C->x, L->mask, x);
  chebyshev(L, b, x, steps, 0);
}

/* ------------------------------------------------------------------------------------------- PCG */
typedef struct { int its; double res; int converged; } ksp_out;

static void apply_pc(csys* S, int which, int pc, int smooth, const double* dinv, const double* r, double* z) {
  const int64_t n = S->mesh.n;
  if (pc == PC_MG) vcycle(&S->H[which], 0, r, z, smooth);
  else if (pc == PC_JACOBI) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) z[i] = dinv[i] * r[i];
  } else vcopy(n, r, z);
}

/* PETSc-style PCG on block `which` (0: A11, 1: A22).  norm 0: preconditioned-residual norm, tolerance
 * max(rtol*||P^-1 b||, atol, reduction * ||P^-1 r_0||); norm 1 (KSP_NORM_UNPRECONDITIONED): ||r||_2 with
 * max(rtol*||b||, atol, reduction * ||r_0||), tested before the preconditioner is applied
 * (cf. oracle/dpp_oracle.py pcg) */
static ksp_out pcg(csys* S, int which, int pc, int smooth, const double* b, double* x, int warm, double rtol, double atol,
                   int max_it, double reduction, int norm, double* w /* 4n work */) {
  const cmesh* M = &S->mesh;
  const int64_t n = M->n;
  const double* A = which ? S->A22 : S->A11;
  const double* dinv = S->H[which].nlev ? S->H[which].lv[0].dinv : NULL;
  double *r = w, *z = w + n, *p = w + 2 * n, *q = w + 3 * n;
  ksp_out out = {0, 0.0, 0};
  if (norm == 2) {
    /* KSP_NORM_NONE: exactly max_it iterations, no convergence test (PETSc: ksp_norm_type none + ksp_max_it);
     * no preconditioner application after the last update */
    if (warm) resid(M, A, x, b, r);
    else { vzero(n, x); vcopy(n, b, r); }
    apply_pc(S, which, pc, smooth, dinv, r, z);
    vcopy(n, z, p);
    double rz = dot(n, r, z);
    while (out.its < max_it) {
      spmv(M, A, p, q);
      const double alpha = rz / dot(n, p, q);
      axpy(n, alpha, p, x);
      axpy(n, -alpha, q, r);
      out.its++;
      if (out.its == max_it) break;
      apply_pc(S, which, pc, smooth, dinv, r, z);
      const double rzn = dot(n, r, z);
      xpby(n, z, rzn / rz, p);
      rz = rzn;
    }
    out.res = sqrt(dot(n, r, r));
    out.converged = 1;
    return out;
  }
  if (norm == 1) {
    const double bnorm = sqrt(dot(n, b, b));
    if (warm) resid(M, A, x, b, r);
    else { vzero(n, x); vcopy(n, b, r); }
    double res = sqrt(dot(n, r, r));
    const double tol = fmax(fmax(rtol * bnorm, atol), reduction * res);
    out.res = res;
    if (res <= tol) { out.converged = 1; return out; }
    apply_pc(S, which, pc, smooth, dinv, r, z);
    vcopy(n, z, p);
    double rz = dot(n, r, z);
    while (out.its < max_it) {
      spmv(M, A, p, q);
      const double alpha = rz / dot(n, p, q);
      axpy(n, alpha, p, x);
      axpy(n, -alpha, q, r);
      out.its++;
      out.res = sqrt(dot(n, r, r));
      if (out.res <= tol) { out.converged = 1; return out; }
      apply_pc(S, which, pc, smooth, dinv, r, z);
      const double rzn = dot(n, r, z);
      xpby(n, z, rzn / rz, p);
      rz = rzn;
    }
    return out;
  }
  double bnorm;
  if (warm) {
    apply_pc(S, which, pc, smooth, dinv, b, z);
    bnorm = sqrt(dot(n, z, z));
    resid(M, A, x, b, r);
    apply_pc(S, which, pc, smooth, dinv, r, z);
  } else {
    vzero(n, x); vcopy(n, b, r);
    apply_pc(S, which, pc, smooth, dinv, r, z);
    bnorm = sqrt(dot(n, z, z));
  }
  double res = sqrt(dot(n, z, z));
  const double tol = fmax(fmax(rtol * bnorm, atol), reduction * res);
  out.res = res;
  if (res <= tol) { out.converged = 1; return out; }
  vcopy(n, z, p);
  double rz = dot(n, r, z);
  while (out.its < max_it) {
    spmv(M, A, p, q);
    const double alpha = rz / dot(n, p, q);
    axpy(n, alpha, p, x);
    axpy(n, -alpha, q, r);
    apply_pc(S, which, pc, smooth, dinv, r, z);
    out.its++;
    out.res = sqrt(dot(n, z, z));
    if (out.res <= tol) { out.converged = 1; return out; }
    const double rzn = dot(n, r, z);
    xpby(n, z, rzn / rz, p);
    rz = rzn;
  }
  return out;
}

/* --------------------------------------------------------------------------------------- C API */
void* dppc_create(int dim, int kind, int nx, int ny, int nz) {
  csys* S = (csys*)calloc(1, sizeof(csys));
  if (!S) return NULL;
  if (mesh_build(&S->mesh, dim, kind, nx, ny, nz)) { free(S); return NULL; }
  const int64_t n = S->mesh.n;
  S->mask[0] = (unsigned char*)calloc((size_t)n, 1);
  S->mask[1] = (unsigned char*)calloc((size_t)n, 1);
  S->g = (double*)calloc((size_t)(2 * n), sizeof(double));
  return S;
}

void dppc_destroy(void* h) {
  csys* S = (csys*)h;
  if (!S) return;
  hier_free(&S->H[0]); hier_free(&S->H[1]);
  for (int l = 0; l < S->ncoarse; ++l) { mesh_free(S->coarse[l]); free(S->coarse[l]); }
  free(S->mask[0]); free(S->mask[1]); free(S->g);
  free(S->A11); free(S->A22); free(S->A12); free(S->A21); free(S->rhs);
  mesh_free(&S->mesh);
  free(S);
}

void dppc_sizes(void* h, int64_t* n, int64_t* ncell, int64_t* nnz, int32_t* m) {
  csys* S = (csys*)h;
  *n = S->mesh.n; *ncell = S->mesh.ncell; *nnz = S->mesh.nnz; *m = S->mesh.m;
}

void dppc_get_mesh(void* h, int32_t* cells, double* coords /* [n][dim] */) {
  csys* S = (csys*)h;
  memcpy(cells, S->mesh.cells, sizeof(int32_t) * (size_t)S->mesh.ncell * S->mesh.m);
  for (int64_t v = 0; v < S->mesh.n; ++v)
    for (int d = 0; d < S->mesh.dim; ++d) coords[v * S->mesh.dim + d] = S->mesh.xyz[3 * v + d];
}

int dppc_set_dirichlet(void* h, int field, const int64_t* nodes, const double* vals, int64_t count) {
  csys* S = (csys*)h;
  if (field < 0 || field > 1) return -1;
  for (int64_t t = 0; t < count; ++t) {
    if (nodes[t] < 0 || nodes[t] >= S->mesh.n) return -1;
    S->mask[field][nodes[t]] = 1;
    S->g[(size_t)field * S->mesh.n + nodes[t]] = vals[t];
  }
  S->assembled = 0;
  return 0;
}

/* K, M, the four eliminated blocks and the lifted right-hand side */
int dppc_assemble(void* h, double k1, double k2, double beta, double mu) {
  csys* S = (csys*)h;
  cmesh* M = &S->mesh;
  const int64_t n = M->n;
  if (assemble_KM(M)) return -2;
  S->a = k1 / mu; S->b = beta / mu; S->c = k2 / mu;
  const size_t nb = sizeof(double) * (size_t)M->nnz;
  if (!S->A11) { S->A11 = (double*)malloc(nb); S->A22 = (double*)malloc(nb); S->A12 = (double*)malloc(nb);
                 S->A21 = (double*)malloc(nb); S->rhs = (double*)malloc(sizeof(double) * (size_t)(2 * n)); }
  make_block(M, S->a, S->b, S->mask[0], S->mask[0], 1, S->A11);
  make_block(M, S->c, S->b, S->mask[1], S->mask[1], 1, S->A22);
  make_block(M, 0.0, -S->b, S->mask[0], S->mask[1], 0, S->A12);
  make_block(M, 0.0, -S->b, S->mask[1], S->mask[0], 0, S->A21);
  /* rhs = -(A u0) on free rows with the un-eliminated operator, u0 = boundary data */
  const double* g1 = S->g; const double* g2 = S->g + n;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double s1 = 0.0, s2 = 0.0;
    for (int64_t p = M->rowptr[r]; p < M->rowptr[r + 1]; ++p) {
      const int32_t c = M->col[p];
      const double km1 = S->a * M->K[p] + S->b * M->M[p], km2 = S->c * M->K[p] + S->b * M->M[p], bm = S->b * M->M[p];
      s1 += km1 * g1[c] - bm * g2[c];
      s2 += km2 * g2[c] - bm * g1[c];
    }
    S->rhs[r] = S->mask[0][r] ? 0.0 : -s1;
    S->rhs[n + r] = S->mask[1][r] ? 0.0 : -s2;
  }
  hier_free(&S->H[0]); hier_free(&S->H[1]);
  S->assembled = 1;
  return 0;
}

int dppc_mg_setup(void* h, int min_cells) {
  csys* S = (csys*)h;
  if (!S->assembled) return -1;
  /* a repeated setup integrates the cached coarse meshes again: every step does the same work (the first one also
   * pays for allocation and first touch, which a timed second step does not) */
  for (int l = 0; l < S->ncoarse; ++l)
    if (assemble_KM(S->coarse[l])) return -2;
  if (hier_build(S, &S->H[0], &S->mesh, S->a, S->b, S->mask[0], min_cells)) return -2;
  if (hier_build(S, &S->H[1], &S->mesh, S->c, S->b, S->mask[1], min_cells)) return -2;
  return S->H[0].nlev;
}

static const double* mat_of(csys* S, int which) {
  switch (which) {
    case MAT_K: return S->mesh.K;
    case MAT_M: return S->mesh.M;
    case MAT_A11: return S->A11;
    case MAT_A22: return S->A22;
    case MAT_A12: return S->A12;
    case MAT_A21: return S->A21;
  }
  return NULL;
}

int dppc_get_csr(void* h, int which, int64_t* rowptr, int32_t* col, double* val) {
  csys* S = (csys*)h;
  const double* v = mat_of(S, which);
  if (!v) return -1;
  memcpy(rowptr, S->mesh.rowptr, sizeof(int64_t) * (size_t)(S->mesh.n + 1));
  memcpy(col, S->mesh.col, sizeof(int32_t) * (size_t)S->mesh.nnz);
  memcpy(val, v, sizeof(double) * (size_t)S->mesh.nnz);
  return 0;
}

void dppc_get_rhs(void* h, double* rhs, double* u0) {
  csys* S = (csys*)h;
  memcpy(rhs, S->rhs, sizeof(double) * (size_t)(2 * S->mesh.n));
  memcpy(u0, S->g, sizeof(double) * (size_t)(2 * S->mesh.n));
}

int dppc_spmv(void* h, int which, const double* x, double* y) {
  csys* S = (csys*)h;
  const double* v = mat_of(S, which);
  if (!v) return -1;
  spmv(&S->mesh, v, x, y);
  return 0;
}

/* seconds per SpMV of matrix `which`, averaged over `reps` after 3 warm-up products */
double dppc_spmv_bench(void* h, int which, int reps) {
  csys* S = (csys*)h;
  const double* v = mat_of(S, which);
  const int64_t n = S->mesh.n;
  double* x = (double*)malloc(sizeof(double) * (size_t)n);
  double* y = (double*)malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) x[i] = 1.0 / (double)(1 + (i % 17));
  for (int t = 0; t < 3; ++t) spmv(&S->mesh, v, x, y);
  const double t0 = omp_get_wtime();
  for (int t = 0; t < reps; ++t) spmv(&S->mesh, v, x, y);
  const double dt = (omp_get_wtime() - t0) / reps;
  free(x); free(y);
  return dt;
}

void dppc_vcycle(void* h, int which, const double* r, double* z, int smooth) {
  csys* S = (csys*)h;
  vcycle(&S->H[which], 0, r, z, smooth);
}

int dppc_pcg(void* h, int which, int pc, const double* b, double* x, int warm, double rtol, double atol, int max_it,
             double reduction, int smooth, int norm, double* resnorm) {
  csys* S = (csys*)h;
  const int64_t n = S->mesh.n;
  if (!S->H[which].nlev) return -1;         /* the Jacobi diagonal lives on level 0 of the hierarchy */
  double* w = (double*)malloc(sizeof(double) * (size_t)(4 * n));
  ksp_out o = pcg(S, which, pc, smooth, b, x, warm, rtol, atol, max_it, reduction, norm, w);
  free(w);
  if (resnorm) *resnorm = o.res;
  return o.converged ? o.its : -2 - o.its;
}

/* Block Picard per dpp_delayed_form (Gauss-Seidel order, warm-started inexact block solves) until the true
 * residual of the monolithic system drops below max(rtol*||F(u0)||, atol).  x_out = u0 + correction. */
int dppc_picard(void* h, int pc, double inner_rtol, double inner_atol, int inner_max_it, double reduction, int smooth,
                int inner_norm, double rtol, double atol, int max_it, double* x_out, int* inner_its, double* resnorm) {
  csys* S = (csys*)h;
  const cmesh* M = &S->mesh;
  const int64_t n = M->n;
  if (!S->assembled || !S->H[0].nlev) return -1;
  double* w = (double*)malloc(sizeof(double) * (size_t)(4 * n));
  double* du = (double*)calloc((size_t)(2 * n), sizeof(double));
  double* b = (double*)malloc(sizeof(double) * (size_t)n);
  double* t = (double*)malloc(sizeof(double) * (size_t)n);
  const double r0 = sqrt(dot(2 * n, S->rhs, S->rhs));
  double res = r0;
  int sweeps = 0, tot = 0;
  while (res > fmax(rtol * r0, atol) && sweeps < max_it) {
    spmv(M, S->A12, du + n, t);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) b[i] = S->rhs[i] - t[i];
    ksp_out o1 = pcg(S, 0, pc, smooth, b, du, sweeps > 0, inner_rtol, inner_atol, inner_max_it, reduction, inner_norm, w);
    spmv(M, S->A21, du, t);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) b[i] = S->rhs[n + i] - t[i];
    ksp_out o2 = pcg(S, 1, pc, smooth, b, du + n, sweeps > 0, inner_rtol, inner_atol, inner_max_it, reduction, inner_norm,
                     w);
    tot += o1.its + o2.its;
    sweeps++;
    /* true residual of the monolithic system */
    double s = 0.0;
    spmv(M, S->A11, du, t); spmv(M, S->A12, du + n, b);
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int64_t i = 0; i < n; ++i) { const double e = S->rhs[i] - t[i] - b[i]; s += e * e; }
    spmv(M, S->A21, du, t); spmv(M, S->A22, du + n, b);
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int64_t i = 0; i < n; ++i) { const double e = S->rhs[n + i] - t[i] - b[i]; s += e * e; }
    res = sqrt(s);
  }
  for (int64_t i = 0; i < 2 * n; ++i) x_out[i] = S->g[i] + du[i];
  if (inner_its) *inner_its = tot;
  if (resnorm) *resnorm = res;
  free(w); free(du); free(b); free(t);
  return res <= fmax(rtol * r0, atol) ? sweeps : -2 - sweeps;
}

int dppc_num_threads(void) { return omp_get_max_threads(); }
void dppc_set_threads(int t) { omp_set_num_threads(t); }
