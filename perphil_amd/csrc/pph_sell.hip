// Stencil-ELL ("SELL") operator format of the scalar blocks and its SpMV.
//
// Every scalar block of this path (A11, A22, A12 and the multigrid level operators) lives on a structured box whose
// CSR pattern is a fixed stencil (27 / 15 / 9 / 7 neighbours, ascending columns).  SELL stores the same matrix as
// val[slot][row]: no column indices, no row pointers (8 B instead of 12 B per stored entry, 20 B per row less),
// every matrix load AND every x load is coalesced across consecutive rows, and the x window of a row block is shared
// through the vector L1.  It replaces PETSc's MatMult (seqaij) on the blocks of the Picard / field-split solves
// (reference src/perphil/solvers/solver.py:71 -> KSPSolve); the CSR arrays remain the export format
// (pph_get_csr, reference src/perphil/solvers/conditioning.py:82-85) and the format of the monolithic Krylov path.
//
// Algorithmic bytes per product: 8 S nrows (values, S = stencil size, pad zeros on the box boundary included)
// + 16 nrows (x read once, y written once).
#include "pph_internal.h"
#include <chrono>
#include <vector>

__device__ inline double sell_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// operands of the epilogue of a row block: requested before the matrix stream, not after the sums
template <int MODE, int RPT, bool CLAMP>
__device__ __forceinline__ void sell_prologue(const double* __restrict__ x, const double* __restrict__ b,
                                              const double* __restrict__ dinv, const double* __restrict__ y,
                                              const double* __restrict__ aux, const double* __restrict__ z0, int64_t n,
                                              int64_t r0, double (&acc)[RPT], double (&bv)[RPT], double (&xr)[RPT],
                                              double (&dv)[RPT], double (&tv)[RPT], double (&av)[RPT], bool (&act)[RPT]) {
  if constexpr (RPT == 2 && !CLAMP) {   // both rows inside: 16-byte loads (sell_ld2)
    acc[0] = acc[1] = 0.0; bv[0] = bv[1] = 0.0; xr[0] = xr[1] = 0.0; dv[0] = dv[1] = 0.0; tv[0] = tv[1] = 0.0; av[0] = av[1] = 0.0;
    act[0] = act[1] = true;
    if (MODE == 1 || (MODE >= 3 && MODE <= 5) || MODE == 7) sell_ld2(b + r0, bv[0], bv[1]);
    if ((MODE >= 2 && MODE <= 4) || MODE == 7) sell_ld2(x + r0, xr[0], xr[1]);
    if (MODE == 3 || MODE == 4) sell_ld2(dinv + r0, dv[0], dv[1]);
    if (MODE == 5 || MODE == 6) { sell_ld2(y + r0, tv[0], tv[1]); sell_ld2(aux + r0, av[0], av[1]); if (z0) sell_ld2(dinv + r0, dv[0], dv[1]); }
    return;
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    acc[i] = 0.0; bv[i] = 0.0; xr[i] = 0.0; dv[i] = 0.0; tv[i] = 0.0; av[i] = 0.0;
    act[i] = !CLAMP || (r0 + i < n);
    if (act[i]) {
      if (MODE == 1 || (MODE >= 3 && MODE <= 5) || MODE == 7) bv[i] = b[r0 + i];
      if ((MODE >= 2 && MODE <= 4) || MODE == 7) xr[i] = x[r0 + i];
      if (MODE == 3 || MODE == 4) dv[i] = dinv[r0 + i];
      if (MODE == 5 || MODE == 6) { tv[i] = y[r0 + i]; av[i] = aux[r0 + i]; if (z0) dv[i] = dinv[r0 + i]; }
    }
  }
}

// what a mode does with the row sums acc = (A x)[r0 .. r0 + RPT) (modes: see sell_rows)
template <int MODE, int RPT>
__device__ __forceinline__ void sell_epilogue(const double (&acc)[RPT], const double (&bv)[RPT], const double (&xr)[RPT],
                                              const double (&dv)[RPT], const double (&tv)[RPT], const double (&av)[RPT],
                                              const bool (&act)[RPT], double w, double* __restrict__ y,
                                              double* __restrict__ aux, double* __restrict__ z0, int64_t r0,
                                              double& dotacc, int64_t dlo, int64_t dhi, double* dotx, int flags) {
  double yo[RPT], ao[RPT], zo[RPT];   // what goes to y, aux, z0
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    yo[i] = 0.0; ao[i] = 0.0; zo[i] = 0.0;
    if (!act[i]) continue;
    const int64_t r = r0 + i;
    if (MODE == 0) yo[i] = acc[i];
    else if (MODE == 1) yo[i] = bv[i] - acc[i];
    else if (MODE == 2) { yo[i] = acc[i]; if (r >= dlo && r < dhi) dotacc += acc[i] * xr[i]; }   // (owned rows: ghost rows of a symmetric slab operator hold no row of this rank)
    else if (MODE == 7) {
      yo[i] = acc[i];
      if (r >= dlo && r < dhi) { dotacc += acc[i] * xr[i]; dotx[0] += acc[i] * bv[i]; dotx[1] += acc[i] * acc[i]; }
    } else if (MODE >= 5) {
      const double tn = (MODE == 5) ? bv[i] - acc[i] : acc[i];
      const double rn = av[i] + ((MODE == 5) ? 1.0 : -1.0) * (tn - tv[i]);   // k_shift
      ao[i] = rn;
      yo[i] = tn;
      zo[i] = dv[i] * rn * w;   // the next block solve's first pre-smoothing (k_cg_update_dev's order)
      if (r >= dlo && r < dhi) dotacc += rn * rn;
    } else {
      const double yn = xr[i] + dv[i] * (bv[i] - acc[i]) * w;   // the order of k_cheb_init: dinv * r / theta
      yo[i] = yn;
      if (MODE == 4 && r >= dlo && r < dhi) dotacc += bv[i] * yn;
    }
  }
  if constexpr (RPT == 2) {
    if (act[0] && act[1] && !(MODE == 0 && (flags & 1))) {
      sell_st2(y + r0, yo[0], yo[1]);
      if (MODE == 5 || MODE == 6) {
        sell_st2(aux + r0, ao[0], ao[1]);
        if (z0) sell_st2(z0 + r0, zo[0], zo[1]);
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    if (!act[i]) continue;
    const int64_t r = r0 + i;
    if (MODE == 0 && (flags & 1)) __builtin_nontemporal_store(yo[i], y + r); else y[r] = yo[i];
    if (MODE == 5 || MODE == 6) {
      aux[r] = ao[i];
      if (z0) z0[r] = zo[i];
    }
  }
}

// MODE 0: y = A x      1: y = b - A x      2: y = A x and per-workgroup partial sums of x.y over the rows [dlo, dhi)
// MODE 3: y = x + w * dinv .* (b - A x)   (one damped-Jacobi / one-step Chebyshev sweep, out of place)
// MODE 4: MODE 3 and the dot product b . y over the rows [dlo, dhi) (CG: r . z from the last kernel of the V-cycle)
// MODE 5: t = b - A x ;  aux += t - y ;  y = t      MODE 6: t = A x ;  aux -= t - y ;  y = t      (the residual
//         bookkeeping of the Picard sweeps, k_shift, in the epilogue of the coupling product) and the sum of
//         aux^2 over the rows [dlo, dhi); z0 != null: also z0 = dinv .* aux * w (pre-smoothing of the new residual)
// MODE 7: MODE 2 and, for the same rows, the sums b.y and y.y to part[pstride + blockIdx.x], part[2 pstride + blockIdx.x]
//         (CG: with b = r the host can form r.r of the NEXT residual from p.Ap, r.Ap, Ap.Ap before the update has run)
// MODE 2 / 4 / 5 / 6 / 7 write one partial sum per workgroup to part[blockIdx.x]; the caller finishes the sum.
// One thread owns RPT consecutive rows; a workgroup a chunk of 256 RPT rows.  Chunks are dealt to the XCDs in groups
// of `group` consecutive chunks (group 1 = plain grid-stride order); workgroups with equal blockIdx % 8 share an XCD.
// DICT: the coefficients come from the row dictionary (struct SellDict): dtab = its table in LDS, cls = the classes;
// x loads, order of the sums and epilogue are the plain kernel's, so the products are bit-identical
template <int KIND, int MODE, int RPT, bool CLAMP, bool SYM = false, bool DICT = false>
__device__ __forceinline__ void sell_rows(const double* __restrict__ val, int64_t ld, const double* __restrict__ x,
                                          const double* __restrict__ b, const double* __restrict__ dinv, double w,
                                          double* __restrict__ y, double* __restrict__ aux, double* __restrict__ z0,
                                          int64_t n, int px, int64_t pxy, int64_t r0, double& dotacc, int64_t dlo,
                                          int64_t dhi, double* dotx = nullptr, int flags = 0,
                                          const uint16_t* __restrict__ cls = nullptr, const double* dtab = nullptr) {
  using ST = SellSt<KIND>;
  double acc[RPT];
  double bv[RPT], xr[RPT], dv[RPT], tv[RPT], av[RPT];
  bool act[RPT];
  sell_prologue<MODE, RPT, CLAMP>(x, b, dinv, y, aux, z0, n, r0, acc, bv, xr, dv, tv, av, act);
  int cb[RPT];   // DICT: offset of the row's class in the table
  if constexpr (DICT) {
    if constexpr (RPT == 2 && !CLAMP) {
      const uint32_t two = *reinterpret_cast<const uint32_t*>(cls + r0);   // (r0 even, cls 256-byte aligned)
      cb[0] = (int)(two & 0xffffu) * ST::S; cb[1] = (int)(two >> 16) * ST::S;
    } else {
#pragma unroll
      for (int i = 0; i < RPT; ++i) cb[i] = act[i] ? (int)cls[r0 + i] * ST::S : 0;
    }
  }
  int slot = 0;
#pragma unroll
  for (int l = 0; l < ST::NL; ++l) {
    const int mask = ST::mask(l);
    if (mask == 0) continue;
    const int64_t L = r0 + (int64_t)ST::dy(l) * px + (int64_t)ST::dz(l) * pxy;
    double xs[RPT + 2];
#ifdef PPH_SELL_PROBE
    if (!CLAMP && RPT == 2) {   // timing probe (wrong results): one aligned 16-byte load per x line
      const double2 t = *reinterpret_cast<const double2*>(x + (L & ~(int64_t)1));
      xs[0] = t.x; xs[1] = t.x; xs[2] = t.y; xs[3] = t.y;
    } else
#endif
#pragma unroll
    for (int e = 0; e < RPT + 2; ++e) {
      bool need = false;
#pragma unroll
      for (int i = 0; i < RPT; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          if (((mask >> d) & 1) && i + d == e) need = true;
      xs[e] = 0.0;
#ifdef PPH_DICT_PROBE
      if (DICT && (flags & 16)) { if (need) xs[e] = 1.0 + e + l + (double)threadIdx.x; } else   // timing probe: no x loads
#endif
      if (need) {
        int64_t idx = L + e - 1;
        if (CLAMP) idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
        xs[e] = x[idx];
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (!((mask >> d) & 1)) continue;
      // symmetric storage: stored slot = slot - S/2 for the diagonal and the upper slots; a lower slot is the mirror
      // slot S - 1 - slot of row r + o (o < 0 its offset): each stored value serves two rows
      constexpr int C0 = ST::S / 2;
      const double* vp = val + (int64_t)(SYM ? (slot >= C0 ? slot - C0 : 0) : slot) * ld + r0;
      double v[RPT];
      if constexpr (DICT) {
#ifdef PPH_DICT_PROBE
        if (flags & 8) {   // timing probe (wrong results): no LDS reads
#pragma unroll
          for (int i = 0; i < RPT; ++i) v[i] = 1.0 + slot;
        } else
#endif
#pragma unroll
        for (int i = 0; i < RPT; ++i) v[i] = act[i] ? dtab[cb[i] + slot] : 0.0;
      } else if (SYM && slot < C0) {
        const int64_t off = (int64_t)(d - 1) + (int64_t)ST::dy(l) * px + (int64_t)ST::dz(l) * pxy;
        const double* mp = val + (int64_t)(ST::S - 1 - slot - C0) * ld;
#if defined(PPH_SELL_PROBE) && PPH_SELL_PROBE >= 2
        if (!CLAMP && RPT == 2) {   // timing probe (wrong results): mirror values as one aligned 16-byte load
          const double2 t = *reinterpret_cast<const double2*>(mp + ((r0 + off) & ~(int64_t)1));
          v[0] = t.x; v[1] = t.y;
        } else
#endif
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int64_t rr = r0 + i + off;
          v[i] = (act[i] && (!CLAMP || (rr >= 0 && rr < n))) ? mp[rr] : 0.0;
        }
      } else if constexpr (RPT == 1) {
        v[0] = act[0] ? vp[0] : 0.0;
      } else {
        // r0 and ld are multiples of RPT and the array is 256-byte aligned: 16-byte loads; rows beyond n fall into the
        // zero padding up to ld (n odd) or are masked
#pragma unroll
        for (int i = 0; i < RPT; i += 2) {
          double2 t = make_double2(0.0, 0.0);
          if (act[i]) {
            if (SYM && slot == C0 && (flags & 2)) {   // experiment: the diagonal is the one stored slot no other row reads
              typedef double v2d __attribute__((ext_vector_type(2)));
              const v2d q = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(vp + i));
              t = make_double2(q.x, q.y);
            } else t = *reinterpret_cast<const double2*>(vp + i);
          }
          v[i] = t.x; v[i + 1] = t.y;
        }
      }
#pragma unroll
      for (int i = 0; i < RPT; ++i) acc[i] += v[i] * xs[i + d];
      ++slot;
    }
  }
  sell_epilogue<MODE, RPT>(acc, bv, xr, dv, tv, av, act, w, y, aux, z0, r0, dotacc, dlo, dhi, dotx, flags);
}

struct SellDictArgs { const uint16_t* cls; const double* tab; const int* state; int ncls; int zconst; };   // zconst: SellDict::zconst

template <int KIND, int MODE, int RPT, bool SYM = false, bool DICT = false>
__global__ __launch_bounds__(256) void k_spmv_sell(const double* __restrict__ val, int64_t ld,
                                                   const double* __restrict__ x, const double* __restrict__ b,
                                                   const double* __restrict__ dinv, const double* __restrict__ wp,
                                                   double* __restrict__ y, double* __restrict__ aux,
                                                   double* __restrict__ z0, int64_t n, int px, int64_t pxy, int64_t halo,
                                                   int64_t nchunks, int64_t chunk0,
                                                   int group, int zwalk, int xmap, double* __restrict__ part,
                                                   int64_t dlo, int64_t dhi, int flags, SellDictArgs da) {
  constexpr int CH = 256 * RPT;
  // DICT: the table of distinct rows -> LDS, if the device-side state says this assembly's dictionary is the one the
  // host sized the launch for; otherwise (check failed, rebuilt with another class count) the plain path below
  extern __shared__ double dtab[];
  bool dok = false;
  if constexpr (DICT) {
    dok = da.state[0] == da.ncls && da.state[1] == 1;
    if (dok) {
      for (int i = threadIdx.x; i < da.ncls * SellSt<KIND>::S; i += 256) dtab[i] = da.tab[i];
      __syncthreads();
    }
  }
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, bpx = gridDim.x >> 3;   // launcher keeps gridDim.x a multiple of 8
  // the smoother weight lives in device memory (refreshed per assembly) so that captured graphs survive a re-assembly
  const double w = (MODE == 3 || MODE == 4 || ((MODE == 5 || MODE == 6) && z0)) ? *wp : 0.0;
  double dotacc = 0.0;
  double dotx[2] = {0.0, 0.0};
  // Chunk order.  zwalk = Z > 0 (3D): a workgroup takes Z work items in a row that sit at the same in-plane position
  // of Z consecutive node planes (chunk, chunk + P, ..., P = chunks per plane rounded: the rows shift by pxy - P CH,
  // one row at 256^3).  The x lines and - with symmetric storage - the operator values a plane reads at offset
  // -pxy are the ones this workgroup loaded for the plane below a moment ago: they come from L1 / this XCD's L2
  // instead of the fabric.  zwalk = 0: chunks dealt to the XCDs in groups of `group` (1 = plane grid-stride order).
  const int64_t P = zwalk > 0 ? (pxy + CH / 2) / CH : 0;
  const bool column = zwalk >= PPH_SELL_COLUMN_WALK && P > 0;
  const int64_t nitems = (zwalk > 0 && P > 0 && !column) ? ((nchunks + P * zwalk - 1) / (P * zwalk)) * P * zwalk : 0;
  // column walk: the P x planes work items in (position-major, plane-minor) order are cut into gridDim.x equal
  // contiguous ranges - a workgroup climbs (a part of) one z column plane by plane, every workgroup the same number
  // of planes, and a cold start (mirror values of the plane below not in L2) happens once per range
  const int64_t planes = column ? (nchunks + P - 1) / P : 0;
  const int64_t ctot = planes * P;
  // xmap: the workgroups of one XCD (equal blockIdx % 8) take CONSECUTIVE in-plane positions, so that the rows a
  // chunk reads from its in-plane neighbours (mirror values and x of the lines next to the chunk) were loaded into
  // the same L2 by the neighbour workgroup in the same plane step
  const int64_t vb = xmap ? (int64_t)xcd * bpx + bx : (int64_t)blockIdx.x;
  const int64_t cq1 = column ? (ctot * (vb + 1)) / gridDim.x : 0;
  for (int64_t q = column ? (ctot * vb) / gridDim.x : (nitems ? vb * zwalk : bx);;
       q += ((nitems || column) ? 1 : bpx)) {
    int64_t chunk;
    if (column) {
      if (q >= cq1) break;
      chunk = (q % planes) * P + q / planes;
      if (chunk >= nchunks) continue;
    } else if (nitems) {
      // items [b Z, b Z + Z) then [(b + gridDim) Z, ...): q walks one item at a time inside a run of Z
      const int64_t run = q / zwalk, t = q % zwalk;
      if (run * zwalk >= nitems) break;
      const int64_t slab = run / P, pos = run % P;
      chunk = slab * P * zwalk + t * P + pos;
      if (t == zwalk - 1) q += (int64_t)(gridDim.x - 1) * zwalk;   // next run of this workgroup
      if (chunk >= nchunks) continue;
    } else {
      const int64_t base = ((q / group) * 8 + xcd) * (int64_t)group;
      if (base >= nchunks) break;
      chunk = base + q % group;
      if (chunk >= nchunks) continue;
    }
    const int64_t c0 = (chunk0 + chunk) * CH;   // (chunk0: first chunk of a sub-range launch, see sell_spmv)
    const int64_t r0 = c0 + (int64_t)threadIdx.x * RPT;
    // a chunk whose rows and x window lie inside [0, n) needs no index clamps and no row masks (all but the first
    // and last few chunks)
    if constexpr (DICT) {
      if (dok) {
        if (c0 >= halo && c0 + CH + halo <= n)
          sell_rows<KIND, MODE, RPT, false, SYM, true>(val, ld, x, b, dinv, w, y, aux, z0, n, px, pxy, r0, dotacc, dlo, dhi, dotx, flags, da.cls, dtab);
        else
          sell_rows<KIND, MODE, RPT, true, SYM, true>(val, ld, x, b, dinv, w, y, aux, z0, n, px, pxy, r0, dotacc, dlo, dhi, dotx, flags, da.cls, dtab);
        continue;
      }
    }
    if (c0 >= halo && c0 + CH + halo <= n)
      sell_rows<KIND, MODE, RPT, false, SYM>(val, ld, x, b, dinv, w, y, aux, z0, n, px, pxy, r0, dotacc, dlo, dhi, dotx, flags);
    else
      sell_rows<KIND, MODE, RPT, true, SYM>(val, ld, x, b, dinv, w, y, aux, z0, n, px, pxy, r0, dotacc, dlo, dhi, dotx, flags);
  }
  if (MODE == 2 || MODE >= 4) {
    __shared__ double lds[4];
    dotacc = sell_wave_sum(dotacc);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = dotacc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
    if (MODE == 7) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        __syncthreads();
        const double t = sell_wave_sum(dotx[k]);
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) part[(int64_t)(k + 1) * PPH_PART_STRIDE + blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Dictionary product, z-walk with the x window in registers (hexahedra, whole-operator launches)
// ------------------------------------------------------------------------------------------------
// With the coefficients in LDS the product of k_spmv_sell<.., DICT> is bound by its 36 x loads per row pair (L1), of which
// 24 repeat what the pair one plane below loaded.  Here a thread keeps the in-plane position of its two rows and climbs
// the planes: the x lines of planes z - 1, z, z + 1 sit in three register sets that rotate, 12 loads per step.  The work
// (in-plane chunks of 512 positions x planes, column-major) is cut into gridDim.x equal contiguous ranges.  Sums, their
// order and the epilogue are k_spmv_sell's: products stay bit-identical (the per-workgroup partial sums of the dot
// modes group other rows, as with any other grid).
template <bool CLAMP>
__device__ __forceinline__ void dictw_load_plane(const double* __restrict__ x, int64_t base, int px, int64_t n, double (&win)[3][4]) {
  // base = first row of the pair in the plane to load; lines dy = -1, 0, +1, entries base + dy px - 1 .. + 2
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int64_t idx = base + (int64_t)(j - 1) * px + e - 1;
      if (CLAMP) idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);   // (k_spmv_sell's clamp: the coefficient there is 0)
      win[j][e] = x[idx];
    }
}

// operands of one step besides the x window: classes and the epilogue's vectors, requested one step ahead
struct DwOps { double bv[2], dv[2], tv[2], av[2]; uint32_t cls2; };   // cls2: the two classes as loaded (low / high half)

// ALL: both rows of every lane are inside the plane (all position chunks but the last of a plane) - no lane masks, no
// branches: the 4-step loop must stay straight-line code, or the compiler's wait-count pass drains every outstanding
// request at each join and the requests of the next step are never in flight while this one computes
// NOCLS: the classes of the pair are those of the plane below (SellDict::zconst), the caller copies them
template <int MODE, bool ALL, bool NOCLS = false>
__device__ __forceinline__ void dictw_fetch(DwOps& o, const uint16_t* __restrict__ cls, const double* __restrict__ b,
                                            const double* __restrict__ dinv, const double* __restrict__ y,
                                            const double* __restrict__ aux, const double* __restrict__ z0, int64_t r0, bool a0,
                                            bool a1) {
  const bool act[2] = {a0, a1};
  if (ALL || (a0 && a1)) {
    o.bv[0] = o.bv[1] = 0.0; o.dv[0] = o.dv[1] = 0.0; o.tv[0] = o.tv[1] = 0.0; o.av[0] = o.av[1] = 0.0;
    if (!NOCLS) o.cls2 = (uint32_t)cls[r0] | ((uint32_t)cls[r0 + 1] << 16);   // (two aligned 2-byte loads: r0 is odd on every other plane)
    if (MODE == 1 || (MODE >= 3 && MODE <= 5) || MODE == 7) sell_ld2(b + r0, o.bv[0], o.bv[1]);
    if (MODE == 3 || MODE == 4) sell_ld2(dinv + r0, o.dv[0], o.dv[1]);
    if (MODE == 5 || MODE == 6) { sell_ld2(y + r0, o.tv[0], o.tv[1]); sell_ld2(aux + r0, o.av[0], o.av[1]); if (z0) sell_ld2(dinv + r0, o.dv[0], o.dv[1]); }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i == 0) o.cls2 = 0;
    o.bv[i] = 0.0; o.dv[i] = 0.0; o.tv[i] = 0.0; o.av[i] = 0.0;
    if (act[i]) {
      o.cls2 |= (uint32_t)cls[r0 + i] << (16 * i);
      if (MODE == 1 || (MODE >= 3 && MODE <= 5) || MODE == 7) o.bv[i] = b[r0 + i];
      if (MODE == 3 || MODE == 4) o.dv[i] = dinv[r0 + i];
      if (MODE == 5 || MODE == 6) { o.tv[i] = y[r0 + i]; o.av[i] = aux[r0 + i]; if (z0) o.dv[i] = dinv[r0 + i]; }
    }
  }
}

template <int MODE, bool ALL>
__device__ __forceinline__ void dictw_step(const double (&lo)[3][4], const double (&mid)[3][4], const double (&hi)[3][4],
                                           const DwOps& o, const double* dtab, double w, double* __restrict__ y, double* __restrict__ aux,
                                           double* __restrict__ z0, int64_t r0, bool a0, bool a1, double& dotacc, int64_t dlo,
                                           int64_t dhi, double* dotx, int flags) {
  double acc[2] = {0.0, 0.0}, xr[2] = {0.0, 0.0};
  const bool act[2] = {ALL || a0, ALL || a1};
  if ((MODE >= 2 && MODE <= 4) || MODE == 7) {   // x[r0 + i]
    xr[0] = act[0] ? mid[1][1] : 0.0;
    xr[1] = act[1] ? mid[1][2] : 0.0;
  }
  {
    const int cb[2] = {(int)(o.cls2 & 0xffffu) * 27, (int)(o.cls2 >> 16) * 27};
    int slot = 0;
#ifdef PPH_DW_SPLITACC
    double pa[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
#endif
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      const double (&win)[3][4] = (l < 3) ? lo : (l < 6 ? mid : hi);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        double v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = dtab[cb[i] + slot];   // (a row that is not active reads class 0 and is not stored)
#pragma unroll
#ifdef PPH_DW_SPLITACC
        for (int i = 0; i < 2; ++i) pa[l / 3][i] += v[i] * win[l % 3][i + d];
#else
        for (int i = 0; i < 2; ++i) acc[i] += v[i] * win[l % 3][i + d];
#endif
        ++slot;
      }
    }
#ifdef PPH_DW_SPLITACC
    for (int i = 0; i < 2; ++i) acc[i] = (pa[0][i] + pa[1][i]) + pa[2][i];
#endif
  }
  sell_epilogue<MODE, 2>(acc, o.bv, xr, o.dv, o.tv, o.av, act, w, y, aux, z0, r0, dotacc, dlo, dhi, dotx, 0);   // (no store-hint experiments here: a branch around the stores costs the pipelining)
}

#ifdef PPH_DW_STAMPS
// diagnostic build only (tools/r4_dict_stamps.py): s_memtime stamps of one wave per workgroup into a buffer nothing else reads
__device__ unsigned long long* g_dw_stamps = nullptr;
#define PPH_DW_STAMP(I) do { if (stp && (I) < 60) stp[(I)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PPH_DW_STAMP(I) do { } while (0)
#endif
template <int MODE, bool ZC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE <= 1 ? 3 : 2, 4))) void k_spmv_dict_walk(const double* __restrict__ val, int64_t ld,
                                                        const double* __restrict__ x, const double* __restrict__ b,
                                                        const double* __restrict__ dinv, const double* __restrict__ wp,
                                                        double* __restrict__ y, double* __restrict__ aux,
                                                        double* __restrict__ z0, int64_t n, int px, int64_t pxy, int planes,
                                                        double* __restrict__ part, int64_t dlo, int64_t dhi, int flags,
                                                        SellDictArgs da) {
  extern __shared__ double dtab[];
#ifdef PPH_DW_STAMPS
  unsigned long long* stp = (g_dw_stamps && threadIdx.x == 0) ? g_dw_stamps + (size_t)blockIdx.x * 64 : nullptr;
  int sti = 3;
  PPH_DW_STAMP(0);
#endif
  const bool dok = da.state[0] == da.ncls && da.state[1] == 1;
  if (dok) {
    for (int i = threadIdx.x; i < da.ncls * 27; i += 256) dtab[i] = da.tab[i];
    __syncthreads();
  }
  PPH_DW_STAMP(1);
  const double w = (MODE == 3 || MODE == 4 || ((MODE == 5 || MODE == 6) && z0)) ? *wp : 0.0;
  double dotacc = 0.0;
  double dotx[2] = {0.0, 0.0};
  const int64_t P = (pxy + 511) / 512;
  const int64_t total = P * planes;
  int64_t q = total * (int64_t)blockIdx.x / gridDim.x;
  const int64_t q1 = total * ((int64_t)blockIdx.x + 1) / gridDim.x;
  while (q < q1) {
    const int64_t pos = q / planes;
    int z = (int)(q - pos * planes);
    const int zend = (int)((q1 - q) < (int64_t)(planes - z) ? z + (q1 - q) : planes);   // this range's steps in the column
    q += zend - z;
    const int64_t p = pos * 512 + (int64_t)threadIdx.x * 2;
    const bool a0 = p < pxy, a1 = p + 1 < pxy;
    if (!dok) {
      // this assembly's dictionary was refused on the device: the stored values, one row at a time
      for (; z < zend; ++z)
        for (int i = 0; i < 2; ++i)
          if (p + i < pxy)
            sell_rows<PPH_CELL_HEX, MODE, 1, true, true>(val, ld, x, b, dinv, w, y, aux, z0, n, px, pxy, (int64_t)z * pxy + p + i,
                                                         dotacc, dlo, dhi, dotx, flags);
      continue;
    }
    // Register sets A, B, C = x lines of planes z - 1, z, z + 1; D = the plane two steps ahead, O0 / O1 = this step's and
    // the next step's other operands: everything a step needs was requested one step earlier, so that a wave always has a
    // step's worth of loads in flight behind the one it computes on (the memory counter retires in order: all requests of
    // step z + 1 are issued before step z waits for its own).  Planes 0, planes - 1 and beyond reach outside [0, n):
    // clamped loads, kept out of the 4-step loop.
    double A[3][4], B[3][4], C[3][4], D[3][4];
    DwOps O0 = {}, O1 = {};   // (O1 is copied before its first request when a range is one step long)
#define PPH_DW_LOADC(W, ZZ)                                                                                     \
    do {                                                                                                          \
      const int zz_ = (ZZ);                                                                                       \
      if (zz_ >= 1 && zz_ <= planes - 2) dictw_load_plane<false>(x, (int64_t)zz_ * pxy + p, px, n, W);            \
      else dictw_load_plane<true>(x, (int64_t)zz_ * pxy + p, px, n, W);                                           \
    } while (0)
#define PPH_DW_LOAD(W, ZZ) dictw_load_plane<false>(x, (int64_t)(ZZ) * pxy + p, px, n, W)
#define PPH_DW_FETCH(O, ZZ) dictw_fetch<MODE, false>(O, da.cls, b, dinv, y, aux, z0, (int64_t)(ZZ) * pxy + p, a0, a1)
#define PPH_DW_STEP(LO, MID, HI, O, ZZ)                                                                         \
    dictw_step<MODE, false>(LO, MID, HI, O, dtab, w, y, aux, z0, (int64_t)(ZZ) * pxy + p, a0, a1, dotacc, dlo, dhi, dotx, flags)
#define PPH_DW_FETCHA(O, ZZ) dictw_fetch<MODE, true>(O, da.cls, b, dinv, y, aux, z0, (int64_t)(ZZ) * pxy + p, true, true)
#define PPH_DW_STEPA(LO, MID, HI, O, ZZ)                                                                        \
    dictw_step<MODE, true>(LO, MID, HI, O, dtab, w, y, aux, z0, (int64_t)(ZZ) * pxy + p, true, true, dotacc, dlo, dhi, dotx, flags)
    PPH_DW_LOADC(A, z - 1);
    PPH_DW_LOADC(B, z);
    PPH_DW_LOADC(C, z + 1);
    PPH_DW_FETCH(O0, z);
#define PPH_DW_SINGLE()                                                                                         \
    do {                                                                                                          \
      PPH_DW_LOADC(D, z + 2);                                                                                     \
      if (z + 1 < zend) PPH_DW_FETCH(O1, z + 1);                                                                  \
      PPH_DW_STEP(A, B, C, O0, z);                                                                                \
      _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                               \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { A[j][e] = B[j][e]; B[j][e] = C[j][e]; C[j][e] = D[j][e]; } \
      O0 = O1;                                                                                                    \
      ++z;                                                                                                        \
    } while (0)
#ifdef PPH_DW_STAMPS
    PPH_DW_STAMP(2);      // range located, first three planes + operands requested
#endif
    // (plane 0 needs the clamped loads: outside the 4-step loop; with constant classes along z also plane 1, whose classes differ)
    while (z < (ZC ? 2 : 1) && z < zend) PPH_DW_SINGLE();
    if (pos * 512 + 512 <= pxy) {   // every lane has both rows in the plane
      if constexpr (ZC) {
        // Round 4.  The class words are the one stream of this kernel that is read once, from HBM, and needed FIRST in its
        // step (class -> table address -> coefficient -> first multiply-add): 2 of 18 bytes per row cost 0.03 of 0.11 ms
        // (profiles/r04_dict_walk_probes.txt, P0 -> P1).  On a box the class of an in-plane position is the same on all
        // planes 2 .. planes - 3 (checked on the class array when the dictionary is built: SellDict::zconst), so inside
        // that range a step takes its classes from the step below: no class load, one word less in flight.
#define PPH_DW_FETCHN(O, OP, ZZ)                                                                                  \
        do {                                                                                                      \
          dictw_fetch<MODE, true, true>(O, da.cls, b, dinv, y, aux, z0, (int64_t)(ZZ) * pxy + p, true, true);     \
          uint32_t c_ = (OP).cls2;                                                                                \
          asm volatile("" : "+v"(c_));   /* (opaque: or the 54 table reads are hoisted out of the loop - 108 registers) */ \
          (O).cls2 = c_;                                                                                          \
        } while (0)
        while (z + 4 <= zend && z + 4 <= planes - 3) {
          PPH_DW_LOAD(D, z + 2); PPH_DW_FETCHN(O1, O0, z + 1); PPH_DW_STEPA(A, B, C, O0, z);
          PPH_DW_LOAD(A, z + 3); PPH_DW_FETCHN(O0, O1, z + 2); PPH_DW_STEPA(B, C, D, O1, z + 1);
          PPH_DW_LOAD(B, z + 4); PPH_DW_FETCHN(O1, O0, z + 3); PPH_DW_STEPA(C, D, A, O0, z + 2);
          PPH_DW_LOAD(C, z + 5); PPH_DW_FETCHN(O0, O1, z + 4); PPH_DW_STEPA(D, A, B, O1, z + 3);
          z += 4;
#ifdef PPH_DW_STAMPS
          PPH_DW_STAMP(sti); ++sti;      // one stamp per four steps
#endif
        }
#undef PPH_DW_FETCHN
      } else {
        while (z + 4 <= zend && z + 5 <= planes - 2 && z >= 1) {
          PPH_DW_LOAD(D, z + 2); PPH_DW_FETCHA(O1, z + 1); PPH_DW_STEPA(A, B, C, O0, z);
          PPH_DW_LOAD(A, z + 3); PPH_DW_FETCHA(O0, z + 2); PPH_DW_STEPA(B, C, D, O1, z + 1);
          PPH_DW_LOAD(B, z + 4); PPH_DW_FETCHA(O1, z + 3); PPH_DW_STEPA(C, D, A, O0, z + 2);
          PPH_DW_LOAD(C, z + 5); PPH_DW_FETCHA(O0, z + 4); PPH_DW_STEPA(D, A, B, O1, z + 3);
          z += 4;
#ifdef PPH_DW_STAMPS
          PPH_DW_STAMP(sti); ++sti;      // one stamp per four steps
#endif
        }
      }
    }
#ifdef PPH_DW_STAMPS
    PPH_DW_STAMP(sti); if (stp) stp[61] = (unsigned long long)sti; ++sti;
#endif
    while (z < zend) PPH_DW_SINGLE();
#ifdef PPH_DW_STAMPS
    PPH_DW_STAMP(sti); ++sti;
#endif
#undef PPH_DW_SINGLE
#undef PPH_DW_LOADC
#undef PPH_DW_LOAD
#undef PPH_DW_FETCH
#undef PPH_DW_STEP
#undef PPH_DW_FETCHA
#undef PPH_DW_STEPA
  }
#ifdef PPH_DW_STAMPS
  if (stp) { stp[62] = __builtin_amdgcn_s_memtime(); stp[63] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); }
#endif
  if (MODE == 2 || MODE >= 4) {
    __shared__ double lds[4];
    dotacc = sell_wave_sum(dotacc);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = dotacc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
    if (MODE == 7) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        __syncthreads();
        const double t = sell_wave_sum(dotx[k]);
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) part[(int64_t)(k + 1) * PPH_PART_STRIDE + blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
      }
    }
  }
}

#ifdef PPH_DW_STAMPS
// (diagnostic build) the stamps go to the context's right-hand-side vector, which the isolated product loop does not touch
int sell_dw_stamps(pph_ctx* ctx, int on) {
  unsigned long long* p = on ? reinterpret_cast<unsigned long long*>(ctx->rhs.p) : nullptr;
  if (on) PPH_HIP(ctx, hipMemsetAsync(ctx->rhs.p, 0, sizeof(double) * 2 * (size_t)ctx->n, ctx->stream));
  PPH_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_dw_stamps), &p, sizeof(p)));
  return PPH_OK;
}
#endif

// launches the walk kernel for a whole-operator product on a usable dictionary; returns the grid, 0 = not applicable
static int sell_launch_dict_walk(pph_ctx* ctx, int mode, const Sell& E, const double* x, const double* b, const double* dinv,
                                 const double* w, double* y, double* aux, double* z0, int64_t n, double* part, int64_t dlo,
                                 int64_t dhi) {
  const int64_t pxy = (int64_t)E.px * E.py;
  if (!ctx->sell_dict_walk || E.kind != PPH_CELL_HEX || E.pz < 4 || pxy < 2048 || n != pxy * E.pz) return 0;
  const SellDict& D = *E.dict;
  const SellDictArgs da = {D.cls.p, D.tab.p, D.state.p, D.ncls, (D.zconst && ctx->sell_dict_zconst && E.pz >= 8) ? 1 : 0};
  const size_t lds = (size_t)D.ncls * 27 * sizeof(double);
  const int64_t items = ((pxy + 511) / 512) * E.pz;
  int64_t g = ctx->sell_dict_blocks >= 8 ? ctx->sell_dict_blocks : 1024;
  if ((mode == 2 || mode >= 4) && g > ctx->part_cap) g = ctx->part_cap;
  // (ranges of at least 12 steps, or the straight-line loop of the kernel hardly runs: level 1 of a 256^3 hierarchy has 4 257 steps)
  if (g * 12 > items) g = items / 12 > ctx->num_cus ? items / 12 : ctx->num_cus;
  if (g > items) g = items;
  const int grid = (int)g;
#define PPH_DW_GO(MM)                                                                                                       \
  do {                                                                                                                       \
    if (da.zconst)                                                                                                           \
      hipLaunchKernelGGL((k_spmv_dict_walk<MM, true>), dim3(grid), dim3(256), lds, ctx->stream, E.val, E.ld, x, b, dinv, w, y, aux, \
                         z0, n, E.px, pxy, E.pz, part, dlo, dhi, ctx->sell_flags, da);                                       \
    else                                                                                                                     \
      hipLaunchKernelGGL((k_spmv_dict_walk<MM, false>), dim3(grid), dim3(256), lds, ctx->stream, E.val, E.ld, x, b, dinv, w, y, aux, \
                         z0, n, E.px, pxy, E.pz, part, dlo, dhi, ctx->sell_flags, da);                                       \
  } while (0)
  switch (mode) {
    case 0: PPH_DW_GO(0); break;
    case 1: PPH_DW_GO(1); break;
    case 2: PPH_DW_GO(2); break;
    case 3: PPH_DW_GO(3); break;
    case 4: PPH_DW_GO(4); break;
    case 5: PPH_DW_GO(5); break;
    case 6: PPH_DW_GO(6); break;
    default: PPH_DW_GO(7); break;
  }
#undef PPH_DW_GO
  return grid;
}


template <int KIND, int RPT>
static void sell_launch_mode(pph_ctx* ctx, int mode, int grid, const Sell& E, const double* x, const double* b,
                             const double* dinv, const double* w, double* y, double* aux, double* z0, int64_t n, int64_t nchunks, int64_t chunk0,
                             int group, double* part, int64_t dlo, int64_t dhi) {
  const int64_t pxy = (int64_t)E.px * E.py;
  const int64_t halo = (E.pz > 1 ? pxy : 0) + E.px + 2;   // reach of the x window of a row (2D: no z lines)
  int zwalk = (E.sym && E.pz > 2 && ctx->sell_zwalk > 0 && chunk0 == 0 && nchunks >= ctx->sell_zwalk_min_chunks) ? ctx->sell_zwalk : 0;   // (no gain on full storage)
  const SellDictArgs nod = {nullptr, nullptr, nullptr, 0, 0};
#define PPH_SELL_GO(MM)                                                                                              \
  hipLaunchKernelGGL((k_spmv_sell<KIND, MM, RPT>), dim3(grid), dim3(256), 0, ctx->stream, E.val, E.ld, x, b, dinv, w, \
                     y, aux, z0, n, E.px, pxy, halo, nchunks, chunk0, group, zwalk, ctx->sell_xmap, part, dlo, dhi, ctx->sell_flags, nod)
  if constexpr (RPT == 2) {
    if (E.sym && E.dict && E.dict->on) {
      // row dictionary: 2 B per row instead of the value streams; the table of distinct rows goes to LDS
      const SellDict& D = *E.dict;
      const SellDictArgs da = {D.cls.p, D.tab.p, D.state.p, D.ncls, 0};
      const size_t lds = (size_t)D.ncls * SellSt<KIND>::S * sizeof(double);
      if (ctx->sell_dict_zwalk >= 0) zwalk = (E.pz > 2 && chunk0 == 0) ? ctx->sell_dict_zwalk : 0;
#define PPH_SELL_GOD(MM)                                                                                                       \
  hipLaunchKernelGGL((k_spmv_sell<KIND, MM, 2, true, true>), dim3(grid), dim3(256), lds, ctx->stream, E.val, E.ld, x, b, dinv, \
                     w, y, aux, z0, n, E.px, pxy, halo, nchunks, chunk0, group, zwalk, ctx->sell_xmap, part, dlo, dhi,         \
                     ctx->sell_flags, da)
      switch (mode) {
        case 0: PPH_SELL_GOD(0); break;
        case 1: PPH_SELL_GOD(1); break;
        case 2: PPH_SELL_GOD(2); break;
        case 3: PPH_SELL_GOD(3); break;
        case 4: PPH_SELL_GOD(4); break;
        case 5: PPH_SELL_GOD(5); break;
        case 6: PPH_SELL_GOD(6); break;
        default: PPH_SELL_GOD(7); break;
      }
#undef PPH_SELL_GOD
      return;
    }
  }
  if (E.sym) {
#define PPH_SELL_GOS(MM)                                                                                                   \
  hipLaunchKernelGGL((k_spmv_sell<KIND, MM, RPT, true>), dim3(grid), dim3(256), 0, ctx->stream, E.val, E.ld, x, b, dinv, w, \
                     y, aux, z0, n, E.px, pxy, halo, nchunks, chunk0, group, zwalk, ctx->sell_xmap, part, dlo, dhi, ctx->sell_flags, nod)
    switch (mode) {
      case 0: PPH_SELL_GOS(0); break;
      case 1: PPH_SELL_GOS(1); break;
      case 2: PPH_SELL_GOS(2); break;
      case 3: PPH_SELL_GOS(3); break;
      case 4: PPH_SELL_GOS(4); break;
      case 5: PPH_SELL_GOS(5); break;
      case 6: PPH_SELL_GOS(6); break;
      default: PPH_SELL_GOS(7); break;
    }
#undef PPH_SELL_GOS
    return;
  }
  switch (mode) {
    case 0: PPH_SELL_GO(0); break;
    case 1: PPH_SELL_GO(1); break;
    case 2: PPH_SELL_GO(2); break;
    case 3: PPH_SELL_GO(3); break;
    case 4: PPH_SELL_GO(4); break;
    case 5: PPH_SELL_GO(5); break;
    case 6: PPH_SELL_GO(6); break;
    default: PPH_SELL_GO(7); break;
  }
#undef PPH_SELL_GO
}

// launches the product; returns the grid (= number of partial sums written in mode 2)
int sell_spmv(pph_ctx* ctx, const Sell& E, int64_t n, int mode, const double* x, const double* b, const double* dinv,
              const double* w, double* y, double* part, int64_t dlo, int64_t dhi, double* aux, double* z0, int64_t cbeg, int64_t cend,
              int grid_cap) {
  const int rpt = (ctx->sell_rpt == 1) ? 1 : 2;
  // [cbeg, cend) (cend < 0: all): the chunks of this launch - a product split into boundary and interior rows
  const int64_t nchunks_all = ceil_div64(n, 256 * rpt);
  const int64_t chunk0 = cend < 0 ? 0 : cbeg;
  const int64_t nchunks = cend < 0 ? nchunks_all : cend - cbeg;
  if (nchunks <= 0) return 0;
  // grid: persistent, up to 4096 workgroups (128^3: 9.38 ms per step against 9.60 with 2048); with the z-walk order of symmetric operators ONE workgroup per CU: the value a
  // plane reads a second time must still be in the XCD's 4 MB L2, and every resident workgroup streams 57 KB per plane
  // step (measured on the 256^3 block: 0.58 ms plain order, 0.51 ms z-walk with 4096 workgroups, 0.47 ms with 256;
  // profiles/r02_sell_sym_probe_256.txt, r02_sell_sym_probe2_256.txt)
  const bool zw = E.sym && E.pz > 2 && ctx->sell_zwalk > 0 && chunk0 == 0 && nchunks >= ctx->sell_zwalk_min_chunks;   // (smaller levels: too few chunks per workgroup)
  const bool dict = rpt == 2 && E.sym && E.dict && E.dict->on;
  if (dict && cend < 0) {
    const int g = sell_launch_dict_walk(ctx, mode, E, x, b, dinv, w, y, aux, z0, n, part, dlo, dhi);
    if (g > 0) return g;
  }
  int cap = (ctx->sell_blocks >= 8 && ctx->sell_blocks <= 8192) ? (ctx->sell_blocks / 8) * 8
                                                                 : (zw ? ((ctx->num_cus + 7) / 8) * 8 : 4096);
  // (a dictionary product streams vectors only: nothing to keep in L2 for a second reader, many waves to hide latency)
  if (dict && ctx->sell_dict_blocks >= 8) cap = (ctx->sell_dict_blocks / 8) * 8;
  if ((mode == 2 || mode >= 4) && cap > PPH_PART_STRIDE) cap = PPH_PART_STRIDE;   // one partial sum per workgroup
  // (a product launched as several row ranges shares the PPH_PART_STRIDE partial sums of a slot: sell_product)
  if (grid_cap >= 8 && cap > grid_cap) cap = (grid_cap / 8) * 8;
  int64_t g = nchunks < cap ? nchunks : cap;
  g = ((g + 7) / 8) * 8;
  const int grid = (int)g;
  int group = ctx->sell_group > 0 ? ctx->sell_group : 1;
#define PPH_SELL_KIND(KK)                                                                                     \
  if (rpt == 1) sell_launch_mode<KK, 1>(ctx, mode, grid, E, x, b, dinv, w, y, aux, z0, n, nchunks, chunk0, group, part, dlo, dhi); \
  else sell_launch_mode<KK, 2>(ctx, mode, grid, E, x, b, dinv, w, y, aux, z0, n, nchunks, chunk0, group, part, dlo, dhi)
  switch (E.kind) {
    case PPH_CELL_QUAD: PPH_SELL_KIND(PPH_CELL_QUAD); break;
    case PPH_CELL_TRI: PPH_SELL_KIND(PPH_CELL_TRI); break;
    case PPH_CELL_HEX: PPH_SELL_KIND(PPH_CELL_HEX); break;
    default: PPH_SELL_KIND(PPH_CELL_TET); break;
  }
#undef PPH_SELL_KIND
  return grid;
}

// ------------------------------------------------------------------------------------------------
// Row dictionary (struct SellDict, pph_internal.h): build, table, check
// ------------------------------------------------------------------------------------------------
static inline int sell_grid(int64_t n) {
  int64_t b = ceil_div64(n, 256);
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// the coefficient k_spmv_sell multiplies x[r + off] with for row r and (full) slot s: the stored value, or - symmetric
// storage, lower slot - the mirror slot of row r + off (0 outside [0, n): the kernel's clamped chunks)
__device__ __forceinline__ double sell_coef(const double* __restrict__ val, int64_t ld, int sym, int S, int s, int64_t r,
                                            int64_t off, int64_t n) {
  const int c0 = S / 2;
  if (!sym) return val[(int64_t)s * ld + r];
  if (s >= c0) return val[(int64_t)(s - c0) * ld + r];
  const int64_t rr = r + off;
  return (rr >= 0 && rr < n) ? val[(int64_t)(S - 1 - s - c0) * ld + rr] : 0.0;
}
__device__ __forceinline__ int64_t sell_off(const Stencil& st, int s, int px, int64_t pxy) {
  return (int64_t)st.d[s][0] + (int64_t)st.d[s][1] * px + (int64_t)st.d[s][2] * pxy;
}
__device__ __forceinline__ unsigned long long dict_mix(unsigned long long h, unsigned long long v) {
  h = (h ^ v) * 0xFF51AFD7ED558CCDull;
  h ^= h >> 29;
  h *= 0xC4CEB9FE1A85EC53ull;
  return h ^ (h >> 32);
}

// pass 1: hash of a row's S coefficients -> open-addressing table; cls[row] = the slot its hash lives in.  Rows whose
// hash is already there (all but a handful) only read.  state[0] counts the distinct hashes; more than `cap`, or a
// probe sequence longer than 64, refuses the dictionary (state[1] = -1) and lets the remaining rows return at once.
__global__ __launch_bounds__(256) void k_dict_build(const double* __restrict__ val, int64_t ld, int sym, Stencil st, int px,
                                                    int64_t pxy, int64_t n, unsigned long long* keys, int64_t* rep,
                                                    uint16_t* __restrict__ cls, int* state, int cap) {
  const int S = st.count;
  volatile int* vstate = state;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
    if (vstate[1] < 0) return;
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    for (int s = 0; s < S; ++s)
      h = dict_mix(h, (unsigned long long)__double_as_longlong(sell_coef(val, ld, sym, S, s, row, sell_off(st, s, px, pxy), n)));
    if (h == 0ull) h = 1ull;
    unsigned slot = (unsigned)(h >> 40) & (PPH_DICT_HASH - 1);
    for (int probe = 0;; ++probe) {
      unsigned long long k = __atomic_load_n(keys + slot, __ATOMIC_RELAXED);
      if (k == 0ull) {
        k = atomicCAS(keys + slot, 0ull, h);
        if (k == 0ull) {   // this row's hash is new: it represents the class
          rep[slot] = row;
          if (atomicAdd(state, 1) + 1 > cap) atomicExch(state + 1, -1);
          k = h;
        }
      }
      if (k == h) break;
      slot = (slot + 1) & (PPH_DICT_HASH - 1);
      if (probe >= 64) { atomicExch(state + 1, -1); break; }
    }
    cls[row] = (uint16_t)slot;
  }
}

// pass 2 (one workgroup): classes = occupied hash slots in slot order; tab[class] = the coefficients of its
// representative row.  refresh: the classes are known from an earlier assembly, only the table is read again.
__global__ __launch_bounds__(256) void k_dict_table(const double* __restrict__ val, int64_t ld, int sym, Stencil st, int px,
                                                    int64_t pxy, int64_t n, const unsigned long long* __restrict__ keys,
                                                    int64_t* rep, uint16_t* __restrict__ map, double* __restrict__ tab,
                                                    int* state, int cap, int refresh, int ncls) {
  const int S = st.count;
  if (refresh) {
    for (int c = threadIdx.x; c < ncls; c += 256) {
      const int64_t row = rep[PPH_DICT_HASH + c];
      for (int s = 0; s < S; ++s) tab[c * S + s] = sell_coef(val, ld, sym, S, s, row, sell_off(st, s, px, pxy), n);
    }
    if (threadIdx.x == 0) { state[0] = ncls; state[1] = 1; }
    return;
  }
  constexpr int PER = PPH_DICT_HASH / 256;
  __shared__ int first[257];
  int mine = 0;
  for (int k = 0; k < PER; ++k) mine += keys[threadIdx.x * PER + k] != 0ull;
  first[threadIdx.x + 1] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    first[0] = 0;
    for (int t = 1; t <= 256; ++t) first[t] += first[t - 1];
  }
  __syncthreads();
  const int total = first[256];
  if (state[1] < 0 || total > cap) {
    if (threadIdx.x == 0) { state[0] = total; state[1] = -1; }
    return;
  }
  int id = first[threadIdx.x];
  for (int k = 0; k < PER; ++k) {
    const int slot = threadIdx.x * PER + k;
    if (keys[slot] == 0ull) continue;
    const int64_t row = rep[slot];
    map[slot] = (uint16_t)id;
    rep[PPH_DICT_HASH + id] = row;
    for (int s = 0; s < S; ++s) tab[id * S + s] = sell_coef(val, ld, sym, S, s, row, sell_off(st, s, px, pxy), n);
    ++id;
  }
  if (threadIdx.x == 0) { state[0] = total; state[1] = 1; }
}

// pass 3: every row against the table entry of its class, bit for bit (a hash collision, or a row that left its class
// in a re-assembly, refuses the dictionary: state[1] = -2); map != null: hash slots -> classes on the way.  Symmetric
// storage, stencil unrolled, table in LDS: this runs after every assembly.
template <int KIND>
__global__ __launch_bounds__(256) void k_dict_verify_sym(const double* __restrict__ val, int64_t ld, int px, int64_t pxy,
                                                         int64_t n, const uint16_t* __restrict__ map,
                                                         uint16_t* __restrict__ cls, const double* __restrict__ tab,
                                                         int* state, int lds_classes, int walk, int* alarm) {
  using ST = SellSt<KIND>;
  constexpr int S = ST::S, C0 = S / 2;
  extern __shared__ double vtab[];
  if (state[1] != 1) return;
  const int ncls = state[0] < lds_classes ? state[0] : lds_classes;
  for (int i = threadIdx.x; i < ncls * S; i += 256) vtab[i] = tab[i];
  __syncthreads();
  bool bad = false;
  // Row order.  3D (walk > 0; the launcher keeps gridDim.x a multiple of 8): a workgroup takes `walk` consecutive planes
  // at one in-plane chunk of 256 positions, the workgroups of an XCD consecutive chunks - the mirror values of a plane
  // are the stored values of the plane below and of the in-plane neighbours, read a step earlier into the same L2
  // (plain grid-stride order: 1.87 x the stored bytes from the fabric).  2D: grid-stride.
  const int64_t P = (pxy + 255) / 256;
  const int64_t planes = walk > 0 ? n / pxy : 0;
  const int64_t nslab = walk > 0 ? (planes + walk - 1) / walk : 0;
  const int64_t vb = walk > 0 ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int64_t nsteps = walk > 0 ? nslab * P * walk : (n + 255) / 256;
  for (int64_t q = walk > 0 ? vb * walk : vb;; ) {
    if (q >= nsteps) break;
    int64_t row;
    bool in = true;
    if (walk > 0) {
      const int64_t item = q / walk, t = q % walk, slab = item / P, pos = item % P;
      const int64_t z = slab * walk + t, pp = pos * 256 + threadIdx.x;
      in = z < planes && pp < pxy;
      row = z * pxy + pp;
      q += (t == walk - 1) ? (int64_t)(gridDim.x - 1) * walk + 1 : 1;
    } else {
      row = q * 256 + threadIdx.x;
      in = row < n;
      q += gridDim.x;
    }
    if (!in) continue;
    int c = cls[row];
    if (map) { c = map[c]; cls[row] = (uint16_t)c; }
    if (c >= ncls) { bad = true; continue; }
    const double* t = vtab + c * S;
    int slot = 0;
#pragma unroll
    for (int l = 0; l < ST::NL; ++l) {
      const int mask = ST::mask(l);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (!((mask >> d) & 1)) continue;
        double v;
        if (slot >= C0) {
          v = val[(int64_t)(slot - C0) * ld + row];
        } else {
          const int64_t rr = row + (int64_t)(d - 1) + (int64_t)ST::dy(l) * px + (int64_t)ST::dz(l) * pxy;
          v = (rr >= 0 && rr < n) ? val[(int64_t)(S - 1 - slot - C0) * ld + rr] : 0.0;
        }
        bad |= __double_as_longlong(v) != __double_as_longlong(t[slot]);
        ++slot;
      }
    }
  }
  if (bad) {
    atomicExch(state + 1, -2);
    // the host learns of a refusal on the device without a read-back per operator: one word in mapped host memory, looked
    // at when a solve ends (sell_dict_poll)
    if (alarm) __hip_atomic_store(alarm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static void dict_launch_verify(pph_ctx* ctx, const Sell& E, SellDict& D, int64_t n, const uint16_t* map, int lds_classes) {
  const int64_t pxy = (int64_t)E.px * E.py;
  const size_t lds = (size_t)lds_classes * sell_slots(E.kind) * sizeof(double);
  if (E.kind == PPH_CELL_HEX || E.kind == PPH_CELL_TET) {
    const int walk = (E.pz >= 8 && n == pxy * E.pz) ? 8 : 0;
    int grid = sell_grid(n);
    if (walk) {   // a few workgroups per CU, a multiple of 8, no more than there are (slab, chunk) items
      const int64_t items = ((E.pz + walk - 1) / walk) * ((pxy + 255) / 256);
      int64_t g = (int64_t)ctx->num_cus * 4;
      if (g > items) g = items;
      grid = (int)(((g + 7) / 8) * 8);
    }
    if (E.kind == PPH_CELL_HEX)
      hipLaunchKernelGGL(k_dict_verify_sym<PPH_CELL_HEX>, dim3(grid), dim3(256), lds, ctx->stream, E.val, E.ld, E.px, pxy, n, map,
                         D.cls.p, D.tab.p, D.state.p, lds_classes, walk, ctx->dict_alarm_dev);
    else
      hipLaunchKernelGGL(k_dict_verify_sym<PPH_CELL_TET>, dim3(grid), dim3(256), lds, ctx->stream, E.val, E.ld, E.px, pxy, n, map,
                         D.cls.p, D.tab.p, D.state.p, lds_classes, walk, ctx->dict_alarm_dev);
  } else if (E.kind == PPH_CELL_QUAD) {
    hipLaunchKernelGGL(k_dict_verify_sym<PPH_CELL_QUAD>, dim3(sell_grid(n)), dim3(256), lds, ctx->stream, E.val, E.ld, E.px, pxy, n,
                       map, D.cls.p, D.tab.p, D.state.p, lds_classes, 0, ctx->dict_alarm_dev);
  } else {
    hipLaunchKernelGGL(k_dict_verify_sym<PPH_CELL_TRI>, dim3(sell_grid(n)), dim3(256), lds, ctx->stream, E.val, E.ld, E.px, pxy, n,
                       map, D.cls.p, D.tab.p, D.state.p, lds_classes, 0, ctx->dict_alarm_dev);
  }
}

// rows of planes 2 .. pz - 4 whose class differs from the class of the row one plane above: none = the class of an in-plane
// position is constant on planes 2 .. pz - 3 (SellDict::zconst; k_spmv_dict_walk then loads no class words inside that range)
__global__ __launch_bounds__(256) void k_dict_zconst(const uint16_t* __restrict__ cls, int64_t pxy, int pz, int* state) {
  if (state[1] != 1 || pz < 8) return;
  const int64_t lo = 2 * pxy, hi = (int64_t)(pz - 3) * pxy;
  int bad = 0;
  for (int64_t r = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += (int64_t)gridDim.x * blockDim.x)
    bad |= cls[r] != cls[r + pxy];
  if (bad) atomicAdd(state + 2, 1);
}

int sell_dict_update(pph_ctx* ctx, Sell* E, SellDict& D, int64_t n) {
  E->dict = nullptr;
  const bool want = ctx->sell_dict && E->val && E->sym && n >= ctx->sell_dict_min_rows && ctx->sell_rpt != 1;
  if (!want) { D.on = false; D.checked = false; return PPH_OK; }
  const bool same = D.val == E->val && D.n == n && D.px == E->px && D.py == E->py && D.bc_epoch == ctx->bc_epoch &&
                    D.cap == ctx->sell_dict_cap;
  // refused for this mesh and these Dirichlet sets (too many distinct rows / a failed check): not tried again.  A dictionary
  // that was only switched OFF (option sell_dict 0, sell_sym 0 or sell_rpt 1 at an assembly in between) is rebuilt.
  if (same && D.tried && !D.on && D.status < 0) return PPH_OK;
  const Stencil st = make_stencil(E->kind);
  const int64_t pxy = (int64_t)E->px * E->py;
  const int grid = sell_grid(n);
  const int cap = ctx->sell_dict_cap < PPH_DICT_CAP ? (ctx->sell_dict_cap < 1 ? 1 : ctx->sell_dict_cap) : PPH_DICT_CAP;
  if (same && D.on && D.checked) {
    // re-assembly whose kernel compared every entry it stored with the table read from the mini operator (facts A and B above)
    D.checked = false;
    E->dict = &D;
    return PPH_OK;
  }
  if (same && D.on) {
    // re-assembly: same classes expected - re-read the table from the representatives, check every row
    hipLaunchKernelGGL(k_dict_table, dim3(1), dim3(256), 0, ctx->stream, E->val, E->ld, E->sym, st, E->px, pxy, n, D.keys.p,
                       D.rep.p, D.map.p, D.tab.p, D.state.p, cap, 1, D.ncls);
    dict_launch_verify(ctx, *E, D, n, nullptr, D.ncls);
    PPH_HIP(ctx, hipGetLastError());
    E->dict = &D;
    return PPH_OK;
  }
  const bool was_on = D.on;
  const int was_ncls = D.ncls;
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (first build only: its cost is reported apart, pph_get_timers out[20])
  const auto t_build0 = std::chrono::steady_clock::now();
  PPH_TRY(D.cls.alloc(ctx, (size_t)E->ld));
  PPH_TRY(D.keys.alloc(ctx, (size_t)PPH_DICT_HASH));
  PPH_TRY(D.rep.alloc(ctx, (size_t)(PPH_DICT_HASH + PPH_DICT_CAP)));
  PPH_TRY(D.map.alloc(ctx, (size_t)PPH_DICT_HASH));
  PPH_TRY(D.tab.alloc(ctx, (size_t)PPH_DICT_CAP * 27));
  PPH_TRY(D.state.alloc(ctx, 4));
  PPH_HIP(ctx, hipMemsetAsync(D.cls.p, 0, (size_t)E->ld * sizeof(uint16_t), ctx->stream));
  PPH_HIP(ctx, hipMemsetAsync(D.keys.p, 0, (size_t)PPH_DICT_HASH * sizeof(unsigned long long), ctx->stream));
  PPH_HIP(ctx, hipMemsetAsync(D.state.p, 0, 4 * sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_dict_build, dim3(grid), dim3(256), 0, ctx->stream, E->val, E->ld, E->sym, st, E->px, pxy, n, D.keys.p,
                     D.rep.p, D.cls.p, D.state.p, cap);
  hipLaunchKernelGGL(k_dict_table, dim3(1), dim3(256), 0, ctx->stream, E->val, E->ld, E->sym, st, E->px, pxy, n, D.keys.p,
                     D.rep.p, D.map.p, D.tab.p, D.state.p, cap, 0, 0);
  dict_launch_verify(ctx, *E, D, n, D.map.p, cap);
  if (E->kind == PPH_CELL_HEX && n == pxy * E->pz)
    hipLaunchKernelGGL(k_dict_zconst, dim3(grid), dim3(256), 0, ctx->stream, D.cls.p, pxy, E->pz, D.state.p);
  PPH_HIP(ctx, hipGetLastError());
  int h[3] = {0, 0, 0};
  PPH_HIP(ctx, hipMemcpyAsync(h, D.state.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->t_dict_build += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
  ctx->n_dict_build++;
  D.ncls = h[0];
  D.on = h[1] == 1;
  D.status = h[1];
  D.zconst = D.on && E->kind == PPH_CELL_HEX && n == pxy * E->pz && E->pz >= 8 && h[2] == 0;
  D.tried = true;
  D.adj_ok = false; D.checked = false;     // (the group is rebuilt by the caller: dict_group_build)
  D.val = E->val; D.n = n; D.px = E->px; D.py = E->py; D.bc_epoch = ctx->bc_epoch; D.cap = ctx->sell_dict_cap;
  if (D.on != was_on || D.ncls != was_ncls) la_release_graphs(ctx);   // captured launches carry the old class count
  if (D.on) E->dict = &D;
  return PPH_OK;
}

// ------------------------------------------------------------------------------------------------
// Check fused into the assembly (round 4).  k_dict_verify_sym read every stored value again (1.9 GB per fine operator, 1.6 ms
// per 256^3 step).  The same guarantee - every coefficient a product takes from the table equals, bit for bit, the one the
// stored-value kernel would load - follows from two cheaper facts:
//   A. every STORED entry of every row equals the stored half of its class's table row: checked by the assembly kernel on
//      the value it is about to store (k_asm_node2 in check mode, pph_assemble.hip);
//   B. for every pair of classes (c, c') that meet - some row of class c has a row of class c' at its lower slot s - the lower
//      half of the table agrees with the mirror: tab[c][s] == tab[c'][S - 1 - s] (and == 0 where the slot leaves [0, n)):
//      which classes meet where depends on the class arrays alone, is recorded once at build time (k_dict_adj) and checked
//      per assembly on the tables (k_dict_tables_check, one workgroup).
// A lower coefficient of row r is the mirror entry stored with row r + o; A makes that tab[cls[r + o]][S - 1 - s], B makes it
// tab[cls[r]][s].  For A the table must exist BEFORE the assembly proper: the representative row of every class and the
// rows its lower slots mirror are assembled first into a mini operator (k_asm_node2 in listed mode, a few hundred rows) and
// the tables are read from there (DictGroup).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dict_adj(const uint16_t* __restrict__ cls, Stencil st, int px, int64_t pxy, int64_t n,
                                                  uint32_t* adj, const int* __restrict__ state) {
  if (state[1] != 1) return;
  const int C0 = st.count / 2;
  const int64_t nr = (n + 63) & ~(int64_t)63;      // (whole waves: the wave-uniform shortcut below needs all 64 lanes)
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < nr; row += (int64_t)gridDim.x * blockDim.x) {
    const bool live = row < n;
    const int c = live ? (int)cls[row] : -1;
    for (int s = 0; s < C0; ++s) {
      const int64_t rr = row + sell_off(st, s, px, pxy);
      const int cp = !live ? -1 : ((rr >= 0 && rr < n) ? (int)cls[rr] : PPH_DICT_CAP);
      // almost every wave holds ONE (class, neighbour class) pair per slot: one lane looks the bit up, not 64 (17 M rows
      // polling the same few words took 2.9 ms per operator)
      const int c0 = __builtin_amdgcn_readfirstlane(c), cp0 = __builtin_amdgcn_readfirstlane(cp);
      const bool uni = __all(c == c0 && cp == cp0);
      if (uni && (threadIdx.x & 63) != 0) continue;
      if (!live) continue;
      uint32_t* w = adj + ((int64_t)c * C0 + s) * PPH_DICT_ADJW + (cp >> 5);
      const uint32_t bit = 1u << (cp & 31);
      if (!(__atomic_load_n(w, __ATOMIC_RELAXED) & bit)) atomicOr(w, bit);
    }
  }
}

// tab[c][s] <- mini[src[c][s]] (0 where src < 0), then fact B on the finished table.  One workgroup per dictionary.
__global__ __launch_bounds__(256) void k_dict_tables_check(const double* __restrict__ mini, const int32_t* __restrict__ src,
                                                           double* __restrict__ tab, const uint32_t* __restrict__ adj, int S,
                                                           int ncls, int* state, int* alarm) {
  const int C0 = S / 2;
  for (int i = threadIdx.x; i < ncls * S; i += 256) tab[i] = src[i] >= 0 ? mini[src[i]] : 0.0;
  __syncthreads();
  bool bad = false;
  for (int i = threadIdx.x; i < ncls * C0; i += 256) {
    const int c = i / C0, s = i % C0;
    const long long mine = __double_as_longlong(tab[c * S + s]);
    const uint32_t* w = adj + (int64_t)i * PPH_DICT_ADJW;
    for (int cp = 0; cp <= PPH_DICT_CAP; ++cp) {
      if (!((w[cp >> 5] >> (cp & 31)) & 1u)) continue;
      const long long other = cp == PPH_DICT_CAP ? 0ll : (cp < ncls ? __double_as_longlong(tab[cp * S + (S - 1 - s)]) : ~mine);
      bad |= mine != other;
    }
  }
  if (threadIdx.x == 0) { state[0] = ncls; }
  if (bad) {
    atomicExch(state + 1, -2);
    if (alarm) __hip_atomic_store(alarm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// After the dictionaries of a group were built: the adjacency bit sets, the list of rows whose entries fill the tables and,
// per dictionary, where each table entry sits in the mini operator.  Synchronises (build path only).
int dict_group_build(pph_ctx* ctx, DictGroup& G, SellDict* const* dicts, int nd, const Sell& shape, int64_t n) {
  G.ok = false;
  const Stencil st = make_stencil(shape.kind);
  const int S = st.count, C0 = S / 2, SS = S - C0;
  const int64_t pxy = (int64_t)shape.px * shape.py;
  if (!ctx->dict_fuse || !shape.sym) return PPH_OK;
  for (int d = 0; d < nd; ++d)
    if (dicts[d] && (!dicts[d]->on || dicts[d]->ncls > PPH_DICT_FUSE_CAP)) return PPH_OK;
  std::vector<uint32_t> list;
  std::vector<int64_t> keys;                      // node of list entry i (linear search is fine: a few hundred entries)
  auto index_of = [&](int64_t node) -> int {
    for (size_t i = 0; i < keys.size(); ++i) if (keys[i] == node) return (int)i;
    keys.push_back(node);
    list.push_back((uint32_t)node);
    return (int)keys.size() - 1;
  };
  std::vector<std::vector<int64_t>> reps((size_t)nd);
  for (int d = 0; d < nd; ++d) {
    SellDict* D = dicts[d];
    if (!D) continue;
    reps[(size_t)d].resize((size_t)D->ncls);
    PPH_HIP(ctx, hipMemcpyAsync(reps[(size_t)d].data(), D->rep.p + PPH_DICT_HASH, sizeof(int64_t) * (size_t)D->ncls, hipMemcpyDeviceToHost,
                                ctx->stream));
  }
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // first pass: the list (so that the mini operator's leading dimension is known), second pass: the source indices
  for (int pass = 0; pass < 2; ++pass) {
    const int64_t ldm = sell_ld((int64_t)list.size());
    for (int d = 0; d < nd; ++d) {
      SellDict* D = dicts[d];
      if (!D) continue;
      std::vector<int32_t> src((size_t)D->ncls * S, -1);
      for (int c = 0; c < D->ncls; ++c) {
        const int64_t r = reps[(size_t)d][(size_t)c];
        const int ir = index_of(r);
        for (int s = 0; s < S; ++s) {
          if (s >= C0) { src[(size_t)c * S + s] = (int32_t)((int64_t)(s - C0) * ldm + ir); continue; }
          const int64_t rr = r + (int64_t)st.d[s][0] + (int64_t)st.d[s][1] * shape.px + (int64_t)st.d[s][2] * pxy;
          if (rr < 0 || rr >= n) continue;
          src[(size_t)c * S + s] = (int32_t)((int64_t)(S - 1 - s - C0) * ldm + index_of(rr));
        }
      }
      if (pass == 1) {
        PPH_TRY(D->src.alloc(ctx, src.size()));
        PPH_HIP(ctx, hipMemcpyAsync(D->src.p, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice, ctx->stream));
        PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));     // (src is a local)
      }
    }
  }
  G.nlist = (int)list.size();
  G.ldm = sell_ld((int64_t)G.nlist);
  G.nd = nd;
  PPH_TRY(G.list.alloc(ctx, (size_t)G.ldm));
  list.resize((size_t)G.ldm, list.empty() ? 0u : list[0]);     // padding rows repeat a valid node
  PPH_HIP(ctx, hipMemcpyAsync(G.list.p, list.data(), sizeof(uint32_t) * list.size(), hipMemcpyHostToDevice, ctx->stream));
  PPH_TRY(G.mini.alloc(ctx, (size_t)3 * SS * (size_t)G.ldm));
  PPH_HIP(ctx, hipMemsetAsync(G.mini.p, 0, sizeof(double) * (size_t)3 * SS * (size_t)G.ldm, ctx->stream));
  for (int d = 0; d < nd; ++d) {
    SellDict* D = dicts[d];
    G.ncls[d] = D ? D->ncls : 0;
    G.cls_of[d] = D ? (const void*)D->cls.p : nullptr;
    if (!D) continue;
    PPH_TRY(D->adj.alloc(ctx, (size_t)D->ncls * C0 * PPH_DICT_ADJW));
    PPH_HIP(ctx, hipMemsetAsync(D->adj.p, 0, sizeof(uint32_t) * (size_t)D->ncls * C0 * PPH_DICT_ADJW, ctx->stream));
    hipLaunchKernelGGL(k_dict_adj, dim3(sell_grid(n)), dim3(256), 0, ctx->stream, D->cls.p, st, shape.px, pxy, n, D->adj.p, D->state.p);
    D->adj_ok = true;
  }
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PPH_HIP(ctx, hipGetLastError());
  G.ok = true;
  return PPH_OK;
}

// before the assembly proper, after the listed launch has filled G.mini: tables + fact B, one small launch per dictionary
int dict_group_tables(pph_ctx* ctx, DictGroup& G, SellDict* const* dicts, int nd, const Sell& shape) {
  const int S = sell_slots(shape.kind), SS = S - S / 2;
  for (int d = 0; d < nd; ++d) {
    SellDict* D = dicts[d];
    if (!D) continue;
    hipLaunchKernelGGL(k_dict_tables_check, dim3(1), dim3(256), 0, ctx->stream, G.mini.p + (size_t)d * SS * (size_t)G.ldm, D->src.p, D->tab.p,
                       D->adj.p, S, D->ncls, D->state.p, ctx->dict_alarm_dev);
    D->checked = true;
  }
  PPH_HIP(ctx, hipGetLastError());
  return PPH_OK;
}

// A per-assembly check that refused a dictionary ON THE DEVICE (state[1] = -2) makes every product launch take its
// one-row-at-a-time fallback - correct, but far slower than the plain kernel the refusal is meant to fall back to, and the
// byte accounting keeps counting 2 B per row.  The check kernels raise one word in mapped host memory; a solve looks at it
// when it ends (it has synchronised by then) and, if raised, reads the status words back and retires the refused
// dictionaries on the host as well.
int sell_dict_poll(pph_ctx* ctx) {
  if (!ctx->dict_alarm || !*ctx->dict_alarm) return PPH_OK;
  *ctx->dict_alarm = 0;
  auto retire = [&](Sell& E, SellDict& D) -> int {
    if (!D.on || !D.state.p) return PPH_OK;
    int h[2] = {0, 0};
    PPH_HIP(ctx, hipMemcpyAsync(h, D.state.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[1] == 1) return PPH_OK;
    D.on = false; D.status = h[1];
    if (E.dict == &D) E.dict = nullptr;
    return 1;
  };
  int n = 0, r;
  if ((r = retire(ctx->S11, ctx->D11)) < 0) return r; n += r;
  if ((r = retire(ctx->S22, ctx->D22)) < 0) return r; n += r;
  if ((r = retire(ctx->S12, ctx->D12)) < 0) return r; n += r;
  if (ctx->S21.dict && !ctx->D12.on) ctx->S21.dict = nullptr;
  for (size_t l = 0; l < ctx->mg.size(); ++l)
    for (int f = 0; f < 2; ++f) {
      MgLevel& L = ctx->mg[l];
      if (l == 0) { if (L.ell[f].dict && !L.ell[f].dict->on) L.ell[f].dict = nullptr; continue; }   // level 0 aliases S11 / S22
      if ((r = retire(L.ell[f], L.dict[f])) < 0) return r; n += r;
    }
  if (n) la_release_graphs(ctx);   // captured launches carry the dictionary kernels
  return PPH_OK;
}

// ------------------------------------------------------------------------------------------------
// CSR <-> SELL (values only; the pattern is the closed-form stencil pattern of pph_mesh.hip)
// ------------------------------------------------------------------------------------------------
template <bool TO_SELL>
__global__ __launch_bounds__(256) void k_sell_convert(const int64_t* __restrict__ rowptr, double* __restrict__ csr,
                                                      double* __restrict__ ell, int64_t ld, Stencil st, int px, int py,
                                                      int pz, int64_t n, int sym, int glo, int ghi) {
  const int c0 = st.count / 2;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(row % px);
    const int64_t t = row / px;
    const int j = (int)(t % py), k = (int)(t / py);
    int64_t o = rowptr[row];
    for (int s = 0; s < st.count; ++s) {
      const int ii = i + st.d[s][0], jj = j + st.d[s][1], kk = k + st.d[s][2];
      const bool in = ii >= 0 && ii < px && jj >= 0 && jj < py && kk >= 0 && kk < pz;
      // CSR export of a symmetric slab operator: ghost rows are empty in CSR (their stored entries are the mirrors of
      // owned rows' entries, pph_assemble.hip: fuse_elim_diag)
      const bool ghost_row = !TO_SELL && sym && ((glo && k == 0) || (ghi && k == pz - 1));
      if (ghost_row) {
        if (in) csr[o] = 0.0;
      } else if (!sym) {
        if (TO_SELL) ell[(int64_t)s * ld + row] = in ? csr[o] : 0.0;
        else if (in) csr[o] = ell[(int64_t)s * ld + row];
      } else if (s >= c0) {
        if (TO_SELL) ell[(int64_t)(s - c0) * ld + row] = in ? csr[o] : 0.0;
        else if (in) csr[o] = ell[(int64_t)(s - c0) * ld + row];
      } else if (!TO_SELL && in) {
        // lower entry (row, col) = upper entry (col, row), stored with the mirror slot at row col
        const int64_t colr = ii + (int64_t)px * (jj + (int64_t)py * kk);
        csr[o] = ell[(int64_t)(st.count - 1 - s - c0) * ld + colr];
      }
      o += in ? 1 : 0;
    }
  }
}

// (re)allocates `buf` for a stencil-ELL operator on `mesh` with the padding rows [n, ld) zeroed, returns the view
int sell_alloc(pph_ctx* ctx, const MeshData& mesh, DevBuf<double>& buf, Sell* out, int sym) {
  const int S = sell_stored(mesh.kind, sym);
  const int64_t ld = sell_ld(mesh.n);
  const size_t want = (size_t)S * (size_t)ld;
  if (buf.n != want || !buf.p) {
    PPH_TRY(buf.alloc(ctx, want));
    PPH_HIP(ctx, hipMemsetAsync(buf.p, 0, want * sizeof(double), ctx->stream));   // padding rows stay zero for good
  }
  out->val = buf.p; out->ld = ld; out->kind = mesh.kind; out->px = mesh.px; out->py = mesh.py; out->pz = mesh.pzl;
  out->sym = sym;
  out->dict = nullptr;   // whoever fills the values next decides (sell_dict_update): a dictionary never outlives the values it was checked on
  return PPH_OK;
}

int sell_from_csr(pph_ctx* ctx, const MeshData& mesh, const double* csr_val, DevBuf<double>& buf, Sell* out, int sym) {
  PPH_TRY(sell_alloc(ctx, mesh, buf, out, sym));
  hipLaunchKernelGGL(k_sell_convert<true>, dim3(sell_grid(mesh.n)), dim3(256), 0, ctx->stream, mesh.rowptr.p,
                     const_cast<double*>(csr_val), buf.p, out->ld, make_stencil(mesh.kind), mesh.px, mesh.py, mesh.pzl,
                     mesh.n, sym, 0, 0);
  PPH_HIP(ctx, hipGetLastError());
  return PPH_OK;
}

int sell_to_csr(pph_ctx* ctx, const MeshData& mesh, const Sell& E, double* csr_val) {
  hipLaunchKernelGGL(k_sell_convert<false>, dim3(sell_grid(mesh.n)), dim3(256), 0, ctx->stream, mesh.rowptr.p, csr_val,
                     const_cast<double*>(E.val), E.ld, make_stencil(mesh.kind), mesh.px, mesh.py, mesh.pzl, mesh.n, E.sym,
                     mesh.glo ? 1 : 0, mesh.ghi ? 1 : 0);
  PPH_HIP(ctx, hipGetLastError());
  return PPH_OK;
}
