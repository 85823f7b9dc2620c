// Internal declarations shared by the translation units of libperphil_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/perphil_hip.h"

#define PPH_WAVE 64

struct PphNcclId { char internal[128]; };  // layout of ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)

// ---------------------------------------------------------------------------------------------
// error plumbing: HIP failures become a negative status + message, never an abort
// ---------------------------------------------------------------------------------------------
void pph_set_error(pph_ctx* ctx, const char* fmt, ...);

#define PPH_HIP(ctx, call)                                                                     \
  do {                                                                                         \
    hipError_t e__ = (call);                                                                   \
    if (e__ != hipSuccess) {                                                                   \
      pph_set_error((ctx), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,   \
                    __LINE__);                                                                 \
      return (e__ == hipErrorOutOfMemory) ? PPH_ERR_NOMEM : PPH_ERR_HIP;                       \
    }                                                                                          \
  } while (0)

#define PPH_TRY(expr)           \
  do {                          \
    int s__ = (expr);           \
    if (s__ < 0) return s__;    \
  } while (0)

#define PPH_REQUIRE(ctx, cond, ...)        \
  do {                                     \
    if (!(cond)) {                         \
      pph_set_error((ctx), __VA_ARGS__);   \
      return PPH_ERR_INVALID;              \
    }                                      \
  } while (0)

// ---------------------------------------------------------------------------------------------
// device buffers
// ---------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(pph_ctx* ctx, size_t count);
  void release();
};

// index set of a reduction: [off1, off1+len1) U [off2, off2+len2) (owned entries of one or two fields)
struct Seg { int64_t off1, len1, off2, len2; };

struct MeshData;

// Stencil-ELL ("SELL") operator storage of a scalar block on a structured level: val[s * ld + row], slot s = the
// s-th offset of the cell kind's stencil in ascending (dz,dy,dx) order - the column order of the CSR pattern -
// and an explicit 0 where the neighbour lies outside the local box.  No column indices and no row pointers:
// col = row + dx + px * (dy + py * dz).  8 B per stored entry instead of 12 B, every load coalesced across rows.
//
// Symmetric storage (sym = 1): the eliminated blocks are symmetric and the stencils are symmetric under negation
// (slot s and S-1-s carry opposite offsets), so only the diagonal and the UPPER slots are stored: val[(s - S/2)][row]
// for s >= S/2.  Entry (r, r + o) with o < 0 is read as entry (r + o, r) from the mirror slot of row r + o: every stored
// value is used by two rows, S/2 + 1 instead of S streams.
#define PPH_SELL_COLUMN_WALK 1000   // sell_zwalk >= this: balanced column walk (pph_sell.hip)
// Row dictionary of a stencil-ELL operator (pph_sell.hip, "sell_dict"): on the reference's meshes (uniform boxes, constant
// coefficients) almost all rows of a block are the SAME 27 numbers - interior rows, rows next to a Dirichlet face, ...
// The dictionary stores every distinct row once (tab[class][slot], all S slots: no mirrors to fetch) and a 2-byte class
// per row; a product then streams 2 B instead of 8 (S/2 + 1) B per row and takes its coefficients from LDS.  Lossless: two
// rows share a class only if all S coefficients the plain kernel would load for them are bitwise equal (checked for every
// row by k_dict_verify after every assembly), the sums run in the plain kernel's order, so products are bit-identical.
// More classes than PPH_DICT_CAP (graded / perturbed meshes, variable coefficients) or a failed check: the operator keeps
// its plain storage and kernel.
#define PPH_DICT_CAP 256      // classes (LDS: 256 x 27 x 8 B = 54 KB)
#define PPH_DICT_HASH 4096    // open-addressing slots of the build
struct SellDict {
  DevBuf<uint16_t> cls;              // [ld] class of a row (hash slot between build and remap)
  DevBuf<unsigned long long> keys;   // [PPH_DICT_HASH] 64-bit row hashes, 0 = empty
  DevBuf<int64_t> rep;               // [PPH_DICT_HASH] a row of the class that won hash slot h; [PPH_DICT_HASH + c]: of class c
  DevBuf<uint16_t> map;              // [PPH_DICT_HASH] hash slot -> class
  DevBuf<double> tab;                // [PPH_DICT_CAP][S]
  DevBuf<int> state;                 // [0] classes  [1] 1 usable / 0 not built / < 0 refused  (read by every product launch)  [2] k_dict_zconst's count
  int ncls = 0;                      // host copy of state[0] after the build
  bool on = false;                   // products use the dictionary
  bool tried = false;                // a build was attempted for the current mesh / Dirichlet sets
  int status = 0;                    // host copy of state[1] after the build
  bool zconst = false;               // hexahedral box: the class of an in-plane position is the same on planes 2 .. pz - 3 (state[2] == 0, k_dict_zconst)
  int64_t n = 0;
  int px = 0, py = 0, bc_epoch = -1, cap = 0;   // ... and the configuration it was built (or refused) for
  const double* val = nullptr;       // the storage it describes
  // check fused into the assembly (round 4, DictGroup below): which classes meet in which lower slot
  DevBuf<uint32_t> adj;              // [ncls][S / 2][PPH_DICT_ADJW] bit c' : a row of class c has a row of class c' at its lower slot s (bit PPH_DICT_CAP: none, outside [0, n))
  DevBuf<int32_t> src;               // [ncls][S]: where tab[c][s] comes from in the group's mini operator (-1: the 0 outside [0, n))
  bool adj_ok = false;
  bool checked = false;              // this assembly's rows were checked by the assembly kernel itself
  void release() { cls.release(); keys.release(); rep.release(); map.release(); tab.release(); state.release(); adj.release(); src.release(); ncls = 0; on = tried = adj_ok = checked = zconst = false; n = 0; val = nullptr; }
};
#define PPH_DICT_ADJW (PPH_DICT_CAP / 32 + 1)
#define PPH_DICT_FUSE_CAP 128      // classes per operator up to which the assembly kernel holds the stored halves of the tables in LDS
// The operators one assembly launch writes together (fine level: A11, A22, A12; a multigrid level: its two operators) and
// what the fused check needs besides their dictionaries: the representative rows of every class and the rows their lower
// slots mirror, assembled FIRST into a mini operator of a few hundred rows by the same kernel (k_asm_node2 in listed mode),
// from which the tables are read; the assembly proper then compares every entry it stores with its row's class entry.
struct DictGroup {
  DevBuf<uint32_t> list;             // node of mini row i
  DevBuf<double> mini;               // [nd][stored slots][ldm]
  int nlist = 0, nd = 0;
  int64_t ldm = 0;
  bool ok = false;
  int ncls[3] = {0, 0, 0};           // class counts the group was built for
  const void* cls_of[3] = {nullptr, nullptr, nullptr};
  void release() { list.release(); mini.release(); nlist = 0; nd = 0; ldm = 0; ok = false; }
};
struct Sell {
  const double* val = nullptr;
  int64_t ld = 0;            // leading dimension: rows rounded up to a multiple of 64
  int kind = -1;             // PPH_CELL_*
  int px = 0, py = 0, pz = 0;  // node dims of the local box
  int sym = 0;               // 1: upper half only (see above)
  const SellDict* dict = nullptr;   // set while the row dictionary of this storage is usable
};
static inline int sell_slots(int kind) {
  return kind == PPH_CELL_QUAD ? 9 : kind == PPH_CELL_TRI ? 7 : kind == PPH_CELL_HEX ? 27 : 15;
}
static inline int sell_stored(int kind, int sym) { return sym ? sell_slots(kind) / 2 + 1 : sell_slots(kind); }
static inline int64_t sell_ld(int64_t n) { return (n + 63) & ~(int64_t)63; }

// stencil of a cell kind as (dy,dz) lines with a 3-bit mask of the dx in {-1,0,+1} present, lines in ascending
// (dz,dy) order: slot numbering = ascending (dz,dy,dx) = the CSR column order (make_stencil)
template <int KIND> struct SellSt;
template <> struct SellSt<PPH_CELL_QUAD> {
  static constexpr int NL = 3, S = 9;
  __host__ __device__ static constexpr int dy(int l) { return l - 1; }
  __host__ __device__ static constexpr int dz(int) { return 0; }
  __host__ __device__ static constexpr int mask(int) { return 7; }
};
template <> struct SellSt<PPH_CELL_TRI> {
  static constexpr int NL = 3, S = 7;
  __host__ __device__ static constexpr int dy(int l) { return l - 1; }
  __host__ __device__ static constexpr int dz(int) { return 0; }
  __host__ __device__ static constexpr int mask(int l) { return l == 0 ? 6 : (l == 1 ? 7 : 3); }
};
template <> struct SellSt<PPH_CELL_HEX> {
  static constexpr int NL = 9, S = 27;
  __host__ __device__ static constexpr int dy(int l) { return l % 3 - 1; }
  __host__ __device__ static constexpr int dz(int l) { return l / 3 - 1; }
  __host__ __device__ static constexpr int mask(int) { return 7; }
};
template <> struct SellSt<PPH_CELL_TET> {
  static constexpr int NL = 9, S = 15;
  __host__ __device__ static constexpr int dy(int l) { return l % 3 - 1; }
  __host__ __device__ static constexpr int dz(int l) { return l / 3 - 1; }
  // (dz,dy): (-1,-1) (-1,0) (-1,1) (0,-1) (0,0) (0,1) (1,-1) (1,0) (1,1)
  __host__ __device__ static constexpr int mask(int l) {
    return (l == 0 || l == 1 || l == 3) ? 3 : (l == 4 ? 7 : ((l == 5 || l == 7 || l == 8) ? 6 : 0));
  }
};

// neighbour offsets of a row of the scalar pattern, in ascending (dz,dy,dx) = ascending column order
struct Stencil {
  int count;
  int8_t d[27][3];
};

static inline Stencil make_stencil(int kind) {
  Stencil s;
  s.count = 0;
  auto push = [&](int dx, int dy, int dz) {
    s.d[s.count][0] = (int8_t)dx; s.d[s.count][1] = (int8_t)dy; s.d[s.count][2] = (int8_t)dz; s.count++;
  };
  if (kind == PPH_CELL_QUAD) {
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) push(dx, dy, 0);
  } else if (kind == PPH_CELL_TRI) {
    // edges: x, y and the "left" diagonal (-1,+1)
    push(0, -1, 0); push(1, -1, 0); push(-1, 0, 0); push(0, 0, 0); push(1, 0, 0); push(-1, 1, 0); push(0, 1, 0);
  } else if (kind == PPH_CELL_HEX) {
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) push(dx, dy, dz);
  } else {
    // Kuhn edges: x, y, z, x+y, x+z, y+z, x+y+z (both signs) + self
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      bool pos = dx >= 0 && dy >= 0 && dz >= 0;
      bool neg = dx <= 0 && dy <= 0 && dz <= 0;
      if (pos || neg) push(dx, dy, dz);
    }
  }
  return s;
}

struct Csr {  // device CSR view (no ownership)
  const int64_t* rowptr = nullptr;
  const int32_t* col = nullptr;
  const double* val = nullptr;
  const float* val32 = nullptr;   // when set: fp32 copy of the values, used instead of `val` (preconditioner operands)
  int64_t nrows = 0;
  int64_t nnz = 0;
  const MeshData* geom = nullptr;  // slab geometry of the rows (halo exchange of x before the product); may be null
  int lanes = 8;    // lanes per row used by the CSR-vector SpMV kernel
  int max_row = 0;  // longest row (0: unknown) - selects the CSR-stream kernel geometry
  Sell ell;         // when ell.val is set the product runs on the stencil-ELL copy of the same matrix
};

// one structured mesh level (local box): geometry, cell->dof map, scalar pattern, K and M
struct MeshData {
  int dim = 0, kind = -1, m = 0;        // m = nodes per cell
  int nx = 0, ny = 0, nz = 0;           // global cells of this level
  int z0 = 0, nzl = 0;                  // local slab: first cell layer, layer count
  int glo = 0, ghi = 0;                 // lowest / highest local node plane is a ghost plane (owned by a neighbour)
  int px = 0, py = 0, pzl = 0;          // local node dims
  int64_t n = 0, ncell = 0, nnzb = 0;   // nodes, cells, nnz of one scalar block
  int max_row = 0;                      // stencil size = longest row of the scalar pattern
  bool km_valid = false;                // K and M hold the integrals of this mesh
  bool pattern_ok = false;              // rowptr / col hold the scalar CSR pattern (built on demand: pph_ensure_pattern)
  bool all_affine = false;              // multilinear cells: every cell has equal parallel edges (exact test at mesh build)
  bool uniform = false;                 // ... and every edge along axis d is (h[d] e_d) to within the rounding of the coordinates themselves
  double hcan[3] = {0, 0, 0};           // (2^-52 x the largest coordinate): the box is uniform, the node kernel integrates on the canonical edges
  DevBuf<double> cx, cy, cz;            // nodal coordinates (SoA)
  DevBuf<int32_t> cells;                // cell -> dof map [ncell][m]
  DevBuf<int64_t> rowptr;               // scalar CSR pattern
  DevBuf<int32_t> col;
  DevBuf<double> K, M;                  // scalar stiffness / mass values
  DevBuf<double> erows;                 // element-matrix rows [cell][a][K row | M row] of the two-pass assembly
  int64_t plane() const { return (int64_t)px * py; }
  int64_t own_begin() const { return glo ? plane() : 0; }           // owned entries = [own_begin, own_end)
  int64_t own_end() const { return n - (ghi ? plane() : 0); }
  void release_geometry() { cx.release(); cy.release(); cz.release(); cells.release(); K.release(); M.release(); erows.release(); }
  void release_all() { release_geometry(); rowptr.release(); col.release(); }
};

// one level of the geometric multigrid hierarchy for a scalar block (structured, coarsening 2)
struct MgLevel {
  MeshData mesh;                 // coarse levels own their mesh; level 0 aliases the context's
  const int64_t* rowptr = nullptr;
  const int32_t* col = nullptr;
  const double* val[2] = {nullptr, nullptr};   // A11-like and A22-like operators on this level
  int64_t n = 0, nnz = 0;
  int px = 0, py = 0, pz = 0;
  DevBuf<double> own_val[2];     // storage of val[] on coarse levels
  DevBuf<double> own_ell[2];     // stencil-ELL storage of the level operators (op_format 1; level 0 aliases the context's)
  SellDict dict[2];              // row dictionaries of own_ell (sell_dict)
  DictGroup dgroup;              // ... and their fused-check group
  Sell ell[2];                   // views used by the products when ell[f].val is set
  DevBuf<float> val32[2];        // fp32 copies of val[] for the smoother / residual SpMVs of the V-cycle
  DevBuf<double> dinv[2];
  DevBuf<uint8_t> mask[2];       // per field: non-zero where the dof is constrained
  DevBuf<uint8_t> rownear;       // rows that need the masks (fused level operators); follows the masks
  DevBuf<uint8_t> rfast[2];      // per field: coarse nodes whose 3^d fine neighbours are all inside, owned and unconstrained (restriction fast path)
  int bc_epoch = -1;             // ctx->bc_epoch the masks / rownear were derived from
  const uint8_t* maskp[2] = {nullptr, nullptr};
  DevBuf<double> x, b, r, d, t, w;  // work vectors of the V-cycle (x, b unused on level 0)
  const MeshData* geom = nullptr;  // slab geometry of this level (level 0: the context's mesh)
  bool replicated = false;       // level holds the whole (global) coarse mesh on every rank
  int gz0 = 0;                   // global index of local node plane 0
  int own_lo = 0, own_hi = 0;    // owned global node planes [own_lo, own_hi)
  double lam[2] = {0, 0};        // upper bound of the spectrum of D^-1 A
};

// captured iteration bodies (pph_la.hip: la_run_graph): valid while every pointer baked into the kernels' arguments is
struct GraphKey {
  const void* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int64_t n = 0;
  int slot = 0, tag = 0;
  uint64_t epoch = 0;
  bool operator==(const GraphKey& o) const {
    for (int i = 0; i < 8; ++i) if (p[i] != o.p[i]) return false;
    return n == o.n && slot == o.slot && tag == o.tag && epoch == o.epoch;
  }
};
struct GraphEntry { GraphKey key; hipGraphExec_t exec = nullptr; uint64_t used = 0; };

// ILU(0) factors of one CSR operator with the level schedule of its rows (pph_ilu.hip)
struct IluData {
  const int64_t* rowptr = nullptr;
  const int32_t* col = nullptr;
  int64_t nrows = 0, nnz = 0;
  DevBuf<double> lu;             // strict lower part of L (unit diagonal implied) and U on the pattern of A
  DevBuf<int64_t> diag;          // position of the diagonal entry of every row
  DevBuf<int32_t> perm;          // rows sorted by level
  DevBuf<double> y;
  std::vector<int64_t> levptr;   // offsets of the non-empty levels in perm
  bool struct_ok = false, valid = false;
  uint64_t epoch = 0;
};
void ilu_release(IluData& I);

struct pph_ctx {
  int device = 0;
  int num_cus = 256;                    // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // the fused assembly's own event pair: pph_assemble_dpp returns without waiting, pph_get_timers resolves the time (round 4)
  hipEvent_t ev_asm0 = nullptr, ev_asm1 = nullptr;
  bool asm_time_pending = false;
  // a final reduction left to its consumer (round 4, single context): the CG update sums the p.Ap partials itself, the
  // direction update the r.z partials of the cycle's last kernel - two launches less per CG iteration (la_defer_final)
  struct PendFinal { const double* part = nullptr; int n = 0, slot = -1, copy_src = -1, copy_dst = -1; bool valid = false; } pend;
  bool defer_next_final = false;     // the next la_spmv_jacobi with a dot slot leaves its final reduction pending
  int fold_finals = 1;               // option "fold_finals" 
  // spectral bounds of the multigrid levels (bit patterns): device array kept across assemblies, read back into pinned host
  // memory WITHOUT a synchronisation - the smoother weights are computed on the device (k_mg_weights); the host copies
  // (MgLevel::lam) are filled when something on the host asks for them (mg_lam_host: multi-step Chebyshev, slabs)
  DevBuf<unsigned long long> mg_lam;
  unsigned long long* h_lam = nullptr;   // [2 * 32] pinned
  hipEvent_t ev_lam = nullptr;
  bool mg_lam_pending = false;
  // halo_overlap: the exchange of a product's operand runs on comm_stream while the rows that need no ghost value
  // are computed; 0 off (one launch after the exchange), 1 overlapped, 2 the same three launches without overlap
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_x = nullptr, ev_h = nullptr;
  int halo_overlap = 1;
  int64_t halo_overlap_min_rows = 200000;
  int64_t n_split = 0;                  // products launched split (statistics)
  int max_split_partials = 0;           // largest total a split product wrote so far (statistics)
  int part_cap = 4096;                  // partial sums the launches of ONE product may write in total (= PPH_PART_STRIDE; tests lower it)
  std::string err;

  // mesh (local box)
  MeshData mesh;
  int ghost_lo = 0, ghost_hi = 0;
  int64_t n = 0, nnzb = 0;              // copies of mesh.n / mesh.nnzb
  bool mesh_ok = false, asm_ok = false, mono_ok = false;
  DevBuf<uint8_t> bcmask[2];            // per field: 1 Dirichlet, 2 ghost, 0 free
  DevBuf<double> dinv0[2];              // fused assembly: 1 / diag of A11, A22 ...
  DevBuf<unsigned long long> lam0;      // ... and their spectral bounds (bit patterns), valid when diag0_valid
  bool diag0_valid = false;
  int asm_fused = 1;                    // multilinear two-pass assembly writes the blocks directly (see pph_launch_assemble_fused)
  int asm_affine = 1;                   // tile kernel: cells with equal parallel edges get their (constant) geometry factor once per cell
  int asm_tile_probe = 0;               // timing probe of the tile kernel: 1 stop after phase A, 2 after phase B (wrong results)
  int64_t asm_tile_min_nodes = 30000;   // levels with fewer nodes use the two-pass kernels (asm_tile 2: tile kernel always)
  int asm_node_xmap = 0;                // node kernel: blocks dealt round-robin to the XCDs (0, default) or one contiguous eighth per XCD (1: 2.0 instead of 4.1 GB read at 256^3, but 4.5 instead of 3.5 ms)
  int64_t asm_node_split_min = 200000;  // node kernel on levels of at least this many nodes: straight-line waves and the others in two launches
  int asm_uniform = 1;                  // node kernel on a uniform box (MeshData::uniform): canonical edges instead of coordinate loads
  int asm_node_probe = 0;               // timing probes (wrong results): 1 skip the straight-line launch, 2 skip the other
  int asm_node = 1;                     // box meshes, stencil-ELL output: one thread per node, registers only (k_asm_node); 0: tile / two-pass kernels
  int asm_tile_xmap = 0;                // tile kernel: x-adjacent tiles on ONE XCD (both halves of a 128-B line of a slot array meet in one L2)
  int asm_tile = 1;                     // multilinear fused assembly: 1 single-pass tile kernel (no element-row buffer), 0 two-pass
  int asm_ring = 0;                     // > 0 (experiment, slower): fused 3D assembly alternates element and node passes over a ring of cell layers, about asm_ring cells per launch
  int asm_keep_km = 0;                  // 1: the fused pass also stores K and M (two more 8 B/nnz streams); 0: they are integrated on demand (pph_get_csr K/M, Darcy projection)
  DevBuf<uint8_t> rownear;              // 1: the row is constrained / ghost or has a constrained column (needs the masks)
  bool bc_dirty = true;                 // masks changed since rownear / a21_alias were derived from them
  int bc_epoch = 0;                     // counts changes of the Dirichlet sets (multigrid levels cache injected masks)
  DevBuf<double> g[2];                  // per field Dirichlet values (dense, 0 elsewhere)
  DevBuf<double> A11, A22, A12, A21;    // eliminated blocks on the scalar pattern
  bool a21_alias = false;               // both fields share one Dirichlet set: A21 == A12, A21 not stored
  const double* A21p() const { return a21_alias ? A12.p : A21.p; }
  // stencil-ELL copies of the blocks (op_format 1).  The fused assembly writes ONLY these; the CSR arrays above are
  // then materialised on demand (pph_ensure_csr_blocks: export, monolithic CSR, Jacobi / 2x2-block preconditioners)
  DevBuf<double> E11, E22, E12, E21;
  Sell S11, S22, S12, S21;              // views of E* (S21 == S12 when a21_alias)
  SellDict D11, D22, D12;               // their row dictionaries (sell_dict; S21 shares D12 when aliased, else none)
  DictGroup DG;                         // fused-check group of the three
  int dict_fuse = 1;                    // option "sell_dict_fuse": the per-assembly check of every row runs inside the assembly kernel (0: k_dict_verify_sym)
  bool ell_ok = false;                  // S* hold the assembled blocks
  bool csr_ok = false;                  // A11 .. A21 hold the assembled blocks
  DevBuf<double> rhs, u0, sol;          // length 2n
  DevBuf<int64_t> mrowptr;              // monolithic CSR
  DevBuf<int32_t> mcol;
  DevBuf<double> mval;
  double a = 0, b = 0, c = 0;           // k1/mu, beta/mu, k2/mu

  // multi-GPU: cell-slab decomposition along z; communication goes through two callbacks so that the
  // same solver code runs over torch.distributed (gloo in tests, nccl = RCCL on the 8-GPU node)
  int rank = 0, world = 1;
  pph_halo_fn halo_cb = nullptr;
  pph_allreduce_fn allreduce_cb = nullptr;
  void* comm_user = nullptr;
  void* nccl_comm = nullptr;            // RCCL communicator (pph_comm_init_rccl); takes precedence over the callbacks
  std::vector<double> h_stage;          // host staging for vector all-reduces
  bool comm_suspended = false;          // true while working on replicated (non-distributed) coarse levels
  int64_t n_halo = 0;                   // halo exchanges of the last solve
  int64_t n_allreduce = 0;              // all-reduces (scalars and replicated-level vectors) of the last solve
  int comm_status = 0;                  // sticky: PPH_ERR_COMM after the first failed exchange / reduction (pph_comm.hip)
  std::string comm_error;               // its message

  // solver workspace
  DevBuf<double> scal;                  // device scalars / reduction partials
  // optional by-products of the level-0 post-smoothing sweep of the fused V-cycle (set by the caller of mg_vcycle,
  // cleared by it): partial dot product rin . zout over the owned rows -> scal[mg_dot_slot]
  int mg_dot_slot = -1;
  Seg mg_dot_seg = {0, 0, 0, 0};
  bool mg_x0_ready = false;             // zout already holds the pre-smoothed first guess dinv .* rin * w (k_cg_update_dev)
  double* h_scal = nullptr;             // pinned host mirror (mapped + coherent)
  double* h_scal_dev = nullptr;         // its device-side address
  unsigned long long* h_seq = nullptr;  // sequence word behind the mirror: number of the last published fetch
  unsigned long long* h_seq_dev = nullptr;
  unsigned long long pub_seq = 0;       // fetches published so far
  DevBuf<unsigned long long> pub_ctr;   // device-side count of publications (k_publish), so that a publication can be
                                        // replayed from a graph: the sequence number is not a kernel argument
  int merge_allreduce = 1;              // slabs: p.Ap, r.Ap, Ap.Ap and the previous r.r in ONE all-reduce per CG iteration (the host forms the next r.r)
  int64_t graph_cg_max_rows = 0;        // CG iteration bodies are replayed from graphs on systems up to this size (0: never - eager is as fast or faster)
  int use_graphs = 1;                   // Krylov iteration bodies are captured into hipGraphs where possible
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0;
  int64_t n_graph_launch = 0, n_graph_capture = 0;
  int fetch_spin = 1;                   // 1: la_fetch publishes through the mapped mirror and polls (no stream synchronisation)
  std::vector<DevBuf<double>> work;     // named work vectors, grown on demand
  std::vector<MgLevel> mg;              // multigrid hierarchy (level 0 = fine)
  IluData ilu[3];                       // ILU(0) factors: 0 monolithic system, 1 A11, 2 A22 (pc_type ilu)
  DevBuf<double> mg_w;                  // [2 l + which]: 1 / theta of the one-step Chebyshev smoother of level l (device copy:
                                        // kernels read it through a pointer, so captured graphs survive a re-assembly)
  std::vector<double> mg_w_host;
  uint64_t mg_epoch = 0;                // bumped whenever the hierarchy's buffers are (re)allocated: captured graphs die
  bool mg_ok = false;                   // hierarchy values (operators, masks, bounds) match the assembled system
  bool mg_struct_ok = false;            // hierarchy structure (level meshes, patterns, buffers) matches mesh + communicator

  // timers (ms)
  double t_mesh = 0, t_asm = 0, t_bc = 0, t_solve = 0;
  // SpMV accounting of the last solve, per kernel variant (0: k_spmv<G,false>, 1: k_spmv<G,true>):
  // launches, algorithmic bytes (12 nnz + 20 nrows per launch) and, when time_spmv is on, the sum of
  // the per-launch durations taken with HIP event pairs recorded on the context stream
  double t_spmv[2] = {0, 0};
  double spmv_bytes[2] = {0, 0};
  int64_t n_spmv[2] = {0, 0};
  bool time_spmv = false;
  bool time_comm = false;               // option "time_comm": every halo exchange / all-reduce bracketed by an event pair on the stream it runs on (callback transport: host clock around the callback)
  double t_comm[2] = {0, 0};            // [0] halo exchanges, [1] all-reduces: summed durations in ms of the last solve
  int64_t n_comm_timed[2] = {0, 0};
  struct EvPair { hipEvent_t e0, e1; int variant; bool fine; };
  double t_spmv_fine = 0, spmv_bytes_fine = 0;   // the fine-level launches among t_spmv / spmv_bytes
  int64_t n_spmv_fine = 0;
  std::vector<EvPair> ev_pool;          // reusable event pairs
  size_t ev_used = 0;                   // pairs recorded since the last harvest
  int spmv_lanes_override = 0;          // 0: pick from the mean row length
  int spmv_blocks = 0;                  // 0: default persistent grid (1024 workgroups)
  int device_scalars = 0;               // 1: the device-scalar CG branch also over the callback transport (tests)
  int mg_fused = 1;                     // V(1,1) cycles on stencil-ELL levels: fused smoother / transfer kernels and the
                                        // single-workgroup tail (pph_mg.hip); 0: the general kernel-per-operation cycle
  DevBuf<double> mg_tail_pack[2];       // operators / inverse diagonals / masks of the tail levels, packed per assembly
  int mg_tail_lt[2] = {-1, -1};         // first tail level the pack was built for
  int64_t mg_tail_rows = 5000;          // levels with at most this many rows are handled inside the tail kernel
  int coarse_max_it = 500;              // iteration limit of the coarsest-level Jacobi-CG (to rtol 1e-12)
  int coarse_failed = 0;                // host-driven coarsest solves of the last solve that stopped at their iteration limit
  int coarse_on_device = 1;             // coarsest multigrid level (<= 4096 rows): CG inside one workgroup, no host round trips
  int spmv_bench_mode = 0;              // pph_spmv_bench protocol: 0 back-to-back, 1-3 interleaved (see pph_api.hip)
  int64_t mg_replicate_below = 40000;   // slabs: multigrid levels with at most this many global nodes are replicated
  int64_t mg_replicate_rows_per_rank = 40000;   // ... and levels with at most this many nodes PER RANK (and at most mg_replicate_cap global nodes): their slab kernels are shorter than one exchange
  int64_t mg_replicate_cap = 1000000;
  int mg_fp32 = 0;                      // 1: V-cycle SpMVs read fp32 copies of the operator values (8 instead of 12 B per non-zero)
  int asm_kernel = 2;                   // multilinear cells: 2 two-pass (element rows + node gather, default), 1 one-pass node gather, 0 cell-centred atomic scatter-add
  // operator format of the scalar blocks inside the block solves / Picard sweeps: 1 stencil-ELL (pph_sell.hip), 0 CSR
  int op_format = 1;
  int sell_sym_slabs = 1;               // ... also on slabs (0: full storage there)
  int sell_sym = 1;                     // stencil-ELL operators store the diagonal and the upper slots only (symmetric blocks)
  int64_t sell_zwalk_min_chunks = 5500; // levels with fewer 512-row chunks keep the plain chunk order (measured: 128^3 = 4200 chunks loses 4 % with the z-walk, 144^3 = 5950 equal, 160^3 gains 8 %, 192^3 4 %, 256^3 20 %)
  int sell_zwalk = 4;                   // > 0 (symmetric operators, 3D): a workgroup walks this many consecutive node planes at one in-plane position
  int sell_xmap = 1;                    // z-walk: consecutive in-plane positions on one XCD
  int sell_flags = 0;                   // experiments: 1 non-temporal y stores (mode 0), 2 non-temporal loads of the diagonal slot
  int sell_dict = 1;                    // row dictionaries for the stencil-ELL blocks (struct SellDict)
  int64_t sell_dict_min_rows = 1000000; // ... of operators with at least this many rows
  int sell_dict_blocks = 1024, sell_dict_zwalk = -1;   // grid (cap) of a dictionary product; chunk z-walk of k_spmv_sell<DICT> (-1: as sell_zwalk)
  int sell_dict_zconst = 1;             // ... without class loads where the classes are constant along z (SellDict::zconst)
  int sell_dict_walk = 1;               // whole-operator dictionary products of hexahedral blocks: x window in registers (k_spmv_dict_walk)
  int sell_dict_cap = PPH_DICT_CAP;     // classes accepted (tests lower it to force the plain path)
  int sell_rpt = 2, sell_blocks = 0, sell_group = 0;   // SELL SpMV tuning: rows per thread, grid cap, XCD chunk group
  int* dict_alarm = nullptr;            // mapped host word the per-assembly dictionary checks raise on a refusal (sell_dict_poll)
  int* dict_alarm_dev = nullptr;
  double t_dict_build = 0;              // ms spent in first builds of row dictionaries (k_dict_build + table + first check + read-back)
  int n_dict_build = 0;
  DevBuf<double> sell_tmp;              // SELL copy of the matrix last selected by pph_spmv / pph_spmv_bench
  DevBuf<double> post_u;                // nodal field of the last pph_error_norms_sampled call (chunked callers upload it once)
  bool post_u_valid = false;
};

// lanes per row of the CSR-vector SpMV for a matrix with the given mean row length
static inline int pph_pick_lanes(const pph_ctx* ctx, int64_t nnz, int64_t nrows) {
  if (ctx->spmv_lanes_override > 0) return ctx->spmv_lanes_override;
  const double avg = (double)nnz / (double)(nrows > 0 ? nrows : 1);
  return avg > 40 ? 16 : (avg > 12 ? 8 : 4);
}

// ---------------------------------------------------------------------------------------------
// kernels / launchers (pph_mesh.hip, pph_assemble.hip, pph_la.hip)
// ---------------------------------------------------------------------------------------------
int pph_launch_mesh(pph_ctx* ctx, MeshData& mesh);
int pph_launch_pattern(pph_ctx* ctx, int dim, int kind, int px, int py, int pz, DevBuf<int64_t>& rowptr,
                       DevBuf<int32_t>& col, int64_t* nnz_out);
int pph_launch_assemble_KM(pph_ctx* ctx, MeshData& mesh);
int pph_ensure_pattern(pph_ctx* ctx, MeshData& mesh);       // scalar CSR pattern on demand (synchronises when it builds)
int pph_mesh_check_affine(pph_ctx* ctx, MeshData& mesh);   // sets mesh.all_affine (synchronises; mesh build only)
int pph_launch_blocks(pph_ctx* ctx, int monolithic);
bool pph_can_fuse_assembly(const pph_ctx* ctx);
int pph_ensure_csr_blocks(pph_ctx* ctx);   // CSR values of A11, A22, A12 (, A21) from the stencil-ELL copies when missing
int pph_launch_assemble_fused(pph_ctx* ctx, int monolithic);
// A1, A2: CSR value arrays (ell_ld == 0) or stencil-ELL arrays with leading dimension ell_ld
int pph_launch_level_operators(pph_ctx* ctx, MeshData& mesh, const uint8_t* m1, const uint8_t* m2, const uint8_t* near,
                               int same, double coefK1, double coefK2, double coefM, double* A1, double* A2,
                               double* dinv1, double* dinv2, unsigned long long* lam, int64_t ell_ld, int ell_sym,
                               DictGroup* group = nullptr, SellDict* dicts = nullptr, const Sell* views = nullptr);
void pph_launch_row_near(pph_ctx* ctx, const MeshData& mesh, const uint8_t* m1, const uint8_t* m2, uint8_t* out);   // (stencil walk: no CSR pattern)

// linear algebra on the context stream; all results that feed control flow go through ctx->scal
void la_spmv(pph_ctx* ctx, const Csr& A, const double* x, double* y);
void la_spmv_resid(pph_ctx* ctx, const Csr& A, const double* x, const double* b, double* y);  // y = b - A x
// dot_slot >= 0: also scal[dot_slot] = b . y over the rows [dlo, dhi)
// x_ghosts_valid: the ghost planes of x already hold the owners' values (no exchange before the product)
// t = (b ? b - A x : A x) ;  R += sign (t - told) ;  told = t ;  scal[slot] = sum of R^2 over [dlo, dhi)   (sign = +1 with
// b, -1 without: the Picard sweeps' coupling product + residual bookkeeping + norm in one pass; `tmp`: n doubles,
// used when A has no stencil-ELL copy)
// z0 != null (stencil-ELL operators): also z0 = dinv0 .* R * (*w0), the first pre-smoothing of the solve that starts from R
void la_spmv_shift(pph_ctx* ctx, const Csr& A, const double* x, const double* b, double* R, double* told, double* tmp,
                   int slot, int64_t dlo, int64_t dhi, double* z0 = nullptr, const double* dinv0 = nullptr,
                   const double* w0 = nullptr);
void la_spmv_jacobi(pph_ctx* ctx, const Csr& A, const double* x, const double* b, const double* dinv,
                    const double* w /* device */, double* y, int dot_slot = -1, int64_t dlo = 0, int64_t dhi = 0,
                    bool x_ghosts_valid = false);
// y = A x and partial sums of dot(x, y) -> scal slot
// (copy_src >= 0: scal[copy_dst] = scal[copy_src] is done by the final reduction's single workgroup)
// defer: the final reduction of the partial sums is left pending for the kernel that consumes the scalar (la_cg_update_dev;
// any other call that needs it launches it: la_flush_final)
void la_spmv_dot(pph_ctx* ctx, const Csr& A, const double* x, double* y, int slot, int copy_src = -1, int copy_dst = -1, bool defer = false);
void la_flush_final(pph_ctx* ctx);
void la_spmv_dot3(pph_ctx* ctx, const Csr& A, const double* x, const double* r, double* y, int slot, int copy_src = -1,
                  int copy_dst = -1);   // + r.y and y.y in slot + 1, slot + 2 (stencil-ELL operators)
// publication of scal[slot .. slot + count) to the host mirror (enqueue) / wait for the last publication
void la_publish(pph_ctx* ctx, int slot, int count);
int la_wait_published(pph_ctx* ctx);
// runs `body` (kernel launches on the context stream only) through a captured hipGraph cached under `key`
// (publishes: the body ends with one la_publish, which every replay repeats)
int la_run_graph(pph_ctx* ctx, const GraphKey& key, const std::function<int()>& body, bool publishes = true);
int ilu_factor(pph_ctx* ctx, IluData& I, const Csr& A);
int ilu_apply(pph_ctx* ctx, IluData& I, const double* r, double* z);
void la_release_graphs(pph_ctx* ctx);
void la_set(pph_ctx* ctx, double* x, double v, int64_t n);
void la_copy(pph_ctx* ctx, double* dst, const double* src, int64_t n);
void la_axpy(pph_ctx* ctx, double* y, double alpha, const double* x, int64_t n);          // y += alpha x
void la_axpby(pph_ctx* ctx, double* y, double alpha, const double* x, double beta, int64_t n);  // y = alpha x + beta y
void la_scale(pph_ctx* ctx, double* y, double alpha, int64_t n);
void la_pointwise_mult(pph_ctx* ctx, double* z, const double* d, const double* r, int64_t n);  // z = d .* r
void la_sub(pph_ctx* ctx, double* z, const double* a, const double* b, int64_t n);          // z = a - b
bool la_device_scalars(const pph_ctx* ctx);
int la_reduce_device(pph_ctx* ctx, int slot, int count);
int la_fetch_raw(pph_ctx* ctx, int slot, int count);
// pub_count > 0: the caller would publish scal[pub_slot .. + pub_count) right after (la_publish); true = done in the same launch
bool la_cg_update_dev(pph_ctx* ctx, double* x, double* r, const double* p, const double* q, int slot_num, int slot_den,
                      int64_t n, int slot_out, Seg sg, double* z0 = nullptr, const double* dinv0 = nullptr,
                      const double* w0 = nullptr, int slot_bad = -1, int pub_slot = -1, int pub_count = 0);   // slot_bad >= 0: scal[slot_bad] += 1 when p.Ap is 0 / NaN
void la_p_update_dev(pph_ctx* ctx, double* p, const double* z, int slot_num, int slot_den, int64_t n);
void la_shift(pph_ctx* ctx, double* R, double* told, const double* tnew, double sign, int64_t n);  // R += sign (tnew - told); told = tnew
void la_block2_apply(pph_ctx* ctx, double* z, const double* binv /*[4][n]*/, const double* r, int64_t n);
// k dots in one pass: out[slot+i] = dot(V_i, w), i < k  (V_i = V + i*ld)
void la_mdot(pph_ctx* ctx, const double* V, int64_t ld, int k, const double* w, int64_t n, int slot);
void la_mdot_seg(pph_ctx* ctx, const double* V, int64_t ld, int k, const double* w, Seg sg, int slot);
void la_dot2_seg(pph_ctx* ctx, const double* x, const double* y, const double* z, Seg sg, int slot);
// w -= sum_i h[i] V_i   (h on host)
void la_maxpy_neg(pph_ctx* ctx, double* w, const double* V, int64_t ld, int k, const double* h, int64_t n);
// x += sum_i y[i] V_i
void la_maxpy(pph_ctx* ctx, double* x, const double* V, int64_t ld, int k, const double* y, int64_t n);
void la_dot(pph_ctx* ctx, const double* x, const double* y, int64_t n, int slot);
void la_dot2(pph_ctx* ctx, const double* x, const double* y, const double* z, int64_t n, int slot);  // x.y , z.z
// fused CG update with Jacobi-type PC: x += alpha p; r -= alpha q; z = dinv.*r (dinv may be null: z=r);
// scal[slot] = r.z, scal[slot+1] = z.z
void la_cg_update(pph_ctx* ctx, double* x, double* r, double* z, const double* p, const double* q,
                  const double* dinv, double alpha, int64_t n, int slot, Seg sg);
void la_extract_diag_inv(pph_ctx* ctx, const Csr& A, double* dinv);
// fetch `count` reduction results starting at slot into ctx->h_scal (synchronises the stream)
int la_fetch(pph_ctx* ctx, int slot, int count);
// ghost planes of v <- owner's values (no-op without neighbours / communicator)
int la_halo(pph_ctx* ctx, const MeshData& g, double* v, hipStream_t on = nullptr, hipEvent_t x_ready = nullptr);
// element-wise sum over all ranks of a device vector (small coarse-level vectors)
int la_allreduce_vec(pph_ctx* ctx, double* v, int64_t n);
int comm_allreduce_device(pph_ctx* ctx, double* dev, int64_t count);
int comm_allreduce_host(pph_ctx* ctx, double* vals, int64_t count);
void comm_release(pph_ctx* ctx);
// adds the elapsed times of all event pairs recorded since the last call to ctx->t_spmv (synchronises)
void la_harvest_spmv_times(pph_ctx* ctx);
void la_reset_spmv_stats(pph_ctx* ctx);

// stencil-ELL operator format (pph_sell.hip)
int sell_spmv(pph_ctx* ctx, const Sell& E, int64_t n, int mode, const double* x, const double* b, const double* dinv,
              const double* w /* device */, double* y, double* part, int64_t dlo = 0, int64_t dhi = 0, double* aux = nullptr, double* z0 = nullptr,
              int64_t cbeg = 0, int64_t cend = -1, int grid_cap = 0);
#ifdef __HIPCC__
// two consecutive rows' entries of a vector as ONE 16-byte access per lane (8-byte alignment is enough for the hardware):
// a wave then touches 1 KB of consecutive bytes per instruction - with one 8-byte store per row, each of the two store
// instructions of a row pair writes every other 8 bytes of the lines it touches (byte-masked partial writes)
typedef double pph_d2 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void sell_st2(double* p, double a, double b) {
  pph_d2 v;
  v.x = a; v.y = b;
  *reinterpret_cast<pph_d2*>(p) = v;
}
__device__ __forceinline__ void sell_ld2(const double* p, double& a, double& b) {
  const pph_d2 v = *reinterpret_cast<const pph_d2*>(p);
  a = v.x; b = v.y;
}
#endif

int sell_alloc(pph_ctx* ctx, const MeshData& mesh, DevBuf<double>& buf, Sell* out, int sym);
// operator bytes a product streams per row: the stored values, or the 2-byte class with a usable row dictionary - of the rows
// of four planes only where the walk kernel takes the classes of the interior planes from the plane below (SellDict::zconst)
static inline double sell_stream_bytes(const pph_ctx* ctx, const Sell& E) {
  const bool dict = ctx->sell_rpt != 1 && E.sym && E.dict && E.dict->on;
  if (!dict) return 8.0 * sell_stored(E.kind, E.sym);
  const bool zc = E.dict->zconst && ctx->sell_dict_zconst && ctx->sell_dict_walk && E.kind == PPH_CELL_HEX && E.pz >= 8;
  return zc ? 2.0 * 4.0 / (double)E.pz : 2.0;
}
// (re)builds the row dictionary of E after its values were (re)written; sets / clears E->dict
int sell_dict_update(pph_ctx* ctx, Sell* E, SellDict& D, int64_t n);
int sell_dict_poll(pph_ctx* ctx);
#ifdef PPH_DW_STAMPS
int sell_dw_stamps(pph_ctx* ctx, int on);   // diagnostic build (tools/r4_dict_stamps.py)
#endif
// fused check (pph_sell.hip): (re)build the group after its dictionaries were built; before an assembly: fill the tables from
// the mini operator and check the class adjacencies (after the listed launch); `dicts`: up to three, null entries allowed
int dict_group_build(pph_ctx* ctx, DictGroup& G, SellDict* const* dicts, int nd, const Sell& shape, int64_t n);
int dict_group_tables(pph_ctx* ctx, DictGroup& G, SellDict* const* dicts, int nd, const Sell& shape);   // retires dictionaries a per-assembly check refused on the device (end of a solve)
int sell_from_csr(pph_ctx* ctx, const MeshData& mesh, const double* csr_val, DevBuf<double>& buf, Sell* out, int sym);
// symmetric storage is used for operators that are symmetric on the local box: single context (a slab's ghost rows
// are empty, which breaks the symmetry of the local matrix) and option sell_sym on
static inline int pph_sell_sym(const pph_ctx* ctx);
static inline int pph_sell_sym_from_csr(const pph_ctx* ctx);
int sell_to_csr(pph_ctx* ctx, const MeshData& mesh, const Sell& E, double* csr_val);

// block values + Dirichlet elimination on any level: out = (row constrained) ? I : coefK*K + coefM*M with
// constrained columns zeroed
void pph_launch_scalar_block(pph_ctx* ctx, const MeshData& mesh, const uint8_t* mask, double coefK, double coefM,
                             double* out);
// Jacobi-preconditioned CG on caller-supplied work vectors (used by the multigrid coarsest level)
int pph_cg_jacobi(pph_ctx* ctx, const Csr& A, const double* b, double* x, const double* dinv, double rtol, double atol,
                  int max_it, double* r, double* z, double* p, double* q, int* its);
// multigrid (pph_mg.hip)
int mg_setup(pph_ctx* ctx);
int mg_lam_host(pph_ctx* ctx);
int mg_transfer_bench(pph_ctx* ctx, int which, int reps, double* out2);   // diagnostic: ms per launch of the fine-level interpolation / restriction   // host copies of the levels' spectral bounds (waits for the read-back if it is still on its way)
// Jacobi-CG of a stencil-ELL operator of at most 4096 rows inside ONE workgroup (the coarsest multigrid level's kernel;
// also the reference's LU blocks on plumbing-size meshes): x = A^-1 b to rtol, zero guess; r, p, q: work vectors of n
void mg_onchip_cg(pph_ctx* ctx, const Sell& E, const double* dinv, const double* b, double* x, double* r, double* p, double* q,
                  int64_t n, double rtol, int max_it);
void mg_release(pph_ctx* ctx);
// z = Vcycle(r) for block `which` (0: A11, 1: A22); r and z have fine-level length n
void mg_vcycle(pph_ctx* ctx, int which, const double* r, double* z, int nsmooth);
bool mg_pre_smoother(pph_ctx* ctx, int which, int nsmooth, const double** dinv, const double** w /* device */,
                     bool* launch_only);

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
// symmetric storage (diagonal + upper slots); on slabs (sell_sym_slabs) the ghost rows keep their entries towards owned
// columns so that owned rows find their lower entries (fuse_elim_diag, pph_assemble.hip)
static inline int pph_sell_sym(const pph_ctx* ctx) { return (ctx->sell_sym && (ctx->world == 1 || ctx->sell_sym_slabs)) ? 1 : 0; }
// ... for operators converted from CSR values, whose ghost rows are empty: single context only
static inline int pph_sell_sym_from_csr(const pph_ctx* ctx) { return (ctx->sell_sym && ctx->world == 1) ? 1 : 0; }

// owned index set of a vector of `nrows` entries living on slab geometry g (null: everything)
static inline Seg pph_owned_seg(const MeshData* g, int64_t nrows) {
  Seg s;
  if (!g) { s.off1 = 0; s.len1 = nrows; s.off2 = 0; s.len2 = 0; return s; }
  s.off1 = g->own_begin(); s.len1 = g->own_end() - g->own_begin();
  if (nrows == 2 * g->n) { s.off2 = g->n + s.off1; s.len2 = s.len1; } else { s.off2 = 0; s.len2 = 0; }
  return s;
}

#define PPH_MAX_SCAL 4096  // reduction slots in ctx->scal
#define PPH_PART_STRIDE 4096  // partial sums per reduction slot behind them (>= the largest grid that writes partial sums)
