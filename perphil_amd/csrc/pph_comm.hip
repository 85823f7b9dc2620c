// RCCL transport of the slab decomposition: neighbour-plane send/recv and scalar all-reduces issued on the
// context's own HIP stream (no host synchronisation per exchange).  librccl is resolved at run time with
// dlopen so that single-GPU use needs no RCCL at all and so that, inside a PyTorch process, the copy
// torch already loaded (same SONAME) is the one used.  The communicator is created from a unique id that
// the Python launcher broadcasts over torch.distributed (perphil_amd/distributed.py).
//
// Replaces PETSc's implicit VecScatter halo exchange and VecDot all-reduce under mpiexec (never exercised
// by the reference): per SpMV one (nx+1)(ny+1)-plane per neighbour and direction
// (0.53 MB at 256^3), grouped ncclSend/ncclRecv between neighbours only; all-reduce for 1-3 scalars.
#include "pph_internal.h"
#include <dlfcn.h>
#include <cstring>

typedef int (*fn_GetUniqueId)(void*);
typedef int (*fn_CommInitRank)(void**, int, PphNcclId, int);
typedef int (*fn_CommDestroy)(void*);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_Send)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_Recv)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_Group)(void);
typedef const char* (*fn_ErrStr)(int);

enum { RCCL_DOUBLE = 8, RCCL_SUM = 0 };  // ncclFloat64, ncclSum (rccl.h)

struct RcclApi {
  void* handle = nullptr;
  fn_GetUniqueId GetUniqueId = nullptr;
  fn_CommInitRank CommInitRank = nullptr;
  fn_CommDestroy CommDestroy = nullptr;
  fn_AllReduce AllReduce = nullptr;
  fn_Send Send = nullptr;
  fn_Recv Recv = nullptr;
  fn_Group GroupStart = nullptr, GroupEnd = nullptr;
  fn_ErrStr GetErrorString = nullptr;
};

static RcclApi g_rccl;

static int rccl_load(pph_ctx* ctx, const char* libpath) {
  if (g_rccl.handle) return PPH_OK;
  const char* cands[4] = {libpath && libpath[0] ? libpath : "librccl.so.1", "librccl.so.1", "/opt/rocm/lib/librccl.so.1",
                          "librccl.so"};
  void* h = nullptr;
  for (int i = 0; i < 4 && !h; ++i) h = dlopen(cands[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    pph_set_error(ctx, "cannot load librccl: %s", dlerror());
    return PPH_ERR_COMM;
  }
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (fn_GetUniqueId)dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (fn_CommInitRank)dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (fn_CommDestroy)dlsym(h, "ncclCommDestroy");
  a.AllReduce = (fn_AllReduce)dlsym(h, "ncclAllReduce");
  a.Send = (fn_Send)dlsym(h, "ncclSend");
  a.Recv = (fn_Recv)dlsym(h, "ncclRecv");
  a.GroupStart = (fn_Group)dlsym(h, "ncclGroupStart");
  a.GroupEnd = (fn_Group)dlsym(h, "ncclGroupEnd");
  a.GetErrorString = (fn_ErrStr)dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.Send || !a.Recv || !a.GroupStart ||
      !a.GroupEnd) {
    pph_set_error(ctx, "librccl lacks a required symbol");
    return PPH_ERR_COMM;
  }
  g_rccl = a;
  return PPH_OK;
}

#define RCCL_TRY(ctx, call)                                                                              \
  do {                                                                                                   \
    int r__ = (call);                                                                                    \
    if (r__ != 0) {                                                                                      \
      pph_set_error((ctx), "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "?"); \
      return PPH_ERR_COMM;                                                                               \
    }                                                                                                    \
  } while (0)

extern "C" int pph_rccl_unique_id(const char* libpath, uint8_t* id128) {
  if (!id128) return PPH_ERR_INVALID;
  PPH_TRY(rccl_load(nullptr, libpath));
  PphNcclId id;
  memset(&id, 0, sizeof(id));
  RCCL_TRY(nullptr, g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, 128);
  return PPH_OK;
}

extern "C" int pph_comm_init_rccl(pph_ctx* ctx, int rank, int world, const uint8_t* id128, const char* libpath) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world && id128, "bad rank/world/id");
  PPH_TRY(rccl_load(ctx, libpath));
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->nccl_comm) { (void)g_rccl.CommDestroy(ctx->nccl_comm); ctx->nccl_comm = nullptr; }
  PphNcclId id;
  memcpy(id.internal, id128, 128);
  RCCL_TRY(ctx, g_rccl.CommInitRank(&ctx->nccl_comm, world, id, rank));
  ctx->rank = rank;
  ctx->world = world;
  ctx->halo_cb = nullptr;
  ctx->allreduce_cb = nullptr;
  mg_release(ctx);
  return PPH_OK;
}

void comm_release(pph_ctx* ctx) {
  if (ctx->nccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->nccl_comm);
  ctx->nccl_comm = nullptr;
}

// ---- transport-independent primitives used by the solver ---------------------------------------------

int la_halo(pph_ctx* ctx, const MeshData& g, double* v) {
  if (ctx->world <= 1 || (!g.glo && !g.ghi)) return PPH_OK;
  const int64_t pl = g.plane();
  const int64_t send_lo = g.glo ? pl : -1, recv_lo = g.glo ? 0 : -1;
  const int64_t send_hi = g.ghi ? g.n - 2 * pl : -1, recv_hi = g.ghi ? g.n - pl : -1;
  if (ctx->nccl_comm) {
    RCCL_TRY(ctx, g_rccl.GroupStart());
    if (g.glo) {
      RCCL_TRY(ctx, g_rccl.Send(v + send_lo, (size_t)pl, RCCL_DOUBLE, ctx->rank - 1, ctx->nccl_comm, ctx->stream));
      RCCL_TRY(ctx, g_rccl.Recv(v + recv_lo, (size_t)pl, RCCL_DOUBLE, ctx->rank - 1, ctx->nccl_comm, ctx->stream));
    }
    if (g.ghi) {
      RCCL_TRY(ctx, g_rccl.Send(v + send_hi, (size_t)pl, RCCL_DOUBLE, ctx->rank + 1, ctx->nccl_comm, ctx->stream));
      RCCL_TRY(ctx, g_rccl.Recv(v + recv_hi, (size_t)pl, RCCL_DOUBLE, ctx->rank + 1, ctx->nccl_comm, ctx->stream));
    }
    RCCL_TRY(ctx, g_rccl.GroupEnd());
    ctx->n_halo++;
    return PPH_OK;
  }
  if (!ctx->halo_cb) return PPH_OK;
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->halo_cb(ctx->comm_user, v, pl, send_lo, recv_lo, send_hi, recv_hi) != 0) {
    pph_set_error(ctx, "halo-exchange callback failed");
    return PPH_ERR_COMM;
  }
  ctx->n_halo++;
  return PPH_OK;
}

// sum of `count` doubles starting at device address `dev` over all ranks, in place, on the context stream
int comm_allreduce_device(pph_ctx* ctx, double* dev, int64_t count) {
  if (ctx->world <= 1) return PPH_OK;
  if (ctx->nccl_comm) {
    RCCL_TRY(ctx, g_rccl.AllReduce(dev, dev, (size_t)count, RCCL_DOUBLE, RCCL_SUM, ctx->nccl_comm, ctx->stream));
    return PPH_OK;
  }
  if (!ctx->allreduce_cb) return PPH_OK;
  ctx->h_stage.resize((size_t)count);
  PPH_HIP(ctx, hipMemcpyAsync(ctx->h_stage.data(), dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->allreduce_cb(ctx->comm_user, ctx->h_stage.data(), count) != 0) {
    pph_set_error(ctx, "all-reduce callback failed");
    return PPH_ERR_COMM;
  }
  PPH_HIP(ctx, hipMemcpyAsync(dev, ctx->h_stage.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PPH_OK;
}

int la_allreduce_vec(pph_ctx* ctx, double* v, int64_t n) { return comm_allreduce_device(ctx, v, n); }

// sum of `count` HOST doubles over all ranks (setup-time collectives)
int comm_allreduce_host(pph_ctx* ctx, double* vals, int64_t count) {
  if (ctx->world <= 1) return PPH_OK;
  if (ctx->nccl_comm) {
    PPH_REQUIRE(ctx, count <= 64, "host all-reduce limited to 64 values");
    double* dev = ctx->scal.p + (PPH_MAX_SCAL - 192);
    PPH_HIP(ctx, hipMemcpyAsync(dev, vals, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    PPH_TRY(comm_allreduce_device(ctx, dev, count));
    PPH_HIP(ctx, hipMemcpyAsync(vals, dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PPH_OK;
  }
  if (!ctx->allreduce_cb) return PPH_OK;
  if (ctx->allreduce_cb(ctx->comm_user, vals, count) != 0) {
    pph_set_error(ctx, "all-reduce callback failed");
    return PPH_ERR_COMM;
  }
  return PPH_OK;
}

// Self-test of the RCCL plumbing on the context's stream: a grouped send/recv of one buffer to this
// rank itself and an in-place all-reduce; returns PPH_OK when the received data and the reduced sum are
// exactly what they must be (world-size-aware).  Used by the launcher before the timed region and by
// the single-GPU test (a 1-rank communicator exercises the same calls).
extern "C" int pph_comm_selftest(pph_ctx* ctx) {
  if (!ctx) return PPH_ERR_INVALID;
  PPH_REQUIRE(ctx, ctx->nccl_comm != nullptr, "no RCCL communicator");
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  const int n = 4096;
  DevBuf<double> a, b;
  PPH_TRY(a.alloc(ctx, n));
  PPH_TRY(b.alloc(ctx, n));
  std::vector<double> h((size_t)n), g((size_t)n, -1.0);
  for (int i = 0; i < n; ++i) h[(size_t)i] = 1000.0 * ctx->rank + i;
  PPH_HIP(ctx, hipMemcpyAsync(a.p, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  PPH_HIP(ctx, hipMemcpyAsync(b.p, g.data(), sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
  RCCL_TRY(ctx, g_rccl.GroupStart());
  RCCL_TRY(ctx, g_rccl.Send(a.p, (size_t)n, RCCL_DOUBLE, ctx->rank, ctx->nccl_comm, ctx->stream));
  RCCL_TRY(ctx, g_rccl.Recv(b.p, (size_t)n, RCCL_DOUBLE, ctx->rank, ctx->nccl_comm, ctx->stream));
  RCCL_TRY(ctx, g_rccl.GroupEnd());
  PPH_HIP(ctx, hipMemcpyAsync(g.data(), b.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; ++i)
    PPH_REQUIRE(ctx, g[(size_t)i] == h[(size_t)i], "RCCL self send/recv returned wrong data at %d", i);
  if (ctx->world > 1) {
    // the halo pattern itself: one grouped exchange with both slab neighbours, received data verified
    DevBuf<double> lo, hi;
    PPH_TRY(lo.alloc(ctx, n));
    PPH_TRY(hi.alloc(ctx, n));
    const bool has_lo = ctx->rank > 0, has_hi = ctx->rank + 1 < ctx->world;
    RCCL_TRY(ctx, g_rccl.GroupStart());
    if (has_lo) {
      RCCL_TRY(ctx, g_rccl.Send(a.p, (size_t)n, RCCL_DOUBLE, ctx->rank - 1, ctx->nccl_comm, ctx->stream));
      RCCL_TRY(ctx, g_rccl.Recv(lo.p, (size_t)n, RCCL_DOUBLE, ctx->rank - 1, ctx->nccl_comm, ctx->stream));
    }
    if (has_hi) {
      RCCL_TRY(ctx, g_rccl.Send(a.p, (size_t)n, RCCL_DOUBLE, ctx->rank + 1, ctx->nccl_comm, ctx->stream));
      RCCL_TRY(ctx, g_rccl.Recv(hi.p, (size_t)n, RCCL_DOUBLE, ctx->rank + 1, ctx->nccl_comm, ctx->stream));
    }
    RCCL_TRY(ctx, g_rccl.GroupEnd());
    for (int side = 0; side < 2; ++side) {
      if (!(side == 0 ? has_lo : has_hi)) continue;
      const int peer = side == 0 ? ctx->rank - 1 : ctx->rank + 1;
      PPH_HIP(ctx, hipMemcpyAsync(g.data(), side == 0 ? lo.p : hi.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
      PPH_HIP(ctx, hipStreamSynchronize(ctx->stream));
      for (int i = 0; i < n; ++i)
        PPH_REQUIRE(ctx, g[(size_t)i] == 1000.0 * peer + i, "RCCL neighbour exchange with rank %d returned wrong data at %d",
                    peer, i);
    }
    lo.release();
    hi.release();
  }
  double v[3] = {1.0, (double)(ctx->rank + 1), 0.5};
  PPH_TRY(comm_allreduce_host(ctx, v, 3));
  const double w = (double)ctx->world;
  PPH_REQUIRE(ctx, ctx->world == 1 || (v[0] == w && v[1] == w * (w + 1) / 2 && v[2] == 0.5 * w),
              "RCCL all-reduce returned %g %g %g for world %d", v[0], v[1], v[2], ctx->world);
  a.release();
  b.release();
  return PPH_OK;
}
