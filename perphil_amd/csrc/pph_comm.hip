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
#include <cstdio>
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

// 0 when librccl can be loaded and has every entry point this transport uses (checked by each rank BEFORE any rank
// enters ncclCommInitRank, so that one rank without the library cannot leave the others waiting inside it)
extern "C" int pph_rccl_available(const char* libpath) { return rccl_load(nullptr, libpath); }

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
  ctx->comm_status = PPH_OK;
  ctx->asm_ok = false; ctx->ell_ok = false; ctx->csr_ok = false; ctx->mono_ok = false;   // assembled for another decomposition
  mg_release(ctx);
  return PPH_OK;
}

void comm_release(pph_ctx* ctx) {
  if (ctx->nccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->nccl_comm);
  ctx->nccl_comm = nullptr;
}

// ---- transport-independent primitives used by the solver ---------------------------------------------

// Communication failures are sticky: once an exchange or reduction failed on this rank the stale ghost planes /
// partial sums make every later result meaningless and the peers may already wait in a collective this rank will
// never join.  comm_fail records the first failure in ctx->comm_status; la_fetch / la_fetch_raw (every Krylov
// iteration passes through one of them) and the end of pph_solve_device turn it into PPH_ERR_COMM.
static int comm_fail(pph_ctx* ctx, const char* what, const char* detail) {
  if (ctx->comm_status == PPH_OK) {
    ctx->comm_status = PPH_ERR_COMM;
    pph_set_error(ctx, "%s failed on rank %d: %s", what, ctx->rank, detail ? detail : "?");
    ctx->comm_error = ctx->err;
  }
  return PPH_ERR_COMM;
}

static const char* rccl_errstr(int r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"; }

// option "time_comm": an event pair around every exchange / reduction on the stream it is issued on (harvested with
// the SpMV pairs by la_harvest_spmv_times: variant 2 = halo exchange, 3 = all-reduce); the callback transport is
// timed with the host clock around the callback (its stream is idle then)
#include <chrono>
static pph_ctx::EvPair* comm_ev_begin(pph_ctx* ctx, int variant, hipStream_t on) {
  if (!ctx->time_comm) return nullptr;
  if (ctx->ev_used == ctx->ev_pool.size()) {
    pph_ctx::EvPair p;
    p.variant = 0; p.fine = false;
    if (hipEventCreate(&p.e0) == hipSuccess && hipEventCreate(&p.e1) == hipSuccess) ctx->ev_pool.push_back(p);
  }
  if (ctx->ev_used >= ctx->ev_pool.size()) return nullptr;
  pph_ctx::EvPair* ev = &ctx->ev_pool[ctx->ev_used++];
  ev->variant = variant; ev->fine = false;
  (void)hipEventRecord(ev->e0, on);
  return ev;
}
static void comm_ev_end(pph_ctx::EvPair* ev, hipStream_t on) { if (ev) (void)hipEventRecord(ev->e1, on); }
struct HostClock {
  pph_ctx* ctx; int which; std::chrono::steady_clock::time_point t0;
  HostClock(pph_ctx* c, int w) : ctx(c), which(w), t0(std::chrono::steady_clock::now()) {}
  ~HostClock() {
    if (!ctx->time_comm) return;
    ctx->t_comm[which] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->n_comm_timed[which]++;
  }
};

// `on` (default: the context's stream): stream the RCCL exchange is issued on.  `x_ready` (callback transport): the
// event after which v holds the planes to send - waited for instead of the whole stream, which may already carry
// the interior rows of the product that consumes v (halo_overlap, pph_la.hip)
int la_halo(pph_ctx* ctx, const MeshData& g, double* v, hipStream_t on, hipEvent_t x_ready) {
  if (ctx->world <= 1 || (!g.glo && !g.ghi)) return PPH_OK;
  if (!on) on = ctx->stream;
  if (ctx->comm_status != PPH_OK) return ctx->comm_status;
  const int64_t pl = g.plane();
  const int64_t send_lo = g.glo ? pl : -1, recv_lo = g.glo ? 0 : -1;
  const int64_t send_hi = g.ghi ? g.n - 2 * pl : -1, recv_hi = g.ghi ? g.n - pl : -1;
  if (ctx->nccl_comm) {
    pph_ctx::EvPair* ev = comm_ev_begin(ctx, 2, on);
    // every call of the group is issued and the group is always closed, also after a failed call
    int bad = g_rccl.GroupStart();
    if (g.glo) {
      int r = g_rccl.Send(v + send_lo, (size_t)pl, RCCL_DOUBLE, ctx->rank - 1, ctx->nccl_comm, on);
      bad = bad ? bad : r;
      r = g_rccl.Recv(v + recv_lo, (size_t)pl, RCCL_DOUBLE, ctx->rank - 1, ctx->nccl_comm, on);
      bad = bad ? bad : r;
    }
    if (g.ghi) {
      int r = g_rccl.Send(v + send_hi, (size_t)pl, RCCL_DOUBLE, ctx->rank + 1, ctx->nccl_comm, on);
      bad = bad ? bad : r;
      r = g_rccl.Recv(v + recv_hi, (size_t)pl, RCCL_DOUBLE, ctx->rank + 1, ctx->nccl_comm, on);
      bad = bad ? bad : r;
    }
    const int re = g_rccl.GroupEnd();
    bad = bad ? bad : re;
    comm_ev_end(ev, on);
    if (bad) return comm_fail(ctx, "RCCL halo exchange", rccl_errstr(bad));
    ctx->n_halo++;
    return PPH_OK;
  }
  if (!ctx->halo_cb) return PPH_OK;
  if ((x_ready ? hipEventSynchronize(x_ready) : hipStreamSynchronize(ctx->stream)) != hipSuccess)
    return comm_fail(ctx, "halo exchange", "synchronisation before the callback");
  HostClock hc(ctx, 0);
  if (ctx->halo_cb(ctx->comm_user, v, pl, send_lo, recv_lo, send_hi, recv_hi) != 0)
    return comm_fail(ctx, "halo exchange", "callback returned an error");
  ctx->n_halo++;
  return PPH_OK;
}

// sum of `count` doubles starting at device address `dev` over all ranks, in place, on the context stream
int comm_allreduce_device(pph_ctx* ctx, double* dev, int64_t count) {
  if (ctx->world <= 1) return PPH_OK;
  if (ctx->comm_status != PPH_OK) return ctx->comm_status;
  ctx->n_allreduce++;
  if (ctx->nccl_comm) {
    pph_ctx::EvPair* ev = comm_ev_begin(ctx, 3, ctx->stream);
    const int r = g_rccl.AllReduce(dev, dev, (size_t)count, RCCL_DOUBLE, RCCL_SUM, ctx->nccl_comm, ctx->stream);
    comm_ev_end(ev, ctx->stream);
    if (r) return comm_fail(ctx, "RCCL all-reduce", rccl_errstr(r));
    return PPH_OK;
  }
  if (!ctx->allreduce_cb) return PPH_OK;
  ctx->h_stage.resize((size_t)count);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return comm_fail(ctx, "all-reduce", "synchronisation before the callback");
  HostClock hc(ctx, 1);
  if (hipMemcpyAsync(ctx->h_stage.data(), dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return comm_fail(ctx, "all-reduce", "device-to-host copy");
  if (ctx->allreduce_cb(ctx->comm_user, ctx->h_stage.data(), count) != 0)
    return comm_fail(ctx, "all-reduce", "callback returned an error");
  if (hipMemcpyAsync(dev, ctx->h_stage.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return comm_fail(ctx, "all-reduce", "host-to-device copy");
  return PPH_OK;
}

int la_allreduce_vec(pph_ctx* ctx, double* v, int64_t n) { return comm_allreduce_device(ctx, v, n); }

// sum of `count` HOST doubles over all ranks (setup-time collectives)
int comm_allreduce_host(pph_ctx* ctx, double* vals, int64_t count) {
  if (ctx->world <= 1) return PPH_OK;
  if (ctx->comm_status != PPH_OK) return ctx->comm_status;
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
  ctx->n_allreduce++;
  HostClock hc(ctx, 1);
  if (ctx->allreduce_cb(ctx->comm_user, vals, count) != 0) return comm_fail(ctx, "all-reduce", "callback returned an error");
  return PPH_OK;
}

// Self-test of the RCCL plumbing on the context's stream.  STRAIGHT-LINE: every rank always issues all three phases -
// a grouped send/recv to itself, one grouped exchange with both slab neighbours (the halo pattern), an all-reduce - so
// that no rank can leave its peers waiting in a collective it skipped; failures of single calls and wrong data are
// only RECORDED on the way, a group that was opened is always closed, the buffers are released on every path, and
// the verdict (PPH_OK or PPH_ERR_COMM with the first failure as message) is returned after the last phase.  The
// launcher then agrees on the verdict across ranks (perphil_amd/distributed.py).  `ranks_seen` (optional) receives
// the world size the all-reduce observed (sum of 1 over all ranks).
extern "C" int pph_comm_selftest2(pph_ctx* ctx, int* ranks_seen) {
  if (!ctx) return PPH_ERR_INVALID;
  if (ranks_seen) *ranks_seen = 0;
  PPH_REQUIRE(ctx, ctx->nccl_comm != nullptr, "no RCCL communicator");
  PPH_HIP(ctx, hipSetDevice(ctx->device));
  // buffers: the reduction-partials area of the context (idle outside a solve) - no allocation, so every rank
  // always sends and receives messages of the same length
  const size_t cnt = 4096;
  double* pa = ctx->scal.p + PPH_MAX_SCAL;
  double* pb = pa + cnt;
  double* plo = pb + cnt;
  double* phi = plo + cnt;
  double* dv = phi + cnt;
  bool ok = true;
  char first[256] = "";
  auto note = [&](const char* what, const char* detail) {
    if (ok) snprintf(first, sizeof(first), "%s: %s", what, detail ? detail : "?");
    ok = false;
  };
  auto rc = [&](int r, const char* what) { if (r != 0) note(what, rccl_errstr(r)); };
  auto hc = [&](hipError_t e, const char* what) { if (e != hipSuccess) note(what, hipGetErrorString(e)); };
  std::vector<double> h(cnt), g(cnt, -1.0);
  for (size_t i = 0; i < cnt; ++i) h[i] = 1000.0 * ctx->rank + (double)i;
  hc(hipMemcpyAsync(pa, h.data(), sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream), "upload");
  hc(hipMemcpyAsync(pb, g.data(), sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream), "upload");
  // phase 1: to self
  rc(g_rccl.GroupStart(), "ncclGroupStart");
  rc(g_rccl.Send(pa, cnt, RCCL_DOUBLE, ctx->rank, ctx->nccl_comm, ctx->stream), "ncclSend(self)");
  rc(g_rccl.Recv(pb, cnt, RCCL_DOUBLE, ctx->rank, ctx->nccl_comm, ctx->stream), "ncclRecv(self)");
  rc(g_rccl.GroupEnd(), "ncclGroupEnd");
  // phase 2: both slab neighbours
  const bool has_lo = ctx->rank > 0, has_hi = ctx->rank + 1 < ctx->world;
  if (ctx->world > 1) {
    rc(g_rccl.GroupStart(), "ncclGroupStart");
    if (has_lo) {
      rc(g_rccl.Send(pa, cnt, RCCL_DOUBLE, ctx->rank - 1, ctx->nccl_comm, ctx->stream), "ncclSend(lower)");
      rc(g_rccl.Recv(plo, cnt, RCCL_DOUBLE, ctx->rank - 1, ctx->nccl_comm, ctx->stream), "ncclRecv(lower)");
    }
    if (has_hi) {
      rc(g_rccl.Send(pa, cnt, RCCL_DOUBLE, ctx->rank + 1, ctx->nccl_comm, ctx->stream), "ncclSend(upper)");
      rc(g_rccl.Recv(phi, cnt, RCCL_DOUBLE, ctx->rank + 1, ctx->nccl_comm, ctx->stream), "ncclRecv(upper)");
    }
    rc(g_rccl.GroupEnd(), "ncclGroupEnd");
  }
  // phase 3: all-reduce of {1, rank + 1, 0.5} (always issued)
  double v[3] = {1.0, (double)(ctx->rank + 1), 0.5};
  hc(hipMemcpyAsync(dv, v, sizeof(v), hipMemcpyHostToDevice, ctx->stream), "upload");
  rc(g_rccl.AllReduce(dv, dv, 3, RCCL_DOUBLE, RCCL_SUM, ctx->nccl_comm, ctx->stream), "ncclAllReduce");
  hc(hipMemcpyAsync(v, dv, sizeof(v), hipMemcpyDeviceToHost, ctx->stream), "download");
  // verification, after all phases
  hc(hipMemcpyAsync(g.data(), pb, sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream), "download");
  hc(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
  for (size_t i = 0; i < cnt && ok; ++i)
    if (g[i] != h[i]) note("self send/recv", "wrong data");
  for (int side = 0; side < 2 && ctx->world > 1; ++side) {
    if (!(side == 0 ? has_lo : has_hi)) continue;
    const int peer = side == 0 ? ctx->rank - 1 : ctx->rank + 1;
    hc(hipMemcpyAsync(g.data(), side == 0 ? plo : phi, sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream), "download");
    hc(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    for (size_t i = 0; i < cnt && ok; ++i)
      if (g[i] != 1000.0 * peer + (double)i) note("neighbour exchange", "wrong data");
  }
  const double w = (double)ctx->world;
  if (!(v[0] == w && v[1] == w * (w + 1) / 2 && v[2] == 0.5 * w)) note("all-reduce", "wrong sum");
  if (ranks_seen) *ranks_seen = (int)v[0];
  if (!ok) {
    pph_set_error(ctx, "RCCL self-test failed on rank %d of %d (%s)", ctx->rank, ctx->world, first);
    return PPH_ERR_COMM;
  }
  return PPH_OK;
}

extern "C" int pph_comm_selftest(pph_ctx* ctx) { return pph_comm_selftest2(ctx, nullptr); }

// communication counters of the last solve (bench.py: config.halo_exchanges_per_step / allreduces_per_step) and the
// sticky status
// option "time_comm": summed durations of the last solve's exchanges and reductions (ms) and how many were timed;
// out[0] halo ms, [1] all-reduce ms, [2] halo exchanges timed, [3] all-reduces timed.  RCCL transport: device time between
// two events on the issuing stream (an overlapped exchange counts its full duration on the communication stream);
// callback transport: host time inside the callback (+ staging copies of the reductions)
extern "C" int pph_comm_times(pph_ctx* ctx, double* out4) {
  if (!ctx || !out4) return PPH_ERR_INVALID;
  la_harvest_spmv_times(ctx);
  out4[0] = ctx->t_comm[0]; out4[1] = ctx->t_comm[1];
  out4[2] = (double)ctx->n_comm_timed[0]; out4[3] = (double)ctx->n_comm_timed[1];
  return PPH_OK;
}

extern "C" int pph_comm_stats(pph_ctx* ctx, int64_t* halo, int64_t* allreduce, int* status) {
  if (!ctx) return PPH_ERR_INVALID;
  if (halo) *halo = ctx->n_halo;
  if (allreduce) *allreduce = ctx->n_allreduce;
  if (status) *status = ctx->comm_status;
  return PPH_OK;
}
