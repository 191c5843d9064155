// comm.hip -- RCCL-over-xGMI communicators for the 2-D cyclic grid.
//
// Replaces the MPI layer of the reference: comm_mod wrappers bcast_dbl / reduce_dbl (= allreduce) /
// allgather_dbl / datacast_dbl (src/comm.F:726-1528) and the hand-rolled reproducible allreduces
// (src/comm.F:2035-2580).  Communicator structure follows eigen_init_cartesian_check
// (src/eigen_libs0.F:579-585): "X" = ranks that share my column coordinate py (size Px),
// "Y" = ranks that share my row coordinate px (size Py), plus world.
//
// librccl is dlopen'ed on first multi-rank init so the single-GPU path carries no RCCL dependency.
// RCCL reductions over a fixed communicator use a fixed ring/tree order, so every rank receives
// bit-identical sums -- the property the reference's hand allreduce exists for (manual 5.5.1).
#include "eigx_context.h"
#include "eigx_comm.h"
#include "../../include/eigenexa_amd.h"
#include <dlfcn.h>
#include <cstring>
#include <vector>

namespace eigx {

namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, ncclUniqueIdBlob, int) = nullptr;
  int (*CommSplit)(void*, int, int, void**, void*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
} api;

bool load_rccl() {
  if (api.lib) return true;
  api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!api.lib) api.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!api.lib) api.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!api.lib) {
    fprintf(stderr, "[eigx] cannot load librccl: %s\n", dlerror());
    return false;
  }
#define EIGX_SYM(field, name)                                                    \
  *(void**)(&api.field) = dlsym(api.lib, name);                                  \
  if (!api.field) { fprintf(stderr, "[eigx] librccl lacks %s\n", name); return false; }
  EIGX_SYM(GetUniqueId, "ncclGetUniqueId");
  EIGX_SYM(CommInitRank, "ncclCommInitRank");
  EIGX_SYM(CommSplit, "ncclCommSplit");
  EIGX_SYM(CommDestroy, "ncclCommDestroy");
  EIGX_SYM(AllReduce, "ncclAllReduce");
  EIGX_SYM(Broadcast, "ncclBroadcast");
  EIGX_SYM(AllGather, "ncclAllGather");
  EIGX_SYM(GroupStart, "ncclGroupStart");
  EIGX_SYM(GroupEnd, "ncclGroupEnd");
  EIGX_SYM(Send, "ncclSend");
  EIGX_SYM(Recv, "ncclRecv");
  EIGX_SYM(GetErrorString, "ncclGetErrorString");
#undef EIGX_SYM
  return true;
}

#define EIGX_NCCL_CHECK(expr)                                                              \
  do {                                                                                     \
    int _r = (expr);                                                                       \
    if (_r != 0) {                                                                         \
      fprintf(stderr, "[eigx] RCCL error %d (%s) at %s:%d\n", _r,                          \
              api.GetErrorString ? api.GetErrorString(_r) : "?", __FILE__, __LINE__);     \
      abort();                                                                             \
    }                                                                                      \
  } while (0)

constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0, kNcclMax = 2;

// Host-staged test transport: the collectives are delegated to callbacks (tests register gloo-backed
// Python functions), so the distributed algorithm can be exercised with several processes on ONE GPU,
// where RCCL refuses duplicate devices.  Never used by bench.py.
struct Callbacks {
  eigx_allreduce_cb allreduce = nullptr;  // (buf, count, op: 0 sum / 2 max, group)
  eigx_bcast_cb bcast = nullptr;          // (buf, count, root, group)
  eigx_allgather_cb allgather = nullptr;  // (send, recv, count, group)
} cbs;
std::vector<double> stage_a, stage_b;
}  // namespace

bool comm_uses_callbacks(const Context& ctx) { return ctx.comm && ctx.comm->callbacks; }

int comm_get_unique_id(void* out128) {
  if (!out128) return EIGX_ERR_BAD_ARG;
  if (!load_rccl()) return EIGX_ERR_INTERNAL;
  const int rc_id = api.GetUniqueId(out128);
  if (rc_id != 0) { fprintf(stderr, "[eigx] ncclGetUniqueId failed: %d\n", rc_id); return EIGX_ERR_INTERNAL; }
  return EIGX_OK;
}

int comm_init(Context& ctx, const void* uid) {
  if (!uid) {
    if (!cbs.allreduce || !cbs.bcast || !cbs.allgather) return EIGX_ERR_BAD_ARG;
    CommState* cs = new CommState();
    cs->callbacks = true;
    ctx.comm = cs;
    return EIGX_OK;
  }
  if (!load_rccl()) return EIGX_ERR_INTERNAL;
  CommState* cs = new CommState();
  ncclUniqueIdBlob id;
  memcpy(id.internal, uid, 128);
  const Grid& g = ctx.grid;
  // a failed communicator is reported, not fatal: bench.py then falls back to independent replicas
  const int rc_init = api.CommInitRank(&cs->world, g.nranks, id, g.rank);
  if (rc_init != 0) {
    fprintf(stderr, "[eigx] ncclCommInitRank failed: %d (%s)\n", rc_init, api.GetErrorString ? api.GetErrorString(rc_init) : "?");
    delete cs;
    return EIGX_ERR_INTERNAL;
  }
  // round 1 uses the world communicator only; X / Y groups are split on demand
  ctx.comm = cs;
  return EIGX_OK;
}

void comm_free(Context& ctx) {
  if (!ctx.comm) return;
  if (ctx.comm->callbacks) { delete ctx.comm; ctx.comm = nullptr; return; }
  if (ctx.comm->x) api.CommDestroy(ctx.comm->x);
  if (ctx.comm->y) api.CommDestroy(ctx.comm->y);
  if (ctx.comm->world) api.CommDestroy(ctx.comm->world);
  delete ctx.comm;
  ctx.comm = nullptr;
}

static void* pick(const Context& ctx, CommGroup grp) {
  return grp == COMM_X ? ctx.comm->x : grp == COMM_Y ? ctx.comm->y : ctx.comm->world;
}
int comm_size(const Context& ctx, CommGroup grp) {
  return grp == COMM_X ? ctx.grid.Px : grp == COMM_Y ? ctx.grid.Py : ctx.grid.nranks;
}

static void cb_roundtrip_begin(double* dev, size_t count, std::vector<double>& h, hipStream_t s) {
  h.resize(count);
  EIGX_HIP_CHECK(hipMemcpyAsync(h.data(), dev, count * 8, hipMemcpyDeviceToHost, s));
  EIGX_HIP_CHECK(hipStreamSynchronize(s));
}
static void cb_roundtrip_end(double* dev, size_t count, std::vector<double>& h, hipStream_t s) {
  EIGX_HIP_CHECK(hipMemcpyAsync(dev, h.data(), count * 8, hipMemcpyHostToDevice, s));
  EIGX_HIP_CHECK(hipStreamSynchronize(s));
}

void comm_allreduce_sum(const Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s) {
  if (comm_size(ctx, grp) == 1 || count == 0) return;
  if (ctx.comm->callbacks) {
    cb_roundtrip_begin(buf, count, stage_a, s);
    cbs.allreduce(stage_a.data(), (long)count, 0, (int)grp);
    cb_roundtrip_end(buf, count, stage_a, s);
    return;
  }
  EIGX_NCCL_CHECK(api.AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, pick(ctx, grp), s));
}
void comm_allreduce_max(const Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s) {
  if (comm_size(ctx, grp) == 1 || count == 0) return;
  if (ctx.comm->callbacks) {
    cb_roundtrip_begin(buf, count, stage_a, s);
    cbs.allreduce(stage_a.data(), (long)count, 2, (int)grp);
    cb_roundtrip_end(buf, count, stage_a, s);
    return;
  }
  EIGX_NCCL_CHECK(api.AllReduce(buf, buf, count, kNcclFloat64, kNcclMax, pick(ctx, grp), s));
}
void comm_bcast(const Context& ctx, CommGroup grp, double* buf, size_t count, int root, hipStream_t s) {
  if (comm_size(ctx, grp) == 1 || count == 0) return;
  if (ctx.comm->callbacks) {
    cb_roundtrip_begin(buf, count, stage_a, s);
    cbs.bcast(stage_a.data(), (long)count, root, (int)grp);
    cb_roundtrip_end(buf, count, stage_a, s);
    return;
  }
  EIGX_NCCL_CHECK(api.Broadcast(buf, buf, count, kNcclFloat64, root, pick(ctx, grp), s));
}
void comm_allgather(const Context& ctx, CommGroup grp, const double* send, double* recv, size_t count,
                    hipStream_t s) {
  if (comm_size(ctx, grp) == 1) {
    if (send != recv && count)
      EIGX_HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
    return;
  }
  if (ctx.comm->callbacks) {
    const size_t np = (size_t)comm_size(ctx, grp);
    cb_roundtrip_begin(const_cast<double*>(send), count, stage_a, s);
    stage_b.resize(count * np);
    cbs.allgather(stage_a.data(), stage_b.data(), (long)count, (int)grp);
    cb_roundtrip_end(recv, count * np, stage_b, s);
    return;
  }
  EIGX_NCCL_CHECK(api.AllGather(send, recv, count, kNcclFloat64, pick(ctx, grp), s));
}

}  // namespace eigx

// 1-rank RCCL self-test: exercises dlopen, ncclCommInitRank from a unique id and the three collectives the
// multi-GPU solvers use, on the library's compute stream (a one-GPU box cannot host more RCCL ranks)
extern "C" int eigx_rccl_selftest(void) {
  using namespace eigx;
  if (!g_ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (!load_rccl()) return EIGX_ERR_INTERNAL;
  EIGX_HIP_CHECK(hipSetDevice(g_ctx.device));
  ncclUniqueIdBlob id;
  EIGX_NCCL_CHECK(api.GetUniqueId(&id));
  void* comm = nullptr;
  EIGX_NCCL_CHECK(api.CommInitRank(&comm, 1, id, 0));
  const int cnt = 1000;
  double *a = nullptr, *b = nullptr;
  EIGX_HIP_CHECK(hipMalloc(&a, cnt * 8));
  EIGX_HIP_CHECK(hipMalloc(&b, cnt * 8));
  std::vector<double> h(cnt), r(cnt);
  for (int i = 0; i < cnt; ++i) h[i] = 0.5 * i - 3.0;
  EIGX_HIP_CHECK(hipMemcpy(a, h.data(), cnt * 8, hipMemcpyHostToDevice));
  hipStream_t s = g_ctx.stream;
  EIGX_NCCL_CHECK(api.AllReduce(a, a, cnt, kNcclFloat64, kNcclSum, comm, s));
  EIGX_NCCL_CHECK(api.Broadcast(a, a, cnt, kNcclFloat64, 0, comm, s));
  EIGX_NCCL_CHECK(api.AllGather(a, b, cnt, kNcclFloat64, comm, s));
  EIGX_NCCL_CHECK(api.AllGather(b, b, cnt, kNcclFloat64, comm, s));  // in-place form used for Z
  EIGX_HIP_CHECK(hipStreamSynchronize(s));
  EIGX_HIP_CHECK(hipMemcpy(r.data(), b, cnt * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < cnt; ++i) bad += (r[i] != h[i]);
  EIGX_NCCL_CHECK(api.CommDestroy(comm));
  EIGX_HIP_CHECK(hipFree(a));
  EIGX_HIP_CHECK(hipFree(b));
  return bad == 0 ? EIGX_OK : EIGX_ERR_INTERNAL;
}

extern "C" int eigx_set_comm_callbacks(eigx_allreduce_cb ar, eigx_bcast_cb bc, eigx_allgather_cb ag) {
  eigx::cbs.allreduce = ar;
  eigx::cbs.bcast = bc;
  eigx::cbs.allgather = ag;
  return 0;
}

namespace eigx {

}  // namespace eigx
