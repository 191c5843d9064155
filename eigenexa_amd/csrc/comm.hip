// comm.hip -- inter-GPU transport for the 2-D cyclic process grid of one xGMI node.
//
// Replaces the MPI layer of the reference: comm_mod wrappers bcast_dbl / reduce_dbl (= allreduce) /
// allgather_dbl / datacast_dbl (src/comm.F:726-1528), the hand-rolled reproducible allreduces
// (src/comm.F:2035-2580) and the X / Y communicator split of eigen_init_cartesian_check
// (src/eigen_libs0.F:579-585): "X" = ranks that share my column coordinate py (size Px),
// "Y" = ranks that share my row coordinate px (size Py), plus world.
//
// MI355X design (eigx_comm.h): the ranks of a node map each other's communication buffers (hipIpcMemHandle)
// and write into them from kernels over xGMI -- one hop, no host, no protocol -- with 8-byte epoch flags; RCCL
// (dlopen'ed, world / X / Y communicators) carries the bulk collectives when every rank owns a GPU.  Handles are
// exchanged through a POSIX shared-memory board named after the 128-byte session id that the host broadcasts
// (MPI_Bcast in the Fortran module, torch.distributed in bench.py), so eigx_init_multi needs nothing else from
// the host.  Every reduction sums the members' contributions in rank order on every rank: replicated results
// are bit-identical (the property the reference's hand allreduce exists for, manual 5.5.1).
//
// Failure model: no collective aborts.  A bounded spin that runs out, a failed mapping or an RCCL error sets a
// sticky failure flag (comm_failed); the solver then returns EIGX_ERR_INTERNAL and later waits return at once.
#include "eigx_context.h"
#include "eigx_comm.h"
#include "../../include/eigenexa_amd.h"
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstring>
#include <vector>

namespace eigx {

namespace {
struct ncclUniqueIdBlob { char internal[128]; };
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

// 0 = not tried, 1 = every entry point resolved, -1 = unusable (a failed load is final: a library that lacks one entry
// point is closed again and never half-used)
int g_rccl_state = 0;
uint64_t g_uid_hash_from_rccl = 0;   // hash of the last session id that ncclGetUniqueId produced in this process

bool load_rccl() {
  if (g_rccl_state != 0) return g_rccl_state > 0;
  g_rccl_state = -1;
  if (getenv("EIGX_NO_RCCL")) return false;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    fprintf(stderr, "[eigx] cannot load librccl: %s\n", dlerror());
    return false;
  }
  RcclApi a;
  a.lib = lib;
  bool ok = true;
#define EIGX_SYM(field, name)                                                    \
  *(void**)(&a.field) = dlsym(lib, name);                                        \
  if (!a.field) { fprintf(stderr, "[eigx] librccl lacks %s\n", name); ok = false; }
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
  if (!ok) { dlclose(lib); return false; }
  api = a;
  g_rccl_state = 1;
  return true;
}

constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0, kNcclMax = 2;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- bootstrap board: a few hundred bytes of POSIX shared memory, one slot per rank ------------------------
struct Board {
  std::atomic<uint32_t> attached;
  std::atomic<uint32_t> abort_flag;   // set by a rank whose communicator failed: every board round ends at once, everywhere
  std::atomic<uint64_t> seq_written[EIGX_MAXP];
  std::atomic<uint64_t> seq_read[EIGX_MAXP];
  unsigned char slot[EIGX_MAXP][128];
};

struct InitBlob {          // what the ranks tell each other at init
  hipIpcMemHandle_t flags_handle;   // 64 bytes
  char bus_id[32];                  // PCI bus id of the rank's GPU: equal ids = shared device
  int pid;
  int ipc_ok;
  int rccl_loaded;                  // librccl resolved completely in this process
  int uid_from_rccl;                // this process made the session id with ncclGetUniqueId (only its maker can know)
};
static_assert(sizeof(InitBlob) <= 128, "board slot too small");
struct BufBlob { hipIpcMemHandle_t handle; uint64_t bytes; int ok; };
static_assert(sizeof(BufBlob) <= 128, "board slot too small");

}  // namespace

// flag block of a rank (u64 words): [channel][kind 0 = data / parity 0, 1 = ready / parity 1][source rank]
constexpr int kFlagWords = CH_COUNT * 2 * EIGX_MAXP;
// + one more line of words behind the flags: word kFlagWords = the rank's sticky failure word.  It lives in the
// peer-mapped block so that a FAILING rank can set it on every peer (comm_fail): their bounded spins poll it and
// return at once instead of running out their time limit.
constexpr int kFlagWordsTotal = kFlagWords + 16;
__host__ __device__ inline int flag_index(int ch, int kind, int src) { return (ch * 2 + kind) * EIGX_MAXP + src; }

struct CommState {
  int P = 1, me = 0;
  Board* board = nullptr;
  uint64_t board_seq = 0;
  bool ipc = false;            // peer windows usable (mapped and self-tested)
  bool shared_device = false;  // two ranks on one GPU (tests): RCCL unusable
  bool failed = false;
  bool board_dead = false;     // a board exchange timed out: the ranks' sequence numbers no longer agree, never use it again
  double timeout_s = 120.0;
  // RCCL
  void* world = nullptr;
  void* x = nullptr;
  void* y = nullptr;
  bool rccl_ok = false;        // communicators exist on every rank and passed the self-test
  bool rccl = false;           // the BULK collectives go through RCCL (otherwise through the peer windows)
  bool step_coll = false;      // the per-step exchange is a collective allgather (RCCL / emulated), not peer writes
  bool err_in_flags = false;   // err_dev is a word of the peer-mapped flag block
  // per-collective counts since init (the reference's COMM_STAT tables, src/eigen_devel.F:364-526): calls and bytes this rank
  // SENT, by kind: 0 per-step exchange, 1 allgather / all-to-all of small pieces, 2 allreduce, 3 large all-to-all / allgather
  double st_calls[4] = {0, 0, 0, 0}, st_bytes[4] = {0, 0, 0, 0};
  bool loop = false;           // EIGX_LOOPBACK (lab): this process plays ONE rank of a P-rank grid alone; every peer window is its own
  bool in_selftest = false;    // init-time self-test: failures stay local (the verdict is voted on), device waits are short
  // init-time transport self-test (recorded for eigx_comm_info)
  int st_ipc_rounds = 0, st_ipc_errors = -1, st_step_rounds = 0, st_step_errors = -1, st_rccl_checks = 0, st_rccl_errors = -1;
  double st_ipc_us = 0.0, st_step_us = 0.0, st_rccl_us = 0.0;
  // peer-mapped flag block + local bookkeeping words
  PeerBuf flags;
  unsigned long long epoch[CH_COUNT] = {0, 0, 0, 0, 0, 0};
  unsigned* counters = nullptr;          // [CH_COUNT] last-workgroup tickets (device)
  int* err_dev = nullptr;                // sticky device-side failure word (a bounded spin ran out)
  unsigned long long* ticks_dev = nullptr;   // accumulated 100 MHz ticks spent in waits / pushes
  std::map<std::string, PeerBuf> bufs;
  std::vector<PeerBuf> retired;          // outgrown buffers: stay mapped until comm_free
  size_t step_msg[2] = {0, 0};   // doubles per message of the two step windows (Y, X)
  double rccl_seconds = 0.0;
  std::vector<hipEvent_t> tev;           // event pairs around RCCL calls
  size_t tev_used = 0;
};

namespace {

void comm_fail(CommState* cs, const char* what) {
  if (!cs->failed) fprintf(stderr, "[eigx] rank %d: communication failure: %s\n", cs->me, what);
  const bool first = !cs->failed;
  cs->failed = true;
  if (cs->board && !cs->in_selftest) cs->board->abort_flag.store(1u, std::memory_order_release);   // host side: peers waiting in a board round leave
  const int one = 1;
  if (cs->err_dev) { if (hipMemcpy(cs->err_dev, &one, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) (void)hipGetLastError(); }
  // ... and the peers: their device-side waits poll their own failure word, which sits in the peer-mapped flag block
  if (first && cs->err_in_flags && cs->flags.mapped && !cs->in_selftest) {
    for (int q = 0; q < cs->P; ++q) {
      if (q == cs->me || !cs->flags.peer[q]) continue;
      if (hipMemcpy((unsigned long long*)cs->flags.peer[q] + kFlagWords, &one, sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        (void)hipGetLastError();
    }
  }
}

#define EIGX_NCCL_TRY(cs, expr)                                                                      \
  do {                                                                                               \
    int _r = (expr);                                                                                 \
    if (_r != 0) {                                                                                   \
      fprintf(stderr, "[eigx] RCCL error %d (%s) at %s:%d\n", _r,                                    \
              api.GetErrorString ? api.GetErrorString(_r) : "?", __FILE__, __LINE__);               \
      comm_fail(cs, "RCCL call failed");                                                             \
    }                                                                                                \
  } while (0)

// every rank contributes len <= 128 bytes and receives everybody's; false on time-out
bool board_exchange(CommState* cs, const void* mine, size_t len, unsigned char (*all)[128]) {
  Board* b = cs->board;
  if (!b || cs->board_dead) return false;
  const uint64_t seq = ++cs->board_seq;
  const double t0 = now_s();
  // a round ends without result when it runs out of time or when any rank has reported a failure
  auto timed_out = [&]() {
    if (now_s() - t0 > cs->timeout_s || b->abort_flag.load(std::memory_order_acquire) != 0) { cs->board_dead = true; return true; }
    return false;
  };
  if (timed_out()) return false;
  for (int q = 0; q < cs->P; ++q)
    while (b->seq_read[q].load(std::memory_order_acquire) + 1 < seq) {
      if (timed_out()) return false;
      usleep(50);
    }
  memset(b->slot[cs->me], 0, 128);
  memcpy(b->slot[cs->me], mine, len);
  b->seq_written[cs->me].store(seq, std::memory_order_release);
  for (int q = 0; q < cs->P; ++q)
    while (b->seq_written[q].load(std::memory_order_acquire) < seq) {
      if (timed_out()) return false;
      usleep(50);
    }
  for (int q = 0; q < cs->P; ++q) memcpy(all[q], b->slot[q], 128);
  b->seq_read[cs->me].store(seq, std::memory_order_release);
  return true;
}

uint64_t fnv1a(const void* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= ((const unsigned char*)p)[i]; h *= 1099511628211ull; }
  return h;
}

// fine-grained device memory: stores from a peer (or from another process on the same GPU) are visible to
// system-scope loads without a kernel boundary (RCCL allocates its protocol buffers the same way)
double* alloc_window(size_t bytes) {
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    // no fallback to coarse-grained memory: kernels poll flags and messages in these windows while the producer is
    // still running, which plain hipMalloc memory does not support; the caller treats this as "no peer windows"
    (void)hipGetLastError();
    return nullptr;
  }
  return (double*)p;
}

// ---- device side -----------------------------------------------------------------------------------------------
typedef unsigned long long u64;

// Bounded spin on an epoch flag.  A spin that runs out sets the sticky error word; once it is set every later wait
// of this rank returns at once (the solver reports the failure at the next stage boundary).
__device__ __forceinline__ bool spin_until(const u64* flag, u64 epoch, int* err, long long limit_ticks) {
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
  const long long t0 = wall_clock64();
  // (relaxed polls, one acquire at the end: an acquire load invalidates caches at every look)
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
    __builtin_amdgcn_s_sleep(1);
    // a peer that failed (or this rank's host) sets the failure word: leave at once
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
    if (wall_clock64() - t0 > limit_ticks) {
      __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return true;
}

struct WaitArgs {
  const u64* flag[EIGX_MAXP];
  int n;
  u64 epoch;
  int* err;
  u64* ticks;
  long long limit_ticks;
};
// one wave: lane q waits for member q's flag.  Every exit is bounded (time-out -> sticky error word).
__global__ void wait_kernel(WaitArgs W) {
  const long long t0 = wall_clock64();
  if ((int)threadIdx.x < W.n) spin_until(W.flag[threadIdx.x], W.epoch, W.err, W.limit_ticks);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (threadIdx.x == 0) atomicAdd(W.ticks, (u64)(wall_clock64() - t0));
}

struct ReadyArgs {
  u64* remote[EIGX_MAXP];       // my "ready" word in member q's flag block
  const u64* local[EIGX_MAXP];  // member q's "ready" word in mine
  int n;
  u64 epoch;
  int* err;
  u64* ticks;
  long long limit_ticks;
};
// barrier in front of a bulk push: nobody writes into a receive buffer before its owner's stream has reached the
// operation (i.e. has finished consuming what the previous operation left there)
__global__ void ready_kernel(ReadyArgs R) {
  const long long t0 = wall_clock64();
  if ((int)threadIdx.x < R.n) {
    __hip_atomic_store(R.remote[threadIdx.x], R.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    spin_until(R.local[threadIdx.x], R.epoch, R.err, R.limit_ticks);
  }
  if (threadIdx.x == 0) atomicAdd(R.ticks, (u64)(wall_clock64() - t0));
}

struct PushArgs {
  const double* src;
  size_t src_stride;            // member q reads src + q * src_stride
  double* dst[EIGX_MAXP];       // destination in member q's window (already offset to my slot)
  u64* flag[EIGX_MAXP];         // my data flag in member q's flag block
  int n;
  size_t count;
  u64 epoch;
  unsigned* counter;
  u64* ticks;
};
__global__ __launch_bounds__(256) void push_kernel(PushArgs A) {
  const long long t0 = wall_clock64();
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int q = 0; q < A.n; ++q) {
    const double* s = A.src + (size_t)q * A.src_stride;
    double* d = A.dst[q];
    if (((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
      const size_t n2 = A.count / 2;
      for (size_t i = t; i < n2; i += nthreads) {
        const double2 v = reinterpret_cast<const double2*>(s)[i];
        __builtin_nontemporal_store(v.x, d + 2 * i);
        __builtin_nontemporal_store(v.y, d + 2 * i + 1);
      }
      if ((A.count & 1) && t == 0) __builtin_nontemporal_store(s[A.count - 1], d + A.count - 1);
    } else {
      for (size_t i = t; i < A.count; i += nthreads) __builtin_nontemporal_store(s[i], d + i);
    }
  }
  // every storing wave drains its stores; the last workgroup to arrive publishes the flags
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __threadfence_system();
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0) {
    const unsigned tk = atomicAdd(A.counter, 1u);
    last = (tk == gridDim.x - 1);
    if (last) *A.counter = 0;
  }
  __syncthreads();
  if (last) {
    if ((int)threadIdx.x < A.n) __hip_atomic_store(A.flag[threadIdx.x], A.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) atomicAdd(A.ticks, (u64)(wall_clock64() - t0));
  }
}

// out[i] = op over members r (in group order) of  in[r * stride + i]      (fixed order: bit-identical on every rank)
__global__ void reduce_members_kernel(const double* __restrict__ in, size_t stride, int n, size_t count, int op,
                                      double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double v = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int r = 1; r < n; ++r) {
      const double x = __hip_atomic_load(in + (size_t)r * stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v = (op == kNcclMax) ? fmax(v, x) : v + x;
    }
    out[i] = v;
  }
}
__global__ void copy_sys_kernel(const double* __restrict__ in, size_t count, double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    out[i] = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- init-time transport self-test (comm_init) -------------------------------------------------------------------
// payload of (source rank, round, index): exact in fp64, different for every triple that could be confused
__device__ __host__ inline double st_value(int rank, int round, size_t i) {
  return (double)((long long)(rank + 1) * 1000003ll + (long long)round * 1009ll + (long long)(i % 977));
}
__global__ void st_fill_kernel(double* buf, size_t count, int rank, int round) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    buf[i] = st_value(rank, round, i);
}
struct StMembers { int r[EIGX_MAXP]; };
// win[q * stride + i] must be member q's payload of this round
__global__ void st_check_kernel(const double* win, size_t stride, size_t count, int n, StMembers M, int round, unsigned* errs) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count * n; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i / count);
    const size_t j = i - (size_t)q * count;
    const double v = __hip_atomic_load(win + (size_t)q * stride + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v != st_value(M.r[q], round, j)) atomicAdd(errs, 1u);
  }
}
// buf[i] must be the sum over the members of their payloads (exact: small integers)
__global__ void st_check_sum_kernel(const double* buf, size_t count, int n, StMembers M, int round, unsigned* errs) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double want = 0.0;
    for (int q = 0; q < n; ++q) want += st_value(M.r[q], round, i);
    if (buf[i] != want) atomicAdd(errs, 1u);
  }
}
// the per-step exchange in miniature (band_reduce.hip kl_kernel's tail): system-scope stores into every rank's step
// window, drained, last workgroup publishes the epoch flag on every rank -- no ready handshake, double-buffered by parity
struct StStepArgs {
  double* slot[EIGX_MAXP];        // my message area (parity 0) in rank q's window
  u64* flag[EIGX_MAXP];           // my arrival flag (parity 0) in rank q's flag block
  size_t parity_stride, count;
  unsigned* counter;
  int n, rank, round, fence;
  u64 epoch;
};
__global__ __launch_bounds__(256) void st_step_push_kernel(StStepArgs A) {
  const int par = (int)(A.epoch & 1);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < A.count; i += (size_t)gridDim.x * blockDim.x) {
    const double v = st_value(A.rank, A.round, i);
    for (int d = 0; d < A.n; ++d)
      __hip_atomic_store(A.slot[d] + (size_t)par * A.parity_stride + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // exactly band_reduce.hip's kl_publish: acknowledged write-through stores, barrier, one count per workgroup, the
  // publisher's flag store alone carries the system-scope release (EIGX_STEP_FENCE=1: a fence in every workgroup)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (A.fence) __threadfence_system();
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0) {
    const unsigned tk = atomicAdd(A.counter, 1u);
    last = (tk == gridDim.x - 1);
    if (last) *A.counter = 0;
  }
  __syncthreads();
  if (last && (int)threadIdx.x < A.n)
    __hip_atomic_store(A.flag[threadIdx.x] + par * EIGX_MAXP, A.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

long long g_comm_bounce = (long long)32 << 20;   // doubles per slice of comm_exchange_big's bounce window (eigx_tune key 9)

long long limit_ticks(const CommState* cs) {   // wall_clock64: 100 MHz
  const double t = (cs->in_selftest && cs->timeout_s > 10.0) ? 10.0 : cs->timeout_s;
  return (long long)(t * 1e8);
}

void* pick(const CommState* cs, CommGroup grp) { return grp == COMM_X ? cs->x : grp == COMM_Y ? cs->y : cs->world; }

void rccl_time_begin(CommState* cs, hipStream_t s) {
  if (cs->tev_used + 2 > cs->tev.size())
    for (int q = 0; q < 2; ++q) { hipEvent_t e; EIGX_HIP_CHECK(hipEventCreate(&e)); cs->tev.push_back(e); }
  EIGX_HIP_CHECK(hipEventRecord(cs->tev[cs->tev_used], s));
}
void rccl_time_end(CommState* cs, hipStream_t s) {
  EIGX_HIP_CHECK(hipEventRecord(cs->tev[cs->tev_used + 1], s));
  cs->tev_used += 2;
}

}  // namespace

void comm_report_failure(Context& ctx, const char* what) {
  if (ctx.comm) comm_fail(ctx.comm, what);
}

int comm_group(const Context& ctx, CommGroup grp, int* members, int* my_index) {
  const Grid& g = ctx.grid;
  auto world_rank = [&](int qx, int qy) { return g.row_major ? qx * g.Py + qy : qx + qy * g.Px; };
  int n = 0, mine = 0;
  if (grp == COMM_X) {
    for (int qx = 0; qx < g.Px; ++qx) members[n++] = world_rank(qx, g.py);
    mine = g.px;
  } else if (grp == COMM_Y) {
    for (int qy = 0; qy < g.Py; ++qy) members[n++] = world_rank(g.px, qy);
    mine = g.py;
  } else {
    for (int q = 0; q < g.nranks; ++q) members[n++] = q;
    mine = g.rank;
  }
  if (my_index) *my_index = mine;
  return n;
}
int comm_size(const Context& ctx, CommGroup grp) {
  return grp == COMM_X ? ctx.grid.Px : grp == COMM_Y ? ctx.grid.Py : ctx.grid.nranks;
}
bool comm_failed(const Context& ctx) {   // synchronous (reads one device word): call at stage boundaries only
  CommState* cs = ctx.comm;
  if (!cs) return false;
  if (!cs->failed && cs->err_dev) {
    int e = 0;
    if (hipMemcpy(&e, cs->err_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); e = 1; }
    if (e != 0) {
      fprintf(stderr, "[eigx] rank %d: a wait for a peer ran out after %.0f s\n", cs->me, cs->timeout_s);
      cs->failed = true;
    }
  }
  return cs->failed;
}
bool comm_shared_device(const Context& ctx) { return ctx.comm && ctx.comm->shared_device; }

int64_t comm_held_bytes(const Context& ctx) {
  const CommState* cs = ctx.comm;
  if (!cs) return 0;
  int64_t t = (int64_t)cs->flags.bytes;
  for (const auto& kv : cs->bufs) t += (int64_t)kv.second.bytes;
  for (const PeerBuf& b : cs->retired) t += (int64_t)b.bytes;
  return t;
}

double comm_seconds(Context& ctx, bool reset) {
  CommState* cs = ctx.comm;
  if (!cs) return 0.0;
  u64 ticks = 0;
  EIGX_HIP_CHECK(hipMemcpy(&ticks, cs->ticks_dev, sizeof(u64), hipMemcpyDeviceToHost));
  for (size_t q = 0; q + 1 < cs->tev_used; q += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, cs->tev[q], cs->tev[q + 1]) == hipSuccess) cs->rccl_seconds += 1e-3 * ms;
    else (void)hipGetLastError();
  }
  cs->tev_used = 0;
  const double sec = 1e-8 * (double)ticks + cs->rccl_seconds;
  if (reset) {
    EIGX_HIP_CHECK(hipMemset(cs->ticks_dev, 0, sizeof(u64)));
    cs->rccl_seconds = 0.0;
  }
  return sec;
}

int comm_get_unique_id(void* out128) {
  if (!out128) return EIGX_ERR_BAD_ARG;
  memset(out128, 0, 128);
  // an ncclUniqueId when RCCL is present (it seeds the RCCL communicators as well); otherwise random bytes --
  // either way the 128 bytes name the session (and its shared-memory board)
  if (load_rccl() && api.GetUniqueId(out128) == 0) { g_uid_hash_from_rccl = fnv1a(out128, 128); return EIGX_OK; }
  FILE* f = fopen("/dev/urandom", "rb");
  size_t got = f ? fread(out128, 1, 128, f) : 0;
  if (f) fclose(f);
  if (got != 128) {
    uint64_t h = fnv1a(&got, sizeof(got)) ^ (uint64_t)getpid() ^ (uint64_t)(now_s() * 1e6);
    for (int i = 0; i < 16; ++i) { h = h * 6364136223846793005ull + 1442695040888963407ull; memcpy((char*)out128 + 8 * i, &h, 8); }
  }
  return EIGX_OK;
}

int comm_init(Context& ctx, const void* uid) {
  if (!uid) return EIGX_ERR_BAD_ARG;
  const Grid& g = ctx.grid;
  if (g.nranks > EIGX_MAXP) {
    fprintf(stderr, "[eigx] at most %d ranks (one xGMI node) are supported\n", EIGX_MAXP);
    return EIGX_ERR_BAD_ARG;
  }
  CommState* cs = new CommState();
  cs->P = g.nranks; cs->me = g.rank;
  if (const char* t = getenv("EIGX_COMM_TIMEOUT_S")) { const double v = atof(t); if (v > 0.0) cs->timeout_s = v; }
  if (getenv("EIGX_LOOPBACK")) {
    // Lab mode (tools/mg_step_rehearsal.py): this process plays rank g.rank of the P-rank grid ALONE.  Every peer
    // window and flag block is its own memory and it signals on behalf of every source, so the complete multi-rank
    // kernel sequence of a rank -- local mat-vec with the folded exchange, waits, the replicated K_A, panel gathers --
    // runs at the true local sizes on an otherwise idle GPU and can be timed; the numbers it computes are meaningless.
    cs->loop = true;
    cs->ipc = true;
    EIGX_HIP_CHECK(hipMalloc(&cs->counters, CH_COUNT * sizeof(unsigned)));
    EIGX_HIP_CHECK(hipMemset(cs->counters, 0, CH_COUNT * sizeof(unsigned)));
    EIGX_HIP_CHECK(hipMalloc(&cs->ticks_dev, sizeof(u64)));
    EIGX_HIP_CHECK(hipMemset(cs->ticks_dev, 0, sizeof(u64)));
    cs->flags.bytes = kFlagWordsTotal * sizeof(u64);
    cs->flags.local = alloc_window(cs->flags.bytes);
    if (!cs->flags.local) EIGX_HIP_CHECK(hipMalloc((void**)&cs->flags.local, cs->flags.bytes));
    EIGX_HIP_CHECK(hipMemset(cs->flags.local, 0, cs->flags.bytes));
    cs->err_dev = (int*)((u64*)cs->flags.local + kFlagWords);
    for (int q = 0; q < cs->P; ++q) cs->flags.peer[q] = cs->flags.local;
    cs->flags.mapped = true;
    ctx.comm = cs;
    fprintf(stderr, "[eigx] LOOPBACK: rank %d of %d alone on this GPU (timing rehearsal, results meaningless)\n", cs->me, cs->P);
    return EIGX_OK;
  }
  // ---- board ------------------------------------------------------------------------------------------
  char name[64];
  snprintf(name, sizeof(name), "/eigx-%016llx", (unsigned long long)fnv1a(uid, 128));
  const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { perror("[eigx] shm_open"); delete cs; return EIGX_ERR_INTERNAL; }
  if (ftruncate(fd, sizeof(Board)) != 0) { perror("[eigx] ftruncate"); close(fd); delete cs; return EIGX_ERR_INTERNAL; }
  void* mp = mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (mp == MAP_FAILED) { perror("[eigx] mmap"); delete cs; return EIGX_ERR_INTERNAL; }
  cs->board = (Board*)mp;
  cs->board->attached.fetch_add(1);
  {
    const double t0 = now_s();
    while (cs->board->attached.load() < (uint32_t)cs->P) {
      if (now_s() - t0 > cs->timeout_s) {
        fprintf(stderr, "[eigx] rank %d: only %u of %d ranks reached eigx_init_multi\n", cs->me, cs->board->attached.load(), cs->P);
        shm_unlink(name);
        munmap(mp, sizeof(Board)); cs->board = nullptr; delete cs; return EIGX_ERR_INTERNAL;
      }
      usleep(100);
    }
  }
  // ---- local words ------------------------------------------------------------------------------------
  EIGX_HIP_CHECK(hipMalloc(&cs->counters, CH_COUNT * sizeof(unsigned)));
  EIGX_HIP_CHECK(hipMemset(cs->counters, 0, CH_COUNT * sizeof(unsigned)));
  EIGX_HIP_CHECK(hipMalloc(&cs->ticks_dev, sizeof(u64)));
  EIGX_HIP_CHECK(hipMemset(cs->ticks_dev, 0, sizeof(u64)));
  // ---- flag block + identity exchange -----------------------------------------------------------------
  InitBlob mine;
  memset(&mine, 0, sizeof(mine));
  mine.pid = (int)getpid();
  if (hipDeviceGetPCIBusId(mine.bus_id, sizeof(mine.bus_id), ctx.device) != hipSuccess) { (void)hipGetLastError(); mine.bus_id[0] = 0; }
  cs->flags.bytes = kFlagWordsTotal * sizeof(u64);
  cs->flags.local = alloc_window(cs->flags.bytes);
  mine.ipc_ok = 0;
  if (cs->flags.local) {
    EIGX_HIP_CHECK(hipMemset(cs->flags.local, 0, cs->flags.bytes));
    if (hipIpcGetMemHandle(&mine.flags_handle, cs->flags.local) == hipSuccess) mine.ipc_ok = 1;
    else (void)hipGetLastError();
    // the sticky failure word is a word of the peer-mapped block: a failing rank sets it on every peer (comm_fail)
    cs->err_dev = (int*)((u64*)cs->flags.local + kFlagWords);
    cs->err_in_flags = true;
  } else {
    // no fine-grained memory: no peer windows (the flag block is polled while peers write it); plain words for RCCL-only use
    EIGX_HIP_CHECK(hipMalloc((void**)&cs->flags.local, cs->flags.bytes));
    EIGX_HIP_CHECK(hipMemset(cs->flags.local, 0, cs->flags.bytes));
    cs->err_dev = (int*)((u64*)cs->flags.local + kFlagWords);
  }
  cs->flags.peer[cs->me] = cs->flags.local;
  if (getenv("EIGX_NO_IPC")) mine.ipc_ok = 0;
  // RCCL is used only if EVERY rank resolved the library and the session id really is an ncclUniqueId (only the process
  // that made it can tell): both facts travel with the bootstrap blob, so the decision is the same everywhere BEFORE
  // anybody enters the blocking ncclCommInitRank
  mine.rccl_loaded = load_rccl() ? 1 : 0;
  mine.uid_from_rccl = (g_uid_hash_from_rccl != 0 && fnv1a(uid, 128) == g_uid_hash_from_rccl) ? 1 : 0;
  unsigned char all[EIGX_MAXP][128];
  ctx.comm = cs;
  auto give_up = [&](const char* why) {   // nothing of a rejected init stays behind
    fprintf(stderr, "[eigx] rank %d: %s\n", cs->me, why);
    if (cs->me == 0) shm_unlink(name);
    cs->board_dead = true;                // no farewell exchange in comm_free: the peers may be gone
    comm_free(ctx);
    return EIGX_ERR_INTERNAL;
  };
  if (!board_exchange(cs, &mine, sizeof(mine), all)) return give_up("bootstrap exchange timed out");
  if (cs->me == 0) shm_unlink(name);   // everybody has it mapped: the name can go (the memory lives until the last munmap)
  bool all_ipc = true, all_rccl_loaded = true, uid_is_nccl = false;
  for (int q = 0; q < cs->P; ++q) {
    InitBlob b; memcpy(&b, all[q], sizeof(b));
    if (!b.ipc_ok) all_ipc = false;
    if (!b.rccl_loaded) all_rccl_loaded = false;
    if (b.uid_from_rccl) uid_is_nccl = true;
    for (int r = 0; r < q; ++r) {
      InitBlob c; memcpy(&c, all[r], sizeof(c));
      if (b.bus_id[0] && strncmp(b.bus_id, c.bus_id, sizeof(b.bus_id)) == 0) cs->shared_device = true;
    }
  }
  // one agreed yes / no over all ranks (a board round); false also when the board timed out
  auto vote = [&](int my_ok, bool* verdict) {
    unsigned char ok_all[EIGX_MAXP][128];
    if (!board_exchange(cs, &my_ok, sizeof(my_ok), ok_all)) return false;
    bool v = true;
    for (int q = 0; q < cs->P; ++q) { int x; memcpy(&x, ok_all[q], sizeof(x)); if (!x) v = false; }
    *verdict = v;
    return true;
  };
  int map_ok = all_ipc ? 1 : 0;
  if (all_ipc) {
    for (int q = 0; q < cs->P; ++q) {
      if (q == cs->me) continue;
      InitBlob b; memcpy(&b, all[q], sizeof(b));
      void* p = nullptr;
      if (hipIpcOpenMemHandle(&p, b.flags_handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        map_ok = 0;
        break;
      }
      cs->flags.peer[q] = (double*)p;     // closed by comm_free whatever the verdict below
    }
  }
  bool ipc_mapped = false;
  if (!vote(map_ok, &ipc_mapped)) return give_up("bootstrap exchange timed out");
  cs->ipc = ipc_mapped;
  cs->flags.mapped = ipc_mapped;
  // ---- RCCL communicators: need one GPU per rank, the library on every rank, a real ncclUniqueId ------------------
  const bool try_rccl = !cs->shared_device && all_rccl_loaded && uid_is_nccl;
  int rccl_ok = 0;
  if (try_rccl) {
    ncclUniqueIdBlob id;
    memcpy(id.internal, uid, 128);
    const int rc_init = api.CommInitRank(&cs->world, g.nranks, id, g.rank);
    if (rc_init != 0) {
      fprintf(stderr, "[eigx] ncclCommInitRank failed: %d (%s)\n", rc_init, api.GetErrorString ? api.GetErrorString(rc_init) : "?");
      cs->world = nullptr;
    } else {
      // X: ranks that share my py, ordered by px; Y: ranks that share my px, ordered by py (src/eigen_libs0.F:579-585)
      const int rx = api.CommSplit(cs->world, g.py, g.px, &cs->x, nullptr);
      const int ry = api.CommSplit(cs->world, g.px, g.py, &cs->y, nullptr);
      if (rx == 0 && ry == 0) rccl_ok = 1;
      else fprintf(stderr, "[eigx] ncclCommSplit failed: %d %d\n", rx, ry);
    }
  }
  bool rccl_up = false;
  if (!vote(rccl_ok, &rccl_up)) return give_up("bootstrap exchange timed out");

  // ---- transport self-test: both transports carry checksummed payloads before the solver relies on them --------------
  // (what a multi-rank test on ONE card cannot show: cross-device IPC mappings, xGMI visibility of system-scope stores
  // and flags, RCCL's X / Y sub-communicators).  EIGX_SELFTEST_ROUNDS=0 skips it; EIGX_SELFTEST_FAIL=ipc|rccl makes the
  // named leg report failure (tests of the fallback ladder).
  int rounds = 400;
  if (const char* e = getenv("EIGX_SELFTEST_ROUNDS")) rounds = atoi(e);
  const char* force_fail = getenv("EIGX_SELFTEST_FAIL");
  hipStream_t ts = nullptr;
  EIGX_HIP_CHECK(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
  unsigned* errs = nullptr;
  EIGX_HIP_CHECK(hipMalloc(&errs, 2 * sizeof(unsigned)));
  EIGX_HIP_CHECK(hipMemset(errs, 0, 2 * sizeof(unsigned)));
  const size_t cnt = 1024;   // doubles per message
  double* sendb = nullptr;
  EIGX_HIP_CHECK(hipMalloc(&sendb, cnt * EIGX_MAXP * sizeof(double)));
  StMembers Mw, Mx, My;
  int nx_ = 0, ny_ = 0;
  { int mi; comm_group(ctx, COMM_WORLD, Mw.r, &mi); nx_ = comm_group(ctx, COMM_X, Mx.r, &mi); ny_ = comm_group(ctx, COMM_Y, My.r, &mi); }
  auto read_errs = [&](int which) {
    unsigned h[2] = {1, 1};
    if (hipStreamSynchronize(ts) != hipSuccess) { (void)hipGetLastError(); return 1u << 30; }
    if (hipMemcpy(h, errs, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 1u << 30; }
    return h[which];
  };
  bool ipc_good = false;
  cs->in_selftest = true;
  if (cs->ipc) {
    int my_ok = 1;
    // Every board round of this block -- the two collective allocations and the two votes -- is made by EVERY rank whatever
    // it has seen locally: checksum errors are local to the receiver, and a rank that skipped a round would pair its next
    // message with the others' current one (garbage verdicts, or a 120-s stall).  Only kernel rounds are skipped.
    // EIGX_SELFTEST_FAIL=ipc makes this leg report failure; with EIGX_SELFTEST_FAIL_RANK=r on rank r alone (tests).
    const char* fail_rank = getenv("EIGX_SELFTEST_FAIL_RANK");
    const bool forced_ipc = force_fail && strcmp(force_fail, "ipc") == 0 && (!fail_rank || atoi(fail_rank) == cs->me);
    cs->rccl = false;
    PeerBuf* w = comm_buffer(ctx, "comm.selftest", (size_t)EIGX_MAXP * cnt * sizeof(double));
    StepPeers sp;
    double* win = comm_step_window(ctx, 0, cnt, &sp);
    if (!w->mapped || cs->failed) my_ok = 0;
    // (1) bulk protocol: ready handshake + push kernel + flag + wait kernel, world all-gather, every round checked
    const double t0 = now_s();
    for (int r = 1; my_ok && r <= rounds; ++r) {
      hipLaunchKernelGGL(st_fill_kernel, dim3(4), dim3(256), 0, ts, sendb, cnt, cs->me, r);
      comm_exchange(ctx, COMM_WORLD, sendb, 0, w, 0, cnt, ts, CH_BULK);
      hipLaunchKernelGGL(st_check_kernel, dim3(8), dim3(256), 0, ts, (const double*)w->local, cnt, cnt, cs->P, Mw, r, errs);
    }
    cs->st_ipc_rounds = my_ok ? rounds : 0;
    cs->st_ipc_errors = my_ok ? (int)read_errs(0) : 1;
    if (forced_ipc && fail_rank) cs->st_ipc_errors += 1;     // (test hook: checksum errors seen by this rank only)
    cs->st_ipc_us = rounds > 0 ? (now_s() - t0) * 1e6 / rounds : 0.0;
    if (cs->st_ipc_errors != 0 || cs->failed) my_ok = 0;
    bool bulk_good = false;
    if (!vote(my_ok, &bulk_good)) { (void)hipFree(errs); (void)hipFree(sendb); (void)hipStreamDestroy(ts); return give_up("bootstrap exchange timed out"); }
    // (2) step protocol: system-scope stores straight into every rank's step window, epoch flags, parity double
    // buffering, no ready handshake -- the per-step exchanges of the reduction in miniature.  Run by all ranks or none.
    if (bulk_good) {
      StStepArgs A;
      A.n = cs->P; A.rank = cs->me; A.count = cnt; A.parity_stride = sp.parity_stride; A.counter = sp.counter;
      A.fence = (getenv("EIGX_STEP_FENCE") && atoi(getenv("EIGX_STEP_FENCE")) != 0) ? 1 : 0;
      for (int q = 0; q < EIGX_MAXP; ++q) { A.slot[q] = sp.slot[q]; A.flag[q] = sp.flag[q]; }
      const u64 base = comm_step_epoch_base(ctx, 0, (u64)rounds + 2);
      const double t1 = now_s();
      for (int r = 1; r <= rounds; ++r) {
        A.round = r; A.epoch = base + r;
        hipLaunchKernelGGL(st_step_push_kernel, dim3(4), dim3(256), 0, ts, A);
        comm_step_wait(ctx, 0, A.epoch, ts);
        hipLaunchKernelGGL(st_check_kernel, dim3(8), dim3(256), 0, ts, (const double*)(win + (A.epoch & 1) * sp.parity_stride), cnt, cnt,
                           cs->P, Mw, r, errs + 1);
      }
      cs->st_step_rounds = rounds;
      cs->st_step_errors = (int)read_errs(1);
      cs->st_step_us = rounds > 0 ? (now_s() - t1) * 1e6 / rounds : 0.0;
    }
    if (!bulk_good || cs->st_ipc_errors != 0 || (cs->st_step_rounds > 0 && cs->st_step_errors != 0) || cs->failed) my_ok = 0;
    if (my_ok && comm_failed(ctx)) my_ok = 0;
    if (forced_ipc) my_ok = 0;
    if (!my_ok && cs->me == 0)
      fprintf(stderr, "[eigx] transport self-test: peer windows FAILED (bulk errors %d, step errors %d%s)\n", cs->st_ipc_errors,
              cs->st_step_errors, (force_fail && strcmp(force_fail, "ipc") == 0) ? ", forced by EIGX_SELFTEST_FAIL" : "");
    // a failed self-test leaves the failure word set; the verdict is what counts from here on
    cs->failed = false;
    { const int zero = 0; EIGX_HIP_CHECK(hipMemcpy(cs->err_dev, &zero, sizeof(int), hipMemcpyHostToDevice)); }
    if (!vote(my_ok, &ipc_good)) { (void)hipFree(errs); (void)hipFree(sendb); (void)hipStreamDestroy(ts); return give_up("bootstrap exchange timed out"); }
  }
  bool rccl_good = false;
  if (rccl_up) {
    int my_ok = 1;
    EIGX_HIP_CHECK(hipMemset(errs, 0, 2 * sizeof(unsigned)));
    double* recvb = nullptr;
    EIGX_HIP_CHECK(hipMalloc(&recvb, cnt * EIGX_MAXP * sizeof(double)));
    const int checks = rounds > 0 ? (rounds < 20 ? rounds : 20) : 0;
    const double t0 = now_s();
    for (int r = 1; r <= checks; ++r) {
      // all-reduce over X, Y and world against the exact sum; all-gather over world; grouped send / receive all-to-all
      struct { void* c; int n; StMembers* M; } grp[3] = {{cs->x, nx_, &Mx}, {cs->y, ny_, &My}, {cs->world, cs->P, &Mw}};
      for (int gi = 0; gi < 3; ++gi) {
        hipLaunchKernelGGL(st_fill_kernel, dim3(4), dim3(256), 0, ts, sendb, cnt, cs->me, r);
        if (api.AllReduce(sendb, sendb, cnt, kNcclFloat64, kNcclSum, grp[gi].c, ts) != 0) my_ok = 0;
        hipLaunchKernelGGL(st_check_sum_kernel, dim3(4), dim3(256), 0, ts, (const double*)sendb, cnt, grp[gi].n, *grp[gi].M, r, errs);
      }
      hipLaunchKernelGGL(st_fill_kernel, dim3(4), dim3(256), 0, ts, sendb, cnt, cs->me, r);
      if (api.AllGather(sendb, recvb, cnt, kNcclFloat64, cs->world, ts) != 0) my_ok = 0;
      hipLaunchKernelGGL(st_check_kernel, dim3(8), dim3(256), 0, ts, (const double*)recvb, cnt, cnt, cs->P, Mw, r, errs);
      EIGX_HIP_CHECK(hipMemsetAsync(recvb, 0, cnt * cs->P * sizeof(double), ts));
      if (api.GroupStart() != 0) my_ok = 0;
      for (int q = 0; q < cs->P; ++q) {
        if (api.Send(sendb, cnt, kNcclFloat64, q, cs->world, ts) != 0) my_ok = 0;
        if (api.Recv(recvb + (size_t)q * cnt, cnt, kNcclFloat64, q, cs->world, ts) != 0) my_ok = 0;
      }
      if (api.GroupEnd() != 0) my_ok = 0;
      hipLaunchKernelGGL(st_check_kernel, dim3(8), dim3(256), 0, ts, (const double*)recvb, cnt, cnt, cs->P, Mw, r, errs);
    }
    cs->st_rccl_checks = checks * 5;
    cs->st_rccl_errors = (int)read_errs(0) + (my_ok ? 0 : 1);
    cs->st_rccl_us = checks > 0 ? (now_s() - t0) * 1e6 / (checks * 5) : 0.0;
    (void)hipFree(recvb);
    if (cs->st_rccl_errors != 0) my_ok = 0;
    if (force_fail && strcmp(force_fail, "rccl") == 0) my_ok = 0;
    if (!my_ok && cs->me == 0) fprintf(stderr, "[eigx] transport self-test: RCCL FAILED (%d errors)\n", cs->st_rccl_errors);
    if (!vote(my_ok, &rccl_good)) { (void)hipFree(errs); (void)hipFree(sendb); (void)hipStreamDestroy(ts); return give_up("bootstrap exchange timed out"); }
  }
  (void)hipFree(errs);
  (void)hipFree(sendb);
  (void)hipStreamDestroy(ts);
  cs->in_selftest = false;
  cs->ipc = ipc_good;
  cs->rccl_ok = rccl_good;

  // ---- the ladder: peer windows -> RCCL -> (caller falls back to independent replicas) -----------------------------
  // Per-step exchanges: peer windows (kernel stores into the peers' HBM + epoch flags) when they passed the self-test,
  // otherwise -- or with EIGX_STEP=coll -- allgathers (ncclAllGather on a node, the same group semantics emulated over
  // the peer windows when ranks share a card).  Bulk collectives (panel gather, reflector gather, the all-to-alls): RCCL
  // wherever its communicators exist and passed the self-test, i.e. on a node with one rank per GPU -- windows of N^2/P
  // doubles are then never peer-mapped (comm_buffer) --, the peer windows where RCCL cannot run (ranks sharing a card:
  // every multi-rank test) or where EIGX_BULK=ipc asks for them.
  if (!cs->ipc && !cs->rccl_ok) {
    fprintf(stderr, "[eigx] rank %d: neither peer windows (hipIpc) nor RCCL are usable between the ranks\n", cs->me);
    comm_free(ctx);
    return EIGX_ERR_INTERNAL;
  }
  const char* want_bulk = getenv("EIGX_BULK");
  const char* want_step = getenv("EIGX_STEP");
  cs->rccl = cs->rccl_ok && !(cs->ipc && want_bulk && strcmp(want_bulk, "ipc") == 0);
  cs->step_coll = !cs->ipc || (want_step && strcmp(want_step, "coll") == 0);
  if (getenv("EIGX_TRACE_COMM") && cs->me == 0) {
    char info[1536];
    comm_info(ctx, info, sizeof(info));
    fprintf(stderr, "[eigx] transport: %s\n", info);
  }
  return EIGX_OK;
}

int comm_info(const Context& ctx, char* buf, int len) {
  const CommState* cs = ctx.comm;
  if (!buf || len <= 0) return EIGX_ERR_BAD_ARG;
  if (!cs) { snprintf(buf, (size_t)len, "{\"ranks\": 1}"); return EIGX_OK; }
  const bool fused = comm_step_wait_fused(ctx);
  snprintf(buf, (size_t)len,
           "{\"ranks\": %d, \"shared_device\": %s, \"peer_windows\": %s, \"rccl\": %s, \"step_exchange\": \"%s\", "
           "\"step_wait\": \"%s\", \"bulk\": \"%s\", \"selftest\": {\"ipc_rounds\": %d, \"ipc_errors\": %d, \"ipc_us_per_round\": %.1f, "
           "\"step_rounds\": %d, \"step_errors\": %d, \"step_us_per_round\": %.1f, \"rccl_checks\": %d, \"rccl_errors\": %d, "
           "\"rccl_us_per_call\": %.1f}, \"since_init\": {\"step_exchanges\": %.0f, \"step_bytes_sent\": %.0f, \"small_exchanges\": %.0f, "
           "\"small_bytes_sent\": %.0f, \"allreduces\": %.0f, \"allreduce_bytes_sent\": %.0f, \"large_exchanges\": %.0f, \"large_bytes_sent\": %.0f}}",
           cs->P, cs->shared_device ? "true" : "false", cs->ipc ? "true" : "false", cs->rccl_ok ? "true" : "false",
           cs->step_coll ? (cs->rccl ? "allgather (RCCL)" : "allgather (peer-window emulation)") : "peer writes (hipIpc windows, kernel stores over xGMI)",
           cs->step_coll ? "stream order" : (fused ? "fused into the consumer kernels" : "wait kernel"), cs->rccl ? "rccl" : "peer windows",
           cs->st_ipc_rounds, cs->st_ipc_errors, cs->st_ipc_us, cs->st_step_rounds, cs->st_step_errors, cs->st_step_us,
           cs->st_rccl_checks, cs->st_rccl_errors, cs->st_rccl_us, cs->st_calls[0], cs->st_bytes[0], cs->st_calls[1], cs->st_bytes[1],
           cs->st_calls[2], cs->st_bytes[2], cs->st_calls[3], cs->st_bytes[3]);
  return EIGX_OK;
}

void comm_free(Context& ctx) {
  CommState* cs = ctx.comm;
  if (!cs) return;
  // nobody unmaps while a peer may still be writing: meet at the board first (bounded) -- unless the communicator has
  // already failed or a board round timed out: the peers may be gone, and waiting for them twice more would stall
  // eigx_free for minutes
  const bool polite = cs->board && !cs->board_dead && !cs->failed;
  if (polite) { int z = 0; unsigned char all[EIGX_MAXP][128]; (void)board_exchange(cs, &z, sizeof(z), all); }
  std::vector<PeerBuf> allb(cs->retired);
  for (auto& kv : cs->bufs) allb.push_back(kv.second);
  // every imported mapping is closed, whether or not the import round as a whole succeeded
  for (PeerBuf& b : allb) {
    for (int q = 0; q < cs->P; ++q)
      if (q != cs->me && b.peer[q] && !cs->loop) { if (hipIpcCloseMemHandle(b.peer[q]) != hipSuccess) (void)hipGetLastError(); }
  }
  for (int q = 0; q < cs->P; ++q)
    if (q != cs->me && cs->flags.peer[q] && !cs->loop) { if (hipIpcCloseMemHandle(cs->flags.peer[q]) != hipSuccess) (void)hipGetLastError(); }
  if (polite && !cs->board_dead) { int z = 0; unsigned char all[EIGX_MAXP][128]; (void)board_exchange(cs, &z, sizeof(z), all); }
  for (PeerBuf& b : allb)
    if (b.local) { if (hipFree(b.local) != hipSuccess) (void)hipGetLastError(); }
  if (cs->flags.local) { if (hipFree(cs->flags.local) != hipSuccess) (void)hipGetLastError(); }   // holds the failure word too
  if (cs->x) api.CommDestroy(cs->x);
  if (cs->y) api.CommDestroy(cs->y);
  if (cs->world) api.CommDestroy(cs->world);
  if (cs->counters) (void)hipFree(cs->counters);
  if (cs->ticks_dev) (void)hipFree(cs->ticks_dev);
  for (hipEvent_t e : cs->tev) (void)hipEventDestroy(e);
  if (cs->board) munmap(cs->board, sizeof(Board));
  delete cs;
  ctx.comm = nullptr;
}

PeerBuf* comm_buffer(Context& ctx, const std::string& name, size_t bytes) {
  CommState* cs = ctx.comm;
  PeerBuf& b = cs->bufs[name];
  if (b.bytes >= bytes && b.local) return &b;
  // (re)allocation is collective: every rank asks for the same size at the same point of the program.
  // A buffer that has to grow is RETIRED, not freed: its mappings stay valid on every peer until comm_free, so no
  // rank can ever store through a stale pointer and no IPC handle is closed and re-opened in the life of the
  // communicator (sizes grow geometrically, so the retired copies add up to less than the live one).
  stage_trace(cs->me, name.c_str(), (long)bytes);
  EIGX_HIP_CHECK(hipDeviceSynchronize());
  stage_trace(cs->me, "  device idle");
  if (b.local) {
    cs->retired.push_back(b);
    b = PeerBuf();
  }
  // head room only for the small windows (they grow with the panel width and the like); the big ones are sized by N
  const size_t want = (bytes < ((size_t)64 << 20)) ? bytes + bytes / 2 + 256 : bytes + 256;
  // only what kernels store into needs a peer mapping: everything when the peer windows carry the bulk collectives too
  // (ranks sharing a card), otherwise just the per-step window -- RCCL takes plain device pointers
  if (cs->loop) {   // lab mode: every "peer" is this buffer
    const size_t want_l = bytes + bytes / 2 + 256;
    EIGX_HIP_CHECK(hipMalloc((void**)&b.local, want_l));
    EIGX_HIP_CHECK(hipMemset(b.local, 0, want_l));
    b.bytes = want_l;
    for (int q = 0; q < cs->P; ++q) b.peer[q] = b.local;
    b.mapped = true;
    return &b;
  }
  const bool map = cs->ipc && (!cs->rccl || ((name == "comm.step" || name == "comm.stepx") && !cs->step_coll));
  b.local = map ? alloc_window(want) : nullptr;
  const bool window_ok = b.local != nullptr;      // false with map: no fine-grained memory -> reported as a failed mapping below
  if (!b.local) EIGX_HIP_CHECK(hipMalloc((void**)&b.local, want));
  stage_trace(cs->me, "  allocated");
  b.bytes = want;
  b.peer[cs->me] = b.local;
  if (map) {
    BufBlob mine; memset(&mine, 0, sizeof(mine));
    mine.bytes = want;
    hipError_t e1 = window_ok ? hipIpcGetMemHandle(&mine.handle, b.local) : hipErrorOutOfMemory;
    mine.ok = (e1 == hipSuccess) ? 1 : 0;
    if (!mine.ok) { fprintf(stderr, "[eigx] rank %d: hipIpcGetMemHandle(%s, %zu bytes): %s\n", cs->me, name.c_str(), want, hipGetErrorString(e1)); (void)hipGetLastError(); }
    unsigned char all[EIGX_MAXP][128];
    stage_trace(cs->me, "  handle exported");
    if (!board_exchange(cs, &mine, sizeof(mine), all)) { comm_fail(cs, "buffer exchange timed out"); return &b; }
    stage_trace(cs->me, "  handles exchanged");
    bool ok = true;
    for (int q = 0; q < cs->P && ok; ++q) {
      BufBlob o; memcpy(&o, all[q], sizeof(o));
      if (!o.ok || o.bytes != want) {
        fprintf(stderr, "[eigx] rank %d: buffer %s: rank %d offers %llu bytes (ok %d), expected %zu\n", cs->me, name.c_str(), q,
                (unsigned long long)o.bytes, o.ok, want);
        ok = false;
      }
    }
    // The ranks import one after the other (a turn per rank, a board round between the turns), so that a failure is
    // attributed to one rank and agreed on by all.  Known limit of this driver: importing 3-GiB windows of three other
    // processes on the same card never returns from hipIpcOpenMemHandle (N = 32768 on a 2 x 2 grid sharing one GPU;
    // 1.6-GiB windows are fine, and so are 6-GiB ones between two bare processes: tools/ipc_big.hip).  A node with one
    // rank per GPU never maps windows of that size (RCCL carries the bulk collectives there, see `map` above).
    for (int turn = 0; turn < cs->P; ++turn) {
      if (turn == cs->me && ok) {
        for (int q = 0; q < cs->P && ok; ++q) {
          if (q == cs->me) continue;
          BufBlob o; memcpy(&o, all[q], sizeof(o));
          void* p = nullptr;
          hipError_t e2 = hipIpcOpenMemHandle(&p, o.handle, hipIpcMemLazyEnablePeerAccess);
          if (e2 != hipSuccess) {
            fprintf(stderr, "[eigx] rank %d: hipIpcOpenMemHandle(%s of rank %d, %zu bytes): %s\n", cs->me, name.c_str(), q, want,
                    hipGetErrorString(e2));
            (void)hipGetLastError();
            ok = false;
            break;
          }
          b.peer[q] = (double*)p;
        }
        stage_trace(cs->me, "  peers imported");
      }
      unsigned char okb = ok ? 1 : 0, oks[EIGX_MAXP][128];
      if (!board_exchange(cs, &okb, 1, oks)) { comm_fail(cs, "buffer exchange timed out"); return &b; }
      for (int q = 0; q < cs->P; ++q) if (!oks[q][0]) ok = false;   // a failure anywhere ends the turns on every rank alike
    }
    if (!ok) comm_fail(cs, "mapping a peer buffer failed");
    b.mapped = ok;
  }
  stage_trace(cs->me, "window mapped");
  return &b;
}

static u64* flag_word(double* block, int ch, int kind, int src) { return (u64*)block + flag_index(ch, kind, src); }
// index under which this rank signals rank q (loopback: it stands in for q itself, so that its own waits are satisfied)
static int sig_idx(const CommState* cs, int q) { return cs->loop ? q : cs->me; }

void comm_exchange(Context& ctx, CommGroup grp, const double* send, size_t send_stride, PeerBuf* recv, size_t recv_off,
                   size_t count, hipStream_t s, CommChannel ch) {
  CommState* cs = ctx.comm;
  int members[EIGX_MAXP], mine = 0;
  const int n = comm_group(ctx, grp, members, &mine);
  if (n == 1) {
    if (count && send != recv->local + recv_off)
      EIGX_HIP_CHECK(hipMemcpyAsync(recv->local + recv_off, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
    return;
  }
  if (cs->failed) return;
  if (ch != CH_STEP2 && ch != CH_STEPX2) { cs->st_calls[1] += 1.0; cs->st_bytes[1] += 8.0 * (double)count * (n - 1); }
  if (cs->rccl) {
    rccl_time_begin(cs, s);
    if (send_stride == 0) {
      EIGX_NCCL_TRY(cs, api.AllGather(send, recv->local + recv_off, count, kNcclFloat64, pick(cs, grp), s));
    } else {
      EIGX_NCCL_TRY(cs, api.GroupStart());
      for (int r = 0; r < n; ++r) {
        EIGX_NCCL_TRY(cs, api.Send(send + (size_t)r * send_stride, count, kNcclFloat64, r, pick(cs, grp), s));
        EIGX_NCCL_TRY(cs, api.Recv(recv->local + recv_off + (size_t)r * count, count, kNcclFloat64, r, pick(cs, grp), s));
      }
      EIGX_NCCL_TRY(cs, api.GroupEnd());
    }
    rccl_time_end(cs, s);
    return;
  }
  if (!recv->mapped) { comm_fail(cs, "exchange into an unmapped buffer"); return; }
  const u64 epoch = ++cs->epoch[ch];
  ReadyArgs R;
  WaitArgs W;
  PushArgs A;
  R.n = W.n = A.n = n;
  R.epoch = W.epoch = A.epoch = epoch;
  R.err = W.err = cs->err_dev;
  R.ticks = W.ticks = A.ticks = cs->ticks_dev;
  R.limit_ticks = W.limit_ticks = limit_ticks(cs);
  A.src = send; A.src_stride = send_stride; A.count = count; A.counter = cs->counters + ch;
  for (int r = 0; r < n; ++r) {
    const int q = members[r];
    R.remote[r] = flag_word(cs->flags.peer[q], ch, 1, sig_idx(cs, q));
    R.local[r] = flag_word(cs->flags.local, ch, 1, q);
    A.dst[r] = recv->peer[q] + recv_off + (size_t)(cs->loop ? r : mine) * count;
    A.flag[r] = flag_word(cs->flags.peer[q], ch, 0, sig_idx(cs, q));
    W.flag[r] = flag_word(cs->flags.local, ch, 0, q);
  }
  hipLaunchKernelGGL(ready_kernel, dim3(1), dim3(64), 0, s, R);
  if (count > 0) {
    size_t blocks = (count * n + 4095) / 4096;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(push_kernel, dim3((unsigned)blocks), dim3(256), 0, s, A);
    hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, s, W);
  }
}

static void allreduce_impl(Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s, CommChannel ch, int op) {
  CommState* cs = ctx.comm;
  const int n = comm_size(ctx, grp);
  if (n == 1 || count == 0 || cs->failed) return;
  cs->st_calls[2] += 1.0; cs->st_bytes[2] += 8.0 * (double)count * (n - 1);
  if (cs->rccl) {
    rccl_time_begin(cs, s);
    EIGX_NCCL_TRY(cs, api.AllReduce(buf, buf, count, kNcclFloat64, op, pick(cs, grp), s));
    rccl_time_end(cs, s);
    return;
  }
  // gather every member's vector, then reduce in member order (same order on every rank)
  PeerBuf* w = comm_buffer(ctx, ch == CH_SIDE ? "comm.ar.side" : "comm.ar", (size_t)EIGX_MAXP * count * sizeof(double));
  comm_exchange(ctx, grp, buf, 0, w, 0, count, s, ch);
  size_t blocks = (count + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(reduce_members_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w->local, count, n, count, op, buf);
}
void comm_allreduce_sum(Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s, CommChannel ch) {
  allreduce_impl(ctx, grp, buf, count, s, ch, kNcclSum);
}
void comm_allreduce_max(Context& ctx, CommGroup grp, double* buf, size_t count, hipStream_t s, CommChannel ch) {
  allreduce_impl(ctx, grp, buf, count, s, ch, kNcclMax);
}

void comm_allgather(Context& ctx, CommGroup grp, const double* send, double* recv, size_t count, hipStream_t s,
                    CommChannel ch) {
  CommState* cs = ctx.comm;
  const int n = comm_size(ctx, grp);
  if (n == 1) {
    if (send != recv && count) EIGX_HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
    return;
  }
  if (cs->failed || count == 0) return;
  if (cs->rccl) {
    rccl_time_begin(cs, s);
    EIGX_NCCL_TRY(cs, api.AllGather(send, recv, count, kNcclFloat64, pick(cs, grp), s));
    rccl_time_end(cs, s);
    return;
  }
  if (ch == CH_BULK && (size_t)n * count > (size_t)g_comm_bounce) {   // large (eigen_h's matrix gather): bounded window
    comm_exchange_big(ctx, grp, send, 0, recv, count, s);
    return;
  }
  PeerBuf* w = comm_buffer(ctx, ch == CH_SIDE ? "comm.ag.side" : "comm.ag", (size_t)n * count * sizeof(double));
  comm_exchange(ctx, grp, send, 0, w, 0, count, s, ch);
  size_t blocks = ((size_t)n * count + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(copy_sys_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w->local, (size_t)n * count, recv);
}

// All-to-all of LARGE pieces (the eigenvector redistributions: N^2 / P doubles per rank) into a plain local buffer:
//   recv[r * count + i] = send_r[(my index) * send_stride + i]   for every member r of the group
// (send_stride = count: all-to-all; 0: allgather).  RCCL: one collective / grouped send + receive, no window at all.  Peer windows: pairwise rounds (round k: send to member me + k,
// receive from member me - k) through ONE bounce window of at most BOUNCE doubles, slice by slice -- every slice is a
// ready / push / wait round of its own and is copied out of the window before the next one is admitted -- so the
// hipIpc-mapped memory stays bounded whatever N is (windows of 2-3 GiB could not be imported on a shared card).
void comm_exchange_big(Context& ctx, CommGroup grp, const double* send, size_t send_stride, double* recv, size_t count,
                       hipStream_t s) {
  CommState* cs = ctx.comm;
  int members[EIGX_MAXP], mine = 0;
  const int n = comm_group(ctx, grp, members, &mine);
  if (count == 0) return;
  if (n == 1) {
    if (send != recv) EIGX_HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s));
    return;
  }
  if (cs->failed) return;
  cs->st_calls[3] += 1.0; cs->st_bytes[3] += 8.0 * (double)count * (n - 1);
  if (cs->rccl) {
    rccl_time_begin(cs, s);
    if (send_stride == 0) {
      EIGX_NCCL_TRY(cs, api.AllGather(send, recv, count, kNcclFloat64, pick(cs, grp), s));
    } else {
      EIGX_NCCL_TRY(cs, api.GroupStart());
      for (int r = 0; r < n; ++r) {
        EIGX_NCCL_TRY(cs, api.Send(send + (size_t)r * send_stride, count, kNcclFloat64, r, pick(cs, grp), s));
        EIGX_NCCL_TRY(cs, api.Recv(recv + (size_t)r * count, count, kNcclFloat64, r, pick(cs, grp), s));
      }
      EIGX_NCCL_TRY(cs, api.GroupEnd());
    }
    rccl_time_end(cs, s);
    return;
  }
  const size_t BOUNCE = (size_t)g_comm_bounce;
  const size_t slice = count < BOUNCE ? count : BOUNCE;
  PeerBuf* w = comm_buffer(ctx, "comm.bounce", slice * sizeof(double));
  if (!w->mapped) { comm_fail(cs, "exchange through an unmapped window"); return; }
  EIGX_HIP_CHECK(hipMemcpyAsync(recv + (size_t)mine * count, send + (size_t)mine * send_stride, count * sizeof(double),
                                hipMemcpyDeviceToDevice, s));
  const CommChannel ch = CH_BULK;
  for (int k = 1; k < n; ++k) {
    const int di = (mine + k) % n, si = (mine - k + n) % n;     // group indices of this round's destination / source
    const int dq = members[di], sq = members[si];               // ... and their world ranks
    for (size_t off = 0; off < count; off += slice) {
      const size_t c = (count - off < slice) ? count - off : slice;
      const u64 epoch = ++cs->epoch[ch];
      ReadyArgs R;
      WaitArgs W;
      PushArgs A;
      R.n = W.n = A.n = 1;
      R.epoch = W.epoch = A.epoch = epoch;
      R.err = W.err = cs->err_dev;
      R.ticks = W.ticks = A.ticks = cs->ticks_dev;
      R.limit_ticks = W.limit_ticks = limit_ticks(cs);
      R.remote[0] = flag_word(cs->flags.peer[sq], ch, 1, sig_idx(cs, dq));   // "my window is free" goes to the rank that fills it
      R.local[0] = flag_word(cs->flags.local, ch, 1, dq);           // ... and I wait for the same word of my destination
      A.src = send + (size_t)di * send_stride + off; A.src_stride = 0; A.count = c; A.counter = cs->counters + ch;
      A.dst[0] = w->peer[dq];
      A.flag[0] = flag_word(cs->flags.peer[dq], ch, 0, sig_idx(cs, sq));
      W.flag[0] = flag_word(cs->flags.local, ch, 0, sq);
      hipLaunchKernelGGL(ready_kernel, dim3(1), dim3(64), 0, s, R);
      size_t blocks = (c + 4095) / 4096;
      if (blocks > 512) blocks = 512;
      hipLaunchKernelGGL(push_kernel, dim3((unsigned)blocks), dim3(256), 0, s, A);
      hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, s, W);
      size_t cb = (c + 255) / 256;
      if (cb > 2048) cb = 2048;
      hipLaunchKernelGGL(copy_sys_kernel, dim3((unsigned)cb), dim3(256), 0, s, (const double*)w->local, c,
                         recv + (size_t)si * count + off);
    }
  }
}

// ---- per-step exchanges (two windows: 0 = Y on CH_STEP, 1 = X on CH_STEPX; see eigx_comm.h) ---------------------------
static const char* step_name(int which) { return which ? "comm.stepx" : "comm.step"; }
static int step_channel(int which) { return which ? CH_STEPX : CH_STEP; }

double* comm_step_window(Context& ctx, int which, size_t msg_doubles, StepPeers* peers) {
  CommState* cs = ctx.comm;
  PeerBuf* w = comm_buffer(ctx, step_name(which), (size_t)2 * cs->P * msg_doubles * sizeof(double));
  cs->step_msg[which] = msg_doubles;
  const int ch = step_channel(which);
  peers->n = cs->P;
  peers->parity_stride = (size_t)cs->P * msg_doubles;
  peers->src_stride = msg_doubles;
  peers->counter = cs->counters + ch;
  for (int q = 0; q < EIGX_MAXP; ++q) { peers->slot[q] = nullptr; peers->flag[q] = nullptr; }
  if (cs->step_coll) {
    // collective form: the producer writes its message into a local send buffer and comm_step_allgather moves it;
    // nothing is stored into a peer by the producer kernel
    double* sendb = ctx.pool.get_t<double>(which ? "comm.stepxsend" : "comm.stepsend", msg_doubles + 8);
    peers->n = 1;
    peers->parity_stride = 0;
    peers->slot[0] = sendb;
    return w->local;
  }
  // Loopback lab mode.  X: this rank writes its message once per "peer", into the slot of each source of its own window
  // (as many stores as on a node; every source's x / W then reads as this rank's).  Y: the row and column sums go to ONE
  // destination each (the row's owner), so all sources alias one message area (src_stride 0) -- every entry a consumer
  // reads has then been written in this step, with one store as on a node.
  if (cs->loop && which == 0) peers->src_stride = 0;
  for (int q = 0; q < cs->P; ++q) {
    const size_t so = (cs->loop && which == 0) ? 0 : (size_t)sig_idx(cs, q) * msg_doubles;
    peers->slot[q] = (w->mapped ? w->peer[q] : w->local) + so;
    peers->flag[q] = flag_word(cs->flags.mapped ? cs->flags.peer[q] : cs->flags.local, ch, 0, sig_idx(cs, q));
  }
  if (!w->mapped) comm_fail(cs, "the per-step exchange needs peer windows (hipIpc) between the ranks");
  return w->local;
}

bool comm_step_collective(const Context& ctx) { return ctx.comm && ctx.comm->step_coll; }

// collective form of a per-step exchange (the north_star's literal "allgather / allreduce on RCCL", src/comm.F:1192-1247
// reduce_dbl): every rank's message of msg_doubles lands in every rank's step window at the given parity; the consumer
// adds the contributions in rank order as with the peer-write form, so the replicated results stay bit-identical.
// Enqueued on s; the consumer kernel follows in stream order (no wait kernel).
void comm_step_allgather(Context& ctx, int which, const double* sendmsg, int parity, hipStream_t s) {
  CommState* cs = ctx.comm;
  if (cs->failed) return;
  PeerBuf* w = &cs->bufs[step_name(which)];
  const size_t msg = cs->step_msg[which];
  const size_t off = (size_t)parity * cs->P * msg;
  if (cs->rccl) {
    rccl_time_begin(cs, s);
    EIGX_NCCL_TRY(cs, api.AllGather(sendmsg, w->local + off, msg, kNcclFloat64, cs->world, s));
    rccl_time_end(cs, s);
    return;
  }
  comm_exchange(ctx, COMM_WORLD, sendmsg, 0, w, off, msg, s, which ? CH_STEPX2 : CH_STEP2);
}

unsigned long long comm_step_epoch_base(Context& ctx, int which, unsigned long long nmsg) {
  CommState* cs = ctx.comm;
  const int ch = step_channel(which);
  const u64 base = cs->epoch[ch];
  cs->epoch[ch] += nmsg;
  // (an upper bound of the messages that follow; each goes to the P - 1 peers: Y in pieces -- every row sum to one
  // owner --, X whole)
  cs->st_calls[0] += (double)nmsg;
  cs->st_bytes[0] += 8.0 * (double)cs->step_msg[which] * (double)nmsg * (which ? (cs->P - 1) : 1.0);
  return base;
}

bool comm_step_wait_fused(const Context& ctx) {
  const CommState* cs = ctx.comm;
  if (!cs || cs->step_coll) return false;
  // The consumer kernels (ka_kernel, the mat-vec launch) spin on the flags in their prologue instead of running behind a
  // one-wave wait kernel: the default when every rank has its own GPU (a rank's spinning workgroups then only share the
  // card with its own side-stream kernels, which never wait for them); on a shared card (tests) and in the loopback lab
  // the wait kernel, unless EIGX_FUSE_WAIT says otherwise.
  const char* e = getenv("EIGX_FUSE_WAIT");
  if (e) return atoi(e) != 0;
  return !cs->shared_device && !cs->loop;
}
StepWait comm_step_wait_args(Context& ctx, int which, unsigned long long epoch) {
  CommState* cs = ctx.comm;
  StepWait w;
  w.flag = (const u64*)cs->flags.local + flag_index(step_channel(which), 0, 0);
  w.err = cs->err_dev; w.ticks = cs->ticks_dev; w.limit_ticks = limit_ticks(cs);
  w.epoch = epoch; w.n = cs->P; w.naps = 1;
  return w;
}

void comm_step_wait(Context& ctx, int which, unsigned long long epoch, hipStream_t s) {
  CommState* cs = ctx.comm;
  WaitArgs W;
  W.n = cs->P; W.epoch = epoch; W.err = cs->err_dev; W.ticks = cs->ticks_dev; W.limit_ticks = limit_ticks(cs);
  const int kind = (int)(epoch & 1);
  for (int q = 0; q < cs->P; ++q) W.flag[q] = flag_word(cs->flags.local, step_channel(which), kind, q);
  hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, s, W);
}

int comm_set_bounce(int doubles) {
  const int old = (int)g_comm_bounce;
  if (doubles >= 1024) g_comm_bounce = doubles;
  return old;
}

}  // namespace eigx

// RCCL self-test on one rank: exercises dlopen, ncclCommInitRank from a unique id, ncclCommSplit and the
// collectives the multi-GPU solvers use, on the library's compute stream (a one-GPU box cannot host more RCCL ranks)
extern "C" int eigx_rccl_selftest(void) {
  using namespace eigx;
  if (!g_ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (!load_rccl()) return EIGX_ERR_INTERNAL;
  EIGX_HIP_CHECK(hipSetDevice(g_ctx.device));
  ncclUniqueIdBlob id;
  if (api.GetUniqueId(&id) != 0) return EIGX_ERR_INTERNAL;
  void *comm = nullptr, *sub = nullptr;
  if (api.CommInitRank(&comm, 1, id, 0) != 0) return EIGX_ERR_INTERNAL;
  if (api.CommSplit(comm, 0, 0, &sub, nullptr) != 0 || !sub) return EIGX_ERR_INTERNAL;
  const int cnt = 1000;
  double *a = nullptr, *b = nullptr;
  EIGX_HIP_CHECK(hipMalloc(&a, cnt * 8));
  EIGX_HIP_CHECK(hipMalloc(&b, cnt * 8));
  std::vector<double> h(cnt), r(cnt);
  for (int i = 0; i < cnt; ++i) h[i] = 0.5 * i - 3.0;
  EIGX_HIP_CHECK(hipMemcpy(a, h.data(), cnt * 8, hipMemcpyHostToDevice));
  hipStream_t s = g_ctx.stream;
  int bad = 0;
  bad += api.AllReduce(a, a, cnt, kNcclFloat64, kNcclSum, sub, s) != 0;
  bad += api.AllReduce(a, a, cnt, kNcclFloat64, kNcclMax, comm, s) != 0;
  bad += api.AllGather(a, b, cnt, kNcclFloat64, sub, s) != 0;
  bad += api.GroupStart() != 0;
  bad += api.Send(b, cnt, kNcclFloat64, 0, comm, s) != 0;
  bad += api.Recv(a, cnt, kNcclFloat64, 0, comm, s) != 0;
  bad += api.GroupEnd() != 0;
  EIGX_HIP_CHECK(hipStreamSynchronize(s));
  EIGX_HIP_CHECK(hipMemcpy(r.data(), a, cnt * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < cnt; ++i) bad += (r[i] != h[i]);
  api.CommDestroy(sub);
  api.CommDestroy(comm);
  EIGX_HIP_CHECK(hipFree(a));
  EIGX_HIP_CHECK(hipFree(b));
  return bad == 0 ? EIGX_OK : EIGX_ERR_INTERNAL;
}
