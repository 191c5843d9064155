// ipc_probe.hip -- feasibility probe for the device-side peer-write transport (lab tool, not product).
// P processes share ONE GPU (the situation of the multi-rank tests); each allocates a window, exports it with
// hipIpcGetMemHandle, maps the others' windows and runs `iters` rounds of
//   push kernel  : write `n` doubles into slot [parity][me] of EVERY rank's window (system-scope stores), then a flag
//   wait kernel  : one wave polls the P flags of the round (bounded spin)
//   check kernel : sums the P slots and compares with the closed form
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/ipc_probe tools/ipc_probe.hip -lrt
// run  : ipc_probe P iters n memkind(0 hipMalloc, 1 fine-grained, 2 uncached)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s failed: %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); _exit(3); } } while (0)
static int g_rank = -1;
constexpr int MAXP = 8;

struct Board {
  std::atomic<int> ready[8];
  hipIpcMemHandle_t h[MAXP];
  double usec[MAXP];
  int bad[MAXP];
};

struct Win { double* peer[MAXP]; };

// window layout (doubles): [flags: 2*MAXP u64][pad to 64][slots: 2 parities x P ranks x nmax]
__device__ __forceinline__ unsigned long long* flag_ptr(double* w, int par, int src) { return (unsigned long long*)w + par * MAXP + src; }
__device__ __forceinline__ double* slot_ptr(double* w, int par, int src, int nmax) { return w + 64 + ((size_t)par * MAXP + src) * nmax; }

__global__ void push_kernel(Win W, int P, int me, int it, int n, int nmax, unsigned* counter) {
  const int par = it & 1;
  for (int q = 0; q < P; ++q) {
    double* dst = slot_ptr(W.peer[q], par, me, nmax);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
      __hip_atomic_store(dst + i, (double)(me * 1000 + it) + 1e-3 * (i & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0) {
    __threadfence_system();
    const unsigned t = atomicAdd(counter, 1u);
    last = (t == (unsigned)(it + 1) * gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x < P)
    __hip_atomic_store(flag_ptr(W.peer[threadIdx.x], par, me), (unsigned long long)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void wait_kernel(Win W, int P, int me, int it, int* err) {
  const int par = it & 1;
  if (threadIdx.x < P) {
    unsigned long long* f = flag_ptr(W.peer[me], par, threadIdx.x);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)(it + 1)) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 300000000LL) { atomicAdd(err, 1000000); break; }   // 3 s at 100 MHz
    }
  }
}

__global__ void check_kernel(Win W, int P, int me, int it, int n, int nmax, int* err) {
  const int par = it & 1;
  int bad = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double s = 0.0, ref = 0.0;
    for (int q = 0; q < P; ++q) {
      s += __hip_atomic_load(slot_ptr(W.peer[me], par, q, nmax) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ref += (double)(q * 1000 + it) + 1e-3 * (i & 1023);
    }
    if (s != ref) ++bad;
  }
  if (bad) atomicAdd(err, bad);
}

static void barrier(Board* b, int idx, int P) {
  b->ready[idx].fetch_add(1);
  while (b->ready[idx].load() < P) usleep(100);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 200;
  const int nmax = argc > 3 ? atoi(argv[3]) : 65536;
  const int kind = argc > 4 ? atoi(argv[4]) : 1;
  Board* b = (Board*)mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset((void*)b, 0, sizeof(Board));
  pid_t pids[MAXP];
  for (int r = 0; r < P; ++r) {
    pids[r] = fork();
    if (pids[r] == 0) { g_rank = r; break; }
  }
  if (g_rank < 0) {
    int rc = 0;
    for (int r = 0; r < P; ++r) { int st = 0; waitpid(pids[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1; }
    for (int r = 0; r < P; ++r) printf("rank %d: %.2f us per round, bad %d\n", r, b->usec[r], b->bad[r]);
    printf("P=%d iters=%d n=%d kind=%d -> %s\n", P, iters, nmax, kind, rc ? "FAILED" : "OK");
    return rc;
  }
  const int me = g_rank;
  CK(hipSetDevice(0));
  const size_t bytes = (64 + (size_t)2 * MAXP * nmax) * 8;
  double* win = nullptr;
  if (kind == 0) CK(hipMalloc(&win, bytes));
  else CK(hipExtMallocWithFlags((void**)&win, bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
  CK(hipMemset(win, 0, bytes));
  CK(hipIpcGetMemHandle(&b->h[me], win));
  barrier(b, 0, P);
  Win W;
  for (int q = 0; q < MAXP; ++q) W.peer[q] = nullptr;
  for (int q = 0; q < P; ++q) {
    if (q == me) { W.peer[q] = win; continue; }
    void* p = nullptr;
    CK(hipIpcOpenMemHandle(&p, b->h[q], hipIpcMemLazyEnablePeerAccess));
    W.peer[q] = (double*)p;
  }
  barrier(b, 1, P);
  unsigned* counter; int* err;
  CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
  CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nblk = (nmax + 2047) / 2048 < 64 ? (nmax + 2047) / 2048 : 64;
  for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up (and is checked), pass 1 is timed
    barrier(b, 2 + pass, P);
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < iters; ++k) {
      const int it = pass * iters + k;
      hipLaunchKernelGGL(push_kernel, dim3(nblk), dim3(256), 0, st, W, P, me, it, nmax, nmax, counter);
      hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, st, W, P, me, it, err);
      hipLaunchKernelGGL(check_kernel, dim3(nblk), dim3(256), 0, st, W, P, me, it, nmax, nmax, err);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
  }
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  b->usec[me] = 1e3 * ms / iters;
  b->bad[me] = herr;
  barrier(b, 4, P);
  for (int q = 0; q < P; ++q) if (q != me) CK(hipIpcCloseMemHandle(W.peer[q]));
  barrier(b, 5, P);
  CK(hipFree(win));
  _exit(herr ? 2 : 0);
}
