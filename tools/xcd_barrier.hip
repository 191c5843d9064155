// xcd_barrier.hip -- lab tool (round 4): what does a barrier among the workgroups of ONE XCD cost?
// The tail of the reduction (active size L <= 1024-2048: a 4-16 MB triangle) pays two dependent kernel launches per step;
// a persistent kernel confined to one XCD (32 CUs, its own 4-MB L2) would replace them by two barriers.  The guide's 4-5 us
// for a grid barrier is chip-wide.  Workgroups are dealt round-robin to the 8 XCDs, so of 8*W launched workgroups the
// W with blockIdx.x % 8 == 0 sit on one XCD (checked with the XCC_ID hardware register); the others leave at once.
// Measured per iteration: (a) a counter barrier alone (agent-scope add + relaxed agent-scope poll), (b) barrier + a 2-KB
// hand-over per workgroup (write-through stores, agent-scope loads of the neighbour's block) -- twice per iteration, as a
// reduction step would need.
// build: hipcc --offload-arch=gfx950 -O2 -o build/xcd_barrier tools/xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]

struct Args {
  unsigned* counter;        // barrier counter (monotone)
  unsigned* xcc;            // [participant] XCC id seen
  double* buf;              // [participant][256] hand-over blocks
  long long* ticks;         // [participant] wall_clock64 ticks (100 MHz) of the timed loop
  int stride;               // participants = workgroups with blockIdx.x % stride == 0
  int W;                    // participants
  int iters;
  int payload;              // 0: barrier only; 1: + hand-over
};

__device__ __forceinline__ void barrier_all(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_barrier(Args A) {
  if (blockIdx.x % A.stride != 0) return;
  const int me = blockIdx.x / A.stride;
  if (me >= A.W) return;
  if (threadIdx.x == 0) A.xcc[me] = xcc_id();
  unsigned epoch = 0;
  barrier_all(A.counter, (++epoch) * A.W);       // everybody resident
  const long long t0 = wall_clock64();
  double acc = 0.0;
  for (int it = 0; it < A.iters; ++it) {
    for (int half = 0; half < 2; ++half) {
      if (A.payload) {
        // write my block write-through, drain, barrier, read the neighbour's block past L1
        __hip_atomic_store(&A.buf[(size_t)me * 256 + threadIdx.x], (double)(it + half + me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      barrier_all(A.counter, (++epoch) * A.W);
      if (A.payload) {
        const int nb = (me + 1) % A.W;
        acc += __hip_atomic_load(&A.buf[(size_t)nb * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) A.ticks[me] = t1 - t0;
  if (acc == 1.2345) A.buf[0] = acc;
}

int main() {
  Args A;
  const int maxw = 256;
  CK(hipMalloc(&A.counter, 256)); CK(hipMalloc(&A.xcc, maxw * 4)); CK(hipMalloc(&A.buf, (size_t)maxw * 256 * 8));
  CK(hipMalloc(&A.ticks, maxw * 8));
  A.iters = 2000;
  struct Case { const char* name; int stride, W; } cases[] = {
    {"one XCD, 16 workgroups", 8, 16}, {"one XCD, 32 workgroups (1 per CU)", 8, 32}, {"one XCD, 64 workgroups (2 per CU)", 8, 64},
    {"whole chip, 64 workgroups", 1, 64}, {"whole chip, 256 workgroups (1 per CU)", 1, 256}};
  for (const Case& c : cases) {
    for (int payload = 0; payload < 2; ++payload) {
      A.stride = c.stride; A.W = c.W; A.payload = payload;
      CK(hipMemset(A.counter, 0, 256)); CK(hipMemset(A.buf, 0, (size_t)maxw * 256 * 8));
      hipLaunchKernelGGL(k_barrier, dim3(c.W * c.stride), dim3(256), 0, 0, A);
      CK(hipDeviceSynchronize());
      unsigned xcc[maxw]; long long tk[maxw];
      CK(hipMemcpy(xcc, A.xcc, c.W * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(tk, A.ticks, c.W * 8, hipMemcpyDeviceToHost));
      unsigned mask = 0; long long tmax = 0;
      for (int i = 0; i < c.W; ++i) { mask |= 1u << xcc[i]; if (tk[i] > tmax) tmax = tk[i]; }
      printf("%-40s %s: %.2f us per barrier%s (XCC ids seen: mask 0x%02x)\n", c.name, payload ? "barrier + 2-KB hand-over" : "barrier alone            ",
             tmax * 0.01 / (2.0 * A.iters), payload ? " + hand-over" : "", mask);
    }
  }
  return 0;
}
