// chain_overlap.hip -- lab probe for the reduction's two-kernels-per-step chain: can the launch / ramp / first-load cost
// of a dependent kernel be hidden by launching it EARLY and letting it spin on a completion counter of its producer?
//   A ("ka-like", 512 WGs): independent loads of a 2-MB panel, then needs B's output Y (reads 4 KB per WG), writes 256 B of X
//   B ("symv-like", 2080 WGs): needs A's output X, streams 32 KB per WG of a 68-MB matrix, writes 64 B of Y per WG
// modes: 0 = one stream, plain loads / stores (today's chain)
//        1 = two streams; A(k+1) is enqueued behind A(k) and spins on B(k)'s counter; B(k) waits for A(k) through an event
//        2 = two streams, no events: B(k) is enqueued behind B(k-1) and spins on A(k)'s counter as well
// In modes 1 and 2 everything that crosses kernels is stored write-through (agent scope) and loaded past L2.
// Every spin is bounded.  The chain carries a checksum so that a wrong ordering shows.
// build: hipcc --offload-arch=gfx950 -O2 -o build/chain_overlap tools/chain_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
constexpr int NA = 512, NB = 2080;

__device__ __forceinline__ double ld_ag(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ bool spin(const u64* c, u64 want, int* err) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > 200000000ll) { *err = 1; return false; }   // 2 s
  }
  return true;
}

// A: X[wg*32 + i] = sum of this WG's 512 Y values * 1e-3 + panel contribution + step
template <int MODE>
__global__ __launch_bounds__(256) void kA(const double* __restrict__ panel, const double* Y, double* X, u64* doneA, const u64* doneB,
                                          u64 step, int* err) {
  __shared__ double red[256];
  const int t = threadIdx.x, wg = blockIdx.x;
  double pv = 0.0;
  for (int j = 0; j < 2; ++j) pv += panel[(size_t)(wg * 512 + j * 256 + t)];   // predecessor-independent loads (2 MB in all)
  if (MODE >= 1 && step > 0) {
    if (t == 0) spin(doneB, (u64)NB * step, err);
    __syncthreads();
  }
  double y = 0.0;
  for (int j = 0; j < 2; ++j) { const double* q = Y + (size_t)((wg * 512 + j * 256 + t) % (NB * 8)); y += MODE ? ld_ag(q) : *q; }
  red[t] = y + 1e-9 * pv;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
  if (t < 32) { const double v = red[0] * 1e-3 + (double)step; if (MODE) st_ag(X + wg * 32 + t, v); else X[wg * 32 + t] = v; }
  if (MODE >= 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(doneA, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// B: Y[wg*8 + i] = X[(wg*8+i) % (NA*32)] + tiny contribution of the streamed tile
template <int MODE>
__global__ __launch_bounds__(256) void kB(const double* __restrict__ M, const double* X, double* Y, const u64* doneA, u64* doneB,
                                          u64 step, int* err) {
  __shared__ double red[256];
  const int t = threadIdx.x, wg = blockIdx.x;
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2* m = (const d2*)(M + (size_t)wg * 4096);
  d2 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = m[j * 256 + t];          // 32 KB per WG, predecessor-independent
  if (MODE >= 2) {
    if (t == 0) spin(doneA, (u64)NA * (step + 1), err);
    __syncthreads();
  }
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j].x + v[j].y;
  red[t] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (t < k) red[t] += red[t + k]; __syncthreads(); }
  if (t < 8) {
    const double* q = X + (wg * 8 + t) % (NA * 32);
    const double x = MODE ? ld_ag(q) : *q;
    const double o = x + 1e-12 * red[0];
    if (MODE) st_ag(Y + wg * 8 + t, o); else Y[wg * 8 + t] = o;
  }
  if (MODE >= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(doneB, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MODE>
static void run(const char* name, int steps) {
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  double *panel, *M, *X, *Y; u64* done; int* err;
  CK(hipMalloc(&panel, (size_t)NA * 512 * 8)); CK(hipMemset(panel, 0, (size_t)NA * 512 * 8));
  CK(hipMalloc(&M, (size_t)NB * 4096 * 8)); CK(hipMemset(M, 0, (size_t)NB * 4096 * 8));
  CK(hipMalloc(&X, NA * 32 * 8)); CK(hipMalloc(&Y, NB * 8 * 8)); CK(hipMalloc(&done, 16)); CK(hipMalloc(&err, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<hipEvent_t> ev(MODE == 1 ? steps : 0);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  float best = 1e30f;
  double xh = 0.0, host_us = 0.0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(X, 0, NA * 32 * 8)); CK(hipMemset(Y, 0, NB * 8 * 8)); CK(hipMemset(done, 0, 16)); CK(hipMemset(err, 0, 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa));
    const auto h0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) {
      if (MODE == 0) {
        hipLaunchKernelGGL(kA<0>, dim3(NA), dim3(256), 0, sa, panel, Y, X, done, done + 1, (u64)k, err);
        hipLaunchKernelGGL(kB<0>, dim3(NB), dim3(256), 0, sa, M, X, Y, done, done + 1, (u64)k, err);
      } else if (MODE == 1) {
        hipLaunchKernelGGL(kA<1>, dim3(NA), dim3(256), 0, sa, panel, Y, X, done, done + 1, (u64)k, err);
        CK(hipEventRecord(ev[k], sa));
        CK(hipStreamWaitEvent(sb, ev[k], 0));
        hipLaunchKernelGGL(kB<1>, dim3(NB), dim3(256), 0, sb, M, X, Y, done, done + 1, (u64)k, err);
      } else {
        hipLaunchKernelGGL(kA<2>, dim3(NA), dim3(256), 0, sa, panel, Y, X, done, done + 1, (u64)k, err);
        hipLaunchKernelGGL(kB<2>, dim3(NB), dim3(256), 0, sb, M, X, Y, done, done + 1, (u64)k, err);
      }
    }
    host_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count() * 1e6 / steps;
    if (MODE) { hipEvent_t eb; CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming)); CK(hipEventRecord(eb, sb)); CK(hipStreamWaitEvent(sa, eb, 0)); CK(hipEventDestroy(eb)); }
    CK(hipEventRecord(e1, sa));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    int eh = 0; CK(hipMemcpy(&eh, err, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&xh, X, 8, hipMemcpyDeviceToHost));
    if (eh) { printf("%-60s SPIN TIMED OUT\n", name); return; }
  }
  printf("%-60s %.2f us per step (2 kernels; host enqueue loop %.2f us per step), X[0] after %d steps = %.9f\n", name,
         1e3 * best / steps, host_us, steps, xh);
  fflush(stdout);
}

int main() {
  const int steps = 2000;
  run<0>("one stream, plain (today)", steps);
  run<1>("A pre-launched + counter; B behind an event", steps);
  run<2>("both pre-launched + counters, no events", steps);
  run<0>("one stream, plain (again)", steps);
  return 0;
}
