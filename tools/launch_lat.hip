// launch_lat.hip -- lab tool: cost of a chain of dependent kernel launches on one stream (the floor under the
// reduction's two kernels per step).  build: hipcc --offload-arch=gfx950 -O2 -o build/launch_lat tools/launch_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Big { double* p; int pad[100]; };
__global__ void k_empty() {}
__global__ void k_big(Big b) { if (b.pad[0] == 12345 && threadIdx.x == 0 && blockIdx.x == 0) b.p[0] = 1.0; }
__global__ void k_touch(double* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
__global__ void k_regs(double* p, int n) {   // a 200-VGPR kernel that does nothing
  double a[80];
#pragma unroll
  for (int i = 0; i < 80; ++i) a[i] = p[(threadIdx.x + i) % n];
  double s = 0;
#pragma unroll
  for (int i = 0; i < 80; ++i) s += a[i] * a[(i + 7) % 80];
  if (s == 1.2345) p[0] = s;
}
template <int N> struct Arg { double* p; int pad[N]; };
template <int N> __global__ void k_arg(Arg<N> b) { if (b.pad[N - 1] == 12345 && threadIdx.x == 0 && blockIdx.x == 0) b.p[0] = 1.0; }
template <int N> __global__ void k_arg_ptr(const Arg<N>* __restrict__ bp) { const Arg<N> b = *bp; if (b.pad[N - 1] == 12345 && threadIdx.x == 0 && blockIdx.x == 0) b.p[0] = 1.0; }
__global__ void k_vgpr(double* p) {   // claims ~210 VGPRs, does nothing else
  asm volatile("v_mov_b32 v209, 0" ::: "v209");
  if (p == nullptr) p[0] = 1.0;
}
__global__ void k_lds(double* p) {    // 32 KB of static LDS, touched once
  __shared__ double s[4096];
  s[threadIdx.x] = 1.0;
  __syncthreads();
  if (s[(threadIdx.x + 1) & 255] == 2.0) p[0] = 1.0;
}
__global__ void k_vgpr_lds_touch(double* p, int n) {   // 210 VGPRs + 32 KB LDS + one load/store per thread
  __shared__ double s[4096];
  asm volatile("v_mov_b32 v209, 0" ::: "v209");
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  s[threadIdx.x] = (i < n) ? p[i] : 0.0;
  __syncthreads();
  if (i < n) p[i] = s[(threadIdx.x + 1) & 255] + 1.0;
}
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double* p; CK(hipMalloc(&p, 1 << 24)); CK(hipMemset(p, 0, 1 << 24));
  Big b; b.p = p; for (int i = 0; i < 100; ++i) b.pad[i] = i;
  const int N = 4000;
  auto run = [&](const char* name, auto launch) {
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < N; ++i) launch();
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %.2f us per launch\n", name, 1e3 * ms / N);
  };
  run("empty, 1 WG x 64", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); });
  run("empty, 512 WG x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, st); });
  run("empty, 2080 WG x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(2080), dim3(256), 0, st); });
  run("408-byte kernarg, 512 WG x 256", [&] { hipLaunchKernelGGL(k_big, dim3(512), dim3(256), 0, st, b); });
  run("touch 128 KB (p[i] += 1), 64 WG x 256", [&] { hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, st, p, 16384); });
  run("touch 1 MB, 512 WG x 256", [&] { hipLaunchKernelGGL(k_touch, dim3(512), dim3(256), 0, st, p, 131072); });
  run("80 dependent-free loads, 512 WG x 256", [&] { hipLaunchKernelGGL(k_regs, dim3(512), dim3(256), 0, st, p, 4096); });
#define ARGRUN(N) { Arg<N> a; a.p = p; for (int i = 0; i < N; ++i) a.pad[i] = i; char nm[64]; snprintf(nm, 64, "%d-byte kernarg, 512 WG x 256", (int)sizeof(a)); \
    run(nm, [&] { hipLaunchKernelGGL(k_arg<N>, dim3(512), dim3(256), 0, st, a); }); }
  ARGRUN(2) ARGRUN(14) ARGRUN(30) ARGRUN(46) ARGRUN(62) ARGRUN(78) ARGRUN(100)
  { Arg<100> a; a.p = p; for (int i = 0; i < 100; ++i) a.pad[i] = i; Arg<100>* dp; CK(hipMalloc(&dp, sizeof(a))); CK(hipMemcpy(dp, &a, sizeof(a), hipMemcpyHostToDevice));
    run("408 bytes behind a pointer, 512 WG x 256", [&] { hipLaunchKernelGGL(k_arg_ptr<100>, dim3(512), dim3(256), 0, st, (const Arg<100>*)dp); }); }
  run("210 VGPRs, 512 WG x 256", [&] { hipLaunchKernelGGL(k_vgpr, dim3(512), dim3(256), 0, st, p); });
  run("210 VGPRs, 2080 WG x 256", [&] { hipLaunchKernelGGL(k_vgpr, dim3(2080), dim3(256), 0, st, p); });
  run("32 KB LDS, 512 WG x 256", [&] { hipLaunchKernelGGL(k_lds, dim3(512), dim3(256), 0, st, p); });
  run("210 VGPRs + 32 KB LDS + touch 1 MB, 512 WG", [&] { hipLaunchKernelGGL(k_vgpr_lds_touch, dim3(512), dim3(256), 0, st, p, 131072); });
  run("alternating empty / touch 1 MB", [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, st); hipLaunchKernelGGL(k_touch, dim3(512), dim3(256), 0, st, p, 131072); });
  return 0;
}
