// read_bw.hip -- lab tool: ceiling of a read-only HBM stream on this part (what the fused SYMV can hope for).
// build: hipcc --offload-arch=gfx950 -O3 -o build/read_bw tools/read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
// every wave streams contiguous 1-KiB rows (64 lanes x 16 B), UNR independent loads in flight per lane
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void rd(const d2* __restrict__ p, size_t n2, double* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  for (; i + (UNR - 1) * stride < n2; i += UNR * stride) {
    d2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) s += v[u].x + v[u].y;
  }
  if (s == 1.23456789) out[0] = s;
}
// SYMV-like: a wave reads 8 columns x 1 KiB with a column stride of ld doubles (tile = 128 rows x 128 cols per workgroup)
template <bool NT>
__global__ __launch_bounds__(256) void rd_tiles(const double* __restrict__ A, int ld, int nt, double* out) {
  const int ty = blockIdx.x / nt, tx = blockIdx.x % nt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* base = A + (size_t)(tx * 128 + wave * 32) * ld + ty * 128 + lane * 2;
  double s = 0.0;
#pragma unroll 1
  for (int g = 0; g < 4; ++g) {
    d2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const d2* q = (const d2*)(base + (size_t)(g * 8 + j) * ld);
      v[j] = NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j].x + v[j].y;
  }
  if (s == 1.23456789) out[0] = s;
}
int main() {
  const size_t bytes = (size_t)8 << 30;   // 8 GiB: far beyond L2 + Infinity Cache
  d2* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes));
  double* out; CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch, double nbytes) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %.0f GB/s\n", name, 3 * nbytes / (ms * 1e-3) / 1e9);
  };
  const size_t n2 = bytes / 16;
  for (int wg : {2048, 4096, 8192, 16384})
    for (int nt = 0; nt < 2; ++nt) {
      char nm[96];
      snprintf(nm, 96, "linear read, %d WGs, 8 loads in flight%s", wg, nt ? ", nontemporal" : "");
      if (nt) time(nm, [&] { hipLaunchKernelGGL((rd<8, true>), dim3(wg), dim3(256), 0, 0, p, n2, out); }, (double)bytes);
      else time(nm, [&] { hipLaunchKernelGGL((rd<8, false>), dim3(wg), dim3(256), 0, 0, p, n2, out); }, (double)bytes);
    }
  time("linear read, 8192 WGs, 16 loads in flight, nt", [&] { hipLaunchKernelGGL((rd<16, true>), dim3(8192), dim3(256), 0, 0, p, n2, out); }, (double)bytes);
  time("linear read, 8192 WGs, 4 loads in flight, nt", [&] { hipLaunchKernelGGL((rd<4, true>), dim3(8192), dim3(256), 0, 0, p, n2, out); }, (double)bytes);
  // Infinity-Cache-resident sizes (the upper triangle of N = 8192 is 268 MB and shrinks): repeated reads of one buffer
  for (size_t mb : {64, 128, 192, 256, 512}) {
    char nm[96];
    snprintf(nm, 96, "linear read of the same %zu MB, 8192 WGs, repeated", mb);
    const size_t m2 = (mb << 20) / 16;
    time(nm, [&] { hipLaunchKernelGGL((rd<8, false>), dim3(8192), dim3(256), 0, 0, p, m2, out); }, (double)(mb << 20));
    snprintf(nm, 96, "linear read of the same %zu MB, 2048 WGs, repeated", mb);
    time(nm, [&] { hipLaunchKernelGGL((rd<8, false>), dim3(2048), dim3(256), 0, 0, p, m2, out); }, (double)(mb << 20));
  }
  {
    const int n = 8192 - 1024, ld = 8192 + 34, nt = n / 128;   // N = 8192-sized matrix (411 MB square, repeated)
    time("128x128 tiles of a 7168^2 block, ld=8226, repeated", [&] { hipLaunchKernelGGL((rd_tiles<false>), dim3(nt * nt), dim3(256), 0, 0, (const double*)p, ld, nt, out); }, 8.0 * n * n);
    const int n2 = 4096, nt2 = n2 / 128;                        // 134 MB square
    time("128x128 tiles of a 4096^2 block, ld=8226, repeated", [&] { hipLaunchKernelGGL((rd_tiles<false>), dim3(nt2 * nt2), dim3(256), 0, 0, (const double*)p, ld, nt2, out); }, 8.0 * n2 * n2);
  }
  {
    const int n = 32768 - 2048, ld = 32768 + 34, nt = n / 128;   // column-major matrix, SYMV-like tiles
    time("128x128 tiles, 8 cols x 1 KiB per wave, ld=32802", [&] { hipLaunchKernelGGL((rd_tiles<false>), dim3(nt * nt), dim3(256), 0, 0, (const double*)p, ld, nt, out); }, 8.0 * n * n);
    time("128x128 tiles, 8 cols x 1 KiB per wave, ld=32802, nt", [&] { hipLaunchKernelGGL((rd_tiles<true>), dim3(nt * nt), dim3(256), 0, 0, (const double*)p, ld, nt, out); }, 8.0 * n * n);
  }
  return 0;
}
