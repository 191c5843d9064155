// ipc_big.hip -- lab probe: how large may a hipIpc-shared window be?  Two processes on GPU 0; each allocates `gb` GiB
// (kind 1 = fine-grained, 0 = hipMalloc), exports it, maps the other's and stores into its far end.  Every step is timed
// and the whole probe is bounded by alarm().
// build: hipcc --offload-arch=gfx950 -O2 -o build/ipc_big tools/ipc_big.hip -lrt ; run: build/ipc_big gb kind
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
static int g_rank = -1;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("rank %d: %s failed: %s\n", g_rank, #x, hipGetErrorString(e_)); fflush(stdout); _exit(3); } } while (0)
struct Board { std::atomic<int> ready[2]; hipIpcMemHandle_t h[2]; std::atomic<int> done[2]; };
__global__ void poke(double* p, size_t n, double v) { if (threadIdx.x == 0 && blockIdx.x == 0) { p[0] = v; p[n - 1] = v; } }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 1.0;
  const int kind = argc > 2 ? atoi(argv[2]) : 1;
  const size_t bytes = (size_t)(gb * (1ull << 30));
  Board* b = (Board*)mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset(b, 0, sizeof(Board));
  const pid_t pid = fork();
  g_rank = pid ? 0 : 1;
  alarm(60);
  CK(hipSetDevice(0));
  double t = now();
  void* p = nullptr;
  if (kind == 1) CK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained)); else CK(hipMalloc(&p, bytes));
  printf("rank %d: %.2f GiB %s alloc %.3f s\n", g_rank, gb, kind ? "fine-grained" : "hipMalloc", now() - t); fflush(stdout);
  t = now();
  CK(hipIpcGetMemHandle(&b->h[g_rank], p));
  printf("rank %d: get handle %.3f s\n", g_rank, now() - t); fflush(stdout);
  b->ready[g_rank].store(1);
  while (!b->ready[1 - g_rank].load()) usleep(100);
  t = now();
  void* q = nullptr;
  CK(hipIpcOpenMemHandle(&q, b->h[1 - g_rank], hipIpcMemLazyEnablePeerAccess));
  printf("rank %d: open handle %.3f s\n", g_rank, now() - t); fflush(stdout);
  t = now();
  hipLaunchKernelGGL(poke, dim3(1), dim3(64), 0, 0, (double*)q, bytes / 8, 1.0 + g_rank);
  CK(hipDeviceSynchronize());
  printf("rank %d: store through the mapping %.3f s\n", g_rank, now() - t); fflush(stdout);
  b->done[g_rank].store(1);
  while (!b->done[1 - g_rank].load()) usleep(100);
  double v[2];
  CK(hipMemcpy(&v[0], p, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&v[1], (char*)p + bytes - 8, 8, hipMemcpyDeviceToHost));
  printf("rank %d: my window now holds %.0f %.0f (expected %d)\n", g_rank, v[0], v[1], 2 - g_rank); fflush(stdout);
  if (pid) { int st; waitpid(pid, &st, 0); }
  return 0;
}
