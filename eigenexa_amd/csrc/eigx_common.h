// eigx_common.h -- shared declarations for the MI355X (gfx950) EigenExa hot-path library.
// Internal header: the public C-ABI is include/eigenexa_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <ctime>

#define EIGX_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      fprintf(stderr, "[eigx] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e),    \
              __FILE__, __LINE__, hipGetErrorString(_e));                           \
      abort();                                                                      \
    }                                                                               \
  } while (0)

namespace eigx {

typedef double d4_t __attribute__((ext_vector_type(4)));

// Process-grid position of this rank in the 2-D cyclic layout
// (reference: src/eigen_libs0.F:526-570 grid rule, :1825-2258 index maps).
// Global row g (0-based) lives on x-rank g % Px at local row g / Px; same for columns with Py.
struct Grid {
  int Px = 1, Py = 1;  // grid shape (x: rows, y: columns)
  int px = 0, py = 0;  // my coordinates
  int rank = 0, nranks = 1;
  int row_major = 0;   // rank order 'R' (src/eigen_libs0.F:2336-2356)
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Leading dimension >= n that is an odd multiple of 32 doubles (256 B): column strides that are multiples of
// large powers of two camp on a few HBM channels (measured: fp64 GEMM 60 -> 45 TFLOP/s at ld = 8192).  Same
// rule as the reference's CSTAB_get_optdim (src/CSTAB.F:73-131) for a different memory system.
// Leading dimensions (in doubles) of column-major device arrays that are streamed from HBM.
// Measured on MI355X (tools/ld_scan.sh, profiles/r03_ld_scan.log): what matters is the byte shift between consecutive
// columns modulo 16 KiB.  A shift of a few hundred bytes (the old "odd multiple of 32 doubles" rule: 256 B) keeps the
// columns a wave walks through on the same few memory channels; 2 - 12 KiB spreads them.  N = 8192 band reduction:
// ld 8224 -> 131.9 ms, 8256 -> 135.5, 8288 -> 129.6, 8448 -> 127.1, 8704 -> 126.5, 9216 -> 126.9 (reproducible to 0.2 ms).
// Rule: from 2048 doubles on, ld mod 2048 in [512, 1536] (shift 4 - 12 KiB); below that the arrays live in L2 / MALL.
static inline int hbm_ld(int l) {   // l: multiple of 32
  if (l < 2048) return l;
  const int m = l % 2048;
  if (m < 512) return l + (512 - m);
  if (m > 1536) return l + (2048 - m) + 512;
  return l;
}
static inline int pad_ld(int n) {
  static const int extra = [] { const char* e = getenv("EIGX_LD_EXTRA"); return e ? atoi(e) : 0; }();   // lab switch (doubles)
  int l = (n + 31) / 32;
  if (l * 32 >= 2048) return hbm_ld(l * 32) + extra;
  if ((l & 1) == 0) ++l;
  return l * 32 + extra;
}
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// number of local indices l with global index l*P+p < n
static inline int local_count(int n, int P, int p) { return n > p ? (n - p + P - 1) / P : 0; }
// ScaLAPACK NUMROC with source process 0: local extent of n indices dealt in blocks of nb to process p of P
// (nb = 1 is the cyclic layout of the EigenExa API: numroc(n, 1, p, P) == local_count(n, P, p))
static inline int numroc(int n, int nb, int p, int P) {
  const int nblocks = n / nb;
  int cnt = (nblocks / P) * nb;
  const int extra = nblocks % P;
  if (p < extra) cnt += nb;
  else if (p == extra) cnt += n % nb;
  return cnt;
}

// ---- kernels / launchers (device pointers, column-major, all on `stream`) -------------------

// C = alpha*op(A)*op(B) + beta*C ; opA/opB in {'N','T'}.
// tri_mode: 0 = full, 1 = only tiles that intersect the upper triangle (global row <= global col)
// of a matrix whose local element (i,j) is global (i*Px+px, j*Py+py), square tile grid, block-triangular order;
// 2 = the same test on the rectangular local block of a 2-D cyclic distribution (full-mode tile order).
// batch > 1: `batch` independent products, operand b at A + b*strideA etc. (elements).
// kmapA (device, optional, opA='N'): A's column for k-index k; cmapC (device, optional): C's column for n.
// tri mode: ownP/ownp = multi-GPU ownership of 128-column tile columns; [tn_lo, tn_hi) restricts the launch to a
// window of tile columns (the look-ahead split of the trailing update).
void dgemm_dev(hipStream_t stream, char opA, char opB, int M, int N, int K, double alpha,
               const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc,
               int tri_mode = 0, const Grid* g = nullptr, const int* kmapA = nullptr,
               const int* cmapC = nullptr, int batch = 1, long strideA = 0, long strideB = 0, long strideC = 0,
               int batch2 = 1, long strideA2 = 0, long strideB2 = 0, long strideC2 = 0, int ownP = 1,
               int ownp = 0, const int* kmapB = nullptr, int tn_lo = 0, int tn_hi = 0x7fffffff);

// Batch of independent gather products C_b = A(:, kmapA_b) * B(:, kmapB_b)^T with their own sizes (the merges of one D&C
// height): ONE launch, the table lives in device memory (blockIdx.y = entry); offsets are in elements from the bases.
struct GemmBatch {
  int M, N, K, pad;
  long offA, offB, offC;
  int offKA, offKB;
};
void dgemm_gather_batch_dev(hipStream_t stream, const GemmBatch* tab_dev, int nbatch, int maxM, int maxN,
                            const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                            const int* kmapA, const int* kmapB);

// EIGX_TRACE_STAGES=1: one stderr line per stage boundary of a solve (rank, wall clock), for locating a stall on a
// process grid; costs one getenv per process
inline void stage_trace(int rank, const char* what, long detail = -1) {
  static const bool on = getenv("EIGX_TRACE_STAGES") != nullptr;
  if (!on) return;
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  if (detail >= 0) fprintf(stderr, "[eigx stage] rank %d t=%.3f %s %ld\n", rank, ts.tv_sec % 100000 + ts.tv_nsec * 1e-9, what, detail);
  else fprintf(stderr, "[eigx stage] rank %d t=%.3f %s\n", rank, ts.tv_sec % 100000 + ts.tv_nsec * 1e-9, what);
  fflush(stderr);
}

// tuning hook (eigx_tune key 0): 2 = LDS-DMA ring GEMM where supported, 1 = register-staged GEMM only
int set_gemm_variant(int v);
// tuning hook (eigx_tune key 6): 1 = the trailing update streams its C tiles with non-temporal loads / stores
int set_gemm_cstream(int v);
// tuning hook (eigx_tune key 1): target number of concurrent Sturm sweeps of the bisection
int set_bisect_threads(int v);
// tuning hook (eigx_tune key 2): super-block factor of the back-transformation (0 = automatic, 1, 2, 4)
int set_bt_q(int v);
// tuning hook (eigx_tune keys 3, 4): largest L that uses the 128 / 256 SYMV tile
int set_symv_threshold(int which, int v);
// tuning hook (eigx_tune key 8): chunk width (roots) of the multi-rank D&C's eigenvector-row buffer, 64 .. 2048
int set_dc_chunk(int v);
// lab switches (eigx_tune keys 15, 16): pipelined D&C passes / one product launch per low height (one GPU)
int set_dc_pipe(int v);
int set_dc_batch(int v);

}  // namespace eigx
