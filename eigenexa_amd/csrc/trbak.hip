// trbak.hip -- Householder back-transformation Z <- H_n ... H_{1+band} Z in compact-WY blocks, gfx950.
//
// Replaces eigen_common_trbakwy / eigen_trbakwy_body / eigen_trbakwy_block_body{,1,2}
// (src/trbakwy4.F:77-819, src/trbakwy4_body.F:107-741).
//
// Same block reflector as the reference: for a block of reflectors u_{j0..j1} (applied j0 first)
//   H_{j1} ... H_{j0} = I - V S^{-1} V^T,  S lower triangular, S_jj = beta_j, S_jk = u_j^T u_k (j > k)
//   (src/trbakwy4_body.F:573-577, :305-313, :687), beta_j = -a(j-band, j) * e(j, band)
//   (src/trbakwy4.F:309-335).
// MI355X re-design (one stream, no panel broadcast / triple buffering needed on one GPU):
//   1. mask-copy the block's reflectors out of `a` into a zero-padded panel V (rows x mb)
//   2. Gram partials G_c = V_c^T V_c over 512-row chunks: one batched fp64 MFMA GEMM launch
//   3. T = S^{-1} by the row recurrence T(k,:) = -(1/beta_k) G(k,0:k) T(0:k,:) in LDS (packed lower
//      triangles of G and T, one workgroup) -- replaces the DTRSM of the reference
//   4. W = V^T Z,  X = T W,  Z -= V X : three fp64 MFMA GEMMs (the reference's dgemm('T','N') +
//      dtrsm + dgemm('N','N'))
#include "eigx_context.h"
#include "../../include/eigenexa_amd.h"

namespace eigx {

namespace {

constexpr int GCH = 512;  // rows per Gram chunk

// V(r, c) = a(r, j0+c) for r <= j0+c-band, else 0 ; rows [0, rows_pad)
__global__ void maskcopy_kernel(const double* __restrict__ A, int lda, int j0, int mbk, int band, int rows_pad,
                                double* __restrict__ V, int ldv) {
  const int c = blockIdx.y;
  if (c >= mbk) return;
  const int len = j0 + c - band + 1;
  const double* src = A + (size_t)(j0 + c) * lda;
  double* dst = V + (size_t)c * ldv;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows_pad; r += gridDim.x * blockDim.x)
    dst[r] = (r < len) ? src[r] : 0.0;
}

__device__ __forceinline__ int tri_idx(int r, int c) { return r * (r + 1) / 2 + c; }  // c <= r

// T = S^{-1}, S = strict_lower(G) + diag(beta); Gpart: [nchunks][mb x mb] column-major partial Grams
__global__ __launch_bounds__(256) void tbuild_kernel(const double* __restrict__ Gpart, int nchunks, int mb, int mbk,
                                                     const double* __restrict__ A, int lda,
                                                     const double* __restrict__ e, int lde, int band, int j0,
                                                     double* __restrict__ T) {
  extern __shared__ double sm[];  // Gl[mb(mb+1)/2] | Tl[mb(mb+1)/2] | binv[mb]
  const int tri = mb * (mb + 1) / 2;
  double* Gl = sm;
  double* Tl = sm + tri;
  double* binv = sm + 2 * tri;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < mbk * mbk; idx += 256) {
    const int r = idx % mbk, c = idx / mbk;
    if (c < r) {
      double v = 0.0;
      for (int q = 0; q < nchunks; ++q) v += Gpart[(size_t)q * mb * mb + (size_t)c * mb + r];
      Gl[tri_idx(r, c)] = v;
    }
  }
  for (int k = tid; k < mbk; k += 256) {
    const int j = j0 + k;
    const double beta = -A[(size_t)j * lda + (j - band)] * e[(size_t)(band - 1) * lde + j];
    binv[k] = (beta != 0.0) ? 1.0 / beta : 1.0;  // beta == 0 <=> u_j == 0: the row of G is zero too
  }
  __syncthreads();
  for (int k = 0; k < mbk; ++k) {
    const double bk = binv[k];
    for (int c = tid; c < k; c += 256) {
      double acc = 0.0;
      for (int l = c; l < k; ++l) acc += Gl[tri_idx(k, l)] * Tl[tri_idx(l, c)];
      Tl[tri_idx(k, c)] = -bk * acc;
    }
    if (tid == 0) Tl[tri_idx(k, k)] = bk;
    __syncthreads();
  }
  for (int idx = tid; idx < mb * mb; idx += 256) {
    const int r = idx % mb, c = idx / mb;
    T[idx] = (r < mbk && c <= r) ? Tl[tri_idx(r, c)] : 0.0;
  }
}

}  // namespace

void trbak_dev(Context& ctx, int n, int nvec, const double* A, int lda, double* Z, int ldz, const double* e,
               int lde, int mb, int band) {
  if (nvec <= 0 || n <= band) return;
  hipStream_t st = ctx.stream;
  if (mb < 8) mb = 8;
  if (mb > 128) mb = 128;  // T-builder keeps two packed mb x mb triangles in LDS
  const int rows_max = n;
  const int ldv = (rows_max + GCH - 1) / GCH * GCH;
  const int maxchunks = ldv / GCH;
  double* V = ctx.pool.get_t<double>("bt.V", (size_t)ldv * mb);
  double* Gpart = ctx.pool.get_t<double>("bt.G", (size_t)maxchunks * mb * mb);
  double* T = ctx.pool.get_t<double>("bt.T", (size_t)mb * mb);
  double* W = ctx.pool.get_t<double>("bt.W", (size_t)mb * nvec);
  double* X = ctx.pool.get_t<double>("bt.X", (size_t)mb * nvec);
  const size_t tshm = ((size_t)mb * (mb + 1) + mb) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    EIGX_HIP_CHECK(hipFuncSetAttribute((const void*)tbuild_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(((size_t)128 * 129 + 128) * sizeof(double))));
    attr = true;
  }
  for (int j0 = band; j0 < n; j0 += mb) {
    const int mbk = (n - j0 < mb) ? n - j0 : mb;
    const int rows = j0 + mbk - 1 - band + 1;  // length of the longest reflector of the block
    const int nchunks = (rows + GCH - 1) / GCH;
    const int rows_pad = nchunks * GCH;
    hipLaunchKernelGGL(maskcopy_kernel, dim3((rows_pad + 255) / 256 > 64 ? 64 : (rows_pad + 255) / 256, mbk),
                       dim3(256), 0, st, A, lda, j0, mbk, band, rows_pad, V, ldv);
    // Gram partials: batch of nchunks products (mbk x mbk, K = GCH)
    dgemm_dev(st, 'T', 'N', mbk, mbk, GCH, 1.0, V, ldv, V, ldv, 0.0, Gpart, mb, 0, nullptr, nullptr, nullptr,
              nchunks, GCH, GCH, (long)mb * mb);
    hipLaunchKernelGGL(tbuild_kernel, dim3(1), dim3(256), tshm, st, Gpart, nchunks, mb, mbk, A, lda, e, lde, band,
                       j0, T);
    dgemm_dev(st, 'T', 'N', mbk, nvec, rows, 1.0, V, ldv, Z, ldz, 0.0, W, mb);
    dgemm_dev(st, 'N', 'N', mbk, nvec, mbk, 1.0, T, mb, W, mb, 0.0, X, mb);
    dgemm_dev(st, 'N', 'N', rows, nvec, mbk, -1.0, V, ldv, X, mb, 1.0, Z, ldz);
  }
  EIGX_HIP_CHECK(hipGetLastError());
}

}  // namespace eigx
