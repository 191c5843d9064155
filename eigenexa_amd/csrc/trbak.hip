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
//   1. zero `a` below every reflector: `a` itself is then the zero-padded panel V of every block
//   2. Gram partials G_c = V_c^T V_c over 512-row chunks of ALL blocks: one two-level batched fp64 MFMA
//      GEMM launch
//   3. T = S^{-1} of ALL blocks in one launch (one workgroup per block): row recurrence
//      T(k,:) = -(1/beta_k) G(k,0:k) T(0:k,:) on packed lower triangles in LDS -- replaces the DTRSM
//   4. W = V^T Z,  X = T W,  Z -= V X : three fp64 MFMA GEMMs (the reference's dgemm('T','N') +
//      dtrsm + dgemm('N','N'))
#include "eigx_context.h"
#include "../../include/eigenexa_amd.h"

namespace eigx {

namespace {

constexpr int GCH = 512;  // rows per Gram chunk

// After the reduction the part of column j below its reflector (rows > j-band) holds stale matrix / band
// entries that nobody reads again (d, e were extracted; `a` is destroyed by contract, src/eigen_sx.F:30-308).
// Zeroing it turns `a` itself into the zero-padded reflector panel V of every block: no copies.
__global__ void zero_below_kernel(double* __restrict__ A, int lda, int n, int band, int rows_pad) {
  const int j = blockIdx.y;
  const int len = j - band + 1;  // reflector length (<= 0: no reflector in this column)
  double* col = A + (size_t)j * lda;
  const int lo = len > 0 ? len : 0;
  const int hi = rows_pad < lda ? rows_pad : lda;
  for (int r = lo + blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += gridDim.x * blockDim.x) col[r] = 0.0;
}

__device__ __forceinline__ int tri_idx(int r, int c) { return r * (r + 1) / 2 + c; }  // c <= r

// T = S^{-1}, S = strict_lower(G) + diag(beta); Gpart: [nchunks][mb x mb] column-major partial Grams
__global__ __launch_bounds__(256) void tbuild_kernel(const double* __restrict__ Gall, int maxchunks, int mb, int n,
                                                     const double* __restrict__ A, int lda,
                                                     const double* __restrict__ e, int lde, int band,
                                                     double* __restrict__ Tall) {
  // block b = blockIdx.x: reflectors j0 .. j0+mbk-1
  const int j0 = band + blockIdx.x * mb;
  const int mbk = (n - j0 < mb) ? n - j0 : mb;
  const int rows = j0 + mbk - band;
  const int nchunks = (rows + GCH - 1) / GCH;
  const double* Gpart = Gall + (size_t)blockIdx.x * maxchunks * mb * mb;
  double* T = Tall + (size_t)blockIdx.x * mb * mb;
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

void trbak_dev(Context& ctx, int n, int nvec, double* A, int lda, double* Z, int ldz, const double* e,
               int lde, int mb, int band) {
  if (nvec <= 0 || n <= band) return;
  hipStream_t st = ctx.stream;
  if (mb < 8) mb = 8;
  if (mb > 128) mb = 128;  // T-builder keeps two packed mb x mb triangles in LDS
  const int nblk = (n - band + mb - 1) / mb;
  const int rows_all = n - band;                       // longest reflector
  int rows_pad = (rows_all + GCH - 1) / GCH * GCH;     // Gram chunks read up to here: must stay inside lda
  const int maxchunks = (rows_all + GCH - 1) / GCH;
  const bool inplace = rows_pad <= lda;
  double* V = A;
  int ldv = lda;
  if (!inplace) {
    // lda too small for the zero padding of the last Gram chunk: work on a padded copy of the reflectors
    ldv = pad_ld(rows_pad);
    V = ctx.pool.get_t<double>("bt.Vall", (size_t)ldv * n);
    EIGX_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)ldv * 8, A, (size_t)lda * 8, (size_t)n * 8, (size_t)n,
                                    hipMemcpyDeviceToDevice, st));
  }
  hipLaunchKernelGGL(zero_below_kernel, dim3(8, n), dim3(256), 0, st, V, ldv, n, band, rows_pad);
  double* Gall = ctx.pool.get_t<double>("bt.G", (size_t)nblk * maxchunks * mb * mb);
  double* Tall = ctx.pool.get_t<double>("bt.T", (size_t)nblk * mb * mb);
  double* W = ctx.pool.get_t<double>("bt.W", (size_t)mb * nvec);
  double* X = ctx.pool.get_t<double>("bt.X", (size_t)mb * nvec);
  const size_t tshm = ((size_t)mb * (mb + 1) + mb) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    EIGX_HIP_CHECK(hipFuncSetAttribute((const void*)tbuild_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(((size_t)128 * 129 + 128) * sizeof(double))));
    attr = true;
  }
  // Gram partials of every block in one launch: batch = chunk (stride GCH rows), batch2 = block
  // (stride mb columns); chunks beyond a block's reflector length multiply zeros.
  const int nfull = (n - band) / mb;  // full blocks; a trailing partial block gets its own launch
  if (nfull > 0)
    dgemm_dev(st, 'T', 'N', mb, mb, GCH, 1.0, V + (size_t)band * ldv, ldv, V + (size_t)band * ldv, ldv, 0.0, Gall, mb,
              0, nullptr, nullptr, nullptr, maxchunks, GCH, GCH, (long)mb * mb, nfull, (long)mb * ldv, (long)mb * ldv,
              (long)maxchunks * mb * mb);
  if (nfull < nblk) {
    const int j0 = band + nfull * mb, mbk = n - j0;
    dgemm_dev(st, 'T', 'N', mbk, mbk, GCH, 1.0, V + (size_t)j0 * ldv, ldv, V + (size_t)j0 * ldv, ldv, 0.0,
              Gall + (size_t)nfull * maxchunks * mb * mb, mb, 0, nullptr, nullptr, nullptr, maxchunks, GCH, GCH,
              (long)mb * mb);
  }
  hipLaunchKernelGGL(tbuild_kernel, dim3(nblk), dim3(256), tshm, st, Gall, maxchunks, mb, n, V, ldv, e, lde, band,
                     Tall);
  for (int b = 0; b < nblk; ++b) {
    const int j0 = band + b * mb;
    const int mbk = (n - j0 < mb) ? n - j0 : mb;
    const int rows = j0 + mbk - band;  // length of the longest reflector of the block
    const double* Vb = V + (size_t)j0 * ldv;
    const double* T = Tall + (size_t)b * mb * mb;
    dgemm_dev(st, 'T', 'N', mbk, nvec, rows, 1.0, Vb, ldv, Z, ldz, 0.0, W, mb);
    dgemm_dev(st, 'N', 'N', mbk, nvec, mbk, 1.0, T, mb, W, mb, 0.0, X, mb);
    dgemm_dev(st, 'N', 'N', rows, nvec, mbk, -1.0, Vb, ldv, X, mb, 1.0, Z, ldz);
  }
  EIGX_HIP_CHECK(hipGetLastError());
}

}  // namespace eigx
