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
#include "eigx_comm.h"
#include "../../include/eigenexa_amd.h"

namespace eigx {

namespace {

constexpr int GCH = 512;  // rows per Gram chunk
int g_bt_q = 0;           // tuning hook (eigx_tune key 2): force the super-block factor (0 = automatic)

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

// T = S^{-1}, S = strict_lower(G) + diag(beta), for one 128-column sub-block of a super-block of mbe columns.
// Gall: [super-block][chunk][mbe x mbe] column-major partial Grams; the sub-block's diagonal block sits at
// (sub*mb, sub*mb) of every partial.  Tall: [super-block][mbe x mbe]; only the sub-block's diagonal block is
// written here (the blocks below the diagonal follow from two small GEMMs per level, see trbak_dev).
__global__ __launch_bounds__(256) void tbuild_kernel(const double* __restrict__ Gall, int maxchunks, int gch, int mb,
                                                     int mbe, int nsub, int n, const double* __restrict__ A, int lda,
                                                     const double* __restrict__ e, int lde, int band,
                                                     double* __restrict__ Tall, int jstart) {
  const int sb = blockIdx.x / nsub, sub = blockIdx.x % nsub;
  const int j0 = jstart + sb * mbe + sub * mb;    // first reflector of this sub-block
  if (j0 >= n) return;
  const int mbk = (n - j0 < mb) ? n - j0 : mb;
  const int rows = j0 + mbk - band;
  const int nchunks = (rows + gch - 1) / gch;
  const double* Gpart = Gall + (size_t)sb * maxchunks * mbe * mbe + (size_t)(sub * mb) * mbe + sub * mb;
  double* T = Tall + (size_t)sb * mbe * mbe + (size_t)(sub * mb) * mbe + sub * mb;
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
      for (int q = 0; q < nchunks; ++q) v += Gpart[(size_t)q * mbe * mbe + (size_t)c * mbe + r];
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
    T[(size_t)c * mbe + r] = (r < mbk && c <= r) ? Tl[tri_idx(r, c)] : 0.0;
  }
}

// Gs[super-block] = sum over the row chunks of the partial Grams (dense mbe x mbe, for the off-diagonal T blocks)
__global__ void gsum_kernel(const double* __restrict__ Gall, int maxchunks, int mbe, double* __restrict__ Gs) {
  const int sb = blockIdx.y;
  const size_t sz = (size_t)mbe * mbe;
  const double* src = Gall + (size_t)sb * maxchunks * sz;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sz; i += (size_t)gridDim.x * blockDim.x) {
    double v = 0.0;
    for (int q = 0; q < maxchunks; ++q) v += src[(size_t)q * sz + i];
    Gs[(size_t)sb * sz + i] = v;
  }
}

}  // namespace

// reflectors [jstart, jend) in blocks of mbe = q * mb columns (q = 1: any range, the last block may be short;
// q > 1: jend - jstart must be a multiple of mbe).  T of a super-block is assembled from the 128-column
// diagonal blocks: S = [[S11, 0], [S21, S22]]  =>  S^-1 = [[T11, 0], [-T22 S21 T11, T22]], S21 = V2^T V1.
// bt_prepare_range builds the T factors (they depend on the reflectors only, so trbak_prepare_dev runs this on the
// side stream while the divide and conquer occupies the compute stream); bt_apply_range applies them to Z.
struct BtRange { int mbe, nblk, gch, maxchunks; size_t msz; double* Tall; };

static BtRange bt_range_geom(Context& ctx, int band, int jstart, int jend, int mb, int q) {
  BtRange g;
  g.mbe = mb * q;
  g.nblk = (jend - jstart + g.mbe - 1) / g.mbe;
  const int rows_all = jend - band;                    // longest reflector of the range
  // Gram row chunks: 512 rows for plain blocks; ~4096 rows (even, as few padding rows as possible) for
  // super-blocks, whose partial Grams are mbe x mbe each
  g.gch = GCH; g.maxchunks = (rows_all + GCH - 1) / GCH;
  if (q > 1) {
    g.maxchunks = (rows_all + 4095) / 4096;
    g.gch = ((rows_all + g.maxchunks - 1) / g.maxchunks + 1) & ~1;
  }
  g.msz = (size_t)g.mbe * g.mbe;
  g.Tall = ctx.pool.get_t<double>(q > 1 ? "bt.Tq" : "bt.T", (size_t)g.nblk * g.msz);
  return g;
}

static void bt_prepare_range(Context& ctx, hipStream_t st, int n, double* V, int ldv, const double* e, int lde, int band,
                             int jstart, int jend, int mb, int q) {
  if (jend <= jstart) return;
  const BtRange g = bt_range_geom(ctx, band, jstart, jend, mb, q);
  const int mbe = g.mbe, nblk = g.nblk, gch = g.gch, maxchunks = g.maxchunks;
  const size_t msz = g.msz;
  double* Tall = g.Tall;
  double* Gall = ctx.pool.get_t<double>(q > 1 ? "bt.Gq" : "bt.G", (size_t)nblk * maxchunks * msz);
  const size_t tshm = ((size_t)mb * (mb + 1) + mb) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    EIGX_HIP_CHECK(hipFuncSetAttribute((const void*)tbuild_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(((size_t)128 * 129 + 128) * sizeof(double))));
    attr = true;
  }
  // Gram partials of every block in one launch: batch = chunk (stride gch rows), batch2 = block (stride mbe
  // columns); chunks beyond a block's reflector length multiply zeros.
  const int nfull = (jend - jstart) / mbe;  // full blocks; a trailing partial block (q = 1 only) gets its own launch
  double* Vr = V + (size_t)jstart * ldv;
  if (nfull > 0)
    dgemm_dev(st, 'T', 'N', mbe, mbe, gch, 1.0, Vr, ldv, Vr, ldv, 0.0, Gall, mbe, 0, nullptr, nullptr, nullptr,
              maxchunks, gch, gch, (long)msz, nfull, (long)mbe * ldv, (long)mbe * ldv, (long)maxchunks * msz);
  if (nfull < nblk) {
    const int j0 = jstart + nfull * mbe, mbk = jend - j0;
    dgemm_dev(st, 'T', 'N', mbk, mbk, gch, 1.0, V + (size_t)j0 * ldv, ldv, V + (size_t)j0 * ldv, ldv, 0.0,
              Gall + (size_t)nfull * maxchunks * msz, mbe, 0, nullptr, nullptr, nullptr, maxchunks, gch, gch,
              (long)msz);
  }
  if (q > 1) EIGX_HIP_CHECK(hipMemsetAsync(Tall, 0, (size_t)nblk * msz * sizeof(double), st));
  // tbuild addresses reflector j0 = band + sb*mbe + sub*mb: shift the matrix view so that "band" means jstart
  hipLaunchKernelGGL(tbuild_kernel, dim3(nblk * q), dim3(256), tshm, st, Gall, maxchunks, gch, mb, mbe, q, n, V, ldv, e,
                     lde, band, Tall, jstart);
  if (q > 1) {
    double* Gs = ctx.pool.get_t<double>("bt.Gs", (size_t)nblk * msz);
    double* Y = ctx.pool.get_t<double>("bt.Y", (size_t)nblk * msz);
    hipLaunchKernelGGL(gsum_kernel, dim3(64, nblk), dim3(256), 0, st, Gall, maxchunks, mbe, Gs);
    // level by level: diagonal blocks of size h are done; the block below-left of each pair follows
    for (int h = mb; h < mbe; h *= 2) {
      const int np = mbe / (2 * h);                 // pairs per super-block
      const long dstep = (long)2 * h * (mbe + 1);   // from one pair to the next along the diagonal
      // Y = G21 * T11   ;   T21 = -T22 * Y
      dgemm_dev(st, 'N', 'N', h, h, h, 1.0, Gs + h, mbe, Tall, mbe, 0.0, Y, h, 0, nullptr, nullptr, nullptr, np, dstep,
                dstep, (long)h * h, nblk, (long)msz, (long)msz, (long)np * h * h);
      dgemm_dev(st, 'N', 'N', h, h, h, -1.0, Tall + (size_t)h * (mbe + 1), mbe, Y, h, 0.0, Tall + h, mbe, 0, nullptr,
                nullptr, nullptr, np, dstep, (long)h * h, dstep, nblk, (long)msz, (long)np * h * h, (long)msz);
    }
  }
}

static void bt_apply_range(Context& ctx, int nvec, const double* V, int ldv, double* Z, int ldz, int band, int jstart,
                           int jend, int mb, int q) {
  if (jend <= jstart) return;
  hipStream_t st = ctx.stream;
  const BtRange g = bt_range_geom(ctx, band, jstart, jend, mb, q);
  const int mbe = g.mbe;
  double* W = ctx.pool.get_t<double>("bt.W", (size_t)(mbe > 512 ? mbe : 512) * nvec);
  double* X = ctx.pool.get_t<double>("bt.X", (size_t)(mbe > 512 ? mbe : 512) * nvec);
  for (int b = 0; b < g.nblk; ++b) {
    const int j0 = jstart + b * mbe;
    const int mbk = (jend - j0 < mbe) ? jend - j0 : mbe;
    const int rows = j0 + mbk - band;  // length of the longest reflector of the block
    const double* Vb = V + (size_t)j0 * ldv;
    const double* T = g.Tall + (size_t)b * g.msz;
    dgemm_dev(st, 'T', 'N', mbk, nvec, rows, 1.0, Vb, ldv, Z, ldz, 0.0, W, mbe);
    dgemm_dev(st, 'N', 'N', mbk, nvec, mbk, 1.0, T, mbe, W, mbe, 0.0, X, mbe);
    dgemm_dev(st, 'N', 'N', rows, nvec, mbk, -1.0, Vb, ldv, X, mbe, 1.0, Z, ldz);
  }
}

// Block plan of the whole back-transformation: a head range of (n - band) mod mbe short reflectors in plain
// 128-column blocks, then super-blocks up to column n -- the long reflectors, where a block costs three passes over
// all of Z, are always inside super-blocks (the remainder used to sit at the long end: 3.5 of 21.4 ms at N=8192).
struct BtPlan { int mb, q, mbe, js, rows_pad; };
static BtPlan bt_plan(int n, int mb, int band) {
  BtPlan p;
  if (mb < 8) mb = 8;
  if (mb > 128) mb = 128;  // T-builder keeps two packed mb x mb triangles in LDS; wider blocks are assembled
  // Super-blocks: every block costs three passes over Z (read for W = V^T Z, read + write for Z -= V X), and at
  // 128 columns those passes, not the MFMA work, bound the back-transformation (AI = mb/6 flop/B).
  int q = 1;
  if (mb == 128 && n >= 2048) q = (n >= 6144) ? 4 : 2;   // measured: N=8192 31.4 -> 26.2 (q=2) -> 23.9 ms (q=4)
  if (g_bt_q > 0 && mb == 128) q = g_bt_q;
  p.mb = mb; p.q = q; p.mbe = mb * q;
  const int total = n - band;                          // number of reflectors = longest reflector
  const int nsup = (q > 1) ? total / p.mbe : 0;
  p.js = (nsup > 0) ? n - nsup * p.mbe : n;            // super-block range [js, n); plain blocks [band, js)
  if (nsup == 0) p.q = 1;
  // Gram chunks read whole chunks of rows: the zero padding below the reflectors must stay inside the leading dimension
  p.rows_pad = (total + GCH - 1) / GCH * GCH;
  if (nsup > 0) {
    const int mc = (total + 4095) / 4096;
    const int gg = ((total + mc - 1) / mc + 1) & ~1;
    if (mc * gg > p.rows_pad) p.rows_pad = mc * gg;
  }
  return p;
}

// T factors of all blocks, on stream s (any stream: the caller orders it after the reduction).  Records ctx.bt_ev.
void trbak_prepare_dev(Context& ctx, int n, double* A, int lda, const double* e, int lde, int mb, int band,
                       hipStream_t s) {
  ctx.bt_ready = false;
  if (n <= band) return;
  const BtPlan p = bt_plan(n, mb, band);
  const bool inplace = p.rows_pad <= lda;
  double* V = A;
  int ldv = lda;
  if (!inplace) {
    // lda too small for the zero padding of the last Gram chunk: work on a padded copy of the reflectors
    ldv = pad_ld(p.rows_pad);
    V = ctx.pool.get_t<double>("bt.Vall", (size_t)ldv * n);
    EIGX_HIP_CHECK(hipMemcpy2DAsync(V, (size_t)ldv * 8, A, (size_t)lda * 8, (size_t)n * 8, (size_t)n,
                                    hipMemcpyDeviceToDevice, s));
  }
  hipLaunchKernelGGL(zero_below_kernel, dim3(8, n), dim3(256), 0, s, V, ldv, n, band, p.rows_pad);
  bt_prepare_range(ctx, s, n, V, ldv, e, lde, band, band, p.js, p.mb, 1);
  if (p.js < n) bt_prepare_range(ctx, s, n, V, ldv, e, lde, band, p.js, n, p.mb, p.q);
  EIGX_HIP_CHECK(hipEventRecord(ctx.bt_ev, s));
  ctx.bt_ready = true;
  ctx.bt_a = A; ctx.bt_n = n; ctx.bt_mb = mb; ctx.bt_band = band; ctx.bt_V = V; ctx.bt_ldv = ldv;
}

void trbak_dev(Context& ctx, int n, int nvec, double* A, int lda, double* Z, int ldz, const double* e,
               int lde, int mb, int band) {
  if (nvec <= 0 || n <= band) { ctx.bt_ready = false; return; }
  hipStream_t st = ctx.stream;
  if (!(ctx.bt_ready && ctx.bt_a == A && ctx.bt_n == n && ctx.bt_mb == mb && ctx.bt_band == band))
    trbak_prepare_dev(ctx, n, A, lda, e, lde, mb, band, st);
  EIGX_HIP_CHECK(hipStreamWaitEvent(st, ctx.bt_ev, 0));
  ctx.bt_ready = false;   // the plan belongs to this reduction's reflectors only
  const BtPlan p = bt_plan(n, mb, band);
  bt_apply_range(ctx, nvec, ctx.bt_V, ctx.bt_ldv, Z, ldz, band, band, p.js, p.mb, 1);
  if (p.js < n) bt_apply_range(ctx, nvec, ctx.bt_V, ctx.bt_ldv, Z, ldz, band, p.js, n, p.mb, p.q);
  EIGX_HIP_CHECK(hipGetLastError());
}

// ---- several GPUs: reflectors stay 2-D cyclic in the ranks' local blocks and stream past in column groups ---------
namespace {
// my rows of my columns of the group [j0, j1): send[ljr*nxc + li], li < nrl (local rows with global row < j1 - band)
__global__ void pack_refl_kernel(const double* __restrict__ A, int lda, int lj0, int mloc, int nrl, int nxc,
                                 double* __restrict__ send) {
  const int ljr = blockIdx.y;
  if (ljr >= mloc) return;
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < nrl; li += gridDim.x * blockDim.x)
    send[(size_t)ljr * nxc + li] = A[(size_t)(lj0 + ljr) * lda + li];
}
// V(r, c - j0) = reflector entry u_c(r) for r <= c - band (taken from its owner's piece), 0 below: the zero-padded panel
__global__ void unpack_refl_kernel(const double* __restrict__ recv, size_t count, int nxc, int j0, int j1, int band,
                                   int rows_pad, int Px, int Py, int row_major, double* __restrict__ V, int ldv) {
  const int c = j0 + blockIdx.y;
  if (c >= j1) return;
  const int qy = c % Py;
  const int lj0 = (j0 - qy + Py - 1) / Py;
  const int ljr = c / Py - lj0;
  const int len = c - band + 1;   // reflector length
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows_pad; r += gridDim.x * blockDim.x) {
    double v = 0.0;
    if (r < len) {
      const int qx = r % Px;
      const int src = row_major ? qx * Py + qy : qx + qy * Px;
      v = __hip_atomic_load(recv + (size_t)src * count + (size_t)ljr * nxc + r / Px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    V[(size_t)(c - j0) * ldv + r] = v;
  }
}
}  // namespace

// Z(:, 0:nvec) <- H_n ... H_{1+band} Z for THIS rank's eigenvector columns.  Aloc(lda, *) is the rank's block of the
// reduced matrix (reflectors in its columns, 2-D cyclic).  The reflectors are gathered group by group -- the plain
// 128-column blocks of the head range first, then a few super-blocks at a time -- into a zero-padded panel that
// every rank builds (the reference broadcasts the same panels block by block: trbakwy_datacast + bcast,
// src/trbakwy4.F:227-650); T factors are formed redundantly (O(mb^2 n) per block), the sweep over Z is local.
void trbak_mg_dev(Context& ctx, int n, int nvec, const double* Aloc, int lda, double* Z, int ldz, const double* e,
                  int lde, int mb, int band) {
  ctx.bt_ready = false;
  if (n <= band) return;
  const Grid& G = ctx.grid;
  hipStream_t st = ctx.stream;
  const BtPlan p = bt_plan(n, mb, band);
  const int total = n - band;
  int rp = p.rows_pad;
  if (rp < total + 2 * ceil_div(total, 4096) + 2) rp = total + 2 * ceil_div(total, 4096) + 2;   // chunk padding of any group
  const int ldv = pad_ld(rp);
  const int group_cols = (p.mbe * 4 > 2048) ? p.mbe * 4 : 2048;          // super-block columns gathered at a time
  const int plain_cols = (2048 / p.mb > 0 ? 2048 / p.mb : 1) * p.mb;      // plain-block columns gathered at a time
  const int gmax = group_cols > plain_cols ? group_cols : plain_cols;
  double* Vg = ctx.pool.get_t<double>("bt.Vgroup", (size_t)ldv * gmax);
  const int mloc_max = ceil_div(gmax, G.Py) + 1;
  const int nxs = (ceil_div(n, G.Px) + 7) / 8 * 8;
  PeerBuf* recv = comm_buffer(ctx, "bt.recv", (size_t)G.nranks * mloc_max * nxs * sizeof(double));
  double* send = ctx.pool.get_t<double>("bt.send", (size_t)mloc_max * nxs);
  auto run_group = [&](int j0, int j1, int q) {
    if (j1 <= j0) return;
    stage_trace(G.rank, "back-transformation: reflector group starting at column", j0);
    const int toprows = j1 - band;                      // longest reflector of the group
    const int nxc = (ceil_div(toprows, G.Px) + 7) / 8 * 8;
    const size_t cnt = (size_t)(ceil_div(j1 - j0, G.Py) + 1) * nxc;
    const int lj0 = (j0 - G.py + G.Py - 1) / G.Py;
    const int lj1 = (j1 - 1 >= G.py) ? (j1 - 1 - G.py) / G.Py : -1;
    const int mloc = lj1 - lj0 + 1;
    const int nrl = local_count(toprows, G.Px, G.px);
    if (mloc > 0 && nrl > 0)
      hipLaunchKernelGGL(pack_refl_kernel, dim3(ceil_div(nrl, 256) < 64 ? ceil_div(nrl, 256) : 64, mloc), dim3(256), 0, st,
                         Aloc, lda, lj0, mloc, nrl, nxc, send);
    comm_exchange(ctx, COMM_WORLD, send, 0, recv, 0, cnt, st, CH_BULK);
    // zero padding: the Gram chunks of the range read whole chunks of rows
    const BtRange g = bt_range_geom(ctx, band, j0, j1, p.mb, q);
    int rows_pad = g.maxchunks * g.gch;
    if (rows_pad > ldv) rows_pad = ldv;
    hipLaunchKernelGGL(unpack_refl_kernel, dim3(ceil_div(rows_pad, 256) < 64 ? ceil_div(rows_pad, 256) : 64, j1 - j0), dim3(256),
                       0, st, (const double*)recv->local, cnt, nxc, j0, j1, band, rows_pad, G.Px, G.Py, G.row_major, Vg, ldv);
    double* Vs = Vg - (size_t)j0 * ldv;                 // so that column j of the panel is Vs + j*ldv
    bt_prepare_range(ctx, st, n, Vs, ldv, e, lde, band, j0, j1, p.mb, q);
    if (nvec > 0) bt_apply_range(ctx, nvec, Vs, ldv, Z, ldz, band, j0, j1, p.mb, q);
  };
  for (int j0 = band; j0 < p.js; j0 += plain_cols) run_group(j0, (j0 + plain_cols < p.js) ? j0 + plain_cols : p.js, 1);
  for (int j0 = p.js; j0 < n; j0 += group_cols) run_group(j0, (j0 + group_cols < n) ? j0 + group_cols : n, p.q);
  EIGX_HIP_CHECK(hipGetLastError());
}

int set_bt_q(int v) { const int old = g_bt_q; g_bt_q = v; return old; }

}  // namespace eigx
