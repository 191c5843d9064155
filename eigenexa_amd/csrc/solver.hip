// solver.hip -- drivers: eigx_sx (pentadiagonal route) and eigx_s (tridiagonal route).
//
// Replaces eigen_sx (src/eigen_sx.F:30-308) and eigen_s -> eigen_FS / eigen_s0
// (src/eigen_libs.F:150-202, src/eigen_FS.F:29-300, src/eigen_s.F:30-307):
//   guards -> eigen_scaling -> band reduction -> band D&C -> back-transformation -> unscale ->
//   a(1:3,1) = flops, seconds, comm seconds (src/eigen_sx.F:285-296).
#include "eigx_context.h"
#include "eigx_comm.h"
#include "../../include/eigenexa_amd.h"
#include <chrono>
#include <cfloat>
#include <limits>
#include <vector>
#include <cstring>

namespace eigx {

void trbak_prepare_dev(Context& ctx, int n, double* A, int lda, const double* e, int lde, int mb, int band,
                       hipStream_t s);
void trbak_dev(Context& ctx, int n, int nvec, double* A, int lda, double* Z, int ldz, const double* e,
               int lde, int mb, int band);
void trbak_mg_dev(Context& ctx, int n, int nvec, const double* Aloc, int lda, double* Z, int ldz, const double* e,
                  int lde, int mb, int band);

namespace {

// max |a_ij| over the upper triangle and a non-finite flag (eigen_scaling, src/eigen_scaling.F:86-150)
// A is the local block of a 2-D cyclic distribution: local (i, j) = global (i*Px + px, j*Py + py); ncl local columns
__global__ __launch_bounds__(256) void absmax_kernel(const double* __restrict__ A, int lda, int ncl, int Px, int px, int Py,
                                                     int py, double* __restrict__ out /* [gridDim.x][2] */) {
  __shared__ double smax[4], sbad[4];
  double mx = 0.0, bad = 0.0;
  for (int j = blockIdx.x; j < ncl; j += gridDim.x) {
    const double* col = A + (size_t)j * lda;
    const int gj = j * Py + py;
    const int iend = gj >= px ? (gj - px) / Px : -1;   // last local row with global row <= gj
    for (int i = threadIdx.x; i <= iend; i += 256) {
      const double v = fabs(col[i]);
      if (!(v <= DBL_MAX)) bad = 1.0;
      else mx = fmax(mx, v);
    }
  }
  for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o, 64)); bad = fmax(bad, __shfl_xor(bad, o, 64)); }
  if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = mx; sbad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    out[2 * blockIdx.x + 1] = fmax(fmax(sbad[0], sbad[1]), fmax(sbad[2], sbad[3]));
  }
}

__global__ void scale_upper_kernel(double* __restrict__ A, int lda, int ncl, int Px, int px, int Py, int py, double s) {
  for (int j = blockIdx.x; j < ncl; j += gridDim.x) {
    double* col = A + (size_t)j * lda;
    const int gj = j * Py + py;
    const int iend = gj >= px ? (gj - px) / Px : -1;
    for (int i = threadIdx.x; i <= iend; i += blockDim.x) col[i] *= s;
  }
}

// two-number reduction of the absmax partials on the device (so that the cross-rank MAX can follow on the stream)
__global__ void absmax_final_kernel(const double* __restrict__ part, int nb, double* __restrict__ out2) {
  __shared__ double smax[4], sbad[4];
  double mx = 0.0, bad = 0.0;
  for (int q = threadIdx.x; q < nb; q += 256) { mx = fmax(mx, part[2 * q]); bad = fmax(bad, part[2 * q + 1]); }
  for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o, 64)); bad = fmax(bad, __shfl_xor(bad, o, 64)); }
  if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = mx; sbad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out2[0] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    out2[1] = fmax(fmax(sbad[0], sbad[1]), fmax(sbad[2], sbad[3]));
  }
}

__global__ void scale_vec_kernel(double* __restrict__ w, int n, double s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] *= s;
}

__global__ void fill_vec_kernel(double* __restrict__ w, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = v;
}

// local index l of process p (of P) -> global index, blocks of nb (nb = 1: cyclic, l*P + p)
__device__ __forceinline__ int bc_l2g(int l, int nb, int P, int p) { return ((l / nb) * P + p) * nb + l % nb; }

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// z(:, j) = e_j for the first gridDim.y columns (modes 'S', 'C': eigen_identity, src/eigen_sx.F:214)
__global__ void identity_kernel(double* __restrict__ z, int ldz, int n, int c0) {
  const int j = blockIdx.y;   // local column; global column c0 + j
  double* col = z + (size_t)j * ldz;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) col[r] = (r == c0 + j) ? 1.0 : 0.0;
}

__device__ __host__ __forceinline__ int bc_owner(int g, int nb, int P) { return (g / nb) % P; }
__device__ __host__ __forceinline__ int bc_g2l(int g, int nb, int P) { return ((g / nb) / P) * nb + g % nb; }
// number of indices g < n that process p owns (NUMROC), usable on the device
__device__ __host__ __forceinline__ int bc_count(int n, int nb, int p, int P) {
  const int nblocks = n / nb;
  int cnt = (nblocks / P) * nb;
  const int extra = nblocks % P;
  if (p < extra) cnt += nb;
  else if (p == extra) cnt += n % nb;
  return cnt;
}

// Eigenvector column block of this rank (columns [c0, c0 + cnt), all n rows) -> pieces for the all-to-all that deals
// the matrix into the callers' 2-D (block-)cyclic blocks: the piece for rank (qx, qy) holds the rows that qx owns of
// those of my columns that qy owns: send[rank][ljr * nrmax + li]   (src/dc_redist1.F / dc_redist2.F play this role
// in the reference, between its D&C layout and the API layout)
__global__ void pack_z_pieces_kernel(const double* __restrict__ Z, int ldz, int n, int c0, int cnt, int nb, int Px, int Py,
                                     int row_major, int nrmax, size_t piece, double* __restrict__ send) {
  const int cl = blockIdx.y, qx = blockIdx.z;
  if (cl >= cnt) return;
  const int c = c0 + cl;
  const int qy = bc_owner(c, nb, Py);
  const int ljr = bc_g2l(c, nb, Py) - bc_count(c0, nb, qy, Py);
  const int dst = row_major ? qx * Py + qy : qx + qy * Px;
  const int nr = bc_count(n, nb, qx, Px);
  double* out = send + (size_t)dst * piece + (size_t)ljr * nrmax;
  const double* col = Z + (size_t)cl * ldz;
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < nr; li += gridDim.x * blockDim.x)
    out[li] = col[bc_l2g(li, nb, Px, qx)];
}
// z_user(li, lj) for my local columns that lie in source rank q's column range [q*zc, min((q+1)*zc, nvec))
__global__ void unpack_z_pieces_kernel(const double* __restrict__ recv, size_t piece, int nrmax, int nvec, int zc, int nb,
                                       int py, int Py, int nr, double* __restrict__ z, int ldz) {
  const int q = blockIdx.z;
  const int g0 = q * zc < nvec ? q * zc : nvec, g1 = (q + 1) * zc < nvec ? (q + 1) * zc : nvec;
  const int l0 = bc_count(g0, nb, py, Py), l1 = bc_count(g1, nb, py, Py);
  const int ljr = blockIdx.y;
  if (ljr >= l1 - l0) return;
  const double* src = recv + (size_t)q * piece + (size_t)ljr * nrmax;
  double* col = z + (size_t)(l0 + ljr) * ldz;
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < nr; li += gridDim.x * blockDim.x)
    col[li] = __hip_atomic_load(src + li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- block-cyclic (nb x nb blocks, a ScaLAPACK descriptor's layout) -> cyclic, as one all-to-all ---------------------
// The element a rank holds at local (li, lj) is global (gi, gj) = (l2g(li), l2g(lj)); in the cyclic layout it belongs to
// rank (gi mod Px, gj mod Py) at local (gi div Px, gj div Py).  The piece for a destination is addressed by the RANK of
// the row / column among the sender's rows / columns that go to that destination: rrank[li], crank[lj] (host tables,
// O(n / P) integers); the receiver holds, for each of its cyclic rows / columns, the sender's grid coordinate and that
// rank (srcx / posr, srcy / posc).  This is what pdgemr2d does for the reference's callers (manual 3.4).
__global__ void bc_pack_kernel(const double* __restrict__ a, int lda, int nr, int nc, int nb, int Px, int px, int Py, int py,
                               int row_major, const int* __restrict__ rrank, const int* __restrict__ crank, int nrp,
                               size_t piece, double* __restrict__ send) {
  const int lj = blockIdx.y;
  if (lj >= nc) return;
  const int gj = bc_l2g(lj, nb, Py, py);
  const int qy = gj % Py;
  const int pc = crank[lj];
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < nr; li += gridDim.x * blockDim.x) {
    const int gi = bc_l2g(li, nb, Px, px);
    const int qx = gi % Px;
    const int dst = row_major ? qx * Py + qy : qx + qy * Px;
    send[(size_t)dst * piece + (size_t)pc * nrp + rrank[li]] = a[(size_t)lj * lda + li];
  }
}
__global__ void bc_unpack_kernel(const double* __restrict__ recv, size_t piece, int nrp, int clr, int clc, int Px, int Py,
                                 int row_major, const int* __restrict__ srcx, const int* __restrict__ posr,
                                 const int* __restrict__ srcy, const int* __restrict__ posc, double* __restrict__ out, int ldo) {
  const int lj = blockIdx.y;
  if (lj >= clc) return;
  const int sy = srcy[lj], pc = posc[lj];
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < clr; li += gridDim.x * blockDim.x) {
    const int sx = srcx[li];
    const int src = row_major ? sx * Py + sy : sx + sy * Px;
    out[(size_t)lj * ldo + li] = __hip_atomic_load(recv + (size_t)src * piece + (size_t)pc * nrp + posr[li], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// one dimension of the index tables: n indices dealt in blocks of nb to P processes (me = p) -> cyclic over the same P
static void bc_tables_1d(int n, int nb, int P, int p, std::vector<int>& rank_of_local, std::vector<int>& src_of_cyc,
                         std::vector<int>& pos_of_cyc, int& max_piece) {
  const int nl = numroc(n, nb, p, P);
  rank_of_local.assign(nl > 0 ? nl : 1, 0);
  std::vector<int> cnt(P, 0);
  for (int l = 0; l < nl; ++l) {                       // my block-cyclic indices in ascending local order
    const int g = ((l / nb) * P + p) * nb + l % nb;
    rank_of_local[l] = cnt[g % P]++;
  }
  // what I receive as cyclic owner p: my cyclic index c is global g = c*P + p, held by block-cyclic process (g/nb) % P
  // at the rank it has among THAT process's indices going to me
  const int nc = local_count(n, P, p);
  src_of_cyc.assign(nc > 0 ? nc : 1, 0);
  pos_of_cyc.assign(nc > 0 ? nc : 1, 0);
  std::vector<int> seen(P, 0);
  for (int c = 0; c < nc; ++c) {                       // ascending global order = ascending local order on every sender
    const int g = c * P + p;
    const int s = (g / nb) % P;
    src_of_cyc[c] = s;
    pos_of_cyc[c] = seen[s]++;
  }
  max_piece = 0;
  for (int q = 0; q < P; ++q) { if (cnt[q] > max_piece) max_piece = cnt[q]; if (seen[q] > max_piece) max_piece = seen[q]; }
}

static int bc_to_cyclic(Context& ctx, const double* a, int lda, int n, int nb, double* out, int ldo, hipStream_t st) {
  const Grid& G = ctx.grid;
  const int P = G.nranks;
  std::vector<int> rrank, srcx, posr, crank, srcy, posc;
  int mr = 0, mc = 0;
  bc_tables_1d(n, nb, G.Px, G.px, rrank, srcx, posr, mr);
  bc_tables_1d(n, nb, G.Py, G.py, crank, srcy, posc, mc);
  // the piece extents must agree on every rank: an upper bound that depends on (n, nb, grid) only
  // (every block of a sender starts at the same residue mod P, so one destination can get ceil(nb/P) rows of EVERY block)
  const int nrp = (numroc(n, nb, 0, G.Px) / nb + 1) * ceil_div(nb, G.Px), ncp = (numroc(n, nb, 0, G.Py) / nb + 1) * ceil_div(nb, G.Py);
  if (mr > nrp || mc > ncp) {   // cannot happen (see the bound above); refuse rather than write past a piece
    fprintf(stderr, "[eigx] internal: block-cyclic piece bound violated (%d > %d or %d > %d)\n", mr, nrp, mc, ncp);
    return EIGX_ERR_INTERNAL;
  }
  const size_t piece = (size_t)nrp * ncp;
  const int nr = numroc(n, nb, G.px, G.Px), nc = numroc(n, nb, G.py, G.Py);
  const int clr = local_count(n, G.Px, G.px), clc = local_count(n, G.Py, G.py);
  const size_t nt = rrank.size() + srcx.size() + posr.size() + crank.size() + srcy.size() + posc.size();
  int* tab = ctx.pool.get_t<int>("mg.bctab", nt);
  int* htab = (int*)ctx.pool.get_host("mg.bctab", nt * sizeof(int));
  size_t o = 0;
  auto put = [&](const std::vector<int>& v) { int* d = tab + o; memcpy(htab + o, v.data(), v.size() * sizeof(int)); o += v.size(); return d; };
  const int* d_rrank = put(rrank); const int* d_srcx = put(srcx); const int* d_posr = put(posr);
  const int* d_crank = put(crank); const int* d_srcy = put(srcy); const int* d_posc = put(posc);
  EIGX_HIP_CHECK(hipMemcpyAsync(tab, htab, nt * sizeof(int), hipMemcpyHostToDevice, st));
  double* sendb = ctx.pool.get_t<double>("mg.xsend", piece * P);
  double* recvb = ctx.pool.get_t<double>("mg.xrecv", piece * P);
  if (nr > 0 && nc > 0)
    hipLaunchKernelGGL(bc_pack_kernel, dim3(8, nc), dim3(256), 0, st, a, lda, nr, nc, nb, G.Px, G.px, G.Py, G.py, G.row_major,
                       d_rrank, d_crank, nrp, piece, sendb);
  comm_exchange_big(ctx, COMM_WORLD, sendb, piece, recvb, piece, st);
  if (clr > 0 && clc > 0)
    hipLaunchKernelGGL(bc_unpack_kernel, dim3(8, clc), dim3(256), 0, st, (const double*)recvb, piece, nrp, clr, clc, G.Px,
                       G.Py, G.row_major, d_srcx, d_posr, d_srcy, d_posc, out, ldo);
  EIGX_HIP_CHECK(hipStreamSynchronize(st));   // the pinned table staging buffer is reused by the next call
  return EIGX_OK;
}

}  // namespace

// Eigenvector column blocks (rank r holds columns [r zc, r zc + zc) of the first nvec, all n rows: zcols(ldz, zcnt) are
// mine, starting at global column zc0) -> the callers' 2-D (block-)cyclic blocks: one all-to-all of
// (rows of qx) x (my columns of qy) pieces.  Enqueued on st.  (eigen_h's split planes go through it one plane at a time.)
void cols_to_cyclic_dev(Context& ctx, int n, int nvec, int nb, int zc, int zc0, int zcnt, const double* zcols, int ldz,
                        double* z_user, int ldz_user, hipStream_t st) {
  const Grid& G = ctx.grid;
  const int P = G.nranks;
  const int nloc_r = numroc(n, nb, G.px, G.Px);
  const int nrmax = numroc(n, nb, 0, G.Px);
  const int ncmax = (zc / (nb * G.Py) + 2) * nb;
  const size_t piece = (size_t)nrmax * ncmax;
  double* sendb = ctx.pool.get_t<double>("mg.xsend", piece * P);
  double* recvb = ctx.pool.get_t<double>("mg.xrecv", piece * P);
  if (zcnt > 0)
    hipLaunchKernelGGL(pack_z_pieces_kernel, dim3(8, zcnt, G.Px), dim3(256), 0, st, zcols, ldz, n, zc0, zcnt, nb, G.Px,
                       G.Py, G.row_major, nrmax, piece, sendb);
  comm_exchange_big(ctx, COMM_WORLD, sendb, piece, recvb, piece, st);
  if (nloc_r > 0)
    hipLaunchKernelGGL(unpack_z_pieces_kernel, dim3(8, ncmax, P), dim3(256), 0, st, (const double*)recvb, piece, nrmax,
                       nvec, zc, nb, G.py, G.Py, nloc_r, z_user, ldz_user);
}

namespace {

// nb = block size of the 2-D block-cyclic layout of a and z over the process grid (1 = the cyclic layout of the
// EigenExa API; a ScaLAPACK caller passes its descriptor's MB = NB and needs no pdgemr2d redistribution, manual 3.4)
int solve_dev(Context& ctx, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb,
              char mode, int band, int nb) {
  if (!ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (n <= 0) {
    fprintf(stderr, "[eigx] warning: non-positive dimension is invalid\n");  // src/eigen_sx.F:95-98
    return EIGX_ERR_BAD_ARG;
  }
  const Grid& G = ctx.grid;
  const int P = G.nranks;
  if (nb < 1) return EIGX_ERR_BAD_ARG;
  const int nloc_r = numroc(n, nb, G.px, G.Px), nloc_c = numroc(n, nb, G.py, G.Py);
  if (lda < (nloc_r > 1 ? nloc_r : 1) || !a || !w) return EIGX_ERR_BAD_ARG;
  if (mode >= 'a' && mode <= 'z') mode = (char)(mode - 'a' + 'A');
  if (nvec == 0) mode = 'N';                      // src/eigen_sx.F:108-110
  if (nvec < 0) nvec = -nvec;
  if (nvec > n) nvec = n;
  const bool want_vec = (mode != 'N');
  if (want_vec && (!z || ldz < (nloc_r > 1 ? nloc_r : 1))) return EIGX_ERR_BAD_ARG;
  if (mf <= 0) mf = 128;
  if (mb <= 0) mb = 128;
  EIGX_HIP_CHECK(hipSetDevice(ctx.device));
  // The library works on its own non-blocking streams: whatever the caller queued on the default stream to fill a
  // (a copy, a generator kernel) has to be complete before the first kernel here reads it.  (Found by a test that
  // filled `a` with an asynchronous copy and called the C-ABI directly: the second solve of a process -- workspace
  // already allocated, nothing else in the way -- overtook the copy.)
  EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));
  hipStream_t st = ctx.stream;
  ctx.errinfo = 0;
  ctx.dc_zero_n = 0;
  for (int q = 0; q < 16; ++q) ctx.timers[q] = 0.0;
  if (P > 1) (void)comm_seconds(ctx, true);
  const double t0 = now_s();

  double* a_user = a;
  double* z_user = z;
  const int ldz_user = ldz;
  // The kernels read columns in 16-byte pieces: an odd leading dimension (eigen_get_matdims mode 'M' with an odd
  // ceil(n/Px) produces one) is served from an internal padded copy; `a` is destroyed by contract anyway.
  // Several GPUs: the cyclic block a(lda, *) is used IN PLACE -- nothing of A is replicated; a block-cyclic caller
  // (nb > 1, the ScaLAPACK interop entry) is converted to the cyclic layout by one all-to-all first.
  const int clr = local_count(n, G.Px, G.px), clc = local_count(n, G.Py, G.py);   // cyclic local extents
  if ((lda & 1) || ((uintptr_t)a & 15) || (P > 1 && nb > 1)) {
    const int ldi = pad_ld(clr + 2);
    double* ai = ctx.pool.get_t<double>("sol.apad", (size_t)ldi * (clc > 0 ? clc : 1));
    if (P > 1 && nb > 1) {
      const int rc_bc = bc_to_cyclic(ctx, a, lda, n, nb, ai, ldi, st);     // one all-to-all: nothing is replicated
      if (rc_bc != EIGX_OK) return rc_bc;
    } else if (clr > 0 && clc > 0) {
      EIGX_HIP_CHECK(hipMemcpy2DAsync(ai, (size_t)ldi * 8, a, (size_t)lda * 8, (size_t)clr * 8, (size_t)clc,
                                      hipMemcpyDeviceToDevice, st));
    }
    a = ai;
    lda = ldi;
  }
  // eigenvector workspace.  One GPU: the caller's z.  Several GPUs: the D&C and the back-transformation work on
  // whole eigenvector COLUMNS (rank r: columns [r*zc, (r+1)*zc) of the n x nvec matrix); the result is dealt back
  // into the caller's cyclic z(ldz, *) at the end.
  int zcols_per_rank = 0;
  if (P > 1 || (want_vec && ((ldz & 1) || ((uintptr_t)z & 15)))) {
    zcols_per_rank = ceil_div(nvec > 0 ? nvec : 1, P);
    if (want_vec) {
      const int ldf = pad_ld(n);
      z = ctx.pool.get_t<double>("mg.Z", (size_t)ldf * (size_t)zcols_per_rank);   // this rank's column block only
      ldz = ldf;
    }
  }

  // ---- eigen_scaling ---------------------------------------------------------------------------
  double sigma = 1.0;
  {
    const int nbk = 512;
    double* part = ctx.pool.get_t<double>("sol.absmax", (size_t)2 * nbk + 8);
    hipLaunchKernelGGL(absmax_kernel, dim3(nbk), dim3(256), 0, st, a, lda, clc, G.Px, G.px, G.Py, G.py, part);
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(256), 0, st, part, nbk, part + 2 * nbk);
    if (P > 1) comm_allreduce_max(ctx, COMM_WORLD, part + 2 * nbk, 2, st);       // src/eigen_scaling.F:118-123
    double hp[2] = {0.0, 0.0};
    EIGX_HIP_CHECK(hipMemcpyAsync(hp, part + 2 * nbk, sizeof(hp), hipMemcpyDeviceToHost, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
    const double anrm = hp[0], bad = hp[1];
    if (bad != 0.0) {  // NaN/Inf in the input: w(:) = NaN and return (src/eigen_sx.F:151-155)
      hipLaunchKernelGGL(fill_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, n,
                         std::numeric_limits<double>::quiet_NaN());
      EIGX_HIP_CHECK(hipStreamSynchronize(st));
      ctx.errinfo = -1;
      return EIGX_ERR_NONFINITE;
    }
    // eigen_scaling (src/eigen_scaling.F:76-81,:127-147) rescales only when max|a| leaves the safe range,
    // to RMIN/RMAX ~ 1e-146/1e+146.  This implementation forms reflector quantities that are cubic in the
    // matrix scale (u^T A u with un-normalised u), so its safe range is narrower and the target is O(1):
    // outside [1e-90, 1e90] the matrix is scaled by the exact power of two nearest to 1/max|a|.
    if (anrm > 0.0 && (anrm < 1e-90 || anrm > 1e90)) {
      int ex = 0;
      (void)frexp(anrm, &ex);
      sigma = ldexp(1.0, -ex);
    }
    if (sigma != 1.0)
      hipLaunchKernelGGL(scale_upper_kernel, dim3(1024), dim3(256), 0, st, a, lda, clc, G.Px, G.px, G.Py, G.py, sigma);
  }

  // ---- forward reduction --------------------------------------------------------------------------
  const int lde = (n + 3) / 4 * 4;  // nme of src/eigen_sx.F:139
  double* d = ctx.pool.get_t<double>("sol.d", (size_t)n);
  double* e = ctx.pool.get_t<double>("sol.e", (size_t)lde * 2);
  const double t1 = now_s();
  // modes that run the D&C: zero its two Q buffers on the side stream underneath the reduction
  if (!(mode == 'N' || mode == 'S' || mode == 'C')) band_dc_prepare(ctx, n);
  band_reduce_dev(ctx, n, a, lda, d, e, lde, mf, band);
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  if (P > 1 && comm_failed(ctx)) return EIGX_ERR_INTERNAL;
  const double t2 = now_s();
  stage_trace(G.rank, "reduction done");

  // ---- divide and conquer --------------------------------------------------------------------------
  // modes (src/eigen_sx.F:200-222): A/X/T/R divide and conquer (X: eigenvalues then re-done by bisection),
  // S/C identity eigenvector matrix + bisection, N bisection only
  const bool do_bt = want_vec && mode != 'T' && mode != 'C' && mode != 'R';  // src/eigen_sx.F:240
  // The T factors of the back-transformation depend on the reflectors only: build them on the side stream while the
  // divide and conquer (launch-bound at its low levels) has the compute stream.  The reduction is complete here
  // (the host synchronised the compute stream above).
  const bool runs_dc = !(mode == 'N' || mode == 'S' || mode == 'C');
  ctx.dc_side_work = nullptr;   // (a solve that failed before its D&C ran may have left one behind)
  if (do_bt && nvec > 0 && P == 1) {
    if (runs_dc) ctx.dc_side_work = [&ctx, n, a, lda, e, lde, mb, band] { trbak_prepare_dev(ctx, n, a, lda, e, lde, mb, band, ctx.bt_stream); };
    else trbak_prepare_dev(ctx, n, a, lda, e, lde, mb, band, ctx.side_stream);
  }
  // several GPUs: this rank's eigenvector columns [zc0, zc0 + zcnt) (the D&C delivers them, all n rows each)
  const int zc0 = (P > 1) ? ((G.rank * zcols_per_rank < nvec) ? G.rank * zcols_per_rank : nvec) : 0;
  const int zcnt = (P > 1) ? ((nvec - zc0 < zcols_per_rank) ? nvec - zc0 : zcols_per_rank) : nvec;
  if (mode == 'N' || mode == 'S' || mode == 'C') {
    if (want_vec && zcnt > 0) hipLaunchKernelGGL(identity_kernel, dim3(8, zcnt), dim3(256), 0, st, z, ldz, n, zc0);
    band_bisect_dev(ctx, n, d, e, lde, band, w);
  } else {
    band_dc_dev(ctx, n, nvec, d, e, lde, band, w, z, ldz);
    if (ctx.dc_side_work) { std::function<void()> f = std::move(ctx.dc_side_work); ctx.dc_side_work = nullptr; f(); }   // not consumed (cannot happen today)
    if (mode == 'X') band_bisect_dev(ctx, n, d, e, lde, band, w);
  }
  const double t3 = now_s();
  stage_trace(G.rank, "eigenvalue stage done");

  // ---- back-transformation ---------------------------------------------------------------------------
  if (do_bt) {
    if (P == 1) {
      trbak_dev(ctx, n, nvec, a, lda, z, ldz, e, lde, mb, band);
    } else {
      // eigenvector columns are split over the ranks; the reflectors stay distributed and stream past in column
      // groups (trbak.hip); no communication inside a group's sweep
      trbak_mg_dev(ctx, n, zcnt, a, lda, z, ldz, e, lde, mb, band);
    }
  }
  stage_trace(G.rank, "back-transformation enqueued");
  if (P > 1 && want_vec) {
    cols_to_cyclic_dev(ctx, n, nvec, nb, zcols_per_rank, zc0, zcnt, z, ldz, z_user, ldz_user, st);
  } else if (want_vec && z != z_user) {
    EIGX_HIP_CHECK(hipMemcpy2DAsync(z_user, (size_t)ldz_user * 8, z, (size_t)ldz * 8, (size_t)n * 8, (size_t)nvec,
                                    hipMemcpyDeviceToDevice, st));
  }
  if (sigma != 1.0 && sigma != 0.0)
    hipLaunchKernelGGL(scale_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, n, 1.0 / sigma);
  stage_trace(G.rank, "exit redistribution enqueued");
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  stage_trace(G.rank, "stream drained");
  if (P > 1 && comm_failed(ctx)) return EIGX_ERR_INTERNAL;
  const double t4 = now_s();

  // ---- statistics (src/eigen_sx.F:285-296) -----------------------------------------------------------
  const double f_red = 4.0 / 3.0 * (double)n * n * n;
  const double f_dc = ctx.timers[11];
  const double f_bt = do_bt ? 2.0 * (double)nvec * n * n : 0.0;
  double ret = f_red + f_dc + f_bt;
  if (f_dc == 0.0) ret = -ret;
  // a(3,1): seconds this rank spent communicating (waits for peers included), as the reference returns
  // (src/eigen_sx.F:285-296); -1 on one GPU, where there is none
  const double t_comm = (P > 1) ? comm_seconds(ctx, false) : -1.0;
  ctx.timers[0] = t4 - t0; ctx.timers[1] = t2 - t1; ctx.timers[2] = t3 - t2; ctx.timers[3] = t4 - t3;
  ctx.timers[4] = (P > 1) ? t_comm : 0.0; ctx.timers[12] = ret;
  const double stats[3] = {ret, t4 - t0, t_comm};
  int nst = nloc_r >= 3 ? 3 : nloc_r;  // a(1:3,1) lives in the first local column
  if (nloc_c == 0) nst = 0;
  if (nst > 0) EIGX_HIP_CHECK(hipMemcpyAsync(a_user, stats, (size_t)nst * 8, hipMemcpyHostToDevice, st));
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  return EIGX_OK;
}

int solve_host(Context& ctx, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb,
               char mode, int band, int nb) {
  if (!ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (nb < 1) return EIGX_ERR_BAD_ARG;
  const int nr = numroc(n, nb, ctx.grid.px, ctx.grid.Px), nc = numroc(n, nb, ctx.grid.py, ctx.grid.Py);
  if (n <= 0 || !a || !w || lda < nr) return EIGX_ERR_BAD_ARG;
  EIGX_HIP_CHECK(hipSetDevice(ctx.device));
  const int ldd = pad_ld(nr + 2);  // device leading dimension: even (16-byte column loads), odd multiple of 32
  const int ncd = nc > 0 ? nc : 1;
  double* ad = ctx.pool.get_t<double>("host.a", (size_t)ldd * ncd);
  double* zd = ctx.pool.get_t<double>("host.z", (size_t)ldd * ncd);
  double* wd = ctx.pool.get_t<double>("host.w", (size_t)n);
  if (nr > 0 && nc > 0)
    EIGX_HIP_CHECK(hipMemcpy2D(ad, (size_t)ldd * 8, a, (size_t)lda * 8, (size_t)nr * 8, (size_t)nc,
                               hipMemcpyHostToDevice));
  const int rc = solve_dev(ctx, n, nvec, ad, ldd, wd, zd, ldd, mf, mb, mode, band, nb);
  EIGX_HIP_CHECK(hipMemcpy(w, wd, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (rc != EIGX_OK) return rc;
  char md = mode;
  if (md >= 'a' && md <= 'z') md = (char)(md - 'a' + 'A');
  int nv = nvec < 0 ? -nvec : nvec;
  if (nv > n) nv = n;
  const int nzc = numroc(nv, nb, ctx.grid.py, ctx.grid.Py);
  if (z && nzc > 0 && nr > 0 && md != 'N')
    EIGX_HIP_CHECK(hipMemcpy2D(z, (size_t)ldz * 8, zd, (size_t)ldd * 8, (size_t)nr * 8, (size_t)nzc,
                               hipMemcpyDeviceToHost));
  // `a` is destroyed by contract; only the statistics come back
  const int nst = (nc > 0) ? (nr >= 3 ? 3 : nr) : 0;
  if (nst > 0) EIGX_HIP_CHECK(hipMemcpy(a, ad, (size_t)nst * 8, hipMemcpyDeviceToHost));
  return EIGX_OK;
}

// ---- KMATH_EIGEN_GEV: generalised symmetric-definite problem A x = lambda B x -----------------------------
// lower triangle := upper triangle (the GEMMs below need the full symmetric A; trpos_utol of the reference,
// src/KMATH_EIGEN_GEV_misc.F:140-173)
__global__ void symmetrize_kernel(double* __restrict__ a, int lda, int n) {
  const int j = blockIdx.y;
  for (int i = j + 1 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    a[(size_t)j * lda + i] = a[(size_t)i * lda + j];
}

// b(:, j) = z(:, j) * w(j)^(-1/2)   (diag_mult, src/KMATH_EIGEN_GEV_misc.F:49-104)
__global__ void scale_cols_rsqrt_kernel(const double* __restrict__ z, int ldz, const double* __restrict__ w,
                                        double* __restrict__ b, int ldb, int n) {
  const int j = blockIdx.y;
  const double s = 1.0 / sqrt(w[j]);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    b[(size_t)j * ldb + i] = z[(size_t)j * ldz + i] * s;
}


// ---- multi-rank KMATH_EIGEN_GEV on the 2-D cyclic blocks -------------------------------------------------------------
// Two building blocks, both O(n^2 / P) memory per rank:
//   dist_transpose : Z = A^T.  Element A(j, i) lives on rank (j % Px, i % Py) and goes to rank (i % Px, j % Py): on a
//                    non-square grid that is a genuine all-to-all.  The rows i that rank (px, .) receives from a source in
//                    process column sy are the i = i0 + t L (L = lcm(Px, Py), i0 by the Chinese remainder theorem, none
//                    if px != sy mod gcd); likewise the columns j = j0 + u L: a piece is the (t, u) rectangle, piece
//                    [u][t].  (role of PDTRAN + trpos_utol, src/KMATH_EIGEN_GEV_1.F:57-58)
//   dist_gemm_nn   : C = A B (SUMMA): for every panel of kb global indices k the ranks of a process ROW allgather their
//                    columns of A(:, k-panel), the ranks of a process COLUMN their rows of B(k-panel, :), and the local
//                    fp64 MFMA GEMM accumulates the panel product.  (role of the three PDGEMMs, :100-139)
struct TrPeers { int i0[EIGX_MAXP], j0[EIGX_MAXP]; };   // per peer (world rank order): first row / column of the piece, -1 = empty
// pack: piece for destination d, element [u][t] = A(j0 + u L, i0 + t L) of my block (row j, column i)
__global__ void tr_pack_kernel(const double* __restrict__ a, int lda, int n, int Px, int Py, int L, TrPeers tp, int nimax,
                               int u0, int ucw, double* __restrict__ send) {
  const int d = blockIdx.z;
  const int i0 = tp.i0[d], j0 = tp.j0[d];
  for (int uu = blockIdx.y; uu < ucw; uu += gridDim.y) {
    const int u = u0 + uu;
    double* dst = send + ((size_t)d * ucw + uu) * nimax;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nimax; t += gridDim.x * blockDim.x) {
      double v = 0.0;
      if (i0 >= 0 && j0 >= 0) {
        const int i = i0 + t * L, j = j0 + u * L;   // I hold row j (local j / Px), column i (local i / Py)
        if (i < n && j < n) v = a[(size_t)(i / Py) * lda + j / Px];
      }
      dst[t] = v;
    }
  }
}
// unpack: Z(i, j) = piece from source s at [u][t]; I hold row i (local i / Px), column j (local j / Py)
__global__ void tr_unpack_kernel(const double* __restrict__ recv, int n, int Px, int Py, int L, TrPeers tp, int nimax,
                                 int u0, int ucw, double* __restrict__ z, int ldz) {
  const int sidx = blockIdx.z;
  const int i0 = tp.i0[sidx], j0 = tp.j0[sidx];
  if (i0 < 0 || j0 < 0) return;
  for (int uu = blockIdx.y; uu < ucw; uu += gridDim.y) {
    const int u = u0 + uu;
    const double* src = recv + ((size_t)sidx * ucw + uu) * nimax;
    const int j = j0 + u * L;
    if (j >= n) return;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nimax; t += gridDim.x * blockDim.x) {
      const int i = i0 + t * L;
      if (i < n) z[(size_t)(j / Py) * ldz + i / Px] = __hip_atomic_load(src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
static int crt_small(int a, int A_, int b, int B_, int L) {   // smallest x < L with x % A_ == a and x % B_ == b, -1 if none
  for (int x = 0; x < L; ++x)
    if (x % A_ == a && x % B_ == b) return x;
  return -1;
}
static int gcd_int(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
// the pieces rank (px, py) exchanges with rank (qx, qy) in Z = A^T (pure arithmetic; eigx_transpose_plan exposes it to the
// CPU tests, which assemble A^T from the pieces for every grid)
static void transpose_plan(int Px, int Py, int px, int py, int qx, int qy, int* send_i0, int* send_j0, int* recv_i0,
                           int* recv_j0, int* step) {
  const int L = Px / gcd_int(Px, Py) * Py;
  // to (qx, qy): its rows i (i % Px == qx) among my columns (i % Py == py); its columns j (j % Py == qy) among my rows
  *send_i0 = crt_small(qx, Px, py, Py, L);
  *send_j0 = crt_small(px, Px, qy, Py, L);
  // from (qx, qy): my rows i (i % Px == px) among its columns (i % Py == qy); my columns j (j % Py == py) among its rows
  *recv_i0 = crt_small(px, Px, qy, Py, L);
  *recv_j0 = crt_small(qx, Px, py, Py, L);
  *step = L;
}

// z(ldz, nc) = (a(lda, nc))^T on the cyclic blocks (both n x n); enqueued on st.  The all-to-all's pieces are uniform, and
// only gcd(Px, Py)^-2 of the rank pairs exchange anything, so the exchange runs in rounds over the pieces' columns u that
// keep the send + receive buffers at about one local block each.
static void dist_transpose(Context& ctx, int n, const double* a, int lda, double* z, int ldz, hipStream_t st) {
  const Grid& G = ctx.grid;
  const int g = gcd_int(G.Px, G.Py);
  const int P = G.nranks, L = G.Px / g * G.Py;
  const int nimax = ceil_div(n, L);
  const int ucw = ceil_div(nimax, g * g);                      // piece columns per round
  const size_t count = (size_t)nimax * ucw;
  double* sendb = ctx.pool.get_t<double>("gev.tsend", count * P);
  double* recvb = ctx.pool.get_t<double>("gev.trecv", count * P);
  TrPeers to, from;
  for (int q = 0; q < P; ++q) {
    const int qx = G.row_major ? q / G.Py : q % G.Px, qy = G.row_major ? q % G.Py : q / G.Px;
    int step_;
    transpose_plan(G.Px, G.Py, G.px, G.py, qx, qy, &to.i0[q], &to.j0[q], &from.i0[q], &from.j0[q], &step_);
  }
  const int gy = ucw < 32768 ? ucw : 32768;
  for (int u0 = 0; u0 < nimax; u0 += ucw) {
    hipLaunchKernelGGL(tr_pack_kernel, dim3(ceil_div(nimax, 256), gy, P), dim3(256), 0, st, a, lda, n, G.Px, G.Py, L, to, nimax, u0,
                       ucw, sendb);
    comm_exchange_big(ctx, COMM_WORLD, sendb, count, recvb, count, st);
    hipLaunchKernelGGL(tr_unpack_kernel, dim3(ceil_div(nimax, 256), gy, P), dim3(256), 0, st, (const double*)recvb, n, G.Px, G.Py, L,
                       from, nimax, u0, ucw, z, ldz);
  }
}

// a(i, j) for i > j (global indices) from t = a^T: the full symmetric matrix out of its upper triangle
__global__ void sym_merge_kernel(double* __restrict__ a, int lda, const double* __restrict__ t, int ldt, int nr, int Px, int px,
                                 int Py, int py) {
  const int lc = blockIdx.y, gj = lc * Py + py;
  for (int lr = blockIdx.x * blockDim.x + threadIdx.x; lr < nr; lr += gridDim.x * blockDim.x)
    if (lr * Px + px > gj) a[(size_t)lc * lda + lr] = t[(size_t)lc * ldt + lr];
}
// b(:, lc) = z(:, lc) * w(global column)^(-1/2) on the local block   (diag_mult, src/KMATH_EIGEN_GEV_misc.F:49-104)
__global__ void scale_cols_rsqrt_cyclic_kernel(const double* __restrict__ z, int ldz, const double* __restrict__ w,
                                               double* __restrict__ b, int ldb, int nr, int Py, int py) {
  const int lc = blockIdx.y;
  const double sc = 1.0 / sqrt(w[lc * Py + py]);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x)
    b[(size_t)lc * ldb + i] = z[(size_t)lc * ldz + i] * sc;
}
// SUMMA panels.  A side: my columns lc0 .. lc0 + kbl - 1 of the panel, rows padded to nrp: out[c * nrp + r]
__global__ void mm_pack_a_kernel(const double* __restrict__ a, int lda, int nr, int nc, int lc0, int nrp, double* __restrict__ out) {
  const int c = blockIdx.y, lc = lc0 + c;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrp; r += gridDim.x * blockDim.x)
    out[(size_t)c * nrp + r] = (r < nr && lc < nc) ? a[(size_t)lc * lda + r] : 0.0;
}
// B side: my rows lr0 .. lr0 + kbl - 1 of the panel for every local column j: out[j * kbl + rr]
__global__ void mm_pack_b_kernel(const double* __restrict__ b, int ldb, int nr, int nc, int lr0, int kbl, double* __restrict__ out) {
  const int j = blockIdx.y;
  for (int rr = blockIdx.x * blockDim.x + threadIdx.x; rr < kbl; rr += gridDim.x * blockDim.x)
    out[(size_t)j * kbl + rr] = (j < nc && lr0 + rr < nr) ? b[(size_t)j * ldb + lr0 + rr] : 0.0;
}
// gathered B rows [q'][j][rr] (k = k0 + rr Px + q') -> panel matrix Bp(pos, j) in the k order of the gathered A columns:
// k - k0 = c Py + q  ->  pos = q kbl_y + c
__global__ void mm_unpack_b_kernel(const double* __restrict__ recv, int Px, int Py, int kbl_x, int kbl_y, int ncp, int kb,
                                   double* __restrict__ Bp) {
  const int j = blockIdx.y, q = blockIdx.z;
  for (int rr = blockIdx.x * blockDim.x + threadIdx.x; rr < kbl_x; rr += gridDim.x * blockDim.x) {
    const int dk = rr * Px + q;
    Bp[(size_t)j * kb + (size_t)(dk % Py) * kbl_y + dk / Py] = recv[((size_t)q * ncp + j) * kbl_x + rr];
  }
}
// C(ldc, nc) = A B on the cyclic blocks (all n x n, A and B complete -- not triangles); synchronous
static int dist_gemm_nn(Context& ctx, int n, const double* A, int lda, const double* B, int ldb, double* C, int ldc) {
  const Grid& G = ctx.grid;
  hipStream_t st = ctx.stream;
  const int nr = local_count(n, G.Px, G.px), nc = local_count(n, G.Py, G.py);
  const int L = G.Px / gcd_int(G.Px, G.Py) * G.Py;
  const int unit = 2 * L;                                    // panels start at multiples of Px and Py; even widths
  // panel width: about n / 8 between 128 and 1024 (the panels are O(n kb / sqrt(P)) of workspace)
  const int kb_want = (n / 8 < 128) ? (n < 128 ? n : 128) : (n / 8 > 1024 ? 1024 : n / 8);
  const int kb = unit * ceil_div(kb_want, unit);
  const int kbl_x = kb / G.Px, kbl_y = kb / G.Py;
  const int nrp = ((nr > 2 ? nr : 2) + 1) & ~1, ncp = nc > 1 ? nc : 1;
  double* sendA = ctx.pool.get_t<double>("gev.sa", (size_t)nrp * kbl_y);
  double* Ap = ctx.pool.get_t<double>("gev.pa", (size_t)nrp * kb);
  double* sendB = ctx.pool.get_t<double>("gev.sb", (size_t)kbl_x * ncp);
  double* recvB = ctx.pool.get_t<double>("gev.rb", (size_t)kb * ncp);
  double* Bp = ctx.pool.get_t<double>("gev.pb", (size_t)kb * ncp);
  for (int k0 = 0; k0 < n; k0 += kb) {
    hipLaunchKernelGGL(mm_pack_a_kernel, dim3(ceil_div(nrp, 256), kbl_y), dim3(256), 0, st, A, lda, nr, nc, k0 / G.Py, nrp, sendA);
    comm_allgather(ctx, COMM_Y, sendA, Ap, (size_t)nrp * kbl_y, st);          // Ap(:, q kbl_y + c) = A(my rows, k0 + c Py + q)
    hipLaunchKernelGGL(mm_pack_b_kernel, dim3(ceil_div(kbl_x, 256), ncp), dim3(256), 0, st, B, ldb, nr, nc, k0 / G.Px, kbl_x, sendB);
    comm_allgather(ctx, COMM_X, sendB, recvB, (size_t)kbl_x * ncp, st);
    hipLaunchKernelGGL(mm_unpack_b_kernel, dim3(ceil_div(kbl_x, 256), ncp, G.Px), dim3(256), 0, st, (const double*)recvB, G.Px, G.Py,
                       kbl_x, kbl_y, ncp, kb, Bp);
    if (nr > 0 && nc > 0) dgemm_dev(st, 'N', 'N', nr, nc, kb, 1.0, Ap, nrp, Bp, kb, k0 == 0 ? 0.0 : 1.0, C, ldc);
  }
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  return comm_failed(ctx) ? EIGX_ERR_INTERNAL : EIGX_OK;
}

// Same sequence as KMATH_EIGEN_GEV_1 (src/KMATH_EIGEN_GEV_1.F:57-139): eigen_s(B, 'X') -> B^(-1/2) := Z_B W_B^(-1/2);
// A' = B^(-1/2)^T A B^(-1/2) by two GEMMs; eigen_s(A', 'X') -> w, Y; Z = B^(-1/2) Y (B-orthonormal).  On entry only
// the upper triangles of a and b are significant; a, b are destroyed (a holds Y, b holds B^(-1/2) on exit, as in
// the reference).  One GPU; all three products run on the fp64 MFMA GEMM.
int gev_dev(Context& ctx, int n, double* a, int lda, double* b, int ldb, double* w, double* z, int ldz);

// Several ranks: the same sequence on the 2-D cyclic blocks, nothing gathered -- two distributed eigen_s solves, the
// symmetrisation of A and the transposed factor by dist_transpose, three SUMMA products with local MFMA GEMMs
// (round 4; the first version gathered A and B on every rank).
static int gev_dev_mg(Context& ctx, int n, double* a, int lda, double* b, int ldb, double* w, double* z, int ldz) {
  const Grid G = ctx.grid;
  const int nr = local_count(n, G.Px, G.px), nc = local_count(n, G.Py, G.py);
  if (n <= 0 || !a || !b || !w || !z || lda < (nr > 1 ? nr : 1) || ldb < (nr > 1 ? nr : 1) || ldz < (nr > 1 ? nr : 1))
    return EIGX_ERR_BAD_ARG;
  EIGX_HIP_CHECK(hipSetDevice(ctx.device));
  EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));
  hipStream_t st = ctx.stream;
  const double t0 = now_s();
  const int ldt = pad_ld((nr > 2 ? nr : 2));
  const int ncd = nc > 0 ? nc : 1;
  double* tb = ctx.pool.get_t<double>("gev.t", (size_t)ldt * ncd);    // A^T, later (B^(-1/2))^T
  double* cb = ctx.pool.get_t<double>("gev.c", (size_t)ldt * ncd);    // C = A B^(-1/2)
  dist_transpose(ctx, n, a, lda, tb, ldt, st);
  if (nr > 0 && nc > 0)
    hipLaunchKernelGGL(sym_merge_kernel, dim3(ceil_div(nr, 256), nc), dim3(256), 0, st, a, lda, (const double*)tb, ldt, nr, G.Px, G.px,
                       G.Py, G.py);
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  if (comm_failed(ctx)) return EIGX_ERR_INTERNAL;
  int rc = solve_dev(ctx, n, n, b, ldb, w, z, ldz, 128, 128, 'X', 1, 1);      // B = Z_B W_B Z_B^T
  if (rc != EIGX_OK) return rc;
  const double t1 = now_s();
  double wmin = 0.0;
  EIGX_HIP_CHECK(hipMemcpy(&wmin, w, 8, hipMemcpyDeviceToHost));
  if (!(wmin > 0.0)) {   // w is replicated bit for bit: every rank takes the same way out
    if (G.rank == 0) fprintf(stderr, "[eigx] Matrix B is not positive definite!\n");            // src/KMATH_EIGEN_GEV_1.F:75-80
    return EIGX_ERR_NOT_SPD;
  }
  if (nr > 0 && nc > 0)
    hipLaunchKernelGGL(scale_cols_rsqrt_cyclic_kernel, dim3(ceil_div(nr, 256), nc), dim3(256), 0, st, (const double*)z, ldz,
                       (const double*)w, b, ldb, nr, G.Py, G.py);
  rc = dist_gemm_nn(ctx, n, a, lda, b, ldb, cb, ldt);                          // C  = A B^(-1/2)
  if (rc != EIGX_OK) return rc;
  dist_transpose(ctx, n, b, ldb, tb, ldt, st);                                 // (B^(-1/2))^T
  rc = dist_gemm_nn(ctx, n, tb, ldt, cb, ldt, z, ldz);                         // A' = B^(-1/2)^T C
  if (rc != EIGX_OK) return rc;
  const double t2 = now_s();
  rc = solve_dev(ctx, n, n, z, ldz, w, a, lda, 128, 128, 'X', 1, 1);            // A' = Y W Y^T, Y in a
  if (rc != EIGX_OK) return rc;
  const double t3 = now_s();
  rc = dist_gemm_nn(ctx, n, b, ldb, a, lda, z, ldz);                           // Z = B^(-1/2) Y
  if (rc != EIGX_OK) return rc;
  const double t4 = now_s();
  ctx.timers[0] = t4 - t0; ctx.timers[1] = t1 - t0; ctx.timers[2] = t2 - t1; ctx.timers[3] = t3 - t2; ctx.timers[4] = t4 - t3;
  return EIGX_OK;
}

int gev_dev(Context& ctx, int n, double* a, int lda, double* b, int ldb, double* w, double* z, int ldz) {
  if (!ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (ctx.grid.nranks != 1) return gev_dev_mg(ctx, n, a, lda, b, ldb, w, z, ldz);
  if (n <= 0 || !a || !b || !w || !z || lda < n || ldb < n || ldz < n || ((lda | ldb | ldz) & 1)) return EIGX_ERR_BAD_ARG;
  EIGX_HIP_CHECK(hipSetDevice(ctx.device));
  EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));
  hipStream_t st = ctx.stream;
  const double t0 = now_s();
  hipLaunchKernelGGL(symmetrize_kernel, dim3(8, n), dim3(256), 0, st, a, lda, n);
  int rc = solve_dev(ctx, n, n, b, ldb, w, z, ldz, 128, 128, 'X', 1, 1);      // B = Z_B W_B Z_B^T
  if (rc != EIGX_OK) return rc;
  const double t1 = now_s();
  double wmin = 0.0;
  EIGX_HIP_CHECK(hipMemcpy(&wmin, w, 8, hipMemcpyDeviceToHost));
  if (!(wmin > 0.0)) {
    fprintf(stderr, "[eigx] Matrix B is not positive definite!\n");            // src/KMATH_EIGEN_GEV_1.F:75-80
    return EIGX_ERR_NOT_SPD;
  }
  hipLaunchKernelGGL(scale_cols_rsqrt_kernel, dim3(8, n), dim3(256), 0, st, z, ldz, w, b, ldb, n);
  const int ldc = pad_ld(n);
  double* c = ctx.pool.get_t<double>("gev.c", (size_t)ldc * n);
  dgemm_dev(st, 'N', 'N', n, n, n, 1.0, a, lda, b, ldb, 0.0, c, ldc);          // C  = A B^(-1/2)
  dgemm_dev(st, 'T', 'N', n, n, n, 1.0, b, ldb, c, ldc, 0.0, z, ldz);          // A' = B^(-1/2)^T C
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  const double t2 = now_s();
  rc = solve_dev(ctx, n, n, z, ldz, w, a, lda, 128, 128, 'X', 1, 1);            // A' = Y W Y^T, Y in a
  if (rc != EIGX_OK) return rc;
  const double t3 = now_s();
  dgemm_dev(st, 'N', 'N', n, n, n, 1.0, b, ldb, a, lda, 0.0, z, ldz);          // Z = B^(-1/2) Y
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  const double t4 = now_s();
  ctx.timers[0] = t4 - t0; ctx.timers[1] = t1 - t0; ctx.timers[2] = t2 - t1; ctx.timers[3] = t3 - t2; ctx.timers[4] = t4 - t3;
  return EIGX_OK;
}

int gev_host(Context& ctx, int n, double* a, int lda, double* b, int ldb, double* w, double* z, int ldz) {
  if (!ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  // host arrays: the rank's 2-D cyclic blocks a(lda, *), b(ldb, *), z(ldz, *) (one rank: the whole matrices)
  const int nr = local_count(n, ctx.grid.Px, ctx.grid.px), nc = local_count(n, ctx.grid.Py, ctx.grid.py);
  if (n <= 0 || !a || !b || !w || !z || lda < nr || ldb < nr || ldz < nr) return EIGX_ERR_BAD_ARG;
  EIGX_HIP_CHECK(hipSetDevice(ctx.device));
  const int ldd = pad_ld(nr + 2);
  const int ncd = nc > 0 ? nc : 1;
  double* ad = ctx.pool.get_t<double>("host.a", (size_t)ldd * ncd);
  double* zd = ctx.pool.get_t<double>("host.z", (size_t)ldd * ncd);
  double* bd = ctx.pool.get_t<double>("host.b", (size_t)ldd * ncd);
  double* wd = ctx.pool.get_t<double>("host.w", (size_t)n);
  if (nr > 0 && nc > 0) {
    EIGX_HIP_CHECK(hipMemcpy2D(ad, (size_t)ldd * 8, a, (size_t)lda * 8, (size_t)nr * 8, (size_t)nc, hipMemcpyHostToDevice));
    EIGX_HIP_CHECK(hipMemcpy2D(bd, (size_t)ldd * 8, b, (size_t)ldb * 8, (size_t)nr * 8, (size_t)nc, hipMemcpyHostToDevice));
  }
  const int rc = gev_dev(ctx, n, ad, ldd, bd, ldd, wd, zd, ldd);
  if (rc != EIGX_OK) return rc;
  EIGX_HIP_CHECK(hipMemcpy(w, wd, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (nr > 0 && nc > 0) {
    EIGX_HIP_CHECK(hipMemcpy2D(z, (size_t)ldz * 8, zd, (size_t)ldd * 8, (size_t)nr * 8, (size_t)nc, hipMemcpyDeviceToHost));
    EIGX_HIP_CHECK(hipMemcpy2D(a, (size_t)lda * 8, ad, (size_t)ldd * 8, (size_t)nr * 8, (size_t)nc, hipMemcpyDeviceToHost));
    EIGX_HIP_CHECK(hipMemcpy2D(b, (size_t)ldb * 8, bd, (size_t)ldd * 8, (size_t)nr * 8, (size_t)nc, hipMemcpyDeviceToHost));
  }
  return EIGX_OK;
}

}  // namespace

int64_t solver_workspace_bytes(const Context& ctx, int n, int lda, int ldz, int mf, int mb) {
  (void)lda; (void)ldz;
  if (mf <= 0) mf = 128;
  if (mb <= 0) mb = 128;
  if (mf > 256) mf = 256;
  const int P = ctx.grid.nranks;
  const int64_t ldn = pad_ld(n);
  const int64_t ldp = pad_ld((n + 127) / 128 * 128 + 128);
  if (P == 1) {
    const int64_t nn = ldn * n;
    // D&C: Qa, Qb, S, S2 ; reduction: panels + partials ; back-transform: V, W, X
    return 8 * (4 * nn + (int64_t)(n + 256) * (3 * mf + 2 * (n / 128 + 2) * 2 + 8) + (int64_t)(n + 512) * mb +
                2 * (int64_t)mb * n);
  }
  // Several GPUs: everything of size n^2 is divided by P (the caller's a and z blocks are n^2/P each as well):
  //   D&C       Qa, Qb row blocks (2 n^2/P), the eigenvector-row chunk buffer (2048 n), Z column block (n^2/P),
  //   exchanges send + receive pieces of the two all-to-alls (2 n^2/P, the peer window with its growth slack),
  //   reduction replicated panels [U|W|U] + gathered panel (ldp (4 m + 2)), tile partial sums, compact panels,
  //             step window (2 P messages of 2 (nx + ny) doubles), panel-gather window,
  //   back-transformation: reflector group (2048 columns), its gather window, W / X / T / Gram blocks.
  const int64_t rp = (n + P - 1) / P, zc = rp;
  const int64_t nxs = (n + ctx.grid.Px - 1) / ctx.grid.Px + 8, nys = (n + ctx.grid.Py - 1) / ctx.grid.Py + 8;
  const int64_t maxseg = (nxs > nys ? nxs : nys) / 128 + 3;
  int64_t w = 0;
  w += 2 * (int64_t)pad_ld((int)rp + 2) * n + (int64_t)n * 2048 + ldn * zc;              // D&C + Z block
  w += (int64_t)(2.6 * (double)((nxs + 8) * ((zc / ctx.grid.Py) + 2) * P)) + 3 * rp * zc;     // the two all-to-alls
  w += (int64_t)(1.5 * (double)(rp * zc < ((int64_t)32 << 20) ? rp * zc : ((int64_t)32 << 20))) + 64;   // their bounce window
  w += ldp * (4 * mf + 2) + 2 * maxseg * 2 * ldp + 3 * ldp + (nxs + nys + 128) * 2 * mf;   // reduction panels / partials
  w += (int64_t)(1.5 * (double)(2 * P * (2 * (nxs + nys) + 8))) + (int64_t)(2.5 * (double)(P + 1) * (mf / ctx.grid.Py + 3) * nxs);
  w += (int64_t)pad_ld(n + 1024) * 2048 + (int64_t)(2.5 * (double)(P + 1) * (2048 / ctx.grid.Py + 2) * nxs);   // reflector groups
  w += 2 * (int64_t)512 * zc + 6 * (int64_t)512 * 512 * ((n + 511) / 512) / 4 + 8 * (int64_t)n;               // W, X, T, Gram
  w += (int64_t)(1.5 * 8 * 3 * n) + 16 * (int64_t)n;                                                              // small allreduces, D&C vectors
  return 8 * w;
}



}  // namespace eigx

using namespace eigx;

extern "C" {

int eigx_transpose_plan(int Px, int Py, int px, int py, int qx, int qy, int* send_i0, int* send_j0, int* recv_i0,
                        int* recv_j0, int* step) {
  if (Px < 1 || Py < 1 || px < 0 || px >= Px || py < 0 || py >= Py || qx < 0 || qx >= Px || qy < 0 || qy >= Py || !send_i0 ||
      !send_j0 || !recv_i0 || !recv_j0 || !step) return EIGX_ERR_BAD_ARG;
  transpose_plan(Px, Py, px, py, qx, qy, send_i0, send_j0, recv_i0, recv_j0, step);
  return EIGX_OK;
}

int eigx_sx(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb, char mode) {
  return eigx_guard(g_ctx, [&] { return solve_host(g_ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode, 2, 1); });
}
int eigx_s(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb, char mode) {
  return eigx_guard(g_ctx, [&] { return solve_host(g_ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode, 1, 1); });
}
int eigx_sx_dev(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb, char mode) {
  return eigx_guard(g_ctx, [&] { return solve_dev(g_ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode, 2, 1); });
}
int eigx_s_dev(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb, char mode) {
  return eigx_guard(g_ctx, [&] { return solve_dev(g_ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode, 1, 1); });
}
// block-cyclic (ScaLAPACK descriptor MB = NB = nb) local blocks in and out; route 2 = eigen_sx, 1 = eigen_s
int eigx_solve_bc(int route, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int nb, int mf, int mb,
                  char mode) {
  if (route != 1 && route != 2) return EIGX_ERR_BAD_ARG;
  return eigx_guard(g_ctx, [&] { return solve_host(g_ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode, route, nb); });
}
int eigx_solve_bc_dev(int route, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int nb, int mf,
                      int mb, char mode) {
  if (route != 1 && route != 2) return EIGX_ERR_BAD_ARG;
  return eigx_guard(g_ctx, [&] { return solve_dev(g_ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode, route, nb); });
}
int eigx_numroc(int n, int nb, int iproc, int nprocs) {
  if (n < 0 || nb < 1 || nprocs < 1 || iproc < 0 || iproc >= nprocs) return -1;
  return numroc(n, nb, iproc, nprocs);
}

int eigx_band_reduce_dev(int n, double* a, int lda, double* d, double* e, int lde, int mf, int band) {
  if (!g_ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  // several ranks: collective; a = this rank's 2-D cyclic block a(lda, *), d / e replicated
  const int nloc = local_count(n, g_ctx.grid.Px, g_ctx.grid.px);
  if (n <= 0 || lda < (nloc > 1 ? nloc : 1) || (lda & 1) || lde < n || (band != 1 && band != 2)) return EIGX_ERR_BAD_ARG;
  return eigx_guard(g_ctx, [&] {
    EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));   // the caller's default-stream work on the arguments (see solve_dev)
    band_reduce_dev(g_ctx, n, a, lda, d, e, lde, mf > 0 ? mf : 128, band);
    EIGX_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
    return (g_ctx.grid.nranks > 1 && comm_failed(g_ctx)) ? EIGX_ERR_INTERNAL : EIGX_OK;
  });
}

int eigx_band_dc_dev(int n, int nvec, const double* d, const double* e, int lde, int band, double* w, double* z,
                     int ldz) {
  if (!g_ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (n <= 0 || nvec < 0 || nvec > n || lde < n || (band != 1 && band != 2) || (nvec > 0 && ldz < n))
    return EIGX_ERR_BAD_ARG;
  if (g_ctx.grid.nranks != 1) return EIGX_ERR_INTERNAL;
  EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));
  band_dc_dev(g_ctx, n, nvec, d, e, lde, band, w, z, ldz);
  return EIGX_OK;
}

int eigx_gev(int n, double* a, int lda, double* b, int ldb, double* w, double* z, int ldz) {
  return eigx_guard(g_ctx, [&] { return gev_host(g_ctx, n, a, lda, b, ldb, w, z, ldz); });
}
int eigx_gev_dev(int n, double* a, int lda, double* b, int ldb, double* w, double* z, int ldz) {
  return eigx_guard(g_ctx, [&] { return gev_dev(g_ctx, n, a, lda, b, ldb, w, z, ldz); });
}

int eigx_band_bisect_dev(int n, const double* d, const double* e, int lde, int band, double* w) {
  if (!g_ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (n <= 0 || lde < n || (band != 1 && band != 2) || !d || !e || !w) return EIGX_ERR_BAD_ARG;
  EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));
  band_bisect_dev(g_ctx, n, d, e, lde, band, w);
  return EIGX_OK;
}

int eigx_trbak_dev(int n, int nvec, const double* a, int lda, double* z, int ldz, const double* e, int lde, int mb,
                   int band) {
  if (!g_ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (n <= 0 || nvec < 0 || lda < n || ldz < n || lde < n || (band != 1 && band != 2)) return EIGX_ERR_BAD_ARG;
  if (g_ctx.grid.nranks != 1) return EIGX_ERR_INTERNAL;
  EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));
  trbak_dev(g_ctx, n, nvec, const_cast<double*>(a), lda, z, ldz, e, lde, mb > 0 ? mb : 128, band);
  EIGX_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  return EIGX_OK;
}

}  // extern "C"
