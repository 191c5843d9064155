// herm.hip -- eigen_h: complex Hermitian eigensolver.  One GPU; with more ranks the cyclic blocks are
// gathered, every rank reduces the replicated problem, and the back-transformation is shared by eigenvector columns.
//
// Replaces eigen_h (src/eigen_h.F:30-322): eigen_scaling_h -> eigen_hrd (Hermitian -> REAL symmetric tridiagonal,
// src/eigen_hrd.F:1-448) -> dc2 (the real tridiagonal D&C of dc.hip, unchanged) -> eigen_hrbakwyx (complex WY
// back-transformation, src/hrbakwy4.F:1-720, src/hrbakwy4_body.F:1-553).
//
// MI355X design: complex data lives in SPLIT PLANES (re, im) in HBM, so every O(n^3) contraction is a handful of real
// fp64 MFMA GEMMs of gemm_f64.hip instead of a ZGEMM port:
//   trailing update  A -= U W^H + W U^H :  Ar -= [Ur Ui Wr Wi][Wr Wi Ur Ui]^T ,  Ai -= [Ui -Ur Wi -Wr][Wr Wi Ur Ui]^T  (K = 4m)
//   back-transform   Z -= V (T (V^H Z)), T = S^-H : 4 real GEMMs each for V^H Z, T Y and V X, 4 small ones for the Gram matrix
// The reduction keeps the upper triangle of A up to date (the trailing update computes the tiles that meet it), and the
// Hermitian mat-vec streams only that triangle (128 x 128 register tiles).  Each column is formed lazily from the panel (dlatrd style), one column per step as the reference does:
//   x = A_eff(0:L, i);  g = -sign(||x||, Re x_{L-1});  u = x, u_{L-1} -= g;  beta = -u_{L-1} g       (src/eigen_hrd_t4.F:40-95)
//   q = A_eff u;  s = u^H q;  alpha = s / (2 beta);  v = (q - alpha u) / conj(beta)                 (src/eigen_hrd_t6_3.F:256-272)
//   A_eff = A - U W^H - W U^H  (panel of m columns, applied every m steps)                          (src/eigen_hrd_t1.F:2-110)
// Two launches per column (finish the previous column + form x | reflector scalars + panel dots + tiled mat-vec): still
// latency-bound per step; the fused / tiled structure of band_reduce.hip is the template.  All cross-workgroup reductions
// are two-phase and deterministic (no atomics).
#include "eigx_context.h"
#include "eigx_common.h"
#include "eigx_comm.h"
#include "../../include/eigenexa_amd.h"
#include <cfloat>
#include <chrono>
#include <limits>
#include <type_traits>
#include <vector>

namespace eigx {

namespace {

constexpr int HT = 256;      // threads per workgroup
constexpr int HS = 512;      // threads of a step workgroup (K1): 64 rows x 8 waves
constexpr int HSW = HS / 64;
constexpr int HM = 128;      // max panel width (LDS arrays)
constexpr int HTL = 128;     // tile edge of the Hermitian mat-vec
constexpr int HTH = 256;     // threads of a mat-vec workgroup (4 waves x 32 tile columns, in units of 4 columns)
constexpr int PDR = 1024;    // rows per panel-dot chunk
constexpr int HMB = 128;     // reflectors per back-transformation block (the packed triangle S^H of a block lives in LDS)

struct HArgs {
  double *Ar, *Ai; int ld; int n;
  double *Ur, *Ui, *Wr, *Wi; int ldp;
  double *xr, *xi;
  double *beta, *d, *e;
  double *pn;      // norm partials [workgroup]
  double *pd;      // panel-dot partials [chunk][HM][4]
  double *yrr, *yri, *ycr, *yci;  // mat-vec partials: row sums [tile column][ldp], column sums [tile row][ldp]
  double *ps;      // partials of u^H q [mat-vec tile][2]
  // several ranks, sharded reduction: this rank holds the tile columns tx (HTL columns each) with tx % P == p, stored
  // compactly (local tile column tx / P); P = 1: the whole matrix
  int P, p;
};
// element offset of global column c in the planes Ar / Ai
__device__ __host__ __forceinline__ size_t hcol(const HArgs& H, int c) {
  return (H.P == 1) ? (size_t)c * H.ld : (size_t)(((c >> 7) / H.P) * 128 + (c & 127)) * H.ld;
}
static_assert(HTL == 128, "hcol assumes 128-column tiles");

__device__ __forceinline__ double hwave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// deterministic block sum of K values over NW waves; result in every thread
template <int K, int NW>
__device__ __forceinline__ void hblock_sum_w(double (&v)[K], double* red /* NW*K */) {
#pragma unroll
  for (int q = 0; q < K; ++q) v[q] = hwave_sum(v[q]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < K; ++q) red[(threadIdx.x >> 6) * K + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < K; ++q) {
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) a += red[w * K + q];
    v[q] = a;
  }
}
// deterministic block sum of K values (256 threads); result in every thread
template <int K>
__device__ __forceinline__ void hblock_sum(double (&v)[K], double* red /* 4*K */) {
#pragma unroll
  for (int q = 0; q < K; ++q) v[q] = hwave_sum(v[q]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < K; ++q) red[(threadIdx.x >> 6) * K + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < K; ++q) v[q] = (red[q] + red[K + q]) + (red[2 * K + q] + red[3 * K + q]);
}

// interleaved complex(8) upper triangle -> split planes of the FULL Hermitian matrix (lower := conj(upper), real diagonal)
__global__ void h_split_kernel(const double* __restrict__ a, int lda, int n, double* __restrict__ Ar,
                               double* __restrict__ Ai, int ld) {
  const int j = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= j; i += gridDim.x * blockDim.x) {
    const double re = a[2 * ((size_t)i + (size_t)j * lda)];
    const double im = (i == j) ? 0.0 : a[2 * ((size_t)i + (size_t)j * lda) + 1];
    Ar[(size_t)i + (size_t)j * ld] = re; Ai[(size_t)i + (size_t)j * ld] = im;
    if (i < j) { Ar[(size_t)j + (size_t)i * ld] = re; Ai[(size_t)j + (size_t)i * ld] = -im; }
  }
}

// max |re|, |im| over the upper triangle and a non-finite flag (eigen_scaling_h, src/eigen_scaling_h.F)
__global__ __launch_bounds__(HT) void h_absmax_kernel(const double* __restrict__ a, int lda, int n, double* __restrict__ out) {
  __shared__ double red[8];
  double v[2] = {0.0, 0.0};
  for (int j = blockIdx.x; j < n; j += gridDim.x)
    for (int i = threadIdx.x; i <= j; i += HT) {
      const double re = fabs(a[2 * ((size_t)i + (size_t)j * lda)]);
      const double im = (i == j) ? 0.0 : fabs(a[2 * ((size_t)i + (size_t)j * lda) + 1]);
      if (!(re <= DBL_MAX) || !(im <= DBL_MAX)) v[1] = 1.0;
      else v[0] = fmax(v[0], fmax(re, im));
    }
  for (int o = 32; o > 0; o >>= 1) { v[0] = fmax(v[0], __shfl_xor(v[0], o, 64)); v[1] = fmax(v[1], __shfl_xor(v[1], o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = v[0]; red[(threadIdx.x >> 6) * 2 + 1] = v[1]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = fmax(fmax(red[0], red[2]), fmax(red[4], red[6]));
    out[2 * blockIdx.x + 1] = fmax(fmax(red[1], red[3]), fmax(red[5], red[7]));
  }
}

__global__ void h_scale_kernel(double* __restrict__ a, int lda, int n, double s) {   // interleaved, upper triangle
  const int j = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= j; i += gridDim.x * blockDim.x) {
    a[2 * ((size_t)i + (size_t)j * lda)] *= s;
    a[2 * ((size_t)i + (size_t)j * lda) + 1] *= s;
  }
}

__global__ void h_fill_kernel(double* p, size_t n, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// Reflector scalars of column i (L = i rows) from ||x||^2 (the norm partials of K1, added in one fixed order by every
// workgroup of the mat-vec and of K1: bit-identical everywhere, so no separate reflector kernel runs) and the pivot.
struct HRef { double nrm2, g, unr, uni, br, bi; };
// (anr, ani) = pivot x_{L-1}
__device__ __forceinline__ HRef h_ref_of(double nrm2, double anr, double ani) {
  HRef f;
  f.nrm2 = nrm2; f.g = 0.0; f.unr = 0.0; f.uni = 0.0; f.br = 1.0; f.bi = 0.0;
  if (f.nrm2 != 0.0) {
    const double mag = sqrt(f.nrm2);
    f.g = (anr >= 0.0) ? -mag : mag;
    f.unr = anr - f.g; f.uni = ani;
    f.br = -f.unr * f.g; f.bi = -f.uni * f.g;
  }
  return f;
}
// u_r = x_r, except the pivot row L-1 (x_{L-1} - g) ; zero vector for a trivial reflector.  (x_r, x_i) = x of row r,
// requested by the caller before the scalars were known (rows >= L - 1: anything)
__device__ __forceinline__ void h_u_of(const HRef& f, int L, int r, double x_r, double x_i, double& ur, double& ui) {
  if (f.nrm2 == 0.0 || r >= L) { ur = 0.0; ui = 0.0; }
  else if (r == L - 1) { ur = f.unr; ui = f.uni; }
  else { ur = x_r; ui = x_i; }
}

template <int NW>
__device__ __forceinline__ void h_paneldot_body(const HArgs& H, const HRef& f, int L, int k, int j, int c) {
  __shared__ double red[4 * NW];
  const int r1 = (c * PDR + PDR < L) ? c * PDR + PDR : L;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  // the first four waves add the rows (the order of the sum does not depend on NW: further waves contribute +0.0)
  for (int r = c * PDR + threadIdx.x; r < r1 && threadIdx.x < HTH; r += HTH) {
    double ur, ui;
    h_u_of(f, L, r, H.xr[r], H.xi[r], ur, ui);
    const double wr = H.Wr[(size_t)r + (size_t)j * H.ldp], wi = H.Wi[(size_t)r + (size_t)j * H.ldp];
    const double pr = H.Ur[(size_t)r + (size_t)j * H.ldp], pi = H.Ui[(size_t)r + (size_t)j * H.ldp];
    v[0] += wr * ur + wi * ui; v[1] += wr * ui - wi * ur;   // conj(w) u
    v[2] += pr * ur + pi * ui; v[3] += pr * ui - pi * ur;   // conj(U_j) u
  }
  hblock_sum_w<4, NW>(v, red);
  if (threadIdx.x == 0) {
    double* o = H.pd + ((size_t)c * HM + j) * 4;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
}

// lane exchanges on the VALU (no LDS crossbar), as in band_reduce.hip: DPP moves inside a row of 16 lanes, gfx950
// v_permlane{16,32}_swap across rows
template <int CTRL>
__device__ __forceinline__ double hdpp_mov(double v) {
  const int lo2 = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi2, lo2);
}
// lanes 0..31 get a[i] + a[i+32], lanes 32..63 get b[i-32] + b[i]
__device__ __forceinline__ double hswapadd32(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// even rows of 16 lanes get a[row] + a[row+1], odd rows get b[row-1] + b[row]
__device__ __forceinline__ double hswapadd16(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// 16 per-lane values (one per tile column of the wave) -> their sums over the 64 lanes.  Halving butterfly: every
// step pairs two columns and two lane groups, so 16 -> 8 -> 4 -> 2 -> 1 values per lane, then the four lanes of a quad
// are added.  Afterwards every lane holds the total of column  8*bit5 + 4*bit4 + 2*bit3 + bit2  (bits of the lane id).
__device__ __forceinline__ double hcolsum16(const double (&v)[16], int lane) {
  double v8[8], v4[4], v2[2], v1;
#pragma unroll
  for (int j = 0; j < 8; ++j) v8[j] = hswapadd32(v[j], v[j + 8]);        // lanes < 32: column j, others: j + 8
#pragma unroll
  for (int j = 0; j < 4; ++j) v4[j] = hswapadd16(v8[j], v8[j + 4]);      // even rows: j, odd rows: j + 4
  {
    const bool hi = lane & 8;                                           // row_ror:8 = lane ^ 8 inside the row of 16
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double keep = hi ? v4[j + 2] : v4[j];
      const double send = hi ? v4[j] : v4[j + 2];
      v2[j] = keep + hdpp_mov<0x128>(send);
    }
  }
  {
    const bool hi = lane & 4;                                           // row_half_mirror: lane i <-> 7 - i of its 8
    const double keep = hi ? v2[1] : v2[0];
    const double send = hi ? v2[0] : v2[1];
    v1 = keep + hdpp_mov<0x141>(send);
  }
  v1 += hdpp_mov<0xB1>(v1);                                             // quad_perm [1,0,3,2]
  v1 += hdpp_mov<0x4E>(v1);                                             // quad_perm [2,3,0,1]
  return v1;
}

// 4 per-lane values -> their sums over the 64 lanes (2 + 1 halving steps, then the 16 lanes of a row are added);
// afterwards every lane holds the total of column  2*bit5 + bit4  of its lane id
__device__ __forceinline__ double hcolsum4(const double (&v)[4], int lane) {
  double v2[2], v1;
#pragma unroll
  for (int j = 0; j < 2; ++j) v2[j] = hswapadd32(v[j], v[j + 2]);
  v1 = hswapadd16(v2[0], v2[1]);
  v1 += hdpp_mov<0xB1>(v1);
  v1 += hdpp_mov<0x4E>(v1);
  v1 += hdpp_mov<0x141>(v1);
  v1 += hdpp_mov<0x140>(v1);
  return v1;
}

// K3: q = A(0:L, 0:L) u from the UPPER triangle only (half the HBM bytes of a GEMV over both triangles): one workgroup
// per 128 x 128 tile (ty <= tx) of the upper block triangle, 1-D grid in row-major tile order.  The registers that hold a
// piece of the tile give the row sums  sum_c A(r,c) u(c)  (combined over the waves through LDS) and the column sums
// sum_r conj(A(r,c)) u(r)  of the mirrored lower-triangle block (halving butterfly over the lanes).  Row partials are
// indexed by tile column (YR[tx][r]), column partials by tile row (YC[ty][c]); K1 of the next column adds the nt + 1
// partials of a row in fixed order.  Every tile also writes its part of u^H q.
typedef double hd2_t __attribute__((ext_vector_type(2)));
// 128 x 128 tile per workgroup of NW = 4 waves; wave w owns the tile columns [32w, 32w+32) as eight units of 4 columns (NW = 8:
// [16w, 16w+16), four units), a lane owns
// the row pair (2 lane, 2 lane + 1) (16-byte loads).  Two units are in flight (register sets av0 / av1); the loop over
// unit pairs is rolled with a trip count the compiler does not know (`npairs`, always 4), otherwise it hoists every load to
// the top and the kernel needs all 256 VGPRs + AGPRs.
// NW = 4 or 8 waves per tile: with 8 a wave owns 16 tile columns (npairs = 2) and the chain of load round trips per tile is
// half as long -- for small active sizes, where a launch has fewer tiles than the chip has CUs and its time is that chain
// (11.7 us per launch below L = 2000 with 4 waves); one workgroup per CU then.
template <int NW>
__global__ __launch_bounds__(NW * 64) void h_hemv_kernel(HArgs H, int L, int k, int nt, int npdc, int npairs, int nparts,
                                                         int ntl) {
  __shared__ double ucr[HTL], uci[HTL], urr[HTL], uri[HTL];
  __shared__ double part[NW][HTL][2];
  __shared__ double sred[NW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: column offsets stay on the scalar unit
  // the first k * npdc workgroups of the launch are the panel-dot workgroups (column j, row chunk c)
  const bool pdot = (int)blockIdx.x < k * npdc;
  // tile index -> (ty, tx), row-major over the upper block triangle
  const int bid = pdot ? 0 : (int)blockIdx.x - k * npdc;
  // (ntl = tiles of this launch; a rank that owns no active tile still sends ONE workgroup for the scalars below)
  const bool idle = !pdot && bid >= ntl;
  int ty = 0, tx = 0;
  if (H.P == 1) {
    const float fn = 2.0f * (float)nt + 1.0f;
    ty = (int)((fn - sqrtf(fn * fn - 8.0f * (float)bid)) * 0.5f);
    if (ty < 0) ty = 0;
    if (ty > nt - 1) ty = nt - 1;
    while (ty > 0 && ty * nt - ty * (ty - 1) / 2 > bid) --ty;
    while ((ty + 1) * nt - (ty + 1) * ty / 2 <= bid) ++ty;
    tx = ty + (bid - (ty * nt - ty * (ty - 1) / 2));
  } else if (!idle && !pdot) {
    // sharded: my tile columns tx = g P + p < nt, column by column; column tx holds the tiles ty = 0 .. tx
    int g = 0, rest = bid;
    for (;;) { const int cnt = g * H.P + H.p + 1; if (rest < cnt) break; rest -= cnt; ++g; }
    tx = g * H.P + H.p; ty = rest;
  }
  const int row0 = ty * HTL, col0 = tx * HTL;
  const size_t cbase = hcol(H, col0);
  const bool diag = (ty == tx);
  const int l0 = 2 * lane, l1 = 2 * lane + 1;           // my rows inside the tile
  const int r0 = row0 + l0, r1 = row0 + l1;
  const int rc = (r0 < L) ? r0 : 0;
  const int wc0 = wave * (HTL / NW);                    // first tile column of this wave
  hd2_t av0r[4], av0i[4], av1r[4], av1i[4];
  auto load4 = [&](hd2_t (&vr)[4], hd2_t (&vi)[4], int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc_ = wc0 + g * 4 + j;
      const size_t co = cbase + (size_t)((col0 + cc_ < L) ? cc_ : 0) * H.ld;   // (beyond L: the tile's first column, masked below)
      vr[j] = *reinterpret_cast<const hd2_t*>(H.Ar + (size_t)rc + co);
      vi[j] = *reinterpret_cast<const hd2_t*>(H.Ai + (size_t)rc + co);
    }
  };
  // Requests first, in the order they are needed: the norm partials and the pivot (reflector scalars), x at this
  // thread's column / row of the tile (u), the first unit of the tile -- one round trip for the whole prologue
  double nr[1] = {0.0};
  if (tid < HT) for (int q = tid; q < nparts; q += HT) nr[0] += H.pn[q];   // the sum K1 forms (further waves add +0.0)
  const double anr = H.xr[L - 1], ani = H.xi[L - 1];
  const int myi = (tid < HTL) ? col0 + tid : row0 + ((tid - HTL) & (HTL - 1));
  const int myc = (myi < L) ? myi : L - 1;
  const double mxr = H.xr[myc], mxi = H.xi[myc];
  if (!pdot && !idle) load4(av0r, av0i, 0);
  // reflector scalars of column i (L = i rows) from the norm partials of K1: every workgroup of the mat-vec and of K1
  // recomputes them in the same order (bit-identical everywhere), so no separate reflector kernel runs
  hblock_sum_w<1, NW>(nr, sred);
  const HRef f = h_ref_of(nr[0], anr, ani);
  if (blockIdx.x == 0 && tid == 0) { H.beta[2 * L] = f.br; H.beta[2 * L + 1] = f.bi; H.e[L] = f.g; }   // column i = L
  if (pdot) {
    h_paneldot_body<NW>(H, f, L, k, (int)blockIdx.x % k, (int)blockIdx.x / k);
    return;
  }
  if (idle) return;
  {
    double a_, b_;
    h_u_of(f, L, myi, mxr, mxi, a_, b_);
    if (tid < HTL) { ucr[tid] = a_; uci[tid] = b_; }
    else if (tid < 2 * HTL) { urr[tid - HTL] = a_; uri[tid - HTL] = b_; }
  }
  __syncthreads();
  const double mur0 = urr[l0], mui0 = uri[l0], mur1 = urr[l1], mui1 = uri[l1];   // u at my rows
  double sr0 = 0.0, si0 = 0.0, sr1 = 0.0, si1 = 0.0;   // row sums over the wave's 32 columns
  double sq[2] = {0.0, 0.0};                            // this tile's part of u^H q (K1 of the next column adds the tiles)
  // interior tiles (strictly above the diagonal, all rows and columns < L) need no masks: most tiles of a launch
  const bool interior = !diag && row0 + HTL <= L && col0 + HTL <= L;
  auto compute4 = [&](auto full_tag, const hd2_t (&vr_)[4], const hd2_t (&vi_)[4], int g) {
    constexpr bool FULL = decltype(full_tag)::value;
    double cr[4], ci[4];
    if constexpr (FULL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = wc0 + g * 4 + j;
        const double vr = ucr[cc], vi = uci[cc];
        sr0 += vr_[j].x * vr - vi_[j].x * vi; si0 += vr_[j].x * vi + vi_[j].x * vr;
        sr1 += vr_[j].y * vr - vi_[j].y * vi; si1 += vr_[j].y * vi + vi_[j].y * vr;
        cr[j] = (vr_[j].x * mur0 + vi_[j].x * mui0) + (vr_[j].y * mur1 + vi_[j].y * mui1);
        ci[j] = (vr_[j].x * mui0 - vi_[j].x * mur0) + (vr_[j].y * mui1 - vi_[j].y * mur1);
      }
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = wc0 + g * 4 + j, c = col0 + cc;
      const bool in0 = r0 < L && c < L, in1 = r1 < L && c < L;
      const bool up0 = in0 && (!diag || l0 <= cc), up1 = in1 && (!diag || l1 <= cc);   // upper incl. diagonal -> row sums
      const bool su0 = in0 && (!diag || l0 < cc), su1 = in1 && (!diag || l1 < cc);     // strictly upper -> column sums
      const double vr = ucr[cc], vi = uci[cc];
      {
        const double xr = up0 ? vr_[j].x : 0.0, xi = (up0 && !(diag && l0 == cc)) ? vi_[j].x : 0.0;
        sr0 += xr * vr - xi * vi; si0 += xr * vi + xi * vr;
      }
      {
        const double xr = up1 ? vr_[j].y : 0.0, xi = (up1 && !(diag && l1 == cc)) ? vi_[j].y : 0.0;
        sr1 += xr * vr - xi * vi; si1 += xr * vi + xi * vr;
      }
      const double y0r = su0 ? vr_[j].x : 0.0, y0i = su0 ? vi_[j].x : 0.0;
      const double y1r = su1 ? vr_[j].y : 0.0, y1i = su1 ? vi_[j].y : 0.0;
      cr[j] = (y0r * mur0 + y0i * mui0) + (y1r * mur1 + y1i * mui1);                   // conj(a) u(row)
      ci[j] = (y0r * mui0 - y0i * mur0) + (y1r * mui1 - y1i * mur1);
    }
    }
    const double tcr = hcolsum4(cr, lane), tci = hcolsum4(ci, lane);
    if ((lane & 15) == 0) {
      const int j = ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);
      const int cc = wc0 + g * 4 + j, c = col0 + cc;
      if (c < L) {
        H.ycr[(size_t)ty * H.ldp + c] = tcr; H.yci[(size_t)ty * H.ldp + c] = tci;
        sq[0] += tcr * ucr[cc] + tci * uci[cc];    // q conj(u)
        sq[1] += tci * ucr[cc] - tcr * uci[cc];
      }
    }
  };
  // software pipeline: av0 holds unit g (loaded one iteration ahead), av1 unit g + 1
  if (interior) {
#pragma unroll 1
    for (int p = 0; p < npairs; ++p) {
      const int g = 2 * p;
      load4(av1r, av1i, g + 1);
      compute4(std::true_type{}, av0r, av0i, g);
      if (p + 1 < npairs) load4(av0r, av0i, g + 2);
      compute4(std::true_type{}, av1r, av1i, g + 1);
    }
  } else {
#pragma unroll 1
    for (int p = 0; p < npairs; ++p) {
      const int g = 2 * p;
      load4(av1r, av1i, g + 1);
      compute4(std::false_type{}, av0r, av0i, g);
      if (p + 1 < npairs) load4(av0r, av0i, g + 2);
      compute4(std::false_type{}, av1r, av1i, g + 1);
    }
  }
  part[wave][l0][0] = sr0; part[wave][l0][1] = si0;
  part[wave][l1][0] = sr1; part[wave][l1][1] = si1;
  __syncthreads();
  if (tid < HTL && row0 + tid < L) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { a0 += part[w][tid][0]; a1 += part[w][tid][1]; }
    H.yrr[(size_t)tx * H.ldp + row0 + tid] = a0;
    H.yri[(size_t)tx * H.ldp + row0 + tid] = a1;
    sq[0] += a0 * urr[tid] + a1 * uri[tid];
    sq[1] += a1 * urr[tid] - a0 * uri[tid];
  }
  hblock_sum_w<2, NW>(sq, &part[0][0][0]);
  if (tid == 0) { H.ps[2 * bid] = sq[0]; H.ps[2 * bid + 1] = sq[1]; }
}

// K1 (one launch per column, plus one at every panel end): finishes the PREVIOUS column (Lp rows, panel slot kp) and
// forms the next one (column i), so a column costs two launches (this one and the mat-vec):
//   p = q - U (W^H u) - W (U^H u)            q = sum of the nt + 1 mat-vec partials of a row, d = panel dots of K3
//   s = u^H p = sum of the tile partials of u^H q (K3) - 2 Re sum_j conj(U_j^H u) (W_j^H u)
//   alpha = s / (2 beta),  v = (p - alpha u) / conj(beta)  -> W(:, kp);  u -> U(:, kp) and column Lp of A
//   x = A_eff(0:i, i) = A(:, i) - sum_{j <= kp} [U_j conj(W(i,j)) + W_j conj(U(i,j))],  d_i = Re A_eff(i,i),  partial ||x||^2
// s and row i of (p, u, v) are needed by every workgroup before its own rows: each workgroup reduces them itself, in the
// same order (bit-identical everywhere).  64 rows per workgroup of 8 waves; wave q takes every eighth partial sum and
// every eighth panel column of those rows (the chain of dependent loads per row, not the bytes, is what this kernel
// costs: everything is requested in one round trip before the first barrier), the
// waves are combined through LDS.  x overwrites the previous x in place: a row is read and written by the one thread
// that owns it, and the previous pivot row (= row i) is not part of the new x.
// Sharded reduction (several ranks): the mat-vec result arrives already summed over tiles and ranks (H.ycr / H.yci point at
// it, ntp = 0: one "partial" per row), likewise s (H.ps, nps = 1) and the raw column i of A (acr / aci: only its owner
// holds it); the reflector goes into column Lp of A on the rank that owns that column.
__global__ __launch_bounds__(HS) void h_step_kernel(HArgs H, int i, int do_x, int Lp, int kp, int ntp, int npdcp,
                                                    int npartsp, double* __restrict__ pn_out, int nps,
                                                    const double* __restrict__ acr, const double* __restrict__ aci) {
  __shared__ double dwr[HM], dwi[HM], dur[HM], dui[HM];
  __shared__ double cwr[HM], cwi[HM], cur[HM], cui[HM];
  __shared__ double comb[HSW][64][4];
  __shared__ double sred[5 * HSW];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int r = blockIdx.x * 64 + lane;
  if (Lp <= 0) {   // first column of a panel: nothing to finish, A_eff = A
    if (q != 0) return;
    double nrm = 0.0;
    if (r <= i) {
      const double xr = acr ? acr[r] : H.Ar[(size_t)r + (size_t)i * H.ld], xi = aci ? aci[r] : H.Ai[(size_t)r + (size_t)i * H.ld];
      if (r < i) { H.xr[r] = xr; H.xi[r] = xi; nrm = xr * xr + xi * xi; }
      else H.d[i] = xr;
    }
    nrm = hwave_sum(nrm);
    if (lane == 0) pn_out[blockIdx.x] = nrm;
    return;
  }
  // ---- requests: everything the kernel needs from memory before the first barrier (one round trip for the common
  // sizes; a straight-line version with every load unconditional and clamped was 2 % slower: more load instructions)
  constexpr int PB = 6;                        // panel columns per wave that are requested up front (kp <= PB * HSW = 48)
  constexpr int QB = 9;                        // mat-vec partials per wave and batch (one batch up to L = 9088)
  const int rc = (r < Lp) ? r : Lp - 1;        // my row, clamped
  double axr = 0.0, axi = 0.0, anr = 0.0, ani = 0.0, xpr = 0.0, xpi = 0.0;
  if (q == 0) {
    if (do_x && r <= i) {
      axr = acr ? acr[r] : H.Ar[(size_t)r + (size_t)i * H.ld];
      axi = aci ? aci[r] : H.Ai[(size_t)r + (size_t)i * H.ld];
    }
    anr = H.xr[Lp - 1]; ani = H.xi[Lp - 1];
    xpr = H.xr[rc]; xpi = H.xi[rc];
  }
  double v5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};    // ||x_prev||^2, s (2), row i of p (2)
  if (tid < HT) for (int t = tid; t < npartsp; t += HT) v5[0] += H.pn[t];   // exactly the sum the mat-vec forms (the other waves add +0.0)
  if (tid < 4 * kp) {   // panel dots: thread (column j, component) adds the chunks, eight requests at a time
    const int j = tid >> 2, comp = tid & 3;
    double a = 0.0;
    for (int c0 = 0; c0 < npdcp; c0 += 8) {
      double t8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t8[e] = H.pd[((size_t)((c0 + e < npdcp) ? c0 + e : npdcp - 1) * HM + j) * 4 + comp];
#pragma unroll
      for (int e = 0; e < 8; ++e) a += (c0 + e < npdcp) ? t8[e] : 0.0;
    }
    (comp == 0 ? dwr : comp == 1 ? dwi : comp == 2 ? dur : dui)[j] = a;
  }
  double rur = 0.0, rui = 0.0, rwr = 0.0, rwi = 0.0;   // row i of panel column tid
  if (do_x && tid < kp) {
    rur = H.Ur[(size_t)i + (size_t)tid * H.ldp]; rui = H.Ui[(size_t)i + (size_t)tid * H.ldp];
    rwr = H.Wr[(size_t)i + (size_t)tid * H.ldp]; rwi = H.Wi[(size_t)i + (size_t)tid * H.ldp];
    cwr[tid] = rwr; cwi[tid] = -rwi; cur[tid] = rur; cui[tid] = -rui;   // conj of row i of the panel
  }
  for (int t = tid; t < nps; t += HS) { v5[1] += H.ps[2 * t]; v5[2] += H.ps[2 * t + 1]; }
  // partial u of row x: u <= R(x) -> column sums of tile (u, R); u > R -> row sums of tile (R, u-1)   (R = x / HTL)
  auto part_r = [&](int u, int x) { return (u <= x / HTL) ? H.ycr[(size_t)u * H.ldp + x] : H.yrr[(size_t)(u - 1) * H.ldp + x]; };
  auto part_i = [&](int u, int x) { return (u <= x / HTL) ? H.yci[(size_t)u * H.ldp + x] : H.yri[(size_t)(u - 1) * H.ldp + x]; };
  if (do_x) for (int u = tid; u < ntp + 1; u += HS) { v5[3] += part_r(u, i); v5[4] += part_i(u, i); }
  // wave q takes the panel columns j = q, q + HSW, ...; the first PB of them are requested here (a column past the end:
  // clamped index, value dropped -- no load behind a branch)
  double pur[PB], pui[PB], pwr[PB], pwi[PB];
#pragma unroll
  for (int e = 0; e < PB; ++e) {
    const int j = q + e * HSW, jc = (j < kp) ? j : 0;
    pur[e] = H.Ur[(size_t)rc + (size_t)jc * H.ldp]; pui[e] = H.Ui[(size_t)rc + (size_t)jc * H.ldp];
    pwr[e] = H.Wr[(size_t)rc + (size_t)jc * H.ldp]; pwi[e] = H.Wi[(size_t)rc + (size_t)jc * H.ldp];
  }
  // and the partials u = q, q + HSW, ...
  double pr = 0.0, pi = 0.0, cr = 0.0, ci = 0.0;
  for (int u0 = q; u0 < ntp + 1; u0 += QB * HSW) {
    double a[QB], b[QB];
#pragma unroll
    for (int e = 0; e < QB; ++e) {
      const int u = u0 + e * HSW, uc = (u < ntp + 1) ? u : ntp;
      a[e] = part_r(uc, rc); b[e] = part_i(uc, rc);
    }
#pragma unroll
    for (int e = 0; e < QB; ++e) { const bool in = u0 + e * HSW < ntp + 1; pr += in ? a[e] : 0.0; pi += in ? b[e] : 0.0; }
  }
  __syncthreads();     // panel dots and row i of the panel are in LDS
  if (tid < kp) {
    const double a0 = dwr[tid], a1 = dwi[tid], a2 = dur[tid], a3 = dui[tid];
    v5[1] += -2.0 * (a2 * a0 + a3 * a1);       // -2 Re conj(du) dw
    if (do_x) {
      v5[3] -= (rur * a0 - rui * a1) + (rwr * a2 - rwi * a3);
      v5[4] -= (rur * a1 + rui * a0) + (rwr * a3 + rwi * a2);
    }
  }
  {
    auto apply = [&](int j, double ur, double ui, double wr, double wi) {
      pr -= (ur * dwr[j] - ui * dwi[j]) + (wr * dur[j] - wi * dui[j]);
      pi -= (ur * dwi[j] + ui * dwr[j]) + (wr * dui[j] + wi * dur[j]);
      if (do_x) {
        cr += (ur * cwr[j] - ui * cwi[j]) + (wr * cur[j] - wi * cui[j]);
        ci += (ur * cwi[j] + ui * cwr[j]) + (wr * cui[j] + wi * cur[j]);
      }
    };
#pragma unroll
    for (int e = 0; e < PB; ++e) {
      const int j = q + e * HSW;
      if (j < kp) apply(j, pur[e], pui[e], pwr[e], pwi[e]);
    }
    for (int j = q + PB * HSW; j < kp; j += HSW)     // wider panels only
      apply(j, H.Ur[(size_t)rc + (size_t)j * H.ldp], H.Ui[(size_t)rc + (size_t)j * H.ldp], H.Wr[(size_t)rc + (size_t)j * H.ldp],
            H.Wi[(size_t)rc + (size_t)j * H.ldp]);
  }
  comb[q][lane][0] = pr; comb[q][lane][1] = pi; comb[q][lane][2] = cr; comb[q][lane][3] = ci;
  // the five scalars, added in the order the mat-vec uses (it recomputes the reflector scalars from the same
  // partials and must get the same bits); its barriers also publish comb
  hblock_sum_w<5, HSW>(v5, sred);
  if (q != 0) return;
  const HRef f = h_ref_of(v5[0], anr, ani);
  const double br = f.br, bi = f.bi, b2 = br * br + bi * bi;
  // alpha = s / (2 beta) = s conj(beta) / (2 |beta|^2);   t / conj(beta) = t beta / |beta|^2
  const double alr = (v5[1] * br + v5[2] * bi) / (2.0 * b2), ali = (v5[2] * br - v5[1] * bi) / (2.0 * b2);
  double nrm = 0.0;
  pr = 0.0; pi = 0.0; cr = 0.0; ci = 0.0;
#pragma unroll
  for (int w = 0; w < HSW; ++w) { pr += comb[w][lane][0]; pi += comb[w][lane][1]; cr += comb[w][lane][2]; ci += comb[w][lane][3]; }
  if (r < Lp) {
    double ur, ui;
    h_u_of(f, Lp, r, xpr, xpi, ur, ui);
    const double tr = pr - (alr * ur - ali * ui), ti = pi - (alr * ui + ali * ur);
    const double wr = (tr * br - ti * bi) / b2, wi = (tr * bi + ti * br) / b2;
    H.Wr[(size_t)r + (size_t)kp * H.ldp] = wr; H.Wi[(size_t)r + (size_t)kp * H.ldp] = wi;
    // the reflector goes into the panel and stays in column Lp of A (rows 0..Lp-1), as the reference leaves it
    H.Ur[(size_t)r + (size_t)kp * H.ldp] = ur; H.Ui[(size_t)r + (size_t)kp * H.ldp] = ui;
    if (H.P == 1 || ((Lp >> 7) % H.P) == H.p) { H.Ar[(size_t)r + hcol(H, Lp)] = ur; H.Ai[(size_t)r + hcol(H, Lp)] = ui; }
    if (do_x) {
      // row i of the column that is being finished: conj v(i), conj u(i)  (i = Lp - 1, the pivot row)
      double uir, uii;
      h_u_of(f, Lp, i, 0.0, 0.0, uir, uii);
      const double sr = v5[3] - (alr * uir - ali * uii), si = v5[4] - (alr * uii + ali * uir);
      const double cvr = (sr * br - si * bi) / b2, cvi = -((sr * bi + si * br) / b2);
      const double cur_ = uir, cui_ = -uii;
      cr += (ur * cvr - ui * cvi) + (wr * cur_ - wi * cui_);
      ci += (ur * cvi + ui * cvr) + (wr * cui_ + wi * cur_);
    }
  }
  if (do_x && r <= i) {
    const double xr = axr - cr, xi = axi - ci;
    if (r < i) { H.xr[r] = xr; H.xi[r] = xi; nrm = xr * xr + xi * xi; }
    else H.d[i] = xr;
  }
  nrm = hwave_sum(nrm);
  if (lane == 0 && do_x) pn_out[blockIdx.x] = nrm;
}

// panel end: P1 = [Ur Ui Wr Wi], P2 = [Ui -Ur Wi -Wr], P3 = [Wr Wi Ur Ui]  (rows < nr, k columns each part)
__global__ void h_pack_kernel(HArgs H, int nr, int k, double* __restrict__ P1, double* __restrict__ P2,
                              double* __restrict__ P3) {
  const int j = blockIdx.y;   // 0..k-1
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += gridDim.x * blockDim.x) {
    const double ur = H.Ur[(size_t)r + (size_t)j * H.ldp], ui = H.Ui[(size_t)r + (size_t)j * H.ldp];
    const double wr = H.Wr[(size_t)r + (size_t)j * H.ldp], wi = H.Wi[(size_t)r + (size_t)j * H.ldp];
    const size_t c0 = (size_t)j * H.ldp + r, st = (size_t)k * H.ldp;
    P1[c0] = ur; P1[c0 + st] = ui; P1[c0 + 2 * st] = wr; P1[c0 + 3 * st] = wi;
    P2[c0] = ui; P2[c0 + st] = -ur; P2[c0 + 2 * st] = wi; P2[c0 + 3 * st] = -wr;
    P3[c0] = wr; P3[c0 + st] = wi; P3[c0 + 2 * st] = ur; P3[c0 + 3 * st] = ui;
  }
}

// back-transformation helpers -----------------------------------------------------------------------------------
// zero rows >= j of reflector column j (what is left there is the old lower triangle); column 0 holds no reflector
__global__ void h_zero_below_kernel(double* __restrict__ Vr, double* __restrict__ Vi, int ld, int n, int rows_pad) {
  const int j = blockIdx.y;
  for (int r = j + blockIdx.x * blockDim.x + threadIdx.x; r < rows_pad; r += gridDim.x * blockDim.x) {
    Vr[(size_t)r + (size_t)j * ld] = 0.0; Vi[(size_t)r + (size_t)j * ld] = 0.0;
  }
}

// T = S^-H for one block: S^H lower triangular, S^H(a,b) = G(a,b) = v_a^H v_b (a > b), S^H(a,a) = conj(beta_a)
// (T^-1 of the compact WY form has the strictly upper part of V^H V and 1/tau = beta on its diagonal; the reference
// keeps the same triangle, src/hrbakwy4_body.F:330-470).  One thread per column of T (forward substitution with the unit
// vector e_c), the packed triangle of S^H in LDS; T (nb x nb, ld = HMB, planes) is then applied by real GEMMs.
__global__ __launch_bounds__(HMB) void h_tinv_kernel(const double* __restrict__ Gall, const double* __restrict__ beta,
                                                      int jfirst, int bw, int n, double* __restrict__ Tall) {
  extern __shared__ double sm[];   // packed lower triangle incl. diagonal: [2][HMB*(HMB+1)/2]
  double* sr = sm; double* si = sm + HMB * (HMB + 1) / 2;
  // one workgroup per block of reflectors: all T factors of the back-transformation in ONE launch
  const int j0 = jfirst + (int)blockIdx.x * bw;
  const int nb = (j0 + bw <= n) ? bw : n - j0;
  const double* Gr = Gall + (size_t)blockIdx.x * 4 * HMB * HMB;
  double* Tr = Tall + (size_t)blockIdx.x * 2 * HMB * HMB;
  double* Ti = Tr + (size_t)HMB * HMB;
  // Gr = [Vr^T Vr, Vr^T Vi; Vi^T Vr, Vi^T Vi] (ld = 2 HMB, second half at offset bw): V^H V = (Vr^T Vr + Vi^T Vi) + i (Vr^T Vi - Vi^T Vr)
  constexpr int LG = 2 * HMB;
  for (int t = threadIdx.x; t < nb * nb; t += HMB) {
    const int a = t % nb, b = t / nb;
    if (a > b) {
      sr[a * (a + 1) / 2 + b] = Gr[a + b * LG] + Gr[(bw + a) + (bw + b) * LG];
      si[a * (a + 1) / 2 + b] = Gr[a + (bw + b) * LG] - Gr[(bw + a) + b * LG];
    }
    else if (a == b) { sr[a * (a + 1) / 2 + a] = beta[2 * (j0 + a)]; si[a * (a + 1) / 2 + a] = -beta[2 * (j0 + a) + 1]; }
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= nb) return;
  double* tr = Tr + (size_t)c * HMB; double* ti = Ti + (size_t)c * HMB;
  for (int a = 0; a < c; ++a) { tr[a] = 0.0; ti[a] = 0.0; }
  for (int a = c; a < nb; ++a) {
    double xr = (a == c) ? 1.0 : 0.0, xi = 0.0;
    const int base = a * (a + 1) / 2;
    for (int b = c; b < a; ++b) {
      const double gr = sr[base + b], gi = si[base + b];
      xr -= gr * tr[b] - gi * ti[b];
      xi -= gr * ti[b] + gi * tr[b];
    }
    const double dr = sr[base + a], di = si[base + a];   // conj(beta_a)
    const double d2 = dr * dr + di * di;
    tr[a] = (xr * dr + xi * di) / d2;                    // x / d = x conj(d) / |d|^2
    ti[a] = (xi * dr - xr * di) / d2;
  }
}

// back-transformation block, stacked operands: Vs = [Vr | Vi] (rows x 2nb) so that every pass over a Z plane feeds an
// M = 2nb (or K = 2nb) GEMM instead of two with nb
__global__ void h_stack_v_kernel(const double* __restrict__ Vr, const double* __restrict__ Vi, int ld, int rows, int nb,
                                 double* __restrict__ Vs, int lds) {
  const int j = blockIdx.y;   // 0 .. 2nb-1
  const double* src = (j < nb) ? Vr + (size_t)j * ld : Vi + (size_t)(j - nb) * ld;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x)
    Vs[(size_t)r + (size_t)j * lds] = src[r];
}
// Yr = A(0:nb) + B(nb:2nb), Yi = B(0:nb) - A(nb:2nb) with A = Vs^T Zr, B = Vs^T Zi (2nb x nvec each)
__global__ void h_ycombine_kernel(const double* __restrict__ A, const double* __restrict__ B, int nb, int nvec,
                                  double* __restrict__ Yr, double* __restrict__ Yi) {
  const int c = blockIdx.y;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < nb; a += gridDim.x * blockDim.x) {
    const size_t o = (size_t)c * (2 * HMB);
    Yr[(size_t)c * HMB + a] = A[o + a] + B[o + nb + a];
    Yi[(size_t)c * HMB + a] = B[o + a] - A[o + nb + a];
  }
}
// XA = [Xr; -Xi], XB = [Xi; Xr] (2nb x nvec, ld = 2 HMB): Zr -= Vs XA, Zi -= Vs XB
__global__ void h_xstack_kernel(const double* __restrict__ Xr, const double* __restrict__ Xi, int nb, int nvec,
                                double* __restrict__ XA, double* __restrict__ XB) {
  const int c = blockIdx.y;
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < nb; a += gridDim.x * blockDim.x) {
    const double xr = Xr[(size_t)c * HMB + a], xi = Xi[(size_t)c * HMB + a];
    const size_t o = (size_t)c * (2 * HMB);
    XA[o + a] = xr; XA[o + nb + a] = -xi;
    XB[o + a] = xi; XB[o + nb + a] = xr;
  }
}

__global__ void h_join_kernel(const double* __restrict__ Zr, const double* __restrict__ Zi, int ldzp, int n, int nvec,
                              double* __restrict__ z, int ldz) {
  const int c = blockIdx.y;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    z[2 * ((size_t)r + (size_t)c * ldz)] = Zr[(size_t)r + (size_t)c * ldzp];
    z[2 * ((size_t)r + (size_t)c * ldz) + 1] = Zi[(size_t)r + (size_t)c * ldzp];
  }
}

__global__ void h_identity_kernel(double* __restrict__ z, int ldz, int nvec) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nvec) z[(size_t)j * ldz + j] = 1.0;
}
__global__ void h_scale_vec_kernel(double* __restrict__ w, int n, double s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] *= s;
}
__global__ void h_fill_vec_kernel(double* __restrict__ w, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = v;
}

double hnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// a, z: device, interleaved complex(8), leading dimensions in complex elements
// full (global) matrix on this rank's GPU.  With more than one rank every rank runs this on the same replicated input:
// the reduction and the real tridiagonal D&C are redundant, the back-transformation (19 % of a solve at N = 8192) is
// split by eigenvector columns and allgathered.  Correct; the reduction is what a sharded form still has to distribute.
// bt_P, bt_p: several ranks that hold the same replicated problem share the back-transformation by eigenvector columns
// (rank bt_p of bt_P takes the columns [bt_p * zc, (bt_p + 1) * zc), zc = ceil(nvec / bt_P)) and allgather the result
static int herm_solve_full(Context& ctx, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb,
                           char mode, int bt_P = 1, int bt_p = 0) {
  if (!ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (n <= 0) {
    fprintf(stderr, "[eigx] warning: non-positive dimension is invalid\n");   // src/eigen_h.F:91-94
    return EIGX_ERR_BAD_ARG;
  }
  if (!a || !w || lda < n) return EIGX_ERR_BAD_ARG;
  if (mode >= 'a' && mode <= 'z') mode = (char)(mode - 'a' + 'A');
  if (nvec == 0) mode = 'N';                      // src/eigen_h.F:104-106
  if (nvec < 0) nvec = -nvec;
  if (nvec > n) nvec = n;
  if (mode != 'N' && mode != 'A' && mode != 'X' && mode != 'S') mode = 'A';   // 'S': identity + bisection + back-transformation (src/eigen_h.F:207-210)
  const bool want_vec = mode != 'N';
  if (want_vec && (!z || ldz < n)) return EIGX_ERR_BAD_ARG;
  int m = mf <= 0 ? 48 : mf;
  if (m > HM) m = HM;
  if (m > n) m = n;
  EIGX_HIP_CHECK(hipSetDevice(ctx.device));
  EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));   // the caller's default-stream work on the arguments (see solve_dev)
  hipStream_t st = ctx.stream;
  ctx.errinfo = 0;
  for (int q = 0; q < 16; ++q) ctx.timers[q] = 0.0;
  const double t0 = hnow();

  // ---- eigen_scaling_h -------------------------------------------------------------------------------------------
  double sigma = 1.0;
  {
    const int nbk = 256;
    double* part = ctx.pool.get_t<double>("h.absmax", (size_t)2 * nbk);
    hipLaunchKernelGGL(h_absmax_kernel, dim3(nbk), dim3(HT), 0, st, a, lda, n, part);
    std::vector<double> hp(2 * nbk);
    EIGX_HIP_CHECK(hipMemcpyAsync(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
    double anrm = 0.0, bad = 0.0;
    for (int q = 0; q < nbk; ++q) { anrm = std::max(anrm, hp[2 * q]); bad = std::max(bad, hp[2 * q + 1]); }
    if (bad != 0.0) {   // NaN / Inf in the input: w(:) = NaN (src/eigen_h.F:147-150)
      hipLaunchKernelGGL(h_fill_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, n,
                         std::numeric_limits<double>::quiet_NaN());
      EIGX_HIP_CHECK(hipStreamSynchronize(st));
      ctx.errinfo = -1;
      return EIGX_ERR_NONFINITE;
    }
    if (anrm > 0.0 && (anrm < 1e-90 || anrm > 1e90)) { int ex = 0; (void)frexp(anrm, &ex); sigma = ldexp(1.0, -ex); }
    if (sigma != 1.0) hipLaunchKernelGGL(h_scale_kernel, dim3(8, n), dim3(256), 0, st, a, lda, n, sigma);
  }

  // ---- workspace -----------------------------------------------------------------------------------------------------
  const int ld = pad_ld(n + 2);
  const int ldp = ld;
  HArgs H;
  H.n = n; H.ld = ld; H.ldp = ldp; H.P = 1; H.p = 0;
  H.Ar = ctx.pool.get_t<double>("h.Ar", (size_t)ld * (n + HMB));   // HMB columns of slack: the batched Gram products of phase A
  // The two planes are streamed together at equal offsets; with plane sizes that are multiples of 16 KiB every pair of
  // requests met on the same memory channel.  Half a period (+128 B) between them: reduction 488 -> 459 ms at N = 8192
  // (tools/ab_herm_skew.sh: 0 -> 488, 272 -> 486, 784 -> 471, 1040 -> 459, 1552 -> 498, 4112 -> 492 ms, reproducible).
  static const int plane_skew = [] { const char* e = getenv("EIGX_H_SKEW"); return e ? atoi(e) : 1040; }();   // doubles
  H.Ai = ctx.pool.get_t<double>("h.Ai", (size_t)ld * (n + HMB) + plane_skew) + plane_skew;
  H.Ur = ctx.pool.get_t<double>("h.UW", (size_t)4 * ldp * m);   // the four panel planes in one buffer (one fill per panel)
  H.Ui = H.Ur + (size_t)ldp * m;
  H.Wr = H.Ui + (size_t)ldp * m;
  H.Wi = H.Wr + (size_t)ldp * m;
  double* P1 = ctx.pool.get_t<double>("h.P1", (size_t)ldp * 4 * m);
  double* P2 = ctx.pool.get_t<double>("h.P2", (size_t)ldp * 4 * m);
  double* P3 = ctx.pool.get_t<double>("h.P3", (size_t)ldp * 4 * m);
  H.xr = ctx.pool.get_t<double>("h.xr", (size_t)ldp); H.xi = ctx.pool.get_t<double>("h.xi", (size_t)ldp);
  H.beta = ctx.pool.get_t<double>("h.beta", (size_t)2 * n + 2);
  const int lde = (n + 3) / 4 * 4;
  H.d = ctx.pool.get_t<double>("h.d", (size_t)n);
  H.e = ctx.pool.get_t<double>("h.e", (size_t)lde);
  const int nwg = ceil_div(n, 64) + 1;
  const int nt_max = ceil_div(n, HTL) + 1;
  const int npdc_max = ceil_div(n, PDR) + 1;
  H.pn = ctx.pool.get_t<double>("h.pn", (size_t)2 * nwg);                         // two parities
  H.ps = ctx.pool.get_t<double>("h.ps", (size_t)nt_max * (nt_max + 1));            // 2 per mat-vec tile
  H.pd = ctx.pool.get_t<double>("h.pd", (size_t)npdc_max * HM * 4);
  H.yrr = ctx.pool.get_t<double>("h.yrr", (size_t)nt_max * ldp);
  H.yri = ctx.pool.get_t<double>("h.yri", (size_t)nt_max * ldp);
  H.ycr = ctx.pool.get_t<double>("h.ycr", (size_t)nt_max * ldp);
  H.yci = ctx.pool.get_t<double>("h.yci", (size_t)nt_max * ldp);

  hipLaunchKernelGGL(h_split_kernel, dim3(8, n), dim3(256), 0, st, a, lda, n, H.Ar, H.Ai, ld);
  hipLaunchKernelGGL(h_fill_kernel, dim3(64), dim3(256), 0, st, H.e, (size_t)lde, 0.0);
  hipLaunchKernelGGL(h_fill_kernel, dim3(64), dim3(256), 0, st, H.beta, (size_t)2 * n + 2, 0.0);
  auto zero_panel = [&]() { hipLaunchKernelGGL(h_fill_kernel, dim3(512), dim3(256), 0, st, H.Ur, (size_t)4 * ldp * m, 0.0); };
  zero_panel();

  // ---- eigen_hrd: Hermitian -> real tridiagonal ---------------------------------------------------------------------
  const double t1 = hnow();
  // two launches per column: K1 (finish the previous column, form this one) and K3 (mat-vec + panel dots); the norm
  // partials alternate between two buffers (K1 reads the previous column's while it writes this column's)
  static const int hemv8_nt = [] { const char* e = getenv("EIGX_H_HEMV8_NT"); return e ? atoi(e) : 22; }();   // lab switch
  int k = 0, par = 0;
  int Lp = 0, ntp = 0, npdcp = 0, npartsp = 0;     // the column that is waiting to be finished (Lp = 0: none)
  double* pnb[2] = {H.pn, H.pn + nwg};
  auto step = [&](int i, int do_x) {
    const int rows = std::max(do_x ? i + 1 : 0, Lp);
    H.pn = pnb[par ^ 1];
    hipLaunchKernelGGL(h_step_kernel, dim3(ceil_div(rows, 64)), dim3(HS), 0, st, H, i, do_x, Lp, Lp ? k - 1 : 0, ntp, npdcp,
                       npartsp, pnb[par], ntp * (ntp + 1) / 2, (const double*)nullptr, (const double*)nullptr);
    return ceil_div(rows, 64);
  };
  for (int i = n - 1; i >= 1; --i) {
    const int L = i;
    const int nparts = step(i, 1);
    const int npdc = ceil_div(L, PDR);
    const int nt = ceil_div(L, HTL);
    H.pn = pnb[par];
    if (nt <= hemv8_nt)   // few tiles (253 for nt = 22): 8 waves per tile, one workgroup per CU
      hipLaunchKernelGGL(h_hemv_kernel<8>, dim3(k * npdc + nt * (nt + 1) / 2), dim3(512), 0, st, H, L, k, nt, npdc, 2, nparts,
                         nt * (nt + 1) / 2);
    else
      hipLaunchKernelGGL(h_hemv_kernel<4>, dim3(k * npdc + nt * (nt + 1) / 2), dim3(HTH), 0, st, H, L, k, nt, npdc, 4, nparts,
                         nt * (nt + 1) / 2);
    Lp = L; ntp = nt; npdcp = npdc; npartsp = nparts;
    par ^= 1;
    ++k;
    if (k == m || i == 1) {
      // trailing update of the remaining i x i block, two real GEMMs with K = 4k; the pending column is finished
      // (u, v into the panel) first
      (void)step(i - 1, 0);
      Lp = 0;
      const int nr = i;
      hipLaunchKernelGGL(h_pack_kernel, dim3(ceil_div(nr, 256), k), dim3(256), 0, st, H, nr, k, P1, P2, P3);
      // only the tiles that meet the upper triangle: after the split nothing reads the strict lower triangle any more
      // (K1 reads A(0:i, i), the mat-vec the tiles ty <= tx with the lower half of a diagonal tile masked, the
      // back-transformation zeroes below the reflectors first)
      dgemm_dev(st, 'N', 'T', nr, nr, 4 * k, -1.0, P1, ldp, P3, ldp, 1.0, H.Ar, ld, 1);
      dgemm_dev(st, 'N', 'T', nr, nr, 4 * k, -1.0, P2, ldp, P3, ldp, 1.0, H.Ai, ld, 1);
      zero_panel();
      k = 0;
    }
  }
  (void)step(0, 1);   // d_0 = Re A(0,0)
  // ---- back-transformation, phase A: the T factors of all blocks (they depend on the reflectors only).  V^H V of a
  // block = (Vr^T Vr + Vi^T Vi) + i (Vr^T Vi - Vi^T Vr) from two batched launches over all blocks and both planes,
  // straight from the reflector columns of A with K = n for every block (rows below a reflector are zero after
  // h_zero_below; 64 separate products of 16 workgroups each took 14 ms at N = 8192), then every T = S^-H in one launch
  // (a per-block single-workgroup kernel on the critical path cost 37 of 149 ms at N=8192).
  int bw = mb <= 0 ? HMB : mb;
  if (bw > HMB) bw = HMB;
  const int lds = ld;
  const int nblk = n > 1 ? ceil_div(n - 1, bw) : 1;
  double* Tall = nullptr;
  if (want_vec && n > 1) {
    double* Gall = ctx.pool.get_t<double>("h.Gall", (size_t)nblk * 4 * HMB * HMB);
    Tall = ctx.pool.get_t<double>("h.Tall", (size_t)nblk * 2 * HMB * HMB);
    const size_t shm = (size_t)2 * (HMB * (HMB + 1) / 2) * sizeof(double);
    static bool attr = false;
    if (!attr) {
      EIGX_HIP_CHECK(hipFuncSetAttribute((const void*)h_tinv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      attr = true;
    }
    hipLaunchKernelGGL(h_zero_below_kernel, dim3(8, n), dim3(256), 0, st, H.Ar, H.Ai, ld, n, n);
    // block b = columns 1 + b bw ... of A; G_b (ld 2 HMB) = [Vr^T Vr, Vr^T Vi; Vi^T Vr, Vi^T Vi] with the second half at
    // offset bw.  The last block may be narrower: its product reads up to bw - 1 columns past the matrix (allocated,
    // any content) into entries of G_b that nobody reads.
    const long sblk = (long)bw * ld, sg = (long)4 * HMB * HMB;
    for (int half = 0; half < 2; ++half)
      dgemm_dev(st, 'T', 'N', bw, bw, n, 1.0, (half ? H.Ai : H.Ar) + (size_t)ld, ld, H.Ar + (size_t)ld, ld, 0.0,
                Gall + (half ? bw : 0), 2 * HMB, 0, nullptr, nullptr, nullptr, nblk, sblk, sblk, sg, 2, 0, (long)(H.Ai - H.Ar),
                (long)bw * 2 * HMB);
    hipLaunchKernelGGL(h_tinv_kernel, dim3(nblk), dim3(HMB), shm, st, Gall, H.beta, 1, bw, n, Tall);
  }
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  const double t2 = hnow();

  // ---- real tridiagonal eigenproblem (dc2 / bisect) --------------------------------------------------------------------
  double* Zr = nullptr;
  double* Zi = nullptr;
  const int ldzp = pad_ld(n + 2);
  if (!want_vec) {
    band_bisect_dev(ctx, n, H.d, H.e, lde, 1, w);
  } else {
    // both planes in one buffer (V^H Z is one batched product over them), room for bt_P column blocks of
    // ceil(nvec / bt_P) each, the imaginary plane half a 16 KiB period off the real one's grid (as for A above)
    const size_t zplane = (size_t)ldzp * (n + bt_P) + 1040;
    Zr = ctx.pool.get_t<double>("h.Zri", 2 * zplane);
    Zi = Zr + zplane;
    if (mode == 'S') {
      // Z = I (the first nvec columns), eigenvalues by bisection: the back-transformation then delivers the unitary
      // matrix of the reduction itself, Z^H A Z = T (src/eigen_h.F:207-210, eigen_identity src/eigen_identity.F)
      hipLaunchKernelGGL(h_fill_kernel, dim3(1024), dim3(256), 0, st, Zr, (size_t)ldzp * nvec, 0.0);
      hipLaunchKernelGGL(h_identity_kernel, dim3(ceil_div(nvec, 256)), dim3(256), 0, st, Zr, ldzp, nvec);
      band_bisect_dev(ctx, n, H.d, H.e, lde, 1, w);
    } else {
      // multi-rank callers reach this point with the gathered (replicated) problem: the real tridiagonal D&C runs
      // replicated too (its distributed form delivers column blocks, which only the real solvers consume)
      GridSwap one_rank(ctx);
      band_dc_dev(ctx, n, nvec, H.d, H.e, lde, 1, w, Zr, ldzp);
    }
    if (mode == 'X') band_bisect_dev(ctx, n, H.d, H.e, lde, 1, w);
    hipLaunchKernelGGL(h_fill_kernel, dim3(1024), dim3(256), 0, st, Zi, (size_t)ldzp * nvec, 0.0);
  }
  const double t3 = hnow();

  // ---- eigen_hrbakwyx: z = H_{n-1}^H ... H_1^H y in blocks of HMB reflectors -------------------------------------------
  if (want_vec && n > 1) {
    double* YA = ctx.pool.get_t<double>("h.YAB", (size_t)2 * 2 * HMB * nvec);
    double* YB = YA + (size_t)2 * HMB * nvec;
    double* Yr = ctx.pool.get_t<double>("h.Yr", (size_t)HMB * nvec);
    double* Yi = ctx.pool.get_t<double>("h.Yi", (size_t)HMB * nvec);
    double* Xr = ctx.pool.get_t<double>("h.Xr", (size_t)HMB * nvec);
    double* Xi = ctx.pool.get_t<double>("h.Xi", (size_t)HMB * nvec);
    double* Vs = ctx.pool.get_t<double>("h.Vs", (size_t)lds * 2 * HMB);
    // phase B: apply the blocks in ascending order -- to this rank's eigenvector columns
    const int zc = ceil_div(nvec, bt_P);
    const int c0 = (bt_p * zc < nvec) ? bt_p * zc : nvec;
    const int cn = (c0 + zc <= nvec) ? zc : nvec - c0;
    double* const Zr_all = Zr;
    double* const Zi_all = Zi;
    const int nvec_all = nvec;
    Zr += (size_t)c0 * ldzp; Zi += (size_t)c0 * ldzp;
    nvec = cn;
    for (int b = 0; b < nblk && nvec > 0; ++b) {
      const int j0 = 1 + b * bw;
      const int nb = (j0 + bw <= n) ? bw : n - j0;
      const int rows = j0 + nb - 1;
      const double* Tr = Tall + (size_t)b * 2 * HMB * HMB;
      const double* Ti = Tr + (size_t)HMB * HMB;
      hipLaunchKernelGGL(h_stack_v_kernel, dim3(8, 2 * nb), dim3(256), 0, st, H.Ar + (size_t)j0 * ld, H.Ai + (size_t)j0 * ld,
                         ld, rows, nb, Vs, lds);
      // Y = V^H Z from YA = Vs^T Zr, YB = Vs^T Zi: one batched launch (256 tiles: the LDS-ring kernel; two launches of
      // 128 tiles each went to the 64 x 64 kernel)
      dgemm_dev(st, 'T', 'N', 2 * nb, nvec, rows, 1.0, Vs, lds, Zr, ldzp, 0.0, YA, 2 * HMB, 0, nullptr, nullptr, nullptr, 2, 0,
                (long)(Zi - Zr), (long)(YB - YA));
      hipLaunchKernelGGL(h_ycombine_kernel, dim3(1, nvec), dim3(128), 0, st, YA, YB, nb, nvec, Yr, Yi);
      // X = T Y : Xr = Tr Yr - Ti Yi ; Xi = Tr Yi + Ti Yr
      dgemm_dev(st, 'N', 'N', nb, nvec, nb, 1.0, Tr, HMB, Yr, HMB, 0.0, Xr, HMB);
      dgemm_dev(st, 'N', 'N', nb, nvec, nb, -1.0, Ti, HMB, Yi, HMB, 1.0, Xr, HMB);
      dgemm_dev(st, 'N', 'N', nb, nvec, nb, 1.0, Tr, HMB, Yi, HMB, 0.0, Xi, HMB);
      dgemm_dev(st, 'N', 'N', nb, nvec, nb, 1.0, Ti, HMB, Yr, HMB, 1.0, Xi, HMB);
      // Z -= V X : Zr -= Vs [Xr; -Xi] ; Zi -= Vs [Xi; Xr]   (K = 2nb)
      hipLaunchKernelGGL(h_xstack_kernel, dim3(1, nvec), dim3(128), 0, st, Xr, Xi, nb, nvec, YA, YB);
      dgemm_dev(st, 'N', 'N', rows, nvec, 2 * nb, -1.0, Vs, lds, YA, 2 * HMB, 1.0, Zr, ldzp);
      dgemm_dev(st, 'N', 'N', rows, nvec, 2 * nb, -1.0, Vs, lds, YB, 2 * HMB, 1.0, Zi, ldzp);
    }
    nvec = nvec_all;
    Zr = Zr_all; Zi = Zi_all;
    if (bt_P > 1) {   // every rank gets every column block (the caller cuts its cyclic block out of the full matrix)
      double* Gr = ctx.pool.get_t<double>("h.ZrG", (size_t)ldzp * zc * bt_P);
      double* Gi = ctx.pool.get_t<double>("h.ZiG", (size_t)ldzp * zc * bt_P);
      comm_allgather(ctx, COMM_WORLD, Zr + (size_t)bt_p * zc * ldzp, Gr, (size_t)zc * ldzp, st);
      comm_allgather(ctx, COMM_WORLD, Zi + (size_t)bt_p * zc * ldzp, Gi, (size_t)zc * ldzp, st);
      Zr = Gr; Zi = Gi;
    }
  }
  if (want_vec) hipLaunchKernelGGL(h_join_kernel, dim3(8, nvec), dim3(256), 0, st, Zr, Zi, ldzp, n, nvec, z, ldz);
  if (sigma != 1.0 && sigma != 0.0)
    hipLaunchKernelGGL(h_scale_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, n, 1.0 / sigma);
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  EIGX_HIP_CHECK(hipGetLastError());
  const double t4 = hnow();

  // ---- statistics (src/eigen_h.F:284-288): a(1,1) = flops, a(2,1) = seconds (real parts) ------------------------------
  const double f_red = 4.0 / 3.0 * (double)n * n * n;
  const double f_dc = ctx.timers[11];
  const double f_bt = want_vec ? 2.0 * (double)nvec * n * n : 0.0;
  const double ret = f_red + f_dc + f_bt;
  ctx.timers[0] = t4 - t0; ctx.timers[1] = t2 - t1; ctx.timers[2] = t3 - t2; ctx.timers[3] = t4 - t3; ctx.timers[12] = ret;
  const double stats[4] = {ret, 0.0, t4 - t0, 0.0};
  EIGX_HIP_CHECK(hipMemcpyAsync(a, stats, (size_t)(n >= 2 ? 4 : 2) * 8, hipMemcpyHostToDevice, st));
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  return EIGX_OK;
}

// =====================================================================================================================
// Several ranks: SHARDED reduction (round 4).  The reference distributes eigen_hrd over the process grid (src/eigen_hrd.F:1-448,
// src/eigen_hrd_t2.F: a rank forms its part of A u and the parts are summed by allreduces over the grid's rows and
// columns); here
//   * a rank holds the tile columns tx (128 columns) of the split planes with tx mod P == rank, stored compactly: n^2 / P
//     per plane (the upper triangle's tile column tx holds tx + 1 tiles, so dealing tile columns round-robin balances);
//   * per column: the tiled Hermitian mat-vec over the rank's own tiles, a local sum of its partial results per row
//     (hs_yreduce_kernel), ONE deterministic allreduce of [y (2 L) | u^H q (2) | the raw next column of A (2 L, from its
//     owner)] over all ranks, then the panel work (h_step_kernel) replicated on every rank from bit-identical inputs --
//     d, e, beta, x and the panels U, W are therefore replicated bit for bit (what the distributed D&C needs);
//   * the trailing update on the rank's tile columns only (one batched GEMM per plane);
//   * T factors of the blocks a rank owns; the real tridiagonal D&C distributed (dc.hip) delivering eigenvector column
//     blocks; the back-transformation on the rank's column block with the reflector blocks streaming past in groups of P
//     (each rank contributes the one it owns to an allgather); exit through the real solvers' all-to-all, plane by plane.
// Nothing of size n^2 is gathered anywhere.
namespace {
__global__ __launch_bounds__(HT) void hs_absmax_kernel(const double* __restrict__ a, int lda, int nr, int nc, int Px, int px, int Py,
                                                       int py, double* __restrict__ out) {
  __shared__ double red[2 * (HT / 64)];
  double mx = 0.0, bad = 0.0;
  for (int lj = blockIdx.x; lj < nc; lj += gridDim.x) {
    const int gj = lj * Py + py;
    for (int li = threadIdx.x; li < nr; li += HT) {
      if (li * Px + px > gj) break;                       // upper triangle only (rows ascend with li)
      const double re = a[2 * ((size_t)lj * lda + li)], im = a[2 * ((size_t)lj * lda + li) + 1];
      if (!(fabs(re) <= DBL_MAX) || !(fabs(im) <= DBL_MAX)) bad = 1.0;
      else mx = fmax(mx, fmax(fabs(re), fabs(im)));
    }
  }
  for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_xor(mx, o, 64)); bad = fmax(bad, __shfl_xor(bad, o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = mx; red[2 * (threadIdx.x >> 6) + 1] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < HT / 64; ++w) { mx = fmax(mx, red[2 * w]); bad = fmax(bad, red[2 * w + 1]); }
    out[2 * blockIdx.x] = mx; out[2 * blockIdx.x + 1] = bad;
  }
}
// entry all-to-all, sender: piece for rank d = [plane][k][lr], k-th of my local columns whose tile column d owns
__global__ void hs_pack_kernel(const double* __restrict__ a, int lda, int nr, const int* __restrict__ cols, int ncmax, int nrmax,
                               double sigma, double* __restrict__ send) {
  const int k = blockIdx.y, d = blockIdx.z;
  const int lc = cols[d * ncmax + k];
  if (lc < 0) return;
  double* dst = send + ((size_t)d * 2 * ncmax + k) * nrmax;
  for (int lr = blockIdx.x * blockDim.x + threadIdx.x; lr < nr; lr += gridDim.x * blockDim.x) {
    dst[lr] = sigma * a[2 * ((size_t)lc * lda + lr)];
    dst[(size_t)ncmax * nrmax + lr] = sigma * a[2 * ((size_t)lc * lda + lr) + 1];
  }
}
// receiver: source s = (sx, .) sent its k-th such column = global column gcols[s * ncmax + k]; its rows are sx, sx + Px, ...
__global__ void hs_unpack_kernel(const double* __restrict__ recv, const int* __restrict__ gcols, int ncmax, int nrmax, int n,
                                 int Px, int Py, int row_major, HArgs H) {
  const int k = blockIdx.y, sidx = blockIdx.z;
  const int gj = gcols[sidx * ncmax + k];
  if (gj < 0) return;
  const int sx = row_major ? sidx / Py : sidx % Px;
  const double* src = recv + ((size_t)sidx * 2 * ncmax + k) * nrmax;
  const size_t co = hcol(H, gj);
  for (int lr = blockIdx.x * blockDim.x + threadIdx.x; lr * Px + sx < n; lr += gridDim.x * blockDim.x) {
    const int gi = lr * Px + sx;
    H.Ar[co + gi] = __hip_atomic_load(src + lr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    H.Ai[co + gi] = __hip_atomic_load(src + (size_t)ncmax * nrmax + lr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// ybuf = [y re (ldp) | y im (ldp) | raw column ci re (ldp) | im (ldp) | s re, s im]: this rank's share of each
__global__ __launch_bounds__(HT) void hs_yreduce_kernel(HArgs H, int L, int nt, int ntl, int ci, double* __restrict__ ybuf) {
  __shared__ double red[2 * (HT / 64)];
  const int ldp = H.ldp;
  const int r = blockIdx.x * HT + threadIdx.x;
  if (r < ldp) {
    double yr = 0.0, yi = 0.0;
    if (r < L) {
      const int R = r >> 7;
      if (R % H.P == H.p)
        for (int u = 0; u <= R; ++u) { yr += H.ycr[(size_t)u * ldp + r]; yi += H.yci[(size_t)u * ldp + r]; }      // tiles (u, R): column sums
      for (int tx = R + ((H.p - R % H.P) + H.P) % H.P; tx < nt; tx += H.P) {                                       // tiles (R, tx): row sums
        yr += H.yrr[(size_t)tx * ldp + r]; yi += H.yri[(size_t)tx * ldp + r];
      }
    }
    ybuf[r] = yr; ybuf[(size_t)ldp + r] = yi;
    double cr = 0.0, cim = 0.0;
    if (ci >= 0 && r <= ci && (ci >> 7) % H.P == H.p) { cr = H.Ar[hcol(H, ci) + r]; cim = H.Ai[hcol(H, ci) + r]; }
    ybuf[(size_t)2 * ldp + r] = cr; ybuf[(size_t)3 * ldp + r] = cim;
  }
  if (blockIdx.x == 0) {
    double sr = 0.0, si = 0.0;
    for (int t = threadIdx.x; t < ntl; t += HT) { sr += H.ps[2 * t]; si += H.ps[2 * t + 1]; }
    for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o, 64); si += __shfl_xor(si, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = sr; red[2 * (threadIdx.x >> 6) + 1] = si; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < HT / 64; ++w) { sr += red[2 * w]; si += red[2 * w + 1]; }
      ybuf[(size_t)4 * ldp] = sr; ybuf[(size_t)4 * ldp + 1] = si;
    }
  }
}
// rows >= j of my reflector columns j (what is left there is the old lower triangle)
__global__ void hs_zero_below_kernel(HArgs H, int n, int nlc) {
  const int lc = blockIdx.y;
  if (lc >= nlc) return;
  const int j = ((lc >> 7) * H.P + H.p) * 128 + (lc & 127);
  if (j >= n) return;
  for (int r = j + blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
    H.Ar[(size_t)lc * H.ld + r] = 0.0; H.Ai[(size_t)lc * H.ld + r] = 0.0;
  }
}
__global__ void hs_identity_block_kernel(double* __restrict__ z, int ldz, int c0, int cn) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < cn) z[(size_t)j * ldz + c0 + j] = 1.0;
}
__global__ void hs_join_cyclic_kernel(const double* __restrict__ zr, const double* __restrict__ zi, int ldt, int nr, int nzc,
                                      double* __restrict__ z, int ldz) {
  const int lj = blockIdx.y;
  if (lj >= nzc) return;
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < nr; li += gridDim.x * blockDim.x) {
    z[2 * ((size_t)lj * ldz + li)] = zr[(size_t)lj * ldt + li];
    z[2 * ((size_t)lj * ldz + li) + 1] = zi[(size_t)lj * ldt + li];
  }
}
}  // namespace

// a, z: this rank's 2-D cyclic blocks (device, interleaved complex)
static int herm_solve_sharded(Context& ctx, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb,
                              char mode) {
  const Grid G = ctx.grid;
  const int P = G.nranks, me = G.rank;
  const int nloc_r = local_count(n, G.Px, G.px), nloc_c = local_count(n, G.Py, G.py);
  if (mode >= 'a' && mode <= 'z') mode = (char)(mode - 'a' + 'A');
  if (nvec == 0) mode = 'N';                      // src/eigen_h.F:104-106
  if (nvec < 0) nvec = -nvec;
  if (nvec > n) nvec = n;
  if (mode != 'N' && mode != 'A' && mode != 'X' && mode != 'S') mode = 'A';
  const bool want_vec = mode != 'N';
  int m = mf <= 0 ? 48 : mf;
  if (m > HM) m = HM;
  if (m > n) m = n;
  hipStream_t st = ctx.stream;
  EIGX_HIP_CHECK(hipStreamSynchronize(nullptr));
  ctx.errinfo = 0;
  for (int q = 0; q < 16; ++q) ctx.timers[q] = 0.0;
  (void)comm_seconds(ctx, true);
  const double t0 = hnow();

  // ---- eigen_scaling_h on the local blocks, maxima combined over the ranks ------------------------------------------
  double sigma = 1.0;
  {
    const int nbk = 64;
    double* part = ctx.pool.get_t<double>("hs.absmax", (size_t)2 * nbk + 2);
    hipLaunchKernelGGL(hs_absmax_kernel, dim3(nbk), dim3(HT), 0, st, a, lda, nloc_r, nloc_c, G.Px, G.px, G.Py, G.py, part);
    std::vector<double> hp(2 * nbk);
    EIGX_HIP_CHECK(hipMemcpyAsync(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
    double two[2] = {0.0, 0.0};
    for (int q = 0; q < nbk; ++q) { two[0] = std::max(two[0], hp[2 * q]); two[1] = std::max(two[1], hp[2 * q + 1]); }
    EIGX_HIP_CHECK(hipMemcpyAsync(part, two, 16, hipMemcpyHostToDevice, st));
    comm_allreduce_max(ctx, COMM_WORLD, part, 2, st);
    EIGX_HIP_CHECK(hipMemcpyAsync(two, part, 16, hipMemcpyDeviceToHost, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
    if (comm_failed(ctx)) return EIGX_ERR_INTERNAL;
    if (two[1] != 0.0) {   // NaN / Inf in the input (on any rank): w(:) = NaN on every rank (src/eigen_h.F:147-150)
      hipLaunchKernelGGL(h_fill_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, n,
                         std::numeric_limits<double>::quiet_NaN());
      EIGX_HIP_CHECK(hipStreamSynchronize(st));
      ctx.errinfo = -1;
      return EIGX_ERR_NONFINITE;
    }
    const double anrm = two[0];
    if (anrm > 0.0 && (anrm < 1e-90 || anrm > 1e90)) { int ex = 0; (void)frexp(anrm, &ex); sigma = ldexp(1.0, -ex); }
  }

  // ---- workspace: the rank's tile columns of the two planes ------------------------------------------------------------
  const int ld = pad_ld(n + 2);
  const int ldp = ld;
  const int nt_all = ceil_div(n, HTL);
  const int ntc = ceil_div(nt_all, P);                 // tile columns per rank (upper bound)
  const int nlc = ntc * HTL;                           // local columns
  HArgs H;
  H.n = n; H.ld = ld; H.ldp = ldp; H.P = P; H.p = me;
  static const int plane_skew = [] { const char* e = getenv("EIGX_H_SKEW"); return e ? atoi(e) : 1040; }();   // doubles
  H.Ar = ctx.pool.get_t<double>("hs.Ar", (size_t)ld * (nlc + HMB));
  H.Ai = ctx.pool.get_t<double>("hs.Ai", (size_t)ld * (nlc + HMB) + plane_skew) + plane_skew;
  H.Ur = ctx.pool.get_t<double>("h.UW", (size_t)4 * ldp * m);
  H.Ui = H.Ur + (size_t)ldp * m;
  H.Wr = H.Ui + (size_t)ldp * m;
  H.Wi = H.Wr + (size_t)ldp * m;
  double* P1 = ctx.pool.get_t<double>("h.P1", (size_t)ldp * 4 * m);
  double* P2 = ctx.pool.get_t<double>("h.P2", (size_t)ldp * 4 * m);
  double* P3 = ctx.pool.get_t<double>("h.P3", (size_t)ldp * 4 * m);
  H.xr = ctx.pool.get_t<double>("h.xr", (size_t)ldp); H.xi = ctx.pool.get_t<double>("h.xi", (size_t)ldp);
  H.beta = ctx.pool.get_t<double>("h.beta", (size_t)2 * n + 2);
  const int lde = (n + 3) / 4 * 4;
  H.d = ctx.pool.get_t<double>("h.d", (size_t)n);
  H.e = ctx.pool.get_t<double>("h.e", (size_t)lde);
  const int nwg = ceil_div(n, 64) + 1;
  const int nt_max = nt_all + 1;
  const int npdc_max = ceil_div(n, PDR) + 1;
  H.pn = ctx.pool.get_t<double>("h.pn", (size_t)2 * nwg);
  H.ps = ctx.pool.get_t<double>("h.ps", (size_t)nt_max * (nt_max + 1));
  H.pd = ctx.pool.get_t<double>("h.pd", (size_t)npdc_max * HM * 4);
  H.yrr = ctx.pool.get_t<double>("h.yrr", (size_t)nt_max * ldp);
  H.yri = ctx.pool.get_t<double>("h.yri", (size_t)nt_max * ldp);
  H.ycr = ctx.pool.get_t<double>("h.ycr", (size_t)nt_max * ldp);
  H.yci = ctx.pool.get_t<double>("h.yci", (size_t)nt_max * ldp);
  double* ybuf = ctx.pool.get_t<double>("hs.ybuf", (size_t)4 * ldp + 8);
  const size_t ycount = (size_t)4 * ldp + 2;
  hipLaunchKernelGGL(h_fill_kernel, dim3(1024), dim3(256), 0, st, H.Ar, (size_t)ld * (nlc + HMB), 0.0);
  hipLaunchKernelGGL(h_fill_kernel, dim3(1024), dim3(256), 0, st, H.Ai, (size_t)ld * (nlc + HMB), 0.0);

  // ---- entry: 2-D cyclic blocks -> tile columns, one all-to-all --------------------------------------------------------
  {
    auto world_of = [&](int qx, int qy) { return G.row_major ? qx * G.Py + qy : qx + qy * G.Px; };
    // the widest list any (source process column, destination) pair has: the same number on every rank
    int ncmax = 1;
    for (int sy = 0; sy < G.Py; ++sy) {
      std::vector<int> cnt(P, 0);
      for (int j = sy; j < n; j += G.Py) ++cnt[(j / HTL) % P];
      for (int d = 0; d < P; ++d) ncmax = std::max(ncmax, cnt[d]);
    }
    const int nrmax = ceil_div(n, G.Px);
    const size_t piece = (size_t)2 * ncmax * nrmax;
    std::vector<int> cols((size_t)2 * P * ncmax, -1);   // [0, P ncmax): my local columns per destination; then global columns per source
    {
      std::vector<int> fill(P, 0);
      for (int lc = 0; lc < nloc_c; ++lc) { const int d = ((lc * G.Py + G.py) / HTL) % P; cols[(size_t)d * ncmax + fill[d]++] = lc; }
      for (int sx = 0; sx < G.Px; ++sx)
        for (int sy = 0; sy < G.Py; ++sy) {
          const int s_ = world_of(sx, sy);
          int f = 0;
          for (int j = sy; j < n; j += G.Py)
            if ((j / HTL) % P == me) cols[(size_t)(P + s_) * ncmax + f++] = j;
        }
    }
    int* tab = ctx.pool.get_t<int>("hs.tab", cols.size());
    EIGX_HIP_CHECK(hipMemcpyAsync(tab, cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));   // (cols is a stack vector)
    double* sendb = ctx.pool.get_t<double>("mg.xsend", piece * P);
    double* recvb = ctx.pool.get_t<double>("mg.xrecv", piece * P);
    if (nloc_r > 0)
      hipLaunchKernelGGL(hs_pack_kernel, dim3(ceil_div(nloc_r, 256), ncmax, P), dim3(256), 0, st, (const double*)a, lda, nloc_r,
                         (const int*)tab, ncmax, nrmax, sigma, sendb);
    comm_exchange_big(ctx, COMM_WORLD, sendb, piece, recvb, piece, st);
    hipLaunchKernelGGL(hs_unpack_kernel, dim3(ceil_div(nrmax, 256), ncmax, P), dim3(256), 0, st, (const double*)recvb,
                       (const int*)(tab + (size_t)P * ncmax), ncmax, nrmax, n, G.Px, G.Py, G.row_major, H);
  }
  hipLaunchKernelGGL(h_fill_kernel, dim3(64), dim3(256), 0, st, H.e, (size_t)lde, 0.0);
  hipLaunchKernelGGL(h_fill_kernel, dim3(64), dim3(256), 0, st, H.beta, (size_t)2 * n + 2, 0.0);
  auto zero_panel = [&]() { hipLaunchKernelGGL(h_fill_kernel, dim3(512), dim3(256), 0, st, H.Ur, (size_t)4 * ldp * m, 0.0); };
  zero_panel();

  // ---- eigen_hrd, sharded -------------------------------------------------------------------------------------------------
  const double t1 = hnow();
  auto ntl_of = [&](int nt) { long c = 0; for (int tx = me; tx < nt; tx += P) c += tx + 1; return (int)c; };   // my tiles of an nt x nt triangle
  int k = 0, par = 0;
  int Lp = 0, ntp = 0, npdcp = 0, npartsp = 0;
  double* pnb[2] = {H.pn, H.pn + nwg};
  HArgs Hk = H;                               // the panel kernel's view: the summed mat-vec result instead of the tile partials
  Hk.ycr = ybuf; Hk.yci = ybuf + ldp; Hk.ps = ybuf + (size_t)4 * ldp;
  // my share of [y | s | raw column ci] of the pending column, summed over the ranks: every rank gets the same bits
  auto exchange = [&](int ci) {
    hipLaunchKernelGGL(hs_yreduce_kernel, dim3(ceil_div(ldp, HT)), dim3(HT), 0, st, H, Lp, ntp, ntl_of(ntp), ci, ybuf);
    comm_allreduce_sum(ctx, COMM_WORLD, ybuf, ycount, st);
  };
  auto step = [&](int i, int do_x) {
    const int rows = std::max(do_x ? i + 1 : 0, Lp);
    Hk.pn = pnb[par ^ 1];
    hipLaunchKernelGGL(h_step_kernel, dim3(ceil_div(rows, 64)), dim3(HS), 0, st, Hk, i, do_x, Lp, Lp ? k - 1 : 0, 0, npdcp, npartsp,
                       pnb[par], Lp ? 1 : 0, (const double*)(ybuf + (size_t)2 * ldp), (const double*)(ybuf + (size_t)3 * ldp));
    return ceil_div(rows, 64);
  };
  for (int i = n - 1; i >= 1; --i) {
    const int L = i;
    exchange(i);
    const int nparts = step(i, 1);
    const int npdc = ceil_div(L, PDR);
    const int nt = ceil_div(L, HTL);
    const int ntl = ntl_of(nt);
    H.pn = pnb[par];
    const int grid = std::max(1, k * npdc + ntl);
    if (nt <= 22)
      hipLaunchKernelGGL(h_hemv_kernel<8>, dim3(grid), dim3(512), 0, st, H, L, k, nt, npdc, 2, nparts, ntl);
    else
      hipLaunchKernelGGL(h_hemv_kernel<4>, dim3(grid), dim3(HTH), 0, st, H, L, k, nt, npdc, 4, nparts, ntl);
    Lp = L; ntp = nt; npdcp = npdc; npartsp = nparts;
    par ^= 1;
    ++k;
    if (k == m || i == 1) {
      exchange(-1);
      (void)step(i - 1, 0);
      Lp = 0;
      const int nr = i;
      hipLaunchKernelGGL(h_pack_kernel, dim3(ceil_div(nr, 256), k), dim3(256), 0, st, H, nr, k, P1, P2, P3);
      // my tile columns of the active block: the full ones in one batched product per plane, then the ragged last one
      const int tfull = nr / HTL;                                   // tile columns 0 .. tfull-1 lie inside nr completely
      const int gfull = (tfull > me) ? (tfull - 1 - me) / P + 1 : 0;
      if (gfull > 0) {
        dgemm_dev(st, 'N', 'T', nr, HTL, 4 * k, -1.0, P1, ldp, P3 + (size_t)me * HTL, ldp, 1.0, H.Ar, ld, 0, nullptr, nullptr,
                  nullptr, gfull, 0, (long)P * HTL, (long)HTL * ld);
        dgemm_dev(st, 'N', 'T', nr, HTL, 4 * k, -1.0, P2, ldp, P3 + (size_t)me * HTL, ldp, 1.0, H.Ai, ld, 0, nullptr, nullptr,
                  nullptr, gfull, 0, (long)P * HTL, (long)HTL * ld);
      }
      if (nr % HTL != 0 && tfull % P == me) {
        const int c0_ = tfull * HTL;
        dgemm_dev(st, 'N', 'T', nr, nr - c0_, 4 * k, -1.0, P1, ldp, P3 + c0_, ldp, 1.0, H.Ar + hcol(H, c0_), ld);
        dgemm_dev(st, 'N', 'T', nr, nr - c0_, 4 * k, -1.0, P2, ldp, P3 + c0_, ldp, 1.0, H.Ai + hcol(H, c0_), ld);
      }
      zero_panel();
      k = 0;
    }
  }
  exchange(0);
  (void)step(0, 1);   // d_0 = Re A(0,0)

  // ---- T factors of my reflector blocks.  Block b = the reflector columns of tile column b: [max(1, 128 b), 128 (b+1)) ----
  const int nblk = nt_all;
  auto blk_j0 = [&](int b) { return b == 0 ? 1 : b * HTL; };
  auto blk_j1 = [&](int b) { return std::min(n, (b + 1) * HTL); };
  double* Tloc = nullptr;
  if (want_vec && n > 1) {
    double* G1 = ctx.pool.get_t<double>("h.Gall", (size_t)4 * HMB * HMB);
    Tloc = ctx.pool.get_t<double>("h.Tall", (size_t)ntc * 2 * HMB * HMB);
    const size_t shm = (size_t)2 * (HMB * (HMB + 1) / 2) * sizeof(double);
    static bool attr = false;
    if (!attr) {
      EIGX_HIP_CHECK(hipFuncSetAttribute((const void*)h_tinv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      attr = true;
    }
    hipLaunchKernelGGL(hs_zero_below_kernel, dim3(8, nlc), dim3(256), 0, st, H, n, nlc);
    for (int b = me; b < nblk; b += P) {
      const int j0 = blk_j0(b), nb = blk_j1(b) - j0;
      if (nb <= 0) continue;
      const size_t co = hcol(H, j0);
      for (int half = 0; half < 2; ++half)
        dgemm_dev(st, 'T', 'N', nb, nb, n, 1.0, (half ? H.Ai : H.Ar) + co, ld, H.Ar + co, ld, 0.0, G1 + (half ? nb : 0), 2 * HMB, 0,
                  nullptr, nullptr, nullptr, 1, 0, 0, 0, 2, 0, (long)(H.Ai - H.Ar), (long)nb * 2 * HMB);
      hipLaunchKernelGGL(h_tinv_kernel, dim3(1), dim3(HMB), shm, st, (const double*)G1, (const double*)H.beta, j0, nb, j0 + nb,
                         Tloc + (size_t)(b / P) * 2 * HMB * HMB);
    }
  }
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  if (comm_failed(ctx)) return EIGX_ERR_INTERNAL;
  const double t2 = hnow();

  // ---- real tridiagonal eigenproblem: distributed D&C (my eigenvector columns, all rows) / bisection --------------------
  const int zc = ceil_div(nvec > 0 ? nvec : 1, P);
  const int c0 = (me * zc < nvec) ? me * zc : nvec;
  const int cn = want_vec ? ((c0 + zc <= nvec) ? zc : nvec - c0) : 0;
  const int ldzp = pad_ld(n + 2);
  double* Zr = nullptr;
  double* Zi = nullptr;
  if (!want_vec) {
    band_bisect_dev(ctx, n, H.d, H.e, lde, 1, w);
  } else {
    const size_t zplane = (size_t)ldzp * (zc + 1) + 1040;
    Zr = ctx.pool.get_t<double>("hs.Zri", 2 * zplane);
    Zi = Zr + zplane;
    if (mode == 'S') {
      hipLaunchKernelGGL(h_fill_kernel, dim3(1024), dim3(256), 0, st, Zr, (size_t)ldzp * zc, 0.0);
      if (cn > 0) hipLaunchKernelGGL(hs_identity_block_kernel, dim3(ceil_div(cn, 256)), dim3(256), 0, st, Zr, ldzp, c0, cn);
      band_bisect_dev(ctx, n, H.d, H.e, lde, 1, w);
    } else {
      band_dc_dev(ctx, n, nvec, H.d, H.e, lde, 1, w, Zr, ldzp);
    }
    if (mode == 'X') band_bisect_dev(ctx, n, H.d, H.e, lde, 1, w);
    hipLaunchKernelGGL(h_fill_kernel, dim3(1024), dim3(256), 0, st, Zi, (size_t)ldzp * zc, 0.0);
  }
  if (comm_failed(ctx)) return EIGX_ERR_INTERNAL;
  const double t3 = hnow();

  // ---- eigen_hrbakwyx on my column block; the reflector blocks stream past in groups of P ---------------------------------
  if (want_vec && n > 1) {
    const int nv = cn > 0 ? cn : 1;
    double* YA = ctx.pool.get_t<double>("h.YAB", (size_t)2 * 2 * HMB * nv);
    double* YB = YA + (size_t)2 * HMB * nv;
    double* Yr = ctx.pool.get_t<double>("h.Yr", (size_t)HMB * nv);
    double* Yi = ctx.pool.get_t<double>("h.Yi", (size_t)HMB * nv);
    double* Xr = ctx.pool.get_t<double>("h.Xr", (size_t)HMB * nv);
    double* Xi = ctx.pool.get_t<double>("h.Xi", (size_t)HMB * nv);
    const int lds = ld;
    const size_t SB = (size_t)lds * 2 * HMB + (size_t)2 * HMB * HMB;      // [Vs | Tr | Ti] of one block
    double* bsend = ctx.pool.get_t<double>("hs.bsend", SB);
    double* brecv = ctx.pool.get_t<double>("hs.brecv", SB * P);
    for (int g0 = 0; g0 < nblk; g0 += P) {
      {
        const int b = g0 + me;
        const int j0 = blk_j0(b), nb = (b < nblk) ? blk_j1(b) - j0 : 0;
        if (nb > 0) {
          hipLaunchKernelGGL(h_stack_v_kernel, dim3(8, 2 * nb), dim3(256), 0, st, H.Ar + hcol(H, j0), H.Ai + hcol(H, j0), ld,
                             j0 + nb - 1, nb, bsend, lds);
          EIGX_HIP_CHECK(hipMemcpyAsync(bsend + (size_t)lds * 2 * HMB, Tloc + (size_t)(b / P) * 2 * HMB * HMB,
                                        (size_t)2 * HMB * HMB * 8, hipMemcpyDeviceToDevice, st));
        }
      }
      comm_allgather(ctx, COMM_WORLD, bsend, brecv, SB, st);
      for (int q = 0; q < P && cn > 0; ++q) {
        const int b = g0 + q;
        if (b >= nblk) break;
        const int j0 = blk_j0(b), nb = blk_j1(b) - j0;
        if (nb <= 0) continue;
        const int rows = j0 + nb - 1;
        const double* Vs = brecv + (size_t)q * SB;
        const double* Tr = Vs + (size_t)lds * 2 * HMB;
        const double* Ti = Tr + (size_t)HMB * HMB;
        dgemm_dev(st, 'T', 'N', 2 * nb, cn, rows, 1.0, Vs, lds, Zr, ldzp, 0.0, YA, 2 * HMB, 0, nullptr, nullptr, nullptr, 2, 0,
                  (long)(Zi - Zr), (long)(YB - YA));
        hipLaunchKernelGGL(h_ycombine_kernel, dim3(1, cn), dim3(128), 0, st, YA, YB, nb, cn, Yr, Yi);
        dgemm_dev(st, 'N', 'N', nb, cn, nb, 1.0, Tr, HMB, Yr, HMB, 0.0, Xr, HMB);
        dgemm_dev(st, 'N', 'N', nb, cn, nb, -1.0, Ti, HMB, Yi, HMB, 1.0, Xr, HMB);
        dgemm_dev(st, 'N', 'N', nb, cn, nb, 1.0, Tr, HMB, Yi, HMB, 0.0, Xi, HMB);
        dgemm_dev(st, 'N', 'N', nb, cn, nb, 1.0, Ti, HMB, Yr, HMB, 1.0, Xi, HMB);
        hipLaunchKernelGGL(h_xstack_kernel, dim3(1, cn), dim3(128), 0, st, Xr, Xi, nb, cn, YA, YB);
        dgemm_dev(st, 'N', 'N', rows, cn, 2 * nb, -1.0, Vs, lds, YA, 2 * HMB, 1.0, Zr, ldzp);
        dgemm_dev(st, 'N', 'N', rows, cn, 2 * nb, -1.0, Vs, lds, YB, 2 * HMB, 1.0, Zi, ldzp);
      }
    }
  }
  // ---- exit: my column block -> the callers' cyclic blocks, plane by plane, then interleave --------------------------------
  if (want_vec) {
    const int ldt = pad_ld((nloc_r > 2 ? nloc_r : 2));
    const int nzc = local_count(nvec, G.Py, G.py);
    double* tr_ = ctx.pool.get_t<double>("hs.zr", (size_t)ldt * (nloc_c > 0 ? nloc_c : 1));
    double* ti_ = ctx.pool.get_t<double>("hs.zi", (size_t)ldt * (nloc_c > 0 ? nloc_c : 1));
    cols_to_cyclic_dev(ctx, n, nvec, 1, zc, c0, cn, Zr, ldzp, tr_, ldt, st);
    cols_to_cyclic_dev(ctx, n, nvec, 1, zc, c0, cn, Zi, ldzp, ti_, ldt, st);
    if (nzc > 0 && nloc_r > 0)
      hipLaunchKernelGGL(hs_join_cyclic_kernel, dim3(ceil_div(nloc_r, 256), nzc), dim3(256), 0, st, (const double*)tr_,
                         (const double*)ti_, ldt, nloc_r, nzc, z, ldz);
  }
  if (sigma != 1.0 && sigma != 0.0)
    hipLaunchKernelGGL(h_scale_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, n, 1.0 / sigma);
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  EIGX_HIP_CHECK(hipGetLastError());
  if (comm_failed(ctx)) return EIGX_ERR_INTERNAL;
  const double t4 = hnow();
  const double f_red = 4.0 / 3.0 * (double)n * n * n;
  const double f_dc = ctx.timers[11];
  const double f_bt = want_vec ? 2.0 * (double)nvec * n * n : 0.0;
  const double ret = f_red + f_dc + f_bt;
  ctx.timers[0] = t4 - t0; ctx.timers[1] = t2 - t1; ctx.timers[2] = t3 - t2; ctx.timers[3] = t4 - t3; ctx.timers[12] = ret;
  if (G.px == 0 && G.py == 0 && nloc_r > 0 && nloc_c > 0) {   // statistics a(1,1), a(2,1) on the owner of the first column
    const double stats[4] = {ret, 0.0, t4 - t0, 0.0};
    EIGX_HIP_CHECK(hipMemcpyAsync(a, stats, (size_t)(nloc_r >= 2 ? 4 : 2) * 8, hipMemcpyHostToDevice, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
  }
  return EIGX_OK;
}

namespace {
// complex (interleaved) versions of the 2-D cyclic layout kernels of solver.hip
__global__ void hz_pack_kernel(const double* __restrict__ a, int lda, int nr, int nc, double* __restrict__ out, int bx) {
  const int lj = blockIdx.y;
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < bx; li += gridDim.x * blockDim.x) {
    const bool ok = li < nr && lj < nc;
    out[2 * ((size_t)lj * bx + li)] = ok ? a[2 * ((size_t)lj * lda + li)] : 0.0;
    out[2 * ((size_t)lj * bx + li) + 1] = ok ? a[2 * ((size_t)lj * lda + li) + 1] : 0.0;
  }
}
__global__ void hz_cyclic_to_full_kernel(const double* __restrict__ recv, int bx, int by, int Px, int Py, int order_r, int n,
                                         double* __restrict__ F, int ldf) {
  const int q = blockIdx.z;
  const int qx = order_r ? q / Py : q % Px, qy = order_r ? q % Py : q / Px;
  const int lj = blockIdx.y;
  const int gj = lj * Py + qy;
  if (gj >= n) return;
  const double* src = recv + 2 * ((size_t)q * bx * by + (size_t)lj * bx);
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < bx; li += gridDim.x * blockDim.x) {
    const int gi = li * Px + qx;
    if (gi < n) { F[2 * ((size_t)gj * ldf + gi)] = src[2 * li]; F[2 * ((size_t)gj * ldf + gi) + 1] = src[2 * li + 1]; }
  }
}
__global__ void hz_full_to_cyclic_kernel(const double* __restrict__ F, int ldf, int nloc_r, int ncols, int Px, int px, int Py,
                                         int py, double* __restrict__ dst, int ldd) {
  const int lj = blockIdx.y;
  const int gj = lj * Py + py;
  if (gj >= ncols) return;
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < nloc_r; li += gridDim.x * blockDim.x) {
    dst[2 * ((size_t)lj * ldd + li)] = F[2 * ((size_t)gj * ldf + (size_t)li * Px + px)];
    dst[2 * ((size_t)lj * ldd + li) + 1] = F[2 * ((size_t)gj * ldf + (size_t)li * Px + px) + 1];
  }
}
}  // namespace

// a, z: this rank's 2-D cyclic blocks (device, interleaved complex), as for eigen_sx / eigen_s
int herm_solve_dev(Context& ctx, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb,
                   char mode) {
  if (!ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  const Grid& G = ctx.grid;
  if (G.nranks == 1) return herm_solve_full(ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode);
  if (n <= 0) return EIGX_ERR_BAD_ARG;
  const int nloc_r = local_count(n, G.Px, G.px), nloc_c = local_count(n, G.Py, G.py);
  if (!a || !w || lda < nloc_r) return EIGX_ERR_BAD_ARG;
  char md = mode;
  if (md >= 'a' && md <= 'z') md = (char)(md - 'a' + 'A');
  int nv = nvec < 0 ? -nvec : nvec;
  if (nv > n) nv = n;
  const bool want_vec = !(md == 'N' || nv == 0);
  if (want_vec && (!z || ldz < nloc_r)) return EIGX_ERR_BAD_ARG;
  EIGX_HIP_CHECK(hipSetDevice(ctx.device));
  // the sharded form (nothing gathered) is the default; EIGX_H_GATHER=1 keeps the first version -- gather the matrix,
  // reduce it on every rank -- for comparisons
  static const bool gather = [] { const char* e = getenv("EIGX_H_GATHER"); return e && atoi(e) != 0; }();
  if (!gather) return herm_solve_sharded(ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode);
  hipStream_t st = ctx.stream;
  const int bx = ceil_div(n, G.Px), by = ceil_div(n, G.Py);
  const int ldf = n + 2;
  double* sendb = ctx.pool.get_t<double>("hm.send", (size_t)2 * bx * by);
  double* recvb = ctx.pool.get_t<double>("hm.recv", (size_t)2 * bx * by * G.nranks);
  double* Af = ctx.pool.get_t<double>("hm.A", (size_t)2 * ldf * n);
  double* Zf = ctx.pool.get_t<double>("hm.Z", (size_t)2 * ldf * n);
  hipLaunchKernelGGL(hz_pack_kernel, dim3(8, by), dim3(256), 0, st, a, lda, nloc_r, nloc_c, sendb, bx);
  comm_allgather(ctx, COMM_WORLD, sendb, recvb, (size_t)2 * bx * by, st);
  hipLaunchKernelGGL(hz_cyclic_to_full_kernel, dim3(8, by, G.nranks), dim3(256), 0, st, recvb, bx, by, G.Px, G.Py,
                     G.row_major, n, Af, ldf);
  int members[EIGX_MAXP], mine = 0;
  const int np = comm_group(ctx, COMM_WORLD, members, &mine);
  const int rc = herm_solve_full(ctx, n, nvec, Af, ldf, w, Zf, ldf, mf, mb, mode, np, mine);
  if (rc != EIGX_OK) return rc;
  if (want_vec) {
    const int nzc = local_count(nv, G.Py, G.py);
    if (nzc > 0 && nloc_r > 0)
      hipLaunchKernelGGL(hz_full_to_cyclic_kernel, dim3(8, nzc), dim3(256), 0, st, Zf, ldf, nloc_r, nv, G.Px, G.px, G.Py,
                         G.py, z, ldz);
  }
  if (G.px == 0 && G.py == 0 && nloc_r > 0 && nloc_c > 0)   // statistics a(1,1), a(2,1) on the owner of the first column
    EIGX_HIP_CHECK(hipMemcpyAsync(a, Af, (size_t)(nloc_r >= 2 ? 4 : 2) * 8, hipMemcpyDeviceToDevice, st));
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  return EIGX_OK;
}

int herm_solve_host(Context& ctx, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb,
                    char mode) {
  if (!ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  const int nr = local_count(n, ctx.grid.Px, ctx.grid.px), nc = local_count(n, ctx.grid.Py, ctx.grid.py);
  if (n <= 0 || !a || !w || lda < nr) return EIGX_ERR_BAD_ARG;
  EIGX_HIP_CHECK(hipSetDevice(ctx.device));
  const int ldd = nr + 2;
  const int ncd = nc > 0 ? nc : 1;
  double* ad = ctx.pool.get_t<double>("hh.a", (size_t)2 * ldd * ncd);
  double* zd = ctx.pool.get_t<double>("hh.z", (size_t)2 * ldd * ncd);
  double* wd = ctx.pool.get_t<double>("hh.w", (size_t)n);
  if (nr > 0 && nc > 0)
    EIGX_HIP_CHECK(hipMemcpy2D(ad, (size_t)ldd * 16, a, (size_t)lda * 16, (size_t)nr * 16, (size_t)nc, hipMemcpyHostToDevice));
  const int rc = herm_solve_dev(ctx, n, nvec, ad, ldd, wd, zd, ldd, mf, mb, mode);
  EIGX_HIP_CHECK(hipMemcpy(w, wd, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (rc != EIGX_OK) return rc;
  char md = mode;
  if (md >= 'a' && md <= 'z') md = (char)(md - 'a' + 'A');
  int nv = nvec < 0 ? -nvec : nvec;
  if (nv > n) nv = n;
  const int nzc = local_count(nv, ctx.grid.Py, ctx.grid.py);
  if (z && nzc > 0 && nr > 0 && md != 'N')
    EIGX_HIP_CHECK(hipMemcpy2D(z, (size_t)ldz * 16, zd, (size_t)ldd * 16, (size_t)nr * 16, (size_t)nzc, hipMemcpyDeviceToHost));
  if (nr > 0 && nc > 0)
    EIGX_HIP_CHECK(hipMemcpy(a, ad, (size_t)(nr >= 2 ? 4 : 2) * 8, hipMemcpyDeviceToHost));   // statistics only: a is destroyed
  return EIGX_OK;
}

}  // namespace eigx

using namespace eigx;

extern "C" {

int eigx_h(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb, char mode) {
  return eigx_guard(g_ctx, [&] { return herm_solve_host(g_ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode); });
}
int eigx_h_dev(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int mf, int mb, char mode) {
  return eigx_guard(g_ctx, [&] { return herm_solve_dev(g_ctx, n, nvec, a, lda, w, z, ldz, mf, mb, mode); });
}

}
