// dc.hip -- divide and conquer for the symmetric band matrix (half-bandwidth 1 or 2), gfx950.
//
// Replaces (reference paths relative to RIKEN-RCCS/EigenExa 2.13):
//   band=1 : eigen_dc2 / MX_PDSTEDC / MX_PDLAED0-3,Z      src/dc2.F, src/mx_pd*.F
//            dc2_FS / FS_EDC / FS_PDLAED0-3 / FS_PDLAEDZ   src/dc2_FS.F, src/FS_*.F90
//   band=2 : eigen_dcx / MY_PDSxEDC / MY_PDLAED0-3,Z       src/dcx.F:81-337, src/my_pd*.F
//            (tear = SVD of the 2x2 coupling block, one rank-one merge per singular triplet,
//             src/my_pdlaed0.F:213-266, :312-408)
//   leaves : LAPACK_EIGEN2/DSYEVD src/lapack_eigen.F:31-61, DSTEQR src/mx_pdlaed0.F:182,
//            DSTEDC src/FS_PDLAED0.F90:178
//   sort   : MY_PDLASRT src/my_pdlasrt.F, FS_PDLASRT src/FS_PDLASRT.F90
//
// MI355X design: the whole tree lives on one GPU; Q (n x n) is block diagonal at every height, so a
// height is processed as one batch:
//   leaves         one workgroup per leaf: parallel cyclic Jacobi on the dense <=64x64 block in LDS
//   z = Q^T w      gather of <= 4 rows of Q per merge
//   deflation      the one inherently serial scan (DLAED2 logic: src/FS_PDLAED2.F90:232-233 tolerance,
//                  :348-383 Givens) runs on the host on 2n doubles per height (the only host round trip)
//   rotations      Givens rotations of eigenvector columns, one thread per row
//   secular eq.    one thread per root, middle-way rational iteration + bisection safeguard
//                  (role of DLAED4 at src/my_pdlaed3.F:276,490, src/FS_PDLAED3.F90:281,700,795)
//   Loewner        Gu-Eisenstat z-hat: one wave per pole, product over the roots
//   eigenvectors   one thread per root, normalised columns of S
//   Q <- Q S       fp64 MFMA GEMM with a column-gather map on A (non-deflated columns) so that no
//                  pack / permute pass touches HBM (role of PDGEMM src/my_pdlaed1.F:310-341 and of the
//                  DGEMM ring src/FS_PDLAED3.F90:833-860)
//   eigenvalues stay unsorted between merges (the next deflation sorts anyway); one final sort +
//   column permutation writes z.
#include "eigx_context.h"
#include "eigx_comm.h"
#include "../../include/eigenexa_amd.h"
#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cstring>

namespace eigx {

namespace {

constexpr int LEAF = 32;

// ================================================================================================
// leaves: cyclic Jacobi with round-robin pairing, one workgroup per leaf
// ================================================================================================
// One WAVE per leaf (64-thread workgroups: the barriers below cost nothing, every CU runs several leaves side by side).
// Round 4: the 256-thread form spent its time in three workgroup barriers per round of <= 16 rotations (1.3 ms for the
// 256 leaves of N = 8192).
constexpr int LEAF_T = 64;
__global__ __launch_bounds__(LEAF_T) void jacobi_leaf_kernel(const double* __restrict__ d,
                                                             const double* __restrict__ e, int lde, int band,
                                                             const int* __restrict__ leaf_off,
                                                             const int* __restrict__ leaf_n, double* __restrict__ D,
                                                             double* __restrict__ Q, int ldq, int r0, int r1) {
  __shared__ double A[LEAF][LEAF + 1];
  __shared__ double V[LEAF][LEAF + 1];
  __shared__ double cs[LEAF / 2][2];
  __shared__ int pq[LEAF / 2][2];
  __shared__ int perm[LEAF];
  const int tid = threadIdx.x;
  const int off = leaf_off[blockIdx.x], m = leaf_n[blockIdx.x];
  double mx = 0.0;
  for (int idx = tid; idx < LEAF * LEAF; idx += LEAF_T) {
    const int r = idx / LEAF, c = idx % LEAF;
    double v = 0.0;
    if (r < m && c < m) {
      const int lo = r < c ? r : c, hi = r < c ? c : r;
      const int dist = hi - lo;
      if (dist == 0) v = d[off + r];
      else if (dist <= band) v = e[(size_t)(dist - 1) * lde + off + hi];  // e(hi, dist) = T(hi-dist, hi)
    }
    A[r][c] = v;
    V[r][c] = (r == c) ? 1.0 : 0.0;
    mx = fmax(mx, fabs(v));
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  __syncthreads();
  const double thr = 2e-18 * mx;
  const double isc = (mx > 0.0) ? ldexp(1.0, -ilogb(mx)) : 1.0;   // the rotation angle is scale-free: keeps dd^2 + b^2 in range
  const int m2 = m + (m & 1);
  const int np = m2 / 2;
  // Thread layout of the update phases: lane -> (row/column index k = tid & 31, pairs t = (tid >> 5) + 2 it, it < 8).  All
  // loads of a phase are issued before its first store (the pairs of a round touch disjoint rows / columns), so a phase
  // costs one LDS round trip instead of one per pair: a leaf is a single wave with nothing else to hide latency behind.
  const int kk = tid & (LEAF - 1), th = tid >> 5;
  const bool kact = kk < m;
  for (int sweep = 0; sweep < 40; ++sweep) {
    int nrot = 0;
    for (int rnd = 0; rnd < m2 - 1; ++rnd) {
      bool rot = false;
      if (tid < np) {
        int p, q;
        if (tid == 0) { p = m2 - 1; q = rnd; }
        else {
          p = rnd + tid; if (p >= m2 - 1) p -= m2 - 1;
          q = rnd - tid; if (q < 0) q += m2 - 1;
        }
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, s = 0.0;
        if (q < m) {
          const double apq = A[p][q];
          if (fabs(apq) > thr) {
            // half-angle form with two reciprocal square roots (hardware estimate + two Newton steps each) instead of
            // three divisions and two square roots: dd = a_qq - a_pp, b = 2 a_pq, r = sqrt(dd^2 + b^2),
            // c = sqrt((1 + |dd|/r)/2), s = sign(dd b) |b| / (2 r c)   (tan of the smaller rotation angle, as before)
            const double dd_ = (A[q][q] - A[p][p]) * isc, b = 2.0 * apq * isc;
            const double x = dd_ * dd_ + b * b;
            double ir = __builtin_amdgcn_rsq(x);
            ir = ir * (1.5 - 0.5 * x * ir * ir);
            ir = ir * (1.5 - 0.5 * x * ir * ir);
            const double c2 = 0.5 + 0.5 * fabs(dd_) * ir;     // in [0.5, 1]
            double ic = __builtin_amdgcn_rsq(c2);
            ic = ic * (1.5 - 0.5 * c2 * ic * ic);
            ic = ic * (1.5 - 0.5 * c2 * ic * ic);
            c = c2 * ic;
            s = 0.5 * b * ir * ic;
            if (dd_ < 0.0) s = -s;
            rot = true;
          }
        }
        cs[tid][0] = c; cs[tid][1] = s;
        pq[tid][0] = p; pq[tid][1] = q;
      }
      const bool any = __ballot(rot) != 0ull;
      __syncthreads();
      if (!any) continue;   // wave-uniform: nothing to rotate in this round
      ++nrot;
      // columns: A <- A J, V <- V J
      {
        double c_[8], s_[8], a0[8], a1[8], v0[8], v1[8];
        int p_[8], q_[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int t = th + 2 * it;
          const bool on = kact && t < np;
          c_[it] = on ? cs[t][0] : 1.0; s_[it] = on ? cs[t][1] : 0.0;
          p_[it] = on ? pq[t][0] : 0; q_[it] = on ? pq[t][1] : 0;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          a0[it] = A[kk][p_[it]]; a1[it] = A[kk][q_[it]];
          v0[it] = V[kk][p_[it]]; v1[it] = V[kk][q_[it]];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it)
          if (s_[it] != 0.0) {
            A[kk][p_[it]] = c_[it] * a0[it] - s_[it] * a1[it];
            A[kk][q_[it]] = s_[it] * a0[it] + c_[it] * a1[it];
            V[kk][p_[it]] = c_[it] * v0[it] - s_[it] * v1[it];
            V[kk][q_[it]] = s_[it] * v0[it] + c_[it] * v1[it];
          }
        __syncthreads();
        // rows: A <- J^T A
#pragma unroll
        for (int it = 0; it < 8; ++it) { a0[it] = A[p_[it]][kk]; a1[it] = A[q_[it]][kk]; }
#pragma unroll
        for (int it = 0; it < 8; ++it)
          if (s_[it] != 0.0) {
            A[p_[it]][kk] = c_[it] * a0[it] - s_[it] * a1[it];
            A[q_[it]][kk] = s_[it] * a0[it] + c_[it] * a1[it];
          }
      }
      __syncthreads();
    }
    if (nrot == 0) break;   // wave-uniform
  }
  // ascending order by rank counting (ties by index)
  if (tid < m) {
    const double v = A[tid][tid];
    int rank = 0;
    for (int i = 0; i < m; ++i) {
      const double u = A[i][i];
      rank += (u < v || (u == v && i < tid)) ? 1 : 0;
    }
    perm[rank] = tid;
    D[off + rank] = v;
  }
  __syncthreads();
  // one Newton-Schulz step V <- V - V (V^T V - I)/2 : removes the O(sqrt(#rotations) eps) loss of
  // orthogonality that the rotation products accumulate (A is free now and holds E = V^T V - I)
  for (int idx = tid; idx < m * m; idx += LEAF_T) {
    const int r = idx / m, c = idx - r * m;
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += V[k][r] * V[k][c];
    A[r][c] = acc - (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int idx = tid; idx < m * m; idx += LEAF_T) {
    const int j = idx / m, r = idx - j * m;
    const int pj = perm[j];
    double acc = 0.0;
    for (int k = 0; k < m; ++k) acc += V[r][k] * A[k][pj];
    if (off + r >= r0 && off + r < r1)   // several GPUs: a rank keeps rows [r0, r1) of Q only
      Q[(size_t)(off + j) * ldq + off + r] = V[r][pj] - 0.5 * acc;
  }
}

// ================================================================================================
// per-merge descriptors (device arrays, one entry per merge of the current height)
// ================================================================================================
struct MergeDev {
  int off, nm, n1, K;
  int rot_beg, rot_end;
  double rho;
  double wv[4];  // weights of rows off+n1-band .. off+n1+band-1 in z = Q^T w
  // z of the NEXT pass, formed ahead of this pass's product (znext kernels): its rows that lie inside this merge
  int zr_n, zr_pad;
  int zr_row[4];
  double zr_w[4];
};

__global__ void zgather_kernel(const MergeDev* __restrict__ md, int band, const double* __restrict__ Q, int ldq,
                               double* __restrict__ z, int r0, int r1) {
  const MergeDev M = md[blockIdx.y];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M.nm) return;
  const int row0 = M.off + M.n1 - band;
  const double* col = Q + (size_t)(M.off + j) * ldq + row0;
  double acc = 0.0;
  for (int t = 0; t < 2 * band; ++t)
    if (row0 + t >= r0 && row0 + t < r1) acc += M.wv[t] * col[t];  // rows owned by this rank (all rows if P = 1)
  z[M.off + j] = acc;
}

__global__ void rotate_kernel(const MergeDev* __restrict__ md, const int* __restrict__ rpj,
                              const int* __restrict__ rjj, const double* __restrict__ rc,
                              const double* __restrict__ rsn, double* __restrict__ Q, int ldq, int r0, int r1) {
  const MergeDev M = md[blockIdx.y];
  if (M.rot_end <= M.rot_beg) return;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M.nm || M.off + r < r0 || M.off + r >= r1) return;
  double* row = Q + M.off + r;
  for (int t = M.rot_beg; t < M.rot_end; ++t) {
    const size_t cp = (size_t)rpj[t] * ldq, cj = (size_t)rjj[t] * ldq;
    const double c = rc[t], s = rsn[t];
    const double x = row[cp], y = row[cj];
    row[cp] = c * x + s * y;
    row[cj] = c * y - s * x;
  }
}

// ---- z of the next pass ahead of this pass's product (one GPU) -----------------------------------------------------
// The next pass needs z' = Q'^T w' with Q' = this pass's result, a few rows of it only: Q'(r, off+j) = sum_i Q(r, nd_i) U(i,j)
// for the K roots, Q'(r, dst) = Q(r, src) for the deflated columns.  So z'(off+j) = sum_i y_i U(i,j) with
// y_i = sum_t w'_t Q(row_t, nd_i): a K x K mat-vec on the eigenvector rows S'(j,i) = U(i,j) that exists BEFORE the
// O(n K^2) product.  With it the host deflates the next pass, and the device solves its secular equations, while the
// product runs (the reference overlaps nothing here: PDLAED2 -> PDLAED3 -> PDGEMM in sequence, src/my_pdlaed1.F:225-341).
// Two-phase and deterministic like the eigenvector norms: partial sums per chunk of ZN_IC poles, then a fixed-order sum.
constexpr int ZN_IC = 256;
__global__ __launch_bounds__(256) void znext1_kernel(const MergeDev* __restrict__ md, const int* __restrict__ nd,
                                                     const double* __restrict__ Qa, int ldq, const double* __restrict__ S,
                                                     int lds, double* __restrict__ zp, int ldn) {
  __shared__ double ys[ZN_IC];
  __shared__ double part[4][64];
  const MergeDev M = md[blockIdx.z];
  const int K = M.K;
  if ((int)(blockIdx.x * 64) >= K || (int)(blockIdx.y * ZN_IC) >= K) return;
  const int i0 = blockIdx.y * ZN_IC;
  const int i1 = (i0 + ZN_IC < K) ? i0 + ZN_IC : K;
  {
    const int i = i0 + threadIdx.x;
    double y = 0.0;
    if (i < i1) {
      const double* col = Qa + (size_t)nd[M.off + i] * ldq;
      for (int t = 0; t < M.zr_n; ++t) y += M.zr_w[t] * col[M.zr_row[t]];
    }
    ys[threadIdx.x] = y;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const bool act = j < K;
  const double* Sp = S + (size_t)M.off * lds + M.off + j;
  double acc = 0.0;
  if (act)
    for (int i = i0 + wave; i < i1; i += 4) acc += ys[i - i0] * Sp[(size_t)i * lds];
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && act)
    zp[(size_t)blockIdx.y * ldn + M.off + j] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}
__global__ void znext2_kernel(const MergeDev* __restrict__ md, const int* __restrict__ dsrc, const double* __restrict__ Qa,
                              int ldq, const double* __restrict__ zp, int ldn, double* __restrict__ z) {
  const MergeDev M = md[blockIdx.y];
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj >= M.nm) return;
  double v = 0.0;
  if (jj < M.K) {
    const int nch = (M.K + ZN_IC - 1) / ZN_IC;
    for (int c = 0; c < nch; ++c) v += zp[(size_t)c * ldn + M.off + jj];
  } else {
    const double* col = Qa + (size_t)dsrc[M.off + jj] * ldq;   // deflated: the column is copied as it is
    for (int t = 0; t < M.zr_n; ++t) v += M.zr_w[t] * col[M.zr_row[t]];
  }
  z[M.off + jj] = v;
}

// row-block exchange of the multi-GPU D&C: pack rows [r0, r0+nr) of Q(:, 0:n) / unpack all ranks' blocks
__global__ void pack_rows_kernel(const double* __restrict__ Q, int ldq, int n, int r0, int nr, int rp,
                                 double* __restrict__ out) {
  const int j = blockIdx.y;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rp; r += gridDim.x * blockDim.x)
    out[(size_t)j * rp + r] = (r < nr) ? Q[(size_t)j * ldq + r0 + r] : 0.0;
}
__global__ void unpack_rows_kernel(const double* __restrict__ in, int n, int rp, double* __restrict__ Q, int ldq) {
  const int j = blockIdx.y, q = blockIdx.z;
  const double* src = in + (size_t)q * rp * n + (size_t)j * rp;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rp; r += gridDim.x * blockDim.x)
    if (q * rp + r < n) Q[(size_t)j * ldq + q * rp + r] = src[r];
}

// ================================================================================================
// secular equation: 8 lanes per root (32 roots per 256-thread workgroup).  f(x) = 1/rho + sum_i z_i^2/(d_i - x),
// d ascending, ||z|| = 1.  Lane `sub` of a root's 8-lane group owns the poles i = sub (mod 8); sums are
// combined with an xor butterfly (bit-identical in all 8 lanes, so the group branches uniformly).
// Writes lambda_j to Dn[off+j] and S'(j,i) = d_i - lambda_j (root index contiguous).
// ================================================================================================
template <int CTRL>
__device__ __forceinline__ double dc_dpp_add(double v) {
  const int lo2 = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return v + __hiloint2double(hi2, lo2);
}

__device__ __forceinline__ double dc_swapadd16(double v) {   // v[row] + v[row ^ 1] over rows of 16 lanes (v_permlane16_swap)
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

__device__ __forceinline__ double group8_sum(double v) {
  // quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror: the 8 lanes of a group by VALU DPP moves instead of
  // three dependent LDS-crossbar shuffles (this kernel is a latency chain: five such sums per secular iteration)
  v = dc_dpp_add<0xB1>(v);
  v = dc_dpp_add<0x4E>(v);
  v = dc_dpp_add<0x141>(v);
  return v;
}

// LPR lanes per root: 8 (32 roots per workgroup) for the small merges, 32 (8 roots per workgroup) for K >= 512, where
// 8 lanes per root leave the chip at under one wave per SIMD (top merge of N=8192: 1.9 ms of dependent fp64 divisions)
template <int LPR>
__device__ __forceinline__ double group_sum(double v) {
  v = group8_sum(v);
  if (LPR == 32) { v = dc_dpp_add<0x140>(v); v = dc_swapadd16(v); }   // row_mirror: 16 lanes; lane swap: the row pair
  return v;
}

// Several GPUs (P > 1): the roots of every merge are split over the ranks by root index -- rank r solves roots
// [K r / P, K (r + 1) / P) -- and nothing K x K is stored: a root is kept as (origin_j, tau_j), from which
// S'(j, i) = (d_i - origin_j) - tau_j is recomputed wherever it is needed; sec = [lambda | origin | tau] (n each) is
// summed over the ranks afterwards (every entry is written by exactly one rank).
template <int LPR>
__global__ __launch_bounds__(256) void secular_kernel(const MergeDev* __restrict__ md,
                                                      const double* __restrict__ dlam,
                                                      const double* __restrict__ wz, double* __restrict__ Dn,
                                                      double* __restrict__ S, int lds, int P, int rank,
                                                      double* __restrict__ sec, int nsec) {
  const MergeDev M = md[blockIdx.y];
  const int K = M.K;
  constexpr int RPW = 256 / LPR;   // roots per workgroup
  __shared__ double so[RPW], stau[RPW];
  const int jbase = (P > 1) ? (int)((long)K * rank / P) : 0;
  const int jend = (P > 1) ? (int)((long)K * (rank + 1) / P) : K;
  if (jbase + (int)(blockIdx.x * RPW) >= jend) return;
  const int sub = threadIdx.x & (LPR - 1);
  const int jraw = jbase + blockIdx.x * RPW + threadIdx.x / LPR;
  const bool act = jraw < jend;
  const int j = act ? jraw : K - 1;  // idle groups recompute the last root (keeps every lane in the shuffles)
  const double* __restrict__ d = dlam + M.off;
  const double* __restrict__ z = wz + M.off;
  double* Sp = S + (size_t)M.off * lds + M.off;  // S'(j,i) at Sp[j + i*lds]
  const double rho = M.rho, rhoinv = 1.0 / rho;
  const double eps = DBL_EPSILON / 2.0;
  if (K == 1) {
    if (threadIdx.x == 0) {
      const double t = rho * z[0] * z[0];
      if (sec) { sec[M.off] = d[0] + t; sec[nsec + M.off] = d[0]; sec[2 * nsec + M.off] = t; }
      else { Dn[M.off] = d[0] + t; Sp[0] = -t; }
    }
    return;
  }
  const bool last = (j == K - 1);
  const int jl = last ? K - 2 : j, jr = jl + 1;
  double origin, lo, hi, tau;
  {
    const double gap = last ? rho : d[j + 1] - d[j];
    const double mid = last ? d[K - 1] + 0.5 * rho : 0.5 * (d[j] + d[j + 1]);
    double c = 0.0;
    for (int i = sub; i < K; i += LPR)
      if (i != jl && i != jr) c += z[i] * z[i] / (d[i] - mid);
    c = group_sum<LPR>(c) + rhoinv;
    const double zl2 = z[jl] * z[jl], zr2 = z[jr] * z[jr];
    const double fmid = c + zl2 / (d[jl] - mid) + zr2 / (d[jr] - mid);
    int org;
    if (last) { org = K - 1; if (fmid <= 0.0) { lo = 0.5 * gap; hi = gap; } else { lo = 0.0; hi = 0.5 * gap; } }
    else if (fmid > 0.0) { org = j; lo = 0.0; hi = 0.5 * gap; }
    else { org = j + 1; lo = -0.5 * gap; hi = 0.0; }
    origin = d[org];
    const double dl = d[jl] - origin, dr = d[jr] - origin;
    const double qa = c, qb = -(c * (dl + dr) + zl2 + zr2), qc = c * dl * dr + zl2 * dr + zr2 * dl;
    tau = 0.5 * (lo + hi);
    const double disc = qb * qb - 4.0 * qa * qc;
    if (disc >= 0.0) {
      const double sq = sqrt(disc);
      const double qq = -0.5 * (qb + (qb >= 0.0 ? sq : -sq));
      const double r1 = (qa != 0.0) ? qq / qa : NAN, r2 = (qq != 0.0) ? qc / qq : NAN;
      if (r1 > lo && r1 < hi) tau = r1;
      else if (r2 > lo && r2 < hi) tau = r2;
    }
  }
  for (int iter = 0; iter < 100; ++iter) {
    double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0, err = 0.0;
    for (int i = sub; i < K; i += LPR) {
      const double t = z[i] / ((d[i] - origin) - tau);
      const double zt = z[i] * t, tt = t * t;
      if (i <= jl) { psi += zt; dpsi += tt; err += fabs(psi); }
      else { phi += zt; dphi += tt; err += fabs(phi); }
    }
    psi = group_sum<LPR>(psi); dpsi = group_sum<LPR>(dpsi);
    phi = group_sum<LPR>(phi); dphi = group_sum<LPR>(dphi);
    err = group_sum<LPR>(err);
    const double wv = rhoinv + phi + psi;
    err = 8.0 * (fabs(phi) + fabs(psi)) + err + 2.0 * rhoinv + fabs(tau) * (dpsi + dphi);
    if (fabs(wv) <= eps * err) break;
    if (wv < 0.0) lo = tau; else hi = tau;
    const double Dl = (d[jl] - origin) - tau, Dr = (d[jr] - origin) - tau;
    const double aa = (Dl + Dr) * wv - Dl * Dr * (dpsi + dphi);
    const double bb = Dl * Dr * wv;
    const double cc = wv - Dl * dpsi - Dr * dphi;
    double eta;
    {
      double disc = aa * aa - 4.0 * bb * cc;
      if (disc < 0.0) disc = 0.0;
      const double sq = sqrt(disc);
      if (cc == 0.0) eta = (aa != 0.0) ? bb / aa : 0.0;
      else if (aa <= 0.0) eta = (aa - sq) / (2.0 * cc);
      else eta = 2.0 * bb / (aa + sq);
    }
    if (wv * eta >= 0.0) eta = -wv / (dpsi + dphi);
    double tnew = tau + eta;
    if (!(tnew > lo && tnew < hi)) tnew = 0.5 * (lo + hi);
    if (tnew == tau) break;
    tau = tnew;
  }
  // S'(j, i) = (d_i - origin_j) - tau_j: the workgroup's RPW roots are consecutive j, so the store is re-mapped to
  // lanes = (root, pole) pairs with the root index fastest: RPW * 8-byte segments instead of one element per line
  if (sec) {
    if (sub == 0 && act) { sec[M.off + j] = origin + tau; sec[nsec + M.off + j] = origin; sec[2 * nsec + M.off + j] = tau; }
    return;
  }
  if (sub == 0) { so[threadIdx.x / LPR] = origin; stau[threadIdx.x / LPR] = tau; if (act) Dn[M.off + j] = origin + tau; }
  __syncthreads();
  {
    const int r = threadIdx.x % RPW, ii = threadIdx.x / RPW;
    const int jr = blockIdx.x * RPW + r;
    if (jr < K) {
      const double o_ = so[r], t_ = stau[r];
      for (int i = ii; i < K; i += 256 / RPW) Sp[jr + (size_t)i * lds] = (d[i] - o_) - t_;
    }
  }
}

// Gu-Eisenstat: zhat_i^2 = prod_j (lam_j - d_i) / prod_{j != i} (d_j - d_i); one wave per pole i
__global__ __launch_bounds__(256) void loewner_kernel(const MergeDev* __restrict__ md,
                                                      const double* __restrict__ dlam,
                                                      const double* __restrict__ wz, const double* __restrict__ S,
                                                      int lds, double* __restrict__ zh) {
  const MergeDev M = md[blockIdx.y];
  const int K = M.K;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= K) return;
  const double* d = dlam + M.off;
  const double* Sp = S + (size_t)M.off * lds + M.off + (size_t)i * lds;  // S'(:, i)
  const double di = d[i];
  double prod = 1.0;
  for (int j = lane; j < K; j += 64) {
    const double num = -Sp[j];
    prod *= (j == i) ? num : num / (d[j] - di);
  }
  for (int o = 32; o > 0; o >>= 1) prod *= __shfl_xor(prod, o, 64);
  if (lane == 0) {
    const double v = sqrt(fabs(prod));
    zh[M.off + i] = (wz[M.off + i] >= 0.0) ? v : -v;
  }
}

// eigenvectors of the rank-one problem: S'(j,i) <- zhat_i / (d_i - lam_j), normalised over i.
// workgroup = 64 roots (lanes) x 4 waves (wave w owns the poles i = w mod 4): coalesced along j.
// eigenvectors of the rank-one update: S'(j,i) <- zh_i / S'(j,i), then every root's row j normalised.
// Two launches over a 2-D grid (64 roots x VEC_IC poles per workgroup) so that the K x K matrix is streamed by the
// whole chip: pass 1 writes the quotients and per-chunk partial norms (deterministic two-phase reduction), pass 2
// re-reduces the partials and scales.  (One workgroup per 64 roots looping over all K poles reached 0.9 TB/s on
// the top merges.)
constexpr int VEC_IC = 256;
__global__ __launch_bounds__(256) void vectors1_kernel(const MergeDev* __restrict__ md, const double* __restrict__ zh,
                                                       double* __restrict__ S, int lds, double* __restrict__ vnp,
                                                       int ldn) {
  __shared__ double part[4][64];
  const MergeDev M = md[blockIdx.z];
  const int K = M.K;
  if ((int)(blockIdx.x * 64) >= K || (int)(blockIdx.y * VEC_IC) >= K) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const bool act = j < K;
  double* Sp = S + (size_t)M.off * lds + M.off + j;
  const double* zz = zh + M.off;
  const int i0 = blockIdx.y * VEC_IC;
  const int i1 = (i0 + VEC_IC < K) ? i0 + VEC_IC : K;
  double nrm = 0.0;
  if (act)
    for (int i = i0 + wave; i < i1; i += 4) {
      const double v = zz[i] / Sp[(size_t)i * lds];
      Sp[(size_t)i * lds] = v;
      nrm += v * v;
    }
  part[wave][lane] = nrm;
  __syncthreads();
  if (wave == 0 && act)
    vnp[(size_t)blockIdx.y * ldn + M.off + j] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}
__global__ __launch_bounds__(256) void vectors2_kernel(const MergeDev* __restrict__ md, double* __restrict__ S, int lds,
                                                       const double* __restrict__ vnp, int ldn) {
  const MergeDev M = md[blockIdx.z];
  const int K = M.K;
  if ((int)(blockIdx.x * 64) >= K || (int)(blockIdx.y * VEC_IC) >= K) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  if (j >= K) return;
  const int nch = (K + VEC_IC - 1) / VEC_IC;
  double nrm = 0.0;
  for (int c = 0; c < nch; ++c) nrm += vnp[(size_t)c * ldn + M.off + j];
  const double sc = 1.0 / sqrt(nrm);
  double* Sp = S + (size_t)M.off * lds + M.off + j;
  const int i0 = blockIdx.y * VEC_IC;
  const int i1 = (i0 + VEC_IC < K) ? i0 + VEC_IC : K;
  for (int i = i0 + wave; i < i1; i += 4) Sp[(size_t)i * lds] *= sc;
}

// ---- several GPUs: the same three steps from (origin, tau) instead of a stored K x K matrix ----------------------
// lambda of the non-deflated roots into the eigenvalue array (the deflated ones are already there)
__global__ void scatter_lambda_kernel(const MergeDev* __restrict__ md, const double* __restrict__ sec, double* __restrict__ Dn) {
  const MergeDev M = md[blockIdx.y];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < M.K) Dn[M.off + j] = sec[M.off + j];
}
// Gu-Eisenstat zhat (see loewner_kernel): lam_j - d_i = (origin_j - d_i) + tau_j
__global__ __launch_bounds__(256) void loewner_mg_kernel(const MergeDev* __restrict__ md, const double* __restrict__ dlam,
                                                         const double* __restrict__ wz, const double* __restrict__ sec,
                                                         int nsec, double* __restrict__ zh) {
  const MergeDev M = md[blockIdx.y];
  const int K = M.K;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= K) return;
  const double* d = dlam + M.off;
  const double* org = sec + nsec + M.off;
  const double* tau = sec + 2 * nsec + M.off;
  const double di = d[i];
  double prod = 1.0;
  for (int j = lane; j < K; j += 64) {
    const double num = (org[j] - di) + tau[j];
    prod *= (j == i) ? num : num / (d[j] - di);
  }
  for (int o = 32; o > 0; o >>= 1) prod *= __shfl_xor(prod, o, 64);
  if (lane == 0) {
    const double v = sqrt(fabs(prod));
    zh[M.off + i] = (wz[M.off + i] >= 0.0) ? v : -v;
  }
}
// normalised eigenvector rows of the roots [j0, j1) of merge number mi0 + blockIdx.z:
//   Sc(j - j0, i) = zh_i / ((d_i - origin_j) - tau_j) / norm_j   at Sc[base + (j - j0) + i * ldsc],
// base = M.off * ldsc when `compact` (all merges of a height side by side), 0 otherwise (one chunk of one merge).
// workgroup = 64 roots (lanes) x 4 waves striding the poles; two sweeps (norm, then store): nothing K x K is read.
__global__ __launch_bounds__(256) void vectors_mg_kernel(const MergeDev* __restrict__ md, int mi0, const double* __restrict__ dlam,
                                                         const double* __restrict__ zh, const double* __restrict__ sec, int nsec,
                                                         int j0, int j1, int compact, double* __restrict__ Sc, int ldsc) {
  __shared__ double part[4][64];
  const MergeDev M = md[mi0 + blockIdx.z];
  const int K = M.K;
  const int jhi = (j1 < K) ? j1 : K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = j0 + blockIdx.x * 64 + lane;
  if (j0 + (int)(blockIdx.x * 64) >= jhi) return;
  const bool act = j < jhi;
  const double* d = dlam + M.off;
  const double* zz = zh + M.off;
  const double oj = act ? sec[nsec + M.off + j] : 0.0, tj = act ? sec[2 * nsec + M.off + j] : 1.0;
  double nrm = 0.0;
  for (int i = wave; i < K; i += 4) {
    const double v = zz[i] / ((d[i] - oj) - tj);
    nrm += v * v;
  }
  part[wave][lane] = nrm;
  __syncthreads();
  const double sc = 1.0 / sqrt((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
  if (!act) return;
  double* dst = Sc + (compact ? (size_t)M.off * ldsc : 0) + (j - j0);
  for (int i = wave; i < K; i += 4) dst[(size_t)i * ldsc] = zz[i] / ((d[i] - oj) - tj) * sc;
}

// row blocks of Q -> column blocks of the sorted eigenvector matrix (all-to-all): rank q packs, for every destination p,
// its rows [r0, r0 + nr) of the eigenvector columns that p owns, sorted order: send[p][c * rp + r]
__global__ void pack_q_for_cols_kernel(const int* __restrict__ perm, const double* __restrict__ Q, int ldq, int r0, int nr,
                                       int rp, int zc, int nvec, double* __restrict__ send) {
  const int c = blockIdx.y, p = blockIdx.z;
  const int g = p * zc + c;       // global (sorted) eigenvector index
  if (c >= zc) return;
  double* dst = send + ((size_t)p * zc + c) * rp;
  if (g >= nvec) { for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rp; r += gridDim.x * blockDim.x) dst[r] = 0.0; return; }
  const double* src = Q + (size_t)perm[g] * ldq + r0;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rp; r += gridDim.x * blockDim.x) dst[r] = (r < nr) ? src[r] : 0.0;
}
// Z(q * rp + r, c) = recv[q][c * rp + r]
__global__ void unpack_cols_kernel(const double* __restrict__ recv, int rp, int zc, int n, double* __restrict__ Z, int ldz) {
  const int c = blockIdx.y, q = blockIdx.z;
  const double* src = recv + ((size_t)q * zc + c) * rp;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rp; r += gridDim.x * blockDim.x)
    if (q * rp + r < n) Z[(size_t)c * ldz + q * rp + r] = __hip_atomic_load(src + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void copycols_kernel(const int* __restrict__ src, const int* __restrict__ dst,
                                const int* __restrict__ row0, const int* __restrict__ nrows,
                                const double* __restrict__ Qa, double* __restrict__ Qb, int ldq) {
  const int p = blockIdx.x;
  const double* s = Qa + (size_t)src[p] * ldq + row0[p];
  double* t = Qb + (size_t)dst[p] * ldq + row0[p];
  for (int r = threadIdx.x; r < nrows[p]; r += blockDim.x) t[r] = s[r];
}

// final: z(:, p) = Q(:, perm[p]), w[p] = scale * D[perm[p]]
__global__ void final_permute_kernel(const int* __restrict__ perm, const double* __restrict__ Q, int ldq, int n,
                                     double* __restrict__ Z, int ldz, int nvec) {
  const int p = blockIdx.y;
  if (p >= nvec) return;
  const double* s = Q + (size_t)perm[p] * ldq;
  double* t = Z + (size_t)p * ldz;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) t[r] = s[r];
}

// ================================================================================================
// host side
// ================================================================================================
struct Node {
  int off, n;
  int left = -1, right = -1;  // children (node indices), -1 for a leaf
  int parent = -1;
  int height = 0;
  int n1 = 0;
  double sig[2] = {0, 0};
  double wv[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
};

// SVD of the band x band upper-triangular-in-(r<=c) coupling block, C = sum sig_k x_k y_k^T
// (role of DGESVD at src/my_pdlaed0.F:226,353)
void svd2(int band, const double* c, double* sig, double* x, double* y) {
  if (band == 1) { sig[0] = fabs(c[0]); x[0] = (c[0] >= 0) ? 1.0 : -1.0; y[0] = 1.0; return; }
  const double c00 = c[0], c10 = c[1], c01 = c[2], c11 = c[3];
  const double g00 = c00 * c00 + c10 * c10, g01 = c00 * c01 + c10 * c11, g11 = c01 * c01 + c11 * c11;
  double cs = 1.0, sn = 0.0;
  if (g01 != 0.0) {
    const double theta = (g11 - g00) / (2.0 * g01);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    cs = 1.0 / sqrt(t * t + 1.0);
    sn = t * cs;
  }
  const double ys[2][2] = {{cs, -sn}, {sn, cs}};
  for (int k = 0; k < 2; ++k) {
    const double u0 = c00 * ys[k][0] + c01 * ys[k][1], u1 = c10 * ys[k][0] + c11 * ys[k][1];
    const double s = hypot(u0, u1);
    sig[k] = s;
    y[2 * k] = ys[k][0]; y[2 * k + 1] = ys[k][1];
    if (s > 0.0) { x[2 * k] = u0 / s; x[2 * k + 1] = u1 / s; } else { x[2 * k] = 0.0; x[2 * k + 1] = 0.0; }
  }
}

struct HostDC {
  int n, band, lde;
  std::vector<double> d, e;  // host copies, torn in place
  std::vector<Node> nodes;
  double& E(int i, int b) { return e[(size_t)(b - 1) * lde + i]; }

  int build(int off, int nn) {
    const int id = (int)nodes.size();
    nodes.push_back(Node());
    nodes[id].off = off; nodes[id].n = nn;
    if (nn <= LEAF) return id;
    const int n1 = nn / 2;
    nodes[id].n1 = n1;
    double c[4] = {0, 0, 0, 0}, sig[2] = {0, 0}, x[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0};
    for (int r = 0; r < band; ++r)
      for (int cc = r; cc < band; ++cc) {
        const int gi = off + n1 + r, gj = off + n1 - band + cc;
        c[r + band * cc] = E(gi, gi - gj);
      }
    svd2(band, c, sig, x, y);
    for (int k = 0; k < band; ++k) {
      for (int r = 0; r < band; ++r)
        for (int cc = 0; cc < band; ++cc) {
          const int i1 = off + n1 - band + r, j1 = off + n1 - band + cc;
          const double v1 = sig[k] * y[band * k + r] * y[band * k + cc];
          if (i1 == j1) d[i1] -= v1; else if (i1 < j1) E(j1, j1 - i1) -= v1;
          const int i2 = off + n1 + r, j2 = off + n1 + cc;
          const double v2 = sig[k] * x[band * k + r] * x[band * k + cc];
          if (i2 == j2) d[i2] -= v2; else if (i2 < j2) E(j2, j2 - i2) -= v2;
        }
      nodes[id].sig[k] = sig[k];
      for (int r = 0; r < band; ++r) {
        nodes[id].wv[k][r] = y[band * k + r];
        nodes[id].wv[k][band + r] = x[band * k + r];
      }
    }
    const int l = build(off, n1);
    const int r = build(off + n1, nn - n1);
    nodes[id].left = l; nodes[id].right = r;
    nodes[l].parent = id; nodes[r].parent = id;
    nodes[id].height = 1 + std::max(nodes[l].height, nodes[r].height);
    return id;
  }
};

// Index order of d[0..nm) by (value, index).  The eigenvalues of a merge arrive as a handful of ascending runs (each
// child: its roots in order, then its deflated values, nearly in order), so this is a natural merge sort: find the runs,
// merge them pairwise; many runs (heavy deflation with reordering) fall back to std::sort.  Same order as sorting the
// (value, index) pairs, which is what every rank of a process grid must agree on.
void sort_index_runs(const double* d, int nm, std::vector<int>& idx, std::vector<int>& tmp, std::vector<int>& bounds) {
  idx.resize(nm);
  for (int i = 0; i < nm; ++i) idx[i] = i;
  bounds.clear();
  bounds.push_back(0);
  for (int i = 1; i < nm; ++i)
    if (d[i] < d[i - 1]) bounds.push_back(i);
  bounds.push_back(nm);
  auto cmp = [d](int a, int b) { return d[a] < d[b] || (d[a] == d[b] && a < b); };
  int nr = (int)bounds.size() - 1;
  if (nr > 16) { std::sort(idx.begin(), idx.end(), cmp); return; }
  tmp.resize(nm);
  while (nr > 1) {
    int w = 0;
    for (int r = 0; r < nr; r += 2) {
      if (r + 1 < nr)
        std::merge(idx.begin() + bounds[r], idx.begin() + bounds[r + 1], idx.begin() + bounds[r + 1], idx.begin() + bounds[r + 2],
                   tmp.begin() + bounds[r], cmp);
      else
        std::copy(idx.begin() + bounds[r], idx.begin() + bounds[r + 1], tmp.begin() + bounds[r]);
      bounds[w++] = bounds[r];
    }
    bounds[w] = nm;
    nr = w;
    idx.swap(tmp);
  }
}

}  // namespace

// several GPUs: width of the chunk buffer in which eigenvector rows are regenerated (eigx_tune key 8; the tests lower it so
// that the chunk-by-chunk path of the big merges runs at small sizes)
// lab switches (eigx_tune keys 15, 16): pipelined passes / one product launch per low height on one GPU
int g_dc_pipe = 1, g_dc_batch = 1, g_dc_side_min = 1024;
int set_dc_pipe(int v) {   // 0 / 1: off / on; v >= 2: merges larger than v use the side stream
  const int old = g_dc_pipe;
  if (v >= 2) g_dc_side_min = v; else g_dc_pipe = v ? 1 : 0;
  return old;
}
int set_dc_batch(int v) { const int old = g_dc_batch; g_dc_batch = v ? 1 : 0; return old; }
int g_dc_chunk = 2048;
int set_dc_chunk(int v) {
  const int old = g_dc_chunk;
  if (v >= 64 && v <= 2048 && v % 64 == 0) g_dc_chunk = v;
  return old;
}

// The two Q buffers must be zero outside the diagonal blocks before the leaves are written: 2 n^2 doubles of memset
// that depend on nothing.  The solver calls this before the reduction; the fills run on the side stream underneath it.
void band_dc_prepare(Context& ctx, int n) {
  const int P = ctx.grid.nranks;
  const int ldq = (P > 1) ? pad_ld((n + P - 1) / P + 2) : pad_ld(n);   // several GPUs: a rank keeps ceil(n/P) rows of Q
  double* Qa = ctx.pool.get_t<double>("dc.Qa", (size_t)ldq * n);
  double* Qb = ctx.pool.get_t<double>("dc.Qb", (size_t)ldq * n);
  EIGX_HIP_CHECK(hipMemsetAsync(Qa, 0, (size_t)ldq * n * 8, ctx.side_stream));
  EIGX_HIP_CHECK(hipMemsetAsync(Qb, 0, (size_t)ldq * n * 8, ctx.side_stream));
  EIGX_HIP_CHECK(hipEventRecord(ctx.dc_ev, ctx.side_stream));
  ctx.dc_zero_n = n; ctx.dc_zero_qa = Qa; ctx.dc_zero_qb = Qb;
}

void band_dc_dev(Context& ctx, int n, int nvec, const double* d_dev, const double* e_dev, int lde, int band,
                 double* w_dev, double* z_dev, int ldz) {
  hipStream_t st = ctx.stream;
  const double eps = DBL_EPSILON / 2.0;
  // Several GPUs (replaces the process tree / ring GEMM of dc2_FS, src/FS_PDLAED0.F90:62-323, src/FS_PDLAED3.F90:526-860):
  //   * Q is row-distributed in contiguous blocks of rp = ceil(n/P) rows and only those rows are ALLOCATED
  //     (buffers of rp x n; the kernels index global rows through a shifted base pointer);
  //   * the secular equation is split by root index (rank r: roots [K r/P, K (r+1)/P) of every merge), the roots
  //     travel as (lambda, origin, tau) triples in one small allreduce; no K x K matrix is stored or sent:
  //     eigenvector rows are regenerated from the triples, a chunk of roots at a time, right before the GEMM
  //     that consumes them;
  //   * the O(n K^2) GEMMs touch the rank's rows only; z = Q^T w needs one allreduce of n doubles per pass;
  //   * deflation runs on every host on bit-identical inputs (O(n) per height);
  //   * at the end an all-to-all turns row blocks of Q into column blocks of the sorted eigenvector matrix:
  //     z_dev receives columns [rank*zc, rank*zc + zc), zc = ceil(nvec/P), all n rows (what the column-parallel
  //     back-transformation works on).
  const int P = ctx.grid.nranks;
  const int rp = (n + P - 1) / P;
  const int r0 = P > 1 ? std::min(n, ctx.grid.rank * rp) : 0;
  const int r1 = P > 1 ? std::min(n, r0 + rp) : n;
  auto clip = [&](int lo, int hi, int& a, int& b) { a = std::max(lo, r0); b = std::min(hi, r1); return b > a; };
  HostDC H;
  H.n = n; H.band = band; H.lde = lde;
  H.d.resize(n);
  H.e.resize((size_t)lde * band);
  EIGX_HIP_CHECK(hipMemcpyAsync(H.d.data(), d_dev, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  EIGX_HIP_CHECK(hipMemcpyAsync(H.e.data(), e_dev, (size_t)lde * band * 8, hipMemcpyDeviceToHost, st));
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  // scale to unit max-norm (MY_PDSxEDC scales by the DLANST norm, src/my_pdsxedc.F:277-290)
  double nrm = 0.0;
  for (int i = 0; i < n; ++i) {
    nrm = std::max(nrm, fabs(H.d[i]));
    for (int b = 1; b <= band; ++b) if (i >= b) nrm = std::max(nrm, fabs(H.E(i, b)));
  }
  const double scl = nrm > 0.0 ? 1.0 / nrm : 1.0;
  for (int i = 0; i < n; ++i) {
    H.d[i] *= scl;
    for (int b = 1; b <= band; ++b) H.E(i, b) = (i >= b) ? H.E(i, b) * scl : 0.0;
  }
  H.nodes.reserve(2 * (n / (LEAF / 2) + 2));
  const int root = H.build(0, n);
  const int maxh = H.nodes[root].height;

  // ---- device workspace --------------------------------------------------------------------------
  // One GPU: the passes are pipelined (see znext1_kernel) -- two arenas and two S buffers alternate from pass to pass.
  const bool pipe = (P == 1) && g_dc_pipe;
  const int ldq = (P > 1) ? pad_ld(rp + 2) : pad_ld(n);
  double* Qa_base = ctx.pool.get_t<double>("dc.Qa", (size_t)ldq * n);
  double* Qb_base = ctx.pool.get_t<double>("dc.Qb", (size_t)ldq * n);
  // several GPUs: element (row, col) of the rank's block lives at base[(row - r0) + col*ldq]; the shifted pointers
  // let every kernel keep global row indices (all of them clip to [r0, r1))
  double* Qa = Qa_base - r0;
  double* Qb = Qb_base - r0;
  // eigenvector rows of the rank-one updates.  One GPU: block diagonal K x K blocks in an n x n array.  Several GPUs:
  // a chunk buffer of n x SCW doubles (all merges of a low height side by side, or one chunk of roots of a big merge)
  const int SCW = g_dc_chunk;
  const int lds_mg = SCW;
  double* Sbuf[2];
  Sbuf[0] = ctx.pool.get_t<double>("dc.S", (P > 1) ? (size_t)n * SCW + 64 : (size_t)ldq * n);
  Sbuf[1] = pipe ? ctx.pool.get_t<double>("dc.S2", (size_t)ldq * n) : Sbuf[0];
  double* sec = (P > 1) ? ctx.pool.get_t<double>("dc.sec", (size_t)3 * n) : nullptr;
  double* dd = ctx.pool.get_t<double>("dc.d", (size_t)n);
  double* de = ctx.pool.get_t<double>("dc.e", (size_t)lde * band);
  double* zh = ctx.pool.get_t<double>("dc.zh", (size_t)n);
  double* vnp = ctx.pool.get_t<double>("dc.vnp", (size_t)n * ((n + VEC_IC - 1) / VEC_IC + 1));  // partial column norms
  double* zp = pipe ? ctx.pool.get_t<double>("dc.zp", (size_t)n * ((n + ZN_IC - 1) / ZN_IC + 1)) : nullptr;
  const int maxmerge = n / (LEAF / 2) + 8;
  int* leafinfo = ctx.pool.get_t<int>("dc.leaf", (size_t)2 * maxmerge);
  int* perm_dev = ctx.pool.get_t<int>("dc.perm", (size_t)n);
  int* iota_dev = ctx.pool.get_t<int>("dc.iota", (size_t)n);  // 0,1,2,...: identity map for the pole index of dense updates
  // One arena holds everything the host exchanges with the device per merge step, mirrored in pinned host
  // memory: a step is one D2H copy (Dcur | z), the host deflation, and one H2D copy (everything up to Dcur).
  //   merge table | product table | ints: nd rpj rjj cps cpd cpr cpn topA topB botA botB dsrc | doubles: dlam wz rc rs Dcur zbuf
  // Arena p&1 belongs to pass p: its Dcur receives the pass's eigenvalues and its zbuf the z of pass p + 1.
  struct Arena {
    char *dev, *host;
    MergeDev *md_dev, *md_h;
    GemmBatch *gb_dev, *gb_h;
    int *nd_dev, *nd_h, *rpj_dev, *rpj_h, *rjj_dev, *rjj_h, *cps_dev, *cps_h, *cpd_dev, *cpd_h, *cpr_dev, *cpr_h, *cpn_dev, *cpn_h;
    int *topA_dev, *topA_h, *topB_dev, *topB_h, *botA_dev, *botA_h, *botB_dev, *botB_h, *dsrc_dev, *dsrc_h;
    double *dlam, *dl_h, *wz, *wz_h, *rc_dev, *rc_h, *rs_dev, *rs_h, *Dcur, *Dh, *zbuf, *zhost;
  } ar[2];
  const size_t npad = ((size_t)n + 7) / 8 * 8;
  const size_t md_bytes = ((size_t)maxmerge * sizeof(MergeDev) + 63) / 64 * 64;
  const size_t gb_bytes = ((size_t)2 * maxmerge * sizeof(GemmBatch) + 63) / 64 * 64;
  const size_t up_bytes = md_bytes + gb_bytes + 12 * npad * sizeof(int) + 5 * npad * sizeof(double);   // H2D part
  const size_t arena_bytes = up_bytes + npad * sizeof(double);                                         // + zbuf
  const size_t down_off = up_bytes - npad * 8;   // byte offset of Dcur: the D2H copy takes [Dcur | zbuf]
  {
    char* dev_all = (char*)ctx.pool.get("dc.arena", 2 * arena_bytes);
    char* host_all = (char*)ctx.pool.get_host("dc.arena", 2 * arena_bytes);
    memset(host_all, 0, 2 * arena_bytes);
    for (int q = 0; q < 2; ++q) {
      Arena& A = ar[q];
      A.dev = dev_all + q * arena_bytes;
      A.host = host_all + q * arena_bytes;
      size_t pos = 0;
      auto carve = [&](size_t bytes, void* pd, void* ph) {
        *(char**)pd = A.dev + pos; *(char**)ph = A.host + pos; pos += bytes;
      };
      carve(md_bytes, &A.md_dev, &A.md_h);
      carve(gb_bytes, &A.gb_dev, &A.gb_h);
      carve(npad * 4, &A.nd_dev, &A.nd_h);      // [n] non-deflated column (global) per merge range, dlam order
      carve(npad * 4, &A.rpj_dev, &A.rpj_h);  carve(npad * 4, &A.rjj_dev, &A.rjj_h);
      carve(npad * 4, &A.cps_dev, &A.cps_h);  carve(npad * 4, &A.cpd_dev, &A.cpd_h);   // copy src / dst / row0 / nrows
      carve(npad * 4, &A.cpr_dev, &A.cpr_h);  carve(npad * 4, &A.cpn_dev, &A.cpn_h);
      carve(npad * 4, &A.topA_dev, &A.topA_h); carve(npad * 4, &A.topB_dev, &A.topB_h);   // first update: columns with a non-zero top part (global index), their pole indices
      carve(npad * 4, &A.botA_dev, &A.botA_h); carve(npad * 4, &A.botB_dev, &A.botB_h);   // same for the bottom part
      carve(npad * 4, &A.dsrc_dev, &A.dsrc_h);   // source column of every deflated output slot
      carve(npad * 8, &A.dlam, &A.dl_h);  carve(npad * 8, &A.wz, &A.wz_h);
      carve(npad * 8, &A.rc_dev, &A.rc_h); carve(npad * 8, &A.rs_dev, &A.rs_h);
      carve(npad * 8, &A.Dcur, &A.Dh);    carve(npad * 8, &A.zbuf, &A.zhost);
    }
  }
  // the merge descriptors also go up once BEFORE the deflation when z is gathered from Q (zgather needs offsets): own
  // staging buffer
  MergeDev* md_h2 = (MergeDev*)ctx.pool.get_host("dc.md2", md_bytes);

  EIGX_HIP_CHECK(hipMemcpyAsync(dd, H.d.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
  EIGX_HIP_CHECK(hipMemcpyAsync(de, H.e.data(), (size_t)lde * band * 8, hipMemcpyHostToDevice, st));
  if (ctx.dc_zero_n == n && ctx.dc_zero_qa == Qa_base && ctx.dc_zero_qb == Qb_base) {
    // zero-filled ahead on the side stream while the reduction ran (band_dc_prepare)
    EIGX_HIP_CHECK(hipStreamWaitEvent(st, ctx.dc_ev, 0));
  } else {
    EIGX_HIP_CHECK(hipMemsetAsync(Qa_base, 0, (size_t)ldq * n * 8, st));
    EIGX_HIP_CHECK(hipMemsetAsync(Qb_base, 0, (size_t)ldq * n * 8, st));
  }
  ctx.dc_zero_n = 0;
  {
    std::vector<int> iota(n);
    for (int q = 0; q < n; ++q) iota[q] = q;
    EIGX_HIP_CHECK(hipMemcpyAsync(iota_dev, iota.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
  }

  // ---- leaves ---------------------------------------------------------------------------------------
  double* Dfinal = ar[1].Dcur;   // pass 0 reads the leaves' eigenvalues from the arena "before" its own
  {
    std::vector<int> lo, ln;
    for (const Node& nd : H.nodes)
      if (nd.left < 0) { lo.push_back(nd.off); ln.push_back(nd.n); }
    const int nl = (int)lo.size();
    std::vector<int> both(lo);
    both.insert(both.end(), ln.begin(), ln.end());
    EIGX_HIP_CHECK(hipMemcpyAsync(leafinfo, both.data(), (size_t)2 * nl * 4, hipMemcpyHostToDevice, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));  // `both` is a stack vector
    hipLaunchKernelGGL(jacobi_leaf_kernel, dim3(nl), dim3(LEAF_T), 0, st, dd, de, lde, band, leafinfo, leafinfo + nl,
                       Dfinal, Qa, ldq, r0, r1);
  }
  // The solver's side work (T factors of the back-transformation: ~2.4 ms of latency-bound kernels on a few dozen CUs) is
  // enqueued when the LAST pass's product starts: beside the leaves or the low heights it made their small dependent
  // kernels wait behind its long workgroups (leaf kernel 0.45 -> 1.4 ms, or a secular launch 0.03 -> 0.7 ms); beside a
  // chip-filling product it costs its own CU time and nothing else.
  auto run_side_work = [&] {
    if (!ctx.dc_side_work) return;
    std::function<void()> f = std::move(ctx.dc_side_work);
    ctx.dc_side_work = nullptr;
    f();
  };

  // ---- merges: the passes (height, k) in order --------------------------------------------------------
  struct Pass { int h, k; std::vector<int> ids; };
  std::vector<Pass> passes;
  for (int h = 1; h <= maxh; ++h) {
    std::vector<int> ids;
    for (int id = 0; id < (int)H.nodes.size(); ++id)
      if (H.nodes[id].left >= 0 && H.nodes[id].height == h) ids.push_back(id);
    if (ids.empty()) continue;
    for (int k = 0; k < band; ++k) passes.push_back(Pass{h, k, ids});
  }
  std::vector<double> Dold(n);
  std::vector<int> ctype(n), ktop(n / 2 + 8), kbot(n / 2 + 8), defl;
  std::vector<MergeDev> mds;
  std::vector<std::pair<double, int>> ord;
  std::vector<int> ordi, ordt, ordb;
  double gemm_flops = 0.0;
  const bool trace = getenv("EIGX_TRACE_DC") != nullptr;
  auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // stream of a pass's secular / eigenvector-row kernels: the high-priority side stream where there is a product to run
  // under (big merges); the low heights are one dependent chain either way, and a second stream only adds event hand-offs
  bool z_ready = false;                         // [Dcur | z] of this pass are already on their way to the host
  double* early_final = nullptr;                // host copy of the final eigenvalues, requested inside the last pass
  for (size_t pi = 0; pi < passes.size(); ++pi) {
    const std::vector<int>& ids = passes[pi].ids;
    const int h = passes[pi].h, k = passes[pi].k;
    Arena& C = ar[pi & 1];          // this pass
    Arena& V = ar[(pi & 1) ^ 1];    // the previous one: its Dcur / zbuf are this pass's input
    double* const S = Sbuf[pi & 1];
    {
      mds.assign(ids.size(), MergeDev());
      int maxnm = 0;
      size_t covered = 0;
      for (size_t q = 0; q < ids.size(); ++q) {
        const Node& nd = H.nodes[ids[q]];
        MergeDev& M = mds[q];
        M.off = nd.off; M.nm = nd.n; M.n1 = nd.n1; M.K = 0; M.rot_beg = 0; M.rot_end = 0; M.rho = nd.sig[k];
        for (int t = 0; t < 4; ++t) M.wv[t] = nd.wv[k][t];
        M.zr_n = 0; M.zr_pad = 0;
        for (int t = 0; t < 4; ++t) { M.zr_row[t] = 0; M.zr_w[t] = 0.0; }
        maxnm = std::max(maxnm, nd.n);
        covered += (size_t)nd.n;
      }
      // rows of the next pass's z inside each merge (only when this pass rewrites every column: balanced heights)
      const bool z_ahead = pipe && covered == (size_t)n && pi + 1 < passes.size();
      if (z_ahead) {
        for (size_t q = 0; q < ids.size(); ++q) {
          const Node& nd = H.nodes[ids[q]];
          MergeDev& M = mds[q];
          if (k + 1 < band) {           // same merges, next singular triplet
            M.zr_n = 2 * band;
            for (int t = 0; t < 2 * band; ++t) { M.zr_row[t] = nd.off + nd.n1 - band + t; M.zr_w[t] = nd.wv[k + 1][t]; }
          } else {                      // the parents' first triplet: `band` rows on this child's side of the tear
            const Node& pn = H.nodes[nd.parent];
            const bool is_left = (pn.left == ids[q]);
            M.zr_n = band;
            for (int t = 0; t < band; ++t) {
              M.zr_row[t] = is_left ? pn.off + pn.n1 - band + t : pn.off + pn.n1 + t;
              M.zr_w[t] = pn.wv[0][is_left ? t : band + t];
            }
          }
        }
      }
      // -- z = Q^T w ------------------------------------------------------------------------------------
      const double tt0 = trace ? now_s() : 0.0;
      if (!z_ready) {
        memcpy(md_h2, mds.data(), mds.size() * sizeof(MergeDev));
        EIGX_HIP_CHECK(hipMemcpyAsync(C.md_dev, md_h2, mds.size() * sizeof(MergeDev), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(zgather_kernel, dim3((maxnm + 255) / 256, (unsigned)ids.size()), dim3(256), 0, st, C.md_dev,
                           band, Qa, ldq, V.zbuf, r0, r1);
        if (P > 1) comm_allreduce_sum(ctx, COMM_WORLD, V.zbuf, (size_t)n, st);
        EIGX_HIP_CHECK(hipMemcpyAsync(V.host + down_off, V.dev + down_off, 2 * npad * 8, hipMemcpyDeviceToHost, st));  // Dcur | z
        EIGX_HIP_CHECK(hipStreamSynchronize(st));
      } else {
        EIGX_HIP_CHECK(hipEventSynchronize(ctx.dc_z_ev));
      }
      z_ready = false;
      const double tt1 = trace ? now_s() : 0.0;
      // -- host deflation -------------------------------------------------------------------------------
      int nrot = 0, ncopy = 0;
      std::copy(V.Dh, V.Dh + n, Dold.begin());
      std::copy(V.Dh, V.Dh + n, C.Dh);       // columns outside this pass's merges keep their eigenvalue
      double* const zhost = V.zhost;
      for (size_t q = 0; q < ids.size(); ++q) {
        MergeDev& M = mds[q];
        const int off = M.off, nm = M.nm;
        double* dloc = &Dold[off];
        double* zloc = &zhost[off];
        double zn = 0.0;
        for (int i = 0; i < nm; ++i) zn += zloc[i] * zloc[i];
        zn = sqrt(zn);
        double rho = M.rho;
        M.rot_beg = nrot;
        int K = 0;
        // column types for the first update: 1 = non-zero only in block 1's rows, 2 = only block 2's, 3 = both
        for (int i2 = 0; i2 < nm; ++i2) ctype[off + i2] = (k == 0) ? (i2 < M.n1 ? 1 : 2) : 3;
        // output slots: roots first, then deflated columns
        defl.clear();
        if (zn > 0.0 && rho > 0.0) {
          for (int i = 0; i < nm; ++i) zloc[i] /= zn;
          rho *= zn * zn;
          double dmax = 0.0, zmax = 0.0;
          for (int i = 0; i < nm; ++i) {
            dmax = std::max(dmax, fabs(dloc[i]));
            zmax = std::max(zmax, fabs(zloc[i]));
          }
          sort_index_runs(dloc, nm, ordi, ordt, ordb);
          const double tol = 8.0 * eps * std::max(dmax, zmax);
          if (rho * zmax > tol) {
            int pj = -1;
            for (int t = 0; t < nm; ++t) {
              const int jj = ordi[t];
              if (rho * fabs(zloc[jj]) <= tol) { defl.push_back(jj); continue; }
              if (pj < 0) { pj = jj; continue; }
              double s = zloc[pj], c = zloc[jj];
              const double tt = dloc[jj] - dloc[pj];
              // |tt c s| = |tt| |z_p z_j| / (z_p^2 + z_j^2): far from the threshold (the usual case) no hypot / divisions are
              // needed to see that this pair does not deflate (they were most of the host time of a low height)
              if (fabs(tt * c * s) > 2.0 * tol * (c * c + s * s)) {
                C.nd_h[off + K] = off + pj;
                ++K;
                pj = jj;
                continue;
              }
              const double tau = hypot(c, s);
              c /= tau; s = -s / tau;
              if (fabs(tt * c * s) <= tol) {
                zloc[jj] = tau; zloc[pj] = 0.0;
                C.rpj_h[nrot] = off + pj; C.rjj_h[nrot] = off + jj; C.rc_h[nrot] = c; C.rs_h[nrot] = s;
                ++nrot;
                if (ctype[off + pj] != ctype[off + jj]) { ctype[off + pj] = 3; ctype[off + jj] = 3; }
                const double dp = dloc[pj] * c * c + dloc[jj] * s * s;
                dloc[jj] = dloc[pj] * s * s + dloc[jj] * c * c;
                dloc[pj] = dp;
                defl.push_back(pj);
                pj = jj;
              } else {
                C.nd_h[off + K] = off + pj;
                ++K;
                pj = jj;
              }
            }
            if (pj >= 0) { C.nd_h[off + K] = off + pj; ++K; }
          } else {
            for (int i = 0; i < nm; ++i) defl.push_back(i);
          }
        } else {
          for (int i = 0; i < nm; ++i) defl.push_back(i);
        }
        M.rot_end = nrot;
        // poles in strictly ascending order
        for (int t = 0; t < K; ++t) { C.dl_h[off + t] = dloc[C.nd_h[off + t] - off]; C.wz_h[off + t] = zloc[C.nd_h[off + t] - off]; }
        for (int t = 1; t < K; ++t) {
          int u = t;
          while (u > 0 && C.dl_h[off + u] < C.dl_h[off + u - 1]) {
            std::swap(C.dl_h[off + u], C.dl_h[off + u - 1]);
            std::swap(C.wz_h[off + u], C.wz_h[off + u - 1]);
            std::swap(C.nd_h[off + u], C.nd_h[off + u - 1]);
            --u;
          }
        }
        M.K = K;
        M.rho = rho;
        {
          int nt_ = 0, nb_ = 0;
          for (int t = 0; t < K; ++t) {
            const int col = C.nd_h[off + t], ty = ctype[col];
            if (ty != 2) { C.topA_h[off + nt_] = col; C.topB_h[off + nt_] = t; ++nt_; }
            if (ty != 1) { C.botA_h[off + nb_] = col; C.botB_h[off + nb_] = t; ++nb_; }
          }
          ktop[q] = nt_; kbot[q] = nb_;
        }
        // deflated columns go behind the K roots
        for (size_t t = 0; t < defl.size(); ++t) {
          const int src = off + defl[t], dst = off + K + (int)t;
          int ca, cb;
          if (clip(off, off + nm, ca, cb)) {
            C.cps_h[ncopy] = src; C.cpd_h[ncopy] = dst; C.cpr_h[ncopy] = ca; C.cpn_h[ncopy] = cb - ca;
            ++ncopy;
          }
          C.dsrc_h[dst] = src;
          C.Dh[dst] = dloc[defl[t]];
        }
        for (int t = 0; t < K; ++t) C.Dh[off + t] = 0.0;  // overwritten by the secular kernel
      }
      int maxK = 0;
      for (const MergeDev& M : mds) maxK = std::max(maxK, M.K);
      const unsigned nmg = (unsigned)ids.size();
      hipStream_t sb = (pipe && maxnm > g_dc_side_min) ? ctx.dc_stream : st;
      // one launch for all products of the pass where each of them is a few 64 x 64 tiles (one GPU, low heights)
      const bool batched = (P == 1) && g_dc_batch && maxK > 0 && mds.size() > 1 &&
                           (long)ceil_div(maxnm, 128) * ceil_div(maxK, 128) < 192;
      int nbatch = 0, bmaxM = 0;
      if (batched) {
        // product table (part of the arena upload).  Gather maps: first update = topA/topB for the top rows, botA/botB
        // (2 npad ints further in the arena) for the bottom rows; second update = nd and the identity
        const int bot_shift = (int)(C.botA_dev - C.topA_dev);
        for (size_t q = 0; q < mds.size(); ++q) {
          const MergeDev& M = mds[q];
          if (M.K <= 0) continue;
          auto add = [&](int row0, int rows, int kk, int offKA, int offKB) {
            if (rows <= 0) return;
            GemmBatch& B = C.gb_h[nbatch++];
            B.M = rows; B.N = M.K; B.K = kk; B.pad = 0;
            B.offA = row0; B.offB = (long)M.off * ldq + M.off; B.offC = (long)M.off * ldq + row0;
            B.offKA = offKA; B.offKB = offKB;
            bmaxM = std::max(bmaxM, rows);
          };
          if (k == 0) {
            add(M.off, M.n1, ktop[q], M.off, M.off);
            add(M.off + M.n1, M.nm - M.n1, kbot[q], M.off + bot_shift, M.off + bot_shift);
            gemm_flops += 2.0 * (double)M.K * ((double)M.n1 * ktop[q] + (double)(M.nm - M.n1) * kbot[q]);
          } else {
            add(M.off, M.nm, M.K, M.off, 0);
            gemm_flops += 2.0 * M.nm * (double)M.K * M.K;
          }
        }
      }
      // -- upload and run the GPU part ---------------------------------------------------------------------
      const double tt2 = trace ? now_s() : 0.0;
      double t_gemm_enq = 0.0;
      memcpy(C.md_h, mds.data(), mds.size() * sizeof(MergeDev));
      EIGX_HIP_CHECK(hipMemcpyAsync(C.dev, C.host, up_bytes, hipMemcpyHostToDevice, sb));   // everything at once
      // several GPUs: chunked eigenvector rows.  compact = all merges of this height fit side by side into S
      // (lds = SCW >= merge size); otherwise the (one or two) big merges go chunk by chunk below.
      const bool mg = P > 1;
      const bool compact = mg && maxnm <= lds_mg;
      if (maxK > 0) {
        if (mg) {
          const int kmx = maxK / P + 2;   // roots per rank and merge
          EIGX_HIP_CHECK(hipMemsetAsync(sec, 0, (size_t)3 * n * 8, st));
          if (maxK >= 512)
            hipLaunchKernelGGL(secular_kernel<32>, dim3((kmx + 7) / 8, nmg), dim3(256), 0, st, C.md_dev, C.dlam, C.wz, C.Dcur, S, ldq, P,
                               ctx.grid.rank, sec, n);
          else
            hipLaunchKernelGGL(secular_kernel<8>, dim3((kmx + 31) / 32, nmg), dim3(256), 0, st, C.md_dev, C.dlam, C.wz, C.Dcur, S, ldq, P,
                               ctx.grid.rank, sec, n);
          comm_allreduce_sum(ctx, COMM_WORLD, sec, (size_t)3 * n, st);
          hipLaunchKernelGGL(scatter_lambda_kernel, dim3((maxK + 255) / 256, nmg), dim3(256), 0, st, C.md_dev, sec, C.Dcur);
          hipLaunchKernelGGL(loewner_mg_kernel, dim3((maxK + 3) / 4, nmg), dim3(256), 0, st, C.md_dev, C.dlam, C.wz, sec, n, zh);
          if (compact)
            hipLaunchKernelGGL(vectors_mg_kernel, dim3((maxK + 63) / 64, 1, nmg), dim3(256), 0, st, C.md_dev, 0, C.dlam, zh, sec, n, 0,
                               maxK, 1, S, lds_mg);
        } else {
          if (maxK >= 512)
            hipLaunchKernelGGL(secular_kernel<32>, dim3((maxK + 7) / 8, nmg), dim3(256), 0, sb, C.md_dev, C.dlam, C.wz, C.Dcur, S, ldq, 1, 0,
                               (double*)nullptr, 0);
          else
            hipLaunchKernelGGL(secular_kernel<8>, dim3((maxK + 31) / 32, nmg), dim3(256), 0, sb, C.md_dev, C.dlam, C.wz, C.Dcur, S, ldq, 1, 0,
                               (double*)nullptr, 0);
          hipLaunchKernelGGL(loewner_kernel, dim3((maxK + 3) / 4, nmg), dim3(256), 0, sb, C.md_dev, C.dlam, C.wz, S, ldq, zh);
          const dim3 vg((maxK + 63) / 64, (maxK + VEC_IC - 1) / VEC_IC, nmg);
          hipLaunchKernelGGL(vectors1_kernel, vg, dim3(256), 0, sb, C.md_dev, zh, S, ldq, vnp, n);
          hipLaunchKernelGGL(vectors2_kernel, vg, dim3(256), 0, sb, C.md_dev, S, ldq, vnp, n);
        }
      }
      if (sb != st) {   // the compute stream takes over: rotations and products need the previous product AND this pass's rows
        EIGX_HIP_CHECK(hipEventRecord(ctx.dc_b_ev, sb));
        EIGX_HIP_CHECK(hipStreamWaitEvent(st, ctx.dc_b_ev, 0));
      }
      if (pipe && pi + 1 == passes.size()) {
        // the eigenvalues are final once the last pass's secular equations are solved: fetch them now and sort on the
        // host under the last product (the serial tail -- copy, sort, two uploads -- was 0.26 ms at N = 8192)
        EIGX_HIP_CHECK(hipMemcpyAsync(V.Dh, C.Dcur, (size_t)n * 8, hipMemcpyDeviceToHost, sb));
        EIGX_HIP_CHECK(hipEventRecord(ctx.dc_z_ev, sb));
        early_final = V.Dh;
      }
      if (nrot > 0)
        hipLaunchKernelGGL(rotate_kernel, dim3((maxnm + 255) / 256, nmg), dim3(256), 0, st, C.md_dev,
                           C.rpj_dev, C.rjj_dev, C.rc_dev, C.rs_dev, Qa, ldq, r0, r1);
      if (z_ahead) {
        if (maxK > 0)
          hipLaunchKernelGGL(znext1_kernel, dim3((maxK + 63) / 64, (maxK + ZN_IC - 1) / ZN_IC, nmg), dim3(256), 0, st, C.md_dev,
                             C.nd_dev, Qa, ldq, S, ldq, zp, n);
        hipLaunchKernelGGL(znext2_kernel, dim3((maxnm + 255) / 256, nmg), dim3(256), 0, st, C.md_dev, C.dsrc_dev, Qa, ldq, zp, n,
                           C.zbuf);
        EIGX_HIP_CHECK(hipMemcpyAsync(C.host + down_off, C.dev + down_off, 2 * npad * 8, hipMemcpyDeviceToHost, st));  // Dcur | z'
        EIGX_HIP_CHECK(hipEventRecord(ctx.dc_z_ev, st));
        z_ready = true;
      }
      if (pi + 1 == passes.size()) run_side_work();
      if (maxK > 0) {
        // the merges of one height are independent: when there are several, spread their GEMMs over the aux
        // streams so that small products run side by side instead of one after another
        // (not in chunked mode: there the chunk buffer S is reused from chunk to chunk and from merge to merge, and only
        // the order of ONE stream keeps a chunk's generation behind the GEMMs that still read the previous one)
        const bool fan = mds.size() > 1 && !(mg && !compact) && !batched;
        if (fan) {
          EIGX_HIP_CHECK(hipEventRecord(ctx.aux_ev[Context::kAux], st));
          for (int q = 0; q < Context::kAux; ++q) EIGX_HIP_CHECK(hipStreamWaitEvent(ctx.aux[q], ctx.aux_ev[Context::kAux], 0));
        }
        int rr = 0;
        const double tg0 = trace ? now_s() : 0.0;
        if (batched) {
          if (k == 0)
            dgemm_gather_batch_dev(st, C.gb_dev, nbatch, bmaxM, maxK, Qa, ldq, S, ldq, Qb, ldq, C.topA_dev, C.topB_dev);
          else
            dgemm_gather_batch_dev(st, C.gb_dev, nbatch, bmaxM, maxK, Qa, ldq, S, ldq, Qb, ldq, C.nd_dev, iota_dev);
        } else
        for (size_t q = 0; q < mds.size(); ++q) {
          const MergeDev& M = mds[q];
          if (M.K <= 0) continue;
          // root chunks [j0, j0 + cw): one chunk = all K roots, except for the big merges of several GPUs, whose
          // eigenvector rows are regenerated chunk by chunk into the n x SCW buffer right before their GEMM
          const bool chunked = mg && !compact;
          int cwq = M.K;
          if (chunked) { cwq = (int)(((size_t)n * SCW / (size_t)M.K) / 64 * 64); if (cwq > M.K) cwq = M.K; if (cwq < 64) cwq = 64; }
          for (int j0 = 0; j0 < M.K; j0 += cwq) {
          const int cw = (M.K - j0 < cwq) ? M.K - j0 : cwq;
          const int ldsb = mg ? (chunked ? ((cw + 1) & ~1) : lds_mg) : ldq;
          const double* Sb = mg ? (chunked ? S : S + (size_t)M.off * lds_mg) : S + (size_t)M.off * ldq + M.off;
          if (chunked) {
            // the chunk buffer is reused: order this chunk's generation after the previous chunk's GEMMs
            hipLaunchKernelGGL(vectors_mg_kernel, dim3((cw + 63) / 64, 1, 1), dim3(256), 0, st, C.md_dev, (int)q, C.dlam, zh, sec, n, j0,
                               j0 + cw, 0, S, ldsb);
          }
          double* Cb = Qb + (size_t)(M.off + j0) * ldq + M.off;
          // Qb(rows, off+j) = sum_i Qa(rows, nd[i]) * U(i,j),  U(i,j) = S'(j,i)
          if (k == 0) {
            // first update: Q = diag(Q1, Q2) up to the Givens-mixed columns, so the top rows only see the
            // columns of type 1/3 and the bottom rows those of type 2/3 (DLAED3's compressed Q2 idea)
            hipStream_t g1 = fan ? ctx.aux[rr++ % Context::kAux] : st;
            hipStream_t g2 = fan ? ctx.aux[rr++ % Context::kAux] : ((mds.size() == 1 && !chunked) ? ctx.aux[0] : st);
            if (!fan && g2 != st) {
              EIGX_HIP_CHECK(hipEventRecord(ctx.aux_ev[Context::kAux], st));
              EIGX_HIP_CHECK(hipStreamWaitEvent(g2, ctx.aux_ev[Context::kAux], 0));
            }
            int ga, gb;
            if (clip(M.off, M.off + M.n1, ga, gb))
              dgemm_dev(g1, 'N', 'T', gb - ga, cw, ktop[q], 1.0, Qa + ga, ldq, Sb, ldsb, 0.0, Cb + (ga - M.off), ldq, 0,
                        nullptr, C.topA_dev + M.off, nullptr, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, C.topB_dev + M.off);
            if (clip(M.off + M.n1, M.off + M.nm, ga, gb))
              dgemm_dev(g2, 'N', 'T', gb - ga, cw, kbot[q], 1.0, Qa + ga, ldq, Sb, ldsb, 0.0, Cb + (ga - M.off), ldq, 0,
                        nullptr, C.botA_dev + M.off, nullptr, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, C.botB_dev + M.off);
            if (!fan && g2 != st) {
              EIGX_HIP_CHECK(hipEventRecord(ctx.aux_ev[0], g2));
              EIGX_HIP_CHECK(hipStreamWaitEvent(st, ctx.aux_ev[0], 0));
            }
            gemm_flops += 2.0 * (double)cw * ((double)M.n1 * ktop[q] + (double)(M.nm - M.n1) * kbot[q]);
          } else {
            hipStream_t gs = fan ? ctx.aux[rr++ % Context::kAux] : st;
            int ga, gb;
            if (clip(M.off, M.off + M.nm, ga, gb))
              dgemm_dev(gs, 'N', 'T', gb - ga, cw, M.K, 1.0, Qa + ga, ldq, Sb, ldsb, 0.0, Cb + (ga - M.off), ldq, 0,
                        nullptr, C.nd_dev + M.off, nullptr, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, iota_dev);
            gemm_flops += 2.0 * M.nm * (double)M.K * cw;
          }
          }
        }
        if (fan)
          for (int q = 0; q < Context::kAux; ++q) {
            EIGX_HIP_CHECK(hipEventRecord(ctx.aux_ev[q], ctx.aux[q]));
            EIGX_HIP_CHECK(hipStreamWaitEvent(st, ctx.aux_ev[q], 0));
          }
        if (trace) t_gemm_enq = now_s() - tg0;
      }
      if (ncopy > 0)
        hipLaunchKernelGGL(copycols_kernel, dim3(ncopy), dim3(256), 0, st, C.cps_dev, C.cpd_dev, C.cpr_dev, C.cpn_dev, Qa, Qb,
                           ldq);
      // merged blocks go back into Qa (blocks that do not merge at this height stay where they are)
      // When the merges of this height cover every column (balanced tree: always), Qb now IS the new Q: swap the
      // buffers instead of copying n^2 doubles back.  Both buffers are zero outside the diagonal blocks (memset at
      // the start; GEMMs and column copies only ever write rows inside their own block).
      if (covered == (size_t)n) {
        std::swap(Qa, Qb);
      } else {
        for (const MergeDev& M : mds) {
          int ca, cb;
          if (!clip(M.off, M.off + M.nm, ca, cb)) continue;
          EIGX_HIP_CHECK(hipMemcpy2DAsync(Qa + (size_t)M.off * ldq + ca, (size_t)ldq * 8,
                                          Qb + (size_t)M.off * ldq + ca, (size_t)ldq * 8, (size_t)(cb - ca) * 8,
                                          (size_t)M.nm, hipMemcpyDeviceToDevice, st));
        }
      }
      Dfinal = C.Dcur;
      // classic flow: the host arrays of this pass are free again once the stream is idle; pipelined: the two arenas
      // alternate, and the wait for the next z orders everything that matters
      if (!pipe || trace) EIGX_HIP_CHECK(hipStreamSynchronize(st));
      if (trace) {
        long sumK = 0, sumN = 0;
        for (const MergeDev& M : mds) { sumK += M.K; sumN += M.nm; }
        long mine = 0;   // secular roots this rank solved (several GPUs: K (r+1)/P - K r/P of every merge)
        for (const MergeDev& M : mds) mine += (P > 1) ? (long)((long)M.K * (ctx.grid.rank + 1) / P - (long)M.K * ctx.grid.rank / P) : M.K;
        fprintf(stderr, "[eigx dc] rank %d/%d height %d pass %d: %zu merges (non-deflated %ld of %ld; secular roots solved here %ld), "
                "z gather + D2H %.3f ms, host deflation %.3f ms, device part %.3f ms (GEMM enqueue %.3f ms)%s\n", ctx.grid.rank, P, h, k,
                ids.size(), sumK, sumN, mine, (tt1 - tt0) * 1e3, (tt2 - tt1) * 1e3, (now_s() - tt2) * 1e3, t_gemm_enq * 1e3,
                batched ? " [one product launch]" : "");
      }
    }
  }
  run_side_work();   // (no pass at all: a matrix of one leaf)

  // ---- final sort + copy-out ----------------------------------------------------------------------------
  stage_trace(ctx.grid.rank, "D&C merges done");
  double* Dh = ar[0].Dh;
  hipStream_t up = st;                          // stream of the two uploads (permutation, eigenvalues)
  if (early_final) {
    EIGX_HIP_CHECK(hipEventSynchronize(ctx.dc_z_ev));
    Dh = early_final;
    up = ctx.dc_stream;                         // not behind the product that is still running on the compute stream
  } else {
    EIGX_HIP_CHECK(hipMemcpyAsync(Dh, Dfinal, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
  }
  ord.resize(n);
  for (int i = 0; i < n; ++i) ord[i] = std::make_pair(Dh[i], i);
  std::sort(ord.begin(), ord.end());
  std::vector<int> perm(n);
  std::vector<double> wh(n);
  for (int i = 0; i < n; ++i) { perm[i] = ord[i].second; wh[i] = ord[i].first * nrm; }
  if (nrm == 0.0) for (int i = 0; i < n; ++i) wh[i] = 0.0;
  EIGX_HIP_CHECK(hipMemcpyAsync(perm_dev, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, up));
  EIGX_HIP_CHECK(hipMemcpyAsync(w_dev, wh.data(), (size_t)n * 8, hipMemcpyHostToDevice, up));
  if (up != st) {
    EIGX_HIP_CHECK(hipEventRecord(ctx.dc_b_ev, up));
    EIGX_HIP_CHECK(hipStreamWaitEvent(st, ctx.dc_b_ev, 0));
  }
  if (nvec > 0 && z_dev && P == 1)
    hipLaunchKernelGGL(final_permute_kernel, dim3(8, nvec), dim3(256), 0, st, perm_dev, Qa, ldq, n, z_dev, ldz, nvec);
  if (nvec > 0 && z_dev && P > 1) {
    // row blocks of Q -> column blocks of the sorted eigenvector matrix: one all-to-all of rp x zc pieces
    const int zc = (nvec + P - 1) / P;
    const size_t piece = (size_t)rp * zc;
    double* sendb = ctx.pool.get_t<double>("mg.xsend", piece * P);
    double* recvb = ctx.pool.get_t<double>("mg.xrecv", piece * P);
    hipLaunchKernelGGL(pack_q_for_cols_kernel, dim3(8, zc, P), dim3(256), 0, st, perm_dev, Qa, ldq, r0, r1 - r0, rp, zc, nvec,
                       sendb);
    stage_trace(ctx.grid.rank, "D&C all-to-all: doubles per piece", (long)piece);
    comm_exchange_big(ctx, COMM_WORLD, sendb, piece, recvb, piece, st);
    hipLaunchKernelGGL(unpack_cols_kernel, dim3(8, zc, P), dim3(256), 0, st, (const double*)recvb, rp, zc, n, z_dev, ldz);
  }
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  if (pipe) EIGX_HIP_CHECK(hipStreamSynchronize(ctx.dc_stream));
  stage_trace(ctx.grid.rank, "D&C done");
  EIGX_HIP_CHECK(hipGetLastError());
  ctx.timers[11] = gemm_flops;
}

}  // namespace eigx
