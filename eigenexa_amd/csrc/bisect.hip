// bisect.hip -- eigenvalues of the symmetric band matrix (band = 1 tridiagonal, 2 pentadiagonal) by Sturm counts,
// gfx950.  Replaces eigen_bisect (src/bisect.F:67-397) and eigen_bisect2 (src/bisect2.F:71-718): mode 'N' of
// eigen_sx / eigen_s (eigenvalues only, src/eigen_sx.F:219-221) and the refinement of mode 'X'.
//
// Same mathematics as the reference:
//   * count(x) = number of eigenvalues < x = number of negative pivots of an LDL^T factorisation of T - xI
//     (Sylvester).  Tridiagonal: the classic three-term recurrence with a pivmin guard.  Pentadiagonal: a
//     4 x 4 window of the running Schur complement with "diagonal-neighbour" pivoting -- the larger of the
//     two leading diagonal entries is the pivot (symmetric interchange inside the window, one column of fill),
//     and a 2 x 2 block pivot when both vanish (the scheme of sturm2_LDLT, src/bisect2.F:398-676).
//   * Gershgorin bounds widened by a few ulps (src/bisect2.F:147-185).
// MI355X re-design: the reference gives each MPI rank n/P eigenvalues and bisects them one after another
// (n Sturm sweeps of length n per rank, 128 iterations); here every sweep is one GPU thread and a round
// evaluates S interior points of every eigenvalue's interval at once (multi-section): round 0 puts n*S points
// on the Gershgorin interval and brackets every eigenvalue by a binary search in the monotone count array, each
// later round shrinks every bracket by S + 1.  n*S ~ 2.6e5 threads, ~7 rounds at N = 8192 instead of 53
// dependent bisection steps.  d / e tiles are staged through LDS (every thread of a workgroup walks the same
// matrix rows).  The invariant count(lb) <= k < count(ub) is kept by construction, so the result brackets the
// k-th eigenvalue even where rounding makes the pivoted count non-monotone.
#include "eigx_context.h"
#include <algorithm>
#include <cfloat>
#include <vector>

namespace eigx {

namespace {

constexpr int BS_TILE = 1024;   // matrix rows staged per LDS tile

struct BisArgs {
  int n, lde, band, S;
  const double* d; const double* e;
  double* lb; double* ub;     // [n] current brackets
  int* cnt;                   // [n * S] counts of the round
  double* scal;               // {glb, gub, pivmin, eps_abs}
};

// Gershgorin interval, pivmin and the absolute tolerance (one workgroup)
__global__ __launch_bounds__(256) void bis_bounds_kernel(BisArgs a) {
  __shared__ double slo[256], shi[256], sem[256];
  const int tid = threadIdx.x;
  double lo = DBL_MAX, hi = -DBL_MAX, em = 0.0;
  for (int i = tid; i < a.n; i += 256) {
    double r = 0.0;
    for (int b = 1; b <= a.band; ++b) {
      if (i - b >= 0) r += fabs(a.e[(size_t)(b - 1) * a.lde + i]);          // T(i-b, i)
      if (i + b < a.n) r += fabs(a.e[(size_t)(b - 1) * a.lde + i + b]);     // T(i, i+b)
      if (i - b >= 0) em = fmax(em, fabs(a.e[(size_t)(b - 1) * a.lde + i]));
    }
    lo = fmin(lo, a.d[i] - r);
    hi = fmax(hi, a.d[i] + r);
  }
  slo[tid] = lo; shi[tid] = hi; sem[tid] = em;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { slo[tid] = fmin(slo[tid], slo[tid + s]); shi[tid] = fmax(shi[tid], shi[tid + s]); sem[tid] = fmax(sem[tid], sem[tid + s]); }
    __syncthreads();
  }
  if (tid == 0) {
    const double eps = DBL_EPSILON;
    const double tn = fmax(fabs(slo[0]), fabs(shi[0]));
    const double x = (fabs(slo[0]) + fabs(shi[0])) * eps;
    const double epsa = eps * sem[0];
    a.scal[0] = (slo[0] - x) - epsa - DBL_MIN;
    a.scal[1] = (shi[0] + x) + epsa + DBL_MIN;
    a.scal[2] = fmax(DBL_MIN * fmax(1.0, sem[0] * sem[0]), DBL_MIN);   // pivmin (dstebz rule)
    a.scal[3] = fmax(eps * tn, DBL_MIN);
  }
}

// ---- Sturm counts over one LDS tile ---------------------------------------------------------------
struct TriState { double q; int cnt; };
struct PenState {
  // lower triangle of the 4 x 4 window of the Schur complement; w41 is zero whenever a row has just entered
  double w11, w21, w22, w31, w32, w33, w41, w42, w43, w44;
  int cnt;
  int trail;        // both leading diagonal entries vanished: the next step eliminates a 2 x 2 block
  double pa, pb, pc;  // the row that could not enter while the 2 x 2 block was pending
};

__device__ __forceinline__ void tri_step(TriState& s, double dd, double e2, double x, double pivmin) {
  double q = (dd - x) - e2 / s.q;
  if (fabs(q) <= pivmin) q = -pivmin;
  s.cnt += (q < 0.0);
  s.q = q;
}

// One elimination step of the pentadiagonal window, then row (c, b, a) enters at the bottom:
// a = T(i,i) - x, b = T(i-1,i), c = T(i-2,i).  Rows beyond the matrix are (0, 0, 1): decoupled positive pivots.
__device__ __forceinline__ void pen_step(PenState& s, double a, double b, double c, double pivmin) {
  if (s.trail) {
    // 2 x 2 block pivot [[0, e0], [e0, 0]] on window rows 1, 2: Schur complement of rows 3, 4
    double e0 = s.w21;
    const bool tiny = fabs(e0) <= pivmin;
    if (tiny) e0 = pivmin;
    const double f0 = s.w31, g0 = s.w41, f1 = s.w32, g1 = s.w42;
    const double r = 1.0 / e0;
    const double n11 = s.w33 - 2.0 * f0 * f1 * r;
    const double n21 = s.w43 - (g0 * f1 + g1 * f0) * r;
    const double n22 = s.w44 - 2.0 * g0 * g1 * r;
    s.cnt += tiny ? 2 : 1;
    // the pending row and the new row enter as window rows 3, 4
    s.w11 = n11; s.w21 = n21; s.w22 = n22;
    s.w31 = s.pc; s.w32 = s.pb; s.w33 = s.pa;
    s.w41 = 0.0; s.w42 = c; s.w43 = b; s.w44 = a;
    s.trail = 0;
    return;
  }
  if (fabs(s.w11) < fabs(s.w22)) {   // symmetric interchange of window rows/columns 1 and 2
    double t = s.w11; s.w11 = s.w22; s.w22 = t;
    t = s.w31; s.w31 = s.w32; s.w32 = t;
    t = s.w41; s.w41 = s.w42; s.w42 = t;
  }
  if (s.w11 == 0.0) {                // both candidates vanish: 2 x 2 block next time, this row waits
    s.trail = 1;
    s.pa = a; s.pb = b; s.pc = c;
    return;
  }
  double d0 = s.w11;
  if (fabs(d0) < pivmin) d0 = -pivmin;
  const double r = 1.0 / d0;
  const double e0 = s.w21, f0 = s.w31, g0 = s.w41;
  s.cnt += (d0 < 0.0);
  const double n11 = s.w22 - e0 * e0 * r;
  const double n21 = s.w32 - e0 * f0 * r;
  const double n22 = s.w33 - f0 * f0 * r;
  const double n31 = s.w42 - e0 * g0 * r;
  const double n32 = s.w43 - f0 * g0 * r;
  const double n33 = s.w44 - g0 * g0 * r;
  s.w11 = n11; s.w21 = n21; s.w22 = n22; s.w31 = n31; s.w32 = n32; s.w33 = n33;
  s.w41 = 0.0; s.w42 = c; s.w43 = b; s.w44 = a;
}

// Inertia of what is left in the window after the last matrix row has entered: the same pivot rule on a dense
// symmetric 4 x 4 block that shrinks by one (or two) rows per step.  Rows of the initial identity that are
// still there are positive pivots and add nothing.
__device__ int pen_finish(PenState& s, double pivmin) {
  if (s.trail) pen_step(s, 1.0, 0.0, 0.0, pivmin);   // pending 2 x 2 block: the waiting row enters, plus a decoupled positive one
  double W[4][4];
  W[0][0] = s.w11;
  W[1][0] = s.w21; W[1][1] = s.w22;
  W[2][0] = s.w31; W[2][1] = s.w32; W[2][2] = s.w33;
  W[3][0] = s.w41; W[3][1] = s.w42; W[3][2] = s.w43; W[3][3] = s.w44;
  int cnt = 0;
  int m = 4;
  while (m > 0) {
    if (m >= 2 && fabs(W[0][0]) < fabs(W[1][1])) {
      double t = W[0][0]; W[0][0] = W[1][1]; W[1][1] = t;
      for (int r = 2; r < m; ++r) { t = W[r][0]; W[r][0] = W[r][1]; W[r][1] = t; }
    }
    if (W[0][0] == 0.0 && m >= 2) {
      double e0 = W[1][0];
      const bool tiny = fabs(e0) <= pivmin;
      if (tiny) e0 = pivmin;
      cnt += tiny ? 2 : 1;
      const double r = 1.0 / e0;
      double N[2][2] = {{0, 0}, {0, 0}};
      for (int i = 2; i < m; ++i)
        for (int j = 2; j <= i; ++j) N[i - 2][j - 2] = W[i][j] - (W[i][0] * W[j][1] + W[i][1] * W[j][0]) * r;
      for (int i = 2; i < m; ++i)
        for (int j = 2; j <= i; ++j) W[i - 2][j - 2] = N[i - 2][j - 2];
      m -= 2;
    } else {
      double d0 = W[0][0];
      if (fabs(d0) < pivmin) d0 = -pivmin;
      cnt += (d0 < 0.0);
      const double r = 1.0 / d0;
      double N[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int i = 1; i < m; ++i)
        for (int j = 1; j <= i; ++j) N[i - 1][j - 1] = W[i][j] - W[i][0] * W[j][0] * r;
      for (int i = 1; i < m; ++i)
        for (int j = 1; j <= i; ++j) W[i - 1][j - 1] = N[i - 1][j - 1];
      m -= 1;
    }
  }
  return cnt;
}

// count(x) for every thread's own x; all threads of the workgroup walk the matrix together through LDS tiles
template <int BAND>
__device__ int sturm_count(const BisArgs& a, double x, double pivmin, double* sd, double* se1, double* se2) {
  const int tid = threadIdx.x;
  TriState ts; ts.q = 1.0; ts.cnt = 0;
  PenState ps;
  ps.w11 = ps.w22 = ps.w33 = ps.w44 = 1.0;
  ps.w21 = ps.w31 = ps.w32 = ps.w41 = ps.w42 = ps.w43 = 0.0;
  ps.cnt = 0; ps.trail = 0; ps.pa = 1.0; ps.pb = ps.pc = 0.0;
  for (int i0 = 0; i0 < a.n; i0 += BS_TILE) {
    const int len = (a.n - i0 < BS_TILE) ? a.n - i0 : BS_TILE;
    __syncthreads();
    for (int t = tid; t < len; t += blockDim.x) {
      const int i = i0 + t;
      sd[t] = a.d[i];
      const double e1 = (i >= 1) ? a.e[i] : 0.0;
      se1[t] = (BAND == 1) ? e1 * e1 : e1;
      if (BAND == 2) se2[t] = (i >= 2) ? a.e[(size_t)a.lde + i] : 0.0;
    }
    __syncthreads();
    if (BAND == 1) {
      for (int t = 0; t < len; ++t) tri_step(ts, sd[t], se1[t], x, pivmin);
    } else {
      for (int t = 0; t < len; ++t) pen_step(ps, sd[t] - x, se1[t], se2[t], pivmin);
    }
  }
  if (BAND == 1) return ts.cnt;
  return ps.cnt + pen_finish(ps, pivmin);
}

// round 0: G = n*S points on the Gershgorin interval
template <int BAND>
__global__ __launch_bounds__(256) void bis_grid_kernel(BisArgs a) {
  __shared__ double sd[BS_TILE], se1[BS_TILE], se2[BS_TILE];
  const long G = (long)a.n * a.S;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const double lo = a.scal[0], hi = a.scal[1], pivmin = a.scal[2];
  const long tc = t < G ? t : G - 1;
  const double x = lo + (hi - lo) * ((double)(tc + 1) / (double)(G + 1));
  const int c = sturm_count<BAND>(a, x, pivmin, sd, se1, se2);
  if (t < G) a.cnt[t] = c;
}

// bracket eigenvalue k (0-based, ascending): smallest grid point with count >= k + 1 is the upper end
__global__ void bis_bracket_kernel(BisArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n) return;
  const long G = (long)a.n * a.S;
  const double lo = a.scal[0], hi = a.scal[1];
  long l = 0, r = G;   // first index in [0, G] with cnt >= k+1 (G = none)
  while (l < r) {
    const long mid = (l + r) >> 1;
    if (a.cnt[mid] >= k + 1) r = mid; else l = mid + 1;
  }
  auto xg = [&](long t) { return lo + (hi - lo) * ((double)(t + 1) / (double)(G + 1)); };
  a.lb[k] = (l == 0) ? lo : xg(l - 1);
  a.ub[k] = (l == G) ? hi : xg(l);
}

// refinement round: thread (k, s) evaluates the s-th interior point of bracket k
template <int BAND>
__global__ __launch_bounds__(256) void bis_refine_kernel(BisArgs a) {
  __shared__ double sd[BS_TILE], se1[BS_TILE], se2[BS_TILE];
  const long G = (long)a.n * a.S;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long tc = t < G ? t : G - 1;
  const int k = (int)(tc / a.S), s = (int)(tc - (long)k * a.S);
  const double lb = a.lb[k], ub = a.ub[k];
  const double x = lb + (ub - lb) * ((double)(s + 1) / (double)(a.S + 1));
  const int c = sturm_count<BAND>(a, x, a.scal[2], sd, se1, se2);
  if (t < G) a.cnt[t] = c;
}

__global__ void bis_update_kernel(BisArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n) return;
  const double lb = a.lb[k], ub = a.ub[k];
  double nlb = lb, nub = ub;
  for (int s = 0; s < a.S; ++s) {
    const double x = lb + (ub - lb) * ((double)(s + 1) / (double)(a.S + 1));
    if (!(x > nlb && x < ub)) continue;          // bracket exhausted in floating point
    if (a.cnt[(long)k * a.S + s] >= k + 1) { nub = x; break; }
    nlb = x;
  }
  a.lb[k] = nlb; a.ub[k] = nub;
}

__global__ void bis_final_kernel(BisArgs a, double* w) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < a.n) w[k] = 0.5 * (a.lb[k] + a.ub[k]);
}

int g_bis_threads = 65536;   // target number of concurrent Sturm sweeps n*S (eigx_tune key 1)

}  // namespace

int set_bisect_threads(int v) { const int old = g_bis_threads; if (v > 0) g_bis_threads = v; return old; }

void band_bisect_dev(Context& ctx, int n, const double* d, const double* e, int lde, int band, double* w) {
  if (n <= 0) return;
  hipStream_t st = ctx.stream;
  BisArgs a;
  a.n = n; a.lde = lde; a.band = band; a.d = d; a.e = e;
  int S = 1;
  while (S < 64 && (long)n * (2 * S) <= g_bis_threads) S *= 2;
  a.S = S;
  a.lb = ctx.pool.get_t<double>("bis.lb", (size_t)n);
  a.ub = ctx.pool.get_t<double>("bis.ub", (size_t)n);
  a.cnt = ctx.pool.get_t<int>("bis.cnt", (size_t)n * S);
  a.scal = ctx.pool.get_t<double>("bis.scal", 8);
  const long G = (long)n * S;
  const int gb = (int)((G + 255) / 256), nb = (n + 255) / 256;
  hipLaunchKernelGGL(bis_bounds_kernel, dim3(1), dim3(256), 0, st, a);
  if (band == 1) hipLaunchKernelGGL(bis_grid_kernel<1>, dim3(gb), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(bis_grid_kernel<2>, dim3(gb), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bis_bracket_kernel, dim3(nb), dim3(256), 0, st, a);
  // after round 0 a bracket is (hi-lo)/(G+1) wide; every round divides it by S+1; 2^-54 of the interval is
  // below half an ulp of its end points
  int rounds = 0;
  for (double width = 1.0 / (double)(G + 1); width > 0x1p-56; width /= (double)(S + 1)) ++rounds;
  rounds += 1;
  for (int r = 0; r < rounds; ++r) {
    if (band == 1) hipLaunchKernelGGL(bis_refine_kernel<1>, dim3(gb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(bis_refine_kernel<2>, dim3(gb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(bis_update_kernel, dim3(nb), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(bis_final_kernel, dim3(nb), dim3(256), 0, st, a, w);
  // the reference sorts the result (lazy_qsort, src/bisect2.F:682-712); brackets of neighbouring eigenvalues can
  // overlap by an ulp, so do the same (n doubles through the host: microseconds)
  std::vector<double> h((size_t)n);
  EIGX_HIP_CHECK(hipMemcpyAsync(h.data(), w, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  EIGX_HIP_CHECK(hipStreamSynchronize(st));
  if (!std::is_sorted(h.begin(), h.end())) {
    std::sort(h.begin(), h.end());
    EIGX_HIP_CHECK(hipMemcpyAsync(w, h.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
  }
  EIGX_HIP_CHECK(hipGetLastError());
}

}  // namespace eigx
