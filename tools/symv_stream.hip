// symv_stream.hip -- lab tool: what the SYMV's load loop alone sustains for different orders of the same loads.
// Upper block triangle of a column-major matrix, tile T = 128*RB per workgroup, 1-D grid over the triangle (row-major),
// wave w owns the tile's columns [w*T/4, (w+1)*T/4), lane = 2 rows (16-byte loads, 1 KiB per wave instruction),
// units of 8 loads, two units in flight (the product kernel's structure).  ORD selects which 8 loads form a unit:
//   0: 8 columns x one 128-row block (product kernel today)        -> 1 KiB pieces per column, next piece one unit later
//   1: 8/RB columns x all RB row blocks of the tile, back to back  -> RB KiB contiguous per column, issued together
//   2: like 0, but the grid walks the triangle column-major (tiles of one tile column consecutive)
// build: hipcc --offload-arch=gfx950 -O3 -o build/symv_stream tools/symv_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// ST: partial-sum stores like the product kernel's (0 none; 1 row sums at the end of the tile + column sums after every
// 8-column group, plain stores; 2 the same with non-temporal stores; 3 row sums only; 4 column sums only)
template <int RB, int ORD, bool NT, int ST = 0>
__global__ __launch_bounds__(256) void rd_sym(const double* __restrict__ A, int ld, int nt, double* out, double* Y = nullptr, int ldp = 0) {
  extern __shared__ double dyn_lds[];   // only to limit the number of resident workgroups (launch parameter)
  constexpr int T = 128 * RB;
  const int bid = blockIdx.x;
  int ty, tx;
  if (ORD == 2) {
    // column-major over the upper block triangle: column tx holds tx+1 tiles
    int c = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
    while (c * (c + 1) / 2 > bid) --c;
    while ((c + 1) * (c + 2) / 2 <= bid) ++c;
    tx = c; ty = bid - c * (c + 1) / 2;
  } else {
    const float fn = 2.0f * (float)nt + 1.0f;
    int r = (int)((fn - sqrtf(fn * fn - 8.0f * (float)bid)) * 0.5f);
    if (r < 0) r = 0;
    if (r > nt - 1) r = nt - 1;
    while (r > 0 && r * nt - r * (r - 1) / 2 > bid) --r;
    while ((r + 1) * nt - (r + 1) * r / 2 <= bid) ++r;
    ty = r; tx = r + (bid - (r * nt - r * (r - 1) / 2));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* base = A + (size_t)(tx * T + wave * (T / 4)) * ld + ty * T + lane * 2;
  constexpr int NU = (T / 4 / 8) * RB;   // units of 8 loads per wave
  auto load = [&](d2 (&v)[8], int u) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int col, rb;
      if (ORD == 1) {
        constexpr int CPU = 8 / RB;           // columns per unit
        col = u * CPU + k / RB; rb = k % RB;
      } else {
        col = (u / RB) * 8 + k; rb = u % RB;
      }
      const d2* q = (const d2*)(base + (size_t)col * ld + rb * 128);
      v[k] = NT ? __builtin_nontemporal_load(q) : *q;
    }
  };
  double s = 0.0;
  d2 a0[8], a1[8];
  load(a0, 0);
  double* YC = Y;                                  // [ty][2][ldp]
  double* YR = Y + (size_t)(nt + 1) * 2 * ldp;     // [tx][2][ldp]
  auto st = [&](double* q, double v) {
    if (ST == 2) __builtin_nontemporal_store(v, q);
    else if (ST == 5) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (ST == 6) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *q = v;
  };
  auto colstore = [&](int u) {
    if ((ST == 1 || ST == 2 || ST == 4 || ST == 5 || ST == 6 || ST == 7) && (u % RB) == RB - 1 && (lane & 7) == 0) {
      const int j = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
      const int c = (ST == 7 ? 0 : tx * T) + wave * (T / 4) + (u / RB) * 8 + j;
      for (int a = 0; a < 2; ++a) {
        double* q = YC + ((size_t)(ST == 7 ? (bid & 15) : ty) * 2 + a) * ldp + c;
        st(q, s);
      }
    }
  };
#pragma unroll 1
  for (int u = 0; u < NU; u += 2) {
    load(a1, u + 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a0[k].x + a0[k].y;
    colstore(u);
    if (u + 2 < NU) load(a0, u + 2);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a1[k].x + a1[k].y;
    colstore(u + 1);
  }
  if (ST == 1 || ST == 2 || ST == 3 || ST == 5 || ST == 6 || ST == 7) {
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256)
      for (int a = 0; a < 2; ++a) {
        double* q = YR + ((size_t)(ST == 7 ? (bid & 15) : tx) * 2 + a) * ldp + (ST == 7 ? 0 : ty * T) + t;
        st(q, s);
      }
  }
  if (s == 1.23456789) { out[0] = s; dyn_lds[0] = s; }
}

// super-tile: a workgroup walks SUP x SUP sub-tiles of 256 x 256 (sub-row by sub-row); row sums are stored once per
// sub-row (accumulated over its SUP sub-tiles in registers), column sums once per super-tile (accumulated in LDS in the
// product kernel; here: one store of 256*SUP values at the end) -> partial-sum volume / SUP at the register footprint of
// the 256 tile
template <int SUP, bool STORES>
__global__ __launch_bounds__(256) void rd_super(const double* __restrict__ A, int ld, int nts, double* out, double* Y, int ldp) {
  extern __shared__ double dyn_lds[];
  constexpr int T = 256, RB = 2;
  const int bid = blockIdx.x;
  const float fn = 2.0f * (float)nts + 1.0f;
  int r = (int)((fn - sqrtf(fn * fn - 8.0f * (float)bid)) * 0.5f);
  if (r < 0) r = 0;
  if (r > nts - 1) r = nts - 1;
  while (r > 0 && r * nts - r * (r - 1) / 2 > bid) --r;
  while ((r + 1) * nts - (r + 1) * r / 2 <= bid) ++r;
  const int sty = r, stx = r + (bid - (r * nts - r * (r - 1) / 2));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* YC = Y;
  double* YR = Y + (size_t)(nts + 1) * 2 * ldp;
  double s = 0.0;
  constexpr int NU = (T / 4 / 8) * RB;
  for (int sy = 0; sy < SUP; ++sy) {
    for (int sx = 0; sx < SUP; ++sx) {
      const int ty = sty * SUP + sy, tx = stx * SUP + sx;
      if (ty > tx) continue;
      const double* base = A + (size_t)(tx * T + wave * (T / 4)) * ld + ty * T + lane * 2;
      auto load = [&](d2 (&v)[8], int u) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int col = (u / RB) * 8 + k, rb = u % RB;
          v[k] = __builtin_nontemporal_load((const d2*)(base + (size_t)col * ld + rb * 128));
        }
      };
      d2 a0[8], a1[8];
      load(a0, 0);
#pragma unroll 1
      for (int u = 0; u < NU; u += 2) {
        load(a1, u + 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a0[k].x + a0[k].y;
        if (u + 2 < NU) load(a0, u + 2);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a1[k].x + a1[k].y;
      }
    }
    if (STORES) {
      __syncthreads();
      for (int t = threadIdx.x; t < T; t += 256)
        for (int a = 0; a < 2; ++a) YR[((size_t)stx * 2 + a) * ldp + (sty * SUP + sy) * T + t] = s;
    }
  }
  if (STORES) {
    __syncthreads();
    for (int t = threadIdx.x; t < T * SUP; t += 256)
      for (int a = 0; a < 2; ++a) YC[((size_t)sty * 2 + a) * ldp + stx * SUP * T + t] = s;
  }
  if (s == 1.23456789) { out[0] = s; dyn_lds[0] = s; }
}


// strip pieces (round 3): a PERSISTENT grid of G workgroups, one job each: job = (strip of H = 1024 rows, column range
// [c0, c1) in units of 8 columns) of the upper triangle, all jobs of (nearly) equal area.  The 4 waves read the SAME 8
// columns at different 128-row blocks (wave w: blocks w and 7 - w of the strip), so row sums are wave-private registers
// over the whole job (stored once at its end: one row-sum slot per job of the strip) and column sums complete over the
// strip's 1024 rows (one column-sum slot per strip: combined over the 4 waves through LDS every 256 columns).
struct StripJob { int s, c0, c1, slot; };
template <bool STORES>
__global__ __launch_bounds__(256) void rd_strip(const double* __restrict__ A, int ld, int L, const StripJob* jobs, double* out,
                                                double* Y, int ldp, int CS) {
  __shared__ double colpart[4][2][256];
  constexpr int H = 1024;
  const StripJob J = jobs[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk[2] = {wave, 7 - wave};
  double s = 0.0;
  const int r0[2] = {J.s * H + blk[0] * 128 + lane * 2, J.s * H + blk[1] * 128 + lane * 2};
  auto load = [&](d2 (&v)[8], int c, int b) {
    // whole unit below the diagonal -> nothing to read (the product kernel skips it as well)
    const bool any = (J.s * H + blk[b] * 128) <= c + 7 && r0[b] < L;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (any) v[k] = __builtin_nontemporal_load((const d2*)(A + (size_t)(c + k) * ld + r0[b]));
      else v[k] = d2{0.0, 0.0};
    }
  };
  d2 a0[8], a1[8];
  load(a0, J.c0, 0);
  int cb = J.c0;   // start of the current 256-column block
#pragma unroll 1
  for (int c = J.c0; c < J.c1; c += 8) {
    load(a1, c, 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a0[k].x + a0[k].y;
    if (c + 8 < J.c1) load(a0, c + 8, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a1[k].x + a1[k].y;
    if (STORES) {
      if ((lane & 7) == 0) { colpart[wave][0][(c - cb) + (lane >> 3)] = s; colpart[wave][1][(c - cb) + (lane >> 3)] = s; }
      if (c + 8 - cb == 256 || c + 8 >= J.c1) {
        __syncthreads();
        const int t = threadIdx.x;
        if (cb + t < J.c1)
          for (int a = 0; a < 2; ++a)
            Y[((size_t)J.s * 2 + a) * ldp + cb + t] = (colpart[0][a][t] + colpart[1][a][t]) + (colpart[2][a][t] + colpart[3][a][t]);
        __syncthreads();
        cb = c + 8;
      }
    }
  }
  if (STORES) {
    for (int b = 0; b < 2; ++b)
      for (int a = 0; a < 2; ++a) {
        double* q = Y + ((size_t)(CS + J.slot) * 2 + a) * ldp + r0[b];
        q[0] = s; q[1] = s;
      }
  }
  if (s == 1.23456789) out[0] = s;
}

__global__ void fill_rand(double* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    p[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 30720;
  const int ld = argc > 2 ? atoi(argv[2]) : 32864;
  const size_t bytes = (size_t)ld * n * 8;
  double* p; CK(hipMalloc(&p, bytes));
  if (argc > 3 && atoi(argv[3])) { hipLaunchKernelGGL(fill_rand, dim3(8192), dim3(256), 0, 0, p, bytes / 8); printf("random data\n"); }
  else CK(hipMemset(p, 0, bytes));
  double* out; CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch, double nbytes) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %.0f GB/s\n", name, 3 * nbytes / (ms * 1e-3) / 1e9);
    fflush(stdout);
  };
#define RUN(RBv, ORDv, NTv) RUNL(RBv, ORDv, NTv, 0)
#define RUNL(RBv, ORDv, NTv, LDSB)                                                                                   \
  {                                                                                                                   \
    const int T = 128 * RBv, nt = n / T;                                                                              \
    const int tiles = nt * (nt + 1) / 2;                                                                              \
    char nm[96];                                                                                                      \
    snprintf(nm, 96, "triangle n=%d ld=%d, T=%d, order %d%s, lds %d KB", n, ld, T, ORDv, NTv ? ", nt" : "", LDSB / 1024); \
    time(nm, [&] { hipLaunchKernelGGL((rd_sym<RBv, ORDv, NTv>), dim3(tiles), dim3(256), LDSB, 0, p, ld, nt, out); },  \
         8.0 * tiles * T * T);                                                                                        \
  }
  if (argc > 4) {   // short form: the product kernel's shape only
    RUNL(2, 0, true, 40 * 1024) RUNL(2, 2, true, 40 * 1024)
    // with the product kernel's partial-sum stores
    const int T = 256, nt = n / T, tiles = nt * (nt + 1) / 2, ldp = n + 64;
    double* Y; CK(hipMalloc(&Y, (size_t)(2 * nt + 2) * 2 * ldp * 8));
#define RUNS(STv)                                                                                                  \
    {                                                                                                               \
      char nm[96];                                                                                                  \
      snprintf(nm, 96, "T=256, order 0, nt, lds 40 KB, stores variant %d", STv);                                    \
      time(nm, [&] { hipLaunchKernelGGL((rd_sym<2, 0, true, STv>), dim3(tiles), dim3(256), 40 * 1024, 0, p, ld, nt, out, Y, ldp); }, \
           8.0 * tiles * T * T);                                                                                    \
    }
    RUNS(0) RUNS(1) RUNS(7) RUNS(0) RUNS(1) RUNS(7) RUNS(3) RUNS(4)
#define RUNSUP(SUPv, STv)                                                                                            \
    {                                                                                                                 \
      const int nts = nt / SUPv, st = nts * (nts + 1) / 2;                                                            \
      double tl = 0; for (int a_ = 0; a_ < nts; ++a_) for (int b_ = a_; b_ < nts; ++b_) tl += (a_ == b_) ? SUPv * (SUPv + 1) / 2 : SUPv * SUPv; \
      char nm[96];                                                                                                    \
      snprintf(nm, 96, "super-tiles %dx%d of 256, nt, lds 40 KB, stores %d", SUPv, SUPv, (int)STv);                    \
      time(nm, [&] { hipLaunchKernelGGL((rd_super<SUPv, STv>), dim3(st), dim3(256), 40 * 1024, 0, p, ld, nts, out, Y, ldp); }, \
           8.0 * tl * T * T);                                                                                         \
    }

    for (int G : {256, 512, 768, 1024}) {
      // equal-area partition of the upper triangle of order n into G jobs of strips of 1024 rows
      const int H = 1024, ns = (n + H - 1) / H;
      std::vector<StripJob> jobs;
      const double total = 0.5 * (double)n * (n + 1);
      auto strip_area = [&](int s_, int c) {   // elements of strip s_ in columns [s_ H, c)
        double a = 0;
        const int rb = s_ * H, re = (rb + H < n) ? rb + H : n;
        const int x = c - rb;
        const int hh = re - rb;
        if (x <= 0) return 0.0;
        if (x <= hh) return 0.5 * (double)x * (x + 1);
        a = 0.5 * (double)hh * (hh + 1) + (double)(x - hh) * hh;
        return a;
      };
      double maxa = 0, suma = 0;
      for (int s_ = 0; s_ < ns; ++s_) {
        const double as = strip_area(s_, n);
        int np = (int)(as / (total / G) + 0.5);
        if (np < 1) np = 1;
        int cprev = s_ * H;
        for (int p_ = 0; p_ < np; ++p_) {
          int c1 = n;
          if (p_ + 1 < np) {
            const double want = as * (p_ + 1) / np;
            int lo = cprev, hi = n;
            while (hi - lo > 8) { const int mid = ((lo + hi) / 2) / 8 * 8; if (strip_area(s_, mid) < want) lo = mid; else hi = mid; }
            c1 = hi / 8 * 8;
          }
          if (c1 > cprev) {
            jobs.push_back({s_, cprev, c1, p_});
            const double ja = strip_area(s_, c1) - strip_area(s_, cprev);
            if (ja > maxa) maxa = ja;
            suma += ja;
          }
          cprev = c1;
        }
      }
      StripJob* dj; CK(hipMalloc(&dj, jobs.size() * sizeof(StripJob)));
      CK(hipMemcpy(dj, jobs.data(), jobs.size() * sizeof(StripJob), hipMemcpyHostToDevice));
      const int CS = ns;
      double* Y2; CK(hipMalloc(&Y2, (size_t)(CS + 160) * 2 * ldp * 8));
      char nm[128];
      snprintf(nm, 128, "strip pieces H=1024, %d jobs (target %d), max/mean area %.3f, no stores", (int)jobs.size(), G, maxa / (suma / jobs.size()));
      time(nm, [&] { hipLaunchKernelGGL((rd_strip<false>), dim3((unsigned)jobs.size()), dim3(256), 0, 0, p, ld, n, dj, out, Y2, ldp, CS); }, 8.0 * suma);
      snprintf(nm, 128, "strip pieces H=1024, %d jobs (target %d), with partial-sum stores", (int)jobs.size(), G);
      time(nm, [&] { hipLaunchKernelGGL((rd_strip<true>), dim3((unsigned)jobs.size()), dim3(256), 0, 0, p, ld, n, dj, out, Y2, ldp, CS); }, 8.0 * suma);
      CK(hipFree(dj)); CK(hipFree(Y2));
    }
    if (argc > 5) { RUNSUP(1, false) RUNSUP(1, true) RUNSUP(2, false) RUNSUP(2, true) RUNSUP(4, false) RUNSUP(4, true) }
    {
      const int T5 = 512, nt5 = n / T5, tiles5 = nt5 * (nt5 + 1) / 2;
      time("T=512, order 0, nt, lds 40 KB, stores variant 1", [&] { hipLaunchKernelGGL((rd_sym<4, 0, true, 1>), dim3(tiles5), dim3(256), 40 * 1024, 0, p, ld, nt5, out, Y, ldp); }, 8.0 * tiles5 * T5 * T5);
    }
    return 0;
  }
  RUN(1, 0, true) RUN(2, 0, false) RUN(2, 0, true) RUN(4, 0, true)
  // resident workgroups per CU limited through the dynamic LDS size: 160 KB / size
  RUNL(2, 0, true, 20 * 1024) RUNL(2, 0, true, 32 * 1024) RUNL(2, 0, true, 40 * 1024) RUNL(2, 0, true, 53 * 1024) RUNL(2, 0, true, 80 * 1024)
  RUNL(1, 0, true, 20 * 1024) RUNL(1, 0, true, 32 * 1024) RUNL(1, 0, true, 40 * 1024)
  RUNL(4, 0, true, 40 * 1024) RUNL(4, 0, true, 80 * 1024)
  return 0;
}
