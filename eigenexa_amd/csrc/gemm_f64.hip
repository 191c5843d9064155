// gemm_f64.hip -- fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64), gfx950 only.
//
// One kernel family serves the three GEMM-shaped stages of the EigenExa hot path:
//   * trailing rank-2k update  A -= [U W][W U]^T on upper-triangle tiles
//       (reference: eigen_common_2update, src/eigen_t1.F:250-306, two dgemm('N','T') per chunk)
//   * WY back-transformation   W = V^T Z ('T','N'),  Z -= V (T W) ('N','N')
//       (reference: eigen_trbakwy_block_body1/2, src/trbakwy4_body.F:504-741)
//   * divide-and-conquer eigenvector update  Q <- Q2 * S ('N','N')
//       (reference: PDGEMM in src/my_pdlaed1.F:310-341, DGEMM ring in src/FS_PDLAED3.F90:833-860)
//
// Design (MI355X-first, not a translation of the reference's chunked BLAS calls):
//   128x128 output tile per 256-thread workgroup, 4 waves as 2x2, 64x64 per wave =
//   4x4 MFMA 16x16 tiles (16 accumulators x 4 f64). K is consumed in slabs of 16 through a
//   double-buffered LDS stage.  The MFMA operands are swapped (first operand = op(B) fragment,
//   second = op(A) fragment) so that the accumulator's lane index runs along the column-major
//   contiguous dimension of C: every epilogue load/store is 16 lanes x 8 B = one full 128-B line.
//   LDS layouts are padded so that the ds_read_b64 fragment reads are bank-conflict free:
//     "MC" operand (contiguous along the tile's m/n index):  [k][128+16]
//     "KC" operand (contiguous along k):                     [m][16+1]
#include "eigx_common.h"

namespace eigx {

namespace {

constexpr int BK = 16;
constexpr int LD_KC = BK + 1;
// tile geometry for a wave tile of WT x WT (WT = 64: 128x128 workgroup tile; WT = 32: 64x64, used when the
// output is too small to fill 256 CUs with 128x128 tiles)
template <int WT> struct Geo {
  static constexpr int BM = 2 * WT, BN = 2 * WT;
  static constexpr int LD_MC = BM + 16;  // doubles; (LD_MC*2) % 64 == 32 -> lanes 16..31 land on the other bank half
  static constexpr int OPER = (BK * LD_MC > BM * LD_KC) ? BK * LD_MC : BM * LD_KC;
  static constexpr int NL = BM * BK / 256;  // slab elements per thread
  static constexpr int NF = WT / 16;        // MFMA tiles per wave per dimension
};

struct GemmArgs {
  int M, N, K;
  double alpha, beta;
  const double* A;
  int lda;
  const double* B;
  int ldb;
  double* C;
  int ldc;
  int tri_mode;
  int Px, px, Py, py;
  const int* kmapA;  // optional: column of A that holds k-index k ('N' A only): A(:, kmapA[k])
  const int* cmapC;  // optional: column of C that receives n-index n: C(:, cmapC[n])
  const int* kmapB;  // optional: column of B that holds k-index k ('T' B only): B(:, kmapB[k])
  long sA, sB, sC;     // batch strides in elements (blockIdx.y = batch index)
  long sA2, sB2, sC2;  // second-level batch strides (blockIdx.z)
  int ownP, ownp;      // tri mode, multi-GPU: this rank updates tile columns tn with tn % ownP == ownp
  int tri_gb;          // gemm2 tri mode: tile-block edge of the XCD-aware order
  int tn_lo, tn_hi;    // tri mode: only tile columns tn_lo <= tn < tn_hi are updated (look-ahead split of the trailing update)
  int c_stream;        // gemm2: C tiles with non-temporal loads / stores (tuning hook, eigx_tune key 6)
  // gemm2, tri mode 2 (rectangular local block of a 2-D cyclic distribution): the launch holds only the tiles of the
  // active region.  Tile rows in groups of rg_h; group q covers tile columns [rg_c0[q], rg_c1) and starts at position
  // rg_start[q] of the tile list (rg_start[nrg] = total); 0 groups: the full rectangle is launched (older form).
  int nrg, rg_h, rg_c1;
  unsigned rg_magic;   // floor(2^32 / Py) + 1: x / Py == umulhi(x, rg_magic) for the x < 2^22 that occur (Py > 1)
  const GemmBatch* btab;  // gemm_f64_kernel only: per-batch sizes / offsets (blockIdx.y), nullptr = one shape for all
};
// tri mode 2: first tile column of row group q = the tile column that holds the first local column whose global index
// reaches the first global row of the group, clipped to the launch's column window [tn_lo, rg_c1]
__host__ __device__ inline int tri2_c0(const GemmArgs& g, int q, int tiles_n) {
  const long grow_min = (long)q * g.rg_h * 128 * g.Px + g.px;
  const unsigned x = grow_min > g.py ? (unsigned)(grow_min - g.py + g.Py - 1) : 0u;
#ifdef __HIP_DEVICE_COMPILE__
  const unsigned lc = (g.Py == 1) ? x : __umulhi(x, g.rg_magic);
#else
  const unsigned lc = (g.Py == 1) ? x : (unsigned)(((unsigned long long)x * g.rg_magic) >> 32);
#endif
  int c = (lc > (unsigned)(g.N - 1)) ? tiles_n : (int)(lc >> 7);
  if (c < g.tn_lo) c = g.tn_lo;
  if (c > g.rg_c1) c = g.rg_c1;
  return c;
}
typedef double d2s_t __attribute__((ext_vector_type(2)));

// column indices (gather map) of the slab starting at k0 for this thread's NL elements
template <int WT>
__device__ __forceinline__ void load_map(int (&kc)[Geo<WT>::NL], const int* __restrict__ kmap, int k0, int Kmax,
                                         int tid) {
  constexpr int BM = Geo<WT>::BM, NL = Geo<WT>::NL;
  const int kb = k0 + tid / BM;
#pragma unroll
  for (int p = 0; p < NL; ++p) {
    const int k = kb + (256 / BM) * p;
    kc[p] = kmap[k < Kmax ? k : 0];
  }
}

// Load this thread's NL elements of a BM x 16 operand slab (rows m0.., k-range k0..) into regs.
//   MC: element (m,k) at P[m + k*ld]     KC: element (m,k) at P[k + m*ld]
template <bool KC, int WT, bool MAP>
__device__ __forceinline__ void load_slab(double (&r)[Geo<WT>::NL], const double* __restrict__ P, int ld, int m0,
                                          int k0, int Mmax, int Kmax, int tid, const int (&kc)[Geo<WT>::NL]) {
  constexpr int BM = Geo<WT>::BM, NL = Geo<WT>::NL;
  if (!KC) {
    const int m = m0 + (tid & (BM - 1));
    const int kb = k0 + tid / BM;  // 0..256/BM-1, then +256/BM per pass
    const bool mok = m < Mmax;
    if (MAP) {
      // gather variant (its own kernel instantiation: a run-time "map or not" select inside this unrolled
      // loop makes hipcc branch around every load and serialise them).  kc[] = column indices of this slab,
      // loaded one slab ahead by load_map so that map and data loads are not two dependent round trips.
#pragma unroll
      for (int p = 0; p < NL; ++p) {
        const int k = kb + (256 / BM) * p;
        r[p] = (mok && k < Kmax) ? P[(size_t)m + (size_t)kc[p] * ld] : 0.0;
      }
    } else {
#pragma unroll
      for (int p = 0; p < NL; ++p) {
        const int k = kb + (256 / BM) * p;
        r[p] = (mok && k < Kmax) ? P[(size_t)m + (size_t)k * ld] : 0.0;
      }
    }
  } else {
    const int k = k0 + (tid & 15);
    const int mb = m0 + (tid >> 4);  // 0..15, then +16 per pass
    const bool kok = k < Kmax;
#pragma unroll
    for (int p = 0; p < NL; ++p) {
      const int m = mb + 16 * p;
      r[p] = (kok && m < Mmax) ? P[(size_t)k + (size_t)m * ld] : 0.0;
    }
  }
}

template <bool KC, int WT>
__device__ __forceinline__ void store_slab(const double (&r)[Geo<WT>::NL], double* __restrict__ S, int tid) {
  constexpr int BM = Geo<WT>::BM, NL = Geo<WT>::NL, LD_MC = Geo<WT>::LD_MC;
  if (!KC) {
    const int m = tid & (BM - 1);
    const int kb = tid / BM;
#pragma unroll
    for (int p = 0; p < NL; ++p) S[(kb + (256 / BM) * p) * LD_MC + m] = r[p];
  } else {
    const int k = tid & 15;
    const int mb = tid >> 4;
#pragma unroll
    for (int p = 0; p < NL; ++p) S[(mb + 16 * p) * LD_KC + k] = r[p];
  }
}

template <bool KC, int WT>
__device__ __forceinline__ double frag(const double* __restrict__ S, int m, int k) {
  return KC ? S[m * LD_KC + k] : S[k * Geo<WT>::LD_MC + m];
}

template <bool A_KC, bool B_KC, int WT, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int BM = Geo<WT>::BM, BN = Geo<WT>::BN, OPER_DOUBLES = Geo<WT>::OPER, NL = Geo<WT>::NL, NF = Geo<WT>::NF;
  g.A += (long)blockIdx.y * g.sA + (long)blockIdx.z * g.sA2;
  g.B += (long)blockIdx.y * g.sB + (long)blockIdx.z * g.sB2;
  g.C += (long)blockIdx.y * g.sC + (long)blockIdx.z * g.sC2;
  if (g.btab) {   // table batch: the entry's own shape (uniform per workgroup: scalar loads)
    const GemmBatch b = g.btab[blockIdx.y];
    g.M = b.M; g.N = b.N; g.K = b.K;
    g.A += b.offA; g.B += b.offB; g.C += b.offC;
    if (GATHER) { g.kmapA += b.offKA; g.kmapB += b.offKB; }
    if ((int)blockIdx.x >= ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN)) return;
  }
  // stage buffer b: A slab at smem + 2*b*OPER_DOUBLES, B slab right behind it

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware tile order (workgroups are dealt round-robin over the 8 XCDs, so bid%8 labels the XCD):
  //  full mode: each XCD gets a contiguous chunk of the column-major tile list (A/B slab reuse in L2);
  //  tri mode : tile COLUMNS are dealt round-robin to XCDs, because column tn of the upper triangle
  //             holds ~tn+1 active tiles and contiguous chunks would leave XCD 7 with 2x the mean work.
  int bid = blockIdx.x;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  int tm, tn;
  if (g.tri_mode == 1 && tiles_n % 8 == 0) {
    const int xcd = bid & 7, idx = bid >> 3;
    tn = 8 * (idx / tiles_m) + xcd;
    tm = idx % tiles_m;
  } else {
    if (g.tri_mode == 0 && ntiles % 8 == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);
    tm = bid % tiles_m;
    tn = bid / tiles_m;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  if (g.tri_mode != 0) {
    if (g.ownP > 1 && (tn % g.ownP) != g.ownp) return;
    if (tn < g.tn_lo || tn >= g.tn_hi) return;
    // skip tiles strictly below the diagonal: min global row > max global col
    const long grow_min = (long)m0 * g.Px + g.px;
    const int jmax = (n0 + BN - 1 < g.N - 1) ? n0 + BN - 1 : g.N - 1;
    const long gcol_max = (long)jmax * g.Py + g.py;
    if (grow_min > gcol_max) return;
  }

  const int fm = lane & 15, fk = lane >> 4;
  // Accumulators start from (beta/alpha)*C so the C tile is fetched while the first operand slabs are
  // in flight and the epilogue is store-only (result = alpha*acc).  acc[i][j][r] holds
  // C(m = m0+wm*64+i*16+(lane&15), n = n0+wn*64+j*16+(lane>>4)+4r).
  d4_t acc[NF][NF];
  const bool use_c = (g.beta != 0.0) && (g.alpha != 0.0);
  if (use_c) {
    const double cscale = g.beta / g.alpha;
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * WT + j * 16 + fk + 4 * r;
        const int nc = (g.cmapC && n < g.N) ? g.cmapC[n] : n;
        const double* cp = g.C + (size_t)nc * g.ldc;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          const int m = m0 + wm * WT + i * 16 + fm;
          acc[i][j][r] = (n < g.N && m < g.M) ? cscale * cp[m] : 0.0;
        }
      }
  } else {
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int i = 0; i < NF; ++i) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
  }

  double ra[NL], rb[NL];
  int kca[NL], kcb[NL];
#pragma unroll
  for (int p = 0; p < NL; ++p) { kca[p] = 0; kcb[p] = 0; }
  const int nk = (g.K + BK - 1) / BK;
  if (GATHER) { load_map<WT>(kca, g.kmapA, 0, g.K, tid); load_map<WT>(kcb, g.kmapB, 0, g.K, tid); }
  load_slab<A_KC, WT, GATHER>(ra, g.A, g.lda, m0, 0, g.M, g.K, tid, kca);
  load_slab<B_KC, WT, GATHER>(rb, g.B, g.ldb, n0, 0, g.N, g.K, tid, kcb);
  if (GATHER && nk > 1) { load_map<WT>(kca, g.kmapA, BK, g.K, tid); load_map<WT>(kcb, g.kmapB, BK, g.K, tid); }
  store_slab<A_KC, WT>(ra, smem, tid);
  store_slab<B_KC, WT>(rb, smem + OPER_DOUBLES, tid);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_slab<A_KC, WT, GATHER>(ra, g.A, g.lda, m0, (kt + 1) * BK, g.M, g.K, tid, kca);
      load_slab<B_KC, WT, GATHER>(rb, g.B, g.ldb, n0, (kt + 1) * BK, g.N, g.K, tid, kcb);
      if (GATHER && kt + 2 < nk) {  // maps of the slab after next: in flight during this slab's MFMAs
        load_map<WT>(kca, g.kmapA, (kt + 2) * BK, g.K, tid);
        load_map<WT>(kcb, g.kmapB, (kt + 2) * BK, g.K, tid);
      }
    }
    const double* as = smem + cur * 2 * OPER_DOUBLES;
    const double* bs = as + OPER_DOUBLES;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double fa[NF], fb[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) fa[i] = frag<A_KC, WT>(as, wm * WT + i * 16 + fm, ks * 4 + fk);
#pragma unroll
      for (int j = 0; j < NF; ++j) fb[j] = frag<B_KC, WT>(bs, wn * WT + j * 16 + fm, ks * 4 + fk);
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < NF; ++i)
          // D[x][y] = sum_k opB(k, n=x) * opA(m=y, k)  ->  lane&15 <-> m (contiguous in C)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_slab<A_KC, WT>(ra, smem + (cur ^ 1) * 2 * OPER_DOUBLES, tid);
      store_slab<B_KC, WT>(rb, smem + (cur ^ 1) * 2 * OPER_DOUBLES + OPER_DOUBLES, tid);
    }
    __syncthreads();
  }

  // epilogue (store-only unless alpha == 0)
  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int j = 0; j < NF; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * WT + j * 16 + fk + 4 * r;
      if (n >= g.N) continue;
      const int nc = g.cmapC ? g.cmapC[n] : n;
      double* cp = g.C + (size_t)nc * g.ldc;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int m = m0 + wm * WT + i * 16 + fm;
        if (m < g.M) {
          double v = alpha * acc[i][j][r];
          if (alpha == 0.0 && beta != 0.0) v = beta * cp[m];
          cp[m] = v;
        }
      }
    }
  }
}


// =================================================================================================
// gemm2: LDS-DMA ring version of the same 128x128 tile (the default for every shape it supports).
//
// Why: with register staging the next K-slab is requested only one slab ahead; under load a slab arrives
// later than one slab of MFMA time, so every slab stalled (49 % of the fp64 MFMA peak on the K = 256 trailing
// update).  Here the operand slabs go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no
// ds_write) into a ring of S = 4 stages of BK = 8, three slabs in flight across the barrier (counted vmcnt,
// raw s_barrier), and the C tile moves in 16-byte accesses:
//   * "MC" operands (contiguous along the tile's m/n index): image [k][128], one 1-KiB row per wave
//     instruction; fragments of two adjacent rows come from ONE ds_read_b128 (conflict-free unpadded), the
//     two rows go to MFMA tiles 2p and 2p+1, so acc[2p] and acc[2p+1] hold adjacent rows of C.
//   * "KC" operands (contiguous along k): image [m][8] with the 16-byte chunks XOR-swizzled by (m>>2)&3 on
//     the SOURCE address (the DMA destination stays lane-linear); fragments by ds_read_b64, conflict-free.
//   * rows/columns beyond M/N read clamped (valid) addresses and are never stored; k >= K reads a zero page.
// =================================================================================================
constexpr int G2_BK = 8;
constexpr int G2_STAGE = 2 * G2_BK * 128;  // doubles per ring stage (A image | B image)

__device__ double g_zero_page[128];        // 1 KiB of zeros: source of every k >= K slab row

// One LDS-DMA wave instruction: lane l copies 16 bytes from its own global address to LDS byte lds_off + 16*l.
// Written as inline asm on purpose: for the builtin form hipcc (ROCm 7.2) makes every later ds_read wait
// vmcnt(0) for ALL outstanding LDS-DMA (no alias information), which drains the ring once per slab; with asm
// the compiler sees no LDS-DMA and the ring is ordered by the counted s_waitcnt vmcnt + s_barrier below.
// M0 is a reserved register (hipcc sets it immediately before each of its own uses and keeps no value in it
// across statements), so writing it here needs -- and accepts -- no clobber entry.
__device__ __forceinline__ void glds16(const double* gp, double* lp) {
  const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lp;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(lds_off) : "memory");
}

// Issue this wave's share (2 wave-instructions) of one operand slab [k0, k0+8).
//   MC: P(x, k) = P[x + col(k)*ld]; wave w loads k-rows 2w, 2w+1; lane l carries x = x0 + 2l, 2l+1
//   KC: P(x, k) = P[k + x*ld];      wave w loads row blocks 2w, 2w+1 (16 rows each); lane l carries row
//       16q + (l>>2), chunk (l&3) ^ ((l>>4)&3)
template <bool KC>
__device__ __forceinline__ void g2_issue(const double* __restrict__ P, int ld, int x0, int Xmax, int k0, int K,
                                         double* img, int wave, int lane, int col0, int col1) {
  if (!KC) {
    int x = x0 + 2 * lane;
    if (x + 1 >= Xmax) x = (Xmax - 1) & ~1;   // clamp to the last aligned pair that starts inside the matrix
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kr = 2 * wave + q;
      const int k = k0 + kr;
      const int col = q ? col1 : col0;        // column of P that holds k-index k (k itself without a gather map)
      const double* src = (k < K) ? P + (size_t)x + (size_t)col * ld : g_zero_page + 2 * lane;
      glds16(src, img + kr * 128);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int blk = 2 * wave + q;
      int x = x0 + blk * 16 + (lane >> 2);
      if (x >= Xmax) x = Xmax - 1;
      const int ch = (lane & 3) ^ ((lane >> 4) & 3);
      const int k = k0 + 2 * ch;
      const double* src = (k < K) ? P + (size_t)k + (size_t)x * ld : g_zero_page + 2 * lane;
      glds16(src, img + blk * 128);
    }
  }
}

// gather maps: columns of this wave's two k-rows of slab k0.  k is wave-uniform, so these are scalar loads --
// written as asm because hipcc emits vector loads for them, which would sit in the same vmcnt queue as the
// LDS-DMA ring and drain it.  The values are valid only after the s_waitcnt lgkmcnt(0) of g2_cols_wait().
__device__ __forceinline__ int g2_sload(const int* p, int idx) {
  int v;
  asm volatile("s_load_dword %0, %1, %2" : "=s"(v) : "s"(p), "s"(idx * 4) : "memory");
  return v;
}
template <bool MAP>
__device__ __forceinline__ void g2_cols(const int* __restrict__ kmap, int k0, int K, int wave, int& c0, int& c1) {
  const int k = k0 + 2 * wave;
  if (MAP) {
    c0 = 0; c1 = 0;
    if (k < K) c0 = g2_sload(kmap, k);
    if (k + 1 < K) c1 = g2_sload(kmap, k + 1);
  } else {
    c0 = k; c1 = k + 1;
  }
}

template <bool A_KC, bool B_KC, bool GATHER>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int S = 4;                    // ring stages
  constexpr int GI = 4;                   // LDS-DMA instructions per wave per slab (2 per operand)
  g.A += (long)blockIdx.y * g.sA + (long)blockIdx.z * g.sA2;
  g.B += (long)blockIdx.y * g.sB + (long)blockIdx.z * g.sB2;
  g.C += (long)blockIdx.y * g.sC + (long)blockIdx.z * g.sC2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;

  // Tile order (speed only).  Workgroups are dealt round-robin over the 8 XCDs (bid % 8 labels the XCD) and an
  // XCD runs ~64 of them at a time, so each XCD walks its own sequence in blocks of GB x GB tiles: the
  // 2*GB operand panels of a block are then shared by GB workgroups through that XCD's L2 instead of every
  // workgroup streaming its own A panel from beyond L2.
  //   full mode: row groups of 8 tile rows, columns inside a group, contiguous share of that list per XCD
  //   tri mode : GB x GB blocks of the upper block triangle (column-major), dealt round-robin to the XCDs
  const int tiles_m = (g.M + 127) / 128;
  const int tiles_n = (g.N + 127) / 128;
  int tm, tn;
  if (g.tri_mode == 1) {
    const int GB = g.tri_gb;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int blk = (seq / (GB * GB)) * 8 + xcd;        // block number in the column-major upper block triangle
    const int w = seq % (GB * GB);
    // blk = bj*(bj+1)/2 + bi, bi <= bj
    int bj = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
    while ((bj + 1) * (bj + 2) / 2 <= blk) ++bj;
    while (bj * (bj + 1) / 2 > blk) --bj;
    const int bi = blk - bj * (bj + 1) / 2;
    tm = bi * GB + (w % GB);
    tn = bj * GB + (w / GB);
    if (tm >= tiles_m || tn >= tiles_n) return;
  } else if (g.tri_mode == 2 && g.nrg > 0) {
    // only the active region was launched: same order as the full mode below (row groups, columns inside a group, a
    // contiguous share of the list per XCD), but a group's columns start where its first tile row reaches the global
    // diagonal.  The full rectangle's list is empty in its lower half, so its contiguous shares left the XCDs that got
    // the bottom row groups idle (24 TFLOP/s per rank on a 2 x 4 grid at N = 32768 against 60 on one GPU).
    int o = blockIdx.x;
    {
      const int ntiles = (int)gridDim.x;
      const int q = ntiles >> 3, r = ntiles & 7, xcd = o & 7;
      o = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (o >> 3);
    }
    // the row group that holds position o of the list (uniform scalar loop, <= 32 short iterations)
    int rg = 0, st0 = 0, c0 = 0;
    for (int q = 0; q < g.nrg; ++q) {
      const int c = tri2_c0(g, q, tiles_n);
      const int rws = (tiles_m - q * g.rg_h < g.rg_h) ? tiles_m - q * g.rg_h : g.rg_h;
      const int cnt = rws * (g.rg_c1 - c);
      rg = q; c0 = c;
      if (o < st0 + cnt) break;
      st0 += cnt;
    }
    const int rem = o - st0;
    const int rows = (tiles_m - rg * g.rg_h < g.rg_h) ? tiles_m - rg * g.rg_h : g.rg_h;
    const int tc = rem / rows;
    tn = c0 + tc;
    tm = rg * g.rg_h + rem - tc * rows;
  } else {
    const int ntiles = tiles_m * tiles_n;
    int o = blockIdx.x;
    {  // contiguous share of the ordered list per XCD (bijective for any ntiles)
      const int q = ntiles >> 3, r = ntiles & 7, xcd = o & 7;
      o = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (o >> 3);
    }
    const int rg = o / (8 * tiles_n);
    const int rem = o - rg * 8 * tiles_n;
    const int rows = (tiles_m - rg * 8 < 8) ? tiles_m - rg * 8 : 8;
    tn = rem / rows;
    tm = rg * 8 + rem - tn * rows;
  }
  const int m0 = tm * 128, n0 = tn * 128;
  if (g.tri_mode != 0) {
    if (g.ownP > 1 && (tn % g.ownP) != g.ownp) return;
    if (tn < g.tn_lo || tn >= g.tn_hi) return;
    const long grow_min = (long)m0 * g.Px + g.px;
    const int jmax = (n0 + 127 < g.N - 1) ? n0 + 127 : g.N - 1;
    const long gcol_max = (long)jmax * g.Py + g.py;
    if (grow_min > gcol_max) return;
  }

  const int fm = lane & 15, fk = lane >> 4;
  const int nk = (g.K + G2_BK - 1) / G2_BK;

  // tile-local column of MFMA tile j at accumulator row rho (rows: (i>>1)*32 + 2*fm + (i&1) for MC A, i*16 + fm for KC A)
  auto ncol = [&](int j, int rho) { return B_KC ? j * 16 + rho : (j >> 1) * 32 + 2 * rho + (j & 1); };

  d4_t acc[4][4];
  const bool use_c = (g.beta != 0.0) && (g.alpha != 0.0);
  if (use_c) {
    const double cscale = g.beta / g.alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + ncol(j, fk + 4 * r);
        const double* cp = g.C + (size_t)n * g.ldc;
        if (!A_KC) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int m = m0 + wm * 64 + p * 32 + 2 * fm;
            double2 v = make_double2(0.0, 0.0);
            if (n < g.N) {
              if (m + 1 < g.M) {
                // trailing update: every C tile is read once and written once per launch -- streamed past L2 so that it
                // does not evict the operand panels the tile block shares there
                if (g.c_stream) { const d2s_t t = __builtin_nontemporal_load(reinterpret_cast<const d2s_t*>(cp + m)); v = make_double2(t.x, t.y); }
                else v = *reinterpret_cast<const double2*>(cp + m);
              } else if (m < g.M) v.x = cp[m];
            }
            acc[2 * p][j][r] = cscale * v.x;
            acc[2 * p + 1][j][r] = cscale * v.y;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + fm;
            acc[i][j][r] = (n < g.N && m < g.M) ? cscale * cp[m] : 0.0;
          }
        }
      }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
  }

  // ---- prologue: the C tile is requested first (above), then S-1 operand slabs; the first counted wait
  // below therefore covers exactly "C tile + slab 0"
  int ca0, ca1, cb0, cb1;   // columns of the slab issued next (gather variant: fetched one iteration ahead)
#pragma unroll
  for (int s = 0; s < S - 1; ++s) {
    double* st = smem + s * G2_STAGE;
    g2_cols<GATHER && !A_KC>(g.kmapA, s * G2_BK, g.K, wave, ca0, ca1);
    g2_cols<GATHER && !B_KC>(g.kmapB, s * G2_BK, g.K, wave, cb0, cb1);
    if (GATHER) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ca0), "+s"(ca1), "+s"(cb0), "+s"(cb1)::"memory");
    g2_issue<A_KC>(g.A, g.lda, m0, g.M, s * G2_BK, g.K, st, wave, lane, ca0, ca1);
    g2_issue<B_KC>(g.B, g.ldb, n0, g.N, s * G2_BK, g.K, st + G2_BK * 128, wave, lane, cb0, cb1);
  }
  g2_cols<GATHER && !A_KC>(g.kmapA, (S - 1) * G2_BK, g.K, wave, ca0, ca1);
  g2_cols<GATHER && !B_KC>(g.kmapB, (S - 1) * G2_BK, g.K, wave, cb0, cb1);

  for (int kt = 0; kt < nk; ++kt) {
    // my DMAs of slab kt have landed (the 2 younger slabs may still be in flight); after the barrier
    // everybody's have, and everybody has finished reading slab kt-1, whose stage is refilled next
    if (GATHER)  // also: the map entries fetched during the previous iteration are in their SGPRs now
      asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)\n\ts_barrier"
                   : "+s"(ca0), "+s"(ca1), "+s"(cb0), "+s"(cb1) : "n"((S - 2) * GI) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((S - 2) * GI) : "memory");
    {
      const int ks = kt + S - 1;
      double* st = smem + (ks % S) * G2_STAGE;
      g2_issue<A_KC>(g.A, g.lda, m0, g.M, ks * G2_BK, g.K, st, wave, lane, ca0, ca1);
      g2_issue<B_KC>(g.B, g.ldb, n0, g.N, ks * G2_BK, g.K, st + G2_BK * 128, wave, lane, cb0, cb1);
      g2_cols<GATHER && !A_KC>(g.kmapA, (ks + 1) * G2_BK, g.K, wave, ca0, ca1);
      g2_cols<GATHER && !B_KC>(g.kmapB, (ks + 1) * G2_BK, g.K, wave, cb0, cb1);
    }
    const double* as = smem + (kt % S) * G2_STAGE;
    const double* bs = as + G2_BK * 128;
#pragma unroll
    for (int ks = 0; ks < G2_BK / 4; ++ks) {
      const int k = ks * 4 + fk;
      double fa[4], fb[4];
      if (!A_KC) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const double2 v = *reinterpret_cast<const double2*>(as + k * 128 + wm * 64 + p * 32 + 2 * fm);
          fa[2 * p] = v.x; fa[2 * p + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = wm * 64 + i * 16 + fm;
          fa[i] = as[m * 8 + (((k >> 1) ^ ((m >> 2) & 3)) << 1) + (k & 1)];
        }
      }
      if (!B_KC) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const double2 v = *reinterpret_cast<const double2*>(bs + k * 128 + wn * 64 + p * 32 + 2 * fm);
          fb[2 * p] = v.x; fb[2 * p + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = wn * 64 + j * 16 + fm;
          fb[j] = bs[n * 8 + (((k >> 1) ^ ((n >> 2) & 3)) << 1) + (k & 1)];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }
  // drain the (zero-page) DMAs still in flight before the LDS allocation is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const double alpha = g.alpha;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 64 + ncol(j, fk + 4 * r);
      if (n >= g.N) continue;
      double* cp = g.C + (size_t)n * g.ldc;
      if (!A_KC) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int m = m0 + wm * 64 + p * 32 + 2 * fm;
          const double2 v = make_double2(alpha * acc[2 * p][j][r], alpha * acc[2 * p + 1][j][r]);
          if (m + 1 < g.M) {
            if (g.c_stream) { d2s_t t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<d2s_t*>(cp + m)); }
            else *reinterpret_cast<double2*>(cp + m) = v;
          } else if (m < g.M) cp[m] = v.x;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + wm * 64 + i * 16 + fm;
          if (m < g.M) cp[m] = alpha * acc[i][j][r];
        }
      }
    }
  }
}

int g_gemm_cstream = 0;   // tuning hook (eigx_tune key 6): stream the C tiles of the trailing update past L2
int g_gemm_variant = 2;   // 2 = gemm2 where supported and worthwhile, 3 = wherever supported (tests), 1 = never

// gemm2 needs 16-byte aligned operand columns (even leading dimensions, aligned bases, even batch strides);
// KC operands additionally need an even K (a 16-byte chunk must not straddle K)
static bool gemm2_ok(const GemmArgs& g, bool a_kc, bool b_kc, int batch, int batch2) {
  if (g_gemm_variant < 2) return false;
  if (g.M < 2 || g.N < 2 || g.K < 1 || g.alpha == 0.0 || g.cmapC) return false;
  if ((g.lda | g.ldb | g.ldc) & 1) return false;
  if (((uintptr_t)g.A | (uintptr_t)g.B | (uintptr_t)g.C) & 15) return false;
  if (batch > 1 && ((g.sA | g.sB | g.sC) & 1)) return false;
  if (batch2 > 1 && ((g.sA2 | g.sB2 | g.sC2) & 1)) return false;
  if ((a_kc || b_kc) && (g.K & 1)) return false;
  if (a_kc && g.lda < g.K) return false;
  if (b_kc && g.ldb < g.K) return false;
  return true;
}

}  // namespace

int set_gemm_variant(int v) { const int old = g_gemm_variant; g_gemm_variant = v; return old; }
int set_gemm_cstream(int v) { const int old = g_gemm_cstream; g_gemm_cstream = v; return old; }

void dgemm_dev(hipStream_t stream, char opA, char opB, int M, int N, int K, double alpha, const double* A,
               int lda, const double* B, int ldb, double beta, double* C, int ldc, int tri_mode,
               const Grid* grid, const int* kmapA, const int* cmapC, int batch, long strideA, long strideB,
               long strideC, int batch2, long strideA2, long strideB2, long strideC2, int ownP, int ownp,
               const int* kmapB, int tn_lo, int tn_hi) {
  if (M <= 0 || N <= 0 || batch <= 0 || batch2 <= 0) return;
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.tri_mode = tri_mode;
  g.kmapA = (opA == 'N' || opA == 'n') ? kmapA : nullptr;
  g.cmapC = cmapC;
  g.kmapB = (opB == 'T' || opB == 't') ? kmapB : nullptr;
  g.sA = strideA; g.sB = strideB; g.sC = strideC;
  g.sA2 = strideA2; g.sB2 = strideB2; g.sC2 = strideC2;
  g.ownP = ownP; g.ownp = ownp; g.tri_gb = 1;
  g.tn_lo = tn_lo; g.tn_hi = tn_hi;
  g.nrg = 0; g.rg_h = 8; g.rg_c1 = 0; g.rg_magic = 0; g.btab = nullptr;
  g.c_stream = (tri_mode != 0 && g_gemm_cstream) ? 1 : 0;
  g.Px = grid ? grid->Px : 1; g.px = grid ? grid->px : 0;
  g.Py = grid ? grid->Py : 1; g.py = grid ? grid->py : 0;
  const bool a_kc = (opA == 'T' || opA == 't');   // op(A)(m,k) = A[k + m*lda]
  const bool b_kc = (opB == 'N' || opB == 'n');   // op(B)(k,n) = B[k + n*ldb]
  const bool gather = g.kmapA || g.kmapB;
  if ((!gather || (g.kmapA && g.kmapB && !a_kc && !b_kc)) && gemm2_ok(g, a_kc, b_kc, batch, batch2)) {
    const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128) * batch * batch2;
    if (tri_mode != 0 || t128 >= 192 || g_gemm_variant == 3) {
      int gx2 = ceil_div(M, 128) * ceil_div(N, 128);
      if (tri_mode == 1) {
        // block-triangular order: needs a square tile grid; GB x GB tile blocks, dealt to the 8 XCDs
        const int t = ceil_div(N > M ? N : M, 128);
        g.tri_gb = (t >= 96) ? 8 : (t >= 24 ? 4 : (t >= 8 ? 2 : 1));
        const int nb = ceil_div(t, g.tri_gb);
        const int nblk = nb * (nb + 1) / 2;
        gx2 = 8 * ceil_div(nblk, 8) * g.tri_gb * g.tri_gb;
      }
      if (tri_mode == 2 && batch == 1 && batch2 == 1 && ownP == 1) {
        // active region of the local block: tile (tm, tn) holds an element on or above the global diagonal iff the first
        // global row of tile row tm <= the last global column of tile column tn; the window [tn_lo, tn_hi) cuts columns
        const int tiles_m = ceil_div(M, 128), tiles_n = ceil_div(N, 128);
        g.rg_h = (tiles_m <= 256) ? 8 : ceil_div(tiles_m, 32);
        g.nrg = ceil_div(tiles_m, g.rg_h);
        g.rg_c1 = tiles_n < tn_hi ? tiles_n : tn_hi;
        g.rg_magic = (unsigned)(((1ull << 32) / (unsigned long long)g.Py) + 1ull);
        int total = 0;
        for (int q = 0; q < g.nrg; ++q) {
          const int rows = (tiles_m - q * g.rg_h < g.rg_h) ? tiles_m - q * g.rg_h : g.rg_h;
          total += rows * (g.rg_c1 - tri2_c0(g, q, tiles_n));
        }
        if (total == 0) return;
        gx2 = total;
      }
      dim3 grd2(gx2, batch, batch2), blk2(256);
      const size_t shmem2 = (size_t)4 * G2_STAGE * sizeof(double);
#define EIGX_LAUNCH2(AK, BK_, GA_)                                                                 \
  do {                                                                                             \
    static bool attr_set2 = false;                                                                 \
    if (!attr_set2) {                                                                              \
      EIGX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm2_kernel<AK, BK_, GA_>,                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem2)); \
      attr_set2 = true;                                                                            \
    }                                                                                              \
    hipLaunchKernelGGL((gemm2_kernel<AK, BK_, GA_>), grd2, blk2, shmem2, stream, g);               \
  } while (0)
      if (gather) EIGX_LAUNCH2(false, false, true);
      else if (a_kc && b_kc) EIGX_LAUNCH2(true, true, false);
      else if (a_kc) EIGX_LAUNCH2(true, false, false);
      else if (b_kc) EIGX_LAUNCH2(false, true, false);
      else EIGX_LAUNCH2(false, false, false);
#undef EIGX_LAUNCH2
      EIGX_HIP_CHECK(hipGetLastError());
      return;
    }
  }
  // 64x64 tiles when 128x128 tiles would leave most of the 256 CUs idle
  const long tiles128 = (long)ceil_div(M, 128) * ceil_div(N, 128) * batch * batch2;
  const bool small = (tri_mode == 0) && tiles128 < 192;
  const int bm = small ? 64 : 128;
  const int tiles = ceil_div(M, bm) * ceil_div(N, bm);
  dim3 grd(tiles, batch, batch2), blk(256);
#define EIGX_LAUNCH(AK, BK_, WT_, GA_)                                                             \
  do {                                                                                             \
    const size_t shmem = (size_t)4 * Geo<WT_>::OPER * sizeof(double);                              \
    static bool attr_set = false;                                                                  \
    if (!attr_set) {                                                                               \
      EIGX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f64_kernel<AK, BK_, WT_, GA_>,          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
      attr_set = true;                                                                             \
    }                                                                                              \
    hipLaunchKernelGGL((gemm_f64_kernel<AK, BK_, WT_, GA_>), grd, blk, shmem, stream, g);          \
  } while (0)
#define EIGX_LAUNCH_T(AK, BK_) do { if (small) EIGX_LAUNCH(AK, BK_, 32, false); else EIGX_LAUNCH(AK, BK_, 64, false); } while (0)
  if (g.kmapA || g.kmapB) {
    // gather variant: only the D&C product Q(:, map) * S^T ('N','T') uses it; both maps are required
    if (a_kc || b_kc || !g.kmapA || !g.kmapB) {
      fprintf(stderr, "[eigx] dgemm gather needs opA='N', opB='T' and both maps\n");
      abort();
    }
    if (small) EIGX_LAUNCH(false, false, 32, true); else EIGX_LAUNCH(false, false, 64, true);
  } else if (a_kc && b_kc) EIGX_LAUNCH_T(true, true);
  else if (a_kc && !b_kc) EIGX_LAUNCH_T(true, false);
  else if (!a_kc && b_kc) EIGX_LAUNCH_T(false, true);
  else EIGX_LAUNCH_T(false, false);
#undef EIGX_LAUNCH_T
#undef EIGX_LAUNCH
  EIGX_HIP_CHECK(hipGetLastError());
}

// One launch for a table of gather products of different shapes (64 x 64 tiles; the D&C's low heights, where a height is
// dozens of products of a few tiles each and the per-product launches were the cost).  Role of the PDGEMM calls of
// src/my_pdlaed1.F:310-341 for all merges of one tree level at once.
void dgemm_gather_batch_dev(hipStream_t stream, const GemmBatch* tab_dev, int nbatch, int maxM, int maxN,
                            const double* A, int lda, const double* B, int ldb, double* C, int ldc,
                            const int* kmapA, const int* kmapB) {
  if (nbatch <= 0 || maxM <= 0 || maxN <= 0) return;
  GemmArgs g;
  g.M = maxM; g.N = maxN; g.K = 0; g.alpha = 1.0; g.beta = 0.0;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.tri_mode = 0; g.kmapA = kmapA; g.cmapC = nullptr; g.kmapB = kmapB;
  g.sA = g.sB = g.sC = 0; g.sA2 = g.sB2 = g.sC2 = 0;
  g.ownP = 1; g.ownp = 0; g.tri_gb = 1; g.tn_lo = 0; g.tn_hi = 0x7fffffff;
  g.nrg = 0; g.rg_h = 8; g.rg_c1 = 0; g.rg_magic = 0; g.c_stream = 0;
  g.Px = 1; g.px = 0; g.Py = 1; g.py = 0;
  g.btab = tab_dev;
  constexpr int WT = 32;
  const int tiles = ceil_div(maxM, Geo<WT>::BM) * ceil_div(maxN, Geo<WT>::BN);
  const size_t shmem = (size_t)4 * Geo<WT>::OPER * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    EIGX_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f64_kernel<false, false, WT, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_f64_kernel<false, false, WT, true>), dim3(tiles, nbatch, 1), dim3(256), shmem, stream, g);
  EIGX_HIP_CHECK(hipGetLastError());
}

}  // namespace eigx
