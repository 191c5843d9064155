// band_reduce.hip -- Householder reduction full -> tridiagonal (NB=1) / pentadiagonal (NB=2), gfx950.
//
// Replaces (reference paths relative to RIKEN-RCCS/EigenExa 2.13):
//   eigen_trd_body   src/eigen_trd.F:349-723   + eigen_trd_au (fused SYMV)  src/eigen_trd_t2.F:161-651,:970-1400
//   eigen_prd_body   src/eigen_prd.F:341-578   + eigen_prd_au (2-vector SYMV) src/eigen_prd_t2.F:90-214,:290-958
//   compute_u        src/eigen_trd_t4.F:81-189, src/eigen_prd_t4x.F:83-373
//   compute_v        src/eigen_trd_t6_3.F:85-330, src/eigen_prd_t6_3.F:76-477
//   local_2update    src/eigen_trd_t5.F:71-137,  src/eigen_prd_t5.F:69-266
//   panel load/store src/eigen_trd_t7.F:72-267,  src/eigen_prd_t7.F:74-250
//   trailing update  eigen_common_2update src/eigen_t1.F:68-309  (here: the MFMA GEMM of gemm_f64.hip)
//
// Same mathematics as the reference (bottom-up, column i annihilated above the band, reflector
// u = x - s e_L, s = -sign(||x||,x_L), beta = -u_L s, stored in column i of `a`; delayed rank-2k update
// per panel of m columns), re-designed for one GPU stream instead of MPI ranks x OpenMP threads:
//
//   per column (NB=1) / column pair (NB=2) a chain of short kernels, all reductions two-phase and
//   deterministic (per-workgroup partials written to HBM, re-reduced redundantly by every consumer
//   workgroup in a fixed order -> bit-identical replicas, the property the reference's hand-written
//   allreduce exists for), no atomics, no host synchronisation inside the reduction:
//     K_A  finish the previous step's W = P T - U M/2 from the SYMV partials, append it to the panel,
//          and form the next effective column(s) x = a_i - U W(i,:)^T - W U(i,:)^T  (lazy panel update),
//          Gram partials of x
//     K_M  (NB=2) apply the first reflector to the second column, partial norm
//     K_B  HBM-bound fused symmetric mat-vec over the upper triangle only: one pass over A gives
//          both  y_row += a(r,c) u(c)  and  y_col += a(r,c) u(r)  for NB vectors at once (K2/K3 of
//          SURVEY.md 2.3); 16-byte coalesced loads down the columns, column sums reduced with a
//          halving butterfly of wave shuffles, partial results per (row strip, column segment)
//     K_P  panel dot products U^T u, W^T u (tall-skinny), reflector store into `a` and the panel
//   per panel: A(0:nr,0:nr) -= [U W][W U]^T on upper-triangle tiles with the fp64 MFMA GEMM.
#include "eigx_context.h"
#include "eigx_comm.h"
#include "../../include/eigenexa_amd.h"
#include <chrono>
#include <cstring>

// Floating-point contraction by SOURCE FORM only (a * b + c written in one expression becomes an fma, nothing is fused
// across statements): the same source then rounds the same way in every template instantiation and role of these kernels.
// With the default "fast" mode the compiler decides per instantiation; after the round-4 refactor of the kernels into role
// bodies the two load forms of the mat-vec (UNC) stopped being bit-identical (test_symv_load_forms_are_bit_identical), and
// the replicas of the multi-GPU path rely on identical rounding on every rank.
#pragma clang fp contract(on)

namespace eigx {

namespace {

// rows per panel-dot chunk: at most 4 chunks (K_A re-reduces them in one batch of loads), at least 512 rows
static inline int pd_rows_for(int toprows) { int r = ((toprows + 3) / 4 + 63) / 64 * 64; return r < 512 ? 512 : r; }
#ifdef EIGX_STAMPS
#define EIGX_ABL(bit) (R.abl & (bit))
// (stamp_me: the workgroup in the middle of its role's range takes the stamps)
#define EIGX_STAMP(slot) do { if (R.dbg && threadIdx.x == 0 && stamp_me) { \
  const unsigned long long _t = __builtin_amdgcn_s_memtime(); atomicAdd(&R.dbg[slot], _t - stamp_prev); stamp_prev = _t; } } while (0)
#define EIGX_STAMP_INIT const bool stamp_me = (bid == nblocks / 2); unsigned long long stamp_prev = __builtin_amdgcn_s_memtime(); (void)stamp_me;
#else
#define EIGX_ABL(bit) false
#define EIGX_STAMP(slot) do {} while (0)
#define EIGX_STAMP_INIT
#endif
constexpr int PD_COLS = 16;    // panel columns per panel-dot workgroup
constexpr int KA_ROWS = 16;    // rows per K_A workgroup
constexpr int KA_SL = 256 / KA_ROWS;  // panel slices per row (KA_ROWS x KA_SL = 256 threads)

// scalar slots in the small device array `sc`
enum { SC_SA = 0, SC_BETA_A = 1, SC_SB = 2, SC_BETA_B = 3, SC_COUNT = 8 };

struct RedArgs {
  double* A; int lda; int n;
  double* UW; int ldp; int m;     // [U | W | U], m columns each
  double* X;                      // ldp x 3 : slot 0 = x_i, slot 1 = x_{i-1}, slot 2 = H_A x_{i-1} (K_M output)
  double* YR; double* YC;         // SYMV partials [tile col | tile row][a][ldp]
  double* KD;                     // panel-dot partials [chunk][kind 2*NB][m], then [chunk] uA.uB at the end
  double* SP;                     // [wg][3] bilinear partials
  double* GP;                     // [K_A workgroup][3] Gram partials of the new columns
  double* sc;                     // scalars
  double* d; double* e; int lde;
  int maxseg, maxrs, maxchunk, gp2_off, kdab_off;
  // multi-GPU (P > 1), 2-D cyclic like the reference (src/eigen_libs0.F:1825-2258): A is this rank's LOCAL block
  // a(lda, *), local (li, lj) = global (li*Px + px, lj*Py + py); nothing of A is replicated.  The per-row panel work of a
  // step (finish W, form the next x: ka_kernel; panel dot products: K_P) is DISTRIBUTED: global rows are dealt to the
  // ranks in groups of KA_ROWS = 16, row r belongs to rank (r / 16) % P at owned index ((r / 16) / P) * 16 + r % 16 --
  // the reference forms v, u and the panel update for a rank's rows only as well (src/eigen_prd_t6_3.F:76-477,
  // src/eigen_prd_t4x.F:215-224, src/eigen_prd_t5.F:69-266).  Two exchanges per step through peer windows (comm.hip):
  //   Y  after the local mat-vec: kl_kernel sums the rank's tile partial sums and stores the sum of local row li /
  //      column lj into the window of the row's OWNER only (rows of the next block columns: into everybody's), plus the
  //      rank's share of the panel dots and 3 bilinear scalars into everybody's;
  //   X  ka_kernel, for its own rows: adds the Py + Px contributions of a row in a fixed order, finishes W, forms the next
  //      x, and stores both into everybody's window, plus the rank's Gram partial sums.
  // They replace the reference's allreduces over X and Y and the row->column transpose per column
  // (src/eigen_prd_t2.F:179,:203, src/comm.F:1377-1528; src/eigen_prd_t6_3.F:174,:296).  What stays replicated is
  // O(L) per step: the reflector store [U | . | U] from x and the copy of the received W rows into the panel.
  int P, me;
  float invP;                     // 1 / P (owner of a row group without an integer division: exact for groups < 2^20)
  int Px, Py, px, py, row_major;
  int nxs, nys;                   // padded maximal local extents = strides inside a Y message
  const double* MSG;              // my Y window: message of source q, parity h at MSG + h*ypar_stride + q*ysrc_stride
  int msg_stride;                 // doubles per Y message: NB*(nxs + nys) + 8 + 2*NB*m
  size_t ypar_stride; int ysrc_stride;
  const double* XW;               // my X window: message of source q, parity h at XW + h*xpar_stride + q*xmsg_stride:
  int xmsg_stride, nown;          //   [x_i | x_{i-1} | W_A | W_B](owned index, stride nown), then 8 doubles: Gram partial sums
  size_t xpar_stride;
  const double* PAN; int ldpan;   // columns of the current panel gathered from their owners (global row order)
  const double* zero16;           // 16 bytes of zeros: where the mat-vec's loads of rows beyond the active block go (see load8)
  int abl;                        // EIGX_STAMPS diagnostic build only: ablation mask (timing experiments)
  int stamp_i;                    // EIGX_STAMPS diagnostic build only: the step (top column i) whose roles record per-workgroup times
  unsigned long long* dbg;        // EIGX_STAMPS diagnostic build only: accumulated s_memtime stamps
};

// SYMV tiling: square tiles of T = 128*RB rows/cols (RB = 1,2,4); tile (ty,tx) with tx >= ty is one
// workgroup.  Small triangles get small tiles so that enough workgroups exist.
struct SymvGeom { int L, T, nt; };

// Tile-size thresholds and the active size above which the matrix is streamed with non-temporal loads
// (eigx_tune keys 3, 4, 5).  History of the A/B runs on one MI355X (tools/gpu_reduce_time.py): the 256 tile + non-temporal
// loads took N=16384 from 972 to 833 ms and N=32768 from 6008 to 5347 ms against the 512 tile; N=8192 loses 1 % with
// non-temporal loads (its 512 MB matrix still profits from the 256 MB Infinity Cache).  Round 2, same box, same call:
// N=32768 4871 ms (512 tile beyond L = 20000) / 4807 (beyond 26000) / 4732 (256 tile throughout) -- the 512 tile's 172
// VGPRs leave two workgroups per CU and its 2080 tiles two "generations" -- so the 512 tile now starts at L = 40000
// (beyond that the 256 tile would give a row more than the 160 partial sums K_A loads in its first batch).  The
// 128 / 256 switch stays at L = 4500: once K_A's load batches were matched to the step (fewer partial sums and tile
// scalars with the larger tile), an A/B on one buffer gave N=8192 133.1 / 133.3 / 134.0 / 134.6 ms (penta) and
// 227.0 / 229.0 / - / 234.1 ms (tri) for a switch at 4500 / 6000 / 7000 / 9000, and 134.5 / 135.9 for 3500 / 3000.  tools/symv_stream.hip (the kernel's load loop alone, same tiles and order)
// reads 6.5-6.9 TB/s; adding the tile's partial-sum stores (1.6 % of the bytes) costs 10-17 % of that on their own,
// and nothing when they stay inside L2.  Walking the triangle tile column by tile column instead
// of row by row changes nothing (A/B on one buffer).
int g_ka_fit = 1;     // K_A (eigx_tune key 10): 1 = load batches matched to the step (launch_ka), 0 = always the largest (A/B)
int g_ka_wgs = 256;   // K_A (eigx_tune key 7): beyond 2 * this many row groups a workgroup takes several of them, ~this many workgroups
int g_symv_t128 = 4500, g_symv_t256 = 40000;
int g_symv_nt = 9000;
// (eigx_tune key 12, once EIGX_FOLD_KL: the Y exchange inside the mat-vec launch -- its last-arriving tiles reduced and pushed.
// In the rehearsal of one rank of a 2 x 4 grid at N = 32768 it made the mat-vec 23 us longer to save a 20-us kernel: the last
// tile's workgroup did a whole row block's and column block's reduction alone, on the critical path.  Removed in round 4.)
int g_symv_unc = 9000;   // the fused mat-vec's branch-free pipelined form up to this active size (eigx_tune key 11)

inline SymvGeom symv_geom(int L) {
  SymvGeom g;
  g.L = L;
  g.T = (L <= g_symv_t128) ? 128 : (L <= g_symv_t256 ? 256 : 512);
  g.nt = (L + g.T - 1) / g.T;
  return g;
}

typedef double d2v_t __attribute__((ext_vector_type(2)));
// 16-byte load of two consecutive doubles; NT = streaming (non-temporal) hint for the once-per-step sweep of a
// triangle that is far larger than L2 + Infinity Cache (A/B on one MI355X, N=32768 reduction: 6.06 -> 5.72 s)
template <bool NT>
__device__ __forceinline__ double2 ld2(const double* p) {
  const d2v_t v = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2v_t*>(p)) : *reinterpret_cast<const d2v_t*>(p);
  return make_double2(v.x, v.y);
}

// v + (v of the lane selected by a DPP control): a VALU data-parallel-primitive move, no LDS crossbar round trip
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return v + __hiloint2double(hi2, lo2);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo2 = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi2, lo2);
}

// gfx950 lane-swap adds (v_permlane32_swap / v_permlane16_swap, VALU): with a' / b' the two registers after the swap,
//   swapadd32(a, b): lanes  0..31 get a[i] + a[i+32],  lanes 32..63 get b[i-32] + b[i]
//   swapadd16(a, b): even rows of 16 lanes get a[row] + a[row+1], odd rows get b[row-1] + b[row]
// i.e. one stage of a halving butterfly on two values, or with a == b the xor-32 / xor-16 stage of an all-reduce.
__device__ __forceinline__ double swapadd32(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swapadd16(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// sum over the 64 lanes, result in every lane, fixed order.  Within a row of 16 lanes by DPP (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror), across the four rows by the gfx950 lane swaps: VALU only,
// instead of 6 dependent ds_bpermute (LDS crossbar) round trips per value on these latency-bound kernels.
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  v = swapadd16(v, v);
  v = swapadd32(v, v);
  return v;
}

// deterministic block sum of K values at once (256 threads, one barrier pair); results in all threads
template <int K>
__device__ __forceinline__ void block_sum_multi(double (&v)[K], double* red /* >= 4*K doubles LDS */) {
#pragma unroll
  for (int q = 0; q < K; ++q) v[q] = wave_sum(v[q]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < K; ++q) red[(threadIdx.x >> 6) * K + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < K; ++q) v[q] = (red[q] + red[K + q]) + (red[2 * K + q] + red[3 * K + q]);
}

__device__ __forceinline__ double sign_of(double mag, double s) { return s >= 0.0 ? fabs(mag) : -fabs(mag); }

// loads / stores of bytes that another rank's kernel writes / reads while this one runs: system scope, cache-bypassing
__device__ __forceinline__ double ld_sys(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// loads / stores of words that another workgroup of the SAME launch reads / wrote: agent scope, past the per-XCD L2s
// (MI355X_MICROARCH.md, hand-off forms: sc1 stores, every storing wave drains, one agent-scope atomic add per workgroup,
// the workgroup whose add came last reads with sc1 loads)
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// =================================================================================================
// K_A : finish previous step (if has_prev) and form the next columns (ncols = 0,1,2).
// Workgroup = 32 rows x 8 slices: slice ks handles panel columns kk = ks (mod 8) and SYMV partials
// t = ks (mod 8) of its row; the 8 slice sums are combined through LDS in a fixed order.
// =================================================================================================
struct KAArgs {
  int has_prev;   // previous step pending: its W columns [kprev, kprev+NB) must be finished
  int iprev;      // top column of the previous step
  int Lprev;      // rows of the previous step's reflectors (uA), uB has Lprev-1
  int kprev;      // panel fill before the previous step
  int nchunk_prev;  // K_P row chunks of the previous step
  int ncols;      // columns to form now (0 = finish only)
  int i;          // top column of the new block (slot 0 <-> column i, slot 1 <-> column i-1)
  int L;          // rows above the new block (Gram sums run over r < L)
  int k;          // panel fill to use for the new columns (= kprev+NB if has_prev)
  int rows;       // rows to cover: max(iprev+1, i+1)
  int nt_prev, lgT_prev;  // SYMV tiling of the previous step: tiles per dimension, log2(tile edge)
  int par;                // multi-GPU: parity of the Y messages that hold the previous step's SYMV partial sums
  int pan_c0;             // multi-GPU: first global column held by the gathered panel R.PAN
  StepWait wait;          // multi-GPU: wait for the Y messages here instead of in a wait kernel (wait.n = 0: no)
  int nchunk_ab;          // multi-GPU: row chunks of the (replicated) reflector store = entries of the uA.uB partial sums
  int xpar;               // multi-GPU: parity of the X message this launch writes
  unsigned long long xepoch;   // ... and its epoch
  int G;                  // row groups (of KA_ROWS rows) per workgroup: the scalar work of a workgroup is done once, then
                          // its G row groups follow in a loop (the next group's loads in flight behind the current one)
};

// multi-GPU: partial number t (0 <= t < Py + Px) of global row r in the step messages of parity `par`:
// t < Py  -> row sums of rank (r % Px, t), which holds row r at local r / Px;
// t >= Py -> column sums of rank (t - Py, r % Py), which holds column r at local r / Py.
// Returns the address for vector 0; vector 1 follows `stride` doubles later.
template <int NB>
__device__ __forceinline__ const double* mg_partial(const RedArgs& R, int par, int t, int r, int& stride) {
  int qx, qy, off;
  if (t < R.Py) { qx = r % R.Px; qy = t; off = r / R.Px; stride = R.nxs; }
  else { qx = t - R.Py; qy = r % R.Py; off = NB * R.nxs + r / R.Py; stride = R.nys; }
  const int src = R.row_major ? qx * R.Py + qy : qx + qy * R.Px;
  return R.MSG + (size_t)par * R.ypar_stride + (size_t)src * R.ysrc_stride + off;
}

// multi-GPU: owner (world rank) and owned index of global row r -- groups of KA_ROWS rows dealt round-robin
__device__ __forceinline__ void own_of(const RedArgs& R, int r, int& src, int& idx) {
  const int g = r >> 4;
  const int q = (int)(((float)g + 0.5f) * R.invP);   // g / P, exact (see RedArgs::invP)
  src = g - q * R.P;
  idx = (q << 4) | (r & 15);
}
static_assert(KA_ROWS == 16, "own_of assumes groups of 16 rows");
// global row of owned index o on rank `me`
__device__ __forceinline__ int own_row(const RedArgs& R, int o) { return (((o >> 4) * R.P + R.me) << 4) | (o & 15); }
// x_i (v = 0) / x_{i-1} (v = 1) at global row r: one GPU from R.X; several GPUs from the owner's X message (parity xpar)
template <bool MG>
__device__ __forceinline__ double xget(const RedArgs& R, int xpar, int v, int r) {
  if (!MG) return R.X[(size_t)v * R.ldp + r];
  int s_, o_;
  own_of(R, r, s_, o_);
  return ld_sys(R.XW + (size_t)xpar * R.xpar_stride + (size_t)s_ * R.xmsg_stride + (size_t)v * R.nown + o_);
}
// bounded spin of a consumer kernel's prologue on the P arrival flags of a step message (lane q of the first wave waits for
// rank q; a time-out sets the sticky error word and the solver reports it), then the workgroup barrier releases the rest.
// The polls are RELAXED system-scope loads: an acquire load invalidates caches at every look (~1.7 us each), and in the
// one-launch-per-step form hundreds of waiting workgroups doing that evicted the working set of the roles still computing
// (N = 32768 on 2 x 4, rehearsal of one rank: 95 us per step with acquire polls, 67 us with relaxed ones; a second level
// of local gate words with one system-scope poller per role gained nothing on top and was dropped).  Every byte of a
// message is read with system-scope loads that bypass the caches, so no acquire is needed behind the flag either.
__device__ __forceinline__ void step_wait_fused(const StepWait& W, bool first_block) {
  const int tid = threadIdx.x;
  if (tid < W.n) {
    const unsigned long long* f = W.flag + (W.epoch & 1) * EIGX_MAXP + tid;
    const long long t0 = wall_clock64();
    if (__hip_atomic_load(W.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < W.epoch) {
        for (int z = 0; z < W.naps; ++z) __builtin_amdgcn_s_sleep(2);   // (~60 ns each)
        if (__hip_atomic_load(W.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;   // a peer failed
        if (wall_clock64() - t0 > W.limit_ticks) {
          __hip_atomic_store(W.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    if (first_block && tid == 0) atomicAdd(W.ticks, (unsigned long long)(wall_clock64() - t0));
  }
  __syncthreads();
}

// (role body: workgroup `bid` of `nblocks`; XPp = where the X message goes, several GPUs only)
template <int NB, bool MG, bool LG, int RPBT, int SPBT, int KBT>
__device__ __forceinline__ void ka_body(const RedArgs& R, const KAArgs& S, const StepPeers* XPp, const int bid, const int nblocks) {
  __shared__ double red[64];
  __shared__ double kd[4][256];        // reduced panel-dot vectors: [UuA, WuA, UuB, WuB][kk]  (m <= 256)
  __shared__ double rowU[2][258], rowW[2][258];  // U(c, kk), W(c, kk) for the new block columns c
  __shared__ double slice[4][KA_ROWS][4];        // per-wave slice sums
  __shared__ int lastw;                          // several GPUs: this workgroup publishes the X message
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int rr = tid & (KA_ROWS - 1), ks = tid / KA_ROWS;
  const int ldp = R.ldp, m = R.m;
  double* Up = R.UW;
  double* Wp = R.UW + (size_t)ldp * m;
  const int nt = S.nt_prev, lgT = S.lgT_prev;      // SYMV tiling of the previous step (T = 1 << lgT)
  constexpr bool mg = MG;   // several GPUs (a template parameter: a run-time branch here costs register copies and waits)
  const bool hp = S.has_prev != 0;
  const int kp = hp ? S.kprev : 0;
  const int kloop = hp ? S.kprev : S.k;
  const int kold = hp ? S.kprev : S.k;  // panel slots that are final in memory
  // The workgroup owns the row groups bid * G + g, g = 0 .. G-1 (KA_ROWS rows each).  Everything that does not
  // depend on the row -- ~85 % of the kernel's instructions: the re-reduction of the tile / panel partial sums, the 2x2
  // algebra, the LDS tables -- is done ONCE per workgroup; with one row group per workgroup the chip ran that overhead
  // 2 (N = 8192) to 8 (N = 32768) times per SIMD, and the PMC counters show the kernel issue-bound there.
  // LG = false: exactly one group per workgroup (the second register set of the loop form disappears: 195 instead of 255
  // VGPRs, two waves per SIMD) -- faster up to 512 groups (N = 8192: 143.7 against 146.8 ms per reduction); LG = true beyond
  // (N = 32768: 5.15 against 5.48 s).
  const int G = LG ? S.G : 1;
  // several GPUs: the workgroup's groups are groups of THIS rank (owned group og <-> global group og * P + me)
  const int r = MG ? ((bid * G) * R.P + R.me) * KA_ROWS + rr : (bid * G) * KA_ROWS + rr;      // row of group 0
  const int rstep = MG ? KA_ROWS * R.P : KA_ROWS;     // rows from a group of the workgroup to its next one
  const StepPeers& XP = *XPp;                         // (dereferenced on the several-GPU paths only)
  EIGX_STAMP_INIT
#ifdef EIGX_STAMPS
#define EIGX_TL(slot) do { if (MG && R.dbg && threadIdx.x == 0 && stamp_me) { const unsigned long long t0_ = __hip_atomic_load(&R.dbg[20], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    const unsigned long long dt_ = (unsigned long long)wall_clock64() - t0_; if (t0_ != 0 && dt_ < 100000ull) atomicAdd(&R.dbg[slot], dt_); } } while (0)
#else
#define EIGX_TL(slot) do {} while (0)
#endif
  EIGX_TL(26);   // (timeline: this workgroup's entry)
#ifdef EIGX_STAMPS
#define EIGX_PW(base) do { if (MG && R.dbg && threadIdx.x == 0 && S.i == R.stamp_i && bid < 1024) R.dbg[(base) + bid] = (unsigned long long)wall_clock64(); } while (0)
#else
#define EIGX_PW(base) do {} while (0)
#endif
  EIGX_PW(64);
#ifdef EIGX_STAMPS
  const unsigned long long clk_w0 = (unsigned long long)wall_clock64(), clk_m0 = __builtin_amdgcn_s_memtime();
#endif

  // ============ phase 0: every load that depends on nothing computed in this kernel ==================
  // The kernel is a latency chain (a few hundred bytes per thread): ALL loads are issued first, in
  // straight-line batches whose extent is cut by wave-uniform conditions (no per-thread branches, no integer
  // divisions, clamped indices; out-of-range entries are dropped by a select afterwards), and nothing is
  // consumed before the last load is issued: the chain costs about one memory round trip.
  // The first batches are unconditional loads (clamped addresses), so their sizes are template parameters that the
  // host matches to the step (launch_ka): every size is correct for every step -- what a batch does not cover goes
  // through the remainder loops below -- a matched one just carries fewer dummy loads (N=8192 reduction 141.1 -> 137.5 ms
  // with the partial-sum batch alone).
  constexpr int KB = KBT;               // panel columns per slice in the first batch (KB * KA_SL columns: 2 / 4 / 8 -> 32 / 64 / 128)
  // SYMV partials per slice in the first batch: 10 (10 * KA_SL = 160 slots) or 5 (80 slots: every step of N <= 10000,
  // half the unconditional loads of that phase); several GPUs: one message entry per slice
  constexpr int RPB = MG ? 1 : RPBT;
  constexpr int CHB = MG ? EIGX_MAXP : 4;   // K_P row chunks (pd_rows_for() never makes more); several GPUs: one share of the panel dots per rank
  constexpr int SPB = MG ? 8 : SPBT;    // folded SP rows per wave in the first batch (covers nt <= 8 * SPB - 1)
  struct RowRegs { double tu[KB], tw[KB], ta[RPB], tb[RPB], uA, uB, ai, aim; };   // what a thread loads for its row of a group
  RowRegs cur, nxt;
  double kdl[4][CHB];
  double spl[SPB][3];
  double abl = 0.0, pcl[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  double bA = 0.0, bB = 0.0;
  double ru[2] = {0.0, 0.0}, rw[2] = {0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < CHB; ++j) kdl[q][j] = 0.0;
#pragma unroll
  for (int j = 0; j < SPB; ++j) { spl[j][0] = 0.0; spl[j][1] = 0.0; spl[j][2] = 0.0; }
  // Everything below is UNCONDITIONAL: no branch (not even a wave-uniform one), clamped addresses that are valid
  // whatever the step, and no loaded value is touched before the marker at the end of the phase.  A select on a
  // loaded value, a branch around a load (its result then has to be copied into the merged register before the
  // branch closes) or a run-time index into a register array makes hipcc wait for the load on the spot: the compiled
  // code of the previous version did ~20 such waits one after the other, 4.8 of the kernel's 9.9 us.  Entries that a
  // lane or a step does not have are loaded from a clamped address and dropped by the masks of the consume phase.
  double ru_raw[2], rw_raw[2];
  double spr[SPB][3];
  unsigned spok = 0;                     // bit j: spr[j] is a real entry of SP for this lane
  // Addresses = wave-uniform base (SGPR pair, one scalar add per load) + ONE 32-bit lane offset per family: the
  // kernel's instruction issue is as long as its memory wait (PMC: SQ_ACTIVE_INST_ANY = SQ_WAIT_INST_ANY), and 64-bit
  // per-lane address arithmetic with clamps was most of it.  The buffers are padded so that unclamped slots stay
  // inside the allocation (band_reduce_impl).
  // row loads of one row group (this thread's row rg of it): unconditional, clamped / padded addresses
  // the row's share of the previous mat-vec's result: partial sums (one GPU) / message entries (several GPUs)
  auto load_rows_msg = [&](int rg, RowRegs& Q) {
    const bool ok = rg < S.rows;
    const bool okp = hp && ok && rg < S.Lprev;
    // SYMV partial sums of the row: t-th partial, t in [0, nt]: t <= ty -> column result of tile row t
    // (column r of tile (t, ty)); t > ty -> row result of tile column t-1 (row r of tile (ty, t-1))
    if (!mg) {   // (compile-time)  one GPU: slot t of the unified array Y = YC (see band_reduce_impl)
      const unsigned voff = (unsigned)ks * (unsigned)(NB * ldp) + (unsigned)(okp ? rg : 0);
#pragma unroll
      for (int j = 0; j < RPB; ++j) {
        const int jb = (j * KA_SL < nt + 1) ? j * KA_SL : 0;        // uniform; unused batches re-read batch 0
        const double* by = R.YC + (size_t)jb * NB * ldp;
        Q.ta[j] = by[voff];
        Q.tb[j] = by[voff + (NB == 2 ? (unsigned)ldp : 0u)];
      }
    } else {
      // several GPUs: slice ks < Py + Px takes one rank's contribution to this row from the step messages
      const int rc = okp ? rg : 0;
      const int t = (ks < R.Px + R.Py) ? ks : 0;
      int stv;
      const double* b = mg_partial<NB>(R, S.par, t, rc, stv);
      Q.ta[0] = ld_sys(b);
      Q.tb[0] = ld_sys(b + (NB == 2 ? stv : 0));
#pragma unroll
      for (int j = 1; j < RPB; ++j) { Q.ta[j] = 0.0; Q.tb[j] = 0.0; }
    }
  };
  // (with_msg = false: the row's local data only -- what a several-GPU launch requests BEFORE it waits for the messages)
  auto load_rows = [&](int rg, RowRegs& Q, bool with_msg = true) {
    const bool ok = rg < S.rows;
    {
      const unsigned voff = (unsigned)ks * (unsigned)ldp + (unsigned)(ok ? rg : 0);
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        // uniform; batches beyond the panel fill re-read batch 0 (same cache lines) instead of touching new memory
        const int jb = (j * KA_SL < kloop) ? j * KA_SL : 0;
        const double* bu = Up + (size_t)jb * ldp;
        const double* bw = Wp + (size_t)jb * ldp;
        Q.tu[j] = bu[voff];
        Q.tw[j] = bw[voff];
      }
    }
    {
      const int ic0 = S.i, ic1 = (S.i > 0) ? S.i - 1 : 0;
      // column i of the (lazily updated) matrix: from A itself, or from the gathered panel on several GPUs
      const double* ci = mg ? R.PAN + (size_t)((ic0 > S.pan_c0 ? ic0 : S.pan_c0) - S.pan_c0) * R.ldpan : R.A + (size_t)ic0 * R.lda;
      const double* cm = mg ? R.PAN + (size_t)((ic1 > S.pan_c0 ? ic1 : S.pan_c0) - S.pan_c0) * R.ldpan : R.A + (size_t)ic1 * R.lda;
      Q.ai = ci[(rg <= ic0) ? rg : ic0];
      Q.aim = cm[(rg <= ic1) ? rg : ic1];
      const int rc = ok ? rg : 0;
      Q.uA = Up[(size_t)kp * ldp + rc];
      Q.uB = Up[(size_t)(kp + (NB == 2 ? 1 : 0)) * ldp + rc];
    }
    if (with_msg) load_rows_msg(rg, Q);
  };
  // masks of the row data (consume side): entries loaded from clamped addresses for rows / steps that have none
  auto mask_rows = [&](int rg, RowRegs& Q) {
    const bool ok = rg < S.rows;
    if (!(S.ncols > 0 && ks == 0 && rg <= S.i)) Q.ai = 0.0;
    if (!(S.ncols > 1 && ks == 0 && rg <= S.i - 1)) Q.aim = 0.0;
    if (!(hp && ks == 0 && ok)) { Q.uA = 0.0; Q.uB = 0.0; }
    if (NB == 1) Q.uB = 0.0;
#pragma unroll
    for (int j = 0; j < RPB; ++j) { if (!hp || (!mg && !(j * KA_SL < nt + 1)) || (mg && j > 0)) { Q.ta[j] = 0.0; Q.tb[j] = 0.0; } }
  };
  load_rows(r, cur, !(MG && S.wait.n > 0));
  {
    const int kku = (tid < S.k) ? tid : 0;
    const int ic0 = S.i, ic1 = (S.i > 0) ? S.i - 1 : 0;
    ru_raw[0] = Up[(size_t)kku * ldp + ic0];
    rw_raw[0] = Wp[(size_t)kku * ldp + ic0];
    ru_raw[1] = Up[(size_t)kku * ldp + ic1];
    rw_raw[1] = Wp[(size_t)kku * ldp + ic1];
  }
  bA = R.sc[SC_BETA_A];
  bB = R.sc[SC_BETA_B];
  if (MG && S.wait.n > 0) {
    // several GPUs, wait folded into this kernel: everything above is local data (panel rows, the next block columns,
    // scalars) and travels while the first wave polls the arrival flags of the Y messages; what follows reads them
    step_wait_fused(S.wait, bid == 0);
    EIGX_STAMP(5);
    EIGX_TL(27);   // (timeline: Y flags seen)
    EIGX_PW(1088);
    load_rows_msg(r, cur);
  }
  // panel dots: thread kk = tid (< kp <= 256) sums entry (kind, kk) over the K_P row chunks
  {
#pragma unroll
    for (int j = 0; j < CHB; ++j) {
#pragma unroll
      for (int q = 0; q < 2 * NB; ++q) {
        if (!mg) {   // (compile-time)
          const double* bk = R.KD + (size_t)((j < S.nchunk_prev ? j : 0) * 2 * NB + q) * m;   // uniform
          kdl[q][j] = bk[(unsigned)tid];
        } else {
          // rank j's share (the rows it owns) from its Y message: behind the row / column sums and the 8 scalar slots
          const double* bk = R.MSG + (size_t)S.par * R.ypar_stride + (size_t)(j < R.P ? j : 0) * R.ysrc_stride +
                             NB * (R.nxs + R.nys) + 8 + q * m;
          kdl[q][j] = ld_sys(bk + (tid < m ? tid : 0));
        }
      }
    }
  }
  // bilinear partials of the SYMV tiles, SP[ty][tx] with the fixed row stride maxseg, tx >= ty only.
  // Folded rows: row f (nt - f tiles) and row nt-1-f (f + 1 tiles) together fill nt + 1 <= 64 lanes;
  // wave w takes the folded rows f = w, w + 4, ...
  if (!mg) {   // (compile-time)
#pragma unroll
    for (int j = 0; j < SPB; ++j) {
      const int f = wave + 4 * j;
      const int cnt = nt - f;          // tiles in row f
      const int f2 = nt - 1 - f;       // partner row (== f for the middle row of an odd nt: skipped)
      const bool first = lane < cnt;
      const int ty = first ? f : f2;
      const int tx = first ? f + lane : f2 + (lane - cnt);
      const bool ok = hp && nt <= 8 * SPB - 1 && 2 * f < nt && (first || ((f2 > f) && tx < nt));
      const double* sp = R.SP + ((size_t)(ok ? ty : 0) * R.maxseg + (ok ? tx : 0)) * 3;
      spr[j][0] = sp[0];
      spr[j][1] = sp[NB == 2 ? 1 : 0];
      spr[j][2] = sp[NB == 2 ? 2 : 0];
      spok |= ok ? (1u << j) : 0u;
    }
  } else {
    // bilinear scalars: one message per rank
    const double* b = R.MSG + (size_t)S.par * R.ypar_stride + (size_t)(tid < R.P ? tid : 0) * R.ysrc_stride + NB * (R.nxs + R.nys);
    spr[0][0] = ld_sys(b);
    spr[0][1] = ld_sys(b + (NB == 2 ? 1 : 0));
    spr[0][2] = ld_sys(b + (NB == 2 ? 2 : 0));
  }
  const int nch_ab = MG ? S.nchunk_ab : S.nchunk_prev;   // entries of the uA.uB partial sums (reflector-store chunks)
  abl = R.KD[R.kdab_off + (tid < nch_ab ? tid : 0)];
  // P(c, a): rows c of the previous step's SYMV result for the new block columns
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int c = (S.i - cc > 0) ? S.i - cc : 0;
    if (!mg) {   // (compile-time)
      const unsigned t = (tid < nt + 1) ? (unsigned)tid : 0u;
      const double* bc = R.YC + c;                                  // uniform
      pcl[cc][0] = bc[t * (unsigned)(NB * ldp)];
      pcl[cc][1] = bc[t * (unsigned)(NB * ldp) + (NB == 2 ? (unsigned)ldp : 0u)];
    } else {
      const int t = (tid < R.Px + R.Py) ? tid : 0;
      int stv;
      const double* b = mg_partial<NB>(R, S.par, t, c, stv);
      pcl[cc][0] = ld_sys(b);
      pcl[cc][1] = ld_sys(b + (NB == 2 ? stv : 0));
    }
  }
  asm volatile("" ::: "memory");   // ---- marker: every load of the phase has been issued
  // ---- masks (consume side): drop what was loaded from clamped addresses for lanes / steps that have no such entry
  {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const bool have = cc < S.ncols;
      ru[cc] = (have && tid < S.k) ? ru_raw[cc] : 0.0;
      rw[cc] = (have && tid < S.k && tid < kold) ? rw_raw[cc] : 0.0;
      if (!(have && hp)) { pcl[cc][0] = 0.0; pcl[cc][1] = 0.0; }
    }
    if (!hp) { bA = 0.0; bB = 0.0; abl = 0.0; }
    if (NB == 1) bB = 0.0;
    mask_rows(r, cur);
#pragma unroll
    for (int j = 0; j < CHB; ++j) {
      if (!(hp && j < S.nchunk_prev)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) kdl[q][j] = 0.0;
      }
    }
#pragma unroll
    for (int j = 0; j < SPB; ++j) {
      const bool ok = (spok >> j) & 1u;
      spl[j][0] = ok ? spr[j][0] : 0.0; spl[j][1] = ok ? spr[j][1] : 0.0; spl[j][2] = ok ? spr[j][2] : 0.0;
    }
    if (mg) { spl[0][0] = hp ? spr[0][0] : 0.0; spl[0][1] = hp ? spr[0][1] : 0.0; spl[0][2] = hp ? spr[0][2] : 0.0; }
  }
  EIGX_STAMP(0);
  // ---- everything is in flight; now consume -----------------------------------------------------------
  double v[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [0..2] SP, [3..5] corrections, [6] uA.uB, [7..10] P(c,a)
  // the two reciprocals of the 2x2 algebra only need the betas: start them here, off the post-reduction chain
  const double tAA = (hp && bA != 0.0) ? 1.0 / bA : 0.0;
  const double tBB = (hp && bB != 0.0) ? 1.0 / bB : 0.0;
  if (hp) {
    if (mg) {
      const int npart = R.Px + R.Py;
      if (tid < R.P) { v[0] = spl[0][0]; if (NB == 2) { v[1] = spl[0][1]; v[2] = spl[0][2]; } }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (cc < S.ncols && tid < npart) { v[7 + 2 * cc] += pcl[cc][0]; if (NB == 2) v[8 + 2 * cc] += pcl[cc][1]; }
      }
    } else {
#pragma unroll
      for (int j = 0; j < SPB; ++j) {
        v[0] += spl[j][0];
        if (NB == 2) { v[1] += spl[j][1]; v[2] += spl[j][2]; }
      }
      if (nt > 8 * SPB - 1) {   // more tiles than the folded batch covers: plain sweep of the upper tile triangle
        for (int ty = wave; ty < nt; ty += 4)
          for (int tx = ty + lane; tx < nt; tx += 64) {
            const double* sp = R.SP + ((size_t)ty * R.maxseg + tx) * 3;
            v[0] += sp[0];
            if (NB == 2) { v[1] += sp[1]; v[2] += sp[2]; }
          }
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (cc < S.ncols) {
          const bool ok = tid < nt + 1;
          v[7 + 2 * cc] += ok ? pcl[cc][0] : 0.0;
          if (NB == 2) v[8 + 2 * cc] += ok ? pcl[cc][1] : 0.0;
          const int c = S.i - cc, ty = c >> lgT;
          for (int t = tid + 256; t < nt + 1; t += 256) {
            const double* base = (t <= ty) ? R.YC + (size_t)t * NB * ldp : R.YR + (size_t)(t - 1) * NB * ldp;
            v[7 + 2 * cc] += base[c];
            if (NB == 2) v[8 + 2 * cc] += base[ldp + c];
          }
        }
      }
    }
    if (NB == 2) {
      v[6] = (tid < nch_ab) ? abl : 0.0;
      for (int c = tid + 256; c < nch_ab; c += 256) v[6] += R.KD[R.kdab_off + c];
    }
  }
  EIGX_STAMP(1);

  // ============ phase 1: publish kd / rowU / rowW in LDS ============================================
  double kdr[4] = {0.0, 0.0, 0.0, 0.0};
  if (hp && tid < kp) {
#pragma unroll
    for (int q = 0; q < 2 * NB; ++q) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < CHB; ++j) acc += kdl[q][j];
      for (int c = CHB; c < S.nchunk_prev; ++c) acc += R.KD[((size_t)c * 2 * NB + q) * m + tid];
      kdr[q] = acc;
      kd[q][tid] = acc;
    }
  }
  if (S.ncols > 0 && tid < S.k) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
      if (cc < S.ncols) { rowU[cc][tid] = ru[cc]; rowW[cc][tid] = rw[cc]; }
  }
  // ============ phase 2: one block reduction for all replicated scalars ================================
  // thread kk = tid holds kd(:, kk) in registers: the corrections need no LDS round trip
  double tm[7] = {0, 0, 0, 0, 0, 0, 0};  // T (tAA,tAB,tBB) and M (m11,m12,m21,m22), in every thread
  double wnew[2][2] = {{0.0, 0.0}, {0.0, 0.0}};  // W(c, kp), W(c, kp+1) of the new block columns
  if (hp) {
    if (tid < kp) {
      v[3] += 2.0 * kdr[0] * kdr[1];
      if (NB == 2) {
        v[4] += kdr[0] * kdr[3] + kdr[1] * kdr[2];
        v[5] += 2.0 * kdr[2] * kdr[3];
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (cc < S.ncols) {
          v[7 + 2 * cc] -= ru[cc] * kdr[1] + rw[cc] * kdr[0];
          if (NB == 2) v[8 + 2 * cc] -= ru[cc] * kdr[3] + rw[cc] * kdr[2];
        }
      }
    }
    // G = (bilinear partials) - (panel corrections) needs only the differences: 8 values to reduce, not 11
    double rv[8] = {v[0] - v[3], v[1] - v[4], v[2] - v[5], v[6], v[7], v[8], v[9], v[10]};
    block_sum_multi<8>(rv, red);     // (its first barrier also publishes kd / rowU / rowW)
    v[7] = rv[4]; v[8] = rv[5]; v[9] = rv[6]; v[10] = rv[7];
    // the 2x2 algebra in every thread (no broadcast, no extra barrier)
    const double gAA = rv[0], gAB = rv[1], gBB = rv[2], uab = rv[3];
    const double tAB = -uab * tAA * tBB;
    // GT = G T ; M = T^T GT
    const double gt11 = gAA * tAA, gt12 = gAA * tAB + gAB * tBB;
    const double gt21 = gAB * tAA, gt22 = gAB * tAB + gBB * tBB;
    tm[0] = tAA; tm[1] = tAB; tm[2] = tBB;
    tm[3] = tAA * gt11;               // m11
    tm[4] = tAA * gt12;               // m12
    tm[5] = tAB * gt11 + tBB * gt21;  // m21
    tm[6] = tAB * gt12 + tBB * gt22;  // m22
    // W(c, new slots) = row c of the previous step's W (same formula as the row loop below)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      if (cc < S.ncols) {
        const double pA = v[7 + 2 * cc], pB = v[8 + 2 * cc];
        const double uA = rowU[cc][kp], uB = (NB == 2) ? rowU[cc][kp + 1] : 0.0;
        const double yA = tm[0] * pA, yB = tm[1] * pA + tm[2] * pB;
        wnew[cc][0] = yA - 0.5 * (uA * tm[3] + uB * tm[5]);
        if (NB == 2) wnew[cc][1] = yB - 0.5 * (uA * tm[4] + uB * tm[6]);
      }
    }
  } else {
    __syncthreads();
  }
  EIGX_STAMP(2);

  // ============ phase 3: the row groups of this workgroup ===============================================
  double gg[3] = {0.0, 0.0, 0.0};
  for (int g = 0; g < G; ++g) {
    const int rg = r + g * rstep;
    const bool okg = rg < S.rows;
    const bool more = LG && (g + 1 < G);
    if (more) load_rows(rg + rstep, nxt);          // the next group's loads travel behind this group's work
    // my share of this row's SYMV partial sums
    double prA = 0.0, prB = 0.0;
    {
      const bool okp = hp && okg && rg < S.Lprev;
      if (hp) {
        if (mg) {
          if (okp && ks < R.Px + R.Py) { prA = cur.ta[0]; if (NB == 2) prB = cur.tb[0]; }
        } else {
#pragma unroll
          for (int j = 0; j < RPB; ++j) {
            const bool ok = okp && (ks + j * KA_SL < nt + 1);
            prA += ok ? cur.ta[j] : 0.0;
            if (NB == 2) prB += ok ? cur.tb[j] : 0.0;
          }
          if (okp) {   // more than RPB * KA_SL = 160 partials per row: not reached by symv_geom below N ~ 80000
            for (int t = ks + RPB * KA_SL; t < nt + 1; t += KA_SL) {
              const double* base = R.YC + (size_t)t * NB * ldp;
              prA += base[rg];
              if (NB == 2) prB += base[ldp + rg];
            }
          }
        }
      }
    }
    {
      double pA = prA, pB = prB, x0 = 0.0, x1 = 0.0;
      auto accum = [&](int kk, double u, double w) {
        if (hp) {
          pA -= u * kd[1][kk] + w * kd[0][kk];
          if (NB == 2) pB -= u * kd[3][kk] + w * kd[2][kk];
        }
        if (S.ncols > 0) {
          x0 += u * rowW[0][kk] + w * rowU[0][kk];
          if (S.ncols > 1) x1 += u * rowW[1][kk] + w * rowU[1][kk];
        }
      };
      if (okg) {
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          const int kk = ks + j * KA_SL;
          if (kk < kloop) accum(kk, cur.tu[j], cur.tw[j]);
        }
        for (int k0 = ks + KB * KA_SL; k0 < kloop; k0 += KB * KA_SL) {   // m > 128 only
          double xu[KB], xw[KB];
#pragma unroll
          for (int j = 0; j < KB; ++j) {
            const int kk = (k0 + j * KA_SL < kloop) ? k0 + j * KA_SL : 0;
            xu[j] = Up[(size_t)kk * ldp + rg];
            xw[j] = Wp[(size_t)kk * ldp + rg];
          }
#pragma unroll
          for (int j = 0; j < KB; ++j)
            if (k0 + j * KA_SL < kloop) accum(k0 + j * KA_SL, xu[j], xw[j]);
        }
      }
      // the wave's 4 slices (lane >> 4) of each row are combined with two shuffles, the 4 waves through LDS
      double q4[4] = {pA, pB, x0, x1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        q4[q] += __shfl_xor(q4[q], 16, 64);
        q4[q] += __shfl_xor(q4[q], 32, 64);
      }
      if (lane < KA_ROWS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) slice[wave][lane][q] = q4[q];
      }
    }
    __syncthreads();
    EIGX_STAMP(3);
    if (ks == 0 && okg) {
      double pA = 0.0, pB = 0.0, x0 = 0.0, x1 = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pA += slice[q][rr][0]; pB += slice[q][rr][1]; x0 += slice[q][rr][2]; x1 += slice[q][rr][3];
      }
      double wA = 0.0, wB = 0.0;
      if (hp) {
        const double uA = cur.uA, uB = cur.uB;
        const double yA = tm[0] * pA, yB = tm[1] * pA + tm[2] * pB;
        wA = yA - 0.5 * (uA * tm[3] + uB * tm[5]);
        wB = (NB == 2) ? yB - 0.5 * (uA * tm[4] + uB * tm[6]) : 0.0;
        if (rg >= S.Lprev) { wA = 0.0; wB = 0.0; }
        Wp[(size_t)kp * ldp + rg] = wA;
        if (NB == 2) Wp[(size_t)(kp + 1) * ldp + rg] = wB;
        if (S.ncols > 0) {
          x0 += uA * wnew[0][0] + wA * rowU[0][kp];
          if (NB == 2) x0 += uB * wnew[0][1] + wB * rowU[0][kp + 1];
          if (S.ncols > 1) {
            x1 += uA * wnew[1][0] + wA * rowU[1][kp];
            if (NB == 2) x1 += uB * wnew[1][1] + wB * rowU[1][kp + 1];
          }
        }
      }
      double xi = 0.0, xim = 0.0;
      if (S.ncols > 0 && rg <= S.i) {
        xi = cur.ai - x0;
        if (!MG) R.X[rg] = xi;
        if (S.ncols > 1 && rg <= S.i - 1) {
          xim = cur.aim - x1;
          if (!MG) R.X[ldp + rg] = xim;
        }
        if (rg < S.L) { gg[0] += xi * xi; gg[1] += xi * xim; gg[2] += xim * xim; }
        if (!MG) {   // (several GPUs: every rank takes d, e from the X messages -- symv_kernel's publisher, mg_tail_kernel)
          if (rg == S.i) R.d[S.i] = xi;
          if (S.ncols > 1 && rg == S.i - 1) { R.e[S.i] = xi; R.d[S.i - 1] = xim; }  // e(i,1) = A_eff(i-1,i)
        }
      }
      if (MG) {
        // my rows of the new x and of the finished W into every rank's X window: the 16 row lanes of the group write 128
        // contiguous bytes per field and destination (write-through stores over xGMI)
        const size_t off = (size_t)S.xpar * XP.parity_stride + (size_t)(((bid * G + g) << 4) | rr);
        for (int d_ = 0; d_ < XP.n; ++d_) {
          double* q_ = XP.slot[d_] + off;
          if (S.ncols > 0) { st_sys(q_, xi); if (NB == 2) st_sys(q_ + R.nown, xim); }
          if (hp) { st_sys(q_ + 2 * (size_t)R.nown, wA); if (NB == 2) st_sys(q_ + 3 * (size_t)R.nown, wB); }
        }
      }
    }
    if (more) {
      __syncthreads();                 // slice[] is written again by the next group
      mask_rows(rg + rstep, nxt);
      cur = nxt;
    }
  }
  if ((MG || S.ncols > 0) && wave == 0) {
    // Gram partials of the new columns over the rows above the block: x_i.x_i, x_i.x_{i-1}, x_{i-1}.x_{i-1}: one entry
    // per workgroup (the row threads of every group are lanes 0..15 of wave 0: no block reduction needed)
#pragma unroll
    for (int q = 0; q < 3; ++q) gg[q] = wave_sum(gg[q]);
    if (lane == 0) {
      if (MG) { st_agent(&R.GP[bid * 3 + 0], gg[0]); st_agent(&R.GP[bid * 3 + 1], gg[1]); st_agent(&R.GP[bid * 3 + 2], gg[2]); }
      else { R.GP[bid * 3 + 0] = gg[0]; R.GP[bid * 3 + 1] = gg[1]; R.GP[bid * 3 + 2] = gg[2]; }
    }
  }
  if (MG) {
    // Publish the X message: every storing wave waits for its write-through stores to be acknowledged, the barrier collects
    // the waves, one agent-scope add counts the workgroup; the workgroup whose add came last sums the rank's Gram partial
    // sums in a fixed order (whoever it is), appends them to the message and raises the flag on every rank (the only store
    // with a system-scope release) -- the protocol of kl_publish, self-tested at init (comm.hip st_step_push_kernel).
    EIGX_STAMP(6);
    EIGX_TL(28);   // (timeline: pushes issued)
    EIGX_PW(2112);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    EIGX_STAMP(12);
    // arrival: a fire-and-forget add (a RETURNING device-scope atomic costs its whole round trip, 2.7 us in the stamps, on
    // the last arriver's path); the role's first workgroup watches the counter with relaxed loads and publishes
    if (tid == 0) {
      __hip_atomic_fetch_add(XP.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lastw = 0;
      if (bid == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(XP.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblocks) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > 200000000ll) break;     // (2 s: never on a healthy run; the consumers' bounded waits report)
        }
        __hip_atomic_store(XP.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lastw = 1;
#ifdef EIGX_STAMPS
        if (R.dbg && S.i == R.stamp_i) R.dbg[3904] = (unsigned long long)wall_clock64();
#endif
#ifdef EIGX_STAMPS
        if (R.dbg) { const unsigned long long t0_ = __hip_atomic_load(&R.dbg[20], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long dt_ = (unsigned long long)wall_clock64() - t0_; if (t0_ != 0 && dt_ < 100000ull) atomicAdd(&R.dbg[29], dt_); }
#endif
      }
    }
    __syncthreads();
    if (lastw) {
      double g3[3] = {0.0, 0.0, 0.0};
      for (int q = tid; q < nblocks; q += 256) {
        g3[0] += ld_agent(&R.GP[3 * q]); g3[1] += ld_agent(&R.GP[3 * q + 1]); g3[2] += ld_agent(&R.GP[3 * q + 2]);
      }
      block_sum_multi<3>(g3, red);
      if (tid < 3) {
        const size_t off = (size_t)S.xpar * XP.parity_stride + 4 * (size_t)R.nown + tid;
        for (int d_ = 0; d_ < XP.n; ++d_) st_sys(XP.slot[d_] + off, g3[tid]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the Gram stores and the flag stores are the first wave's)
      if (tid < XP.n && XP.flag[tid])
        __hip_atomic_store(XP.flag[tid] + S.xpar * EIGX_MAXP, S.xepoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef EIGX_STAMPS
      if (R.dbg && tid == 0 && S.i == R.stamp_i) R.dbg[3905] = (unsigned long long)wall_clock64();
      if (R.dbg && tid == 0) {
        atomicAdd(&R.dbg[13], __builtin_amdgcn_s_memtime() - stamp_prev); atomicAdd(&R.dbg[14], 1ull);
        const unsigned long long t0 = __hip_atomic_load(&R.dbg[20], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long dt = (unsigned long long)wall_clock64() - t0;
        if (t0 != 0 && dt < 100000ull) atomicAdd(&R.dbg[23], dt);
      }
#endif
    }
  }
  EIGX_STAMP(4);
#ifdef EIGX_STAMPS
  if (R.dbg && threadIdx.x == 0 && stamp_me) {   // shader clock of this workgroup's lifetime: s_memtime ticks per 10-ns wall tick
    atomicAdd(&R.dbg[30], (unsigned long long)wall_clock64() - clk_w0); atomicAdd(&R.dbg[31], __builtin_amdgcn_s_memtime() - clk_m0);
  }
  if (R.dbg && threadIdx.x == 0 && stamp_me) atomicAdd(&R.dbg[7], 1ull);
#endif
}

template <int NB, bool MG, bool LG, int RPBT, int SPBT, int KBT>
__global__ __launch_bounds__(256) void ka_kernel(RedArgs R, KAArgs S) {
  ka_body<NB, MG, LG, RPBT, SPBT, KBT>(R, S, nullptr, blockIdx.x, gridDim.x);
}

// =================================================================================================
// K_B : fused symmetric mat-vec on the upper triangle, NV vectors, plus (same launch, extra workgroup
// rows) K_P : panel dot products and reflector store.
// grid: 1-D, npd*(ncg+1) panel workgroups followed by the nt(nt+1)/2 tile workgroups (row-major upper block triangle)
//   tile (ty, tx), tx >= ty, T = 128*RB:
//                      rows [ty*T, +T) x columns [tx*T, +T) of [0,L).  Wave w owns the tile columns
//                      [w*T/4, +T/4) in groups of 8; a lane owns rows 2*lane, 2*lane+1 of each of the RB
//                      128-row blocks (16-byte loads, 1 KiB per wave-instruction, down the columns).
//                      Column sums: halving butterfly of wave shuffles once per column group.
//                      Row sums: per-wave registers, combined over the 4 waves through LDS.
//   panel workgroup (row chunk, column group cg):
//                      cg < ncg : panel columns [cg*PD_COLS, +PD_COLS) of U and W against the NV new vectors
//                      cg == ncg: store the reflectors into the panel (both U copies) and into `a`, uA.uB
// =================================================================================================
struct KBArgs { int i, L, nt, ngp, k, ncg, toprows, pdr, npd;
  int ng;   // column groups per wave (= T / 32): a kernel ARGUMENT so that the pipeline loops stay rolled (hipcc would
            // otherwise unroll them and hoist loads: more registers, lower occupancy)
  // multi-GPU tile enumeration of the local block (see symv_kernel): Lr / Lc = local rows / columns below L,
  // ntc tile columns; the last (clipped) tile column has nty_last tiles and comes first in the grid; tile column
  // tx < ntc - 1 has slope*tx + c1 tiles when Px divides Py (slope = Py / Px), otherwise slope = 0 and the kernel counts
  int Lr, Lc, ntc, nty_last, slope, c1;
  // several GPUs: the panel dots run over the rows THIS rank owns (nown_L of them below L, chunks of pdr owned indices:
  // npd x ncg workgroups), the reflector store over all rows (npd_s chunks of pdr_s rows behind them); the store chunks
  // also copy the W columns [wk, wk + NV) that the previous ka_kernel launches sent (wk < 0: nothing to copy)
  int nown_L, npd_s, pdr_s, wk;
  int xpar;            // parity of the X message that holds this step's x (and the W columns wk..)
  StepWait xwait;      // wait for that message in the prologue (xwait.n = 0: a wait kernel ran, or stream order)
};

// tiles of tile column tx of the local block (T-row tile ty exists iff its first global row <= the last global
// column of the tile column):  floor((last global column - px) / (Px T)) + 1
__host__ __device__ inline int mg_nty(int tx, int T, int Lc, int Px, int px, int Py, int py) {
  const int lc = (tx * T + T - 1 < Lc - 1) ? tx * T + T - 1 : Lc - 1;
  const long num = (long)lc * Py + py - px;
  return num >= 0 ? (int)(num / ((long)Px * T)) + 1 : 0;
}

// K_L (multi-GPU only): reduce this rank's SYMV partials over its tiles -- row sums of the local rows, column sums
// of the local columns, bilinear scalars, its share of the panel dots -- and WRITE the result into the step windows
// (system-scope stores over xGMI; the ranks of a node are all directly linked): a row's / column's sum goes to the rank
// that owns that global row for the panel work, everything else to every rank; then publish the step's flag on every rank.
struct KLArgs {
  int L, Lr, Lc, T, ntc, nbr, par;
  unsigned long long epoch;
  StepPeers peers;
  // second level of the panel dots: the mat-vec launch's K_P workgroups take short chunks of the rank's rows and write kd2;
  // kl_kernel's extra workgroups sum the npd2 chunks and send the rank's share to everybody
  const double* kd2; int npd2, kfill;
  int ntr;     // tile rows that hold at least one tile (= the largest mg_nty over the tile columns)
  int fence;   // 1: every pushing workgroup runs a system-scope fence behind its stores (EIGX_STEP_FENCE=1); 0: see kl_publish
};

// one chunk of KL_ROWS = 128 local rows (rows = true) or local columns starting at l0: sum this rank's tile partial sums of
// each and write the result into the step window of the row's owner (of every rank for the rows of the next block
// columns, L-1 and L-2: every rank needs P(c, :) of those, see ka_kernel).  Half q of the workgroup takes every second
// partial sum of a row (up to ~64 of them sit behind a cold L2: one thread per row walking them one after the other is a
// chain of memory round trips), the two halves are combined through LDS in a fixed order.  128 rows per workgroup (round
// 4; was 64 with four quarters): at N = 32768 on 2 x 4 the kl and ka roles of a step launch then fit the chip together
// (196 + 256 workgroups of 512 slots), so every ka workgroup has its local data in flight while kl still runs.
constexpr int KL_ROWS = 128;
template <int NB>
__device__ __forceinline__ void kl_chunk(const RedArgs& R, const KLArgs& K, bool rows, int l0, double (*comb)[KL_ROWS][2], int kl_bid = 0) {
  const int ldp = R.ldp;
  const int lane = threadIdx.x & (KL_ROWS - 1), q = threadIdx.x / KL_ROWS;    // q = 0, 1
  const int T = K.T;
  const int l = l0 + lane;
  double pA = 0.0, pB = 0.0;
  // this thread's partial sums: tile index t0, t0 + 2, ... < tend of the row's / column's partial-sum array P (stride
  // NB * ldp per tile).  They sit behind a cold L2 (written by tiles on other XCDs): ALL loads of a batch of 16 per vector
  // are issued before the first add -- clamped indices, masked afterwards -- so a row costs one memory round trip, not
  // one per tile (the earlier loop with a serial remainder took ~20 us per launch at 32-64 tiles per row).
#ifdef EIGX_STAMPS
#define EIGX_KLW(base) do { if (R.dbg && threadIdx.x == 0 && K.L == R.stamp_i - 1 && kl_bid < 256) R.dbg[(base) + kl_bid] = (unsigned long long)wall_clock64(); } while (0)
#else
#define EIGX_KLW(base) do {} while (0)
#endif
  EIGX_KLW(3136);
  const bool act = rows ? (l < K.Lr) : (l < K.Lc);
  const double* P = rows ? R.YR : R.YC;
  int t0 = q, tend = 0;
  if (act) {
    if (rows) {
      // tiles (ty, tx) with tx >= txmin, the tile column that holds the first local column at or right of the tile row's
      // first global row
      const int ty = l / T;
      const long g0 = (long)ty * T * R.Px + R.px;
      const long cneed = g0 > R.py ? (g0 - R.py + R.Py - 1) / R.Py : 0;
      if (cneed <= K.Lc - 1) { t0 = (int)(cneed / T) + q; tend = K.ntc; }
    } else {
      tend = mg_nty(l / T, T, K.Lc, R.Px, R.px, R.Py, R.py);
    }
  }
  const int lc = act ? l : 0;
  for (int tb = t0; tb < tend; tb += 32) {
    double a[16], b[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int t = (tb + 2 * e < tend) ? tb + 2 * e : tb;
      a[e] = P[((size_t)t * NB + 0) * ldp + lc];
      b[e] = (NB == 2) ? P[((size_t)t * NB + 1) * ldp + lc] : 0.0;
    }
    asm volatile("" ::: "memory");   // every load of the batch issued
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const bool ok = tb + 2 * e < tend;
      pA += ok ? a[e] : 0.0;
      pB += ok ? b[e] : 0.0;
    }
  }
  comb[q][lane][0] = pA; comb[q][lane][1] = pB;
  __syncthreads();
  EIGX_KLW(3392);
  const size_t pbase = (size_t)K.par * K.peers.parity_stride;
  if (q == 0 && act) {
    const double sA = comb[0][lane][0] + comb[1][lane][0];
    const double sB = comb[0][lane][1] + comb[1][lane][1];
    const size_t off = pbase + (rows ? 0 : (size_t)NB * R.nxs) + l;
    const int stv = rows ? R.nxs : R.nys;
    const int gidx = rows ? l * R.Px + R.px : l * R.Py + R.py;     // global row this sum belongs to
    int own, oi;
    own_of(R, gidx, own, oi);
    const bool all = (gidx >= K.L - 2);                            // rows of the next block columns (gidx < L here)
    for (int d = 0; d < K.peers.n; ++d) {
      if (K.peers.n == 1 || all || d == own) {                     // (n == 1: collective form, one local send buffer)
        st_sys(K.peers.slot[d] + off, sA);
        if (NB == 2) st_sys(K.peers.slot[d] + off + stv, sB);
      }
    }
  }
  __syncthreads();   // comb is reused by the caller's next chunk
}

// the three bilinear scalars of this rank: sum over its tiles, written into every rank's step window.  A workgroup of its
// own (round 4: as a side job of the first row chunk's workgroup its serial walk over the tile columns -- a memory round
// trip each -- kept that one workgroup busy for 8.7 us while the other ~200 had drained their stores after 4.6: the Y
// flags went up at ~10 us).  Wave q takes the tile columns q, q + 4, ...; ALL loads of up to 16 of them are issued before
// the first add (clamped indices, masked afterwards); lane = tile row.
template <int NB>
__device__ __forceinline__ void kl_scalars(const RedArgs& R, const KLArgs& K, double* red) {
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  double v[3] = {0.0, 0.0, 0.0};
  for (int ty0 = 0; ty0 < K.ntr; ty0 += 64) {
    const int ty = ty0 + lane;
    for (int tb = q; tb < K.ntc; tb += 64) {
      double a[16][3];
      bool ok[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tx = tb + 4 * e;
        const int txc = tx < K.ntc ? tx : tb;
        ok[e] = tx < K.ntc && ty < mg_nty(txc, K.T, K.Lc, R.Px, R.px, R.Py, R.py);
        const double* sp = R.SP + ((size_t)(ok[e] ? ty : 0) * R.maxseg + txc) * 3;
        a[e][0] = sp[0]; a[e][1] = sp[1]; a[e][2] = sp[2];
      }
      asm volatile("" ::: "memory");   // every load of the batch issued
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[0] += ok[e] ? a[e][0] : 0.0; v[1] += ok[e] ? a[e][1] : 0.0; v[2] += ok[e] ? a[e][2] : 0.0;
      }
    }
  }
  block_sum_multi<3>(v, red);
  if (threadIdx.x < 3) {
    const size_t off = (size_t)K.par * K.peers.parity_stride + (size_t)NB * (R.nxs + R.nys) + threadIdx.x;
    for (int d = 0; d < K.peers.n; ++d) st_sys(K.peers.slot[d] + off, v[threadIdx.x]);
  }
}

// after a workgroup's pushes: drain, count, and let the role's watching workgroup publish the step's flag on every rank
// once the count is complete (the counter returns to zero for the next launch)
// The payload stores are system-scope write-through stores (st_sys): each storing wave waits until they are acknowledged
// (vmcnt(0)), the workgroup barrier collects the waves, ONE agent-scope add counts the workgroup, and only the publishing
// workgroup's flag store carries a system-scope release.  A system-scope fence in EVERY pushing workgroup (the earlier
// form, kept behind EIGX_STEP_FENCE=1) writes back whatever is dirty in the XCD's L2 -- the mat-vec's partial sums of
// the whole launch -- several hundred times per step: 20-38 us per launch in the rehearsal of one rank at N = 32768.  The
// init-time self-test (comm.hip, st_step_push_kernel) runs this very protocol with checksummed payloads before the solver
// relies on it.
__device__ __forceinline__ void kl_publish(const KLArgs& K, unsigned* counter, bool watcher, unsigned total, int* lastw) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (K.fence) __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    // (fire-and-forget add + a watching workgroup: see ka_body)
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lastw = 0;
    if (watcher) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 200000000ll) break;
      }
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *lastw = 1;
    }
  }
  __syncthreads();
  // (collective form of the exchange: one local destination, no flag -- comm_step_allgather follows in stream order)
  if (*lastw && (int)threadIdx.x < K.peers.n && K.peers.flag[threadIdx.x])
    __hip_atomic_store(K.peers.flag[threadIdx.x] + K.par * EIGX_MAXP, K.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (role body of the step launch: workgroup `bid` of `nblocks` = local rows / KL_ROWS + local columns / KL_ROWS + 2 NB + 1)
template <int NB>
__device__ __forceinline__ void kl_body(const RedArgs& R, const KLArgs& K, const int bid, const int nblocks) {
  // KL_ROWS rows (columns) per workgroup
  __shared__ double comb[2][KL_ROWS][2];
  __shared__ double red[16];
  __shared__ int last;
  if (bid == nblocks - 1) {
    kl_scalars<NB>(R, K, red);
  } else if (bid >= nblocks - 1 - 2 * NB) {
    // panel dots, second level: one workgroup per kind q; entry (q, panel column kk) = sum over this rank's row chunks in
    // chunk order (deterministic), ALL chunk loads of a thread in one batch (<= 12 chunks: one memory round trip); the
    // rank's share goes into every rank's window (ka_kernel adds the P shares in rank order)
    const int m = R.m, tid = threadIdx.x;
    const int q = bid - (nblocks - 1 - 2 * NB);
    constexpr int MAXC = 12;
    if (tid < K.kfill) {
      double v[MAXC];
#pragma unroll
      for (int e = 0; e < MAXC; ++e) v[e] = K.kd2[((size_t)((e < K.npd2) ? e : 0) * 2 * NB + q) * m + tid];
      asm volatile("" ::: "memory");
      double acc = 0.0;
#pragma unroll
      for (int e = 0; e < MAXC; ++e) acc += (e < K.npd2) ? v[e] : 0.0;
      const size_t off = (size_t)K.par * K.peers.parity_stride + (size_t)NB * (R.nxs + R.nys) + 8 + (size_t)q * m + tid;
      for (int d = 0; d < K.peers.n; ++d) st_sys(K.peers.slot[d] + off, acc);
    }
  } else {
    const bool rows = bid < K.nbr;
    kl_chunk<NB>(R, K, rows, (rows ? bid : bid - K.nbr) * KL_ROWS, comb, bid);
    // (rows / columns that no tile covers get explicit zeros above: tend = 0)
  }
  // every storing wave drains its stores; the last workgroup to arrive publishes the flag on every rank
#ifdef EIGX_STAMPS
  { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const int kl_bid = bid; EIGX_KLW(3648); }
#endif
  kl_publish(K, K.peers.counter, bid == 0, (unsigned)nblocks, &last);
#ifdef EIGX_STAMPS
  if (R.dbg && last && threadIdx.x == 0) {
    const unsigned long long t0 = __hip_atomic_load(&R.dbg[20], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long dt = (unsigned long long)wall_clock64() - t0;
    if (t0 != 0 && dt < 100000ull) atomicAdd(&R.dbg[22], dt);
  }
#endif
}

// Reflector scalars of a step, computed by EVERY workgroup of the mat-vec launch in the same order (bit-identical
// replicas) from the Gram partials that K_A left in GP (gpt = this thread's two preloaded entries) and four elements of x.
// NV = 1: s = -sign(||x||, x_piv), beta = ||x||^2 - s x_piv.
// NV = 2: two sequential Householder steps on the column pair (x0 = column i, x1 = column i-1) without a
// kernel in between: reflector A from g00 = x0.x0; gamma = uA.x1 / betaA from g01; the reflected second
// column x1' = x1 - gamma uA is formed on the fly wherever it is needed; its norm above the pivot row
// follows from the isometry of H_A: ||x1'(0:L-1)||^2 = g11 - x1'(L-1)^2.  If that difference cancels
// (more than 3/4 of the column's weight in the pivot row) the sum is taken explicitly instead -- same
// order in every workgroup, so the replicas stay bit-identical either way.
struct HouseScalars { double sA, sB, betaA, betaB, gammaB, eL1; };
template <int NV, bool MG = false>
__device__ __forceinline__ HouseScalars house_scalars(const RedArgs& R, int ngp, int L, const double (&gpt)[2][3], double x0L,
                                                      double x1L, double x0P, double x1P, double* red, int xpar = 0) {
  const int tid = threadIdx.x;
  const int pivB = L - 2;
  double sA, sB = 0.0, betaA, betaB = 0.0, gammaB = 0.0, eL1 = 0.0;
  {
    double gv[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = tid + 256 * j < ngp;
      gv[0] += ok ? gpt[j][0] : 0.0;
      if (NV == 2) { gv[1] += ok ? gpt[j][1] : 0.0; gv[2] += ok ? gpt[j][2] : 0.0; }
    }
    for (int q = tid + 512; q < ngp; q += 256) {   // more than 512 K_A workgroups (N > 8192)
      gv[0] += R.GP[3 * q];
      if (NV == 2) { gv[1] += R.GP[3 * q + 1]; gv[2] += R.GP[3 * q + 2]; }
    }
    if (EIGX_ABL(128)) { gv[0] = 1.0; gv[1] = 0.1; gv[2] = 1.0; }   // diagnostic build: no scalar reduction
    else block_sum_multi<3>(gv, red);
    if (EIGX_ABL(2048)) {   // diagnostic build: the scalar reduction a second time (what one such phase costs)
      double g2[3] = {gv[0] + tid, gv[1], gv[2]};
      __syncthreads();
      block_sum_multi<3>(g2, red);
      if (g2[0] == 1.2345678) gv[0] = g2[1];
      __syncthreads();
    }
    if (gv[0] > 0.0) { sA = -sign_of(sqrt(gv[0]), x0L); betaA = gv[0] - sA * x0L; }
    else { sA = x0L; betaA = 0.0; }
    if (NV == 2) {
      if (betaA != 0.0) gammaB = (gv[1] - sA * x1L) / betaA;
      eL1 = x1L - gammaB * (x0L - sA);              // x1'(L-1) = T(i-2, i-1)
      double hB = gv[2] - eL1 * eL1;                // ||x1'(0:L-1)||^2
      if (!(hB >= 0.25 * gv[2])) {
        double h[1] = {0.0};
        for (int j = tid; j < L - 1; j += 256) {
          const double t = xget<MG>(R, xpar, 1, j) - gammaB * xget<MG>(R, xpar, 0, j);
          h[0] += t * t;
        }
        block_sum_multi<1>(h, red);
        hB = h[0];
      }
      const double xP = x1P - gammaB * x0P;         // x1'(L-2)
      if (hB > 0.0 && pivB >= 0) { sB = -sign_of(sqrt(hB), xP); betaB = hB - sB * xP; }
      else { sB = (pivB >= 0) ? xP : 0.0; betaB = 0.0; }
    }
  }
  HouseScalars hs;
  hs.sA = sA; hs.sB = sB; hs.betaA = betaA; hs.betaB = betaB; hs.gammaB = gammaB; hs.eL1 = eL1;
  return hs;
}

// u_A(j) = x0(j) - [j == L-1] sA ;  u_B(j) = x1'(j) - [j == L-2] sB, u_B(L-1) = 0 ;
// zero when the reflector is trivial or j >= L.  raw0 / raw1 = x0(j), x1(j).
template <int NV>
__device__ __forceinline__ double u_fix(const HouseScalars& h, int L, int a, int j, double raw0, double raw1) {
  if (j >= L) return 0.0;
  if (a == 0) return (h.betaA != 0.0) ? raw0 - (j == L - 1 ? h.sA : 0.0) : 0.0;
  if (j >= L - 1 || h.betaB == 0.0) return 0.0;
  return raw1 - h.gammaB * raw0 - (j == L - 2 ? h.sB : 0.0);
}

// K_P role of a mat-vec launch: panel dot products U^T u, W^T u over one row chunk (cg < ncg: panel columns
// [cg*PD_COLS, +PD_COLS)), or (cg == ncg) the reflector store into the panel (both U copies), into `a`, and the
// partial uA.uB of the chunk.  Several GPUs (B = the launch's arguments): the dots run over the rows this rank owns
// (chunks of owned indices; the P shares are added by ka_kernel), the store over all rows -- it also copies the W columns
// that the owners sent with this step's x into the local panel, so that [U | W | U] stays complete on every rank.
template <int NV, bool MG>
__device__ __forceinline__ void kp_role(const RedArgs& R, const KBArgs& B, const HouseScalars& hs, int i, int L, int k, int ncg,
                                        int toprows, int pdr, int chunk, int cg, double* red) {
  // No LDS staging: u_A, u_B are recomputed from the x vectors (L2-hot) next to every U/W load, so the
  // chunk length is free and there are never more than 4 row chunks to re-reduce in K_A.
  const int m = R.m;
  const int ldp = R.ldp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rbase = chunk * pdr;
  double* Up = R.UW;
  double* Wp = R.UW + (size_t)ldp * m;
  double* U2 = R.UW + (size_t)2 * ldp * m;
  if (cg == ncg) {
    // reflector store: panel slots (both U copies), column(s) of `a`, partial uA.uB
    double ab[1] = {0.0};
    const int rend = (rbase + pdr < toprows) ? rbase + pdr : toprows;
    for (int r = rbase + tid; r < rend; r += 256) {
      double x0, x1 = 0.0;
      if (MG) {
        int s_, o_;
        own_of(R, r, s_, o_);
        const double* xm = R.XW + (size_t)B.xpar * R.xpar_stride + (size_t)s_ * R.xmsg_stride + o_;
        x0 = ld_sys(xm);
        if (NV == 2) x1 = ld_sys(xm + R.nown);
        if (B.wk >= 0) {   // rows of the W columns that the previous ka_kernel finished, from their owners
          Wp[(size_t)B.wk * ldp + r] = ld_sys(xm + 2 * (size_t)R.nown);
          if (NV == 2) Wp[(size_t)(B.wk + 1) * ldp + r] = ld_sys(xm + 3 * (size_t)R.nown);
        }
      } else {
        x0 = R.X[r];
        if (NV == 2) x1 = R.X[ldp + r];
      }
      const double uA = u_fix<NV>(hs, L, 0, r, x0, x1);
      const double uB = (NV == 2) ? u_fix<NV>(hs, L, NV - 1, r, x0, x1) : 0.0;
      Up[(size_t)k * ldp + r] = uA;
      U2[(size_t)k * ldp + r] = uA;
      if (NV == 2) { Up[(size_t)(k + 1) * ldp + r] = uB; U2[(size_t)(k + 1) * ldp + r] = uB; }
      if (r < L) {
        if (!MG) {
          R.A[(size_t)i * R.lda + r] = uA;
          if (NV == 2) R.A[(size_t)(i - 1) * R.lda + r] = uB;  // row L-1 gets 0 (src/eigen_prd_t4x.F:333-343)
        } else if (r % R.Px == R.px) {
          // the owners of columns i, i-1 keep their rows of the reflectors (2-D cyclic, as in the reference)
          if (i % R.Py == R.py) R.A[(size_t)(i / R.Py) * R.lda + r / R.Px] = uA;
          if (NV == 2 && (i - 1) % R.Py == R.py) R.A[(size_t)((i - 1) / R.Py) * R.lda + r / R.Px] = uB;
        }
      }
      ab[0] += uA * uB;
    }
    if (NV == 2) {
      block_sum_multi<1>(ab, red);
      if (tid == 0) R.KD[R.kdab_off + chunk] = ab[0];
    }
    return;
  }
  // panel dots: wave w owns panel columns kk0 .. kk0+3; lanes stride the chunk's rows, KPG row groups (10 loads each)
  // per batch.  A/B at N = 8192 in alternating processes: 4 groups instead of 2 -> tridiagonal 213.5 -> 212.6 ms,
  // pentadiagonal 126.4 -> 127.0 ms; the order of the sums does not depend on it.
  // Several GPUs: the row index runs over this rank's owned indices o (global row own_row(o), ascending with o; x from the
  // rank's own slot of the X window, contiguous in o), rend = the owned rows below L.
  constexpr int KPG = (NV == 1) ? 4 : 2;
  const int kk0 = cg * PD_COLS + wave * 4;
  if (kk0 >= k) return;
  const int Lrows = MG ? B.nown_L : L;
  const int rend = (rbase + pdr < Lrows) ? rbase + pdr : Lrows;
  const double* xme = MG ? R.XW + (size_t)B.xpar * R.xpar_stride + (size_t)R.me * R.xmsg_stride : R.X;
  const int xst = MG ? R.nown : ldp;
  double su[4][2], sw[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) { su[c][0] = su[c][1] = sw[c][0] = sw[c][1] = 0.0; }
  for (int r0 = rbase + lane; r0 < rend; r0 += 64 * KPG) {
    double x0[KPG], x1[KPG], tu[KPG][4], tw[KPG][4];
#pragma unroll
    for (int j = 0; j < KPG; ++j) {
      const int ro = (r0 + 64 * j < rend) ? r0 + 64 * j : rbase;
      const int r = MG ? own_row(R, ro) : ro;
      x0[j] = MG ? ld_sys(xme + ro) : xme[ro];
      x1[j] = (NV == 2) ? (MG ? ld_sys(xme + xst + ro) : xme[xst + ro]) : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int kk = (kk0 + c < k) ? kk0 + c : kk0;
        tu[j][c] = Up[(size_t)kk * ldp + r];
        // several GPUs: the W columns that the ka role of THIS launch finished (its plain stores into the panel are not
        // visible across the XCDs before the launch ends) come from the rank's own slot of the X message like x itself
        if (MG && B.wk >= 0 && kk >= B.wk && kk < B.wk + NV) tw[j][c] = ld_sys(xme + (size_t)(2 + kk - B.wk) * xst + ro);
        else tw[j][c] = Wp[(size_t)kk * ldp + r];
      }
    }
#pragma unroll
    for (int j = 0; j < KPG; ++j) {
      const int ro = r0 + 64 * j;
      if (ro < rend) {
        const int r = MG ? own_row(R, ro) : ro;
        const double a = u_fix<NV>(hs, L, 0, r, x0[j], x1[j]);
        const double b = (NV == 2) ? u_fix<NV>(hs, L, NV - 1, r, x0[j], x1[j]) : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          su[c][0] += tu[j][c] * a; sw[c][0] += tw[j][c] * a;
          if (NV == 2) { su[c][1] += tu[j][c] * b; sw[c][1] += tw[j][c] * b; }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    su[c][0] = wave_sum(su[c][0]); sw[c][0] = wave_sum(sw[c][0]);
    if (NV == 2) { su[c][1] = wave_sum(su[c][1]); sw[c][1] = wave_sum(sw[c][1]); }
  }
  if (lane == 0) {
    double* kdp = R.KD + (size_t)chunk * 2 * NV * m;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int kk = kk0 + c;
      if (kk < k) {
        kdp[0 * m + kk] = su[c][0]; kdp[1 * m + kk] = sw[c][0];
        if (NV == 2) { kdp[2 * m + kk] = su[c][1]; kdp[3 * m + kk] = sw[c][1]; }
      }
    }
  }
}

template <int K> struct IC { static constexpr int value = K; };

// UNC: branch-free loads + pinned order (true two-unit pipeline, counted waits); false: the loads of a unit sit behind
// wave-uniform branches and every use waits for everything in flight.  The first form wins where the launch is latency-bound
// (N = 8192: reduction 133.3 -> 131.0 ms), the second where it is bandwidth-bound (N = 32768, same box, alternating
// processes: 4751 ms against 5326 ms with the pipeline -- more requests in flight per CU than the memory system likes);
// the launch picks by active size (g_symv_unc, eigx_tune key 11).
template <int NV, int RB, bool NTL, bool MG, bool UNC>
__device__ __forceinline__ void symv_body(const RedArgs& R, const KBArgs& B, const int bid0, const int nblocks) {
  constexpr int T = 128 * RB;
  // NTL: non-temporal A loads, chosen by the launch for triangles far beyond L2 + Infinity Cache (g_symv_nt)
  constexpr int DYN = 4 * NV * T;
  __shared__ __attribute__((aligned(16))) double dyn[DYN + NV * T];  // [4 waves][NV][T] row sums ; then uc[NV][T]
  __shared__ double red[32];   // [0, 12): block reductions; folded exchange: flags in [0, 2), its scalar reduction in [8, 20)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldp = R.ldp;
  const int L = B.L, i = B.i;
  // 1-D grid: the K_P workgroups [chunk][column group 0..ncg], then the nt(nt+1)/2 tiles of the upper block triangle
  // (row-major): no empty workgroups for the lower triangle
  // (several GPUs: npd x ncg dot workgroups over the rank's own rows, then npd_s store workgroups over all rows)
  const int nkp = MG ? B.npd * B.ncg + B.npd_s : B.npd * (B.ncg + 1);
  const bool panel_role = bid0 < nkp;     // K_P first: its workgroups are the long ones at small L
  if (MG && B.xwait.n > 0) step_wait_fused(B.xwait, bid0 == 0);
#ifdef EIGX_STAMPS
  if (MG && R.dbg && threadIdx.x == 0 && B.i == R.stamp_i) {
    const unsigned long long t_ = (unsigned long long)wall_clock64();
    atomicMax(&R.dbg[3907], t_); atomicMax(&R.dbg[3906], ~t_);
  }
#endif // this step's x (X message) must be in: here, or a wait kernel ran
  const int bid = bid0 - nkp;
  int tyv = 0, txv = 0;
  if (!panel_role && MG) {
    // multi-GPU: tiles of the LOCAL block a(Lr, Lc) that touch the global upper triangle.  Grid order: the tiles of
    // the last (clipped) tile column, then tile column 0, 1, ...; column tx holds mg_nty(tx) tiles, ty = 0 .. nty-1.
    if (bid < B.nty_last) {
      txv = B.ntc - 1;
      tyv = bid;
    } else {
      const int b2 = bid - B.nty_last;
      int tx;
      if (B.slope > 0) {
        // Px divides Py: nty(tx) = slope tx + c1, column tx starts at slope tx (tx - 1) / 2 + c1 tx
        const float sl = (float)B.slope, h = (float)B.c1 - 0.5f * sl;
        tx = (int)((-h + sqrtf(h * h + 2.0f * sl * (float)b2)) / sl);
        if (tx < 0) tx = 0;
        if (tx > B.ntc - 2) tx = B.ntc - 2;
        auto start = [&](int q) { return B.slope * (q * (q - 1) / 2) + B.c1 * q; };
        while (tx > 0 && start(tx) > b2) --tx;
        while (tx < B.ntc - 2 && start(tx + 1) <= b2) ++tx;
        tyv = b2 - start(tx);
      } else {
        int acc = 0;
        tx = 0;
        for (; tx < B.ntc - 2; ++tx) {
          const int c = mg_nty(tx, T, B.Lc, R.Px, R.px, R.Py, R.py);
          if (b2 < acc + c) break;
          acc += c;
        }
        tyv = b2 - acc;
      }
      txv = tx;
    }
  } else if (!panel_role) {
    // row ty starts at ty*nt - ty(ty-1)/2: invert with a float sqrt and fix up by at most one step each way
    const float fn = 2.0f * (float)B.nt + 1.0f;
    int ty = (int)((fn - sqrtf(fn * fn - 8.0f * (float)bid)) * 0.5f);
    if (ty < 0) ty = 0;
    if (ty > B.nt - 1) ty = B.nt - 1;
    while (ty > 0 && ty * B.nt - ty * (ty - 1) / 2 > bid) --ty;
    while ((ty + 1) * B.nt - (ty + 1) * ty / 2 <= bid) ++ty;
    tyv = ty;
    txv = ty + (bid - (ty * B.nt - ty * (ty - 1) / 2));
  } else if (MG) {
    const int q = bid0, nd = B.npd * B.ncg;
    if (q < nd) { tyv = B.nt + q / B.ncg; txv = q - (q / B.ncg) * B.ncg; }   // dots: (row chunk of owned rows, column group)
    else { tyv = B.nt + (q - nd); txv = B.ncg; }                             // store: row chunk of all rows
  } else {
    const int q = bid0;
    tyv = B.nt + q / (B.ncg + 1);   // nt + row chunk
    txv = q - (q / (B.ncg + 1)) * (B.ncg + 1);   // column group
  }

  // ---- SYMV role: issue everything that needs no scalar before the (latency-bound) scalar reduction:
  // raw x values of this tile's columns / this lane's rows and the first 8-column unit of A
  const int ty = tyv, tx = txv;
  const int row0 = ty * T, col0 = tx * T;
  // local <-> global indices (multi-GPU: 2-D cyclic; one GPU: identity) and the local extents below L
  const int Lr = MG ? B.Lr : L, Lc = MG ? B.Lc : L;
  auto grow = [&](int lr) { return MG ? lr * R.Px + R.px : lr; };
  auto gcol = [&](int lc) { return MG ? lc * R.Py + R.py : lc; };
#ifdef EIGX_STAMPS
  const bool stamp_me = (bid0 == nblocks / 2); unsigned long long stamp_prev = __builtin_amdgcn_s_memtime(); (void)stamp_me;
#endif
  const int wcol0 = wave * (T / 4);       // first tile column of this wave
  // Loads that feed the reflector scalars go FIRST (vmcnt retires in order: behind the A-tile loads below they
  // would make the scalar phase wait for HBM); the empty asm keeps the compiler from sinking them.
  const int pivA = L - 1, pivB = L - 2;
  double gpt[2][3];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = (tid + 256 * j < B.ngp) ? tid + 256 * j : 0;
    if (!MG) {
      gpt[j][0] = R.GP[3 * q];
      if (NV == 2) { gpt[j][1] = R.GP[3 * q + 1]; gpt[j][2] = R.GP[3 * q + 2]; }
    } else {
      // several GPUs: one triple per rank (B.ngp = P), behind the four row fields of its X message
      const double* gq = R.XW + (size_t)B.xpar * R.xpar_stride + (size_t)q * R.xmsg_stride + 4 * (size_t)R.nown;
      gpt[j][0] = ld_sys(gq);
      if (NV == 2) { gpt[j][1] = ld_sys(gq + 1); gpt[j][2] = ld_sys(gq + 2); }
    }
  }
  const double x0L = xget<MG>(R, B.xpar, 0, L - 1);
  const double x1L = (NV == 2) ? xget<MG>(R, B.xpar, 1, L - 1) : 0.0;
  const double x1P = (NV == 2 && pivB >= 0) ? xget<MG>(R, B.xpar, 1, pivB) : 0.0;
  const double x0P = (NV == 2 && pivB >= 0) ? xget<MG>(R, B.xpar, 0, pivB) : 0.0;
  asm volatile("" ::: "memory");
  double craw[NV][(T + 255) / 256];       // raw x at the tile's columns (thread t -> column t, t+256)
  double rraw[NV][RB][2];                 // raw x at this lane's rows
  double2 av0[8], av1[8];
  if (!panel_role) {
#pragma unroll
    for (int q = 0; q < (T + 255) / 256; ++q) {
      const int c = col0 + tid + 256 * q;
      const bool ok = (tid + 256 * q < T) && c < Lc;
      const int gc = ok ? gcol(c) : 0;
      craw[0][q] = ok ? xget<MG>(R, B.xpar, 0, gc) : 0.0;
      if (NV == 2) craw[NV - 1][q] = ok ? xget<MG>(R, B.xpar, 1, gc) : 0.0;
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int r0 = row0 + rb * 128 + lane * 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool ok = r0 + h < Lr;
        const int gr = ok ? grow(r0 + h) : 0;
        rraw[0][rb][h] = ok ? xget<MG>(R, B.xpar, 0, gr) : 0.0;
        if (NV == 2) rraw[NV - 1][rb][h] = ok ? xget<MG>(R, B.xpar, 1, gr) : 0.0;
      }
    }
    {
      // (unconditional loads: see load8 below)
      const int r0 = row0 + lane * 2;
      const bool rok = r0 < Lr;
      const double* Ap = (rok || !UNC) ? R.A + (rok ? r0 : 0) : R.zero16;
      const size_t cstride = (rok || !UNC) ? (size_t)R.lda : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = col0 + wcol0 + j;
        if (UNC) {
          const int cc = (c < Lc) ? c : Lc - 1;
          av0[j] = ld2<NTL>(Ap + (size_t)cc * cstride);
        } else {
          if (rok && c < Lc) av0[j] = ld2<NTL>(Ap + (size_t)c * R.lda);
          else av0[j] = make_double2(0.0, 0.0);
        }
      }
    }
  }

#ifdef EIGX_STAMPS
  if (!panel_role) { if (R.dbg && threadIdx.x == 0 && stamp_me) { const unsigned long long _t = __builtin_amdgcn_s_memtime(); atomicAdd(&R.dbg[8], _t - stamp_prev); stamp_prev = _t; } }
#endif
  // ---- reflector scalars (every workgroup, same order) -------------------------------------------
  // NV = 1: s = -sign(||x||, x_piv), beta = ||x||^2 - s x_piv from the Gram partials of K_A.
  // NV = 2: two sequential Householder steps on the column pair (x0 = column i, x1 = column i-1) without a
  // kernel in between: reflector A from g00 = x0.x0; gamma = uA.x1 / betaA from g01; the reflected second
  // column x1' = x1 - gamma uA is formed on the fly wherever it is needed; its norm above the pivot row
  // follows from the isometry of H_A: ||x1'(0:L-1)||^2 = g11 - x1'(L-1)^2.  If that difference cancels
  // (more than 3/4 of the column's weight in the pivot row) the sum is taken explicitly instead -- same
  // order in every workgroup, so the replicas stay bit-identical either way.
  double sA, sB = 0.0, betaA, betaB = 0.0, gammaB = 0.0, eL1 = 0.0;
  {
    HouseScalars hs = house_scalars<NV, MG>(R, B.ngp, L, gpt, x0L, x1L, x0P, x1P, red, B.xpar);
    sA = hs.sA; sB = hs.sB; betaA = hs.betaA; betaB = hs.betaB; gammaB = hs.gammaB; eL1 = hs.eL1;
  }
  // the store-role panel workgroup of chunk 0 publishes the scalars (it exists on every rank)
  if (panel_role && tx == B.ncg && ty == B.nt && tid == 0) {
    R.sc[SC_SA] = sA; R.sc[SC_BETA_A] = betaA;
    if (NV == 1) {
      R.e[i] = sA;                               // e(i,1) = T(i-1,i)
    } else {
      R.sc[SC_SB] = sB; R.sc[SC_BETA_B] = betaB;
      R.e[R.lde + i] = sA;                       // e(i,2)   = T(i-2,i)
      R.e[i - 1] = eL1;                          // e(i-1,1) = T(i-2,i-1)
      if (i - 1 >= 2) R.e[R.lde + i - 1] = sB;   // e(i-1,2) = T(i-3,i-1)
    }
    if (MG) {
      // several GPUs: the diagonal block of the band that ka_kernel writes on one GPU, on every rank from the X message
      R.d[i] = xget<MG>(R, B.xpar, 0, i);
      if (NV == 2) { R.e[i] = xget<MG>(R, B.xpar, 0, i - 1); R.d[i - 1] = xget<MG>(R, B.xpar, 1, i - 1); }   // e(i,1) = A_eff(i-1,i)
    }
  }
  // u_A(j) = x0(j) - [j == pivA] sA ;  u_B(j) = x1'(j) - [j == pivB] sB, u_B(L-1) = 0 ;
  // zero when the reflector is trivial or j >= L.  raw0/raw1 = x0(j), x1(j).
  auto ufix2 = [&](int a, int j, double raw0, double raw1) -> double {
    if (j >= L) return 0.0;
    if (a == 0) return (betaA != 0.0) ? raw0 - (j == pivA ? sA : 0.0) : 0.0;
    if (j >= L - 1 || betaB == 0.0) return 0.0;
    return raw1 - gammaB * raw0 - (j == pivB ? sB : 0.0);
  };
  if (panel_role) {
    // ================================================================ K_P
    HouseScalars hs;
    hs.sA = sA; hs.sB = sB; hs.betaA = betaA; hs.betaB = betaB; hs.gammaB = gammaB; hs.eL1 = eL1;
    kp_role<NV, MG>(R, B, hs, i, L, B.k, B.ncg, B.toprows, (MG && tx == B.ncg) ? B.pdr_s : B.pdr, ty - B.nt, tx, red);
    return;
  }

  // ================================================================== K_B (SYMV tile)
  EIGX_STAMP(9);
  // tiles that the diagonal crosses mask element-wise; a tile is interior iff its last global row < its first global column
  const bool diag = MG ? (grow(row0 + T - 1) >= gcol(col0)) : (tx == ty);
  double* yrs = dyn;            // [4 waves][NV][T]
  double* ucs = dyn + DYN;      // [NV][T] : u_a at the tile's columns
#pragma unroll
  for (int q = 0; q < (T + 255) / 256; ++q) {
    const int t = tid + 256 * q;
    if (t < T) {
#pragma unroll
      for (int a = 0; a < NV; ++a) ucs[a * T + t] = (col0 + t < Lc) ? ufix2(a, gcol(col0 + t), craw[0][q], craw[NV - 1][q]) : 0.0;
    }
  }
  __syncthreads();

  double ux[NV][RB][2], yr[NV][RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int r0 = row0 + rb * 128 + lane * 2;
#pragma unroll
    for (int a = 0; a < NV; ++a) {
      ux[a][rb][0] = (r0 < Lr) ? ufix2(a, grow(r0), rraw[0][rb][0], rraw[NV - 1][rb][0]) : 0.0;
      ux[a][rb][1] = (r0 + 1 < Lr) ? ufix2(a, grow(r0 + 1), rraw[0][rb][1], rraw[NV - 1][rb][1]) : 0.0;
      yr[a][rb][0] = 0.0; yr[a][rb][1] = 0.0;
    }
  }
  constexpr int NG = T / 32;              // column groups of 8 per wave
  double yc[NV][8];
  double sp[3] = {0.0, 0.0, 0.0};

  // Every load of a unit is issued UNCONDITIONALLY: a branch around a load makes the compiler lose count of what is in
  // flight and wait for everything (s_waitcnt vmcnt(0)) at each use, i.e. the unit requested ahead would have to arrive
  // before the current one is consumed.  Lanes whose rows lie beyond the active block read a page of zeros; columns beyond
  // it re-read the last active column: those are genuine upper-triangle elements (finite), and they only ever meet a zero
  // u (column >= Lc), a select by index (tiles on the diagonal), or sums that are not stored.
  auto load8 = [&](double2 (&av)[8], int g, int rb) {
    const int r0 = row0 + rb * 128 + lane * 2;
    const bool rok = r0 < Lr;
    const double* Ap = (rok || !UNC) ? R.A + (rok ? r0 : 0) : R.zero16;      // (base and column stride per lane: no select at the loads)
    const size_t cstride = (rok || !UNC) ? (size_t)R.lda : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = col0 + wcol0 + g * 8 + j;
      if (UNC) {
        const int cc = (c < Lc) ? c : Lc - 1;
        av[j] = ld2<NTL>(Ap + (size_t)cc * cstride);
      } else {
        if (rok && c < Lc) av[j] = ld2<NTL>(Ap + (size_t)c * R.lda);
        else av[j] = make_double2(0.0, 0.0);
      }
    }
  };
  // the unit after the last one: UNC re-reads the last unit (unused) instead of branching around the loads
  auto load8_next = [&](double2 (&av)[8], int gnext, int glast, bool more) {
    if (UNC) load8(av, more ? gnext : glast, 0);
    else if (more) load8(av, gnext, 0);
  };
  // one unit = (column group g, row block rb): accumulate; after the last row block reduce the 8 column sums
  auto compute8 = [&](const double2 (&av)[8], int g, auto rbc) {
    constexpr int rb = decltype(rbc)::value;
    const int r0 = row0 + rb * 128 + lane * 2, r1 = r0 + 1;
    const int tc0 = wcol0 + g * 8;
    if (rb == 0) {
#pragma unroll
      for (int a = 0; a < NV; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) yc[a][j] = 0.0;
    }
    if (EIGX_ABL(64)) {   // diagnostic build: loads only (what the load loop alone sustains)
#pragma unroll
      for (int j = 0; j < 8; ++j) yr[0][rb][0] += av[j].x + av[j].y;
      return;
    }
    if (!diag) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int a = 0; a < NV; ++a) {
          const double uc = ucs[a * T + tc0 + j];
          yr[a][rb][0] += av[j].x * uc;
          yr[a][rb][1] += av[j].y * uc;
          yc[a][j] += av[j].x * ux[a][rb][0] + av[j].y * ux[a][rb][1];
        }
      }
    } else {
      const int g0 = grow(r0), g1 = MG ? g0 + R.Px : r1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = gcol(col0 + tc0 + j);
        const bool cin = col0 + tc0 + j < Lc;   // (columns beyond the block were loaded from the last active column: drop them by index too)
        const double ax_s = (cin && g0 < c) ? av[j].x : 0.0, ay_s = (cin && g1 < c) ? av[j].y : 0.0;     // strict upper
        const double ax_d = (cin && g0 <= c) ? av[j].x : 0.0, ay_d = (cin && g1 <= c) ? av[j].y : 0.0;   // with diagonal
#pragma unroll
        for (int a = 0; a < NV; ++a) {
          const double uc = ucs[a * T + tc0 + j];
          yr[a][rb][0] += ax_s * uc;
          yr[a][rb][1] += ay_s * uc;
          yc[a][j] += ax_d * ux[a][rb][0] + ay_d * ux[a][rb][1];
        }
      }
    }
    if (rb == RB - 1) {
      // halving butterfly: 8 column sums over 64 lanes in 4+2+1+3 shuffles
      double fin[NV];
#pragma unroll
      for (int a = 0; a < NV; ++a) {
        double v4[4], v2[2], v1;
        const bool hi8 = lane & 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) v4[j] = swapadd32(yc[a][j], yc[a][j + 4]);   // lanes < 32 keep column j, the others j+4
#pragma unroll
        for (int j = 0; j < 2; ++j) v2[j] = swapadd16(v4[j], v4[j + 2]);         // even rows keep j, odd rows j+2
        {
          const double keep = hi8 ? v2[1] : v2[0];
          const double send = hi8 ? v2[0] : v2[1];
          v1 = keep + dpp_mov<0x128>(send);     // row_ror:8 = lane ^ 8 within the row of 16
        }
        // the 8 lanes of a group all end up with the group's total (quad_perm x2, row_half_mirror): VALU only
        v1 = dpp_add<0xB1>(v1);
        v1 = dpp_add<0x4E>(v1);
        v1 = dpp_add<0x141>(v1);
        fin[a] = v1;
      }
      // lane with (lane&7)==0 holds column j = 4*bit5 + 2*bit4 + bit3 : complete over the tile's rows
      if ((lane & 7) == 0) {
        const int j = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
        const int c = col0 + tc0 + j;
        if (c < Lc) {
#pragma unroll
          for (int a = 0; a < NV; ++a) {
            R.YC[((size_t)ty * NV + a) * ldp + c] = fin[a];
          }
          const double ua = ucs[tc0 + j];
          sp[0] += ua * fin[0];
          if (NV == 2) { sp[1] += ua * fin[NV - 1]; sp[2] += ucs[(NV - 1) * T + tc0 + j] * fin[NV - 1]; }
        }
      }
    }
  };

  // software pipeline, two units in flight, all register indices static.  The empty asm statements pin the order
  // "request the next unit, THEN consume the current one": without them the scheduler sinks the (now branch-free)
  // loads down to their first use to shorten live ranges, and nothing is in flight while a unit is consumed.
#define EIGX_PIN do { if (UNC) asm volatile("" ::: "memory"); } while (0)
  {
    // av0 holds unit (g = 0, rb = 0), loaded at kernel entry
    if (RB == 1) {
      for (int g = 0; g < NG; g += 2) {
        load8(av1, g + 1, 0);
        EIGX_PIN;
        compute8(av0, g, IC<0>());
        load8_next(av0, g + 2, g + 1, g + 2 < NG);
        EIGX_PIN;
        compute8(av1, g + 1, IC<0>());
      }
    } else if (RB == 2) {
      const int ng = B.ng;
#pragma unroll 1
      for (int g = 0; g < ng; ++g) {
        load8(av1, g, 1);
        EIGX_PIN;
        compute8(av0, g, IC<0>());
        load8_next(av0, g + 1, g, g + 1 < ng);
        EIGX_PIN;
        compute8(av1, g, IC<(RB > 1 ? 1 : 0)>());
      }
    } else {
      const int ng = B.ng;
#pragma unroll 1
      for (int g = 0; g < ng; ++g) {
        load8(av1, g, 1);
        EIGX_PIN;
        compute8(av0, g, IC<0>());
        load8(av0, g, 2);
        EIGX_PIN;
        compute8(av1, g, IC<(RB > 1 ? 1 : 0)>());
        load8(av1, g, 3);
        EIGX_PIN;
        compute8(av0, g, IC<(RB > 2 ? 2 : 0)>());
        load8_next(av0, g + 1, g, g + 1 < ng);
        EIGX_PIN;
        compute8(av1, g, IC<(RB > 3 ? 3 : 0)>());
      }
    }
  }

  EIGX_STAMP(10);
  if (EIGX_ABL(256)) {   // diagnostic build: no epilogue
    if (yr[0][0][0] == 1.2345678) R.SP[0] = yr[0][0][1] + yr[NV - 1][RB - 1][0];
    return;
  }
  // ---- row sums: combine the 4 waves through LDS; bilinear row part ----------------------------------
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int a = 0; a < NV; ++a) {
      double* dst = yrs + ((size_t)wave * NV + a) * T + rb * 128 + lane * 2;
      dst[0] = yr[a][rb][0];
      dst[1] = yr[a][rb][1];
    }
    sp[0] += ux[0][rb][0] * yr[0][rb][0] + ux[0][rb][1] * yr[0][rb][1];
    if (NV == 2) {
      sp[1] += ux[0][rb][0] * yr[NV - 1][rb][0] + ux[0][rb][1] * yr[NV - 1][rb][1];
      sp[2] += ux[NV - 1][rb][0] * yr[NV - 1][rb][0] + ux[NV - 1][rb][1] * yr[NV - 1][rb][1];
    }
  }
  __syncthreads();
  for (int rep_ = 0; rep_ < (EIGX_ABL(4096) ? 2 : 1); ++rep_) {   // diagnostic build: the combine + stores twice
  if (rep_) {
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int a = 0; a < NV; ++a) {
        double* dst = yrs + ((size_t)wave * NV + a) * T + rb * 128 + lane * 2;
        dst[0] = yr[a][rb][0];
        dst[1] = yr[a][rb][1];
      }
    __syncthreads();
  }
  if (!EIGX_ABL(1024))
  for (int t = tid; t < T; t += 256) {
    const int r = row0 + t;
    if (r < Lr) {
#pragma unroll
      for (int a = 0; a < NV; ++a) {
        const double s = (yrs[((size_t)0 * NV + a) * T + t] + yrs[((size_t)1 * NV + a) * T + t]) +
                         (yrs[((size_t)2 * NV + a) * T + t] + yrs[((size_t)3 * NV + a) * T + t]);
        R.YR[((size_t)tx * NV + a) * ldp + r] = s;
      }
    }
  }
  }
  if (EIGX_ABL(8192)) {   // diagnostic build: the block reduction twice
    double s2[3] = {sp[0], sp[1], sp[2]};
    block_sum_multi<3>(s2, red);
    __syncthreads();
    if (s2[0] == 1.2345678) sp[0] = s2[1];
  }
  if (!EIGX_ABL(512)) block_sum_multi<3>(sp, red);
  if (tid == 0) {
    const size_t w = (size_t)ty * R.maxseg + tx;
    R.SP[w * 3 + 0] = sp[0]; R.SP[w * 3 + 1] = sp[1]; R.SP[w * 3 + 2] = sp[2];
  }
  EIGX_STAMP(11);
#ifdef EIGX_STAMPS
  if (R.dbg && threadIdx.x == 0 && stamp_me) atomicAdd(&R.dbg[15], 1ull);
  if (MG && R.dbg && threadIdx.x == 0 && B.i == R.stamp_i) {
    const unsigned long long t_ = (unsigned long long)wall_clock64();
    atomicMax(&R.dbg[3908], t_); atomicMax(&R.dbg[3909], ~t_); atomicAdd(&R.dbg[3910], 1ull);
  }
#endif
}

template <int NV, int RB, bool NTL, bool MG, bool UNC>
__global__ __launch_bounds__(256) void symv_kernel(RedArgs R, KBArgs B) {
  symv_body<NV, RB, NTL, MG, UNC>(R, B, blockIdx.x, gridDim.x);
}

// Several GPUs: ONE launch per step.  Roles in dispatch order (a 1-D grid is dispatched in workgroup order, so a role never
// waits for a workgroup that could be kept off the CUs by a later one):
//   [0, nkl)          kl_body of the PREVIOUS step: sums of this rank's tile partial sums to the row owners (Y message);
//   [nkl, nkl + nka)  ka_body: spins until every rank's Y message is in, finishes W and forms x for this rank's rows,
//                     sends them to everybody (X message);
//   the rest          symv_body of THIS step (K_P workgroups, then tiles): spins until every rank's X message is in.
// Between the ranks the order is enforced by the flags, inside a rank by the flags as well (its own messages count): what
// kl reads (the previous launch's partial sums) is complete at the launch boundary and is not overwritten before the X
// flags -- which follow the Y flags, which this rank raises after its last kl workgroup has read.  Against three launches
// with a wait each, this saves two launch boundaries per step and lets the consumers' local prologue loads run while
// they wait.  With wait kernels (ranks sharing a card: tests) or the collective exchanges the same kernel is launched
// role by role (the other counts zero).
struct MGStep { int nkl, nka; StepPeers xpeers; };
template <int NV, int RB, bool NTL, bool UNC>
__global__ __launch_bounds__(256) void mg_step_kernel(RedArgs R, KLArgs KL, KAArgs S, KBArgs B, MGStep M) {
  int bid = blockIdx.x;
#ifdef EIGX_STAMPS
  if (R.dbg && bid == 0 && threadIdx.x == 0 && M.nkl > 0 && M.nka > 0 && (int)gridDim.x > M.nkl + M.nka) {
    // timeline of a full step launch on the 100-MHz wall clock: workgroup 0's entry time; its distance to the previous one
    const unsigned long long t = (unsigned long long)wall_clock64();
    const unsigned long long prev = R.dbg[20];
    if (prev != 0 && t - prev < 100000ull) { atomicAdd(&R.dbg[21], t - prev); atomicAdd(&R.dbg[24], 1ull); }
    __hip_atomic_store(&R.dbg[20], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
  if (bid < M.nkl) { kl_body<NV>(R, KL, bid, M.nkl); return; }
  bid -= M.nkl;
  if (bid < M.nka) { ka_body<NV, true, false, 1, 8, 8>(R, S, &M.xpeers, bid, M.nka); return; }
  bid -= M.nka;
  symv_body<NV, RB, NTL, true, UNC>(R, B, bid, (int)gridDim.x - M.nkl - M.nka);
}

// zero-fill helper
__global__ void fill_kernel(double* p, size_t n, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---- multi-GPU helpers ---------------------------------------------------------------------------------------
// compact copies of the panel for the local trailing update: UWr(li, :) = [U | W](global row of li, :),
// UWc(lj, :) = [W | U](global row = global column of lj, :)   (the reference keeps the same four arrays: ur, vr
// row-distributed and uyr, vyr column-distributed, src/eigen_t1.F:68-309)
__global__ void compact_panel_kernel(const double* __restrict__ UW, int ldp, int m, int nrl, int ncl, int Px, int px,
                                     int Py, int py, double* __restrict__ UWr, int ldr, double* __restrict__ UWc, int ldc) {
  const int k = blockIdx.y;                    // 0 .. 2m-1
  const double* rsrc = UW + (size_t)k * ldp;         // [U | W] column k
  const double* csrc = UW + (size_t)(k + m) * ldp;   // [W | U] column k
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < (nrl > ncl ? nrl : ncl); l += gridDim.x * blockDim.x) {
    if (l < nrl) UWr[(size_t)k * ldr + l] = rsrc[(size_t)l * Px + px];
    if (l < ncl) UWc[(size_t)k * ldc + l] = csrc[(size_t)l * Py + py];
  }
}

// my rows of my columns of the panel, columns [clo, chi] (global), rows < toprows (global): send[ljr*nxs + li]
__global__ void pack_panel_kernel(const double* __restrict__ A, int lda, int lj0, int mloc, int nrl, int nxs,
                                  double* __restrict__ send) {
  const int ljr = blockIdx.y;
  if (ljr >= mloc) return;
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < nrl; li += gridDim.x * blockDim.x)
    send[(size_t)ljr * nxs + li] = A[(size_t)(lj0 + ljr) * lda + li];
}
// PAN(r, c - clo) = the owner's copy:  recv[src][ (c/Py - lj0(src's py)) * nxs + r/Px ]
__global__ void unpack_panel_kernel(const double* __restrict__ recv, size_t count, int nxs, int clo, int chi, int toprows,
                                    int Px, int Py, int row_major, double* __restrict__ PAN, int ldpan) {
  const int c = clo + blockIdx.y;
  if (c > chi) return;
  const int qy = c % Py;
  const int lj0 = (clo - qy + Py - 1) / Py;
  const int ljr = c / Py - lj0;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < toprows; r += gridDim.x * blockDim.x) {
    const int qx = r % Px;
    const int src = row_major ? qx * Py + qy : qx + qy * Px;
    PAN[(size_t)(c - clo) * ldpan + r] = ld_sys(recv + (size_t)src * count + (size_t)ljr * nxs + r / Px);
  }
}

// several GPUs, once per panel: the W columns [wk, wk + NB) that the panel-closing ka_kernel finished, rows < rows, from
// their owners' X messages (parity xpar) into the local panel -- the trailing update needs the complete [U | W | U]
template <int NB>
__global__ void mg_wcopy_kernel(RedArgs R, int xpar, int wk, int rows, StepWait W) {
  if (W.n > 0) step_wait_fused(W, blockIdx.x == 0);
  double* Wp = R.UW + (size_t)R.ldp * R.m;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    int s_, o_;
    own_of(R, r, s_, o_);
    const double* xm = R.XW + (size_t)xpar * R.xpar_stride + (size_t)s_ * R.xmsg_stride + o_;
    Wp[(size_t)wk * R.ldp + r] = ld_sys(xm + 2 * (size_t)R.nown);
    if (NB == 2) Wp[(size_t)(wk + 1) * R.ldp + r] = ld_sys(xm + 3 * (size_t)R.nown);
  }
}
// several GPUs, once per reduction: the last diagonal block (the final ka_kernel formed its <= NB columns, no mat-vec
// follows that would publish them): d(i), and for two columns e(i,1) = A_eff(i-1,i), d(i-1)
template <int NB>
__global__ void mg_tail_kernel(RedArgs R, int xpar, int i, int ncols, StepWait W) {
  if (W.n > 0) step_wait_fused(W, blockIdx.x == 0);
  if (threadIdx.x == 0 && ncols > 0) {
    R.d[i] = xget<true>(R, xpar, 0, i);
    if (ncols > 1) { R.e[i] = xget<true>(R, xpar, 0, i - 1); R.d[i - 1] = xget<true>(R, xpar, 1, i - 1); }
  }
}

template <int NB>
void band_reduce_impl(Context& ctx, int n, double* A, int lda, double* d, double* e, int lde, int m) {
  hipStream_t st = ctx.stream;
  if (m < NB) m = NB;
  if (m > 256) m = 256;
  if (NB == 2 && (m & 1)) ++m;
  const int ldp = pad_ld((n + 127) / 128 * 128 + 128);
  const Grid& G = ctx.grid;
  const bool mg = G.nranks > 1;
  RedArgs R;
  R.A = A; R.lda = lda; R.n = n; R.ldp = ldp; R.m = m;
  R.d = d; R.e = e; R.lde = lde;
  R.P = G.nranks; R.me = G.rank; R.invP = 1.0f / (float)G.nranks;
  R.Px = G.Px; R.Py = G.Py; R.px = G.px; R.py = G.py; R.row_major = G.row_major;
  R.nxs = (ceil_div(n, G.Px) + 7) / 8 * 8;
  R.nys = (ceil_div(n, G.Py) + 7) / 8 * 8;
  R.msg_stride = NB * (R.nxs + R.nys) + 8 + 2 * NB * m;          // Y: row sums | column sums | 8 scalar slots | share of the panel dots
  R.ypar_stride = (size_t)G.nranks * R.msg_stride; R.ysrc_stride = R.msg_stride;
  R.nown = ceil_div(ceil_div(n, KA_ROWS), G.nranks) * KA_ROWS;   // rows a rank can own (groups of KA_ROWS dealt round-robin)
  R.xmsg_stride = 4 * R.nown + 8;                                // X: x_i | x_{i-1} | W_A | W_B (owned rows) | Gram partial sums
  R.xpar_stride = (size_t)G.nranks * R.xmsg_stride;
  R.MSG = nullptr; R.XW = nullptr; R.PAN = nullptr; R.ldpan = ldp;
  int maxseg = 0;
  if (!mg) {
    for (int L = n; L >= 1; --L) {  // nt is not monotone in L: scan
      const SymvGeom g = symv_geom(L);
      if (g.nt > maxseg) maxseg = g.nt;
    }
  } else {
    maxseg = ceil_div(R.nxs > R.nys ? R.nxs : R.nys, 128) + 1;
  }
  maxseg += 1;
  R.maxseg = maxseg;
  R.maxrs = maxseg;
  R.maxchunk = 4 + 1;
  // + 128 columns: K_A loads panel columns kk < 128 of the U and of the W region without clamping kk (masked afterwards)
  R.UW = ctx.pool.get_t<double>("red.UW", (size_t)ldp * (m * 3 + 128));
  R.X = ctx.pool.get_t<double>("red.X", (size_t)ldp * 3);
  if (!mg) {
    // One GPU: tile (ty, tx), tx >= ty, writes its column sums to slot ty and its row sums to slot tx + 1 of ONE array
    // Y[slot][vector][ldp] (YR = YC + one slot): the partial sums of a row r in tile row ty(r) are then simply the
    // slots 0 .. nt -- slots t <= ty(r) hold column results of tiles (t, ty), slots t > ty(r) row results of tiles
    // (ty, t-1), each written by exactly one tile -- and K_A addresses them with one uniform stride, no select.
    // 161 slots at least: K_A's first batch loads slots < RPB * KA_SL = 160 unclamped.
    const int nslot = (maxseg + 2 > 162) ? maxseg + 2 : 162;
    R.YC = ctx.pool.get_t<double>("red.Y", (size_t)nslot * NB * ldp);
    R.YR = R.YC + (size_t)NB * ldp;
  } else {
    R.YR = ctx.pool.get_t<double>("red.YR", (size_t)maxseg * NB * ldp);
    R.YC = ctx.pool.get_t<double>("red.YC", (size_t)R.maxrs * NB * ldp);
  }
  R.kdab_off = R.maxchunk * 2 * NB * m;
  R.KD = ctx.pool.get_t<double>("red.KD", (size_t)R.kdab_off + R.maxchunk + 8 + 512);   // + slack: K_A loads kk < 256 unclamped
  // several GPUs: first-level panel dots of up to 12 chunks of the rank's own rows (kl_kernel sums them and sends the
  // rank's share to everybody), then the uA.uB partial sums of up to 4 P + 4 chunks of the replicated reflector store
  const int maxchunk2 = 12;
  const int kdab2_off = maxchunk2 * 2 * NB * m;
  if (mg) {
    R.KD = ctx.pool.get_t<double>("red.KD2", (size_t)kdab2_off + 4 * G.nranks + 4 + 8 + 512);
    R.kdab_off = kdab2_off;
  }
  const size_t sp_count = (size_t)(maxseg * maxseg) * 3 + 8;
  R.SP = ctx.pool.get_t<double>("red.SP", sp_count);
  const int maxgp = (n + KA_ROWS - 1) / KA_ROWS + 2;
  R.gp2_off = 0;
  R.GP = ctx.pool.get_t<double>("red.GP", (size_t)maxgp * 3 + 8);
  R.sc = ctx.pool.get_t<double>("red.sc", SC_COUNT + 8);
  R.zero16 = R.sc + SC_COUNT;             // zero-filled below with the scalars, never written afterwards
  R.dbg = nullptr;
  R.abl = getenv("EIGX_ABL") ? atoi(getenv("EIGX_ABL")) : 0;
  R.stamp_i = getenv("EIGX_STAMP_I") ? atoi(getenv("EIGX_STAMP_I")) : -1;
#ifdef EIGX_STAMPS
  R.dbg = ctx.pool.get_t<unsigned long long>("red.dbg", 4096);
  EIGX_HIP_CHECK(hipMemsetAsync(R.dbg, 0, 4096 * sizeof(unsigned long long), st));
#endif
  // ---- multi-GPU state: step window, gathered panel, compact panels ------------------------------------------
  StepPeers peers, xpeers;
  unsigned long long epoch = 0;          // epoch of the Y message that the NEXT K_A consumes
  unsigned long long xepoch = 0;         // epoch of the X message that the last K_A launch wrote
  PeerBuf* panr = nullptr;               // receive window of the panel gather: [rank][mloc_max][nxs]
  double *pan = nullptr, *pan_send = nullptr, *UWr = nullptr, *UWc = nullptr;
  // a gathered panel holds the m columns of the next panel and, when fewer than NB + 1 columns would remain below it,
  // those too (they are finished without another trailing update)
  auto panel_lo = [&](int itop) { const int clo = itop - m + 1; return clo <= NB ? 0 : clo; };
  const int mloc_max = ceil_div(m + NB, G.Py) + 1;
  const size_t pan_count = (size_t)mloc_max * R.nxs;
  const int ldr = pad_ld(R.nxs + 2), ldc = pad_ld(R.nys + 2);
  if (mg) {
    R.MSG = comm_step_window(ctx, 0, (size_t)R.msg_stride, &peers);
    R.ysrc_stride = (int)peers.src_stride;
    R.XW = comm_step_window(ctx, 1, (size_t)R.xmsg_stride, &xpeers);
    epoch = comm_step_epoch_base(ctx, 0, (unsigned long long)(n / NB + 2));
    xepoch = comm_step_epoch_base(ctx, 1, (unsigned long long)(n / NB + n / m + 8));
    panr = comm_buffer(ctx, "red.panr", (size_t)G.nranks * pan_count * sizeof(double));
    pan = ctx.pool.get_t<double>("red.pan", (size_t)ldp * (m + NB));
    pan_send = ctx.pool.get_t<double>("red.pansend", pan_count);
    UWr = ctx.pool.get_t<double>("red.UWr", (size_t)ldr * 2 * m);
    UWc = ctx.pool.get_t<double>("red.UWc", (size_t)ldc * 2 * m);
    R.PAN = pan;
    EIGX_HIP_CHECK(hipMemsetAsync(pan_send, 0, pan_count * sizeof(double), st));
  }
  // gather the panel columns [clo, chi], rows < chi + 1, from their owners into `pan` (enqueued on stream s)
  auto gather_panel = [&](int clo, int chi, hipStream_t s, CommChannel ch) {
    const int toprows = chi + 1;
    const int nxc = (ceil_div(toprows, G.Px) + 7) / 8 * 8;       // row stride of this gather (same on every rank)
    const size_t cnt = (size_t)mloc_max * nxc;
    const int lj0 = (clo - G.py + G.Py - 1) / G.Py;
    const int lj1 = chi >= G.py ? (chi - G.py) / G.Py : -1;
    const int mloc = lj1 - lj0 + 1;
    const int nrl = local_count(toprows, G.Px, G.px);
    if (mloc > 0 && nrl > 0)
      hipLaunchKernelGGL(pack_panel_kernel, dim3(ceil_div(nrl, 256) < 64 ? ceil_div(nrl, 256) : 64, mloc), dim3(256), 0, s,
                         (const double*)A, lda, lj0, mloc, nrl, nxc, pan_send);
    comm_exchange(ctx, COMM_WORLD, pan_send, 0, panr, 0, cnt, s, ch);
    hipLaunchKernelGGL(unpack_panel_kernel, dim3(ceil_div(toprows, 256) < 64 ? ceil_div(toprows, 256) : 64, chi - clo + 1),
                       dim3(256), 0, s, (const double*)panr->local, cnt, nxc, clo, chi, toprows, G.Px, G.Py,
                       G.row_major, pan, ldp);
  };
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, R.UW, (size_t)ldp * m * 3, 0.0);
  hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, st, e, (size_t)lde * NB, 0.0);
  hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, st, R.sc, (size_t)SC_COUNT + 8, 0.0);
  hipLaunchKernelGGL(fill_kernel, dim3(16), dim3(256), 0, st, R.SP, sp_count, 0.0);

  const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  KAArgs S;
  // ka_kernel's first batches of loads are unconditional (clamped), so their sizes are template parameters matched to
  // the step: partial sums of a row (nt + 1 slots: 1 / 2 / 3 / 5 / 10 batches of KA_SL = 16), folded rows of tile
  // scalars (nt <= 15 / 31 / 63: 2 / 4 / 8 per wave), panel columns (k <= 32 / 64 / more: 2 / 4 / 8 per slice)
  auto launch_ka = [&](int nwg, const KAArgs& K) {   // (one GPU)
    const bool fit = g_ka_fit != 0;
    const int nslot = fit ? K.nt_prev + 1 : 1 << 30, ntp = fit ? K.nt_prev : 1 << 30;
    const int kk = fit ? (K.has_prev ? K.kprev : K.k) : 1 << 30;
#define EIGX_KA3(RPBV, SPBV, KBV)                                                                                        \
    do {                                                                                                                \
      if (K.G > 1) hipLaunchKernelGGL((ka_kernel<NB, false, true, RPBV, SPBV, KBV>), dim3(nwg), dim3(256), 0, st, R, K); \
      else hipLaunchKernelGGL((ka_kernel<NB, false, false, RPBV, SPBV, KBV>), dim3(nwg), dim3(256), 0, st, R, K);        \
    } while (0)
#define EIGX_KA2(RPBV, SPBV)                                                                                             \
    do {                                                                                                                \
      if (kk <= 2 * KA_SL) EIGX_KA3(RPBV, SPBV, 2);                                                                      \
      else if (kk <= 4 * KA_SL) EIGX_KA3(RPBV, SPBV, 4);                                                                 \
      else EIGX_KA3(RPBV, SPBV, 8);                                                                                      \
    } while (0)
    if (nslot <= 1 * KA_SL && ntp <= 15) EIGX_KA2(1, 2);
    else if (nslot <= 2 * KA_SL && ntp <= 31) EIGX_KA2(2, 4);
    else if (nslot <= 3 * KA_SL) EIGX_KA2(3, 8);
    else if (nslot <= 5 * KA_SL) EIGX_KA2(5, 8);
    else EIGX_KA2(10, 8);
#undef EIGX_KA2
#undef EIGX_KA3
  };
  S.has_prev = 0; S.iprev = 0; S.Lprev = 0; S.kprev = 0; S.nchunk_prev = 0; S.nt_prev = 0; S.lgT_prev = 7;
  S.par = 0; S.pan_c0 = 0; S.G = 1;
  S.wait.n = 0; S.wait.flag = nullptr; S.wait.err = nullptr; S.wait.ticks = nullptr; S.wait.limit_ticks = 0; S.wait.epoch = 0; S.wait.naps = 1;
  S.nchunk_ab = 0; S.xpar = 0; S.xepoch = 0;
  const StepWait no_wait = S.wait;
  const bool fuse_wait = mg && comm_step_wait_fused(ctx);
  const bool step_coll = mg && comm_step_collective(ctx);   // per-step exchanges as allgathers (RCCL / emulated)
  const int step_fence = (getenv("EIGX_STEP_FENCE") && atoi(getenv("EIGX_STEP_FENCE")) != 0) ? 1 : 0;
  // K_A's grid.  Several GPUs: K_A runs over this rank's row groups only, one group per workgroup
  auto ka_grid = [&](int rows, int& Gout) {
    const int ng = (rows + KA_ROWS - 1) / KA_ROWS;
    if (mg) { Gout = 1; const int own = local_count(ng, G.nranks, G.rank); return own > 0 ? own : 1; }
    Gout = (ng > 2 * g_ka_wgs) ? (ng + g_ka_wgs - 1) / g_ka_wgs : 1;
    const int nwg = (ng + Gout - 1) / Gout;
    return nwg > 0 ? nwg : 1;
  };
  // rows below L that this rank owns (they come first in its owned index order)
  auto own_count = [&](int L) {
    const int gfull = L / KA_ROWS, rem = L % KA_ROWS;
    return local_count(gfull, G.nranks, G.rank) * KA_ROWS + ((rem > 0 && gfull % G.nranks == G.rank) ? rem : 0);
  };
  // ---- several GPUs: the step launch (mg_step_kernel) with whatever roles are due --------------------------------
  // fuse_wait (every rank on its own GPU): one launch per step, [kl of the previous step | K_A | mat-vec], the consumers
  // spin on the flags in their prologues.  Otherwise (ranks sharing a card: wait kernels; collective exchanges) the roles
  // are launched one by one with the wait kernel / allgather between them.
  // polls of a spinning consumer: naps of ~60 ns between two looks at the flags (lab knob EIGX_SPIN_NAPS="ka,x"; with
  // relaxed polls 1 .. 64 naps all give the same time within 2 %)
  int naps_ka = 1, naps_x = 2;
  if (const char* e = getenv("EIGX_SPIN_NAPS")) { if (sscanf(e, "%d,%d", &naps_ka, &naps_x) < 2) naps_x = naps_ka; }
  MGStep MS;
  memset(&MS, 0, sizeof(MS));
  MS.xpeers = xpeers;
  KLArgs KLnone;
  memset(&KLnone, 0, sizeof(KLnone));
  KBArgs Bnone;
  memset(&Bnone, 0, sizeof(Bnone));
  Bnone.xwait = no_wait;
  // symv_T: tile edge of the mat-vec role (0: no mat-vec in this launch)
  auto launch_roles = [&](int nkl, const KLArgs& KLa, int nka, const KAArgs& Ka, int nsymv, const KBArgs& Ba, int symv_T, bool nt_loads, bool unc) {
    MGStep M = MS;
    M.nkl = nkl; M.nka = nka;
    const int gx = nkl + nka + nsymv;
    if (gx <= 0) return;
#define EIGX_STEP(RBv, NTv)                                                                                             \
  do {                                                                                                                  \
    if (unc) hipLaunchKernelGGL((mg_step_kernel<NB, RBv, NTv, true>), dim3(gx), dim3(256), 0, st, R, KLa, Ka, Ba, M);    \
    else hipLaunchKernelGGL((mg_step_kernel<NB, RBv, NTv, false>), dim3(gx), dim3(256), 0, st, R, KLa, Ka, Ba, M);       \
  } while (0)
    if (symv_T == 0) hipLaunchKernelGGL((mg_step_kernel<NB, 1, false, true>), dim3(gx), dim3(256), 0, st, R, KLa, Ka, Ba, M);
    else if (symv_T == 128) EIGX_STEP(1, false);
    else if (symv_T == 256 && !nt_loads) EIGX_STEP(2, false);
    else if (symv_T == 256) EIGX_STEP(2, true);
    else if (!nt_loads) EIGX_STEP(4, false);
    else EIGX_STEP(4, true);
#undef EIGX_STEP
  };
  int wk_last = -1;                       // W columns that the last K_A launch finished (-1: none), for the next mat-vec's copy
  int last_ka_i = -1, last_ka_ncols = 0;
  bool kl_pending = false;                // fuse_wait: the last mat-vec's kl role rides in the next launch
  KLArgs KLp = KLnone;
  int nkl_p = 0;
  // a K_A launch writes X message ++xepoch
  auto ka_mg_begin = [&](KAArgs& K) {
    ++xepoch;
    K.xpar = (int)(xepoch & 1); K.xepoch = xepoch;
    wk_last = K.has_prev ? K.kprev : -1;
    last_ka_i = K.i; last_ka_ncols = K.ncols;
    K.wait = (fuse_wait && K.has_prev) ? comm_step_wait_args(ctx, 0, epoch) : no_wait;
    K.wait.naps = naps_ka;
  };
  // roles one by one (no fused waits): [kl] was launched behind its mat-vec; here: wait / allgather Y, then [K_A]
  auto ka_mg_alone = [&](int nwg, const KAArgs& K, bool prof_it) {
    if (K.has_prev && !step_coll) {
      if (prof_it) ctx.prof_begin(3, 0.0, st);
      comm_step_wait(ctx, 0, epoch, st);
      if (prof_it) ctx.prof_end(st);
    }
    if (prof_it) ctx.prof_begin(4, 0.0, st);
    launch_roles(0, KLnone, nwg, K, 0, Bnone, 0, false, true);
    if (step_coll) comm_step_allgather(ctx, 1, xpeers.slot[0], K.xpar, st);
    if (prof_it) ctx.prof_end(st);
  };
  // consumer side of the X message outside the step launch: a StepWait for the kernel's prologue, or a wait kernel in
  // front of it (then n = 0)
  auto x_wait = [&](int prof_kind) -> StepWait {
    if (!mg || step_coll) return no_wait;
    if (fuse_wait) { StepWait w = comm_step_wait_args(ctx, 1, xepoch); w.naps = naps_x; return w; }
    if (prof_kind >= 0) ctx.prof_begin(prof_kind, 0.0, st);
    comm_step_wait(ctx, 1, xepoch, st);
    if (prof_kind >= 0) ctx.prof_end(st);
    return no_wait;
  };
  int k = 0;        // panel fill
  int i = n - 1;    // top column of the current block
  if (mg) {
    S.pan_c0 = panel_lo(n - 1);
    gather_panel(S.pan_c0, n - 1, st, CH_BULK);
  }
  double t_symv_bytes = 0.0;
  long n_symv = 0, n_k1 = 0;
  bool prof_step = false;   // several ranks: the previous step's mat-vec was sampled -> sample its wait and K_A as well
  double k1_flops = 0.0;
  while (true) {
    const int L = i - NB + 1;  // rows above the block
    const bool do_step = (L >= 1);
    // columns to form: a full block while reflectors remain, otherwise the last <= NB columns
    const int ncols = do_step ? NB : (i >= 0 ? (i + 1 < NB ? i + 1 : NB) : 0);
    S.ncols = ncols;
    S.i = i;
    S.L = do_step ? L : 0;
    S.k = k;
    S.rows = i + 1;
    if (S.has_prev && S.iprev + 1 > S.rows) S.rows = S.iprev + 1;
    // row groups per workgroup: one wave per SIMD at most (1024 SIMDs = 256 workgroups of 4 waves); the scalar work
    // of a workgroup is done once for all its groups
    const int nb_ka = ka_grid(S.rows, S.G);
    const bool need_ka = S.rows > 0 && (S.has_prev || ncols > 0);
    bool ka_held = false;      // several GPUs, fuse_wait: K_A rides in the step launch below
    if (need_ka && !mg) {
      launch_ka(nb_ka, S);
    } else if (need_ka) {
      ka_mg_begin(S);
      if (fuse_wait) ka_held = true;
      else ka_mg_alone(nb_ka, S, prof_step);
    }
    if (!do_step) {
      if (ka_held) launch_roles(kl_pending ? nkl_p : 0, KLp, nb_ka, S, 0, Bnone, 0, false, true);
      kl_pending = false;
      prof_step = false;
      break;
    }
    KBArgs B;
    B.i = i; B.L = L; B.k = k;
    B.ncg = (k + PD_COLS - 1) / PD_COLS;
    B.toprows = i + 1;
    B.pdr = pd_rows_for(B.toprows);
    B.nown_L = 0; B.npd_s = 0; B.pdr_s = 0; B.wk = -1; B.xpar = 0; B.xwait = no_wait;
    int npd = (B.toprows + B.pdr - 1) / B.pdr;
    if (mg) {
      // several GPUs: the panel dots over the rank's own rows below L in <= 8 chunks (>= 512 rows each; kl adds them up
      // and sends the share), the reflector store over all rows in <= 4 P chunks (the local tile stream is 1 / P of
      // one GPU's: a long K_P chunk would outlast it)
      B.nown_L = own_count(L);
      int r_ = ((B.nown_L + 7) / 8 + 63) / 64 * 64;
      B.pdr = r_ < 512 ? 512 : r_;
      npd = (B.nown_L + B.pdr - 1) / B.pdr;
      r_ = ((B.toprows + 4 * G.nranks - 1) / (4 * G.nranks) + 63) / 64 * 64;
      B.pdr_s = r_ < 512 ? 512 : r_;
      B.npd_s = (B.toprows + B.pdr_s - 1) / B.pdr_s;
      B.wk = wk_last;
      B.xpar = (int)(xepoch & 1);
      B.xwait = x_wait(prof_step ? 5 : -1);
    }
    prof_step = false;
    B.npd = npd;
    B.ngp = mg ? G.nranks : nb_ka;
    B.Lr = L; B.Lc = L; B.ntc = 0; B.nty_last = 0; B.slope = 0; B.c1 = 0;
    int T, ntiles;
    if (!mg) {
      const SymvGeom g = symv_geom(L);
      T = g.T;
      B.nt = g.nt;
      ntiles = g.nt * (g.nt + 1) / 2;                        // tiles of the upper block triangle
    } else {
      // local block below L: Lr x Lc; tile edge from the size of the local trapezoid
      B.Lr = local_count(L, G.Px, G.px);
      B.Lc = local_count(L, G.Py, G.py);
      const double leq = sqrt((double)(B.Lr > 1 ? B.Lr : 1) * (double)(B.Lc > 1 ? B.Lc : 1));
      T = (leq <= g_symv_t128) ? 128 : (leq <= g_symv_t256 ? 256 : 512);
      B.ntc = ceil_div(B.Lc, T);
      B.nt = B.ntc;
      B.nty_last = B.ntc > 0 ? mg_nty(B.ntc - 1, T, B.Lc, G.Px, G.px, G.Py, G.py) : 0;
      ntiles = B.nty_last;
      for (int tx = 0; tx + 1 < B.ntc; ++tx) ntiles += mg_nty(tx, T, B.Lc, G.Px, G.px, G.Py, G.py);
      if (G.Py % G.Px == 0) {
        B.slope = G.Py / G.Px;
        B.c1 = (int)(((long)(T - 1) * G.Py + G.py - G.px) / ((long)G.Px * T)) + 1;
      }
    }
    B.ng = T / 32;
    const int gx = ntiles + (mg ? npd * B.ncg + B.npd_s : npd * (B.ncg + 1));   // + K_P workgroups
    const bool prof = ctx.prof_stride > 0 && (n_symv % ctx.prof_stride) == 0;
    if (prof) ctx.prof_begin(0, 8.0 * ((double)L * (L + 1) / 2) / R.P, st);  // this rank's share of the triangle
    const bool nt_loads = (mg ? sqrt((double)B.Lr * B.Lc) : (double)L) > g_symv_nt;
    const bool unc = (mg ? sqrt((double)B.Lr * B.Lc) : (double)L) <= g_symv_unc;   // latency-bound sizes: the true two-unit pipeline
    if (!mg) {
#define EIGX_SYMV(RBv, NTv)                                                                                         \
  do {                                                                                                              \
    if (unc) hipLaunchKernelGGL((symv_kernel<NB, RBv, NTv, false, true>), dim3(gx), dim3(256), 0, st, R, B);        \
    else hipLaunchKernelGGL((symv_kernel<NB, RBv, NTv, false, false>), dim3(gx), dim3(256), 0, st, R, B);           \
  } while (0)
      if (T == 128) EIGX_SYMV(1, false);
      else if (T == 256 && !nt_loads) EIGX_SYMV(2, false);
      else if (T == 256) EIGX_SYMV(2, true);
      else if (!nt_loads) EIGX_SYMV(4, false);
      else EIGX_SYMV(4, true);
#undef EIGX_SYMV
      if (prof) ctx.prof_end(st);
    } else {
      // the step launch: [kl of the previous step | K_A | this mat-vec] (fuse_wait), or the mat-vec alone
      if (ka_held) launch_roles(kl_pending ? nkl_p : 0, KLp, nb_ka, S, gx, B, T, nt_loads, unc);
      else launch_roles(0, KLnone, 0, S, gx, B, T, nt_loads, unc);
      kl_pending = false;
      if (prof) ctx.prof_end(st);
      // the Y exchange behind the mat-vec: kl reduces this rank's tile partial sums, writes them into the owners' windows
      // and publishes the flag -- in the next launch (fuse_wait) or right away
      KLArgs KL;
      memset(&KL, 0, sizeof(KL));
      ++epoch;
      KL.L = L; KL.Lr = B.Lr; KL.Lc = B.Lc; KL.T = T; KL.ntc = B.ntc;
      KL.nbr = ceil_div(B.Lr > 0 ? B.Lr : 1, KL_ROWS);
      KL.par = (int)(epoch & 1);
      KL.epoch = epoch;
      KL.peers = peers;
      KL.kd2 = R.KD; KL.npd2 = npd; KL.kfill = k;
      KL.fence = step_fence;
      KL.ntr = 0;
      for (int tx = 0; tx < B.ntc; ++tx) { const int c_ = mg_nty(tx, T, B.Lc, G.Px, G.px, G.Py, G.py); if (c_ > KL.ntr) KL.ntr = c_; }
      const int nkl = KL.nbr + ceil_div(B.Lc > 0 ? B.Lc : 1, KL_ROWS) + 2 * NB + 1;   // + the scalar workgroup
      if (fuse_wait) { kl_pending = true; KLp = KL; nkl_p = nkl; }
      else {
        if (prof) ctx.prof_begin(2, 8.0 * R.msg_stride, st);
        launch_roles(nkl, KL, 0, S, 0, Bnone, 0, false, true);
        if (step_coll) comm_step_allgather(ctx, 0, peers.slot[0], KL.par, st);
        if (prof) ctx.prof_end(st);
      }
      prof_step = prof && !fuse_wait;
      S.par = KL.par;
    }
    t_symv_bytes += 8.0 * ((double)L * (L + 1) / 2);
    ++n_symv;
    // bookkeeping for the next K_A
    S.has_prev = 1; S.iprev = i; S.Lprev = L; S.kprev = k;
    S.nchunk_prev = mg ? G.nranks : npd;   // shares of the panel dots that the next K_A adds: one per rank / one per K_P row chunk
    S.nchunk_ab = mg ? B.npd_s : npd;
    S.nt_prev = B.nt; S.lgT_prev = (T == 128) ? 7 : (T == 256 ? 8 : 9);
    k += NB;
    i -= NB;
    if (k >= m && i - NB + 1 >= 1) {
      // panel full and more reflectors to come: finish W, trailing update, start a new panel
      KAArgs F = S;
      prof_step = false;   // (the panel-closing K_A is not part of the sampled step breakdown)
      F.ncols = 0; F.i = i; F.L = 0; F.k = k; F.rows = S.iprev + 1;
      const int nb_kf = ka_grid(F.rows, F.G);
      F.wait.n = 0;
      if (!mg) launch_ka(nb_kf, F);
      else {
        ka_mg_begin(F);
        if (fuse_wait) launch_roles(kl_pending ? nkl_p : 0, KLp, nb_kf, F, 0, Bnone, 0, false, true);
        else ka_mg_alone(nb_kf, F, false);
        kl_pending = false;
        // the last W columns of the panel, finished by their owners just now, into the local panel (all rows)
        const StepWait xw = x_wait(-1);
        hipLaunchKernelGGL((mg_wcopy_kernel<NB>), dim3(ceil_div(F.rows, 256) < 64 ? ceil_div(F.rows, 256) : 64), dim3(256), 0, st, R,
                           F.xpar, F.kprev, F.rows, xw);
        wk_last = -1;
      }
      const int nr = i + 1;
      if (ctx.prof_stride > 0) ctx.prof_begin(1, 2.0 * (double)nr * nr * m / R.P, st);
      if (!mg) {
        dgemm_dev(st, 'N', 'T', nr, nr, 2 * m, -1.0, R.UW, ldp, R.UW + (size_t)ldp * m, ldp, 1.0, A, lda, 1);
      } else {
        // Local trailing update A_loc -= [U W](rows) [W U](cols)^T: needs no communication (src/eigen_t1.F:250-306).
        // Look-ahead: the local tile columns that hold the NEXT panel, global columns (i-m, i], are updated first;
        // they are then gathered from their owners on the side stream (the reference's panel-load allgather,
        // src/eigen_prd_t7.F:114-128) while the compute stream updates the rest of the trailing matrix.
        const int nrl = local_count(nr, G.Px, G.px), ncl = local_count(nr, G.Py, G.py);
        const int clo = panel_lo(i);
        const int lj0 = (clo - G.py + G.Py - 1) / G.Py;
        const int tb0 = lj0 / 128;
        const int lmax = nrl > ncl ? nrl : ncl;
        if (lmax > 0)
          hipLaunchKernelGGL(compact_panel_kernel, dim3(ceil_div(lmax, 256) < 64 ? ceil_div(lmax, 256) : 64, 2 * m), dim3(256),
                             0, st, (const double*)R.UW, ldp, m, nrl, ncl, G.Px, G.px, G.Py, G.py, UWr, ldr, UWc, ldc);
        if (nrl > 0 && ncl > 0)
          dgemm_dev(st, 'N', 'T', nrl, ncl, 2 * m, -1.0, UWr, ldr, UWc, ldc, 1.0, A, lda, 2, &G, nullptr, nullptr, 1, 0, 0,
                    0, 1, 0, 0, 0, 1, 0, nullptr, tb0, 0x7fffffff);
        EIGX_HIP_CHECK(hipEventRecord(ctx.aux_ev[0], st));
        if (tb0 > 0 && nrl > 0 && ncl > 0)
          dgemm_dev(st, 'N', 'T', nrl, ncl, 2 * m, -1.0, UWr, ldr, UWc, ldc, 1.0, A, lda, 2, &G, nullptr, nullptr, 1, 0, 0,
                    0, 1, 0, 0, 0, 1, 0, nullptr, 0, tb0);
        hipStream_t sd = ctx.side_stream;
        EIGX_HIP_CHECK(hipStreamWaitEvent(sd, ctx.aux_ev[0], 0));
        gather_panel(clo, i, sd, CH_SIDE);
        EIGX_HIP_CHECK(hipEventRecord(ctx.aux_ev[1], sd));
        EIGX_HIP_CHECK(hipStreamWaitEvent(st, ctx.aux_ev[1], 0));
        S.pan_c0 = clo;
      }
      if (ctx.prof_stride > 0) ctx.prof_end(st);
      k1_flops += 2.0 * (double)nr * nr * m;  // 2*nr*nr*(2m)/2 : upper triangle only
      ++n_k1;
      hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, R.UW, (size_t)ldp * m * 3, 0.0);
      S.has_prev = 0;
      k = 0;
    }
  }
  if (mg && last_ka_ncols > 0) {
    // the last <= NB columns: no mat-vec follows whose publisher would take d, e from the X message
    const StepWait xw = x_wait(-1);
    hipLaunchKernelGGL((mg_tail_kernel<NB>), dim3(1), dim3(64), 0, st, R, (int)(xepoch & 1), last_ka_i, last_ka_ncols, xw);
  }
  if (getenv("EIGX_TRACE_ENQUEUE")) {
    const double te = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
    const double ts = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[eigx] reduction: enqueue loop %.1f ms, then %.1f ms until the stream drained\n", (te - t_begin) * 1e3, (ts - te) * 1e3);
  }
#ifdef EIGX_STAMPS
  {
    unsigned long long h[32];
    EIGX_HIP_CHECK(hipStreamSynchronize(st));
    EIGX_HIP_CHECK(hipMemcpy(h, R.dbg, sizeof(h), hipMemcpyDeviceToHost));
    if (mg && h[24]) fprintf(stderr, "[eigx stamps] step launch timeline (100-MHz clock, averages over %llu full launches, from workgroup 0's entry): Y flags "
                             "raised at %.2f us, X flags at %.2f us, next launch's workgroup 0 enters at %.2f us\n", h[24], 0.01 * h[22] / h[24],
                             0.01 * h[23] / h[24], 0.01 * h[21] / h[24]);
    if (R.stamp_i >= 0 && mg) {
      // one sampled step (EIGX_STAMP_I = its top column): per-workgroup times of the three roles on the 100-MHz clock
      std::vector<unsigned long long> big(4096);
      EIGX_HIP_CHECK(hipMemcpy(big.data(), R.dbg, 4096 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull; int nw = 0, nk = 0;
      for (int b = 0; b < 256; ++b) if (big[3136 + b]) { ++nk; if (big[3136 + b] < t0) t0 = big[3136 + b]; }
      for (int b = 0; b < 1024; ++b) if (big[64 + b]) { ++nw; if (big[64 + b] < t0) t0 = big[64 + b]; }
      double mk[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
      for (int b = 0; b < nk; ++b) for (int q = 0; q < 3; ++q) { const double v = 0.01 * (big[3136 + 256 * q + b] - t0); if (v > mk[q]) mk[q] = v; }
      for (int b = 0; b < nw; ++b) for (int q = 0; q < 3; ++q) { const double v = 0.01 * (big[64 + 1024 * q + b] - t0); if (v > mx[q]) mx[q] = v; }
      fprintf(stderr, "[eigx stamps] step with top column %d, us after the launch's first workgroup: kl role (%d workgroups) latest entry %.2f, partial sums in %.2f, "
              "stores drained %.2f | ka role (%d) latest entry %.2f, Y seen %.2f, pushes issued %.2f; count complete %.2f, X flags stored %.2f | mat-vec role "
              "(%llu tiles): X seen first %.2f last %.2f, tile ends first %.2f last %.2f\n", R.stamp_i, nk, mk[0], mk[1], mk[2], nw, mx[0], mx[1], mx[2],
              0.01 * (big[3904] - t0), 0.01 * (big[3905] - t0), big[3910], 0.01 * (~big[3906] - t0), 0.01 * (big[3907] - t0), 0.01 * (~big[3909] - t0),
              0.01 * (big[3908] - t0));
    }
    if (h[30]) fprintf(stderr, "[eigx stamps] K_A workgroup in the middle: %.2f us of wall clock per launch, %.0f s_memtime ticks: %.0f MHz\n",
                       0.01 * h[30] / h[7], (double)h[31] / h[7], (double)h[31] / (0.01 * h[30]));
    if (mg && h[24]) fprintf(stderr, "[eigx stamps] ka role, workgroup in the middle: enters at %.2f us, sees the Y flags at %.2f, has issued its pushes at %.2f; "
                             "the watcher sees the count complete at %.2f us\n", 0.01 * h[26] / h[7], 0.01 * h[27] / h[7], 0.01 * h[28] / h[7], 0.01 * h[29] / h[14]);
    if (mg) fprintf(stderr, "[eigx stamps] K_A several GPUs: wait %.0f | to the end of the pushes %.0f, drain + barrier %.0f | last arriver (from its previous stamp "
                    "to the flag store) %.0f x %llu\n", (double)h[5] / h[7], (double)h[6] / h[7], (double)h[12] / h[7], (double)h[13] / (h[14] ? h[14] : 1), h[14]);
    fprintf(stderr, "[eigx stamps] NB=%d n=%d K_A launches %llu: avg cycles issue %.0f consume %.0f reduce %.0f rows %.0f tail %.0f | SYMV %llu: "
            "entry %.0f scalars %.0f stream %.0f tail %.0f\n", NB, n, h[7], (double)h[0] / h[7], (double)h[1] / h[7], (double)h[2] / h[7],
            (double)h[3] / h[7], (double)h[4] / h[7], h[15], (double)h[8] / (h[15] ? h[15] : 1), (double)h[9] / (h[15] ? h[15] : 1),
            (double)h[10] / (h[15] ? h[15] : 1), (double)h[11] / (h[15] ? h[15] : 1));
  }
#endif
  EIGX_HIP_CHECK(hipGetLastError());
  ctx.timers[6] = (double)n_k1;
  ctx.timers[7] = k1_flops;
  ctx.timers[9] = (double)n_symv;
  ctx.timers[10] = t_symv_bytes;
}

}  // namespace

int set_symv_threshold(int which, int v) {
  if (which == 5) { const int old = g_symv_unc; g_symv_unc = v; return old; }
  if (which == 6) return 0;   // (removed: folded step exchange)
  int& t = (which == 4) ? g_ka_fit : (which == 3) ? g_ka_wgs : (which == 2) ? g_symv_nt : (which ? g_symv_t256 : g_symv_t128);
  const int old = t; t = v; return old;
}

void band_reduce_dev(Context& ctx, int n, double* A, int lda, double* d, double* e, int lde, int m, int band) {
  if (band == 1) band_reduce_impl<1>(ctx, n, A, lda, d, e, lde, m);
  else band_reduce_impl<2>(ctx, n, A, lda, d, e, lde, m);
}

}  // namespace eigx
