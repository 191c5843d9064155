/* eigenexa_amd.h -- C-ABI of the MI355X-native EigenExa hot path (libeigenexa_amd.so).
 *
 * Plain C, plain pointers and sizes; no torch / HIP types in any signature (streams are the
 * library's own).  Every entry point names the reference interface it replaces
 * (paths relative to the RIKEN-RCCS/EigenExa 2.13 tree).
 *
 * Conventions kept from the reference:
 *   - all matrices column-major, fp64; indices/sizes 32-bit int;
 *   - the matrix is distributed 2-D cyclically (block size 1) over a Px x Py process grid,
 *     global (i,j) (0-based) -> rank (i%Px, j%Py), local (i/Px, j/Py)   (src/eigen_libs0.F:1825-2258);
 *   - only the upper triangle (global row <= global col) of `a` is read; `a` is destroyed and on
 *     return a(1,1)=flop count, a(2,1)=elapsed seconds, a(3,1)=communication seconds or -1
 *     (src/eigen_sx.F:285-296);
 *   - w(1:n) ascending eigenvalues, replicated; z(ldz, *) cyclic eigenvectors;
 *   - `mode`: only the first character is used: 'A' all eigenpairs, 'N' eigenvalues only,
 *     'X' eigenpairs + refined eigenvalues (src/eigen_sx.F:103-118).
 *   - errors: no status argument in the reference; here every function returns 0 on success and a
 *     negative code on a precondition failure (the Fortran module drops it to keep the
 *     reference's silent-return behaviour, src/eigen_sx.F:82-131).
 *
 * Two families of solver entry points:
 *   eigx_sx / eigx_s          host arrays in, host arrays out     (drop-in for the Fortran API)
 *   eigx_sx_dev / eigx_s_dev  device (HBM-resident) arrays        (what bench.py times)
 * and, on top of them: eigx_solve_bc[_dev] (block-cyclic local blocks of a ScaLAPACK descriptor),
 * eigx_gev[_dev] (KMATH_EIGEN_GEV), eigx_h[_dev] (complex Hermitian eigen_h).
 */
#ifndef EIGENEXA_AMD_H
#define EIGENEXA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EIGX_OK 0
#define EIGX_ERR_NOT_INITIALIZED (-1)
#define EIGX_ERR_BAD_ARG (-2)
#define EIGX_ERR_TOO_LARGE (-3)
#define EIGX_ERR_NO_DEVICE (-4)
#define EIGX_ERR_NONFINITE (-5)
#define EIGX_ERR_INTERNAL (-6)
#define EIGX_ERR_NOT_SPD (-7)
#define EIGX_ERR_NO_MEMORY (-8)   /* a workspace allocation failed; on several ranks the others return EIGX_ERR_INTERNAL at once */

/* ---- life cycle -------------------------------------------------------------------------- */

/* replaces eigen_init(comm, order)  src/eigen_libs.F:70-104 -> eigen_init0 src/eigen_libs0.F:296-376.
 * Single-rank form: 1x1 grid on HIP device `device`. */
int eigx_init(int device);

/* Multi-rank form (one process per GPU, all on one xGMI node).  `nranks` processes call this collectively; the
 * grid is Px = largest divisor of nranks <= sqrt(nranks), Py = nranks/Px, column-major rank order
 * (src/eigen_libs0.F:526-570).  `session_id` is the 128-byte id created by eigx_get_rccl_unique_id() on rank 0
 * and broadcast by the caller (MPI_Bcast in a Fortran/MPI host, torch.distributed in bench.py): an ncclUniqueId
 * when RCCL is installed (it also seeds the world / X / Y RCCL communicators), random bytes otherwise.  The ranks
 * find each other through a POSIX shared-memory board named after it, exchange hipIpcMemHandles of their
 * communication windows and map them: kernels then write into the peers' HBM over xGMI directly.
 * Replaces MPI_Comm_dup/MPI_Comm_split of eigen_init_comm_setup/eigen_init_cartesian_check,
 * src/eigen_libs0.F:382-428, :579-715.  Several ranks may share one GPU (the tests do): RCCL is then not used and
 * every collective goes through the peer windows.
 * Transport ladder, decided at init by a self-test with checksummed payloads on both transports (a few hundred
 * ready / push / flag / wait rounds and step-window rounds over the peer windows; all-reduces over the X, Y and world
 * RCCL communicators, all-gather and grouped send / receive): peer windows for everything if they pass (the form
 * every multi-rank test runs), RCCL for whatever they cannot carry (the per-step exchange then is one ncclAllGather
 * of the step messages -- the reference's reduce_dbl over X and Y, src/comm.F:1192-1247, as one collective), an error
 * return if neither works (bench.py then runs independent replicas).  eigx_comm_info reports what was chosen.
 * Environment: EIGX_COMM_TIMEOUT_S (default 120) bounds every wait for a peer; EIGX_BULK=rccl moves the bulk
 * collectives to RCCL; EIGX_STEP=coll selects the collective form of the per-step exchange; EIGX_FUSE_WAIT=1 folds the
 * step wait into the consumer kernel; EIGX_NO_IPC / EIGX_NO_RCCL disable a transport; EIGX_SELFTEST_ROUNDS (default 400,
 * 0 = skip), EIGX_SELFTEST_FAIL=ipc|rccl (make that leg report failure: tests of the ladder). */
int eigx_init_multi(int device, int rank, int nranks, const void* session_id, char order);
/* visible HIP devices (0 without a GPU): lets an MPI host map its node-local rank to a device
 * (eigen_libs_mod.F90: MPI_Comm_split_type + modulo) */
int eigx_get_device_count(void);
int eigx_get_rccl_unique_id(void* out128);
/* Explicit Px x Py process grid for the NEXT eigx_init_multi call (one-shot; 0, 0 clears): the 2-D cartesian
 * communicator form of eigen_init (eigen_init_cartesian_check, src/eigen_libs0.F:579-715), which the reference's
 * benchmark driver builds for its -x option (benchmark/main2.f:193-211). */
int eigx_set_grid_dims(int px, int py);

/* replaces eigen_get_comm src/eigen_libs0.F:1655-1669 as far as a GPU library can: what the caller needs to build
 * its own row / column communicators -- the colour and key of this rank in the X group (ranks sharing my column
 * coordinate: colour = y_id, key = x_id) and in the Y group (colour = x_id, key = y_id), 1-based ids. */
int eigx_get_comm(int* x_color, int* x_key, int* y_color, int* y_key);

/* Seconds this rank spent in communication (pushes, waits for peers, RCCL calls) during the last solve:
 * the a(3,1) statistic of src/eigen_sx.F:285-296 and the "COMM_STAT" tables of src/eigen_devel.F:364-526. */
double eigx_comm_seconds(void);

/* JSON text describing the transports in use (per-step exchange, its wait, bulk collectives) and the counts / errors /
 * microseconds per round of the init-time self-test, plus calls and bytes sent per kind of collective since init (the
 * reference's COMM_STAT tables, src/eigen_devel.F:364-526); "{"ranks": 1}" on one GPU.  The reference prints the analogous
 * communicator facts at init (src/eigen_libs0.F:774-1109 measures its collectives there). */
int eigx_comm_info(char* buf, int len);

/* 1-rank RCCL self-test (dlopen, communicator from a unique id, ncclCommSplit, allreduce / allgather / send-recv on
 * the library stream): validates the RCCL plumbing on a one-GPU box.  Returns 0 on success. */
int eigx_rccl_selftest(void);

/* replaces eigen_free  src/eigen_libs.F:204-216 */
int eigx_free(void);

/* replaces eigen_get_version src/eigen_libs0.F:175 ; version = 100*major+minor of this library */
int eigx_get_version(int* version, char* date32, char* vcode32);

/* replaces eigen_get_procs / eigen_get_id  src/eigen_libs0.F:1575-1655 (1-based ids like the reference) */
int eigx_get_procs(int* procs, int* x_procs, int* y_procs);
int eigx_get_id(int* id, int* x_id, int* y_id);

/* replaces eigen_get_errinfo src/eigen_libs0.F:1689-1698 */
int eigx_get_errinfo(int64_t* info);

/* replaces eigen_get_matdims(n, nx, ny, m_forward, m_backward, mode) src/eigen_libs.F:106-148,
 * src/eigen_libs0.F:1254-1371.  Returns local array extents that are >= the reference's for the
 * same (n, grid) so existing callers' allocations stay valid; nx = ny = -1 if too large.
 * Mode 'O' (the default): where the reference nudges nx off A64FX cache-set aliasing (src/CSTAB.F:73-131), nx here is
 * moved off MI355X memory-channel aliasing -- from 2048 on, nx mod 2048 lies in [512, 1536] (consecutive columns
 * 4 - 12 KiB apart modulo 16 KiB); the solvers work in place on a(nx, *), and this is worth 2 - 9 % of a solve. */
int eigx_get_matdims(int n, int* nx, int* ny, int m_forward, int m_backward, char mode);
/* the same rule for an explicit x_procs x y_procs grid; pure arithmetic (usable before eigx_init, without a GPU) */
int eigx_matdims_for_grid(int n, int x_procs, int y_procs, int m_forward, int m_backward, char mode, int* nx, int* ny);

/* replaces eigen_memory_internal src/eigen_libs0.F:1395-1549: bytes of device workspace a solve needs */
int64_t eigx_memory_internal(int n, int lda, int ldz, int m_forward, int m_backward);
/* bytes of device memory the library holds right now (pooled workspace + communication windows); the tests check it
 * against eigx_memory_internal and that it scales like 1/P on several ranks.  -1 before eigx_init. */
int64_t eigx_held_bytes(void);
/* the same for the pooled workspace buffers whose name starts with `prefix` (e.g. "gev." = what KMATH_EIGEN_GEV holds
 * beside the two eigen_s solves: the tests check that it is a few n^2 / P, nothing gathered).  -1 before eigx_init. */
int64_t eigx_held_bytes_named(const char* prefix);
/* pure arithmetic, no GPU needed: the pieces of the distributed transpose Z = A^T on the 2-D cyclic layout (the PDTRAN of
 * src/KMATH_EIGEN_GEV_1.F:57) that rank (px, py) of a Px x Py grid exchanges with rank (qx, qy).  A piece is the set of
 * elements Z(i, j) = A(j, i) with i = i0 + t step, j = j0 + u step (step = lcm(Px, Py)); send_*: the piece I pack for
 * (qx, qy), recv_*: the one I get from it; i0 or j0 = -1: that pair exchanges nothing.  The CPU tests assemble A^T from
 * the pieces for every grid up to 8 ranks. */
int eigx_transpose_plan(int x_procs, int y_procs, int px, int py, int qx, int qy, int* send_i0, int* send_j0, int* recv_i0,
                        int* recv_j0, int* step);

/* ---- index helpers (pure functions; 1-based like the reference, src/eigen_libs0.F:1744-2356) - */
int eigx_loop_start(int istart, int nnod, int inod);
int eigx_loop_end(int iend, int nnod, int inod);
int eigx_translate_l2g(int ictr, int nnod, int inod);
int eigx_translate_g2l(int ictr, int nnod, int inod);
int eigx_owner_node(int ictr, int nnod, int inod);
int eigx_owner_index(int ictr, int nnod, int inod);

/* ---- solvers ------------------------------------------------------------------------------ */

/* replaces eigen_sx(n,nvec,a,lda,w,z,ldz,m_forward,m_backward,mode) src/eigen_sx.F:30-308
 * (pentadiagonal route: eigen_prd -> eigen_dcx -> eigen_common_trbakwy(nb=2)). Host arrays. */
int eigx_sx(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int m_forward,
            int m_backward, char mode);

/* replaces eigen_s(...) src/eigen_libs.F:150-202 -> eigen_FS src/eigen_FS.F:29-300 /
 * eigen_s0 src/eigen_s.F:30-307 (tridiagonal route: eigen_trd -> dc2 -> trbakwy(nb=1)). Host arrays. */
int eigx_s(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int m_forward,
           int m_backward, char mode);

/* Same solvers on device-resident arrays (a_dev, w_dev, z_dev are HBM pointers of this rank's GPU).  lda >= the local
 * row count (any parity: an odd lda or a base that is not 16-byte aligned is served from an internal padded copy); like
 * the reference `a` is destroyed, and its padding rows (local rows beyond the matrix, up to lda) are scratch as well. */
/* Device entry points synchronise the (legacy) default stream on entry, so whatever the caller queued there to fill
 * the arguments is complete; a caller that fills them on another stream synchronises that stream itself.  They return
 * after the result is complete. */
int eigx_sx_dev(int n, int nvec, double* a_dev, int lda, double* w_dev, double* z_dev, int ldz,
                int m_forward, int m_backward, char mode);
int eigx_s_dev(int n, int nvec, double* a_dev, int lda, double* w_dev, double* z_dev, int ldz,
               int m_forward, int m_backward, char mode);

/* ScaLAPACK interop without a redistribution step.  The reference asks block-cyclic callers to
 * convert with pdgemr2d into its cyclic layout first (manual 3.4; benchmark/ev_test.f:68-84 does the reverse for the
 * check).  Here the layout is only an index map at the entry and exit of the solver, so the local blocks of a
 * descriptor with MB = NB = nb, RSRC = CSRC = 0 on the eigx process grid (eigx_get_procs / eigx_get_id, same grid
 * as a BLACS grid of that shape and order) are accepted as they are: a is the local numroc(n,nb,px,Px) x
 * numroc(n,nb,py,Py) block, z comes back as the local block of the n x nvec eigenvector matrix in the same
 * distribution, w replicated.  route: 2 = eigen_sx, 1 = eigen_s.  nb = 1 is the cyclic layout of eigx_sx / eigx_s. */
int eigx_solve_bc(int route, int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int nb,
                  int m_forward, int m_backward, char mode);
int eigx_solve_bc_dev(int route, int n, int nvec, double* a_dev, int lda, double* w_dev, double* z_dev, int ldz,
                      int nb, int m_forward, int m_backward, char mode);
/* NUMROC(n, nb, iproc, 0, nprocs) of ScaLAPACK (TOOLS/numroc.f): local extent; -1 for invalid arguments */
int eigx_numroc(int n, int nb, int iproc, int nprocs);

/* replaces eigen_h(n,nvec,a,lda,w,z,ldz,m_forward,m_backward,mode) src/eigen_h.F:30-322 (complex Hermitian:
 * eigen_scaling_h -> eigen_hrd -> dc2 -> eigen_hrbakwyx).  a, z are complex(8) arrays passed as interleaved (re, im)
 * doubles, column-major, lda / ldz in COMPLEX elements; upper triangle of a significant; a is destroyed
 * (a(1,1) = flops, a(2,1) = seconds); w real, ascending; z = eigenvectors (unitary).  mode 'A', 'N', 'X'.  With more
 * than one rank a and z are the 2-D cyclic local blocks as for eigx_sx; the blocks are gathered and every rank solves
 * the replicated problem in this version (first cut: see csrc/herm.hip). */
int eigx_h(int n, int nvec, double* a, int lda, double* w, double* z, int ldz, int m_forward, int m_backward, char mode);
int eigx_h_dev(int n, int nvec, double* a_dev, int lda, double* w_dev, double* z_dev, int ldz, int m_forward,
               int m_backward, char mode);

/* ---- stage entry points (device arrays; used by the parity tests and the profiler) ---------- */

/* replaces eigen_trd(n,a,lda,d,e,m) src/eigen_trd.F:82-113 (band=1) and
 * eigen_prd(n,a,lda,d,e,nme,m) src/eigen_prd.F:80-115 (band=2).
 * Out: d_dev[n]; e_dev[band*lde] with e(i,b) = band entry T(i-b,i) (1-based, zero for i<=b);
 * reflectors stay in a_dev columns, their 1/beta recoverable from e (src/trbakwy4.F:309-335). */
int eigx_band_reduce_dev(int n, double* a_dev, int lda, double* d_dev, double* e_dev, int lde,
                         int m_forward, int band);

/* replaces eigen_dc2 src/dc2.F (band=1) / eigen_dcx src/dcx.F:81-337 (band=2): eigen-decomposition of
 * the symmetric band matrix (d,e); w_dev ascending, z_dev(ldz, n) eigenvectors. */
int eigx_band_dc_dev(int n, int nvec, const double* d_dev, const double* e_dev, int lde, int band,
                     double* w_dev, double* z_dev, int ldz);

/* replaces KMATH_EIGEN_GEV(n,a,lda,b,ldb,w,z,ldz) src/KMATH_EIGEN_GEV.F:1-64 (-> KMATH_EIGEN_GEV_1.F:1-159): generalised
 * symmetric-definite problem A x = lambda B x through two eigen_s solves and three GEMMs.  Upper triangles of a, b
 * significant; w ascending; z B-orthonormal (z^T B z = I); a, b destroyed.  EIGX_ERR_NOT_SPD if B is not positive
 * definite (the reference prints "Matrix B is not positive definite!" and returns).  Host / device-resident arrays;
 * leading dimensions of the device form must be even.  Several ranks: a, b, z are the ranks' 2-D cyclic blocks as for
 * eigx_sx; first version: the blocks are gathered and every rank solves the replicated problem (the reference's is
 * distributed, src/KMATH_EIGEN_GEV_1.F:57-139). */
int eigx_gev(int n, double* a, int lda, double* b, int ldb, double* w, double* z, int ldz);
int eigx_gev_dev(int n, double* a_dev, int lda, double* b_dev, int ldb, double* w_dev, double* z_dev, int ldz);

/* replaces eigen_bisect(d,e,w,n,mode) src/bisect.F:67-397 (band=1) / eigen_bisect2(d,e,f,w,n,mode)
 * src/bisect2.F:71-718 (band=2): all eigenvalues of the band matrix by Sturm counts, w_dev ascending.
 * Used by modes 'N', 'S', 'C' (alone) and 'X' (after the divide and conquer), src/eigen_sx.F:200-222. */
int eigx_band_bisect_dev(int n, const double* d_dev, const double* e_dev, int lde, int band, double* w_dev);

/* replaces eigen_common_trbakwy(n,nvec,a,lda,z,ldz,e,m,nb) src/trbakwy4.F:77-222 */
int eigx_trbak_dev(int n, int nvec, const double* a_dev, int lda, double* z_dev, int ldz,
                   const double* e_dev, int lde, int m_backward, int band);

/* replaces the BLAS dgemm call sites of the path (src/eigen_t1.F:285-295, src/trbakwy4_body.F:604-608,
 * :721-725, src/FS_PDLAED3.F90:833-860): C = alpha*op(A)*op(B)+beta*C on device arrays.
 * tri_upper != 0 restricts the update to 128x128 tiles touching the upper triangle. */
int eigx_dgemm_dev(char opa, char opb, int m, int n, int k, double alpha, const double* a_dev, int lda,
                   const double* b_dev, int ldb, double beta, double* c_dev, int ldc, int tri_upper);

/* same product with column gathers, as used by the D&C eigenvector update Q <- Q(:, nondeflated) * S
 * (replaces the copy into the compressed Q2 + PDGEMM of src/my_pdlaed2.F / src/my_pdlaed1.F:310-341):
 * A(:, kmap_a[k]) supplies k-index k (opa = 'N' only), B(:, kmap_b[k]) likewise (opb = 'T' only);
 * either map may be NULL.  Maps are device int arrays of length k. */
int eigx_dgemm_gather_dev(char opa, char opb, int m, int n, int k, double alpha, const double* a_dev, int lda,
                          const double* b_dev, int ldb, double beta, double* c_dev, int ldc,
                          const int* kmap_a_dev, const int* kmap_b_dev);

/* timers of the last solve, seconds: [0] total [1] reduction [2] d&c [3] back-transform [4] comm
 * (reference: TIMER_PRINT lines, src/eigen_sx.F:167-174, :300-304).  kernel-level stats for bench.py:
 * [5] trailing-update kernel seconds (sum of launches) [6] its launch count [7] its flops
 * [8] symv kernel seconds [9] its launch count [10] its algorithmic bytes */
int eigx_get_timers(double* out16);

/* Sampled in-library profiling for bench.py (the reference prints per-kernel timers under TIMER_PRINT=2,
 * src/eigen_trd.F:710-714): eigx_profile(stride>0) brackets every stride-th launch of the fused symmetric
 * mat-vec kernel and every trailing-update GEMM launch with HIP events on the library's compute stream;
 * eigx_profile_read returns {symv launches sampled, their algorithmic bytes, their seconds,
 * trailing-update launches, their flops, their seconds} accumulated since the last eigx_profile call. */
int eigx_profile(int stride);
int eigx_profile_read(double* out6);
/* the same events by kind: out[3 k + {0, 1, 2}] = {launches sampled, units, seconds} for kind k < nkinds;
 * kinds: 0 fused mat-vec, 1 trailing update, and on several ranks the rest of a sampled reduction step:
 * 2 local reduce + push of the step message (kl_kernel, or kl_kernel + the allgather in the collective form),
 * 3 wait for the peers' messages (wait kernel), 4 ka_kernel (with the wait when it is fused into it). */
int eigx_profile_read_kinds(double* out, int nkinds);

/* Tuning hook for A/B measurements (tools/, tests/): key 0 = GEMM kernel (2 = LDS-DMA ring kernel where it
 * applies [default], 1 = register-staged kernel everywhere); key 1 = target number of concurrent Sturm
 * sweeps of the bisection (default 65536); key 2 = super-block factor of the back-transformation (0 = automatic);
 * keys 3 / 4 = largest active size L that uses the 128 / 256 tile of the fused symmetric mat-vec, key 5 = active
 * size above which it streams the matrix with non-temporal loads; key 6 = 1: the trailing update streams its C tiles
 * past L2; key 7 = workgroups of the column-formation kernel beyond which a workgroup loops over row groups;
 * key 8 = chunk width (roots, 64 .. 2048) of the multi-rank D&C's eigenvector-row buffer; key 9 = doubles per slice of
 * the bounce window of the multi-rank eigenvector redistributions (keys 7-9 exist so that the tests reach the
 * large-N code paths at small sizes); key 10 = 0: the column-formation kernel always uses its largest load batches (A/B).  Returns the previous value, or -1 for an unknown key.  Not part of the
 * reference's interface. */
int eigx_tune(int key, int value);

/* device synchronisation helper for hosts without a HIP binding */
int eigx_device_synchronize(void);

/* device memory helpers for hosts without a HIP binding (Fortran callers, ctypes tests) */
void* eigx_malloc_dev(int64_t bytes);
int eigx_free_dev(void* p);
int eigx_memcpy_h2d(void* dst_dev, const void* src_host, int64_t bytes);
int eigx_memcpy_d2h(void* dst_host, const void* src_dev, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* EIGENEXA_AMD_H */
