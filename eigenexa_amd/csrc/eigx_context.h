// eigx_context.h -- library-global state (the reference keeps the same kind of module-global state:
// TRD_COMM_WORLD, x_nnod, ... in src/eigen_devel.F:53-61; one live grid at a time, not re-entrant).
#pragma once
#include "eigx_common.h"
#include <functional>
#include <map>
#include <vector>
#include <string>

namespace eigx {

// Workspace cache: named device buffers that persist between solves (hipMalloc of multi-GB buffers
// costs milliseconds; the reference allocates per call on the host where that is free).
// Thrown by the workspace pool when the device is out of memory; caught at the C-ABI boundary (eigx_guard), which
// tells the other ranks (their bounded waits return at once instead of running out their time limit) and returns
// EIGX_ERR_NO_MEMORY.  The reference aborts the whole job here (eigen_abort -> MPI_Abort, src/eigen_devel.F:148-164).
struct DeviceAllocError { size_t bytes; std::string name; };

struct Pool {
  struct Buf { void* p = nullptr; size_t bytes = 0; };
  std::map<std::string, Buf> bufs;
  void* get(const std::string& name, size_t bytes) {
    Buf& b = bufs[name];
    if (b.bytes < bytes) {
      if (b.p) EIGX_HIP_CHECK(hipFree(b.p));
      b.p = nullptr;
      b.bytes = 0;
      size_t want = bytes + bytes / 16 + 256;
      // EIGX_TEST_FAIL_ALLOC=<buffer name>: this allocation fails (tests of the failure path)
      static const char* fail_name = getenv("EIGX_TEST_FAIL_ALLOC");
      if ((fail_name && name == fail_name) || hipMalloc(&b.p, want) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        throw DeviceAllocError{want, name};
      }
      b.bytes = want;
    }
    return b.p;
  }
  template <typename T> T* get_t(const std::string& name, size_t count) {
    return (T*)get(name, count * sizeof(T));
  }
  // pinned host staging buffers (one D2H / H2D copy per D&C merge step instead of sixteen pageable ones)
  std::map<std::string, Buf> hbufs;
  void* get_host(const std::string& name, size_t bytes) {
    Buf& b = hbufs[name];
    if (b.bytes < bytes) {
      if (b.p) EIGX_HIP_CHECK(hipHostFree(b.p));
      b.p = nullptr;
      size_t want = bytes + bytes / 16 + 256;
      EIGX_HIP_CHECK(hipHostMalloc(&b.p, want, hipHostMallocDefault));
      b.bytes = want;
    }
    return b.p;
  }
  void release() {
    for (auto& kv : bufs)
      if (kv.second.p) EIGX_HIP_CHECK(hipFree(kv.second.p));
    bufs.clear();
    for (auto& kv : hbufs)
      if (kv.second.p) EIGX_HIP_CHECK(hipHostFree(kv.second.p));
    hbufs.clear();
  }
};

struct CommState;  // comm.hip (peer windows + RCCL communicators for world / X / Y groups)

struct Context {
  bool initialized = false;
  int device = 0;
  Grid grid;
  hipStream_t stream = nullptr;       // compute stream
  hipStream_t side_stream = nullptr;  // collectives / copies overlapped with compute
  static constexpr int kAux = 6;
  hipStream_t aux[kAux] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // concurrent small GEMMs (D&C levels)
  hipEvent_t aux_ev[kAux + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // back-transformation plan prepared ahead (trbak_prepare_dev on the side stream during the D&C): event + key
  hipEvent_t bt_ev = nullptr;
  hipEvent_t dc_ev = nullptr;          // D&C buffers zero-filled ahead (band_dc_prepare)
  // D&C pipeline (one GPU): the secular / Loewner / eigenvector-row kernels of the NEXT pass run on dc_stream under the
  // big product of the current one; dc_b_ev = their completion, dc_z_ev = [new eigenvalues | next z] are on the host
  hipStream_t dc_stream = nullptr;
  // work the solver wants enqueued on the side stream when the D&C's last product starts (the T factors of the
  // back-transformation; see band_dc_dev for why not earlier)
  std::function<void()> dc_side_work;
  // stream of that work (== side_stream; a CU-masked stream of its own changed nothing: profiles/r04_bt_mask_ab.log)
  hipStream_t bt_stream = nullptr;
  hipEvent_t dc_b_ev = nullptr, dc_z_ev = nullptr;
  int dc_zero_n = 0; const double* dc_zero_qa = nullptr; const double* dc_zero_qb = nullptr;
  bool bt_ready = false;
  const double* bt_a = nullptr; double* bt_V = nullptr;
  int bt_n = 0, bt_mb = 0, bt_band = 0, bt_ldv = 0;
  Pool pool;
  CommState* comm = nullptr;
  int64_t errinfo = 0;
  double timers[16] = {0};
  // sampled HIP-event timing of the two roofline kernels (bench.py): every prof_stride-th launch of the
  // fused SYMV kernel and every trailing-update GEMM is bracketed by events on the compute stream
  int prof_stride = 0;  // 0 = off
  std::vector<hipEvent_t> prof_ev;   // pairs
  std::vector<double> prof_units;    // bytes (kind 0) or flops (kind 1) of the bracketed launch
  std::vector<int> prof_kind;
  size_t prof_used = 0;
  void prof_begin(int kind, double units, hipStream_t st) {
    if (prof_used + 2 > prof_ev.size()) {
      for (int q = 0; q < 2; ++q) { hipEvent_t e; EIGX_HIP_CHECK(hipEventCreate(&e)); prof_ev.push_back(e); }
    }
    prof_kind.push_back(kind);
    prof_units.push_back(units);
    EIGX_HIP_CHECK(hipEventRecord(prof_ev[prof_used], st));
  }
  void prof_end(hipStream_t st) {
    EIGX_HIP_CHECK(hipEventRecord(prof_ev[prof_used + 1], st));
    prof_used += 2;
  }
};

extern Context g_ctx;

// A nested one-GPU solve inside a multi-rank entry point (gathered, replicated problems: KMATH_EIGEN_GEV, eigen_h) runs with
// the one-rank grid; the guard puts the caller's grid back on EVERY way out -- a workspace allocation that fails inside
// the nested solve unwinds through here (DeviceAllocError), and a rank left on the one-rank grid would believe
// nranks == 1 while its communicator is still multi-rank.
struct GridSwap {
  Context& c;
  Grid saved;
  explicit GridSwap(Context& ctx) : c(ctx), saved(ctx.grid) { c.grid = Grid(); }
  ~GridSwap() { c.grid = saved; }
  GridSwap(const GridSwap&) = delete;
  GridSwap& operator=(const GridSwap&) = delete;
};

void comm_report_failure(Context& ctx, const char* what);   // comm.hip: sets this rank's and every peer's sticky failure word (P > 1)

// C-ABI boundary guard of the solver entry points: a failed workspace allocation becomes an error code
template <class F>
int eigx_guard(Context& ctx, F&& f) {
  try {
    return f();
  } catch (const DeviceAllocError& e) {
    fprintf(stderr, "[eigx] out of device memory: workspace '%s' needs %zu bytes\n", e.name.c_str(), e.bytes);
    comm_report_failure(ctx, "out of device memory on this rank");
    if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    return -8;   // EIGX_ERR_NO_MEMORY
  }
}

// comm.hip
int comm_get_unique_id(void* out128);
int comm_init(Context& ctx, const void* unique_id);
void comm_free(Context& ctx);

// band_reduce.hip: A (upper triangle) -> band (d, e(:,1..band)); reflectors left in A's columns
void band_reduce_dev(Context& ctx, int n, double* A, int lda, double* d, double* e, int lde, int m, int band);

// dc.hip: zero-fill of the D&C's Q buffers on the side stream, ahead of band_dc_dev (optional)
void band_dc_prepare(Context& ctx, int n);
// dc.hip: eigen-decomposition of the band matrix (d, e(:,1..band)); w ascending, z(ldz, nvec)
void band_dc_dev(Context& ctx, int n, int nvec, const double* d, const double* e, int lde, int band, double* w,
                 double* z, int ldz);

// bisect.hip: eigenvalues only of the band matrix by Sturm counts (multi-section); w ascending
void band_bisect_dev(Context& ctx, int n, const double* d, const double* e, int lde, int band, double* w);

// solver.hip
// eigenvector column blocks -> the callers' 2-D (block-)cyclic blocks (one all-to-all); see solver.hip
void cols_to_cyclic_dev(Context& ctx, int n, int nvec, int nb, int zc, int zc0, int zcnt, const double* zcols, int ldz,
                        double* z_user, int ldz_user, hipStream_t st);
int64_t solver_workspace_bytes(const Context& ctx, int n, int lda, int ldz, int mf, int mb);

}  // namespace eigx
