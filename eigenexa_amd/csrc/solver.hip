// solver.hip -- drivers eigx_sx / eigx_s (placeholder until the stages land).
#include "eigx_context.h"
#include "../../include/eigenexa_amd.h"

namespace eigx {
int64_t solver_workspace_bytes(const Context&, int n, int lda, int ldz, int mf, int mb) {
  (void)lda; (void)ldz; (void)mf; (void)mb;
  return (int64_t)n * n * 8 * 3;
}
}

extern "C" {
int eigx_sx(int, int, double*, int, double*, double*, int, int, int, char) { return EIGX_ERR_INTERNAL; }
int eigx_s(int, int, double*, int, double*, double*, int, int, int, char) { return EIGX_ERR_INTERNAL; }
int eigx_sx_dev(int, int, double*, int, double*, double*, int, int, int, char) { return EIGX_ERR_INTERNAL; }
int eigx_s_dev(int, int, double*, int, double*, double*, int, int, int, char) { return EIGX_ERR_INTERNAL; }
int eigx_band_reduce_dev(int n, double* a, int lda, double* d, double* e, int lde, int mf, int band) {
  using namespace eigx;
  if (!g_ctx.initialized) return EIGX_ERR_NOT_INITIALIZED;
  if (n <= 0 || lda < n || (lda & 1) || lde < n || (band != 1 && band != 2)) return EIGX_ERR_BAD_ARG;
  if (g_ctx.grid.nranks != 1) return EIGX_ERR_INTERNAL;
  band_reduce_dev(g_ctx, n, a, lda, d, e, lde, mf > 0 ? mf : 128, band);
  EIGX_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  return EIGX_OK;
}
int eigx_band_dc_dev(int, int, const double*, const double*, int, int, double*, double*, int) { return EIGX_ERR_INTERNAL; }
int eigx_trbak_dev(int, int, const double*, int, double*, int, const double*, int, int, int) { return EIGX_ERR_INTERNAL; }
}
