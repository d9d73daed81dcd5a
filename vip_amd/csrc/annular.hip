// annular.hip -- per-frame library PCA of one annulus segment (psfsub/pca_local.py:830-909).
#include "common.h"

namespace vipmi {

int annular_residuals_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t npx, const int32_t* lib_idx,
                          const int32_t* lib_len, int64_t max_lib, int64_t ncomp, float* residuals) {
  (void)ctx; (void)A; (void)n; (void)npx; (void)lib_idx; (void)lib_len; (void)max_lib; (void)ncomp; (void)residuals;
  set_error("annular_residuals: not implemented yet");
  return VIPMI_ERR_UNSUPPORTED;
}

}  // namespace vipmi
