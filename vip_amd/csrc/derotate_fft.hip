// derotate_fft.hip -- FFT fast path of the 3-shear rotation (power-of-two padded lengths).
#include "common.h"
#include "rot_common.h"

namespace vipmi {

bool derotate_fft_supported(const RotGeom& g) { (void)g; return false; }

int derotate_fft(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                 float* out, int mask_nan, int mask_zero) {
  (void)ctx; (void)in; (void)d_frames; (void)g; (void)n; (void)out; (void)mask_nan; (void)mask_zero;
  set_error("derotate: FFT path not built");
  return VIPMI_ERR_UNSUPPORTED;
}

}  // namespace vipmi
