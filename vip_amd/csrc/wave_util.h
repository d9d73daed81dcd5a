// wave_util.h -- full-wave (64 lanes) float64 reductions on DPP, shared by the eigensolvers.
#pragma once
#include <hip/hip_runtime.h>

namespace vipmi {

// Full-wave (64 lanes) sum of a double, result in every lane, without touching the LDS crossbar:
// xor-butterflies inside a 16-lane row with DPP (quad_perm xor1/xor2, row_half_mirror, row_mirror --
// all lanes of a row then hold the row sum), then the four row sums are read with v_readlane and added.
// (__shfl_xor on a double is two ds_bpermute round trips per step: ~18 dependent LDS round trips for
// the three dot products of one rotation, which dominated the eigensolver.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
  v += dpp_f64<0x141>(v);   // row_half_mirror      (i <-> 7-i)
  v += dpp_f64<0x140>(v);   // row_mirror           (i <-> 15-i)
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

}  // namespace vipmi
