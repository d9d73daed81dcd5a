// derotate.hip -- cube_derotate / frame_rotate(imlib='vip-fft') (preproc/derotation.py:51-328, 331-399) and rotate_fft /
// _fft_shear (derotation.py:542-640): host driver (canvas geometry, angle split, path selection).
//
// The reference rotates every frame with three FFT shears on a 4x zero-padded canvas:
//   canvas (L x L, frame centred so that pixel N//2 sits on L//2) -> optional rot90 by q quarter
//   turns about pixel L//2 -> x-shear(a) -> y-shear(b) -> x-shear(a), a = tan(d/2), b = -sin(d),
//   each shear = per-line circular sinc shift with period Le (even), complex field carried through,
//   real part cropped at the end.
// Device paths (both the "real-split" decomposition: three passes of REAL shifts + rank-one Nyquist corrections):
//   power-of-two Le (frames of 128 / 256 / 512 / 1024 px): wave-resident FFT shears, derotate_fft2.hip;
//   any other size: derotate_direct2.hip (power-of-two circular convolutions from 129 px, exact correlations below).
// (The complex-field formulations of round 1 -- one complex transform per line, 44 N^3 correlations -- are gone: they
//  survived only as cross-checks of the real-split paths, which are tested against the float64 CPU restatement directly.)
#include "common.h"
#include "rot_common.h"

namespace vipmi {

int derotate_fft2(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                  float* out, int mask_nan, int mask_zero, float mask_v);  // derotate_fft2.hip (real-split, default)
int derotate_direct2(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                     float* out, int mask_nan, int mask_zero, float mask_v);  // derotate_direct2.hip (real-split correlations)

// host-side geometry / angle split (derotation.py:154-158, cosmetics.py:210-215, derotation.py:577-602)
static void rot_geometry(int N, RotGeom& g) {
  int n1 = (int)(N * 1.5);
  if (n1 % 2 != N % 2) n1 += 1;
  int L = (int)llround(n1 * (4.0 / 1.5));
  if (L % 2 != n1 % 2) L -= 1;
  g.N = N;
  g.L = L;
  g.Le = (L % 2 == 0) ? L : L - 1;
  g.off = L / 2 - N / 2;
  g.c = L / 2;
  g.Lc = (L % 2 == 0) ? L : L - 1;     // rot90 pivot sum: index i -> Lc - i
  g.alt0 = g.Lc - g.off - N + 1;
}

static double rint_half_even(double x) { return nearbyint(x); }   // default FE_TONEAREST = np.rint

static RotFrame rot_frame(double angle) {
  double a = angle;
  while (a < 0) a += 360;
  while (a > 360) a -= 360;
  double d;
  int q = 0;
  if (a > 45) {
    d = fmod(a, 90.0);
    if (d > 45) d = -(90 - d);
    q = (int)rint_half_even(a / 90.0);
  } else {
    d = a;
  }
  const double rad = d * (M_PI / 180.0);   // np.deg2rad
  RotFrame r;
  r.a = tan(rad / 2);
  r.b = -sin(rad);
  r.q = q & 3;
  r.pad = 0;
  return r;
}

int derotate_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n, int64_t N,
                 float* out, int mask_nan, int mask_zero, int method, float mask_v) {
  VIPMI_REQUIRE(in && out && angles_host, "derotate: null pointer");
  VIPMI_REQUIRE(n > 0 && N >= 2 && N <= 4096, "derotate: bad sizes n=%ld N=%ld", (long)n, (long)N);
  VIPMI_REQUIRE(in != out, "derotate: in-place operation not supported");
  StageScope sc(ctx, "derotate");
  RotGeom g;
  rot_geometry((int)N, g);
  static thread_local std::vector<RotFrame> h;
  h.resize(n);
  for (int64_t i = 0; i < n; ++i) h[i] = rot_frame(-angles_host[i]);   // cube_derotate: -angle_list[i]
  RotFrame* d_frames = nullptr;
  VIPMI_TRY(ws(ctx, "rot_frames", (size_t)n, &d_frames));
  VIPMI_TRY(ctx->upload_async("rot_frames", h.data(), sizeof(RotFrame) * n, d_frames));
  // power-of-two frames: N = Le/4 centred at 3Le/8 (the wave-resident FFT kernels rely on this alignment)
  bool use_fft = (g.Le == 512 || g.Le == 1024 || g.Le == 2048 || g.Le == 4096) && g.L == g.Le && g.N * 4 == g.Le &&
                 g.off * 8 == 3 * g.Le;
  if (method == VIPMI_ROT_DIRECT) use_fft = false;
  if (method == VIPMI_ROT_FFT && !use_fft) {
    set_error("derotate: FFT path needs a power-of-two padded length (frame size 128/256/512/1024), got Le=%d", g.Le);
    return VIPMI_ERR_UNSUPPORTED;
  }
  if (use_fft) return derotate_fft2(ctx, in, d_frames, g, n, out, mask_nan, mask_zero, mask_v);     // real-split FFT shears
  // any other padded length: the same decomposition as convolutions / correlations (derotate_direct2.hip)
  return derotate_direct2(ctx, in, d_frames, g, n, out, mask_nan, mask_zero, mask_v);
}

}  // namespace vipmi
