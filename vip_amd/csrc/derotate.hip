// derotate.hip -- cube_derotate / frame_rotate(imlib='vip-fft') (preproc/derotation.py:51-328,
// 331-399) and rotate_fft / _fft_shear (derotation.py:542-640), host driver + the generic
// ("direct") device path.
//
// The reference rotates every frame with three FFT shears on a 4x zero-padded canvas:
//   canvas (L x L, frame centred so that pixel N//2 sits on L//2) -> optional rot90 by q quarter
//   turns about pixel L//2 -> x-shear(a) -> y-shear(b) -> x-shear(a), a = tan(d/2), b = -sin(d),
//   each shear = per-line circular sinc shift with period Le (even), complex field carried through,
//   real part cropped at the end.
// A circular sinc shift by s of a line x (period Le, Le even) is the correlation
//   y[m] = sum_j x[j] D(m - j - s),  D(t) = sin(pi t)/(Le sin(pi t/Le)) * (cos(pi t/Le) - i sin(pi t/Le))
// (closed form of (1/Le) sum_k exp(2 pi i f_k t) over numpy's fftfreq set, Nyquist term included --
// that term is what makes the field complex).  The direct path evaluates exactly this sum, using the
// zero structure of the problem: shear 1 has N non-zero inputs per line, shear 2 has N non-zero inputs
// and N needed outputs, shear 3 has N needed outputs.  It works for ANY frame size (the padded length
// Le is arbitrary, e.g. 402 for 101-pixel frames) and costs O(N Le) per line; power-of-two Le (frames of
// 128/256/512/1024 px) take the FFT path in derotate_fft.hip instead.
#include "common.h"
#include "rot_common.h"

namespace vipmi {

namespace {

// D(n - s) for integer offset n, float64 evaluation, float32 storage
__device__ __forceinline__ float2 dirichlet(int n, double s, int Le) {
  const double t = (double)n - s;
  const double th = t / (double)Le;
  const double sr = sinpi(th), cr = cospi(th);
  const double st = sinpi(t);
  double re, im;
  if (fabs(sr) < 1e-290) {
    re = 1.0;
    im = 0.0;
  } else {
    re = st * cr / ((double)Le * sr);
    im = -st / (double)Le;
  }
  return make_float2((float)re, (float)im);
}

// ---- shear 1: rows (real input straight from the frame, rot90 folded into the index map) ----
// A1[f][yrel][X], X in [0,Le)
__global__ __launch_bounds__(256) void shear1_direct(const float* __restrict__ in,
                                                     const RotFrame* __restrict__ fr, RotGeom g,
                                                     float2* __restrict__ A1, int f0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* T = reinterpret_cast<float2*>(smem);     // Le entries
  float* x = smem + 2 * g.Le;                      // N entries
  const int fl = blockIdx.y, f = f0 + fl, yrel = blockIdx.x;
  const RotFrame p = fr[f];
  const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
  const int c0 = (p.q == 2 || p.q == 3) ? g.alt0 : g.off;
  const int Y = r0 + yrel;
  const double s = p.a * (double)(Y - g.c);
  for (int n = threadIdx.x; n < g.Le; n += blockDim.x) T[n] = dirichlet(n, s, g.Le);
  const float* frame = in + (int64_t)f * g.N * g.N;
  for (int j = threadIdx.x; j < g.N; j += blockDim.x) {
    const int X = c0 + j;
    int fy, fx;
    rot_src(p.q, Y, X, g, fy, fx);
    float v = 0.f;
    if (fy >= 0 && fy < g.N && fx >= 0 && fx < g.N) v = frame[fy * g.N + fx];
    x[j] = (v == v) ? v : 0.f;
  }
  __syncthreads();
  float2* orow = A1 + ((int64_t)fl * g.N + yrel) * g.Le;
  for (int X = threadIdx.x; X < g.Le; X += blockDim.x) {
    int idx = X - c0;
    idx %= g.Le;
    if (idx < 0) idx += g.Le;
    float re = 0.f, im = 0.f;
    for (int j = 0; j < g.N; ++j) {
      const float2 t = T[idx];
      const float v = x[j];
      re = fmaf(v, t.x, re);
      im = fmaf(v, t.y, im);
      idx = (idx == 0) ? g.Le - 1 : idx - 1;
    }
    orow[X] = make_float2(re, im);
  }
}

// ---- shear 2: columns.  A2[f][m][X] for output rows Yo = off + m ----
__global__ __launch_bounds__(256) void shear2_direct(const float2* __restrict__ A1,
                                                     const RotFrame* __restrict__ fr, RotGeom g,
                                                     float2* __restrict__ A2, int f0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* T = reinterpret_cast<float2*>(smem);     // Le entries
  float2* x = T + g.Le;                            // N entries
  const int fl = blockIdx.y, f = f0 + fl, X = blockIdx.x;
  const RotFrame p = fr[f];
  const int r0 = (p.q == 1 || p.q == 2) ? g.alt0 : g.off;
  const double s = p.b * (double)(X - g.c);
  for (int n = threadIdx.x; n < g.Le; n += blockDim.x) T[n] = dirichlet(n, s, g.Le);
  for (int j = threadIdx.x; j < g.N; j += blockDim.x) x[j] = A1[((int64_t)fl * g.N + j) * g.Le + X];
  __syncthreads();
  for (int m = threadIdx.x; m < g.N; m += blockDim.x) {
    int idx = (g.off + m) - r0;
    idx %= g.Le;
    if (idx < 0) idx += g.Le;
    float re = 0.f, im = 0.f;
    for (int j = 0; j < g.N; ++j) {
      const float2 t = T[idx];
      const float2 v = x[j];
      re += v.x * t.x - v.y * t.y;
      im += v.x * t.y + v.y * t.x;
      idx = (idx == 0) ? g.Le - 1 : idx - 1;
    }
    A2[((int64_t)fl * g.N + m) * g.Le + X] = make_float2(re, im);
  }
}

// ---- shear 3: rows Yo = off + m, real part at X = off + j; restores the NaN / zero mask ----
__global__ __launch_bounds__(256) void shear3_direct(const float2* __restrict__ A2,
                                                     const RotFrame* __restrict__ fr, RotGeom g,
                                                     const float* __restrict__ in, float* __restrict__ out,
                                                     int f0, int mask_nan, int mask_zero) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* T = reinterpret_cast<float2*>(smem);     // Le entries
  float2* x = T + g.Le;                            // Le entries
  const int fl = blockIdx.y, f = f0 + fl, m = blockIdx.x;
  const RotFrame p = fr[f];
  const int Y = g.off + m;
  const double s = p.a * (double)(Y - g.c);
  for (int n = threadIdx.x; n < g.Le; n += blockDim.x) {
    T[n] = dirichlet(n, s, g.Le);
    x[n] = A2[((int64_t)fl * g.N + m) * g.Le + n];
  }
  __syncthreads();
  const int64_t obase = ((int64_t)f * g.N + m) * g.N;
  for (int j = threadIdx.x; j < g.N; j += blockDim.x) {
    int idx = g.off + j;            // offset (Xo - Xin) for Xin = 0
    float re = 0.f;
    for (int X = 0; X < g.Le; ++X) {
      const float2 t = T[idx];
      const float2 v = x[X];
      re += v.x * t.x - v.y * t.y;
      idx = (idx == 0) ? g.Le - 1 : idx - 1;
    }
    const float src = in[obase + j];
    if (mask_nan && !(src == src)) re = __uint_as_float(0x7fc00000u);
    if (mask_zero && src == 0.f) re = 0.f;
    out[obase + j] = re;
  }
}

}  // namespace

int derotate_direct(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                    float* out, int mask_nan, int mask_zero) {
  const int64_t per_frame = (int64_t)g.N * g.Le;          // float2 elements per intermediate
  int64_t budget = ctx->opt("rot_ws_mb", 4096) * (int64_t)(1 << 20);
  int64_t chunk = budget / (2 * per_frame * (int64_t)sizeof(float2));
  if (chunk < 1) chunk = 1;
  if (chunk > n) chunk = n;
  if (chunk > 65535) chunk = 65535;
  float2 *A1 = nullptr, *A2 = nullptr;
  VIPMI_TRY(ws(ctx, "rot_a1", (size_t)(chunk * per_frame), &A1));
  VIPMI_TRY(ws(ctx, "rot_a2", (size_t)(chunk * per_frame), &A2));
  const size_t lds1 = (size_t)(2 * g.Le + g.N) * sizeof(float);
  const size_t lds2 = (size_t)(g.Le + g.N) * sizeof(float2);
  const size_t lds3 = (size_t)(2 * g.Le) * sizeof(float2);
  VIPMI_REQUIRE(lds3 <= 160 * 1024, "derotate(direct): padded length %d too large", g.Le);
  VIPMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shear1_direct),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  VIPMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shear2_direct),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  VIPMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(shear3_direct),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
  for (int64_t f0 = 0; f0 < n; f0 += chunk) {
    const unsigned nf = (unsigned)((n - f0) < chunk ? (n - f0) : chunk);
    hipLaunchKernelGGL(shear1_direct, dim3(g.N, nf), dim3(256), lds1, ctx->stream, in, d_frames, g, A1, (int)f0);
    hipLaunchKernelGGL(shear2_direct, dim3(g.Le, nf), dim3(256), lds2, ctx->stream, A1, d_frames, g, A2, (int)f0);
    hipLaunchKernelGGL(shear3_direct, dim3(g.N, nf), dim3(256), lds3, ctx->stream, A2, d_frames, g, in, out,
                       (int)f0, mask_nan, mask_zero);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  return VIPMI_OK;
}

int derotate_fft(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                 float* out, int mask_nan, int mask_zero);   // derotate_fft.hip
bool derotate_fft_supported(const RotGeom& g);
int derotate_fft2(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                  float* out, int mask_nan, int mask_zero);  // derotate_fft2.hip (real-split, default)
int derotate_direct2(vipmi_ctx* ctx, const float* in, const RotFrame* d_frames, const RotGeom& g, int64_t n,
                     float* out, int mask_nan, int mask_zero);  // derotate_direct2.hip (real-split correlations)

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
                 float* out, int mask_nan, int mask_zero, int method) {
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
  bool use_fft = derotate_fft_supported(g);
  if (method == VIPMI_ROT_DIRECT) use_fft = false;
  if (method == VIPMI_ROT_FFT && !use_fft) {
    set_error("derotate: FFT path needs a power-of-two padded length (frame size 128/256/512/1024), got Le=%d", g.Le);
    return VIPMI_ERR_UNSUPPORTED;
  }
  if (use_fft) {
    // rot_variant: 0 = real-split two-for-one transforms (default), 1 = complex field as the reference carries it
    if (ctx->opt("rot_variant", 0) == 1) return derotate_fft(ctx, in, d_frames, g, n, out, mask_nan, mask_zero);
    return derotate_fft2(ctx, in, d_frames, g, n, out, mask_nan, mask_zero);
  }
  // any other padded length: real-split correlations (rot_variant 1: the complex-field correlation of this file)
  if (ctx->opt("rot_variant", 0) == 1) return derotate_direct(ctx, in, d_frames, g, n, out, mask_nan, mask_zero);
  return derotate_direct2(ctx, in, d_frames, g, n, out, mask_nan, mask_zero);
}

}  // namespace vipmi
