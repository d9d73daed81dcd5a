// Interpolating rotation: cube_derotate / frame_rotate(imlib='opencv') of the reference (preproc/derotation.py:279-305),
// i.e. cv2.getRotationMatrix2D + cv2.warpAffine(float32, INTER_NEAREST | LINEAR | CUBIC | LANCZOS4, any border mode).
// opencv-python (pyproject.toml:56, unpinned) is absent from this image, so the kernel follows OpenCV's published
// algorithm (modules/imgproc/src/imgwarp.cpp, 4.x float path) and its parity against cv2 itself is NOT pinned:
//   * the affine map is inverted in double; source coordinates are evaluated in 1/1024-pixel fixed point
//     (round-to-nearest-even of M*x*1024 per column and of (M*y + b)*1024 per row, plus a rounding offset) and
//     truncated to 1/32 pixel; the 32 x 32 sub-pixel phases index separable float weight tables;
//   * taps outside the frame follow cv::borderInterpolate (constant 0 / replicate / reflect / reflect-101 / wrap);
//     NaN pixels are zeros (derotation.py:218).
// warp_affine_kernel: one thread per output pixel, taps from global memory (used for nearest; A/B option "warp_direct");
// warp_tile_kernel: source tiles staged in LDS.  4 B read + 4 B written per pixel, 4-10x cheaper than the 3-shear FFT
// rotation: the fast, lower-fidelity option the reference documents (README.rst:183); the default remains vip-fft.
#include <cmath>
#include <cstring>
#include <string>
#include "common.h"

namespace vipmi {

struct WarpFrame {
  double m[6];                   // dst (x, y) -> src: X = m0 x + m1 y + m2, Y = m3 x + m4 y + m5
};

constexpr int WARP_AB_BITS = 10, WARP_INTER_BITS = 5, WARP_TAB = 1 << WARP_INTER_BITS;

// cv::borderInterpolate: source index of an out-of-range coordinate, -1 = the constant border (value 0)
__device__ __forceinline__ int warp_border(int p, int len, int mode) {
  if ((unsigned)p < (unsigned)len) return p;
  switch (mode) {
    case VIPMI_BORDER_REPLICATE: return p < 0 ? 0 : len - 1;
    case VIPMI_BORDER_REFLECT:
    case VIPMI_BORDER_REFLECT101: {
      const int delta = mode == VIPMI_BORDER_REFLECT101;
      if (len == 1) return 0;
      do {
        p = p < 0 ? -p - 1 + delta : len - 1 - (p - len) - delta;
      } while ((unsigned)p >= (unsigned)len);
      return p;
    }
    case VIPMI_BORDER_WRAP:
      p %= len;
      return p < 0 ? p + len : p;
    default: return -1;
  }
}

__device__ __forceinline__ float warp_fetch(const float* __restrict__ src, int N, int gx, int gy, int border) {
  gx = warp_border(gx, N, border);
  gy = warp_border(gy, N, border);
  if (gx < 0 || gy < 0) return 0.f;
  const float v = src[(size_t)gy * N + gx];
  return v == v ? v : 0.f;
}

template <int TAPS>
__global__ __launch_bounds__(256) void warp_affine_kernel(const float* __restrict__ in, const WarpFrame* __restrict__ frames,
                                                          const float* __restrict__ tab, int N, float* __restrict__ out,
                                                          int border) {
  __shared__ float w1[WARP_TAB * (TAPS > 1 ? TAPS : 1)];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  if (TAPS > 1)
    for (int i = tid; i < WARP_TAB * TAPS; i += 256) w1[i] = tab[i];
  __syncthreads();
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= N || y >= N) return;
  const WarpFrame f = frames[blockIdx.z];
  const float* src = in + (size_t)blockIdx.z * N * N;
  const int round_delta = TAPS == 1 ? (1 << WARP_AB_BITS) / 2 : (1 << WARP_AB_BITS) / WARP_TAB / 2;
  const double sc = (double)(1 << WARP_AB_BITS);
  const int X0 = __double2int_rn((f.m[1] * y + f.m[2]) * sc) + round_delta + __double2int_rn(f.m[0] * x * sc);
  const int Y0 = __double2int_rn((f.m[4] * y + f.m[5]) * sc) + round_delta + __double2int_rn(f.m[3] * x * sc);
  float v = 0.f;
  if (TAPS == 1) {
    v = warp_fetch(src, N, X0 >> WARP_AB_BITS, Y0 >> WARP_AB_BITS, border);
  } else {
    const int X = X0 >> (WARP_AB_BITS - WARP_INTER_BITS), Y = Y0 >> (WARP_AB_BITS - WARP_INTER_BITS);
    const int sx = (X >> WARP_INTER_BITS) - (TAPS / 2 - 1), sy = (Y >> WARP_INTER_BITS) - (TAPS / 2 - 1);
    const float* wx = w1 + (X & (WARP_TAB - 1)) * TAPS;
    const float* wy = w1 + (Y & (WARP_TAB - 1)) * TAPS;
    if (border != VIPMI_BORDER_CONSTANT || (sx + TAPS > 0 && sx < N && sy + TAPS > 0 && sy < N)) {
#pragma unroll
      for (int r = 0; r < TAPS; ++r) {
        const float wr = wy[r];
#pragma unroll
        for (int c = 0; c < TAPS; ++c) v += warp_fetch(src, N, sx + c, sy + r, border) * (wr * wx[c]);
      }
    }
  }
  out[(size_t)blockIdx.z * N * N + (size_t)y * N + x] = v;
}

// 32 x 32 output tile per workgroup (4 pixels per thread) with its source footprint staged in LDS: the footprint of a
// rotated tile is at most 31 (|cos| + |sin|) + TAPS <= 44 + TAPS pixels wide, so a box of (46 + TAPS)^2 at most is
// loaded once (coalesced rows, border and NaN pixels already replaced by 0, ~10 independent loads in flight per thread
// to cover the HBM latency) and the TAPS^2 taps of every pixel are LDS reads without bounds checks, evaluated
// separably (row sums with the x weights, then the y weights: TAPS (TAPS + 1) FMAs instead of 2 TAPS^2 operations
// with OpenCV's product table; the difference is float rounding).  Reading the taps from global memory makes every
// wave load touch ~20 cache lines of a slanted source line: 7 ms for 400 x 512^2 lanczos4 frames.  The fixed-point
// coordinates are sums of a monotone function of x and one of y, so their extremes over the tile are at its corners.
template <int TAPS>
__global__ __launch_bounds__(256) void warp_tile_kernel(const float* __restrict__ in, const WarpFrame* __restrict__ frames,
                                                        const float* __restrict__ tab, int N, float* __restrict__ out,
                                                        int border) {
  constexpr int TS = 32, PPT = 4, BOX = 46 + TAPS, LS = BOX + 1, H = TAPS / 2 - 1;
  __shared__ float w1[WARP_TAB * TAPS];
  __shared__ float tile[BOX * LS];
  __shared__ int colx[TS], coly[TS], rowx[TS], rowy[TS];
  const int tid = threadIdx.x, tx = tid & (TS - 1), ty = tid >> 5;
  for (int i = tid; i < WARP_TAB * TAPS; i += 256) w1[i] = tab[i];
  const WarpFrame f = frames[blockIdx.z];
  const float* src = in + (size_t)blockIdx.z * N * N;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int x = x0 + tx;
  const int round_delta = (1 << WARP_AB_BITS) / WARP_TAB / 2;
  const double sc = (double)(1 << WARP_AB_BITS);
  // fixed-point coordinate = (row term of y) + (column term of x); the tile's terms are exchanged through LDS
  const int bxx = __double2int_rn(f.m[0] * x * sc), byy = __double2int_rn(f.m[3] * x * sc);
  int ax[PPT], ay[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int y = y0 + ty + 8 * p;
    ax[p] = __double2int_rn((f.m[1] * y + f.m[2]) * sc) + round_delta;
    ay[p] = __double2int_rn((f.m[4] * y + f.m[5]) * sc) + round_delta;
    if (tx == 0) {
      rowx[ty + 8 * p] = ax[p];
      rowy[ty + 8 * p] = ay[p];
    }
  }
  if (ty == 0) {
    colx[tx] = bxx;
    coly[tx] = byy;
  }
  __syncthreads();
  const int bx = ((min(rowx[0], rowx[TS - 1]) + min(colx[0], colx[TS - 1])) >> WARP_AB_BITS) - H;   // first column / row
  const int by = ((min(rowy[0], rowy[TS - 1]) + min(coly[0], coly[TS - 1])) >> WARP_AB_BITS) - H;
  const int bw = ((max(rowx[0], rowx[TS - 1]) + max(colx[0], colx[TS - 1])) >> WARP_AB_BITS) + TAPS - H - bx;  // <= 45 + TAPS
  const int bh = min(((max(rowy[0], rowy[TS - 1]) + max(coly[0], coly[TS - 1])) >> WARP_AB_BITS) + TAPS - H - by, BOX);
  {
    const int c = tid & 63, gx = warp_border(bx + c, N, border);
    const bool okx = c < bw && gx >= 0;
    for (int r = tid >> 6; r < bh; r += 4) {
      const int gy = warp_border(by + r, N, border);
      float v = 0.f;
      if (okx && gy >= 0) v = src[(size_t)gy * N + gx];
      if (c < BOX) tile[r * LS + c] = v == v ? v : 0.f;
    }
  }
  __syncthreads();
  if (x >= N) return;
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int y = y0 + ty + 8 * p;
    if (y >= N) break;
    const int X = (ax[p] + bxx) >> (WARP_AB_BITS - WARP_INTER_BITS), Y = (ay[p] + byy) >> (WARP_AB_BITS - WARP_INTER_BITS);
    const int sx = min(max((X >> WARP_INTER_BITS) - H - bx, 0), BOX - TAPS);
    const int sy = min(max((Y >> WARP_INTER_BITS) - H - by, 0), BOX - TAPS);
    const float* wx = w1 + (X & (WARP_TAB - 1)) * TAPS;
    const float* wy = w1 + (Y & (WARP_TAB - 1)) * TAPS;
    float wxr[TAPS];
#pragma unroll
    for (int c = 0; c < TAPS; ++c) wxr[c] = wx[c];
    const float* t = tile + sy * LS + sx;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < TAPS; ++r) {
      float srow = 0.f;
#pragma unroll
      for (int c = 0; c < TAPS; ++c) srow = fmaf(t[r * LS + c], wxr[c], srow);
      v = fmaf(srow, wy[r], v);
    }
    out[(size_t)blockIdx.z * N * N + (size_t)y * N + x] = v;
  }
}

// separable weights of the 32 sub-pixel phases (imgwarp.cpp: interpolateLinear / interpolateCubic / interpolateLanczos4)
static void warp_weights(int taps, std::vector<float>& tab) {
  tab.assign((size_t)WARP_TAB * taps, 0.f);
  for (int i = 0; i < WARP_TAB; ++i) {
    const float x = (float)i / WARP_TAB;
    float* c = tab.data() + (size_t)i * taps;
    if (taps == 2) {
      c[0] = 1.f - x;
      c[1] = x;
    } else if (taps == 4) {
      const float A = -0.75f;
      c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
      c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
      c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
      c[3] = 1.f - c[0] - c[1] - c[2];
    } else {
      if (x < 1.1920929e-7f) {
        c[3] = 1.f;
        continue;
      }
      const double s45 = 0.70710678118654752440084436210485, pi = 3.1415926535897932384626433832795;
      const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
      const double y0 = -(x + 3) * pi * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
      float sum = 0.f;
      for (int k = 0; k < 8; ++k) {
        const double yk = -(x + 3 - k) * pi * 0.25;
        c[k] = (float)((cs[k][0] * s0 + cs[k][1] * c0) / (yk * yk));
        sum += c[k];
      }
      sum = 1.f / sum;
      for (int k = 0; k < 8; ++k) c[k] *= sum;
    }
  }
}

int rotate_interp_f32(vipmi_ctx* ctx, const float* in, const double* angles_host, int64_t n, int64_t N, double cx,
                      double cy, int interp, int border, float* out) {
  VIPMI_REQUIRE(in && out && angles_host, "rotate_interp: null pointer");
  VIPMI_REQUIRE(n >= 1 && n <= 65535 && N >= 1 && N <= 16384, "rotate_interp: bad sizes (n=%lld, N=%lld)", (long long)n,
                (long long)N);
  VIPMI_REQUIRE(in != out, "rotate_interp: in-place rotation is not supported");
  const int taps = interp == VIPMI_INTERP_NEAREST ? 1 : interp == VIPMI_INTERP_BILINEAR ? 2 : interp == VIPMI_INTERP_BICUBIC ? 4
                   : interp == VIPMI_INTERP_LANCZOS4 ? 8 : 0;
  VIPMI_REQUIRE(taps != 0, "rotate_interp: unknown interpolation %d", interp);
  VIPMI_REQUIRE(border >= VIPMI_BORDER_CONSTANT && border <= VIPMI_BORDER_WRAP, "rotate_interp: unknown border mode %d", border);
  StageScope scope(ctx, "warp");
  std::vector<WarpFrame> h((size_t)n);
  const double pi = 3.14159265358979323846;
  for (int64_t i = 0; i < n; ++i) {
    // forward matrix of cv2.getRotationMatrix2D((cx, cy), -angle, 1), inverted as warpAffine does
    const double ang = -angles_host[i] * pi / 180.0;
    const double a = std::cos(ang), b = std::sin(ang);
    double M[6] = {a, b, (1 - a) * cx - b * cy, -b, a, b * cx + (1 - a) * cy};
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11;
    M[1] *= -D;
    M[3] *= -D;
    M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1;
    M[5] = b2;
    std::memcpy(h[i].m, M, sizeof(M));
  }
  WarpFrame* d_frames = nullptr;
  VIPMI_TRY(ws(ctx, "warp_frames", (size_t)n, &d_frames));
  VIPMI_TRY(ctx->upload_async("warp_frames", h.data(), sizeof(WarpFrame) * n, d_frames));
  float* d_tab = nullptr;
  if (taps > 1) {
    std::vector<float> tab;
    warp_weights(taps, tab);
    void* p = nullptr;
    const std::string nm = "warp_tab" + std::to_string(taps);
    VIPMI_TRY(ctx->upload_cached(nm.c_str(), nm, tab.data(), tab.size() * sizeof(float), &p));
    d_tab = (float*)p;
  }
  dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(N, 4), (unsigned)n), block(64, 4);
  dim3 tgrid((unsigned)cdiv(N, 32), (unsigned)cdiv(N, 32), (unsigned)n);
  const bool direct = ctx->opt("warp_direct", 0) != 0;         // A/B switch: taps from global memory
  switch (taps) {
    case 1: warp_affine_kernel<1><<<grid, block, 0, ctx->stream>>>(in, d_frames, d_tab, (int)N, out, border); break;
    case 2:
      if (direct) warp_affine_kernel<2><<<grid, block, 0, ctx->stream>>>(in, d_frames, d_tab, (int)N, out, border);
      else warp_tile_kernel<2><<<tgrid, 256, 0, ctx->stream>>>(in, d_frames, d_tab, (int)N, out, border);
      break;
    case 4:
      if (direct) warp_affine_kernel<4><<<grid, block, 0, ctx->stream>>>(in, d_frames, d_tab, (int)N, out, border);
      else warp_tile_kernel<4><<<tgrid, 256, 0, ctx->stream>>>(in, d_frames, d_tab, (int)N, out, border);
      break;
    default:
      if (direct) warp_affine_kernel<8><<<grid, block, 0, ctx->stream>>>(in, d_frames, d_tab, (int)N, out, border);
      else warp_tile_kernel<8><<<tgrid, 256, 0, ctx->stream>>>(in, d_frames, d_tab, (int)N, out, border);
      break;
  }
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace vipmi
