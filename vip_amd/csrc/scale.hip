// scale.hip -- prepare_matrix pieces (var/shapes.py:740-781 matrix_scaling via
// sklearn.preprocessing.scale; var/shapes.py:38-113 mask_circle; annulus gather/scatter
// psfsub/pca_local.py:713,787) plus small conversion helpers.  All HBM-bound streaming kernels.
//
// Scaling semantics (sklearn 1.7.2 `scale`): mean and population std (ddof=0) of the ORIGINAL
// values along the axis; x <- (x - mean) [/ std]; std < 10*eps(float32) -> 1.  Statistics are
// accumulated in float64 (two-pass for the variance), so the data-dependent re-centring passes
// sklearn performs in float32 (|delta| ~ 1e-7 * scale) are unnecessary here.
#include "common.h"

namespace vipmi {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- temporal (axis 0): one thread per pixel column, coalesced across threads ----
__global__ void temp_stats_kernel(const float* __restrict__ in, int n, int64_t P, int with_std,
                                  float* __restrict__ mean, float* __restrict__ inv_std) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P;
       p += (int64_t)gridDim.x * blockDim.x) {
    // eight loads in flight per thread (one load per dependent add left the pass at a third of the HBM rate); same order of additions
    double s = 0;
    int f = 0;
    for (; f + 8 <= n; f += 8) {
      float v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = in[(int64_t)(f + u) * P + p];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v8[u];
    }
    for (; f < n; ++f) s += (double)in[(int64_t)f * P + p];
    const double mu = s / n;
    mean[p] = (float)mu;
    if (with_std) {
      double v = 0;
      f = 0;
      for (; f + 8 <= n; f += 8) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = in[(int64_t)(f + u) * P + p];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double d = (double)v8[u] - mu;
          v += d * d;
        }
      }
      for (; f < n; ++f) {
        double d = (double)in[(int64_t)f * P + p] - mu;
        v += d * d;
      }
      float sd = (float)sqrt(v / n);
      if (sd < 10.f * 1.1920929e-07f) sd = 1.f;
      inv_std[p] = sd;
    }
  }
}

// blockIdx.x over the pixels, blockIdx.y over groups of 8 frames: the per-pixel statistics are loaded once per thread and no
// 64-bit e % P is paid per element (the flat-index loop spent more instructions on it than on everything else)
__global__ void temp_apply_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int64_t P,
                                  const float* __restrict__ mean, const float* __restrict__ sd) {
  const int f0 = blockIdx.y * 8;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const float m = mean[p];
    const float s = sd ? sd[p] : 1.f;
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (f0 + u < n) ? in[(int64_t)(f0 + u) * P + p] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float r = v[u] - m;
      if (sd) r /= s;
      if (f0 + u < n) out[(int64_t)(f0 + u) * P + p] = r;
    }
  }
}

// ---- spatial (axis 1): one workgroup per frame ----
__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

__global__ __launch_bounds__(1024) void spat_scale_kernel(const float* __restrict__ in,
                                                          float* __restrict__ out, int64_t P,
                                                          int with_std) {
  __shared__ double sh[16];
  const float* row = in + (int64_t)blockIdx.x * P;
  float* orow = out + (int64_t)blockIdx.x * P;
  double s = 0;
  for (int64_t p = threadIdx.x; p < P; p += blockDim.x) s += (double)row[p];
  const double mu = block_sum(s, sh) / (double)P;
  float sd = 1.f;
  if (with_std) {
    double v = 0;
    for (int64_t p = threadIdx.x; p < P; p += blockDim.x) {
      double d = (double)row[p] - mu;
      v += d * d;
    }
    sd = (float)sqrt(block_sum(v, sh) / (double)P);
    if (sd < 10.f * 1.1920929e-07f) sd = 1.f;
  }
  const float muf = (float)mu;
  for (int64_t p = threadIdx.x; p < P; p += blockDim.x) {
    float v = row[p] - muf;
    if (with_std) v /= sd;
    orow[p] = v;
  }
}

__global__ void mask_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int64_t P,
                            const uint8_t* __restrict__ mask, float fill) {
  const int64_t f0 = (int64_t)blockIdx.y * 8;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const bool m = mask[p] != 0;
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (!m && f0 + u < n) ? in[(f0 + u) * P + p] : fill;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (f0 + u < n) out[(f0 + u) * P + p] = v[u];
  }
}

// blockIdx.x over the pixel list, blockIdx.y over groups of FPB frames: no 64-bit division per element (the flat-index version
// spent most of its 55 us per annulus of C3 on e / npx and e % npx), the pixel index loaded once per thread, FPB independent
// loads in flight per thread (one element per thread -- a workgroup per frame and 256 pixels -- was slower than the division: 84 us)
constexpr int GS_FPB = 8;
__global__ void gather_kernel(const float* __restrict__ cube, int64_t n, int64_t P,
                              const int32_t* __restrict__ pix, int64_t npx, float* __restrict__ A) {
  const int64_t f0 = (int64_t)blockIdx.y * GS_FPB;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < npx; j += (int64_t)gridDim.x * blockDim.x) {
    const int32_t p = pix[j];                 // negative = padding column (npx rounded up to 4)
    float v[GS_FPB];
#pragma unroll
    for (int u = 0; u < GS_FPB; ++u) v[u] = (p >= 0 && f0 + u < n) ? cube[(f0 + u) * P + p] : 0.f;
#pragma unroll
    for (int u = 0; u < GS_FPB; ++u)
      if (f0 + u < n) A[(f0 + u) * npx + j] = v[u];
  }
}

__global__ void scatter_kernel(const float* __restrict__ A, int64_t n, int64_t P,
                               const int32_t* __restrict__ pix, int64_t npx, float* __restrict__ cube) {
  const int64_t f0 = (int64_t)blockIdx.y * GS_FPB;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < npx; j += (int64_t)gridDim.x * blockDim.x) {
    const int32_t p = pix[j];
    if (p < 0) continue;
    float v[GS_FPB];
#pragma unroll
    for (int u = 0; u < GS_FPB; ++u) v[u] = (f0 + u < n) ? A[(f0 + u) * npx + j] : 0.f;
#pragma unroll
    for (int u = 0; u < GS_FPB; ++u)
      if (f0 + u < n) cube[(f0 + u) * P + p] = v[u];
  }
}

// evecs: rows = eigenvectors (float64, k x n used).  Emits E in both float32 layouts the
// projection kernels read, and 1/sigma.  Components with eval <= 1e-12*eval[0] (numerically null:
// their projection coefficient is zero anyway) are zeroed so that garbage directions never enter
// the reconstruction.
__global__ void convert_evecs_kernel(const double* __restrict__ evecs, const double* __restrict__ evals,
                                     int n, int k, float* __restrict__ Ekn, int nld,
                                     float* __restrict__ Enk, int kld, float* __restrict__ inv_sigma) {
  const double thr = evals[0] * 1e-12;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kld * nld; e += gridDim.x * blockDim.x) {
    const int c = e / nld, f = e % nld;
    float v = 0.f;
    if (c < k && f < n && evals[c] > thr) v = (float)evecs[(int64_t)c * n + f];
    if (c < k) Ekn[(int64_t)c * nld + f] = v;   // [k][nld]  (subtract kernel's Ct)
    if (f < n) Enk[(int64_t)f * kld + c] = v;   // [n][kld]  (rowspace kernel's Wt)
    if (f == 0 && c < k) {
      // singular value = sqrt(eigenvalue of G) ; eigenvalue = column norm of G V
      const double ev = evals[c];
      inv_sigma[c] = (ev > thr && ev > 0) ? (float)(1.0 / sqrt(ev)) : 0.f;
    }
  }
}

__global__ void convert_coeffs_kernel(const double* __restrict__ C, int n, int k, float* __restrict__ Ct,
                                      int nld) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < k * nld; e += gridDim.x * blockDim.x) {
    const int c = e / nld, f = e % nld;
    Ct[e] = (f < n) ? (float)C[(int64_t)f * k + c] : 0.f;
  }
}

__global__ void scale_rows_kernel(const float* __restrict__ in, const float* __restrict__ rs, int64_t k,
                                  int64_t P, float* __restrict__ out) {
  const int64_t total = k * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x)
    out[e] = in[e] * rs[e / P];
}

__global__ void lincomb_kernel(const float* __restrict__ x, const float* __restrict__ y, float a, float b,
                               int64_t total, float* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = y ? a * x[e] + b * y[e] : a * x[e];
}

int grid_for(int64_t total, int cap) {
  int64_t b = cdiv(total, 256);
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

int scale_f32(vipmi_ctx* ctx, const float* in, float* out, int64_t n, int64_t P, int mode) {
  VIPMI_REQUIRE(in && out, "scale: null pointer");
  VIPMI_REQUIRE(n > 0 && P > 0, "scale: bad sizes");
  VIPMI_REQUIRE(mode >= VIPMI_SCALE_TEMP_MEAN && mode <= VIPMI_SCALE_SPAT_STANDARD,
                "Scaling mode not recognized");
  StageScope sc(ctx, "scale");
  if (mode == VIPMI_SCALE_TEMP_MEAN || mode == VIPMI_SCALE_TEMP_STANDARD) {
    const int with_std = mode == VIPMI_SCALE_TEMP_STANDARD;
    float *mean = nullptr, *sd = nullptr;
    VIPMI_TRY(ws(ctx, "scale_mean", (size_t)P, &mean));
    VIPMI_TRY(ws(ctx, "scale_sd", (size_t)P, &sd));
    hipLaunchKernelGGL(temp_stats_kernel, dim3(grid_for(P, 8192)), dim3(256), 0, ctx->stream, in, (int)n,
                       P, with_std, mean, sd);
    VIPMI_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(temp_apply_kernel, dim3(grid_for(P, 2048), (unsigned)cdiv(n, 8)), dim3(256), 0, ctx->stream, in, out,
                       (int)n, P, mean, with_std ? sd : nullptr);
    VIPMI_CHECK_HIP(hipGetLastError());
  } else {
    const int with_std = mode == VIPMI_SCALE_SPAT_STANDARD;
    hipLaunchKernelGGL(spat_scale_kernel, dim3((unsigned)n), dim3(1024), 0, ctx->stream, in, out, P,
                       with_std);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  return VIPMI_OK;
}

int apply_mask_f32(vipmi_ctx* ctx, const float* in, float* out, int64_t n, int64_t P,
                   const uint8_t* mask, float fill) {
  VIPMI_REQUIRE(in && out && mask, "apply_mask: null pointer");
  VIPMI_REQUIRE(n > 0 && P > 0, "apply_mask: bad sizes");
  hipLaunchKernelGGL(mask_kernel, dim3(grid_for(P, 2048), (unsigned)cdiv(n, 8)), dim3(256), 0, ctx->stream, in, out, n, P,
                     mask, fill);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int gather_f32(vipmi_ctx* ctx, const float* cube, int64_t n, int64_t P, const int32_t* pix,
               int64_t npx, float* A) {
  VIPMI_REQUIRE(cube && pix && A && n > 0 && P > 0 && npx > 0, "gather: bad arguments");
  hipLaunchKernelGGL(gather_kernel, dim3(grid_for(npx, 1024), (unsigned)cdiv(n, GS_FPB)), dim3(256), 0, ctx->stream, cube, n, P,
                     pix, npx, A);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int scatter_f32(vipmi_ctx* ctx, const float* A, int64_t n, int64_t P, const int32_t* pix, int64_t npx,
                float* cube) {
  VIPMI_REQUIRE(cube && pix && A && n > 0 && P > 0 && npx > 0, "scatter: bad arguments");
  hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(npx, 1024), (unsigned)cdiv(n, GS_FPB)), dim3(256), 0, ctx->stream, A, n, P,
                     pix, npx, cube);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

// Ekn: [k][nld] with nld = roundup(n,32);  Enk: [n][kld] with kld = roundup(k,32)
int convert_evecs(vipmi_ctx* ctx, const double* evecs, const double* evals, int64_t n, int64_t k,
                  float* Ekn, float* Enk, float* inv_sigma) {
  const int nld = (int)cdiv(n, 32) * 32, kld = (int)cdiv(k, 32) * 32;
  hipLaunchKernelGGL(convert_evecs_kernel, dim3(grid_for((int64_t)kld * nld, 1024)), dim3(256), 0,
                     ctx->stream, evecs, evals, (int)n, (int)k, Ekn, nld, Enk, kld, inv_sigma);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int lincomb_f32(vipmi_ctx* ctx, const float* x, const float* y, float a, float b, int64_t total, float* out) {
  VIPMI_REQUIRE(x && out && total > 0, "lincomb: bad arguments");
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, ctx->stream, x, y, a, b, total, out);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int convert_coeffs(vipmi_ctx* ctx, const double* C, int64_t n, int64_t k, float* Ct, int nld) {
  hipLaunchKernelGGL(convert_coeffs_kernel, dim3(grid_for((int64_t)k * nld, 1024)), dim3(256), 0, ctx->stream,
                     C, (int)n, (int)k, Ct, nld);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

int scale_rows(vipmi_ctx* ctx, const float* in, const float* rowscale, int64_t k, int64_t P, float* out) {
  hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(k * P, 8192)), dim3(256), 0, ctx->stream, in, rowscale,
                     k, P, out);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}

}  // namespace vipmi
