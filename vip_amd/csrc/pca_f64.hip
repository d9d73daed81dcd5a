// pca_f64.hip -- full-frame ADI PCA of a FLOAT64 cube (psfsub/pca_fullfr.py:1552-1737 with a float64 `cube`: the reference keeps
// the caller's dtype through prepare_matrix / svd_wrapper, SURVEY a9 "f64 if f64 in").
//
// Rounding a cube of detector counts (7000 +- 45) to float32 costs 2^-24 * 7000 = 4e-4 per sample before anything is computed, and the
// float32 projection then cancels counts of 7e3 down to residuals of ~45: the final frame ends 2e-3 from the reference's float64
// result (golden g28).  Both losses come from the OFFSET, not from the signal, so the offset is carried in float64 and the float32
// kernels only ever see what is left:
//     mu_p = mean over the frames of M[., p]  (float64),      D = float32(M - 1 mu^T)          (|D| ~ 45: rounding 4e-6)
//   scaling 'temp-mean' / 'temp-standard': the scaled matrix IS D (/ sigma): the ordinary float32 pipeline on it is the reference's.
//   scaling None: the decomposition is that of M = D + 1 mu^T,
//     G = M M^T = D D^T + 1 (D mu)^T + (D mu) 1^T + |mu|^2 1 1^T         (D D^T: the exact Gram kernels; the rest in float64)
//     residual = (I - E^T E) M = [D - E^T (E D)] + r mu^T,   r = 1 - E^T (E 1)   (float64, n numbers)
//   -- the bracket is the ordinary projection of the small matrix D, and the rank-one term is one more "component" of the
//   subtraction kernel (coefficients -r, image mu): nothing of size 7e3 is ever subtracted from anything of size 7e3 in float32.
// Everything after the residuals (FFT derotation, median) is the float32 path: residuals are small whatever the counts were.
#include "common.h"

namespace vipmi {

namespace {

// one thread per pixel column (coalesced across threads): temporal mean (and population std) in float64, D = float32((x - mu) / sd)
// mode 0 / 1: centre only; 2: 'temp-standard' (sklearn: sd < 10 eps(float64) -> 1).  mask: 1 = masked (the sample is 0).
// pix (optional, round 6: the annular front): column p of D is pixel pix[p] of the cube (rows of Psrc samples; -1 = a zero column):
// the gather of all segments' pixels and the centring in one pass.
__global__ void center_f64_kernel(const double* __restrict__ Msrc, int n, int64_t P, const uint8_t* __restrict__ mask, int mode,
                                  float* __restrict__ D, double* __restrict__ mu, float* __restrict__ mu32,
                                  const int32_t* __restrict__ pix = nullptr, int64_t Psrc = 0) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = pix ? pix[p] : p;
    const double* M = Msrc + src - p;                 // M[f * ldm + p] below is Msrc[f * ldm + src]
    const int64_t ldm = pix ? Psrc : P;
    if ((mask && mask[p]) || src < 0) {
      for (int f = 0; f < n; ++f) D[(int64_t)f * P + p] = 0.f;
      mu[p] = 0.0;
      if (mu32) mu32[p] = 0.f;
      continue;
    }
    double s = 0.0;                          // (eight loads in flight; the frames still summed in index order)
    int f0 = 0;
    for (; f0 + 8 <= n; f0 += 8) {
      double v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = M[(int64_t)(f0 + u) * ldm + p];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v8[u];
    }
    for (; f0 < n; ++f0) s += M[(int64_t)f0 * ldm + p];
    const double m = s / n;
    double sd = 1.0;
    if (mode == 2) {
      double v = 0.0;
      for (int f = 0; f < n; ++f) {
        const double d = M[(int64_t)f * ldm + p] - m;
        v += d * d;
      }
      sd = sqrt(v / n);
      if (sd < 10.0 * 2.220446049250313e-16) sd = 1.0;
    }
    const double inv = 1.0 / sd;
    f0 = 0;
    for (; f0 + 8 <= n; f0 += 8) {
      double v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = M[(int64_t)(f0 + u) * ldm + p];
#pragma unroll
      for (int u = 0; u < 8; ++u) D[(int64_t)(f0 + u) * P + p] = (float)((v8[u] - m) * inv);
    }
    for (; f0 < n; ++f0) D[(int64_t)f0 * P + p] = (float)((M[(int64_t)f0 * ldm + p] - m) * inv);
    mu[p] = m;
    if (mu32) mu32[p] = (float)m;
  }
}

__device__ __forceinline__ double block_sum256(double v, double* sh) {      // 256 threads, fixed order: deterministic
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// g[f] = sum_p D[f, p] mu[p] (block f < n), g[n] = sum_p mu[p]^2 (block n): float64
__global__ __launch_bounds__(256) void offset_dots_kernel(const float* __restrict__ D, const double* __restrict__ mu, int n, int64_t P,
                                                          double* __restrict__ g) {
  __shared__ double sh[4];
  const int f = blockIdx.x;
  double s = 0.0;
  if (f < n) {
    const float* row = D + (int64_t)f * P;
    for (int64_t p = threadIdx.x; p < P; p += 256) s += (double)row[p] * mu[p];
  } else {
    for (int64_t p = threadIdx.x; p < P; p += 256) s += mu[p] * mu[p];
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) g[f] = s;
}

// G[i, j] += g[i] u[j] + u[i] g[j] + g[n] u[i] u[j]   (the Gram matrix of D + u mu^T from that of D; u == nullptr: ones)
// (blockIdx.y = matrix of a batch: G[y][n][n], g[y][n + 1])
__global__ void gram_offset_kernel(double* __restrict__ G, const double* __restrict__ g, int n, const double* __restrict__ u = nullptr) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * n) return;
  G += (size_t)blockIdx.y * n * n;
  g += (size_t)blockIdx.y * (n + 1);
  const int i = e / n, j = e - i * n;
  if (u) G[e] += g[i] * u[j] + u[i] * g[j] + g[n] * u[i] * u[j];
  else G[e] += g[i] + g[j] + g[n];
}

// ---- the spatial scalings ('spat-mean' / 'spat-standard': sklearn scale(axis=1), var/shapes.py:740-781) of a float64 cube ----
// The scaled matrix is  S (M - m 1^T),  m = the frames' means, S = diag(u), u = 1 / the frames' standard deviations (ones for
// 'spat-mean').  With mu = the per-pixel temporal mean of M and mubar = mean(mu) = mean(m):
//     S (M - m 1^T) = S [(M - 1 mu^T) - (m - mubar 1) 1^T] + u (mu - mubar 1)^T = D + u mu'^T
// -- the same "small matrix + rank-one offset" shape as scaling None, with the vector u in the place of 1: every entry of D is
// formed in float64 from the counts and rounded once, the offset mu' never meets it in float32.
// One 1024-thread workgroup per frame: st[f] = mean of the (masked) frame, st[n + 1 + f] = u[f]; all sums in float64, fixed order.
// (ld: row length of M; the statistics run over its first P samples -- the zero columns that pad a segment matrix stay out)
__global__ __launch_bounds__(1024) void spat_stats_f64_kernel(const double* __restrict__ M, int n, int64_t P, const uint8_t* __restrict__ mask,
                                                              int with_std, double* __restrict__ st, int64_t ld) {
  __shared__ double sh[16];
  const int f = blockIdx.x;
  const double* row = M + (int64_t)f * ld;
  auto total = [&](double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    return t;
  };
  double s = 0.0;
  for (int64_t p = threadIdx.x; p < P; p += 1024) s += (mask && mask[p]) ? 0.0 : row[p];
  const double m = total(s) / (double)P;
  double u = 1.0;
  if (with_std) {
    double v = 0.0;
    for (int64_t p = threadIdx.x; p < P; p += 1024) {
      const double d = ((mask && mask[p]) ? 0.0 : row[p]) - m;
      v += d * d;
    }
    double sd = sqrt(total(v) / (double)P);
    if (sd < 10.0 * 2.220446049250313e-16) sd = 1.0;       // sklearn _handle_zeros_in_scale
    u = 1.0 / sd;
  }
  if (threadIdx.x == 0) {
    st[f] = m;
    st[n + 1 + f] = u;
  }
}
// st[n] = mubar = mean of the frames' means (one workgroup)
__global__ __launch_bounds__(256) void spat_mubar_kernel(int n, double* __restrict__ st) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int f = threadIdx.x; f < n; f += 256) s += st[f];
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) st[n] = s / n;
}
// D[f, p] = float32(((x - mu[p]) - (m[f] - mubar)) u[f]), x = the (masked) sample; then mu[p] -= mubar, mu32 = float32(mu).
// One thread per pixel column, like center_f64_kernel (which has filled mu).
__global__ void spat_apply_f64_kernel(const double* __restrict__ M, int n, int64_t P, const uint8_t* __restrict__ mask,
                                      const double* __restrict__ st, float* __restrict__ D, double* __restrict__ mu,
                                      float* __restrict__ mu32, int64_t ld) {
  const double mubar = st[n];
  const double* u = st + n + 1;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const bool dead = mask && mask[p];
    const double base = mu[p] - mubar;                  // (mu is zero at masked pixels)
    int f0 = 0;
    for (; f0 + 8 <= n; f0 += 8) {
      double v8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v8[q] = dead ? 0.0 : M[(int64_t)(f0 + q) * ld + p];
#pragma unroll
      for (int q = 0; q < 8; ++q) D[(int64_t)(f0 + q) * ld + p] = (float)(((v8[q] - base) - st[f0 + q]) * u[f0 + q]);
    }
    for (; f0 < n; ++f0) D[(int64_t)f0 * ld + p] = (float)((((dead ? 0.0 : M[(int64_t)f0 * ld + p]) - base) - st[f0]) * u[f0]);
    mu[p] = base;
    mu32[p] = (float)base;
  }
}

// the same dot products for the column SEGMENTS of one matrix D[n][ld] (annular PCA: every segment its own decomposition):
// g[seg][f] = sum over the segment's columns of D[f, p] mu[p], g[seg][n] = sum mu[p]^2; blockIdx.y = segment, columns
// seg_slice[seg] * klen .. seg_slice[seg + 1] * klen
__global__ __launch_bounds__(256) void offset_dots_seg_kernel(const float* __restrict__ D, const double* __restrict__ mu, int n, int64_t ld,
                                                              const int32_t* __restrict__ seg_slice, int klen, double* __restrict__ g) {
  __shared__ double sh[4];
  const int f = blockIdx.x, seg = blockIdx.y;
  const int64_t p0 = (int64_t)seg_slice[seg] * klen, p1 = (int64_t)seg_slice[seg + 1] * klen;
  double s = 0.0;
  if (f < n) {
    const float* row = D + (int64_t)f * ld;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) s += (double)row[p] * mu[p];
  } else {
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) s += mu[p] * mu[p];
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) g[(size_t)seg * (n + 1) + f] = s;
}

// r = 1 - E^T (E 1) over the components that convert_evecs keeps (eigenvalue > 1e-12 of the largest); row k of the subtraction's
// coefficient matrix Ct[k][nld] = -r (zero padded): the rank-one term r mu^T as one more component.  One workgroup.
// (u != nullptr, the spatial scalings: the offset is u mu^T, so r = u - E^T (E u) and e1 = E u)
__global__ __launch_bounds__(256) void offset_coeff_kernel(const double* __restrict__ evecs, const double* __restrict__ evals, int n,
                                                           int k, float* __restrict__ Ct_row, int nld, float* __restrict__ e1_out,
                                                           const double* __restrict__ u = nullptr) {
  extern __shared__ double e1[];          // [k]
  const double thr = evals[0] * 1e-12;
  for (int c = threadIdx.x; c < k; c += blockDim.x) {
    double s = 0.0;
    if (evals[c] > thr)
      for (int f = 0; f < n; ++f) s += u ? evecs[(int64_t)c * n + f] * u[f] : evecs[(int64_t)c * n + f];
    e1[c] = s;
    e1_out[c] = (float)s;                 // E 1: the offset's share of the principal components (pcs_offset_kernel)
  }
  __syncthreads();
  for (int f = threadIdx.x; f < nld; f += blockDim.x) {
    double r = 0.0;
    if (f < n) {
      double s = 0.0;
      for (int c = 0; c < k; ++c) s += evecs[(int64_t)c * n + f] * e1[c];
      r = (u ? u[f] : 1.0) - s;
    }
    Ct_row[f] = (float)(-r);
  }
}

// pcs[c] = (T[c] + e1[c] mu) / sigma_c   (V = S^-1 E^T M with M = D + 1 mu^T; e1 == nullptr: the scaled modes, V = S^-1 E^T D)
__global__ void pcs_offset_kernel(const float* __restrict__ T, const float* __restrict__ isig, const float* __restrict__ e1,
                                  const float* __restrict__ mu32, int64_t k, int64_t P, float* __restrict__ out) {
  const int64_t total = k * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = e / P, p = e - c * P;
    out[e] = (T[e] + (e1 ? e1[c] * mu32[p] : 0.f)) * isig[c];
  }
}
// recon = (the matrix the decomposition saw) - residuals = D (+ mu) - residuals
// (u: the offset is u[f] mu -- the spatial scalings)
__global__ void recon_offset_kernel(const float* __restrict__ D, const float* __restrict__ mu32, const float* __restrict__ res,
                                    int64_t n, int64_t P, float* __restrict__ out, const double* __restrict__ u = nullptr) {
  const int64_t total = n * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const float off = mu32 ? (u ? (float)u[e / P] : 1.f) * mu32[e % P] : 0.f;
    out[e] = D[e] + off - res[e];
  }
}

}  // namespace

// The two building blocks on their own (the annular front applies them to every segment matrix of a float64 cube):
// D = float32((M - 1 mu^T) / sd), mu (float64) and optionally float32(mu) -- mode 0 / 1: centre, 2: 'temp-standard'
int center_f64(vipmi_ctx* ctx, const double* M, int64_t n, int64_t P, int mode, float* D, double* mu, float* mu32) {
  VIPMI_REQUIRE(M && D && mu && n > 0 && P > 0 && mode >= 0 && mode <= 2, "center_f64: bad arguments");
  StageScope sc(ctx, "scale");
  const int64_t blocks = cdiv(P, 256);
  hipLaunchKernelGGL(center_f64_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, ctx->stream, M, (int)n, P,
                     (const uint8_t*)nullptr, mode, D, mu, mu32);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}
// G (= D D^T, n x n float64) += 1 (D mu)^T + (D mu) 1^T + |mu|^2 1 1^T : the Gram matrix of D + 1 mu^T
// (u: G += u (D mu)^T + (D mu) u^T + |mu|^2 u u^T, the Gram matrix of D + u mu^T -- the spatial scalings; nullptr: ones)
int gram_offset_f64(vipmi_ctx* ctx, const float* D, const double* mu, int64_t n, int64_t P, double* G, const double* u) {
  VIPMI_REQUIRE(D && mu && G && n > 0 && P > 0, "gram_offset_f64: bad arguments");
  StageScope sc(ctx, "gram");
  double* g = nullptr;
  VIPMI_TRY(ws(ctx, "pca64_g", (size_t)n + 1, &g));
  hipLaunchKernelGGL(offset_dots_kernel, dim3((unsigned)n + 1), dim3(256), 0, ctx->stream, D, mu, (int)n, P, g);
  hipLaunchKernelGGL(gram_offset_kernel, dim3((unsigned)cdiv(n * n, 256)), dim3(256), 0, ctx->stream, G, g, (int)n, u);
  VIPMI_CHECK_HIP(hipGetLastError());
  return VIPMI_OK;
}
// The spatial scalings of a float64 matrix M[n][P] whose first Preal columns are samples (the rest zero padding):
// D = float32 of diag(u) [(M - 1 mu^T) - (m - mubar 1) 1^T], mu <- mu - mubar (zero in the padding), mu32 = float32(mu), u[n] = the
// frames' inverse standard deviations (with_std = 0: ones).  The scaled matrix is D + u mu^T.
int spat_center_f64(vipmi_ctx* ctx, const double* M, int64_t n, int64_t P, int64_t Preal, int with_std, float* D, double* mu, float* mu32,
                    double* u) {
  VIPMI_REQUIRE(M && D && mu && mu32 && u && n > 0 && P > 0 && Preal > 0 && Preal <= P, "spat_center_f64: bad arguments");
  StageScope sc(ctx, "scale");
  double* st = nullptr;
  VIPMI_TRY(ws(ctx, "pca64_spat", (size_t)2 * n + 1, &st));
  const int64_t blocks = cdiv(P, 256), blocks_r = cdiv(Preal, 256);
  hipLaunchKernelGGL(center_f64_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, ctx->stream, M, (int)n, P,
                     (const uint8_t*)nullptr, 0, D, mu, mu32);
  hipLaunchKernelGGL(spat_stats_f64_kernel, dim3((unsigned)n), dim3(1024), 0, ctx->stream, M, (int)n, Preal, (const uint8_t*)nullptr,
                     with_std ? 1 : 0, st, P);
  hipLaunchKernelGGL(spat_mubar_kernel, dim3(1), dim3(256), 0, ctx->stream, (int)n, st);
  hipLaunchKernelGGL(spat_apply_f64_kernel, dim3((unsigned)(blocks_r < 65535 ? blocks_r : 65535)), dim3(256), 0, ctx->stream, M, (int)n,
                     Preal, (const uint8_t*)nullptr, st, D, mu, mu32, P);
  VIPMI_CHECK_HIP(hipGetLastError());
  VIPMI_CHECK_HIP(hipMemcpyAsync(u, st + n + 1, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  return VIPMI_OK;
}

// The float64 front of annular PCA for ALL segments (round 6): D_all[n][Ptot] = float32 of the centred (mode 0 / 1) or standardised
// (mode 2) pixel columns pix_all of the float64 cube, mu / mu32 their temporal means, G_all[seg] = the Gram matrix of the segment's
// columns of D_all (ONE ragged int8 product) and, mode 0 (no scaling), of D + 1 mu^T (the offset terms of every segment in float64).
int annular_gram_all_f64(vipmi_ctx* ctx, const double* cube, int64_t n, int64_t P, const int32_t* pix_all, int64_t Ptot, int64_t klen,
                         const int32_t* seg_slice, int64_t nseg, int mode, float* D_all, double* mu, float* mu32, double* G_all) {
  VIPMI_REQUIRE(cube && pix_all && seg_slice && D_all && mu && mu32 && G_all, "annular_gram_all_f64: null pointer");
  VIPMI_REQUIRE(n > 0 && P > 0 && Ptot > 0 && nseg > 0 && nseg <= 65535 && mode >= 0 && mode <= 2, "annular_gram_all_f64: bad arguments");
  {
    StageScope sc(ctx, "scale");
    const int64_t blocks = cdiv(Ptot, 256);
    hipLaunchKernelGGL(center_f64_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, ctx->stream, cube, (int)n, Ptot,
                       (const uint8_t*)nullptr, mode, D_all, mu, mu32, pix_all, P);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  VIPMI_TRY(gram_i8_ragged_f32(ctx, D_all, n, Ptot, klen, seg_slice, nseg, G_all));
  if (mode == 0) {
    StageScope sc(ctx, "gram");
    double* g = nullptr;
    VIPMI_TRY(ws(ctx, "ann64_g", (size_t)nseg * (n + 1), &g));
    hipLaunchKernelGGL(offset_dots_seg_kernel, dim3((unsigned)n + 1, (unsigned)nseg), dim3(256), 0, ctx->stream, D_all, mu, (int)n, Ptot,
                       seg_slice, (int)klen, g);
    hipLaunchKernelGGL(gram_offset_kernel, dim3((unsigned)cdiv(n * n, 256), (unsigned)nseg), dim3(256), 0, ctx->stream, G_all, g, (int)n);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  return VIPMI_OK;
}

int pca_fullframe_f64(vipmi_ctx* ctx, const double* cube, const double* angles_host, int64_t n, int64_t N, int64_t ncomp, int scaling,
                      const uint8_t* mask, int collapse_mode, float* frame, float* pcs, float* recon, float* residuals,
                      float* residuals_der) {
  VIPMI_REQUIRE(cube && angles_host && frame, "pca_fullframe_f64: null pointer");
  VIPMI_REQUIRE(n > 0 && N > 1, "pca_fullframe_f64: bad sizes");
  VIPMI_REQUIRE(ncomp > 0, "Number of PCs too low. It should be > 0.");
  if (scaling < 0 || scaling > 4) {
    set_error("pca_fullframe_f64: unknown scaling mode %d", scaling);
    return VIPMI_ERR_UNSUPPORTED;
  }
  const bool spat = scaling == VIPMI_SCALE_SPAT_MEAN || scaling == VIPMI_SCALE_SPAT_STANDARD;
  const int64_t P = N * N;
  const int64_t k = ncomp > n ? n : ncomp;          // pca_fullfr.py:876-881 (clamp, not an error)
  VIPMI_REQUIRE(k <= P, "%ld PCs cannot be obtained from a matrix with size [%ld,%ld].", (long)k, (long)n, (long)P);
  const bool offset = scaling == 0 || spat;         // the matrix of the decomposition is D + u mu^T (u = 1 without scaling)
  const int64_t kk = offset ? k + 1 : k;            // components of the subtraction: the k PCs (+ the rank-one offset term)
  float *D = nullptr, *T = nullptr;
  double *mu = nullptr, *G = nullptr, *evals = nullptr, *evecs = nullptr, *g = nullptr, *spat_st = nullptr;
  const double* u = nullptr;                        // the offset's frame vector (spatial scalings; nullptr = ones)
  VIPMI_TRY(ws(ctx, "pca64_D", (size_t)n * P, &D));
  VIPMI_TRY(ws(ctx, "pca64_mu", (size_t)P, &mu));
  VIPMI_TRY(ws(ctx, "pca64_T", (size_t)kk * P, &T));
  VIPMI_TRY(ws(ctx, "pca_G", (size_t)n * n, &G));
  VIPMI_TRY(ws(ctx, "pca_evals", (size_t)n, &evals));
  VIPMI_TRY(ws(ctx, "pca_evecs", (size_t)n * n, &evecs));
  {
    StageScope sc(ctx, "scale");
    const int64_t blocks = cdiv(P, 256);
    hipLaunchKernelGGL(center_f64_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, ctx->stream, cube, (int)n, P, mask,
                       spat ? 0 : scaling, D, mu, offset ? T + (size_t)k * P : (float*)nullptr);
    if (spat) {
      VIPMI_TRY(ws(ctx, "pca64_spat", (size_t)2 * n + 1, &spat_st));
      hipLaunchKernelGGL(spat_stats_f64_kernel, dim3((unsigned)n), dim3(1024), 0, ctx->stream, cube, (int)n, P, mask,
                         scaling == VIPMI_SCALE_SPAT_STANDARD ? 1 : 0, spat_st, P);
      hipLaunchKernelGGL(spat_mubar_kernel, dim3(1), dim3(256), 0, ctx->stream, (int)n, spat_st);
      hipLaunchKernelGGL(spat_apply_f64_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, ctx->stream, cube, (int)n, P,
                         mask, spat_st, D, mu, T + (size_t)k * P, P);
      u = spat_st + n + 1;
    }
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  VIPMI_TRY(gram_f32(ctx, D, n, D, n, P, P, G));
  if (offset) {
    StageScope sc(ctx, "gram");
    VIPMI_TRY(ws(ctx, "pca64_g", (size_t)n + 1, &g));
    hipLaunchKernelGGL(offset_dots_kernel, dim3((unsigned)n + 1), dim3(256), 0, ctx->stream, D, mu, (int)n, P, g);
    hipLaunchKernelGGL(gram_offset_kernel, dim3((unsigned)cdiv(n * n, 256)), dim3(256), 0, ctx->stream, G, g, (int)n, u);
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  float* res = residuals;
  if (!res) VIPMI_TRY(ws(ctx, "pca_res", (size_t)n * P, &res));
  ctx->gate_armed = ctx->gate != nullptr;
  VIPMI_TRY(eigh_leading(ctx, G, 1, n, k, nullptr, evals, evecs, false));
  VIPMI_TRY(ctx->gate_enter());
  const int nld = (int)cdiv(n, 32) * 32, kld = (int)cdiv(k, 32) * 32;
  float *Ekn = nullptr, *Enk = nullptr, *isig = nullptr;
  VIPMI_TRY(ws(ctx, "pca64_Ekn", (size_t)(cdiv(kk, 32) * 32) * nld, &Ekn));
  VIPMI_TRY(ws(ctx, "pca_Enk", (size_t)nld * kld, &Enk));
  VIPMI_TRY(ws(ctx, "pca_isig", (size_t)kld, &isig));
  VIPMI_TRY(convert_evecs(ctx, evecs, evals, n, k, Ekn, Enk, isig));
  {
    StageScope sc(ctx, "project");
    VIPMI_TRY(rowspace_gemm_t(ctx, Enk, kld, D, k, n, P, nullptr, T));           // T[c] = E[c] D, c < k  (row k: mu, from the centring)
    float* e1 = nullptr;
    if (offset) {
      VIPMI_TRY(ws(ctx, "pca64_e1", (size_t)k, &e1));
      hipLaunchKernelGGL(offset_coeff_kernel, dim3(1), dim3(256), sizeof(double) * (size_t)k, ctx->stream, evecs, evals, (int)n, (int)k,
                         Ekn + (size_t)k * nld, nld, e1, u);
      VIPMI_CHECK_HIP(hipGetLastError());
    }
    VIPMI_TRY(subtract_gemm_t(ctx, D, Ekn, nld, T, n, kk, P, res, nullptr));
    const float* mu32 = offset ? T + (size_t)k * P : nullptr;
    if (pcs) {
      const int64_t blocks = cdiv(k * P, 2048);
      hipLaunchKernelGGL(pcs_offset_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, ctx->stream, T, isig, e1, mu32, k,
                         P, pcs);
    }
    if (recon) {
      const int64_t blocks = cdiv(n * P, 2048);
      hipLaunchKernelGGL(recon_offset_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, ctx->stream, D, mu32, res, n, P,
                         recon, u);
    }
    VIPMI_CHECK_HIP(hipGetLastError());
  }
  float* der = residuals_der;
  if (!der) VIPMI_TRY(ws(ctx, "pca_der", (size_t)n * P, &der));
  // pca(): mask_center_px without rot_options -> mask_val=0 (pca_fullfr.py:412-415)
  VIPMI_TRY(derotate_f32(ctx, res, angles_host, n, N, der, mask ? 0 : 1, mask ? 1 : 0, VIPMI_ROT_AUTO));
  VIPMI_TRY(collapse_f32(ctx, der, n, P, collapse_mode, nullptr, 50, frame));
  VIPMI_TRY(ctx->gate_leave());
  if (mask) {
    if (residuals_der) VIPMI_TRY(apply_mask_f32(ctx, der, der, n, P, mask, 0.f));
    VIPMI_TRY(apply_mask_f32(ctx, frame, frame, 1, P, mask, 0.f));
  }
  return VIPMI_OK;
}

}  // namespace vipmi
